// CPU laboratory for the TOPOLOGY of the acceleration structure (no GPU needed): rebuilds what csrc/device/bvh_build.hip + bvh8.hip
// build (Morton-63 order, PLOC with a +-16 window, greedy 8-wide collapse, octant-ordered slots, 8-bit quantised child boxes),
// variants of it, and counts what the per-lane walk of pt_kernels.hip would do -- node visits and triangle tests per ray -- on
// seeded incoherent rays (cosine-distributed bounce rays leaving random surface points).  It decides which builder changes are
// worth a GPU run; the numbers the design quotes are the device's own counters (MiPtStats).
//
//   g++ -O2 -std=c++17 -fopenmp -Itests/host_shim -Ivk_gltf_renderer_amd/csrc/device -o /tmp/lab/bvh_lab tools/lab/bvh_lab.cpp tests/host_shim/reinsert_on_host.cpp
//   /tmp/lab/bvh_lab /tmp/lab/atrium.bin [options]      (input: tools/lab/dump_tris.py)
// options: builder=ploc|sah|lbvh  radius=16  leaf=2  collapse=greedy|sahdp  reinsert=N  preinsert=N (the device's parallel passes; dump2=<prefix> writes the records before and after for tools/test_reinsert_gpu.hip)  rays=200000  order=octant|dist  split=F
#include <algorithm>
#include <cassert>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <numeric>
#include <random>
#include <string>
#include <vector>

extern "C" long long dev_reinsert(float* nodes, int numInner, int root, int passes, int rounds, int threads, int* movesPerPass, int* wantedPerPass);  // tests/host_shim/reinsert_on_host.cpp

struct V3
{
  float x, y, z;
  float operator[](int i) const { return (&x)[i]; }
  float& operator[](int i) { return (&x)[i]; }
};
static V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static V3 normalize(V3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }

struct Box
{
  V3 lo{FLT_MAX, FLT_MAX, FLT_MAX}, hi{-FLT_MAX, -FLT_MAX, -FLT_MAX};
  void grow(V3 p) { for(int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
  void grow(const Box& b) { for(int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
  float area() const { float ex = hi.x - lo.x, ey = hi.y - lo.y, ez = hi.z - lo.z; return ex * ey + ey * ez + ez * ex; }
  bool valid() const { return lo.x <= hi.x; }
};
static Box unite(const Box& a, const Box& b) { Box r = a; r.grow(b); return r; }

struct Tri { V3 p0, p1, p2; };

// ---- BVH2: nodes[i] = {child refs (>= 0 inner, < 0 leaf ~ref index), boxes of both children, count} --------------------------------
struct Node2
{
  int  c[2];
  Box  b[2];
  int  cnt;
  int  parent = -1;
};
struct Bvh2
{
  std::vector<Node2> nodes;
  int                root = -1;
  std::vector<int>   refTri;  // leaf ref -> triangle index (a triangle may have several refs after splitting)
  std::vector<Box>   refBox;
};

static uint64_t expandBits21(uint32_t v)
{
  uint64_t x = v & 0x1fffffu;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

static std::vector<int> mortonOrder(const std::vector<Box>& boxes)
{
  Box cb;
  for(const Box& b : boxes) cb.grow((b.lo + b.hi) * 0.5f);
  std::vector<std::pair<uint64_t, int>> keys(boxes.size());
  for(size_t i = 0; i < boxes.size(); ++i)
  {
    V3       c = (boxes[i].lo + boxes[i].hi) * 0.5f;
    uint32_t q[3];
    for(int a = 0; a < 3; ++a)
    {
      float ext = cb.hi[a] - cb.lo[a];
      float n   = ext > 0 ? (c[a] - cb.lo[a]) / ext : 0.0f;
      q[a]      = uint32_t(std::min(std::max(n * 2097152.0f, 0.0f), 2097151.0f));
    }
    keys[i] = {(expandBits21(q[0]) << 2) | (expandBits21(q[1]) << 1) | expandBits21(q[2]), int(i)};
  }
  std::stable_sort(keys.begin(), keys.end());
  std::vector<int> order(boxes.size());
  for(size_t i = 0; i < keys.size(); ++i) order[i] = keys[i].second;
  return order;
}

// PLOC as bvh_build.hip does it (mutual nearest neighbours in a +-radius window over the Morton order).
static Bvh2 buildPloc(const std::vector<int>& refTri, const std::vector<Box>& refBox, int radius)
{
  Bvh2 B;
  B.refTri = refTri;
  B.refBox = refBox;
  const int n = int(refBox.size());
  std::vector<int> order = mortonOrder(refBox);
  std::vector<int> cid(n), cnt(n, 1);
  std::vector<Box> cb(n);
  for(int i = 0; i < n; ++i) { cid[i] = ~order[i]; cb[i] = refBox[order[i]]; }
  B.nodes.reserve(n);
  int m = n;
  std::vector<int> nn(n), cid2(n), cnt2(n);
  std::vector<Box> cb2(n);
  while(m > 1)
  {
#pragma omp parallel for schedule(static)
    for(int i = 0; i < m; ++i)
    {
      float best = FLT_MAX;
      int   bj   = -1;
      for(int d = -radius; d <= radius; ++d)
      {
        int g = i + d;
        if(d == 0 || g < 0 || g >= m) continue;
        float a = unite(cb[i], cb[g]).area();
        if(a < best) { best = a; bj = g; }
      }
      nn[i] = bj;
    }
    int p = 0;
    for(int i = 0; i < m; ++i)
    {
      int  j      = nn[i];
      bool mutual = j >= 0 && nn[j] == i;
      if(mutual && i > j) continue;
      if(mutual)
      {
        Node2 N;
        N.c[0] = cid[i]; N.c[1] = cid[j]; N.b[0] = cb[i]; N.b[1] = cb[j]; N.cnt = cnt[i] + cnt[j];
        int k = int(B.nodes.size());
        B.nodes.push_back(N);
        cid2[p] = k; cb2[p] = unite(cb[i], cb[j]); cnt2[p] = N.cnt;
      }
      else { cid2[p] = cid[i]; cb2[p] = cb[i]; cnt2[p] = cnt[i]; }
      ++p;
    }
    m = p;
    std::swap(cid, cid2); std::swap(cb, cb2); std::swap(cnt, cnt2);
  }
  B.root = cid[0];
  for(int i = 0; i < int(B.nodes.size()); ++i)
    for(int k = 0; k < 2; ++k)
      if(B.nodes[i].c[k] >= 0) B.nodes[B.nodes[i].c[k]].parent = i;
  return B;
}

// Top-down binned SAH (32 bins), leaves of one reference: the quality yardstick.
static Bvh2 buildSah(const std::vector<int>& refTri, const std::vector<Box>& refBox)
{
  Bvh2 B;
  B.refTri = refTri;
  B.refBox = refBox;
  const int n = int(refBox.size());
  std::vector<int> idx(n);
  std::iota(idx.begin(), idx.end(), 0);
  B.nodes.reserve(n);
  std::function<int(int, int, Box&)> rec = [&](int lo, int hi, Box& outBox) -> int {
    Box bb, cb;
    for(int i = lo; i < hi; ++i) { bb.grow(refBox[idx[i]]); cb.grow((refBox[idx[i]].lo + refBox[idx[i]].hi) * 0.5f); }
    outBox = bb;
    if(hi - lo == 1) return ~idx[lo];
    constexpr int NB = 32;
    float bestCost = FLT_MAX; int bestAxis = -1, bestBin = -1;
    for(int a = 0; a < 3; ++a)
    {
      float ext = cb.hi[a] - cb.lo[a];
      if(!(ext > 0)) continue;
      Box bins[NB]; int cnts[NB] = {};
      for(int i = lo; i < hi; ++i)
      {
        const Box& b = refBox[idx[i]];
        int k = std::min(NB - 1, int(((b.lo[a] + b.hi[a]) * 0.5f - cb.lo[a]) / ext * NB));
        bins[k].grow(b); cnts[k]++;
      }
      float rightA[NB]; int rightC[NB]; Box acc; int c = 0;
      for(int k = NB - 1; k > 0; --k) { if(cnts[k]) acc.grow(bins[k]); c += cnts[k]; rightA[k] = c ? acc.area() : 0; rightC[k] = c; }
      acc = Box(); c = 0;
      for(int k = 0; k < NB - 1; ++k)
      {
        if(cnts[k]) acc.grow(bins[k]); c += cnts[k];
        if(c == 0 || rightC[k + 1] == 0) continue;
        float cost = acc.area() * c + rightA[k + 1] * rightC[k + 1];
        if(cost < bestCost) { bestCost = cost; bestAxis = a; bestBin = k; }
      }
    }
    int mid;
    if(bestAxis < 0) mid = (lo + hi) / 2;
    else
    {
      float ext = cb.hi[bestAxis] - cb.lo[bestAxis];
      mid = int(std::partition(idx.begin() + lo, idx.begin() + hi, [&](int t) {
              const Box& b = refBox[t];
              int k = std::min(NB - 1, int(((b.lo[bestAxis] + b.hi[bestAxis]) * 0.5f - cb.lo[bestAxis]) / ext * NB));
              return k <= bestBin; }) - idx.begin());
      if(mid == lo || mid == hi) mid = (lo + hi) / 2;
    }
    int me = int(B.nodes.size());
    B.nodes.push_back(Node2());
    Box b0, b1;
    int c0 = rec(lo, mid, b0), c1 = rec(mid, hi, b1);
    Node2& N = B.nodes[me];
    N.c[0] = c0; N.c[1] = c1; N.b[0] = b0; N.b[1] = b1; N.cnt = hi - lo;
    if(c0 >= 0) B.nodes[c0].parent = me;
    if(c1 >= 0) B.nodes[c1].parent = me;
    return me;
  };
  Box rb;
  B.root = rec(0, n, rb);
  return B;
}

static Box nodeBox(const Bvh2& B, int ref) { return ref < 0 ? B.refBox[~ref] : unite(B.nodes[ref].b[0], B.nodes[ref].b[1]); }
static int refCount(const Bvh2& B, int ref) { return ref < 0 ? 1 : B.nodes[ref].cnt; }

static double sahCost2(const Bvh2& B)
{
  double rootA = nodeBox(B, B.root).area(), c = 0;
  for(const Node2& N : B.nodes) c += unite(N.b[0], N.b[1]).area() / rootA * 1.0;
  return c;  // inner-node term only (leaves have one reference each)
}

// ---- reinsertion (Bittner et al. 2013 flavour, simplified): take a node whose removal saves the most area, re-insert its subtree at the
// position that minimises the area increase (branch and bound over the tree).  `passes` sweeps over a fraction of the nodes.
static void refitUp(Bvh2& B, int node)
{
  while(node >= 0)
  {
    Node2& N = B.nodes[node];
    for(int k = 0; k < 2; ++k) N.b[k] = nodeBox(B, N.c[k]);
    N.cnt = refCount(B, N.c[0]) + refCount(B, N.c[1]);
    node  = N.parent;
  }
}
static void reinsertPass(Bvh2& B, double fraction, std::vector<int>& leafParent)
{
  const int nn = int(B.nodes.size());
  // candidates: inner nodes (not root, not children of root), by area * (some inefficiency measure); simple: by area
  std::vector<std::pair<float, int>> cand;
  for(int i = 0; i < nn; ++i)
    if(i != B.root && B.nodes[i].parent >= 0 && B.nodes[i].parent != B.root) cand.push_back({unite(B.nodes[i].b[0], B.nodes[i].b[1]).area(), i});
  std::sort(cand.begin(), cand.end(), [](auto& a, auto& b) { return a.first > b.first; });
  int todo = int(cand.size() * fraction);
  for(int t = 0; t < todo; ++t)
  {
    int x = cand[t].second;          // subtree to move
    int p = B.nodes[x].parent;
    if(p < 0 || p == B.root) continue;
    int g = B.nodes[p].parent;
    if(g < 0) continue;
    // detach: sibling replaces p in g; p becomes the free node that will sit above x at the new position
    int sibSlot = B.nodes[p].c[0] == x ? 1 : 0;
    int sib     = B.nodes[p].c[sibSlot];
    int gslot   = B.nodes[g].c[0] == p ? 0 : 1;
    B.nodes[g].c[gslot] = sib;
    if(sib >= 0) B.nodes[sib].parent = g; else leafParent[~sib] = g;
    refitUp(B, g);
    Box xb = nodeBox(B, x);
    // search best position: branch and bound on induced cost
    struct Item { float induced; int ref; int parent; int slot; };
    float bestCost = FLT_MAX; int bestParent = -1, bestSlot = -1;
    std::vector<Item> stack;
    // start at root's two children positions
    {
      stack.push_back({0.0f, B.root, -1, 0});
    }
    float xa = xb.area();
    while(!stack.empty())
    {
      Item it = stack.back(); stack.pop_back();
      if(it.induced + xa >= bestCost) continue;
      Box  nb     = nodeBox(B, it.ref);
      float direct = unite(nb, xb).area();
      float total  = it.induced + direct;
      if(it.parent >= 0 && total < bestCost) { bestCost = total; bestParent = it.parent; bestSlot = it.slot; }
      if(it.ref >= 0)
      {
        float ind = it.induced + direct - nb.area();
        if(ind + xa < bestCost)
          for(int k = 0; k < 2; ++k) stack.push_back({ind, B.nodes[it.ref].c[k], it.ref, k});
      }
    }
    if(bestParent < 0)
    {  // put back where it was
      bestParent = g; bestSlot = gslot;
    }
    // insert p between bestParent and its child at bestSlot
    int old = B.nodes[bestParent].c[bestSlot];
    B.nodes[bestParent].c[bestSlot] = p;
    B.nodes[p].parent = bestParent;
    B.nodes[p].c[0] = old; B.nodes[p].c[1] = x;
    if(old >= 0) B.nodes[old].parent = p; else leafParent[~old] = p;
    B.nodes[x].parent = p;
    refitUp(B, p);
  }
}

// ---- 8-wide node ------------------------------------------------------------------------------------------------------------------
struct Node8
{
  Box      box[8];      // decoded (quantised) child boxes; !valid() = empty slot
  int      child[8];    // inner: index of the Node8; leaf: -1; empty: -2
  int      triBase[8], triCnt[8];
};
struct Bvh8
{
  std::vector<Node8> nodes;
  std::vector<int>   tris;  // triangle indices in leaf order
  double             sah = 0;
};

struct CollapseCfg { int scaleMantissaBits = 0; int maxLeaf = 2; bool sahdp = false; bool quantise = true; float cLeafTri = 0.24f; int width = 8; bool optimalSlots = false; };

// optimal (SAH, dynamic programming) collapse after Ylitie et al. 2017, section 3: cost(n, i) = cheapest way to represent subtree n as
// at most i roots of 8-wide (sub)trees; cLeaf per triangle and 1 per inner node, weighted by area.
struct Dp
{
  const Bvh2& B;
  int         maxLeaf;
  float       cTri;
  std::vector<float> cost;   // [node][i], i = 1..7 -> index node*8+i  (i = 1: the subtree as ONE child: either a leaf or an inner node)
  std::vector<int8_t> split; // for i >= 2: how many roots go to the left child; for i == 1: 0 = leaf, 1 = inner node
  std::vector<float> areaN;
  Dp(const Bvh2& b, int ml, float ct) : B(b), maxLeaf(ml), cTri(ct)
  {
    const int n = int(B.nodes.size());
    cost.assign(size_t(n) * 8, FLT_MAX); split.assign(size_t(n) * 8, 0); areaN.resize(n);
    // post-order
    std::vector<int> order; order.reserve(n);
    std::vector<int> st{B.root};
    while(!st.empty()) { int r = st.back(); st.pop_back(); if(r < 0) continue; order.push_back(r); st.push_back(B.nodes[r].c[0]); st.push_back(B.nodes[r].c[1]); }
    for(int k = n - 1; k >= 0; --k) solve(order[k]);
  }
  float leafCost(int ref) const { return nodeBox(B, ref).area() * cTri * refCount(B, ref); }
  float get(int ref, int i) const
  {
    if(ref < 0) return leafCost(ref);  // a single reference is one leaf child whatever i
    return cost[size_t(ref) * 8 + std::min(i, 7)];
  }
  void solve(int nd)
  {
    const Node2& N = B.nodes[nd];
    float* c = &cost[size_t(nd) * 8];
    int8_t* s = &split[size_t(nd) * 8];
    // distribute i roots over the two children
    for(int i = 2; i <= 7; ++i)
    {
      float best = FLT_MAX; int bk = 1;
      for(int k = 1; k < i; ++k)
      {
        float v = get(N.c[0], k) + get(N.c[1], i - k);
        if(v < best) { best = v; bk = k; }
      }
      c[i] = best; s[i] = int8_t(bk);
    }
    // as ONE child: a leaf (if small enough) or an inner 8-wide node whose children are the best 8-root forest... 8 = k + (8-k)
    float inner = FLT_MAX; int bk = 1;
    for(int k = 1; k < 8; ++k)
    {
      float v = get(N.c[0], k) + get(N.c[1], 8 - k);
      if(v < inner) { inner = v; bk = k; }
    }
    inner += areaN[nd] = unite(N.b[0], N.b[1]).area();
    float leaf = N.cnt <= maxLeaf ? leafCost(nd) : FLT_MAX;
    if(leaf <= inner) { c[1] = leaf; s[1] = 0; }
    else { c[1] = inner; s[1] = 1; }
    s[0] = int8_t(bk);  // the 8-split of the inner-node option
    for(int i = 2; i <= 7; ++i)
      if(c[1] < c[i]) { c[i] = c[1]; s[i] = -1; }  // fewer roots than allowed is fine too
  }
  // the roots of the forest that represents `ref` with at most i roots
  void roots(int ref, int i, std::vector<int>& out) const
  {
    if(ref < 0) { out.push_back(ref); return; }
    if(i == 1 || split[size_t(ref) * 8 + i] == -1) { out.push_back(ref); return; }
    int k = split[size_t(ref) * 8 + i];
    roots(B.nodes[ref].c[0], k, out);
    roots(B.nodes[ref].c[1], i - k, out);
  }
  bool isLeafRoot(int ref) const { return ref < 0 || split[size_t(ref) * 8 + 1] == 0; }
  void children(int ref, std::vector<int>& out) const
  {
    int k = split[size_t(ref) * 8 + 0];
    roots(B.nodes[ref].c[0], k, out);
    roots(B.nodes[ref].c[1], 8 - k, out);
  }
};

static void collectRefs(const Bvh2& B, int ref, std::vector<int>& out)
{
  if(ref < 0) { out.push_back(~ref); return; }
  collectRefs(B, B.nodes[ref].c[0], out);
  collectRefs(B, B.nodes[ref].c[1], out);
}

static Bvh8 collapse(const Bvh2& B, const CollapseCfg& cfg)
{
  Bvh8 W;
  std::unique_ptr<Dp> dp;
  if(cfg.sahdp) dp.reset(new Dp(B, cfg.maxLeaf, cfg.cLeafTri));
  struct Item { int ref; int node8; };
  std::vector<int> level{B.root};
  W.nodes.push_back(Node8());
  std::vector<int> levelNode{0};
  const double rootA = nodeBox(B, B.root).area();
  while(!level.empty())
  {
    std::vector<int> next, nextNode;
    for(size_t li = 0; li < level.size(); ++li)
    {
      int ref = level[li];
      std::vector<int> kids;
      if(cfg.sahdp) dp->children(ref, kids);
      else
      {
        kids = {B.nodes[ref].c[0], B.nodes[ref].c[1]};
        while(int(kids.size()) < cfg.width)
        {
          int best = -1; float bestA = -1;
          for(size_t k = 0; k < kids.size(); ++k)
            if(kids[k] >= 0 && refCount(B, kids[k]) > cfg.maxLeaf)
            {
              float a = nodeBox(B, kids[k]).area();
              if(a > bestA) { bestA = a; best = int(k); }
            }
          if(best < 0) break;
          int r = kids[best];
          kids[best] = B.nodes[r].c[0];
          kids.push_back(B.nodes[r].c[1]);
        }
      }
      const int count = int(kids.size());
      std::vector<Box> kb(count);
      Box nb;
      for(int k = 0; k < count; ++k) { kb[k] = nodeBox(B, kids[k]); nb.grow(kb[k]); }
      W.sah += nb.area() / rootA;
      // octant slot assignment, as k_collapse_emit
      int candOfSlot[8]; bool slotUsed[8] = {}, done[8] = {};
      std::fill(candOfSlot, candOfSlot + 8, -1);
      for(int round = 0; round < count; ++round)
      {
        float bestCost = -FLT_MAX; int bc = -1, bs = -1;
        for(int c = 0; c < count; ++c)
        {
          if(done[c]) continue;
          float d[3];
          for(int a = 0; a < 3; ++a) d[a] = 0.5f * (kb[c].lo[a] + kb[c].hi[a]) - 0.5f * (nb.lo[a] + nb.hi[a]);
          for(int sl = 0; sl < 8; ++sl)
          {
            if(slotUsed[sl]) continue;
            float cost = ((sl & 1) ? d[0] : -d[0]) + ((sl & 2) ? d[1] : -d[1]) + ((sl & 4) ? d[2] : -d[2]);
            if(cost > bestCost) { bestCost = cost; bc = c; bs = sl; }
          }
        }
        candOfSlot[bs] = bc; slotUsed[bs] = true; done[bc] = true;
      }
      if(cfg.optimalSlots)
      {
        // maximise the sum of dot(centroid - centre, diagonal(slot)) over all assignments: DP over subsets of slots
        float cost[8][8];
        for(int c = 0; c < count; ++c)
        {
          float d[3];
          for(int a = 0; a < 3; ++a) d[a] = 0.5f * (kb[c].lo[a] + kb[c].hi[a]) - 0.5f * (nb.lo[a] + nb.hi[a]);
          for(int sl = 0; sl < 8; ++sl) cost[c][sl] = ((sl & 1) ? d[0] : -d[0]) + ((sl & 2) ? d[1] : -d[1]) + ((sl & 4) ? d[2] : -d[2]);
        }
        std::vector<float> best(256, -FLT_MAX); std::vector<int> from(256, -1);
        best[0] = 0;
        for(int mask = 0; mask < 256; ++mask)
        {
          if(best[mask] == -FLT_MAX) continue;
          int c = __builtin_popcount(mask);
          if(c >= count) continue;
          for(int sl = 0; sl < 8; ++sl)
            if(!(mask & (1 << sl)) && best[mask] + cost[c][sl] > best[mask | (1 << sl)]) { best[mask | (1 << sl)] = best[mask] + cost[c][sl]; from[mask | (1 << sl)] = sl; }
        }
        int bm = -1; float bv = -FLT_MAX;
        for(int mask = 0; mask < 256; ++mask) if(__builtin_popcount(mask) == count && best[mask] > bv) { bv = best[mask]; bm = mask; }
        std::fill(candOfSlot, candOfSlot + 8, -1);
        for(int c = count - 1, mask = bm; c >= 0; --c) { int sl = from[mask]; candOfSlot[sl] = c; mask &= ~(1 << sl); }
      }
      Node8 N;
      // quantisation frame
      float scale[3];
      for(int a = 0; a < 3; ++a)
      {
        float ext = nb.hi[a] - nb.lo[a];
        int   e   = ext > 0 ? int(std::ceil(std::log2(double(ext) / 255.0))) : -126;
        scale[a]  = std::ldexp(1.0f, e);
        if(cfg.scaleMantissaBits > 0 && ext > 0)  // scale = m * 2^e with an m of that many bits (device: a bf16-like scale per axis instead of an exponent byte)
        {
          const double want = double(ext) / 255.0;
          int          e2   = int(std::floor(std::log2(want)));
          const double unit = std::ldexp(1.0, e2 - cfg.scaleMantissaBits);
          scale[a]          = float(std::ceil(want / unit) * unit);
        }
      }
      for(int sl = 0; sl < 8; ++sl)
      {
        N.child[sl] = -2; N.triBase[sl] = 0; N.triCnt[sl] = 0; N.box[sl] = Box();
        if(candOfSlot[sl] < 0) continue;
        int        k = candOfSlot[sl];
        Box        q = kb[k];
        if(cfg.quantise)
          for(int a = 0; a < 3; ++a)
          {
            q.lo[a] = nb.lo[a] + std::floor((kb[k].lo[a] - nb.lo[a]) / scale[a]) * scale[a];
            q.hi[a] = nb.lo[a] + std::ceil((kb[k].hi[a] - nb.lo[a]) / scale[a]) * scale[a];
          }
        N.box[sl] = q;
        bool leaf = cfg.sahdp ? dp->isLeafRoot(kids[k]) : (kids[k] < 0 || refCount(B, kids[k]) <= cfg.maxLeaf);
        if(leaf)
        {
          std::vector<int> refs;
          collectRefs(B, kids[k], refs);
          N.child[sl] = -1; N.triBase[sl] = int(W.tris.size()); N.triCnt[sl] = int(refs.size());
          for(int r : refs) W.tris.push_back(B.refTri[r]);
        }
        else
        {
          N.child[sl] = int(W.nodes.size() + next.size());  // placeholder, fixed below (BFS order = append order)
          next.push_back(kids[k]);
        }
      }
      W.nodes[levelNode[li]] = N;
    }
    // allocate the next level's nodes and patch indices: children were numbered W.nodes.size() + position in `next` at the time;
    // since nodes are appended only here, recompute
    size_t base = W.nodes.size();
    // fix child indices of this level: they were assigned as (size at that time + index), size was constant during the level => ok
    for(size_t i = 0; i < next.size(); ++i) { W.nodes.push_back(Node8()); nextNode.push_back(int(base + i)); }
    level.swap(next); levelNode.swap(nextNode);
  }
  return W;
}

// ---- rays ----------------------------------------------------------------------------------------------------------------------------
struct Ray { V3 o, d; };
static bool hitTri(const Tri& T, const Ray& r, float tmax, float& t)
{
  V3 e1 = T.p1 - T.p0, e2 = T.p2 - T.p0, p = cross(r.d, e2);
  float det = dot(e1, p);
  if(std::fabs(det) < 1e-20f) return false;
  float inv = 1.0f / det;
  V3 s = r.o - T.p0;
  float u = dot(s, p) * inv;
  if(u < 0 || u > 1) return false;
  V3 q = cross(s, e1);
  float v = dot(r.d, q) * inv;
  if(v < 0 || u + v > 1) return false;
  t = dot(e2, q) * inv;
  return t > 0 && t < tmax;
}
static bool slab(const Box& b, const Ray& r, V3 idir, float tmax, float& tn)
{
  float t0 = 0, t1 = tmax;
  for(int a = 0; a < 3; ++a)
  {
    float ta = (b.lo[a] - r.o[a]) * idir[a], tb = (b.hi[a] - r.o[a]) * idir[a];
    if(ta > tb) std::swap(ta, tb);
    t0 = std::max(t0, ta); t1 = std::min(t1, tb);
  }
  tn = t0;
  return t0 <= t1;
}

struct WalkStats { double nodes = 0, tris = 0, rays = 0, hits = 0, maxStack = 0, deep = 0, maxGroups = 0, coplanar = 0, behind = 0; };  // coplanar: tests of a triangle in whose plane the ray STARTS; behind: whose plane the ray never reaches (origin and direction on the same side)  // deep: visits made with more than 12 GROUPS (a node's pending children: the device's stack entry) waiting

// mode 0: octant order, triangles tested immediately (the ideal of the device walk); 1: distance order; 2: octant order with the
// device's deferral model: leaf hits are parked and tested only after `defer` further node visits (tmax tightens late)
static void walk(const Bvh8& W, const std::vector<Tri>& tris, const std::vector<uint8_t>& alpha, const Ray& r, int mode, int defer, uint32_t seed, WalkStats& S)
{
  V3 idir{1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
  const uint32_t octinv = 7u ^ ((idir.x < 0 ? 1u : 0u) | (idir.y < 0 ? 2u : 0u) | (idir.z < 0 ? 4u : 0u));
  float tmax = FLT_MAX;
  struct Entry { int node; float tn; float gmin; int from; };  // from: the node that pushed it; gmin: smallest entry distance among the hit children of the node that pushed this entry
  std::vector<Entry> stack;
  stack.push_back({0, 0, 0, -1});
  struct Parked { int base, cnt, due; };
  std::vector<Parked> parked;
  int visits = 0;
  auto testLeaf = [&](int base, int cnt) {
    for(int k = 0; k < cnt; ++k)
    {
      int   ti = W.tris[base + k];
      float t;
      S.tris += 1;
      {
        const Tri& T  = tris[ti];
        const V3   n  = cross(T.p1 - T.p0, T.p2 - T.p0);
        const float nl = std::sqrt(dot(n, n)) + 1e-30f, dist = dot(n, r.o - T.p0) / nl, dn = dot(n, r.d) / nl;
        if(std::fabs(dist) < 2e-3f) S.coplanar += 1;
        else if(dist * dn >= 0) S.behind += 1;
      }
      if(hitTri(tris[ti], r, tmax, t))
      {
        if(alpha[ti])
        {
          uint32_t h = (seed ^ uint32_t(ti) * 2654435761u); h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
          if((h & 0xffff) < 0x9999) continue;  // 60 % of the candidates on alpha-tested cards are passed through
        }
        tmax = t;
      }
    }
  };
  while(!stack.empty())
  {
    Entry e = stack.back(); stack.pop_back();
    if((mode == 1 || mode == 3) && e.tn > tmax) continue;  // mode 3: octant order, but a child whose entry distance lies beyond the current hit is dropped when popped
    if(mode == 4 && e.gmin > tmax) continue;                 // mode 4: ... only when its whole GROUP lies beyond the hit (one distance per stack entry on the device)
    const Node8& N = W.nodes[e.node];
    S.nodes += 1; ++visits;
    // due parked leaves
    if(mode == 2)
    {
      for(size_t k = 0; k < parked.size();)
        if(parked[k].due <= visits) { testLeaf(parked[k].base, parked[k].cnt); parked[k] = parked.back(); parked.pop_back(); }
        else ++k;
    }
    struct H { int slot; float tn; };
    H hit[8]; int nh = 0;
    for(int sl = 0; sl < 8; ++sl)
    {
      if(N.child[sl] == -2) continue;
      float tn;
      if(!slab(N.box[sl], r, idir, tmax, tn)) continue;
      if(N.child[sl] == -1)
      {
        if(mode == 2) parked.push_back({N.triBase[sl], N.triCnt[sl], visits + defer});
        else testLeaf(N.triBase[sl], N.triCnt[sl]);
      }
      else hit[nh++] = {sl, tn};
    }
    // push in reverse priority so that the nearest is popped first
    if(mode == 1 || mode == 5) std::sort(hit, hit + nh, [](const H& a, const H& b) { return a.tn > b.tn; });  // (mode 5: the order alone, nothing dropped when popped)
    else std::sort(hit, hit + nh, [&](const H& a, const H& b) { return (uint32_t(a.slot) ^ octinv) < (uint32_t(b.slot) ^ octinv); });
    float gmin = FLT_MAX;
    for(int k = 0; k < nh; ++k) gmin = std::min(gmin, hit[k].tn);
    for(int k = 0; k < nh; ++k) stack.push_back({N.child[hit[k].slot], hit[k].tn, gmin, e.node});
    S.maxStack = std::max(S.maxStack, double(stack.size()));
    {
      int groups = 0;
      for(size_t k = 0; k < stack.size(); ++k) groups += (k == 0 || stack[k].from != stack[k - 1].from) ? 1 : 0;
      S.maxGroups = std::max(S.maxGroups, double(groups));
      if(groups > 12) S.deep += 1;
    }
    if(mode == 2 && stack.empty())
    {
      for(auto& p : parked) testLeaf(p.base, p.cnt);
      parked.clear();
      // (a tightened tmax cannot prune anything any more)
    }
  }
  if(mode == 2) for(auto& p : parked) testLeaf(p.base, p.cnt);
  S.rays += 1;
  if(tmax < FLT_MAX) S.hits += 1;
}

// any-hit (shadow) walk: order 0 = octant order (what the device does), 1 = largest child box first, 2 = nearest first, 3 = leaf children first then octant
static void walkAny(const Bvh8& W, const std::vector<Tri>& tris, const std::vector<uint8_t>& alpha, const Ray& r, int order, uint32_t seed, WalkStats& S)
{
  V3 idir{1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
  const uint32_t octinv = 7u ^ ((idir.x < 0 ? 1u : 0u) | (idir.y < 0 ? 2u : 0u) | (idir.z < 0 ? 4u : 0u));
  std::vector<int> stack{0};
  bool occluded = false;
  while(!stack.empty() && !occluded)
  {
    int nd = stack.back(); stack.pop_back();
    const Node8& N = W.nodes[nd];
    S.nodes += 1;
    struct H { int slot; float key; };
    H hit[8]; int nh = 0;
    for(int sl = 0; sl < 8 && !occluded; ++sl)
    {
      if(N.child[sl] == -2) continue;
      float tn;
      if(!slab(N.box[sl], r, idir, FLT_MAX, tn)) continue;
      if(N.child[sl] == -1)
      {
        for(int k = 0; k < N.triCnt[sl] && !occluded; ++k)
        {
          int ti = W.tris[N.triBase[sl] + k]; float t;
          S.tris += 1;
          if(hitTri(tris[ti], r, FLT_MAX, t))
          {
            if(alpha[ti]) { uint32_t h = (seed ^ uint32_t(ti) * 2654435761u); h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; if((h & 0xffff) < 0x9999) continue; }
            occluded = true;
          }
        }
      }
      else hit[nh++] = {sl, order == 1 ? N.box[sl].area() : (order == 2 ? -tn : (order == 3 ? float(sl) : (order == 4 ? -float(uint32_t(sl) ^ octinv) : (order == 5 ? -float(sl) : float(uint32_t(sl) ^ octinv)))))};
    }
    std::sort(hit, hit + nh, [](const H& a, const H& b) { return a.key < b.key; });  // highest key popped first
    for(int k = 0; k < nh; ++k) stack.push_back(N.child[hit[k].slot]);
  }
  S.rays += 1;
  if(occluded) S.hits += 1;
}

int main(int argc, char** argv)
{
  if(argc < 2) { fprintf(stderr, "usage: bvh_lab tris.bin [key=value ...]\n"); return 1; }
  std::map<std::string, std::string> opt;
  for(int i = 2; i < argc; ++i) { std::string a = argv[i]; size_t e = a.find('='); if(e != std::string::npos) opt[a.substr(0, e)] = a.substr(e + 1); }
  auto get = [&](const char* k, const char* d) { return opt.count(k) ? opt[k] : std::string(d); };
  FILE* f = fopen(argv[1], "rb");
  if(!f) { perror("open"); return 1; }
  int n = 0;
  if(fread(&n, 4, 1, f) != 1) return 1;
  std::vector<Tri> tris(n);
  std::vector<uint8_t> alpha(n);
  if(fread(tris.data(), sizeof(Tri), n, f) != size_t(n) || fread(alpha.data(), 1, n, f) != size_t(n)) return 1;
  float cam[6] = {};
  if(fread(cam, 4, 6, f) != 6) return 1;
  fclose(f);

  // references (optionally pre-split: a triangle whose box area exceeds split x mean is cut along its box's longest axis, recursively)
  std::vector<int> refTri; std::vector<Box> refBox;
  const float splitF = std::stof(get("split", "0"));
  const bool  splitGrid = get("splitmode", "mid") == "grid";
  const int   splitDepth = std::stoi(get("splitdepth", "6"));
  Box sceneB; for(int i = 0; i < n; ++i) { sceneB.grow(tris[i].p0); sceneB.grow(tris[i].p1); sceneB.grow(tris[i].p2); }
  {
    double meanA = 0;
    std::vector<Box> tb(n);
    for(int i = 0; i < n; ++i) { tb[i].grow(tris[i].p0); tb[i].grow(tris[i].p1); tb[i].grow(tris[i].p2); meanA += tb[i].area(); }
    meanA /= n;
    for(int i = 0; i < n; ++i)
    {
      if(splitF <= 0 || tb[i].area() <= splitF * meanA) { refTri.push_back(i); refBox.push_back(tb[i]); continue; }
      // clip polygon against box halves recursively
      std::function<void(std::vector<V3>, int)> rec = [&](std::vector<V3> poly, int depth) {
        Box b; for(V3 p : poly) b.grow(p);
        if(depth >= splitDepth || b.area() <= splitF * meanA) { refTri.push_back(i); refBox.push_back(b); return; }
        int ax = 0; float ext = 0;
        for(int a = 0; a < 3; ++a) if(b.hi[a] - b.lo[a] > ext) { ext = b.hi[a] - b.lo[a]; ax = a; }
        float mid = 0.5f * (b.lo[ax] + b.hi[ax]);
        if(splitGrid)  // the coarsest plane of the scene's recursive bisection that crosses the box (Karras & Aila 2013): splits of neighbouring triangles coincide
        {
          const float o = sceneB.lo[ax], w = sceneB.hi[ax] - sceneB.lo[ax];
          double lo = (b.lo[ax] - o) / w, hi = (b.hi[ax] - o) / w;
          for(int lvl = 1; lvl <= 24; ++lvl)
          {
            const double cell = std::ldexp(1.0, -lvl);
            const double k = std::ceil(lo / cell);
            if(k * cell < hi && k * cell > lo)
            {
              // several planes of this level may cross: take the one nearest the middle
              const double km = std::round(0.5 * (lo + hi) / cell);
              const double kk = (km * cell > lo && km * cell < hi) ? km : k;
              mid = float(o + kk * cell * w);
              break;
            }
          }
          if(!(mid > b.lo[ax] && mid < b.hi[ax])) mid = 0.5f * (b.lo[ax] + b.hi[ax]);
        }
        std::vector<V3> L, R;
        for(size_t k = 0; k < poly.size(); ++k)
        {
          V3 a = poly[k], c = poly[(k + 1) % poly.size()];
          bool ia = a[ax] <= mid, ic = c[ax] <= mid;
          if(ia) L.push_back(a); if(!ia || a[ax] == mid) R.push_back(a);
          if(ia != ic) { float t = (mid - a[ax]) / (c[ax] - a[ax]); V3 p = a + (c - a) * t; p[ax] = mid; L.push_back(p); R.push_back(p); }
        }
        if(L.size() >= 3) rec(L, depth + 1);
        if(R.size() >= 3) rec(R, depth + 1);
      };
      rec({tris[i].p0, tris[i].p1, tris[i].p2}, 0);
    }
  }
  printf("triangles %d references %zu\n", n, refTri.size());

  const std::string builder = get("builder", "ploc");
  Bvh2 B = builder == "sah" ? buildSah(refTri, refBox) : buildPloc(refTri, refBox, std::stoi(get("radius", "16")));
  printf("BVH2 %s: inner-node SAH %.2f\n", builder.c_str(), sahCost2(B));
  const int passes = std::stoi(get("reinsert", "0"));
  if(passes > 0)
  {
    std::vector<int> leafParent(refTri.size(), -1);
    for(int i = 0; i < int(B.nodes.size()); ++i) for(int k = 0; k < 2; ++k) if(B.nodes[i].c[k] < 0) leafParent[~B.nodes[i].c[k]] = i;
    for(int p = 0; p < passes; ++p)
    {
      reinsertPass(B, std::stod(get("fraction", "0.02")), leafParent);
      printf("  reinsertion pass %d: SAH %.2f\n", p + 1, sahCost2(B));
    }
  }
  const int ppasses = std::stoi(get("preinsert", "0"));  // the DEVICE's parallel reinsertion (csrc/device/bvh_reinsert.h through tests/host_shim)
  if(ppasses > 0)
  {
    const int ni = int(B.nodes.size());
    std::vector<float> rec(size_t(ni) * 16);
    for(int i = 0; i < ni; ++i)
    {
      const Node2& N = B.nodes[i];
      float* f = &rec[size_t(i) * 16];
      for(int k = 0; k < 2; ++k)
      {
        f[4 * k + 0] = N.b[k].lo.x; f[4 * k + 1] = N.b[k].hi.x; f[4 * k + 2] = N.b[k].lo.y; f[4 * k + 3] = N.b[k].hi.y;
        f[8 + 2 * k] = N.b[k].lo.z; f[9 + 2 * k] = N.b[k].hi.z;
        memcpy(&f[12 + k], &N.c[k], 4);
      }
      memcpy(&f[14], &N.cnt, 4);
    }
    auto dump = [&](const std::string& path) {  // the records as the device holds them: int32 numInner, int32 root, numInner x 16 floats (tools/test_reinsert_gpu.hip)
      FILE* o = fopen(path.c_str(), "wb");
      if(!o) { perror("dump"); return; }
      fwrite(&ni, 4, 1, o); fwrite(&B.root, 4, 1, o); fwrite(rec.data(), 4, rec.size(), o);
      fclose(o);
    };
    if(opt.count("dump2")) dump(opt["dump2"] + ".in");
    std::vector<int> done(ppasses, 0), wanted(ppasses, 0);
    const long long total = dev_reinsert(rec.data(), ni, B.root, ppasses, std::stoi(get("rounds", "8")), std::stoi(get("threads", "0")), done.data(), wanted.data());
    for(int i = 0; i < ni; ++i)
    {
      Node2& N = B.nodes[i];
      const float* f = &rec[size_t(i) * 16];
      for(int k = 0; k < 2; ++k)
      {
        N.b[k].lo = V3{f[4 * k + 0], f[4 * k + 2], f[8 + 2 * k]};
        N.b[k].hi = V3{f[4 * k + 1], f[4 * k + 3], f[9 + 2 * k]};
        memcpy(&N.c[k], &f[12 + k], 4);
      }
      memcpy(&N.cnt, &f[14], 4);
    }
    if(opt.count("dump2")) dump(opt["dump2"] + ".expected");
    B.nodes[B.root].parent = -1;
    for(int i = 0; i < ni; ++i) for(int k = 0; k < 2; ++k) if(B.nodes[i].c[k] >= 0) B.nodes[B.nodes[i].c[k]].parent = i;
    printf("  parallel reinsertion: %lld moves;", total);
    for(int q = 0; q < ppasses; ++q) printf(" %d/%d", done[q], wanted[q]);
    printf("  -> SAH %.2f\n", sahCost2(B));
  }
  CollapseCfg cfg;
  cfg.maxLeaf  = std::stoi(get("leaf", "2"));
  cfg.sahdp    = get("collapse", "greedy") == "sahdp";
  cfg.cLeafTri = std::stof(get("ctri", "0.24"));
  cfg.quantise = get("quantise", "1") == "1";
  cfg.scaleMantissaBits = std::stoi(get("scalebits", "0"));
  cfg.width    = std::stoi(get("width", "8"));
  cfg.optimalSlots = get("slots", "greedy") == "optimal";
  Bvh8 W = collapse(B, cfg);
  {  // depth: the device collapses level by level (two kernels, a scan and a host round trip per level of the 8-wide tree)
    std::vector<std::pair<int, int>> st{{B.root, 1}};
    int maxD2 = 0; double sumLeafD = 0, leaves = 0;
    while(!st.empty())
    {
      auto [nd, d] = st.back(); st.pop_back();
      maxD2 = std::max(maxD2, d);
      for(int k = 0; k < 2; ++k) { if(B.nodes[nd].c[k] >= 0) st.push_back({B.nodes[nd].c[k], d + 1}); else { sumLeafD += d; leaves += 1; } }
    }
    std::vector<std::pair<int, int>> s8{{0, 1}};
    int maxD8 = 0;
    while(!s8.empty())
    {
      auto [nd, d] = s8.back(); s8.pop_back();
      maxD8 = std::max(maxD8, d);
      for(int sl = 0; sl < 8; ++sl) if(W.nodes[nd].child[sl] >= 0) s8.push_back({W.nodes[nd].child[sl], d + 1});
    }
    printf("depth: BVH2 max %d, mean leaf depth %.1f; 8-wide levels %d\n", maxD2, sumLeafD / leaves, maxD8);
  }
  double leafChildren = 0, innerChildren = 0;
  for(const Node8& N : W.nodes) for(int s = 0; s < 8; ++s) { if(N.child[s] == -1) leafChildren++; else if(N.child[s] >= 0) innerChildren++; }
  printf("BVH8 (%s, leaf <= %d): %zu nodes, SAH(nodes) %.2f, fill %.2f children/node (%.2f leaf), %.2f tris/leaf\n", cfg.sahdp ? "sahdp" : "greedy", cfg.maxLeaf,
         W.nodes.size(), W.sah, (leafChildren + innerChildren) / W.nodes.size(), leafChildren / W.nodes.size(), W.tris.size() / std::max(1.0, leafChildren));

  // rays: bounce rays from random surface points (area-weighted), cosine-distributed
  const int nrays = std::stoi(get("rays", "200000"));
  std::vector<double> cdf(n);
  double acc = 0;
  for(int i = 0; i < n; ++i) { V3 c = cross(tris[i].p1 - tris[i].p0, tris[i].p2 - tris[i].p0); acc += 0.5 * std::sqrt(dot(c, c)); cdf[i] = acc; }
  std::vector<Ray> rays(nrays);
  std::mt19937 rng(12345);
  std::uniform_real_distribution<float> U(0, 1);
  // seed rays by tracing from the camera would be closer to the real distribution; use surface points seen from the eye when possible:
  // here: half area-weighted surface points, half points hit by random rays from the eye
  for(int i = 0; i < nrays; ++i)
  {
    int t = int(std::lower_bound(cdf.begin(), cdf.end(), U(rng) * acc) - cdf.begin());
    t = std::min(t, n - 1);
    float a = U(rng), b = U(rng);
    if(a + b > 1) { a = 1 - a; b = 1 - b; }
    V3 p = tris[t].p0 + (tris[t].p1 - tris[t].p0) * a + (tris[t].p2 - tris[t].p0) * b;
    V3 nrm = normalize(cross(tris[t].p1 - tris[t].p0, tris[t].p2 - tris[t].p0));
    if(U(rng) < 0.5f) nrm = nrm * -1.0f;
    float u1 = U(rng), u2 = U(rng), rr = std::sqrt(u1), ph = 6.2831853f * u2;
    V3 tx = normalize(std::fabs(nrm.x) > 0.5f ? cross(nrm, V3{0, 1, 0}) : cross(nrm, V3{1, 0, 0})), ty = cross(nrm, tx);
    V3 d = tx * (rr * std::cos(ph)) + ty * (rr * std::sin(ph)) + nrm * std::sqrt(std::max(0.0f, 1 - u1));
    rays[i] = {p + nrm * 1e-3f, normalize(d)};
  }
  for(int mode : {0, 2, 3, 4, 1, 5})
  {
    WalkStats S;
    const int defer = std::stoi(get("defer", "3"));
#pragma omp parallel
    {
      WalkStats L;
#pragma omp for schedule(dynamic, 256)
      for(int i = 0; i < nrays; ++i) walk(W, tris, alpha, rays[i], mode, defer, uint32_t(i) * 7919u + 17u, L);
#pragma omp critical
      { S.nodes += L.nodes; S.tris += L.tris; S.rays += L.rays; S.hits += L.hits; S.maxStack = std::max(S.maxStack, L.maxStack); S.deep += L.deep; S.maxGroups = std::max(S.maxGroups, L.maxGroups); S.coplanar += L.coplanar; S.behind += L.behind; }
    }
    const double cnode = 59 + 22 * cfg.width;
    printf("walk %-28s: %.2f node visits + %.2f triangle tests per ray  (cost (59+22w)n+56t = %.0f; hit rate %.3f, max stack %.0f; groups waiting: max %.0f, more than 12 at %.3f %% of the visits; of the triangle tests %.1f %% start in the triangle's plane, %.1f %% face away from it)\n",
           mode == 0 ? "octant order, immediate" : (mode == 5 ? "distance order, no cull" : mode == 1 ? "distance order, immediate" : (mode == 3 ? "octant order + cull at pop" : (mode == 4 ? "octant order + group cull" : "octant order, deferred"))), S.nodes / S.rays, S.tris / S.rays,
           (cnode * S.nodes + 56 * S.tris) / S.rays, S.hits / S.rays, S.maxStack, S.maxGroups, 100.0 * S.deep / S.nodes, 100.0 * S.coplanar / std::max(S.tris, 1.0), 100.0 * S.behind / std::max(S.tris, 1.0));
  }
  // shadow rays: from the same surface points, half towards a fixed sun direction (through the skylight), half uniform over the sphere
  {
    std::vector<Ray> sh(nrays);
    V3 sun = normalize(V3{0.0f, std::cos(0.35f), std::sin(0.35f)});
    for(int i = 0; i < nrays; ++i)
    {
      V3 d;
      if(i & 1) d = sun;
      else { float z = 1 - 2 * U(rng), rr = std::sqrt(std::max(0.0f, 1 - z * z)), ph = 6.2831853f * U(rng); d = V3{rr * std::cos(ph), z, rr * std::sin(ph)}; }
      sh[i] = {rays[i].o, d};
    }
    for(int half : {0, 1, 2}) for(int order : {0, 1, 2, 3, 4, 5})
    {
      WalkStats S;
#pragma omp parallel
      {
        WalkStats L;
#pragma omp for schedule(dynamic, 256)
        for(int i = 0; i < nrays; ++i) if(half == 0 || (half == 1) == bool(i & 1)) walkAny(W, tris, alpha, sh[i], order, uint32_t(i) * 7919u + 17u, L);
#pragma omp critical
        { S.nodes += L.nodes; S.tris += L.tris; S.rays += L.rays; S.hits += L.hits; }
      }
      printf("any-hit %s %-24s: %.2f node visits + %.2f triangle tests per ray (occluded %.3f)\n", half == 0 ? "all rays " : (half == 1 ? "sun rays " : "any dir. "), order == 0 ? "octant order" : (order == 1 ? "largest box first" : (order == 2 ? "nearest first" : (order == 3 ? "slot order, 7 first" : (order == 4 ? "octant order reversed" : "slot order, 0 first")))),
             S.nodes / S.rays, S.tris / S.rays, S.hits / S.rays);
    }
  }
  return 0;
}
