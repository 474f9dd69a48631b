// Parallel reinsertion over the BVH2 the builder leaves in HBM (bvh_build.hip), between the PLOC clustering and the 8-wide collapse:
// every node looks for the place in the tree where its subtree would cost the least surface area, the moves that do not get in each
// other's way are carried out, the boxes are refitted, and the pass repeats (Meister & Bittner 2018, "Parallel Reinsertion for Bounding
// Volume Hierarchy Optimization"; the search is Bittner, Hapala, Havran 2013).  What it buys, counted by tools/lab/bvh_lab.cpp on the
// atrium workload under the SAH-optimal collapse: see DESIGN.md section 3.
//
// This header holds the work of ONE thread of each phase as plain functions over the node array, so that the same code is compiled
// into the kernels of bvh_build.hip and -- through tests/host_shim -- into the CPU-only test tier and the laboratory:
//   reinsertParents   one thread per inner node       parent links (inner nodes and leaves)
//   reinsertSearch    one thread per node and leaf    the best new position of its subtree on the FROZEN tree, and what it saves
//   reinsertLock      one thread per node and leaf    atomicMax of (saving, id) on every node the move reads or rewrites: the path from
//                                                     the old position over the common ancestor down to the new one
//   reinsertApply     one thread per node and leaf    a move that holds all its locks is carried out (child links and child boxes only)
//   reinsertUnlock    one thread per node and leaf    (lock, apply, unlock) repeat a few rounds per search for the moves that gave way
//   reinsertRefit     one thread per leaf             bottom-up boxes and triangle counts, the second arrival at a node continues upward
// The outcome is a pure function of the input tree: the search reads a tree nobody writes, the locks are maxima of unique keys, and two
// moves that both hold their locks touch disjoint records.
//
// Node record (bvh_build.hip, 4 x float4): [0] = lo0.x hi0.x lo0.y hi0.y   [1] = lo1.x hi1.x lo1.y hi1.y   [2] = lo0.z hi0.z lo1.z hi1.z
// [3] = child0 child1 triangles-below 0 (int bits).  Child references: >= 0 inner node, < 0 leaf (~triangle).  A node's own box lives in
// its parent's record (the root's nowhere: it is never needed).
#pragma once
#include <hip/hip_runtime.h>
#include "pt_ticket.h"

#include <cstdint>

namespace pt {

#ifndef PT_DEV
#define PT_DEV __device__ __forceinline__
#endif

constexpr int REINSERT_STACK = 48;  // search stack per thread; a subtree that does not fit is not searched further (the search is a heuristic)

struct Bvh2Tree
{
  float4* nodes;       // numInner x 4
  int*    parent;      // numInner (root: -1)
  int*    leafParent;  // numInner + 1
  int     numInner;
  int     root;
};
struct ReinsertMove
{
  int   target;  // the reference (inner >= 0, leaf < 0) above which the subtree goes; valid when lca >= 0
  int   lca;     // the deepest node whose box the move leaves untouched (the old parent itself when the target is below the sibling); -1 = stay
  float gain;    // surface area saved, in the units of the boxes
};
struct RBox
{
  float lo[3], hi[3];
};

PT_DEV int   r2ChildRef(const float4* nodes, int node, int k) { const float4 n3 = nodes[size_t(node) * 4 + 3]; return __float_as_int(k == 0 ? n3.x : n3.y); }
PT_DEV int   r2Count(const float4* nodes, int ref) { return ref >= 0 ? __float_as_int(nodes[size_t(ref) * 4 + 3].z) : 1; }
PT_DEV RBox  r2ChildBox(const float4* nodes, int node, int k)
{
  const float4 a = nodes[size_t(node) * 4 + size_t(k)], c = nodes[size_t(node) * 4 + 2];
  RBox b;
  b.lo[0] = a.x; b.hi[0] = a.y; b.lo[1] = a.z; b.hi[1] = a.w;
  b.lo[2] = k == 0 ? c.x : c.z;
  b.hi[2] = k == 0 ? c.y : c.w;
  return b;
}
// (component stores: the two slots of a record share its third float4, and during the refit they are written by different threads)
PT_DEV void r2SetChildBox(float4* nodes, int node, int k, const RBox& b)
{
  float* f = reinterpret_cast<float*>(nodes + size_t(node) * 4);
  f[4 * k + 0] = b.lo[0]; f[4 * k + 1] = b.hi[0]; f[4 * k + 2] = b.lo[1]; f[4 * k + 3] = b.hi[1];
  f[8 + 2 * k] = b.lo[2]; f[9 + 2 * k] = b.hi[2];
}
PT_DEV void r2SetChildRef(float4* nodes, int node, int k, int ref) { reinterpret_cast<int*>(nodes + size_t(node) * 4)[12 + k] = ref; }
PT_DEV RBox r2Union(const RBox& a, const RBox& b)
{
  RBox r;
  for(int c = 0; c < 3; ++c)
  {
    r.lo[c] = fminf(a.lo[c], b.lo[c]);
    r.hi[c] = fmaxf(a.hi[c], b.hi[c]);
  }
  return r;
}
PT_DEV float r2Area(const RBox& b)
{
  // Three products and two sums, each rounded once: hipcc fuses a * b + c into one fma by default -- also through __fmul_rn / __fadd_rn, which are a
  // plain * and + in this ROCm -- and the host compiler of the CPU test tier does not; with the pragma the device's records equal the host's bit for
  // bit (tools/test_reinsert_gpu.hip).  The only products of these phases are here.
#ifdef __clang__
#pragma clang fp contract(off)
#endif
  const float ex = b.hi[0] - b.lo[0], ey = b.hi[1] - b.lo[1], ez = b.hi[2] - b.lo[2];
  return ex * ey + ey * ez + ez * ex;
}
PT_DEV int r2ParentOf(const Bvh2Tree& T, int ref) { return ref >= 0 ? T.parent[ref] : T.leafParent[~ref]; }
PT_DEV int r2SlotOf(const Bvh2Tree& T, int parentNode, int ref) { return r2ChildRef(T.nodes, parentNode, 0) == ref ? 0 : 1; }
// one index space for the locks and the moves: inner nodes first, then the leaves
PT_DEV int r2IdOf(const Bvh2Tree& T, int ref) { return ref >= 0 ? ref : T.numInner + ~ref; }
PT_DEV int r2RefOf(const Bvh2Tree& T, int id) { return id < T.numInner ? id : ~(id - T.numInner); }

// ---- phase 1: parent links -------------------------------------------------------------------------------------------------------
PT_DEV void reinsertParents(const Bvh2Tree& T, int node)
{
  for(int k = 0; k < 2; ++k)
  {
    const int c = r2ChildRef(T.nodes, node, k);
    if(c >= 0)
      T.parent[c] = node;
    else
      T.leafParent[~c] = node;
  }
  if(node == T.root)
    T.parent[node] = -1;
}

// ---- phase 2: the best position of the subtree `x` ---------------------------------------------------------------------------------
// Taking x out removes its parent p (saves area(p)) and lets the ancestors a1 = parent(p), a2, ... shrink to the union of what is left
// below them; putting it above a node y makes a new parent of area(y u x) and lets the nodes between y and the common ancestor grow.  The
// search climbs from p, and at every level walks the subtree on the other side (depth first, pruned by the best saving found so far: a
// position below n cannot save more than `budget` - growth down to n - area(x)).
PT_DEV void r2SearchSubtree(const Bvh2Tree& T, int topParent, int topSlot, bool topIsCandidate, const RBox& xb, float xArea, float budget, int lca, ReinsertMove& best)
{
  struct Item
  {
    int   parent, slot;
    float grown;  // growth of the nodes above this one (inside the subtree being searched)
  };
  Item stack[REINSERT_STACK];
  int  sp     = 0;
  stack[sp++] = Item{topParent, topSlot, 0.0f};
  bool top    = true;
  while(sp > 0)
  {
    const Item it     = stack[--sp];
    const bool isTop  = top;
    top               = false;
    if(budget - it.grown - xArea <= best.gain)
      continue;
    const int   ref    = r2ChildRef(T.nodes, it.parent, it.slot);
    const RBox  nb     = r2ChildBox(T.nodes, it.parent, it.slot);
    const float merged = r2Area(r2Union(nb, xb));
    if(!isTop || topIsCandidate)
    {
      const float gain = budget - it.grown - merged;
      if(gain > best.gain)
      {
        best.gain   = gain;
        best.target = ref;
        best.lca    = lca;
      }
    }
    if(ref >= 0)
    {
      const float grown = it.grown + (merged - r2Area(nb));
      if(budget - grown - xArea > best.gain && sp + 2 <= REINSERT_STACK)
      {
        stack[sp++] = Item{ref, 1, grown};
        stack[sp++] = Item{ref, 0, grown};
      }
    }
  }
}
PT_DEV ReinsertMove reinsertSearch(const Bvh2Tree& T, int id)
{
  ReinsertMove best{0, -1, 0.0f};
  const int    x = r2RefOf(T, id);
  const int    p = r2ParentOf(T, x);
  if(x == T.root || p < 0 || p == T.root)
    return best;  // (the root's children stay: taking one out would take the root with it)
  const int   xs = r2SlotOf(T, p, x);
  const RBox  xb = r2ChildBox(T.nodes, p, xs);
  const float xArea = r2Area(xb);
  int         a     = T.parent[p];
  RBox        shrunk = r2ChildBox(T.nodes, p, 1 - xs);                       // what is left of p: its other child
  float       budget = r2Area(r2ChildBox(T.nodes, a, r2SlotOf(T, a, p)));    // area(p), saved whatever the target
  // below the sibling (the sibling itself is where x already is)
  r2SearchSubtree(T, p, 1 - xs, false, xb, xArea, budget, p, best);
  int below = p;
  for(;;)
  {
    const int  bs = r2SlotOf(T, a, below);
    r2SearchSubtree(T, a, 1 - bs, true, xb, xArea, budget, a, best);  // the other side of this ancestor; `a` itself keeps its box
    if(a == T.root)
      break;
    shrunk          = r2Union(shrunk, r2ChildBox(T.nodes, a, 1 - bs));
    const int up    = T.parent[a];
    budget += r2Area(r2ChildBox(T.nodes, up, r2SlotOf(T, up, a))) - r2Area(shrunk);  // one level higher `a` shrinks too
    below = a;
    a     = up;
  }
  return best;
}

// ---- phases 3 + 4: locks and moves -------------------------------------------------------------------------------------------------
// The records a move depends on: x, its parent p and grandparent, the chain from p up to the common ancestor, and the chain from the
// target up to (not including) the common ancestor.  Two moves whose sets are disjoint can be carried out in either order -- and cannot
// close a cycle: a target inside another moved subtree has that subtree's root on its chain.
PT_DEV unsigned long long r2Key(const ReinsertMove& m, int id) { return (static_cast<unsigned long long>(__float_as_uint(m.gain)) << 32) | static_cast<uint32_t>(id); }
// f(node id, topo): topo = the move rewrites this node's links (x, its parent and grandparent, the target and the target's parent); the other nodes of
// the two chains only have their boxes changed by it.
template <typename F>
PT_DEV void r2ForEachLocked(const Bvh2Tree& T, int id, const ReinsertMove& m, F f)
{
  const int x = r2RefOf(T, id);
  const int p = r2ParentOf(T, x);
  const int q = r2ParentOf(T, m.target);
  f(id, true);
  f(T.parent[p], true);
  for(int n = p;; n = T.parent[n])
  {
    f(n, n == p || n == q);
    if(n == m.lca)
      break;
  }
  for(int r = m.target;;)
  {
    f(r2IdOf(T, r), r == m.target || r == q);
    const int up = r2ParentOf(T, r);
    if(up == m.lca)
      break;  // (below the sibling the chain ends at the sibling, higher up at the ancestor's other child)
    r = up;
  }
}
PT_DEV bool r2Wanted(const ReinsertMove& m) { return m.lca >= 0 && m.gain > 0.0f; }
// Several lock rounds per search: a move that loses a record to a better one gives way for this round only -- unless the better one is then
// carried out (its records stay TAKEN until the next search), it tries again in the next round, so that one long path through the upper
// tree does not cost a pass to everything it crosses.
constexpr unsigned long long REINSERT_TAKEN   = ~0ull;
constexpr unsigned long long REINSERT_CROSSED = ~0ull - 1ull;
PT_DEV bool r2Blocked(unsigned long long lock, bool topo)
{
  (void)topo;
  return lock == REINSERT_TAKEN;
}
PT_DEV void reinsertLock(const Bvh2Tree& T, ReinsertMove* moves, unsigned long long* locks, int id)
{
  const ReinsertMove m = moves[id];
  if(!r2Wanted(m))
    return;
  bool taken = false;
  r2ForEachLocked(T, id, m, [&](int n, bool topo) { taken = taken || r2Blocked(locks[n], topo); });
  if(taken)
  {
    moves[id].lca = -1;  // one of its records belongs to a move already carried out: the next search decides again
    return;
  }
  const unsigned long long key = r2Key(m, id);
  r2ForEachLocked(T, id, m, [&](int n, bool) { atomicMax(&locks[n], key); });  // (a CROSSED node keeps its mark: no key is that large)
}
PT_DEV void reinsertUnlock(unsigned long long* locks, int id)
{
  if(locks[id] < REINSERT_CROSSED)
    locks[id] = 0ull;
}
// returns whether the move was carried out
PT_DEV bool reinsertApply(const Bvh2Tree& T, ReinsertMove* moves, unsigned long long* locks, int id)
{
  const ReinsertMove m = moves[id];
  if(!r2Wanted(m))
    return false;
  const unsigned long long key = r2Key(m, id);
  bool                     all = true;
  r2ForEachLocked(T, id, m, [&](int n, bool) { all = all && locks[n] == key; });
  if(!all)
    return false;
  moves[id].lca = -1;
  const int  x = r2RefOf(T, id), p = r2ParentOf(T, x), g = T.parent[p], y = m.target, q = r2ParentOf(T, y);
  const int  xs = r2SlotOf(T, p, x), gs = r2SlotOf(T, g, p), qs = r2SlotOf(T, q, y);
  const int  s  = r2ChildRef(T.nodes, p, 1 - xs);
  const RBox xb = r2ChildBox(T.nodes, p, xs), sb = r2ChildBox(T.nodes, p, 1 - xs), yb = r2ChildBox(T.nodes, q, qs);
  // the sibling takes p's place ...
  r2SetChildRef(T.nodes, g, gs, s);
  r2SetChildBox(T.nodes, g, gs, sb);
  // ... and p goes between the target and its parent (q may be g: the target is then p's old sibling slot's neighbour, another slot)
  r2SetChildRef(T.nodes, p, 0, y);
  r2SetChildBox(T.nodes, p, 0, yb);
  r2SetChildRef(T.nodes, p, 1, x);
  r2SetChildBox(T.nodes, p, 1, xb);
  r2SetChildRef(T.nodes, q, qs, p);
  r2SetChildBox(T.nodes, q, qs, r2Union(yb, xb));
  // (the parent links stay those of the searched tree until the pass ends: the chains of the moves still waiting are read from them)
  r2ForEachLocked(T, id, m, [&](int n, bool) { locks[n] = REINSERT_TAKEN; });
  return true;
}

// ---- phase 5: boxes and triangle counts, bottom-up -----------------------------------------------------------------------------------
// (parent links rebuilt by phase 1 after the moves; `arrive` zeroed)
// The words one thread of the refit writes and another reads -- a node's triangle count, the child boxes in its parent's record -- go through
// AGENT-SCOPE relaxed atomic loads and stores (global_load / global_store with sc1: coherent at the device's memory side, past the per-XCD L2s),
// ordered against the ticket by waiting for the stores' acknowledgement.  Round 4 wrote them as plain stores between two __threadfence() -- on a
// device with eight non-coherent L2s that is an L2 write-back + invalidate per thread and tree level, and the refit was 80 % of a reinsertion pass
// (2.6 ms at 0.26 M triangles, 27 ms at 2.8 M: profiles/r05_build_times.txt).  Same values, same order of operations: the records stay bit-identical
// to the host run of this function (tools/test_reinsert_gpu.hip).
#if defined(__HIP_DEVICE_COMPILE__)
PT_DEV float r2CohLoad(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
PT_DEV int   r2CohLoad(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
PT_DEV void  r2CohStore(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
PT_DEV void  r2CohStore(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// (how the data words are ordered against the ticket, and on which targets: pt_ticket.h)
PT_DEV unsigned int r2Ticket(unsigned int* p) { return ticketArrive(p); }
#else  // the host build of the CPU test tier (tests/host_shim): one coherent memory
PT_DEV float r2CohLoad(const float* p) { const int i = __atomic_load_n(reinterpret_cast<const int*>(p), __ATOMIC_SEQ_CST); float v; __builtin_memcpy(&v, &i, 4); return v; }
PT_DEV int   r2CohLoad(const int* p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
PT_DEV void  r2CohStore(float* p, float v) { int i; __builtin_memcpy(&i, &v, 4); __atomic_store_n(reinterpret_cast<int*>(p), i, __ATOMIC_SEQ_CST); }
PT_DEV void  r2CohStore(int* p, int v) { __atomic_store_n(p, v, __ATOMIC_SEQ_CST); }
PT_DEV unsigned int r2Ticket(unsigned int* p) { return __atomic_fetch_add(p, 1u, __ATOMIC_SEQ_CST); }
#endif
PT_DEV RBox r2CohChildBox(float4* nodes, int node, int k)
{
  const float* f = reinterpret_cast<const float*>(nodes + size_t(node) * 4);
  RBox         b;
  b.lo[0] = r2CohLoad(f + 4 * k + 0); b.hi[0] = r2CohLoad(f + 4 * k + 1); b.lo[1] = r2CohLoad(f + 4 * k + 2); b.hi[1] = r2CohLoad(f + 4 * k + 3);
  b.lo[2] = r2CohLoad(f + 8 + 2 * k); b.hi[2] = r2CohLoad(f + 9 + 2 * k);
  return b;
}
PT_DEV void reinsertRefit(const Bvh2Tree& T, unsigned int* arrive, int leaf)
{
  int cur = T.leafParent[leaf];
  while(cur >= 0)
  {
    if(r2Ticket(&arrive[cur]) == 0u)
      return;  // first arrival: the other subtree's thread finishes this node
    int* const w  = reinterpret_cast<int*>(T.nodes + size_t(cur) * 4);
    const int  c0 = w[12], c1 = w[13];  // (child links: nobody writes them in this phase)
    const int  n0 = c0 >= 0 ? r2CohLoad(reinterpret_cast<const int*>(T.nodes + size_t(c0) * 4) + 14) : 1;
    const int  n1 = c1 >= 0 ? r2CohLoad(reinterpret_cast<const int*>(T.nodes + size_t(c1) * 4) + 14) : 1;
    r2CohStore(w + 14, n0 + n1);
    const int up = T.parent[cur];
    if(up >= 0)
    {
      const RBox   u = r2Union(r2CohChildBox(T.nodes, cur, 0), r2CohChildBox(T.nodes, cur, 1));
      const int    k = r2SlotOf(T, up, cur);
      float* const f = reinterpret_cast<float*>(T.nodes + size_t(up) * 4);
      r2CohStore(f + 4 * k + 0, u.lo[0]); r2CohStore(f + 4 * k + 1, u.hi[0]); r2CohStore(f + 4 * k + 2, u.lo[1]); r2CohStore(f + 4 * k + 3, u.hi[1]);
      r2CohStore(f + 8 + 2 * k, u.lo[2]); r2CohStore(f + 9 + 2 * k, u.hi[2]);
    }
    cur = up;
  }
}

}  // namespace pt
