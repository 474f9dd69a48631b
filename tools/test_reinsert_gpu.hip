// Stand-alone GPU check of the builder's reinsertion passes (no Python, no torch: starts in a second on a fresh box).  Compiles bvh_build.hip
// ITSELF into this program, so what runs is the builder's own reinsertBvh2 with its k_re_* kernels, feeds it BVH2 records the laboratory dumped
// (tools/lab/bvh_lab.cpp dump2=<prefix>: <prefix>.in as clustered, <prefix>.expected after `passes` parallel passes of the same phase functions on the
// host, tests/host_shim) and compares byte for byte -- the outcome is a pure function of the input tree.  Then times the passes, also on K copies of
// the tree under a balanced top (the street workload's size without shipping its 425 MB of records).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ivk_gltf_renderer_amd/csrc/device -Wno-unused-function -o tools/_scratch/test_reinsert_gpu tools/test_reinsert_gpu.hip
//   tools/_scratch/test_reinsert_gpu <prefix> <passes> <rounds> [copies]
// (the -DREINSERT_SOFT_LOCKS flavour: add the flag here AND to the laboratory's build, so that <prefix>.expected comes from the same flavour)
#include "bvh_build.hip"

#include <chrono>
#include <cstdio>
#include <vector>

static bool readRecords(const std::string& path, int& numInner, int& root, std::vector<float>& rec)
{
  FILE* f = fopen(path.c_str(), "rb");
  if(!f) { perror(path.c_str()); return false; }
  bool ok = fread(&numInner, 4, 1, f) == 1 && fread(&root, 4, 1, f) == 1;
  if(ok) { rec.resize(size_t(numInner) * 16); ok = fread(rec.data(), 4, rec.size(), f) == rec.size(); }
  fclose(f);
  return ok;
}
static double areaCost(const std::vector<float>& rec, int numInner)  // sum over the inner nodes of the area of the union of their child boxes
{
  double c = 0;
  for(int i = 0; i < numInner; ++i)
  {
    const float* f = &rec[size_t(i) * 16];
    const double ex = double(std::max(f[1], f[5])) - std::min(f[0], f[4]), ey = double(std::max(f[3], f[7])) - std::min(f[2], f[6]), ez = double(std::max(f[9], f[11])) - std::min(f[8], f[10]);
    c += ex * ey + ey * ez + ez * ex;
  }
  return c;
}
// every leaf once, every inner node once, child boxes exact unions of what is below, triangle counts
static bool validTree(const std::vector<float>& rec, int numInner, int root)
{
  std::vector<int> seenInner(numInner, 0), seenLeaf(numInner + 1, 0), stack{root};
  std::vector<int> order;
  seenInner[root] = 1;
  while(!stack.empty())
  {
    const int i = stack.back(); stack.pop_back();
    order.push_back(i);
    for(int k = 0; k < 2; ++k)
    {
      int ref; memcpy(&ref, &rec[size_t(i) * 16 + 12 + k], 4);
      if(ref >= 0) { if(ref >= numInner || seenInner[ref]++) return false; stack.push_back(ref); }
      else { if(~ref > numInner || seenLeaf[~ref]++) return false; }
    }
  }
  if(int(order.size()) != numInner) return false;
  for(int l = 0; l <= numInner; ++l) if(seenLeaf[l] != 1) return false;
  std::vector<int> cnt(numInner, 0);
  for(size_t o = order.size(); o-- > 0;)  // children before parents
  {
    const int i = order[o];
    const float* f = &rec[size_t(i) * 16];
    int total = 0;
    for(int k = 0; k < 2; ++k)
    {
      int ref; memcpy(&ref, &f[12 + k], 4);
      total += ref >= 0 ? cnt[ref] : 1;
      if(ref >= 0)
      {
        const float* c = &rec[size_t(ref) * 16];
        const float u[6] = {std::min(c[0], c[4]), std::max(c[1], c[5]), std::min(c[2], c[6]), std::max(c[3], c[7]), std::min(c[8], c[10]), std::max(c[9], c[11])};
        const float b[6] = {f[4 * k], f[4 * k + 1], f[4 * k + 2], f[4 * k + 3], f[8 + 2 * k], f[9 + 2 * k]};
        for(int a = 0; a < 6; ++a) if(u[a] != b[a]) return false;
      }
    }
    int stored; memcpy(&stored, &f[14], 4);
    if(stored != total) return false;
    cnt[i] = total;
  }
  return true;
}
static bool runOnDevice(std::vector<float>& rec, int numInner, int root, int passes, int rounds, uint32_t& moves, double& ms)
{
  float4* d = nullptr;
  if(hipMalloc(&d, rec.size() * 4) != hipSuccess || hipMemcpy(d, rec.data(), rec.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return false;
  std::string err;
  const auto  t0 = std::chrono::steady_clock::now();
  const bool  ok = pt::reinsertBvh2(d, numInner, root, passes, rounds, nullptr, err, &moves);
  (void)hipDeviceSynchronize();
  ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if(!ok) fprintf(stderr, "reinsertBvh2: %s\n", err.c_str());
  const bool back = hipMemcpy(rec.data(), d, rec.size() * 4, hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(d);
  return ok && back;
}

int main(int argc, char** argv)
{
  if(argc < 4) { fprintf(stderr, "usage: test_reinsert_gpu <prefix> <passes> <rounds> [copies]\n"); return 2; }
  const std::string prefix = argv[1];
  const int passes = atoi(argv[2]), rounds = atoi(argv[3]), copies = argc > 4 ? atoi(argv[4]) : 0;
  int ni = 0, root = 0, ni2 = 0, root2 = 0;
  std::vector<float> in, expected;
  if(!readRecords(prefix + ".in", ni, root, in) || !readRecords(prefix + ".expected", ni2, root2, expected) || ni != ni2 || root != root2) return 2;
  std::vector<float> got = in;
  uint32_t moves = 0;
  double   ms    = 0;
  (void)hipFree(nullptr);
  { std::vector<float> warm = in; uint32_t m; double t; if(!runOnDevice(warm, ni, root, 1, rounds, m, t)) return 1; }  // (first launches: code objects load)
  if(!runOnDevice(got, ni, root, passes, rounds, moves, ms)) return 1;
  size_t diff = 0, first = got.size();
  for(size_t i = 0; i < got.size(); ++i) if(memcmp(&got[i], &expected[i], 4) != 0) { ++diff; if(first == got.size()) first = i; }
  std::vector<float> again = in;
  uint32_t moves2 = 0; double ms2 = 0;
  if(!runOnDevice(again, ni, root, passes, rounds, moves2, ms2)) return 1;
  printf("REINSERT %s: tree after the device's passes %s; a second run of the device gives %s records; expected (host) tree %s\n", prefix.c_str(), validTree(got, ni, root) ? "VALID" : "BROKEN",
         memcmp(again.data(), got.data(), got.size() * 4) == 0 ? "THE SAME" : "OTHER", validTree(expected, ni, root) ? "valid" : "BROKEN");
  if(diff) printf("  first differing word: node %zu word %zu: device %.9g host %.9g\n", first / 16, first % 16, got[first], expected[first]);
  printf("REINSERT %s: %d inner nodes, %d passes x %d rounds on the device: %u moves, %.1f ms (%.2f ms per pass), area cost %.6g -> %.6g (host: %.6g), %zu of %zu words differ from the host's records -> %s\n",
         prefix.c_str(), ni, passes, rounds, moves, ms, ms / passes, areaCost(in, ni), areaCost(got, ni), areaCost(expected, ni), diff, got.size(), diff == 0 ? "IDENTICAL" : "DIFFERENT");
  if(copies > 1)
  {
    // K translated copies side by side under a balanced top tree: nodes of copy c at [c * ni, (c + 1) * ni), leaves at c * (ni + 1) + leaf, top nodes behind
    const int K = copies, total = K * ni + (K - 1);
    std::vector<float> big(size_t(total) * 16);
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    { const float* f = &in[size_t(root) * 16]; lo[0] = std::min(f[0], f[4]); hi[0] = std::max(f[1], f[5]); lo[1] = std::min(f[2], f[6]); hi[1] = std::max(f[3], f[7]); lo[2] = std::min(f[8], f[10]); hi[2] = std::max(f[9], f[11]); }
    const float step = (hi[0] - lo[0]) * 1.02f;
    for(int c = 0; c < K; ++c)
      for(int i = 0; i < ni; ++i)
      {
        const float* s = &in[size_t(i) * 16];
        float*       d = &big[(size_t(c) * ni + i) * 16];
        memcpy(d, s, 64);
        d[0] += c * step; d[1] += c * step; d[4] += c * step; d[5] += c * step;
        for(int k = 0; k < 2; ++k)
        {
          int ref; memcpy(&ref, &s[12 + k], 4);
          ref = ref >= 0 ? ref + c * ni : ~(~ref + c * (ni + 1));
          memcpy(&d[12 + k], &ref, 4);
        }
      }
    struct Sub { int ref; float lo[3], hi[3]; int cnt; };
    std::vector<Sub> level;
    for(int c = 0; c < K; ++c) { Sub s; s.ref = root + c * ni; for(int a = 0; a < 3; ++a) { s.lo[a] = lo[a]; s.hi[a] = hi[a]; } s.lo[0] += c * step; s.hi[0] += c * step; s.cnt = ni + 1; level.push_back(s); }
    int next = K * ni;
    while(level.size() > 1)
    {
      std::vector<Sub> up;
      for(size_t i = 0; i + 1 < level.size(); i += 2)
      {
        float* d = &big[size_t(next) * 16];
        const Sub &a = level[i], &b = level[i + 1];
        d[0] = a.lo[0]; d[1] = a.hi[0]; d[2] = a.lo[1]; d[3] = a.hi[1]; d[8] = a.lo[2]; d[9] = a.hi[2];
        d[4] = b.lo[0]; d[5] = b.hi[0]; d[6] = b.lo[1]; d[7] = b.hi[1]; d[10] = b.lo[2]; d[11] = b.hi[2];
        const int cnt = a.cnt + b.cnt;
        memcpy(&d[12], &a.ref, 4); memcpy(&d[13], &b.ref, 4); memcpy(&d[14], &cnt, 4); d[15] = 0;
        Sub s; s.ref = next++; s.cnt = cnt;
        for(int x = 0; x < 3; ++x) { s.lo[x] = std::min(a.lo[x], b.lo[x]); s.hi[x] = std::max(a.hi[x], b.hi[x]); }
        up.push_back(s);
      }
      if(level.size() & 1) up.push_back(level.back());
      level.swap(up);
    }
    const double before = areaCost(big, total);
    if(!runOnDevice(big, total, level[0].ref, passes, rounds, moves, ms)) return 1;
    printf("REINSERT %s x %d copies: %d inner nodes, %d passes x %d rounds: %u moves, %.1f ms (%.2f ms per pass), area cost %.6g -> %.6g, tree %s\n", prefix.c_str(), K, total, passes, rounds, moves,
           ms, ms / passes, before, areaCost(big, total), validTree(big, total, level[0].ref) ? "VALID" : "BROKEN");
  }
  return diff == 0 ? 0 : 1;
}
