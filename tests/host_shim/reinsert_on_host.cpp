// TEST INFRASTRUCTURE ONLY.  The per-thread phase functions of the BVH2 reinsertion (csrc/device/bvh_reinsert.h) compiled for the host
// through the stand-in <hip/hip_runtime.h> of this directory, one OpenMP loop per kernel of bvh_build.hip, so that the CPU-only test tier
// (tests/test_bvh_reinsert.py) and the laboratory (tools/lab/bvh_lab.cpp) run the code the device runs.  Never loaded by the product.
#include "bvh_reinsert.h"

#include <omp.h>

#include <cstring>
#include <vector>

using namespace pt;

extern "C" {
// nodes: numInner x 16 floats, the builder's records (rewritten in place).  Returns the number of moves carried out over all passes;
// movesPerPass (optional, `passes` ints) receives each pass's count, wantedPerPass the moves with a positive saving before the locks.
__attribute__((visibility("default"))) long long dev_reinsert(float* nodes, int numInner, int root, int passes, int rounds, int threads, int* movesPerPass, int* wantedPerPass)
{
  if(numInner < 1)
    return 0;
  const int numLeaves = numInner + 1, ids = numInner + numLeaves;
  std::vector<int> parent(numInner), leafParent(numLeaves);
  std::vector<ReinsertMove> moves(ids);
  std::vector<unsigned long long> locks(ids);
  std::vector<unsigned int> arrive(numInner);
  Bvh2Tree T{reinterpret_cast<float4*>(nodes), parent.data(), leafParent.data(), numInner, root};
  long long total = 0;
  threads = threads > 0 ? threads : omp_get_max_threads();
  for(int pass = 0; pass < passes; ++pass)
  {
#pragma omp parallel for num_threads(threads) schedule(static)
    for(int i = 0; i < numInner; ++i) reinsertParents(T, i);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 256)
    for(int id = 0; id < ids; ++id) moves[id] = reinsertSearch(T, id);
    std::memset(locks.data(), 0, sizeof(unsigned long long) * ids);
    int done = 0, wanted = 0;
    for(int id = 0; id < ids; ++id) wanted += r2Wanted(moves[id]) ? 1 : 0;
    for(int round = 0; round < rounds; ++round)
    {
#pragma omp parallel for num_threads(threads) schedule(dynamic, 256)
      for(int id = 0; id < ids; ++id) reinsertLock(T, moves.data(), locks.data(), id);
      int won = 0;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 256) reduction(+ : won)
      for(int id = 0; id < ids; ++id) won += reinsertApply(T, moves.data(), locks.data(), id) ? 1 : 0;
#pragma omp parallel for num_threads(threads) schedule(static)
      for(int id = 0; id < ids; ++id) reinsertUnlock(locks.data(), id);
      done += won;
      if(won == 0)
        break;
    }
#pragma omp parallel for num_threads(threads) schedule(static)
    for(int i = 0; i < numInner; ++i) reinsertParents(T, i);
    std::memset(arrive.data(), 0, sizeof(unsigned int) * numInner);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 256)
    for(int l = 0; l < numLeaves; ++l) reinsertRefit(T, arrive.data(), l);
    if(movesPerPass) movesPerPass[pass] = done;
    if(wantedPerPass) wantedPerPass[pass] = wanted;
    total += done;
    if(done == 0)
      break;
  }
  return total;
}
}
