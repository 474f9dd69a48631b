// Stand-alone check of the wave-level work distribution (csrc/device/pt_feed.h): every flat index of a queue of `total`
// entries must be handed out exactly once, for persistent grids of 1024- and 256-thread workgroups.
// build: hipcc --offload-arch=gfx950 -O3 -Ivk_gltf_renderer_amd/csrc/device tools/test_feed.hip -o tools/test_feed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "pt_feed.h"

template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_consume(uint32_t total, uint32_t* heads, uint32_t* marks, int idleLanes)
{
  pt::WaveFeed feed;
  pt::feedInit(feed, total);
  if(!pt::feedBlockHasWork(feed))
    return;
  // a wave asks for work for a varying subset of its lanes, like the trace kernels do
  uint32_t round = 0;
  while(!feed.exhausted)
  {
    const bool     idle = ((threadIdx.x + round * 7u) % 64u) < uint32_t(idleLanes);
    const uint32_t flat = pt::feedTake(feed, idle, heads);
    if(flat != 0xffffffffu)
      atomicAdd(&marks[flat], 1u);
    ++round;
  }
}

int main()
{
  int bad = 0;
  for(int blockSize : {1024, 256})
    for(uint32_t total : {0u, 1u, 5u, 63u, 64u, 70u, 100u, 171u, 325u, 492u, 1303u, 1921u, 4095u, 19200u, 20000u, 27648u, 300000u, 1000003u, 16711680u})
      for(int idle : {64, 20})
      {
        uint32_t *heads, *marks;
        hipMalloc(&heads, 8 * sizeof(uint32_t));
        hipMalloc(&marks, (size_t(total) + 1) * sizeof(uint32_t));
        hipMemset(heads, 0, 8 * sizeof(uint32_t));
        hipMemset(marks, 0, (size_t(total) + 1) * sizeof(uint32_t));
        const int grid = 2048 * 256 / blockSize;
        if(blockSize == 1024)
          hipLaunchKernelGGL(k_consume<1024>, dim3(grid), dim3(1024), 0, 0, total, heads, marks, idle);
        else
          hipLaunchKernelGGL(k_consume<256>, dim3(grid), dim3(256), 0, 0, total, heads, marks, idle);
        hipDeviceSynchronize();
        std::vector<uint32_t> h(size_t(total) + 1);
        hipMemcpy(h.data(), marks, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost);
        size_t miss = 0, dup = 0;
        for(uint32_t i = 0; i < total; ++i)
        {
          miss += h[i] == 0;
          dup += h[i] > 1;
        }
        if(miss || dup || h[total] != 0)
        {
          ++bad;
          printf("FAIL block %d total %u idle %d: missing %zu duplicated %zu overrun %u\n", blockSize, total, idle, miss, dup, h[total]);
        }
        hipFree(heads);
        hipFree(marks);
      }
  printf(bad ? "feed test: %d failing configurations\n" : "feed test: all configurations hand out every index exactly once\n", bad);
  return bad != 0;
}
