// Micro-benchmark: dependent gathers of 80-B records (the BVH8 node fetch pattern) on one GPU.
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench_gather.hip -o gpurun_out/mb_gather ; run: gpurun_out/mb_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

__device__ int g_activeLanes = 64;
template <int LOADS, bool COHERENT>
__global__ void __launch_bounds__(256) k_chase(const uint4* nodes, uint32_t numNodes, int iters, uint32_t* out)
{
  if(int(threadIdx.x & 63) >= g_activeLanes)
    return;
  uint32_t idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u % numNodes;
  if(COHERENT)
    idx = (blockIdx.x * 4 + threadIdx.x / 64) * 2654435761u % numNodes;
  uint32_t acc = 0;
  for(int i = 0; i < iters; ++i)
  {
    const uint4* N = nodes + size_t(idx) * 5;
    uint4        v[5];
#pragma unroll
    for(int k = 0; k < LOADS; ++k)
      v[k] = N[k];
    uint32_t h = 0;
#pragma unroll
    for(int k = 0; k < LOADS; ++k)
      h += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    acc += h;
    idx = (idx * 1664525u + h) % numNodes;  // depends on the loaded data
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int LOADS, bool COHERENT>
void run(const char* name, const uint4* d, uint32_t n, uint32_t* out, int blocksPerCU)
{
  const int cus = 256, iters = 512;
  dim3      grid(cus * blocksPerCU), block(256);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((k_chase<LOADS, COHERENT>), grid, block, 0, 0, d, n, 16, out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k_chase<LOADS, COHERENT>), grid, block, 0, 0, d, n, iters, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  int active = 64;
  hipMemcpyFromSymbol(&active, HIP_SYMBOL(g_activeLanes), sizeof(int));
  double fetches = double(grid.x) * 4.0 * active * iters;
  printf("%-28s nodes=%8u waves/SIMD=%d loads=%d : %7.3f ms  %8.2f G fetch/s  %7.1f GB/s  lat/iter %.0f ns\n", name, n, blocksPerCU, LOADS, ms,
         fetches / ms / 1e6, fetches * LOADS * 16 / ms / 1e6, ms * 1e6 / iters);
}

int main()
{
  for(uint32_t n : {65536u})
  {
    std::vector<uint4> h(size_t(n) * 5);
    uint32_t           s = 12345;
    for(auto& v : h)
    {
      s   = s * 1664525u + 1013904223u;
      v.x = s; v.y = s >> 3; v.z = s >> 7; v.w = s >> 11;
    }
    uint4*    d;
    uint32_t* out;
    hipMalloc(&d, h.size() * sizeof(uint4));
    hipMalloc(&out, sizeof(uint32_t) * 256 * 256 * 8);
    hipMemcpy(d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice);
    for(int lanes : {64, 32, 16, 8, 4, 1})
    {
      hipMemcpyToSymbol(HIP_SYMBOL(g_activeLanes), &lanes, sizeof(int));
      printf("active lanes per wave = %d\n", lanes);
      run<5, false>("divergent 80B", d, n, out, 4);
      run<5, true>("coherent 80B", d, n, out, 4);
    }
    hipFree(d);
    hipFree(out);
  }
  return 0;
}
