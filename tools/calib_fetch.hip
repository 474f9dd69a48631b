// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on THIS renderer's access patterns (MI355X_MICROARCH.md: "FETCH_SIZE reports
// exactly 1/2 of the bytes of a wide coalesced streaming read ... other access widths and WRITE_SIZE are uncalibrated: calibrate on a
// known byte count in your own access pattern").  Every kernel below moves a known number of bytes through a 6 GiB buffer -- far
// beyond L2 (32 MiB) and the Infinity Cache (256 MiB), every cache line touched exactly once per launch -- in one of the patterns
// the path tracer has: wide streams (queue entries, path state: 16 B per lane, consecutive), and gathers of 16-B (texel footprints),
// 48-B (vertices, triangles) and 80-B (BVH8 nodes) records at random places.
//   build : hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o gpurun_out/calib_fetch
//   run   : gpurun_out/calib_fetch                                   -> one line per kernel: bytes asked for, 64-B lines touched, time
//           rocprofv3 --pmc FETCH_SIZE -- gpurun_out/calib_fetch     -> raw counter per kernel (tools/calib_fetch_report.py joins the two)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                   \
  do                                                                                               \
  {                                                                                                \
    hipError_t e_ = (x);                                                                           \
    if(e_ != hipSuccess)                                                                           \
    {                                                                                              \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                      \
      exit(1);                                                                                     \
    }                                                                                              \
  } while(0)

// record index -> a pseudo-random other record index, a bijection on [0, n) for n a power of two (odd multiplier + xor-shift rounds)
__device__ __forceinline__ uint32_t scramble(uint32_t i, uint32_t mask)
{
  i = (i * 2654435761u) & mask;
  i ^= i >> 7;
  i = (i * 40503u + 12345u) & mask;  // odd multiplier: bijection mod 2^k
  i ^= i >> 11;
  return i & mask;
}

__global__ void __launch_bounds__(256) calib_stream_read16(const uint4* __restrict__ src, size_t n, uint32_t* __restrict__ sink)
{
  uint32_t acc = 0;
  for(size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256)
  {
    const uint4 v = src[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if(acc == 0x12345678u)
    sink[0] = acc;
}
// REC16 x 16-B words per record, records of `strideWords` 16-B words (>= REC16), one record per thread and iteration at a scrambled index
template <int REC16>
__global__ void __launch_bounds__(256) calib_gather(const uint4* __restrict__ src, uint32_t numRecords, uint32_t strideWords, uint32_t* __restrict__ sink)
{
  uint32_t acc = 0;
  for(uint32_t i = blockIdx.x * 256u + threadIdx.x; i < numRecords; i += gridDim.x * 256u)
  {
    const uint4* r = src + size_t(scramble(i, numRecords - 1u)) * strideWords;
#pragma unroll
    for(int k = 0; k < REC16; ++k)
    {
      const uint4 v = r[k];
      acc += v.x ^ v.y ^ v.z ^ v.w;
    }
  }
  if(acc == 0x12345678u)
    sink[0] = acc;
}
__global__ void __launch_bounds__(256) calib_stream_write16(uint4* __restrict__ dst, size_t n)
{
  for(size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256)
    dst[i] = make_uint4(uint32_t(i), 1u, 2u, 3u);
}
__global__ void __launch_bounds__(256) calib_scatter_write16(uint4* __restrict__ dst, uint32_t numRecords, uint32_t strideWords)
{
  for(uint32_t i = blockIdx.x * 256u + threadIdx.x; i < numRecords; i += gridDim.x * 256u)
    dst[size_t(scramble(i, numRecords - 1u)) * strideWords] = make_uint4(i, 1u, 2u, 3u);
}

int main()
{
  const size_t bytes = size_t(6) << 30, words = bytes / 16;
  uint4*       buf   = nullptr;
  uint32_t*    sink  = nullptr;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(buf, 1, bytes));
  CHECK(hipDeviceSynchronize());
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  const dim3 grid(256 * 16), block(256);
  auto       timeIt = [&](const char* name, double askedBytes, double lines64, auto&& launch) {
    launch();  // warm-up (also so that the counter run sees two identical dispatches per kernel)
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    printf("{\"kernel\": \"%s\", \"asked_bytes\": %.0f, \"lines64_bytes\": %.0f, \"ms\": %.4f, \"asked_GBps\": %.1f}\n", name, askedBytes, lines64 * 64.0, ms, askedBytes / ms / 1e6);
  };
  // 1. wide stream: every 16-B word of the buffer once
  timeIt("calib_stream_read16", double(bytes), double(bytes / 64), [&] { hipLaunchKernelGGL(calib_stream_read16, grid, block, 0, 0, buf, words, sink); });
  // 2. gathers: 2^25 records (one per thread-iteration), record pitch 128 B (8 words) so that no two records share a 64-B or a 128-B line
  //    16-B record: 1 line; 48-B record (starts at the pitch: inside one 64-B line); 80-B record: 2 lines
  const uint32_t nrec = 1u << 25, pitch = 8;  // 2^25 x 128 B = 4 GiB of the buffer
  timeIt("calib_gather<1>", double(nrec) * 16, double(nrec) * 1, [&] { hipLaunchKernelGGL(calib_gather<1>, grid, block, 0, 0, buf, nrec, pitch, sink); });
  timeIt("calib_gather<3>", double(nrec) * 48, double(nrec) * 1, [&] { hipLaunchKernelGGL(calib_gather<3>, grid, block, 0, 0, buf, nrec, pitch, sink); });
  timeIt("calib_gather<5>", double(nrec) * 80, double(nrec) * 2, [&] { hipLaunchKernelGGL(calib_gather<5>, grid, block, 0, 0, buf, nrec, pitch, sink); });
  // 3. writes: wide stream, and 16-B scatters (one per 128-B pitch)
  timeIt("calib_stream_write16", double(bytes), double(bytes / 64), [&] { hipLaunchKernelGGL(calib_stream_write16, grid, block, 0, 0, buf, words); });
  timeIt("calib_scatter_write16", double(nrec) * 16, double(nrec) * 1, [&] { hipLaunchKernelGGL(calib_scatter_write16, grid, block, 0, 0, buf, nrec, pitch); });
  CHECK(hipFree(buf));
  CHECK(hipFree(sink));
  return 0;
}
