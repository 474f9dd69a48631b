// Micro-benchmark: issue cost of the vector instructions the BVH walks are made of, on one MI355X.
// For each instruction: 16 independent copies per loop body (no dependent chain inside a body), W waves per SIMD, 1024 SIMDs.
// Prints shader cycles per wave64 instruction per SIMD (s_memtime ticks of the longest wave x SIMDs / instructions issued) --
// 2.0 = the SIMD-32 full rate of MI355X_MICROARCH.md, 4.0 = half rate -- and the effective clock (s_memtime against wall time).
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench_valu.hip -o gpurun_out/mb_valu ; run: gpurun_out/mb_valu
// Not part of the product: it tells bench.py's "issue_frac" which cycle count a vector instruction stands for (LABNOTES.md section 4).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define KERNEL(NAME, ASM)                                                                                                          \
  __global__ void __launch_bounds__(256) NAME(int iters, unsigned long long* ticks, float* sink)                                   \
  {                                                                                                                                \
    float v[16];                                                                                                                   \
    for(int i = 0; i < 16; ++i)                                                                                                    \
      v[i] = float(threadIdx.x + i);                                                                                               \
    float    a = 1.0001f, b = 0.5f;                                                                                                \
    uint32_t lanesel = (threadIdx.x * 4u) & 255u;                                                                                  \
    (void)lanesel;                                                                                                                 \
    const unsigned long long t0 = __builtin_readcyclecounter();                                                                    \
    for(int it = 0; it < iters; ++it)                                                                                              \
    {                                                                                                                              \
      REP16(ASM)                                                                                                                   \
    }                                                                                                                              \
    const unsigned long long t1 = __builtin_readcyclecounter();                                                                    \
    float s = 0.0f;                                                                                                                \
    for(int i = 0; i < 16; ++i)                                                                                                    \
      s += v[i];                                                                                                                   \
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                               \
    if((threadIdx.x & 63) == 0)                                                                                                    \
      ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;                                                          \
  }

#define A_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
#define A_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
#define A_SUB(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
#define A_MAX(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
#define A_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
#define A_MIN3(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
#define A_MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
#define A_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(a) : );
#define A_CNDMASK_S(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(v[i]) : "v"(a) : );
#define A_CNDMASK_CMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(a) : "vcc");
#define A_CNDMASK_FMA(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n\tv_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
#define A_CNDMASK_EXEC(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, exec" : "+v"(v[i]) : "v"(a) : );
#define A_CVTUB(i) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(v[i]));
#define A_CVTUB0(i) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(v[i]));
#define A_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(v[i]) : "v"(a));
#define A_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
#define A_OR3(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
#define A_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
#define A_ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
#define A_LSHL(i) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(v[i]));
#define A_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(v[i]) : "v"(a));
#define A_BFE(i) asm volatile("v_bfe_u32 %0, %0, 5, 3" : "+v"(v[i]));
#define A_BFM(i) asm volatile("v_bfm_b32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
#define A_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
#define A_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
#define A_CMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(v[i]), "v"(a) : "vcc");
#define A_CMPX(i) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(v[i]), "v"(a) : "s20", "s21");
#define A_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
#define A_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(a));
#define A_DPP(i) asm volatile("v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v[i]));
#define A_SDWA(i) asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "+v"(v[i]));
#define A_POPC(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
#define A_FFBH(i) asm volatile("v_ffbh_u32 %0, %0" : "+v"(v[i]));
#define A_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
#define A_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i & 7]) : "v"(q));
#define A_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(q));
#define A_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(q));
#define A_BPERM(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(v[i]) : "v"(lanesel));
#define A_READLANE(i) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(v[i]) : "s20");
#define A_SALU(i) asm volatile("s_add_u32 s20, s20, 1" : : : "s20", "scc");
#define A_MIX(i) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_add_u32 s20, s20, 1" : "+v"(v[i]) : "v"(a), "v"(b) : "s20", "scc");

KERNEL(k_fma, A_FMA) KERNEL(k_mul, A_MUL) KERNEL(k_sub, A_SUB) KERNEL(k_max, A_MAX) KERNEL(k_max3, A_MAX3) KERNEL(k_min3, A_MIN3) KERNEL(k_med3, A_MED3)
KERNEL(k_cndmask, A_CNDMASK) KERNEL(k_cndmask_sgpr, A_CNDMASK_S) KERNEL(k_cmp_cndmask_x2, A_CNDMASK_CMP) KERNEL(k_cndmask_fma_x2, A_CNDMASK_FMA) KERNEL(k_cvt_ubyte1, A_CVTUB) KERNEL(k_cvt_ubyte0, A_CVTUB0) KERNEL(k_alignbit, A_ALIGNBIT) KERNEL(k_and, A_AND) KERNEL(k_or3, A_OR3)
KERNEL(k_and_or, A_ANDOR) KERNEL(k_add_u32, A_ADDU) KERNEL(k_lshl, A_LSHL) KERNEL(k_lshl_add, A_LSHLADD) KERNEL(k_bfe, A_BFE) KERNEL(k_bfm, A_BFM) KERNEL(k_mul_lo, A_MULLO)
KERNEL(k_mad_u24, A_MAD24) KERNEL(k_cmp_vcc, A_CMP) KERNEL(k_cmp_sgpr, A_CMPX) KERNEL(k_rcp, A_RCP) KERNEL(k_mov, A_MOV) KERNEL(k_max_dpp, A_DPP) KERNEL(k_cvt_sdwa, A_SDWA)
KERNEL(k_bcnt, A_POPC) KERNEL(k_ffbh, A_FFBH) KERNEL(k_perm, A_PERM) KERNEL(k_bpermute_waited, A_BPERM) KERNEL(k_readlane, A_READLANE) KERNEL(k_salu, A_SALU)
KERNEL(k_fma_plus_salu, A_MIX)

typedef float f32x2 __attribute__((ext_vector_type(2)));
#define KERNEL_PK(NAME, ASM)                                                                                                       \
  __global__ void __launch_bounds__(256) NAME(int iters, unsigned long long* ticks, float* sink)                                   \
  {                                                                                                                                \
    f32x2 p[8];                                                                                                                    \
    for(int i = 0; i < 8; ++i)                                                                                                     \
      p[i] = f32x2{float(threadIdx.x + i), float(i)};                                                                              \
    f32x2                    q  = {1.0001f, 0.5f};                                                                                 \
    const unsigned long long t0 = __builtin_readcyclecounter();                                                                    \
    for(int it = 0; it < iters; ++it)                                                                                              \
    {                                                                                                                              \
      REP16(ASM)                                                                                                                   \
    }                                                                                                                              \
    const unsigned long long t1 = __builtin_readcyclecounter();                                                                    \
    float s = 0.0f;                                                                                                                \
    for(int i = 0; i < 8; ++i)                                                                                                     \
      s += p[i].x + p[i].y;                                                                                                        \
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                               \
    if((threadIdx.x & 63) == 0)                                                                                                    \
      ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;                                                          \
  }
KERNEL_PK(k_pk_fma, A_PKFMA) KERNEL_PK(k_pk_mul, A_PKMUL) KERNEL_PK(k_pk_add, A_PKADD)

typedef void (*Kern)(int, unsigned long long*, float*);
static void run(const char* name, Kern k, int wavesPerSimd, unsigned long long* dTicks, float* dSink)
{
  const int  cus = 256, iters = 49152;
  const int  blocks = cus * wavesPerSimd;  // 256 threads = 4 waves = one per SIMD
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, 64, dTicks, dSink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, iters, dTicks, dSink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> t(size_t(blocks) * 4);
  hipMemcpy(t.data(), dTicks, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double sum = 0, mx = 0;
  for(auto v : t)
  {
    sum += double(v);
    mx = v > mx ? double(v) : mx;
  }
  const double mean = sum / double(t.size());
  const double instPerWave = double(iters) * 16.0;
  // one SIMD hosts wavesPerSimd of these waves at a time: cycles per instruction per SIMD = wave ticks / (instructions of all its waves)
  printf("%-20s waves/SIMD=%d  wave ticks mean %10.0f max %10.0f  cycles/inst/SIMD %6.2f  (by wall time at 2.4 GHz: %6.2f)  %7.3f ms  ticks/us %7.1f\n", name, wavesPerSimd, mean,
         mx, mean / (instPerWave * wavesPerSimd), double(ms) * 1e-3 * 2.4e9 / (instPerWave * wavesPerSimd), ms, mx / (double(ms) * 1e3));
}

int main()
{
  unsigned long long* dTicks;
  float*              dSink;
  hipMalloc(&dTicks, sizeof(unsigned long long) * 256 * 8 * 4);
  hipMalloc(&dSink, sizeof(float) * 256 * 8 * 256);
#define RUN(K) for(int w : {1, 4, 8}) run(#K, K, w, dTicks, dSink);
  RUN(k_fma) RUN(k_pk_fma) RUN(k_pk_mul) RUN(k_pk_add) RUN(k_mul) RUN(k_sub) RUN(k_max) RUN(k_max3) RUN(k_min3) RUN(k_med3) RUN(k_cndmask) RUN(k_cndmask_sgpr) RUN(k_cmp_cndmask_x2) RUN(k_cndmask_fma_x2) RUN(k_cvt_ubyte1) RUN(k_cvt_ubyte0)
  RUN(k_cvt_sdwa) RUN(k_alignbit) RUN(k_and) RUN(k_or3) RUN(k_and_or) RUN(k_add_u32) RUN(k_lshl) RUN(k_lshl_add) RUN(k_bfe) RUN(k_bfm) RUN(k_mul_lo) RUN(k_mad_u24)
  RUN(k_cmp_vcc) RUN(k_cmp_sgpr) RUN(k_rcp) RUN(k_mov) RUN(k_max_dpp) RUN(k_bcnt) RUN(k_ffbh) RUN(k_perm) RUN(k_bpermute_waited) RUN(k_readlane) RUN(k_salu) RUN(k_fma_plus_salu)
  return 0;
}
