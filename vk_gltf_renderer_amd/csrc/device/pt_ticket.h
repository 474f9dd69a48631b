// The one place that states how the builder's bottom-up phases order cross-thread data around an arrival ticket
// (refit of bvh_reinsert.h, k_dp_solve of bvh8.hip): "the second thread to arrive at a node continues upward and reads what the first one wrote".
//
// Data words written by one thread and read by another go through AGENT-scope relaxed atomic stores / loads (global_store / global_load with sc1: they
// bypass the eight per-XCD L2s, which are not coherent with each other, and meet at the memory side).  What is left to order is
//   release:  this thread's data stores are ACKNOWLEDGED before its ticket (an agent-scope relaxed fetch_add) is performed;
//   acquire:  the ticket's value has RETURNED before any data load below it is issued.
// On gfx90a / gfx942 / gfx950 a wave's stores and loads are both counted by vmcnt, sc1 stores are write-through, and the compiler does not move a
// monotonic atomic across the s_waitcnt builtin, so `s_waitcnt vmcnt(0) lgkmcnt(0) expcnt(0)` on either side of the ticket is exactly that -- without
// the L2 write-back + invalidate a device-scope fence costs per thread and tree level (refit pass 3.93 -> 1.30 ms, profiles/r05_refit_ordering.txt).
// That argument is specific to those targets (gfx10+ counts stores in vscnt, and nothing here is a fence in the memory model's sense), so every other
// target -- and -DMI_TICKET_FENCES -- gets the memory model's own agent-scope release / acquire fences instead: same records, slower.
#pragma once
#include <hip/hip_runtime.h>

namespace pt {

#if defined(__HIP_DEVICE_COMPILE__)
#if (defined(__gfx90a__) || defined(__gfx942__) || defined(__gfx950__)) && !defined(MI_TICKET_FENCES)
__device__ __forceinline__ void ticketRelease() { __builtin_amdgcn_s_waitcnt(0); }
__device__ __forceinline__ void ticketAcquire() { __builtin_amdgcn_s_waitcnt(0); }
#else
__device__ __forceinline__ void ticketRelease() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
__device__ __forceinline__ void ticketAcquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
#endif
// Arrival ticket of a node: returns how many threads arrived before this one.
__device__ __forceinline__ unsigned int ticketArrive(unsigned int* p)
{
  ticketRelease();
  const unsigned int t = __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  ticketAcquire();
  return t;
}
#elif defined(__HIPCC__)
__device__ unsigned int ticketArrive(unsigned int* p);  // (host pass of hipcc: kernels are parsed, never emitted)
#endif

}  // namespace pt
