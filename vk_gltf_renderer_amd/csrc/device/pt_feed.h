// Wave-level dynamic work distribution of the persistent trace kernels (pt_kernels.hip); also exercised stand-alone by
// tools/test_feed.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#ifndef PT_DEV
#define PT_DEV __device__ __forceinline__
#endif

namespace pt {

PT_DEV uint32_t feedLaneId() { return __lane_id(); }

// Wave-level dynamic work fetch (persistent waves): a wave owns a private range [cur, end) of flat queue indices and takes
// the next piece from one of 8 head counters when the range runs dry.  Head h serves the h-th eighth of the queue (its own
// XCD's first) and counts RAYS; a fetch is ONE agent-scope atomicAdd on a line shared by all XCDs -- served beyond the XCD's
// L2, several microseconds each (and the XCDs' L2s are not coherent, so peeking at a head is just as expensive) -- so the
// piece size is guided by what the wave learnt from its own previous fetch: 1/(2 x waves per head) of what its home head had
// left, between one wave-load (64) and 1024 rays; pieces stolen from other heads are small.  Long queues start with big
// pieces (few atomics), every queue ends with small ones (no tail of one wave grinding through a large piece), queues
// shorter than the machine spread one wave-load per wave.
struct WaveFeed
{
  uint32_t cur, end, total, share;  // share = rays per head (multiple of 64)
  uint32_t deadHeads;               // heads this wave has seen exhausted
  uint32_t homeSeen;                // progress of the home head as of this wave's last fetch from it
  bool     exhausted;
};
PT_DEV void feedInit(WaveFeed& f, uint32_t total)
{
  f.cur = f.end = 0;
  f.total       = total;
  f.share       = ((total + 7u) / 8u + 63u) & ~63u;
  f.deadHeads   = 0;
  f.homeSeen    = 0;
  f.exhausted   = total == 0;
}
// blocks beyond the ones the queue can feed (one wave per 64 rays) leave at once
PT_DEV bool feedBlockHasWork(const WaveFeed& f) { return blockIdx.x * (blockDim.x / 64u) < (f.total + 63u) / 64u; }
PT_DEV bool feedNextChunk(WaveFeed& f, uint32_t* heads)
{
  uint32_t start = 0xffffffffu, count = 0, dead = f.deadHeads, homeSeen = f.homeSeen;
  if(feedLaneId() == 0)
  {
    const uint32_t h0          = blockIdx.x & 7u;
    const uint32_t wavesPerHead = max(1u, gridDim.x * (blockDim.x / 64u) / 8u);
    for(uint32_t k = 0; k < 8u && start == 0xffffffffu; ++k)
    {
      const uint32_t h = (h0 + k) & 7u;
      if(dead & (1u << h))
        continue;
      const uint32_t base = h * f.share;
      uint32_t       len  = 0u;  // (written with branches on purpose: `base >= total ? 0 : min(share, total - base)` lost its guard in
                                 //  the optimiser inside the trace kernels and handed out ranges beyond the end of the queue)
      if(base < f.total)
      {
        len = f.total - base;
        if(len > f.share)
          len = f.share;
      }
      uint32_t want = 128u;  // stolen from another XCD's head: small
      if(k == 0u)
      {
        want = ((len - min(homeSeen, len)) / (2u * wavesPerHead)) & ~63u;
        want = min(max(want, 64u), 1024u);
      }
      // Heads are looked at before they are touched (a coherent load; loads do not queue up behind each other the way
      // thousands of atomics on one exhausted counter do at the end of every launch) ...
      // The home head is taken from blindly while the wave knows that plenty is left, and looked at first near its end.
      const bool blind = k == 0u && len - min(homeSeen, len) > 32u * wavesPerHead;
      if(len != 0u && (blind || __hip_atomic_load(&heads[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < len))
      {
        const uint32_t s0 = atomicAdd(&heads[h], want);
        if(k == 0u)
          homeSeen = s0 + want;
        if(s0 < len)
        {
          start = base + s0;
          count = min(want, len - s0);
          break;
        }
      }
      dead |= 1u << h;
    }
  }
  start       = uint32_t(__shfl(int(start), 0));
  count       = uint32_t(__shfl(int(count), 0));
  f.deadHeads = uint32_t(__shfl(int(dead), 0));
  f.homeSeen  = uint32_t(__shfl(int(homeSeen), 0));
  if(start == 0xffffffffu)
  {
    f.exhausted = true;
    return false;
  }
  f.cur = start;
  f.end = start + count;
  return true;
}
// Hands flat indices to the lanes whose `idle` predicate is set; returns the index or 0xffffffff (none left for this lane).
PT_DEV uint32_t feedTake(WaveFeed& f, bool idle, uint32_t* heads)
{
  unsigned long long mask = __ballot(idle);
  uint32_t           need = uint32_t(__popcll(mask));
  uint32_t           rank = uint32_t(__popcll(mask & ((1ull << feedLaneId()) - 1ull)));
  uint32_t           mine = 0xffffffffu, assigned = 0;
  while(need > 0 && !f.exhausted)
  {
    if(f.cur == f.end && !feedNextChunk(f, heads))
      break;
    uint32_t take = min(need, f.end - f.cur);
    if(idle && rank >= assigned && rank < assigned + take)
      mine = f.cur + (rank - assigned);
    assigned += take;
    f.cur += take;
    need -= take;
  }
  return mine;
}

}  // namespace pt
