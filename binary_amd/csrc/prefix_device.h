// prefix_device.h — the cross-workgroup prefix protocol shared by the single-pass kernels (query_fused.hip,
// query_pipe.hip): workspace layout, status words, bounded waits, error words. gfx950 only.
#ifndef BIVX_PREFIX_DEVICE_H_
#define BIVX_PREFIX_DEVICE_H_

#include "common.h"

namespace bivx {
namespace {

constexpr unsigned kFMaxTiles = 65536;                 // tiles of 1024 queries per launch (ordered output)
constexpr unsigned kFMaxGroups = kFMaxTiles / kWave;   // groups of 64 tiles
constexpr unsigned kFlatTiles = 1024;                  // launches up to this many tiles sweep the tile words directly
constexpr uint64_t kStValid = 1ull << 63;
// workspace words: the two counters and the status array sit on cache lines of their own, so that the atomics
// on the counters do not queue behind (or in front of) the sweeps' polls of the first status words
constexpr uint32_t kWsTicket = 0, kWsCarry = 8, kWsDone = 16, kWsNeedSort = 24, kWsTodo = 26, kWsTodoDone = 28,
                   kWsOrder = 30, kWsStatus = 32;
// ws[kWsTodo]: tiles the pipelined kernel left for k_fill_tiles (count); their numbers follow the status words
// ws[kWsOrder]: sequence number of the last launch whose batch k_probe_order found position-sorted (query_pipe.hip)
// ws[kWsNeedSort]: sequence number of the last launch that left lists for k_sort_hits to order (never cleared:
// every launch carries a fresh number)
// the pipelined kernels' list of slices behind the status array (u64 index; u32 entries, 16 per tile: a tile has 15
// slices and in the worst case every one of them is listed)
constexpr uint32_t kWsList = kWsStatus + kFMaxGroups + kFMaxTiles, kWsListWords = kFMaxTiles * 8;
constexpr uint32_t kDoneShift = 44;  // unordered output: ws[kWsDone] = departures << 44 | ids reserved by this launch
// k_query_fused flags: index-owned workspace; last launch of the call; bits 8-15: log2 of the bound on a prefix
// wait in ticks of the 100 MHz constant clock (0 = kWaitLog2Default)
constexpr int kFlagSelfClean = 1, kFlagFinal = 2, kFlagWaitShift = 8;
constexpr uint32_t kWaitLog2Default = 31;  // 2^31 x 10 ns = 21 s: only a device that stopped making progress gets there

// A workgroup that cannot produce a valid result says so in the index's error block (host memory mapped into the
// device: the host reads it without a copy after any synchronisation) instead of returning quietly; every
// synchronising entry point and bivx_stream_status turn a raised word into BIVX_E_TIMEOUT (capi.hip).
__device__ __forceinline__ void raise_error(uint32_t *err, uint32_t which) {
  __hip_atomic_store(err + which, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ uint64_t ld_status(const uint64_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_status(uint64_t *p, uint64_t w) {
  __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


__device__ __forceinline__ uint64_t wave_total64(uint64_t x) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) x += __shfl_xor((unsigned long long)x, d, kWave);
  return x;
}

// ---- the sweeps as functions (query_pipe.hip; query_fused.hip keeps its inline form) -------------------------
struct PrefixCtx {
  uint64_t *status, *group;
  uint32_t ntiles;
  uint64_t wait_ticks;
  uint32_t *err;
};

// Bounded by wall time, not by a poll count: a predecessor whose counting walks chromosome-wide windows may
// legitimately take seconds. When the bound expires the tile goes on with a wrong prefix — a hung GPU helps
// nobody — and raises the error word, which no entry point lets pass as success.
__device__ __forceinline__ uint64_t wait_word(const PrefixCtx &c, const uint64_t *p, uint64_t w) {
  if (!(w & kStValid)) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t spins = 1;; ++spins) {
      __builtin_amdgcn_s_sleep(1);
      w = ld_status(p);
      if (w & kStValid) break;
      if ((spins & 15u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > c.wait_ticks) break;
    }
    if (!(w & kStValid)) raise_error(c.err, kErrTimeout);
  }
  return w & ~kStValid;
}

// Hits of the earlier tiles of `tile`'s group (two-level launches), summed by one wavefront.
__device__ __forceinline__ uint64_t sum_in_group(const PrefixCtx &c, uint32_t tile, int lane) {
  const uint32_t g = tile >> 6, r = tile & 63u;
  const uint64_t *mine = &c.status[(g << 6) + (uint32_t)lane];
  return wave_total64((uint32_t)lane < r ? wait_word(c, mine, ld_status(mine)) : 0ull);
}

// One wavefront publishes the total of `tile` (and, for the 64th tile of a group, the group's).
__device__ __forceinline__ void publish_tile(const PrefixCtx &c, uint32_t tile, uint64_t total, int lane) {
  if (lane == 0) st_status(&c.status[tile], kStValid | total);
  if (c.ntiles > kFlatTiles && (tile & 63u) == 63u) {
    const uint64_t in_group = sum_in_group(c, tile, lane);
    if (lane == 0) st_status(&c.group[tile >> 6], kStValid | (in_group + total));
  }
}

// Hits of all tiles before `tile`, summed by one wavefront (every lane returns the sum). Launches of up to
// kFlatTiles tiles sweep the tile words directly; larger ones go through the groups of 64. The first round of
// words and the in-group word are loaded together: one memory round trip when everything is published.
__device__ __forceinline__ uint64_t tiles_before(const PrefixCtx &c, uint32_t tile, int lane) {
  const bool flat = c.ntiles <= kFlatTiles;
  const uint64_t *words = flat ? c.status : c.group;
  const uint32_t nwords = flat ? tile : tile >> 6;
  uint64_t w[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t t = j * kWave + lane;
    w[j] = t < nwords ? ld_status(&words[t]) : kStValid;
  }
  const uint64_t in_group = flat ? 0ull : sum_in_group(c, tile, lane);
  uint64_t sum = 0;
  for (uint32_t t0 = 0; t0 < nwords; t0 += 4 * kWave) {
    if (t0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t t = t0 + j * kWave + lane;
        w[j] = t < nwords ? ld_status(&words[t]) : kStValid;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t t = t0 + j * kWave + lane;
      sum += t < nwords ? wait_word(c, &words[t], w[j]) : 0ull;
    }
  }
  return wave_total64(sum) + in_group;
}

// ---- the same without waiting (query_pipe.hip's service wavefront polls): false if a word is not published yet ---
__device__ __forceinline__ bool try_sum_in_group(const PrefixCtx &c, uint32_t tile, int lane, uint64_t &sum) {
  const uint32_t g = tile >> 6, r = tile & 63u;
  const uint64_t w = (uint32_t)lane < r ? ld_status(&c.status[(g << 6) + (uint32_t)lane]) : kStValid;
  if (!__all((w & kStValid) != 0)) return false;
  sum = wave_total64(w & ~kStValid);
  return true;
}

__device__ __forceinline__ bool try_tiles_before(const PrefixCtx &c, uint32_t tile, int lane, uint64_t &sum) {
  const bool flat = c.ntiles <= kFlatTiles;
  const uint64_t *words = flat ? c.status : c.group;
  const uint32_t nwords = flat ? tile : tile >> 6;
  uint64_t acc = 0;
  for (uint32_t t0 = 0; t0 < nwords; t0 += 4 * kWave) {
    uint64_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t t = t0 + j * kWave + lane;
      w[j] = t < nwords ? ld_status(&words[t]) : kStValid;
    }
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ok = ok && (w[j] & kStValid) != 0;
      acc += w[j] & ~kStValid;
    }
    if (!__all(ok)) return false;
  }
  uint64_t in_group = 0;
  if (!flat && !try_sum_in_group(c, tile, lane, in_group)) return false;
  sum = wave_total64(acc) + in_group;
  return true;
}

}  // namespace
}  // namespace bivx

#endif  // BIVX_PREFIX_DEVICE_H_
