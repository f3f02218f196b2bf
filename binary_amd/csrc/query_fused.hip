// query_fused.hip — the single-pass query kernel (gfx950, wave64): count, prefix across workgroups, offsets and
// hit ids in one launch. Entry points behind it: bivx_query_dev / _f / _s (canonical CSR, optionally with ids
// ordered inside the kernel), bivx_query_dev_u (per-query begin/count, no cross-workgroup wait) and
// bivx_count_dev (zero-capacity buffer). Replaces a batch of IntervalTree::find_overlaps calls (reference
// interval_tree.hpp:306-334). Device building blocks: query_device.h.
#include <atomic>
#include <cstdlib>

#include "query_device.h"
#include "prefix_device.h"

namespace bivx {
namespace {

// ---- single-pass kernel ----------------------------------------------------------------------------------
// A workgroup owns kFTile = 1024 consecutive queries. It counts them (remembering each short window's hit
// mask and the ids of its first hits), publishes its hit total, sums the totals of ALL earlier tiles, then
// writes offsets and hit ids. The prefix is a wide sweep, not a serial look-back chain: on MI355X every poll of
// another XCD's status word goes to memory (per-XCD L2s are not coherent), so the number of dependent polls, not
// their width, is what costs. Launches of up to kFlatTiles tiles (1 M queries) sweep the tile words directly
// (four loads in flight per lane). Larger launches use two levels: tiles form groups of 64, a tile reads the
// words of the earlier tiles of its group (one load per lane) and the words of all earlier GROUPS, and the 64th
// tile of a group publishes the group's total as soon as it has its in-group sum. One launch covers up to
// kFMaxTiles tiles (64 M queries); larger batches run as consecutive launches, each starting from the running
// total its predecessor left in offsets[q_begin].
//   ws[kWsTicket] (low 32 bits): tile ticket. ws[kWsDone]: tiles that have left. ws[kWsStatus + g]: kStValid |
//   hits of group g; ws[kWsStatus + kFMaxGroups + t]: kStValid | hits of tile t. Each is written and polled as ONE
//   8-byte agent-scope atomic, so the value needs no separate fence. Tiles take tickets in launch order: every
//   predecessor of a polling tile is already resident, the wait cannot deadlock; spins are bounded anyway.
#ifndef BIVX_FUSED_THREADS
#define BIVX_FUSED_THREADS 1024
#endif
constexpr int kFThreads = BIVX_FUSED_THREADS;
#ifndef BIVX_FUSED_WAVES
#define BIVX_FUSED_WAVES 8  // waves per SIMD the register allocation is held to: 8 = two workgroups per CU, 64 VGPRs
#endif
constexpr int kFWaves = kFThreads / kWave;
// One query per thread. (More per thread was measured and did not pay: a wavefront here is latency-bound, and the
// output staging below relies on the 64 lists of a wavefront being adjacent.)
constexpr int kFTile = kFThreads;
constexpr uint32_t kStage = 512;     // ids a wavefront lays out in LDS per round before streaming them out
#ifndef BIVX_GATHER
#define BIVX_GATHER 8
#endif
constexpr uint32_t kGatherMax = BIVX_GATHER;  // ids a lane fetches per step when it replays a window in phase 2
static_assert(kMaxRec * kLight <= kStage / 2, "a replayed list must fit half the output stage");
#ifndef BIVX_STAGE_MIN
#define BIVX_STAGE_MIN 128
#endif
constexpr uint32_t kStageMin = BIVX_STAGE_MIN;  // ... when it has at least this many (2 per lane); below that lanes store directly
// Diagnostic build only (-DBIVX_STAMPS): per-tile wall-clock stamps (100 MHz constant counter) written to a
// buffer no other code reads; the product build has no stamp.
#ifdef BIVX_STAMPS
constexpr unsigned kStampTiles = 1024;
constexpr unsigned kStampSlots = 12;
__device__ unsigned long long g_stamps[kStampTiles * kStampSlots];
#define BIVX_STAMP(k) \
  if (threadIdx.x == 0) g_stamps[(blockIdx.x % kStampTiles) * kStampSlots + (k)] = __builtin_amdgcn_s_memrealtime()
#else
#define BIVX_STAMP(k)
#endif

// 8 waves per SIMD (two workgroups of 1024 threads per CU): keeps the kernel within 64 VGPRs.
// S: every query's ids leave in ascending order (sorted on their way through the output stage; no second pass).
// MS: the index has chromosomes with several segments; queries record up to kMaxRec windows for the replay.
// U: unordered output (bivx_query_dev_u). A tile reserves its output range with ONE atomic add on a running
//    total and waits for nobody: no ticket, no status words, no prefix sweep. `offsets` then receives begin[q]
//    (q words) and `counts` count[q]; ranges of different tiles lie in the buffer in whatever order the tiles
//    got there, inside a tile they are in query order. The last tile to leave stores the total in *total_out.
template <bool LDS_DESC, bool F, bool S, bool MS, bool U>
// (the filtered variants carry the filter's words on top: they are built for 4 waves per SIMD, one workgroup per CU,
//  which is what keeps them out of scratch; sv2nl batches are 1e4 - 1e5 records)
__global__ __launch_bounds__(kFThreads, F ? 4 : BIVX_FUSED_WAVES) void k_query_fused(IndexView v, const uint32_t *__restrict__ qchrom,
                                                           const uint32_t *__restrict__ qlow,
                                                           const uint32_t *__restrict__ qhigh, size_t q_begin,
                                                           size_t q_end, uint64_t *__restrict__ offsets,
                                                           uint32_t *__restrict__ hits, uint64_t cap,
                                                           uint64_t *__restrict__ ws, int flags,
                                                           uint32_t *__restrict__ counts,
                                                           uint64_t *__restrict__ total_out, uint32_t seq,
                                                           uint32_t skip_seq) {
  // launched behind k_query_pipe_dense (query_pipe.hip): that kernel did the launch's work if the order probe
  // left this number
  if (skip_seq != 0 && __hip_atomic_load(reinterpret_cast<const uint32_t *>(ws + kWsOrder), __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT) == skip_seq)
    return;
  const bool self_clean = (flags & kFlagSelfClean) != 0;
  __shared__ SegDesc s_seg[LDS_DESC ? kLdsSegs : 1];
  __shared__ uint2 s_cs[LDS_DESC ? kLdsChroms : 1];
  __shared__ uint32_t s_tile;
  __shared__ uint32_t s_wsum[kFWaves];
  __shared__ uint64_t s_wsum64[kFWaves];  // the same in 64 bits, for tiles with very long hit lists
  __shared__ uint64_t s_base;
  // ids of each query's first kKeepN hits (thread-private slots). Eight where LDS allows (1-2 % faster than four:
  // fewer ids are re-read in phase 2); the several-segment kernel spends that LDS on recorded windows instead.
  constexpr uint32_t kKeepN = MS ? 4 : 8;
  // rows the wavefront-cooperative path keeps in flight: the id-ordering variants are the tightest on registers
  // (four rows cost them 8 more bytes of scratch per lane and 4 % on configs 2-3)
  constexpr uint32_t kRowsN = S ? 1 : kRows;
  // ids fetched per replay step: the variants that carry more state (several recorded windows, id ordering, filter
  // words) fetch four, which is what keeps every one of them inside 64 VGPRs / 80 SGPRs without scratch
  constexpr uint32_t kGather = (MS || S || F) && kGatherMax > 4 ? 4 : kGatherMax;
  __shared__ uint4 s_keep[kFThreads * (kKeepN / 4)];
  __shared__ uint32_t s_out[kFWaves][kStage];  // per-wavefront staging of the output ids
  __shared__ uint4 s_xrec[MS ? kFThreads : 1];  // a query's third recorded window (thread-private slots)
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;

  BIVX_STAMP(0);
  // ordered output: tiles take tickets, so that every predecessor of a waiting tile is resident.
  // unordered output: tiles never wait for each other, any tile may be any block.
  if (threadIdx.x == 0)
    s_tile = U ? blockIdx.x : atomicAdd(reinterpret_cast<unsigned int *>(ws + kWsTicket), 1u);
  const SegDesc *segs;
  const uint2 *cs;
  stage_descriptors<LDS_DESC>(v, s_seg, s_cs, segs, cs);
  __syncthreads();
  const uint32_t tile = s_tile;
  uint64_t *group = ws + kWsStatus, *status = group + kFMaxGroups;
  BIVX_STAMP(1);
  // A ticket beyond the grid means the workspace was not zero when the launch began (a launch that died half-way,
  // a caller workspace that was not cleared): nothing this launch writes can be trusted. Say so and leave without
  // indexing the status array or the queries with the ticket.
  if (tile >= gridDim.x) {
    if (threadIdx.x == 0) raise_error(v.err, kErrWorkspace);
    return;
  }

  // phase 1: count the thread's query
  const size_t q = q_begin + (size_t)tile * kFThreads + threadIdx.x;
  Query qy = load_query<F>(v, cs, qchrom, qlow, qhigh, q, q < q_end);
  Replay rp;
  uint32_t *const kept = reinterpret_cast<uint32_t *>(&s_keep[threadIdx.x * (kKeepN / 4)]);
  uint32_t *const xrec = reinterpret_cast<uint32_t *>(&s_xrec[MS ? threadIdx.x : 0]);
  // the wavefront's slab for neighbouring windows: the 64 lanes' keep slots taken together (2 KiB = kSlabSlots
  // records); only the kernels whose chromosomes have one segment each use it
  constexpr bool kSlab = !MS && kKeepN * 4 * kWave >= kSlabSlots * 8;
  uint4 *const slab = &s_keep[(threadIdx.x & ~(kWave - 1)) * (kKeepN / 4)];
  const uint32_t cnt = enumerate_hits<Mode::Count, F, MS, kKeepN, kRowsN, kSlab>(v, segs, qy, nullptr, 0, 0, &rp, kept,
                                                                              xrec, slab);
  const uint32_t tsum = cnt;

  BIVX_STAMP(2);
  // workgroup exclusive scan of the per-thread sums
  const uint32_t incl = wave_scan_incl(tsum);
  if (lane == kWave - 1) s_wsum[wave] = incl;
  // 32-bit sums are exact while every query of the tile has fewer than 2^22 hits (1024 * 2^22 = 2^32). A tile
  // with a larger list (chromosome-wide queries on a very large index) redoes the scan in 64 bits.
  const bool wide = __syncthreads_or(tsum >= (1u << 22)) != 0;
  uint64_t total = 0, local = 0;
  if (!wide) {
    uint32_t wbase = 0, total32 = 0;
#pragma unroll
    for (int w = 0; w < kFWaves; ++w) {
      const uint32_t s = s_wsum[w];
      if (w < wave) wbase += s;
      total32 += s;
    }
    total = total32;
    local = wbase + incl - tsum;
  } else {
    uint64_t incl64 = tsum;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const uint64_t o = __shfl_up((unsigned long long)incl64, d, kWave);
      if (lane >= d) incl64 += o;
    }
    if (lane == kWave - 1) s_wsum64[wave] = incl64;
    __syncthreads();
    uint64_t wbase = 0;
    for (int w = 0; w < kFWaves; ++w) {
      const uint64_t s = s_wsum64[w];
      if (w < wave) wbase += s;
      total += s;
    }
    local = wbase + incl64 - tsum;
  }

  // prefix across tiles: wave 0 publishes this tile's total and sums every earlier tile's
  if (U) {
    if (threadIdx.x == 0)
      s_base = atomicAdd(reinterpret_cast<unsigned long long *>(ws + kWsTicket), (unsigned long long)total);
  } else if (wave == 0) {
    BIVX_STAMP(3);
    if (lane == 0) st_status(&status[tile], kStValid | (uint64_t)total);
    // Bounded by wall time, not by a poll count: a predecessor whose phase 1 walks chromosome-wide windows may
    // legitimately take seconds. When the bound expires the tile goes on with a wrong prefix — a hung GPU helps
    // nobody — and raises the error word, which no entry point lets pass as success.
    const uint32_t wl = ((uint32_t)flags >> kFlagWaitShift) & 0xFFu;
    const uint64_t wait_ticks = 1ull << (wl ? wl : kWaitLog2Default);
    auto wait_word = [&](const uint64_t *p, uint64_t w) -> uint64_t {
      if (!(w & kStValid)) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        for (uint32_t spins = 1;; ++spins) {
          __builtin_amdgcn_s_sleep(1);
          w = ld_status(p);
          if (w & kStValid) break;
          if ((spins & 15u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > wait_ticks) break;
        }
        if (!(w & kStValid)) raise_error(v.err, kErrTimeout);
      }
      return w & ~kStValid;
    };
    auto wave_total = [&](uint64_t x) -> uint64_t {
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) x += __shfl_xor((unsigned long long)x, d, kWave);
      return x;
    };
    // Launches of up to kFlatTiles tiles (1 M queries) sweep the tile words directly, one level: measured 3 us
    // faster there than two levels, whose second level is one more dependent round trip. Larger launches go
    // through the groups: the earlier tiles of this tile's group (one word per lane), then all earlier groups.
    const bool flat = gridDim.x <= kFlatTiles;
    const uint32_t g = tile >> 6, r = tile & 63u;
    const uint64_t *words = flat ? status : group;
    const uint32_t nwords = flat ? tile : g;
    // The loads of the first round of words and of the in-group word leave together: a sweep that finds everything
    // published costs one memory round trip, not one per level (the words are agent-scope atomics, which the
    // compiler keeps in program order).
    uint64_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t t = j * kWave + lane;
      w[j] = t < nwords ? ld_status(&words[t]) : kStValid;
    }
    uint64_t in_group = 0;
    if (!flat) {
      const uint64_t *mine = &status[(g << 6) + (uint32_t)lane];
      in_group = wave_total((uint32_t)lane < r ? wait_word(mine, ld_status(mine)) : 0ull);
      if (r == 63u && lane == 0) st_status(&group[g], kStValid | (in_group + total));
    }
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
        sum += t < nwords ? wait_word(&words[t], w[j]) : 0ull;
      }
    }
    sum = wave_total(sum) + in_group;
    if (lane == 0) s_base = sum + (q_begin ? offsets[q_begin] : 0ull);
    BIVX_STAMP(4);
  }
  __syncthreads();
  BIVX_STAMP(5);

  // phase 2: offsets and hit ids.
  // A query whose window was recorded replays its hit mask: the ids of its first hits wait in LDS, later ones
  // are re-read next to their records. When every lane of a wavefront replays (the common case), the ids are
  // first laid out in LDS exactly as they will sit in the output — the 64 lists are adjacent there — and then
  // streamed out with coalesced stores, kStage ids per round, instead of 64 lanes each storing 4 bytes at a time
  // into 64 different lines. Other wavefronts (several segments, long windows) enumerate again, directly.
  const uint64_t pos = s_base + local;
  {
    if (q < q_end) {
      stream_store(offsets + q, pos);
      if (U) counts[q] = cnt;
      else if (q == q_end - 1) offsets[q_end] = pos + cnt;
    }
    // The replay cursor walks the lane's recorded windows in segment order: `mrem` holds the bits of the current
    // window that are not consumed yet. replay(k0, k1, put) hands the ids of hits k0 .. k1-1 (consecutive calls
    // continue where the last one stopped) to put(k, id). kGather ids are fetched per step with all their loads in
    // flight together: one load per hit in a while-loop made every lane wait a full memory latency per id, which
    // was most of phase 2 when queries have ~16 hits.
    uint64_t mrem = rp.mask;
    uint32_t cur_al = rp.al, cur_rec = 1;
    bool cur_packed = rp.packed;
    auto replay = [&](uint32_t k0, uint32_t k1, auto put) {
      if (kSlab && rp.lds) {  // every id is in the wavefront's LDS slab: there are no loads to batch
        const uint2 *s2 = reinterpret_cast<const uint2 *>(slab) + (rp.al - rp.lbase);
        for (uint32_t k = k0; k < k1; ++k) {
          const uint32_t j = (uint32_t)__ffsll((long long)mrem) - 1u;
          mrem &= mrem - 1;
          put(k, s2[j].y);
        }
        return;
      }
      for (uint32_t k = k0; k < k1; k += kGather) {
        uint32_t slot[kGather], ids[kGather], pk = 0;
#pragma unroll
        for (uint32_t i = 0; i < kGather; ++i) {
          slot[i] = 0;
          if (k + i < k1) {
            if (MS) {
              while (mrem == 0 && cur_rec < kMaxRec) {  // next recorded window (there is one: k < the hit count)
                const uint32_t *w = cur_rec == 1 ? kept : xrec;
                cur_al = w[0] & ~1u;
                cur_packed = (w[0] & 1u) != 0;
                mrem = (uint64_t)w[1] | (uint64_t)w[2] << 32;
                ++cur_rec;
              }
            }
            slot[i] = (uint32_t)__ffsll((long long)mrem) - 1u;
            if (MS) {
              slot[i] += cur_al;
              pk |= (cur_packed ? 1u : 0u) << i;
            }
            mrem &= mrem - 1;
          }
        }
#pragma unroll
        for (uint32_t i = 0; i < kGather; ++i) {
          if (k + i < k1) {
            if (rp.kept && k + i < kKeepN) ids[i] = kept[k + i];
            else if (MS) ids[i] = (pk >> i & 1u) ? v.rec[slot[i]].y : v.id[slot[i]];
            else ids[i] = rp.packed ? v.rec[rp.al + slot[i]].y : v.id[rp.al + slot[i]];
          }
        }
#pragma unroll
        for (uint32_t i = 0; i < kGather; ++i)
          if (k + i < k1) put(k + i, ids[i]);
      }
    };
    const bool all_replay = __all(rp.ok);
    const uint64_t wpos0 = __shfl((unsigned long long)pos, 0, kWave);
    const uint32_t loff = (uint32_t)(pos - wpos0);
    const uint32_t wtotal = __shfl(loff + cnt, kWave - 1, kWave);
    if (cap == 0) {
      // a pure count (bivx_count_dev): the offsets are all that is asked for
    } else if (S && all_replay) {
      // Rounds of consecutive lanes whose lists fit half the stage together (a replayed list has at most
      // kMaxRec * kLight ids, which fits by itself): ids go to one half in slot order, every lane rank-sorts its own list into the other half, and that half is
      // streamed out coalesced.
      uint32_t *in = s_out[wave], *outb = s_out[wave] + kStage / 2;
      uint32_t first = 0;
      while (first < (uint32_t)kWave) {
        const uint32_t base = __shfl(loff, (int)first, kWave);
        const uint64_t fit = __ballot((uint32_t)lane >= first && loff + cnt - base <= kStage / 2);
        const uint64_t nofit = ~fit & (~0ull << first);
        const uint32_t next = nofit ? (uint32_t)__ffsll((long long)nofit) - 1u : (uint32_t)kWave;
        const bool mine = (uint32_t)lane >= first && (uint32_t)lane < next && cnt != 0;
        const uint32_t rel = loff - base;
        if (mine) replay(0u, cnt, [&](uint32_t k, uint32_t id) { in[rel + k] = id; });
        wave_sync_lds();
        if (mine) rank_sort_list<kFusedRankBlock>(in, outb, rel, cnt);
        wave_sync_lds();
        const uint32_t nthis = __shfl(loff + cnt, (int)next - 1, kWave) - base;
        for (uint32_t i = lane; i < nthis; i += kWave) {
          const uint64_t p = wpos0 + base + i;
          if (p < cap) stream_store(hits + p, outb[i]);
        }
        wave_sync_lds();
        first = next;
      }
    } else if (all_replay && wtotal >= kStageMin) {
      uint32_t *buf = s_out[wave];
      uint32_t kdone = 0;  // a lane's hits enter the stage in order, over one or more consecutive rounds
      for (uint32_t base = 0; base < wtotal; base += kStage) {
        if (kdone < cnt && loff < base + kStage) {
          const uint32_t room = base + kStage - loff;
          const uint32_t kend = cnt < room ? cnt : room;
          replay(kdone, kend, [&](uint32_t k, uint32_t id) { buf[loff + k - base] = id; });
          kdone = kend;
        }
        wave_sync_lds();
        const uint32_t nthis = wtotal - base < kStage ? wtotal - base : kStage;
        for (uint32_t i = lane; i < nthis; i += kWave) {
          const uint64_t p = wpos0 + base + i;
          if (p < cap) stream_store(hits + p, buf[i]);
        }
        wave_sync_lds();
      }
    } else {
      // few ids per lane (or a wavefront that holds general-path queries): every lane stores its own list
      if (rp.ok) {
        replay(0u, cnt, [&](uint32_t k, uint32_t id) {
          if (pos + k < cap) hits[pos + k] = id;
        });
        qy.nseg = 0;
      }
      if (!all_replay)
        (void)enumerate_hits<Mode::Fill, F, false, kKeep, kRowsN>(v, segs, qy, hits, pos, cap, nullptr);
      // A wavefront with general-path queries wrote its lists in index order. Ordering them here would pull the
      // whole bitonic machinery into this kernel (12-20 bytes of scratch per lane in every S variant); it asks the
      // conditional k_sort_hits launch that follows this kernel to do it instead (rare: long windows only).
      if (S && lane == 0)
        __hip_atomic_store(reinterpret_cast<uint32_t *>(ws + kWsNeedSort), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  BIVX_STAMP(8);   // wave 0 is through with its ids
  // self-cleaning workspace: every tile bumps `done` when it leaves (its sweep is long over); the tile that
  // sees gridDim.x - 1 knows nobody reads the words any more and zeroes them for the next launch. Off the
  // critical path: nothing waits for this but the end of the kernel.
  // Unordered output finds the last tile with the same word, and sums the tiles' totals in it on the way (one
  // atomic carries both: departures in the high bits, ids in the low kDoneShift bits), so the last tile knows the
  // launch total without reading a word other tiles are still adding to — no fence anywhere: an agent-scope fence
  // writes back and invalidates the XCD's whole L2 on this chip, which cost more than the prefix it replaced.
  // Only wave 0 takes part: the other wavefronts leave as soon as their ids are out (a barrier here kept all
  // sixteen for the departure atomic's round trip: 0.7-1.2 us per tile).
  if ((self_clean || U) && wave == 0) {
    uint32_t last = 0;
    uint64_t launch_total = 0;
    if (lane == 0) {
      if (U) {
        const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long *>(ws + kWsDone),
                                                 (1ull << kDoneShift) | (unsigned long long)total);
        if ((old >> kDoneShift) == gridDim.x - 1) {
          last = 1;
          launch_total = (old & ((1ull << kDoneShift) - 1)) + total;  // ids reserved by this launch
        }
      } else if (atomicAdd(reinterpret_cast<unsigned int *>(ws + kWsDone), 1u) == gridDim.x - 1) {
        last = 1;
      }
    }
    BIVX_STAMP(9);   // departure counted
    if (__shfl(last, 0, kWave)) {
      if (!U)
        for (uint32_t t = (uint32_t)lane; t < gridDim.x; t += kWave) {
          status[t] = 0;
          if (t < (gridDim.x + kWave - 1) / kWave) group[t] = 0;
        }
      if (lane == 0) {
        if (U) {  // running total over the call's launches; the reservation counter restarts after the last one
          const uint64_t sum = ws[kWsCarry] + launch_total;
          *total_out = sum;
          ws[kWsCarry] = (flags & kFlagFinal) ? 0 : sum;
        }
        if (!U || (flags & kFlagFinal)) ws[kWsTicket] = 0;
        ws[kWsDone] = 0;
      }
    }
  }
  BIVX_STAMP(6);
#ifdef BIVX_STAMPS
  if (threadIdx.x == 0) g_stamps[(blockIdx.x % kStampTiles) * kStampSlots + 7] = tile;
#endif
}

}  // namespace

size_t fused_workspace_bytes(size_t q) {
  (void)q;
  // header words, group words, tile words, then the pipelined kernels' list of slices left for k_fill_slices
  return ((size_t)kWsList + kWsListWords) * sizeof(uint64_t);
}

int launch_query_fused(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow,
                       const uint32_t *d_qhigh, size_t q, uint64_t *d_offsets, uint32_t *d_hits, uint64_t cap,
                       void *d_ws, bool self_clean, bool sort_ids, hipStream_t s, uint32_t *d_counts,
                       uint64_t *d_total) {
  const bool unordered = d_counts != nullptr;  // begin/count output, see k_query_fused
  if (q == 0) {
    BIVX_HIP(hipMemsetAsync(unordered ? d_total : d_offsets, 0, sizeof(uint64_t), s));
    return 0;
  }
  uint64_t *ws = static_cast<uint64_t *>(d_ws);
  // ordered output: a launch is limited to the tiles one prefix sweep covers; unordered output has no such limit
  // (only the departure count's 20 bits in ws[kWsDone])
  unsigned max_tiles = unordered ? (1u << 19) : kFMaxTiles, max_tiles_pipe = kFMaxTiles;
  if (const char *e = std::getenv("BIVX_MAX_TILES_PER_LAUNCH")) {  // test knob: forces the chained-launch path
    const long v = std::atol(e);
    if (v >= 1 && v < (long)max_tiles) max_tiles = (unsigned)v;
    if (v >= 1 && v < (long)max_tiles_pipe) max_tiles_pipe = (unsigned)v;
  }
  // the common case (one segment per chromosome, no filter, few ids per query) has a pipelined kernel with tiles of its
  // own size (begin / count output through it is one launch: there is no entry q_end to chain launches through)
  const size_t pipe_tile = pipe_queries_per_launch() / kFMaxTiles;
  const bool use_pipe = pipe_eligible(v, q, cap, sort_ids, unordered) && !(unordered && q > pipe_tile * max_tiles_pipe);
  // ... and so has the case of many ids per query if the batch is position-sorted, which a device-side probe finds out:
  // both kernels are launched, one of them returns at once
  const bool try_dense = !use_pipe && pipe_dense_eligible(v, q, cap, sort_ids, unordered);
  // ... and so has everything else that is large (several segments per chromosome, fused filters, many ids per query in
  // any order): a position-sorted batch is left to the dense kernel (which leaves the word for this one to see)
  const bool use_ms = !use_pipe && pipe_ms_eligible(v, q, cap, unordered);
  const size_t ms_tile = pipe_ms_queries_per_launch() / kFMaxTiles;
  // (a launch pair k_query_pipe_dense | k_query_pipe_ms is cut at the smaller of the two kernels' limits)
  const size_t per_launch = use_ms ? ms_tile * max_tiles_pipe
                                   : use_pipe || try_dense ? pipe_tile * max_tiles_pipe : (size_t)max_tiles * kFTile;
  // caller's workspace: zeroed in front of every launch (ordered output), or once per call (unordered output:
  // the running total lives in it across the call's launches)
  if (!self_clean && unordered && !use_pipe) BIVX_HIP(hipMemsetAsync(d_ws, 0, (size_t)kWsStatus * sizeof(uint64_t), s));
  for (size_t q0 = 0; q0 < q; q0 += per_launch) {
    const size_t q1 = q0 + per_launch < q ? q0 + per_launch : q;
    const size_t tile_q = use_pipe ? pipe_queries_per_launch() / kFMaxTiles : (size_t)kFTile;
    const unsigned tiles = (unsigned)((q1 - q0 + tile_q - 1) / tile_q);
    const size_t tile_small = use_ms ? ms_tile : use_pipe || try_dense ? pipe_queries_per_launch() / kFMaxTiles : (size_t)kFTile;
    if (!self_clean && (!unordered || use_pipe))
      BIVX_HIP(hipMemsetAsync(d_ws, 0, ((q1 - q0 + tile_small - 1) / tile_small + kFMaxGroups + kWsStatus) * sizeof(uint64_t), s));
    const dim3 grid(tiles), block(kFThreads);
    const bool lds = fits_lds(v), flt = v.flt_kind != BIVX_FILTER_NONE;
    // BIVX_PREFIX_WAIT_LOG2 (tests): bound of a prefix wait as log2 of 10 ns ticks; 1 makes every wait that is not
    // satisfied at once expire, which is how the error path is exercised
    int wait_log2 = 0;
    if (const char *e = std::getenv("BIVX_PREFIX_WAIT_LOG2")) {
      const long w = std::atol(e);
      if (w > 0 && w < 64) wait_log2 = (int)w;
    }
    const int flags = (self_clean ? kFlagSelfClean : 0) | (q1 == q ? kFlagFinal : 0) | (wait_log2 << kFlagWaitShift);
    // Ordering ids inside the kernel pays while a wavefront's 64 lists fit half its output stage (one round, all
    // lanes busy); the buffer capacity is the only bound on the hit count the host has. Denser results are
    // ordered by k_sort_hits afterwards, whose stage is eight times larger.
    const bool sort_inside = sort_ids && !unordered && !use_ms && cap <= (uint64_t)kFusedSortMaxAvg * q;
    static std::atomic<uint32_t> launch_seq{1};
    uint32_t seq = launch_seq.fetch_add(1);
    if (seq == 0) seq = launch_seq.fetch_add(1);  // 0 is what a cleared workspace holds
    uint32_t skip_seq = 0;
    if (try_dense) {
      if (int rc = launch_query_pipe_dense(v, d_qchrom, d_qlow, d_qhigh, q0, q1, d_offsets, d_hits, cap, ws, flags, seq, s))
        return rc;
      skip_seq = seq;
    }
    if (use_pipe) {
      if (int rc = launch_query_pipe(v, d_qchrom, d_qlow, d_qhigh, q0, q1, d_offsets, d_hits, cap, ws, flags,
                                     sort_ids ? seq : 0u, d_counts, d_total, s))
        return rc;
    } else if (use_ms) {
      if (int rc = launch_query_pipe_ms(v, d_qchrom, d_qlow, d_qhigh, q0, q1, d_offsets, d_hits, cap, ws, flags, skip_seq, s))
        return rc;
    } else {
#define BIVX_LAUNCH_FUSED_V(L, FL, SO, MSV, UV)                                                               \
  hipLaunchKernelGGL((k_query_fused<L, FL, SO, MSV, UV>), grid, block, 0, s, v, d_qchrom, d_qlow, d_qhigh, q0, \
                     q1, d_offsets, d_hits, cap, ws, flags, d_counts, d_total, seq, skip_seq)
#define BIVX_LAUNCH_FUSED(L, FL, SO)                       \
  if (unordered) {                                         \
    if (v.max_segs > 1)                                    \
      BIVX_LAUNCH_FUSED_V(L, FL, false, true, true);       \
    else                                                   \
      BIVX_LAUNCH_FUSED_V(L, FL, false, false, true);      \
  } else if (v.max_segs > 1) {                             \
    BIVX_LAUNCH_FUSED_V(L, FL, SO, true, false);           \
  } else {                                                 \
    BIVX_LAUNCH_FUSED_V(L, FL, SO, false, false);          \
  }
#ifdef BIVX_ONLY_MAIN  // development builds (tools/resource_usage.py): only the headline instantiation
    BIVX_LAUNCH_FUSED_V(true, false, false, false, false);
#else
    switch ((lds ? 4 : 0) | (flt ? 2 : 0) | (sort_inside ? 1 : 0)) {
      case 0: BIVX_LAUNCH_FUSED(false, false, false); break;
      case 1: BIVX_LAUNCH_FUSED(false, false, true); break;
      case 2: BIVX_LAUNCH_FUSED(false, true, false); break;
      case 3: BIVX_LAUNCH_FUSED(false, true, true); break;
      case 4: BIVX_LAUNCH_FUSED(true, false, false); break;
      case 5: BIVX_LAUNCH_FUSED(true, false, true); break;
      case 6: BIVX_LAUNCH_FUSED(true, true, false); break;
      default: BIVX_LAUNCH_FUSED(true, true, true); break;
    }
#endif
#undef BIVX_LAUNCH_FUSED
#undef BIVX_LAUNCH_FUSED_V
    }
    if (sort_ids && !unordered && !use_pipe) {  // (the pipelined kernel and k_fill_slices order every list themselves)
      BIVX_HIP(hipGetLastError());
      // ordered inside the kernel: the pass only runs if a wavefront asked for it (it compares the word with seq)
      if (int rc = launch_sort_hits(d_offsets + q0, d_hits, q1 - q0, cap, s,
                                    sort_inside ? reinterpret_cast<const uint32_t *>(ws + kWsNeedSort) : nullptr, seq))
        return rc;
    }
  }
  BIVX_HIP(hipGetLastError());
  return 0;
}

#ifdef BIVX_STAMPS
extern "C" int bivx_debug_stamps(unsigned long long *out, size_t n) {
  if (n > (size_t)kStampTiles * kStampSlots) n = (size_t)kStampTiles * kStampSlots;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace bivx
