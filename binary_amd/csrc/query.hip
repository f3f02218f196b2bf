// query.hip — batched overlap queries against the built index (gfx950, wave64).
//
// Replaces IntervalTree::find_overlaps / find_overlap (reference interval_tree.hpp:290-334) for a batch.
// One lane per query. Per (chromosome, length class) segment ONE 16-byte bucket-directory load bounds the
// candidate window [a, b): a = first slot of the cell holding x = q.low - maxlen, b = end of the cell
// holding q.high. Every candidate is then tested with the reference predicate low <= q.high &&
// q.low <= high; no binary search is needed because slots outside [x, q.high] fail the predicate by
// themselves. All candidate loads are independent of each other (16-byte loads, several in flight).
//   - packed segments: 8-byte records (low's 16 low bits | 16-bit length, id), two per load: the line that says
//     "hit" also carries the id
//   - other segments:  8-byte (low, high) pairs, two per load; ids in a parallel array
//   - windows longer than kLight slots are read by the whole wavefront (coalesced rows, __ballot compaction),
//     so a chromosome-scale query costs O(window / 64) wave steps instead of stalling one lane; very long
//     ones are first trimmed by a wavefront 64-ary search (__ballot picks the gap)
// Segment descriptors are staged through LDS. Integer compare/index work only: no MFMA anywhere.
//
// Two ways in:  k_query<Count|Fill|Any>  (two-pass API: count -> offsets scan -> fill), and
//               k_query_fused            (single pass: count, prefix across workgroups, fill into a
//                                         caller-sized buffer).
#include "common.h"

namespace bivx {
namespace {

constexpr int kQThreads = 256;
constexpr int kQWaves = kQThreads / kWave;
constexpr uint32_t kLight = 64;      // window slots a lane reads by itself; longer windows go to the wavefront
#ifndef BIVX_TRIM
#define BIVX_TRIM 512
#endif
constexpr uint32_t kTrim = BIVX_TRIM;  // wavefront windows longer than this are first trimmed by a 64-ary search
constexpr uint32_t kRowsWide = 4;    // rows of 64 slots the wavefront-cooperative path keeps in flight ...
constexpr uint32_t kRowsLean = 1;    // ... and in the one-segment single-pass kernel, which must stay spill-free in 64 VGPRs
constexpr uint32_t kLdsSegs = 128;   // descriptors staged in LDS (4 KiB) ...
constexpr uint32_t kLdsChroms = 511; // ... with chrom_seg (2 KiB); larger indexes read them from global

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ SegDesc load_seg(const SegDesc *p) {
  const uint4 *q = reinterpret_cast<const uint4 *>(p);
  const uint4 u = q[0], w = q[1];
  SegDesc d;
  d.begin = u.x; d.end = u.y; d.base = u.z; d.last = u.w;
  d.shift = w.x; d.table_off = w.y; d.maxlen = w.z; d.ncell = w.w;
  return d;
}

// Candidate window of query [lo, hi] in one segment, straight from the bucket directory.
// Every hit has low <= hi and low >= high - maxlen >= lo - maxlen (entries with low > high count as length 0
// and obey the same bound), so all hits lie in the cells ca .. cb-1; slots of the two edge cells that are
// not hits fail the predicate, which is evaluated on every candidate anyway.
struct Window {
  uint32_t a, b;        // candidate slots [a, b)
  uint32_t cell0_low;   // coordinate of the start of cell ca: every candidate's low is >= it
  uint32_t span;        // number of cells, 0 = empty window
  bool narrow;          // the window's cells cover at most 65536 coordinates (packed records are decodable)
};

__device__ __forceinline__ Window seg_window(const IndexView &v, const SegDesc &d, uint32_t lo, uint32_t hi) {
  Window w{0u, 0u, 0u, 0u, false};
  const uint32_t x = lo > d.maxlen ? lo - d.maxlen : 0u;
  if (hi < d.base || x > d.last || hi < x) return w;
  const uint32_t sh = d.shift & 31u;
  const uint32_t *t = v.table + d.table_off;
  const uint32_t ca = x <= d.base ? 0u : (x - d.base) >> sh;
  const uint32_t cb = hi >= d.last ? d.ncell : ((hi - d.base) >> sh) + 1u;
  // directory entries ca .. ca+3 in one 16-byte load (4-byte aligned; the table carries 3 spare entries)
  const u32x4_a4 tq = *reinterpret_cast<const u32x4_a4 *>(t + ca);
  w.span = cb - ca;
  w.a = tq.x;
  w.b = w.span == 1 ? tq.y : w.span == 2 ? tq.z : w.span == 3 ? tq.w : t[cb];
  w.cell0_low = d.base + (ca << sh);
  w.narrow = ((uint64_t)w.span << sh) <= 65536ull;
  return w;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t x) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d, kWave);
  return x;
}
__device__ __forceinline__ uint32_t wave_min(uint32_t x) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) x = min(x, (uint32_t)__shfl_xor(x, d, kWave));
  return x;
}

// ---- fused post-filters (include/bivx.h, bivx_filter) -----------------------------------------------------------
// sv2nl's per-mapper check_condition, applied to a candidate that already overlaps the query; (lo, hi) is the
// validated NL record, (low, high) the SV record as stored in the tree (reference standalone/sv2nl/):
//   DUP  source/mapper.cpp:50-55   is_contained(sv, nl) && distance_less(nl, sv, d)        (helper.hpp:16-40)
//   INV  source/mapper.cpp:57-79   neither contains the other, within d, then the strand side rule
//   TRA  source/mapper.cpp:144-156 same ordered chromosome pair and both breakpoints within d (helper.hpp:76-82)
__device__ __forceinline__ uint32_t absdiff(uint32_t a, uint32_t b) { return a >= b ? a - b : b - a; }

__device__ __forceinline__ bool filter_accept(const IndexView &v, uint32_t lo, uint32_t hi, uint32_t qaux, uint32_t low,
                                              uint32_t high, uint32_t id) {
  const uint32_t d = v.flt_dist;
  if (v.flt_kind == BIVX_FILTER_SV2NL_TRA) {
    const uint32_t ia = v.flt_iaux[id];
    if ((ia >> 1) != (qaux >> 1)) return false;
    const uint32_t q1 = (qaux & 1u) ? hi : lo, q2 = (qaux & 1u) ? lo : hi;
    const uint32_t i1 = (ia & 1u) ? high : low, i2 = (ia & 1u) ? low : high;
    return absdiff(q1, i1) <= d && absdiff(q2, i2) <= d;
  }
  const bool sv_has_nl = low <= lo && high >= hi;
  const bool near = absdiff(lo, low) <= d && absdiff(hi, high) <= d;
  if (v.flt_kind == BIVX_FILTER_SV2NL_DUP) return sv_has_nl && near;
  // INV
  const bool nl_has_sv = lo <= low && hi >= high;
  if (sv_has_nl || nl_has_sv || !near) return false;
  if (!v.flt_strand) return true;
  const bool s1 = (qaux & 1u) != 0, s2 = (qaux & 2u) != 0;
  return lo <= low ? (s1 && !s2) : (!s1 && s2);
}

// hit mask of a short window over 8-byte (low, high) pairs; bit j <-> slot al + j, al = a rounded down to 2
template <bool F>
__device__ __forceinline__ uint64_t light_mask_pairs(const IndexView &v, uint32_t a, uint32_t b, uint32_t lo,
                                                     uint32_t hi, uint32_t qaux, uint32_t &al) {
  const uint4 *pairs = reinterpret_cast<const uint4 *>(v.se);
  al = a & ~1u;
  uint64_t mask = 0;
#pragma unroll 1
  for (uint32_t c0 = 0; c0 < kLight; c0 += 8) {
    if (al + c0 < b) {
      uint4 r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t s = al + c0 + 2 * j;
        if (s < b) r[j] = pairs[s >> 1];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t s = al + c0 + 2 * j;
        if (s < b) {
          if (s >= a && r[j].x <= hi && r[j].y >= lo &&
              (!F || filter_accept(v, lo, hi, qaux, r[j].x, r[j].y, v.id[s])))
            mask |= 1ull << (c0 + 2 * j);
          if (s + 1 < b && r[j].z <= hi && r[j].w >= lo &&
              (!F || filter_accept(v, lo, hi, qaux, r[j].z, r[j].w, v.id[s + 1])))
            mask |= 1ull << (c0 + 2 * j + 1);
        }
      }
    }
  }
  return mask;
}

// the same over packed records. A packed record is 8 bytes: (low & 0xFFFF | (high - low) << 16, id) — the
// interval's low 16 coordinate bits and its length, and its append-order id right beside it, so the cache line
// that answers "is it a hit" also says which interval it is. A window whose cells cover at most 65536
// coordinates decodes low uniquely: low = cell0_low + ((record - cell0_low) & 0xFFFF).
// bit j <-> slot al + j, al = a rounded down to 2. If `keep` is given, the ids of the first kKeep hits are
// written there (ascending slot order) as they are found.
constexpr uint32_t kKeep = 4;

// one chunk = 8 consecutive slots starting at the even slot c = al + c0, as four 16-byte loads
__device__ __forceinline__ void packed_load_chunk(const IndexView &v, uint32_t c, uint32_t b, uint4 (&r)[4]) {
  const uint4 *pairs = reinterpret_cast<const uint4 *>(v.rec);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t s = c + 2 * j;
    if (s < b) r[j] = pairs[s >> 1];
  }
}

// evaluates the predicate on a loaded chunk; returns the chunk's 8-bit hit mask (bit k <-> slot c + k)
template <bool F>
__device__ __forceinline__ uint32_t packed_eval_chunk(const IndexView &v, const Window &w, uint32_t lo,
                                                      uint32_t hi, uint32_t qaux, uint32_t c, const uint4 (&r)[4],
                                                      uint32_t *keep, uint32_t &n) {
  uint32_t m = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t s = c + 2 * j;
    if (s < w.b) {
      const uint32_t rr[2] = {r[j].x, r[j].z}, ii[2] = {r[j].y, r[j].w};
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const uint32_t i = s + e;
        // the record keeps low's 16 low bits; every candidate's low lies in [cell0_low, cell0_low + 65536)
        const uint32_t low = w.cell0_low + ((rr[e] - w.cell0_low) & 0xFFFFu);
        const uint32_t high = low + (rr[e] >> 16);
        if (i >= w.a && i < w.b && low <= hi && high >= lo &&
            (!F || filter_accept(v, lo, hi, qaux, low, high, ii[e]))) {
          m |= 1u << (2 * j + e);
          if (keep) {
            if (n < kKeep) keep[n] = ii[e];
            ++n;
          }
        }
      }
    }
  }
  return m;
}

template <bool F>
__device__ __forceinline__ uint64_t light_mask_packed(const IndexView &v, const Window &w, uint32_t lo,
                                                      uint32_t hi, uint32_t qaux, uint32_t &al, uint32_t *keep) {
  al = w.a & ~1u;
  uint64_t mask = 0;
  uint32_t n = 0;
#pragma unroll 1
  for (uint32_t c0 = 0; c0 < kLight; c0 += 8) {
    if (al + c0 < w.b) {
      uint4 r[4];
      packed_load_chunk(v, al + c0, w.b, r);
      mask |= (uint64_t)packed_eval_chunk<F>(v, w, lo, hi, qaux, al + c0, r, keep, n) << c0;
    }
  }
  return mask;
}

// First slot in [a, b) whose low is >= x (b if there is none), found by the whole wavefront: a 64-ary search — the
// 64 lanes probe 64 evenly spaced slots, one __ballot tells which gap holds the answer, repeat. Used to trim long
// candidate windows (many intervals starting inside one directory cell) to the slots whose low lies in
// [q.low - maxlen, q.high] before they are scanned. All 64 lanes must call it with the same arguments.
__device__ __forceinline__ uint32_t wave_lower_bound_low(const uint2 *se, uint32_t a, uint32_t b, uint32_t x, int lane) {
  while (b - a > (uint32_t)kWave) {
    const uint32_t step = (b - a + kWave - 1) / kWave;
    const uint32_t p = a + step * (uint32_t)lane;
    const bool ge = p < b ? se[p].x >= x : true;  // lows ascend inside a segment: the ballot is 0..01..1
    const uint64_t m = __ballot(ge);
    const uint32_t first = m ? (uint32_t)__ffsll((long long)m) - 1u : (uint32_t)kWave;
    const uint32_t nb = first < (uint32_t)kWave ? min(a + step * first, b) : b;
    const uint32_t na = first > 0 ? a + step * (first - 1) + 1 : a;
    a = na;
    b = nb;
  }
  const uint32_t p = a + (uint32_t)lane;
  const uint64_t m = __ballot(p < b ? se[p].x >= x : true);
  const uint32_t first = m ? (uint32_t)__ffsll((long long)m) - 1u : (uint32_t)kWave;
  return min(a + first, b);
}

// orders one wavefront's LDS / global accesses: what lanes wrote before is visible to all lanes after
__device__ __forceinline__ void wave_sync_mem() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

enum class Mode { Count, Fill, Any };

// Per-lane query state shared by every kernel.
struct Query {
  uint32_t lo, hi;
  uint32_t s0, nseg;  // segments [s0, s0 + nseg) of the query's chromosome
  uint32_t aux;       // per-query word of the fused post-filter (0 without a filter)
};

// What a lane remembers from a counting pass so that the fill needs no second look at the intervals:
// valid when the query touched one segment and its window fitted the lane budget.
struct Replay {
  uint32_t al;    // aligned first slot of the (first recorded) window
  uint64_t mask;  // bit j set: slot al + j is a hit
  bool ok;
  bool kept;      // ids of the first min(hits, kKeep) hits were written to the caller's `keep` slots
  bool packed;    // the window was read from packed records: ids sit in rec[].y
  uint32_t nrec;  // windows recorded (those with hits): the first one above, later ones in the lane's LDS slots
};
// A query over several segments (several length classes on its chromosome) records up to kMaxRec windows: the
// second goes to the lane's `keep` slot (free: ids are only kept for one-segment queries), the third to its
// `xrec` slot, each as three words (al | packed, mask low, mask high); al is even, so bit 0 is free.
constexpr uint32_t kMaxRec = 3;

// The whole hit enumeration of one query per lane, wavefront-converged (all 64 lanes must call it).
//   Count: returns the number of hits (and fills *rp).   Any: returns the smallest hit id (BIVX_NO_HIT if none).
//   Fill:  writes hit ids to hits_base[dst_pos ..), in index order, only positions below `cap`;
//          returns the number of hits.
template <Mode M, bool F, bool MS = false, uint32_t kHeavyRows = kRowsWide>
__device__ __forceinline__ uint32_t enumerate_hits(const IndexView &v, const SegDesc *segs, const Query &qy,
                                                   uint32_t *hits_base, uint64_t dst_pos, uint64_t cap,
                                                   Replay *rp, uint32_t *keep = nullptr, uint32_t *xrec = nullptr) {
  const int lane = threadIdx.x & (kWave - 1);
  uint32_t acc = (M == Mode::Any) ? BIVX_NO_HIT : 0u;
  const uint32_t lo = qy.lo, hi = qy.hi;
  if (M == Mode::Count && rp) {
    rp->al = 0;
    rp->mask = 0;
    rp->ok = MS || qy.nseg <= 1;
    rp->kept = false;
    rp->packed = false;
    rp->nrec = 0;
  }
  // the segment loop is wavefront-uniform so the cooperative part may use __ballot / __shfl
  for (uint32_t k = 0; __any(k < qy.nseg); ++k) {
    Window w{0u, 0u, 0u, 0u, false};
    uint32_t shf = 0, xlow = 0;
    if (k < qy.nseg) {
      const SegDesc d = load_seg(segs + qy.s0 + k);
      w = seg_window(v, d, lo, hi);
      shf = d.shift;
      xlow = lo > d.maxlen ? lo - d.maxlen : 0u;
    }
    const bool nonempty = w.span != 0 && w.b > w.a;
    const bool packed = (shf & kSegPacked) != 0 && w.narrow;
    const bool heavy = nonempty && (w.b - (w.a & ~1u)) > kLight;
    if (nonempty && !heavy) {
      uint32_t al;
      uint64_t mask;
      const bool want = M == Mode::Count && packed && keep != nullptr && qy.nseg == 1;
      if (packed) {
        mask = light_mask_packed<F>(v, w, lo, hi, qy.aux, al, want ? keep : nullptr);
      } else {
        mask = light_mask_pairs<F>(v, w.a, w.b, lo, hi, qy.aux, al);
      }
      if (M == Mode::Count) {
        acc += (uint32_t)__popcll(mask);
        if (rp && (!MS || mask)) {  // several segments: only windows with hits are worth a record
          if (!MS || rp->nrec == 0) {
            rp->al = al;
            rp->mask = mask;
            rp->kept = want;
            rp->packed = packed;
          } else if (rp->nrec < kMaxRec) {
            uint32_t *slot = rp->nrec == 1 ? keep : xrec;
            slot[0] = al | (packed ? 1u : 0u);
            slot[1] = (uint32_t)mask;
            slot[2] = (uint32_t)(mask >> 32);
          } else {
            rp->ok = false;
          }
          ++rp->nrec;
        }
      } else {
        while (mask) {
          const uint32_t j = (uint32_t)__ffsll((long long)mask) - 1u;
          mask &= mask - 1;
          const uint32_t hid = packed ? v.rec[al + j].y : v.id[al + j];  // packed: the line is already here
          if (M == Mode::Any) acc = min(acc, hid);
          if (M == Mode::Fill) {
            if (dst_pos + acc < cap) hits_base[dst_pos + acc] = hid;
            ++acc;
          }
        }
      }
    }
    if (M == Mode::Count && rp && heavy) rp->ok = false;
    uint64_t hm = __ballot(heavy);
    while (hm) {
      const int src = __ffsll((long long)hm) - 1;
      hm &= hm - 1;
      uint32_t ca = __shfl(w.a, src, kWave), cb = __shfl(w.b, src, kWave);
      const uint32_t cl = __shfl(lo, src, kWave), ch = __shfl(hi, src, kWave);
      const uint32_t cx = F ? __shfl(qy.aux, src, kWave) : 0u;
      if (cb - ca > kTrim) {  // long window: trim it to the slots with low in [q.low - maxlen, q.high]
        ca = wave_lower_bound_low(v.se, ca, cb, __shfl(xlow, src, kWave), lane);
        if (ch != 0xFFFFFFFFu) cb = wave_lower_bound_low(v.se, ca, cb, ch + 1u, lane);
      }
      auto is_hit = [&](uint32_t j) {
        const uint2 e = v.se[j];
        return e.x <= ch && e.y >= cl &&
               (!F || filter_accept(v, cl, ch, cx, e.x, e.y, v.id[j]));
      };
      if (M == Mode::Count) {
        uint32_t c = 0;
        for (uint32_t j0 = ca + lane; j0 < cb; j0 += kHeavyRows * kWave) {  // kHeavyRows rows of 64 slots in flight
          uint2 e[kHeavyRows];
#pragma unroll
          for (uint32_t r = 0; r < kHeavyRows; ++r)
            if (j0 + r * kWave < cb) e[r] = v.se[j0 + r * kWave];
#pragma unroll
          for (uint32_t r = 0; r < kHeavyRows; ++r) {
            const uint32_t j = j0 + r * kWave;
            if (j < cb && e[r].x <= ch && e[r].y >= cl && (!F || filter_accept(v, cl, ch, cx, e[r].x, e[r].y, v.id[j]))) ++c;
          }
        }
        c = wave_sum(c);
        if (lane == src) acc += c;
      } else if (M == Mode::Any) {
        uint32_t m = BIVX_NO_HIT;
        for (uint32_t j = ca + lane; j < cb; j += kWave)
          if (is_hit(j)) m = min(m, v.id[j]);
        m = wave_min(m);
        if (lane == src) acc = min(acc, m);
      } else {
        // ballot compaction keeps ascending slot order: the output does not depend on which path ran
        // kHeavyRows rows of 64 slots are loaded together, ids included (a row's ids are one coalesced load; fetching
        // them only for hits would put a dependent load between the ballot and the store of every row).
        const uint64_t pos0 = __shfl((unsigned long long)(dst_pos + acc), src, kWave);
        uint32_t written = 0;
        for (uint32_t j0 = ca + lane; j0 < cb + lane; j0 += kHeavyRows * kWave) {  // wavefront-uniform trip count
          uint2 e[kHeavyRows];
          uint32_t idv[kHeavyRows];
#pragma unroll
          for (uint32_t r = 0; r < kHeavyRows; ++r) {
            const uint32_t j = j0 + r * kWave;
            if (j < cb) {
              e[r] = v.se[j];
              idv[r] = v.id[j];
            }
          }
#pragma unroll
          for (uint32_t r = 0; r < kHeavyRows; ++r) {
            const uint32_t j = j0 + r * kWave;
            const bool hit = j < cb && e[r].x <= ch && e[r].y >= cl &&
                             (!F || filter_accept(v, cl, ch, cx, e[r].x, e[r].y, idv[r]));
            const uint64_t m = __ballot(hit);
            if (hit) {
              const uint64_t p = pos0 + written + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
              if (p < cap) hits_base[p] = idv[r];
            }
            written += (uint32_t)__popcll(m);
          }
        }
        if (lane == src) acc += written;
      }
    }
  }
  return acc;
}

// Stages chrom_seg and the descriptors through LDS (block-cooperative); returns the pointers to use.
template <bool LDS_DESC>
__device__ __forceinline__ void stage_descriptors(const IndexView &v, SegDesc *s_seg, uint32_t *s_cs,
                                                  const SegDesc *&segs, const uint32_t *&cs) {
  if (LDS_DESC) {
    const uint4 *src = reinterpret_cast<const uint4 *>(v.seg);
    uint4 *dst = reinterpret_cast<uint4 *>(s_seg);
    for (uint32_t t = threadIdx.x; t < v.nseg * 2; t += blockDim.x) dst[t] = src[t];
    for (uint32_t t = threadIdx.x; t <= v.nchrom; t += blockDim.x) s_cs[t] = v.chrom_seg[t];
    segs = s_seg;
    cs = s_cs;
  } else {
    segs = v.seg;
    cs = v.chrom_seg;
  }
}

template <bool F>
__device__ __forceinline__ Query load_query(const IndexView &v, const uint32_t *cs, const uint32_t *qchrom,
                                            const uint32_t *qlow, const uint32_t *qhigh, size_t q, bool valid) {
  Query qy{0u, 0u, 0u, 0u, 0u};
  if (valid) {
    qy.lo = qlow[q];
    qy.hi = qhigh[q];
    if (F && v.flt_qaux) qy.aux = v.flt_qaux[q];
    const uint32_t c = qchrom ? qchrom[q] : 0u;
    if (c < v.nchrom) {
      qy.s0 = cs[c];
      qy.nseg = cs[c + 1] - qy.s0;
    }
  }
  return qy;
}

// ---- two-pass kernels ----------------------------------------------------------------------------------
template <Mode M, bool LDS_DESC, bool F>
__global__ __launch_bounds__(kQThreads) void k_query(IndexView v, const uint32_t *__restrict__ qchrom,
                                                     const uint32_t *__restrict__ qlow,
                                                     const uint32_t *__restrict__ qhigh, size_t nq,
                                                     const uint64_t *__restrict__ offsets,
                                                     uint32_t *__restrict__ out) {
  __shared__ SegDesc s_seg[LDS_DESC ? kLdsSegs : 1];
  __shared__ uint32_t s_cs[LDS_DESC ? kLdsChroms + 1 : 1];
  const SegDesc *segs;
  const uint32_t *cs;
  stage_descriptors<LDS_DESC>(v, s_seg, s_cs, segs, cs);
  if (LDS_DESC) __syncthreads();
  const size_t q = (size_t)blockIdx.x * kQThreads + threadIdx.x;
  const bool valid = q < nq;
  const Query qy = load_query<F>(v, cs, qchrom, qlow, qhigh, q, valid);
  uint64_t pos = 0;
  if (M == Mode::Fill && valid) pos = offsets[q];
  const uint32_t r = enumerate_hits<M, F>(v, segs, qy, out, pos, ~0ull, nullptr);
  if (valid && M != Mode::Fill) out[q] = r;
}

// ---- per-query ascending-id ordering of a CSR hit list ------------------------------------------------

constexpr uint32_t kRankMax = 48;     // lists up to this long are rank-sorted by their lane (fast path)
constexpr uint32_t kRankBlock = 8;   // elements ranked per sweep of a list (held in registers)
#ifndef BIVX_FUSED_RANK_BLOCK
#define BIVX_FUSED_RANK_BLOCK 8
#endif
constexpr uint32_t kFusedSortMaxAvg = 6;   // ids per query (by buffer capacity) up to which k_query_fused orders ids itself
constexpr uint32_t kFusedRankBlock = BIVX_FUSED_RANK_BLOCK;  // the same inside k_query_fused, which lives in 64 VGPRs
constexpr uint32_t kSortLane = 24;    // <= this many hits: the owning lane insertion-sorts in place
constexpr uint32_t kSortLds = 4096;   // <= this many: the wavefront bitonic-sorts through LDS (16 KiB per wavefront)

// Ascending sort of s[0..n) by one wavefront: the normalised bitonic network (every comparator puts the
// minimum at the lower index), so virtual +inf pads at indices >= n never move and comparators that
// touch them are simply skipped — any n sorts in place, in LDS or in global memory.
template <typename IdxT>
__device__ __forceinline__ void wave_bitonic_sort(uint32_t *s, IdxT n, int lane) {
  IdxT np2 = 1;
  while (np2 < n) np2 <<= 1;
  for (IdxT k = 2; k <= np2; k <<= 1) {
    for (IdxT t = lane; t < n; t += kWave) {
      const IdxT p = t ^ (k - 1);
      if (p > t && p < n) {
        const uint32_t x = s[t], y = s[p];
        if (x > y) {
          s[t] = y;
          s[p] = x;
        }
      }
    }
    wave_sync_mem();
    for (IdxT j = k >> 2; j > 0; j >>= 1) {
      for (IdxT t = lane; t < n; t += kWave) {
        const IdxT p = t ^ j;
        if (p > t && p < n) {
          const uint32_t x = s[t], y = s[p];
          if (x > y) {
            s[t] = y;
            s[p] = x;
          }
        }
      }
      wave_sync_mem();
    }
  }
}

// One lane's list in[off .. off + n) rank-sorted into outb[off ..): rank = how many elements are smaller (ids
// are distinct inside a query). kRankBlock elements are ranked per sweep of the list, so a list costs
// n * ceil(n / kRankBlock) independent LDS reads — no dependent chain, unlike an insertion sort, and an eighth
// of the n^2 reads of the plain form (LDS bandwidth is what bounds this when lists are ~16 long).
template <uint32_t kRankBlock>
__device__ __forceinline__ void rank_sort_list(const uint32_t *in, uint32_t *outb, uint32_t off, uint32_t n) {
  for (uint32_t i0 = 0; i0 < n; i0 += kRankBlock) {
    uint32_t x[kRankBlock], rank[kRankBlock];
#pragma unroll
    for (uint32_t k = 0; k < kRankBlock; ++k) {
      x[k] = i0 + k < n ? in[off + i0 + k] : 0u;
      rank[k] = 0;
    }
    for (uint32_t j = 0; j < n; ++j) {
      const uint32_t y = in[off + j];
#pragma unroll
      for (uint32_t k = 0; k < kRankBlock; ++k) rank[k] += y < x[k] ? 1u : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < kRankBlock; ++k)
      if (i0 + k < n) outb[off + rank[k]] = x[k];
  }
}

// Sorts the 64 hit lists of one wavefront, hits[o0 .. o1) per lane (adjacent in memory, lane order), ascending,
// in place. `lds` is the wavefront's own stage of LDSN words. All 64 lanes must call it.
template <uint32_t LDSN, uint32_t RB>
__device__ __forceinline__ void wave_sort_lists(uint32_t *lds, uint64_t o0, uint64_t o1, uint32_t *hits, int lane) {
  const uint64_t cnt = o1 - o0;
  // Fast path, the usual case: every list of the wavefront is short and the 64 lists fit half the stage. The
  // region is loaded with coalesced reads, every lane rank-sorts its own list out of LDS into the other half,
  // which is streamed back coalesced.
  const uint64_t wb = __shfl((unsigned long long)o0, 0, kWave);
  const uint64_t we = __shfl((unsigned long long)o1, kWave - 1, kWave);
  if (__all(cnt <= kRankMax) && we - wb <= LDSN / 2) {
    const uint32_t wtotal = (uint32_t)(we - wb);
    uint32_t *in = lds, *outb = lds + LDSN / 2;
    if (__any(cnt > 1)) {
      for (uint32_t i = lane; i < wtotal; i += kWave) in[i] = hits[wb + i];
      wave_sync_mem();
      rank_sort_list<RB>(in, outb, (uint32_t)(o0 - wb), (uint32_t)cnt);
      wave_sync_mem();
      for (uint32_t i = lane; i < wtotal; i += kWave) hits[wb + i] = outb[i];
      wave_sync_mem();
    }
    return;
  }
  if (cnt > 1 && cnt <= kSortLane) {
    uint32_t *h = hits + o0;
    for (uint32_t j = 1; j < (uint32_t)cnt; ++j) {
      const uint32_t x = h[j];
      uint32_t i = j;
      while (i > 0 && h[i - 1] > x) {
        h[i] = h[i - 1];
        --i;
      }
      h[i] = x;
    }
  }
  uint64_t hm = __ballot(cnt > kSortLane);
  while (hm) {
    const int src = __ffsll((long long)hm) - 1;
    hm &= hm - 1;
    const uint64_t b0 = __shfl((unsigned long long)o0, src, kWave);
    const uint64_t n = __shfl((unsigned long long)cnt, src, kWave);
    uint32_t *h = hits + b0;
    if (n <= LDSN) {
      for (uint32_t i = lane; i < (uint32_t)n; i += kWave) lds[i] = h[i];
      wave_sync_mem();
      wave_bitonic_sort<uint32_t>(lds, (uint32_t)n, lane);
      for (uint32_t i = lane; i < (uint32_t)n; i += kWave) h[i] = lds[i];
    } else {
      wave_bitonic_sort<uint64_t>(h, n, lane);  // very long hit lists: same network in global memory
    }
    wave_sync_mem();
  }
}

__global__ __launch_bounds__(kQThreads) void k_sort_hits(const uint64_t *__restrict__ offsets,
                                                         uint32_t *__restrict__ hits, size_t nq, uint64_t cap) {
  __shared__ uint32_t lds[kQWaves][kSortLds];
  const size_t q = (size_t)blockIdx.x * kQThreads + threadIdx.x;
  uint64_t o0, o1;
  if (q < nq) {
    o0 = offsets[q];
    o1 = offsets[q + 1];
  } else {
    o0 = o1 = offsets[nq];  // lanes past the batch own an empty list at the very end: regions stay monotone
  }
  o0 = o0 < cap ? o0 : cap;  // a CSR that did not fit its buffer: nothing beyond the buffer is touched
  o1 = o1 < cap ? o1 : cap;
  wave_sort_lists<kSortLds, kRankBlock>(lds[threadIdx.x >> 6], o0, o1, hits, threadIdx.x & (kWave - 1));
}

// ---- single-pass kernel ----------------------------------------------------------------------------------
// A workgroup owns kFTile = 1024 consecutive queries. It counts them (remembering each short window's hit
// mask and the ids of its first hits), publishes its hit total, sums the totals of ALL earlier tiles, then
// writes offsets and hit ids. The prefix is a two-level sweep, not a serial look-back chain: on MI355X every
// poll of another XCD's status word goes to memory (per-XCD L2s are not coherent), so the number of dependent
// polls, not their width, is what costs. Tiles form groups of 64. A tile reads the words of the earlier tiles of
// its group (one load per lane) and the words of all earlier GROUPS (one load per lane up to 64 groups, i.e.
// 4 M queries; four in flight beyond); the 64th tile of a group publishes the group's total as soon as it has
// its in-group sum. So a tile waits for at most two levels, and one launch covers up to kFMaxTiles tiles (64 M
// queries). Larger batches run as consecutive launches; each starts from the running total its predecessor left
// in offsets[q_begin].
//   ws[kWsTicket] (low 32 bits): tile ticket. ws[kWsDone]: tiles that have left. ws[kWsStatus + g]: kStValid |
//   hits of group g; ws[kWsStatus + kFMaxGroups + t]: kStValid | hits of tile t. Each is written and polled as ONE
//   8-byte agent-scope atomic, so the value needs no separate fence. Tiles take tickets in launch order: every
//   predecessor of a polling tile is already resident, the wait cannot deadlock; spins are bounded anyway.
#ifndef BIVX_FUSED_THREADS
#define BIVX_FUSED_THREADS 1024
#endif
constexpr int kFThreads = BIVX_FUSED_THREADS;
constexpr int kFWaves = kFThreads / kWave;
constexpr int kFR = 1;  // queries per thread. (More per thread was measured and did not pay: a wavefront here is
                        // latency-bound, and the output staging below assumes the 64 lists of a wavefront are adjacent.)
constexpr int kFTile = kFThreads * kFR;
constexpr unsigned kFMaxTiles = 65536;                 // tiles per launch (ordered output)
constexpr unsigned kFMaxGroups = kFMaxTiles / kWave;   // groups of 64 tiles
constexpr unsigned kFlatTiles = 1024;                  // launches up to this many tiles sweep the tile words directly
constexpr uint32_t kStage = 512;     // ids a wavefront lays out in LDS per round before streaming them out
#ifndef BIVX_GATHER
#define BIVX_GATHER 8
#endif
constexpr uint32_t kGather = BIVX_GATHER;  // ids a lane fetches per step when it replays a window in phase 2
constexpr uint32_t kStageMin = 320;  // ... when it has at least this many (5 per lane); below that lanes store directly
constexpr uint64_t kStValid = 1ull << 63;
// workspace words: the two counters and the status array sit on cache lines of their own, so that the atomics
// on the counters do not queue behind (or in front of) the sweeps' polls of the first status words
constexpr uint32_t kWsTicket = 0, kWsCarry = 8, kWsDone = 16, kWsTimeouts = 24, kWsStatus = 32;
// (kWsTimeouts is never cleared by the kernel)
constexpr uint32_t kDoneShift = 44;  // unordered output: ws[kWsDone] = departures << 44 | ids reserved by this launch
constexpr uint32_t kSpinCap = 1u << 20;
constexpr int kFlagSelfClean = 1, kFlagFinal = 2;  // k_query_fused flags: index-owned workspace; last launch of the call

__device__ __forceinline__ uint64_t ld_status(const uint64_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_status(uint64_t *p, uint64_t w) {
  __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Diagnostic build only (-DBIVX_STAMPS): per-tile wall-clock stamps (100 MHz constant counter) written to a
// buffer no other code reads; the product build has no stamp.
#ifdef BIVX_STAMPS
constexpr unsigned kStampTiles = 1024;
__device__ unsigned long long g_stamps[kStampTiles * 8];
#define BIVX_STAMP(k) \
  if (threadIdx.x == 0) g_stamps[(blockIdx.x % kStampTiles) * 8 + (k)] = __builtin_amdgcn_s_memrealtime()
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
__global__ __launch_bounds__(kFThreads, 8) void k_query_fused(IndexView v, const uint32_t *__restrict__ qchrom,
                                                           const uint32_t *__restrict__ qlow,
                                                           const uint32_t *__restrict__ qhigh, size_t q_begin,
                                                           size_t q_end, uint64_t *__restrict__ offsets,
                                                           uint32_t *__restrict__ hits, uint64_t cap,
                                                           uint64_t *__restrict__ ws, int flags,
                                                           uint32_t *__restrict__ counts,
                                                           uint64_t *__restrict__ total_out) {
  const bool self_clean = (flags & kFlagSelfClean) != 0;
  __shared__ SegDesc s_seg[LDS_DESC ? kLdsSegs : 1];
  __shared__ uint32_t s_cs[LDS_DESC ? kLdsChroms + 1 : 1];
  __shared__ uint32_t s_tile;
  __shared__ uint32_t s_last;  // this tile finished its prefix sweep last: it zeroes the workspace for the next call
  __shared__ uint32_t s_wsum[kFWaves];
  __shared__ uint64_t s_base;
  __shared__ uint64_t s_launch_total;  // unordered output, last tile only
  __shared__ uint4 s_keep[kFR][kFThreads];  // ids of each query's first kKeep hits (thread-private slots)
  __shared__ uint32_t s_out[kFWaves][kStage];  // per-wavefront staging of the output ids
  __shared__ uint4 s_xrec[MS ? kFThreads : 1];  // a query's third recorded window (thread-private slots)
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;

  BIVX_STAMP(0);
  if (threadIdx.x == 0) s_last = 0;
  // ordered output: tiles take tickets, so that every predecessor of a waiting tile is resident.
  // unordered output: tiles never wait for each other, any tile may be any block.
  if (threadIdx.x == 0)
    s_tile = U ? blockIdx.x : atomicAdd(reinterpret_cast<unsigned int *>(ws + kWsTicket), 1u);
  const SegDesc *segs;
  const uint32_t *cs;
  stage_descriptors<LDS_DESC>(v, s_seg, s_cs, segs, cs);
  __syncthreads();
  const uint32_t tile = s_tile;
  uint64_t *group = ws + kWsStatus, *status = group + kFMaxGroups;
  BIVX_STAMP(1);
  // A ticket beyond the grid means the workspace was not zeroed (a caller bug): leave without touching memory
  // rather than index the status array and the queries with it.
  if (tile >= gridDim.x) return;

  // phase 1: count the thread's kFR consecutive queries
  const size_t q0 = q_begin + ((size_t)tile * kFThreads + threadIdx.x) * kFR;
  Query qy[kFR];
  Replay rp[kFR];
  uint32_t cnt[kFR];
  uint32_t tsum = 0;
#pragma unroll
  for (int r = 0; r < kFR; ++r) qy[r] = load_query<F>(v, cs, qchrom, qlow, qhigh, q0 + r, q0 + r < q_end);
#pragma unroll
  for (int r = 0; r < kFR; ++r) {
    cnt[r] = enumerate_hits<Mode::Count, F, MS, MS ? kRowsWide : kRowsLean>(v, segs, qy[r], nullptr, 0, 0, &rp[r],
                                             reinterpret_cast<uint32_t *>(&s_keep[r][threadIdx.x]),
                                             reinterpret_cast<uint32_t *>(&s_xrec[MS ? threadIdx.x : 0]));
    tsum += cnt[r];
  }

  BIVX_STAMP(2);
  // workgroup exclusive scan of the per-thread sums
  uint32_t incl = tsum;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d, kWave);
    if (lane >= d) incl += o;
  }
  if (lane == kWave - 1) s_wsum[wave] = incl;
  __syncthreads();
  uint32_t wbase = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kFWaves; ++w) {
    const uint32_t s = s_wsum[w];
    if (w < wave) wbase += s;
    total += s;
  }
  const uint32_t local = wbase + incl - tsum;

  // prefix across tiles: wave 0 publishes this tile's total and sums every earlier tile's
  if (U) {
    if (threadIdx.x == 0)
      s_base = atomicAdd(reinterpret_cast<unsigned long long *>(ws + kWsTicket), (unsigned long long)total);
  } else if (wave == 0) {
    BIVX_STAMP(3);
    if (lane == 0) st_status(&status[tile], kStValid | (uint64_t)total);
    auto wait_word = [&](const uint64_t *p, uint64_t w) -> uint64_t {
      uint32_t spins = 0;
      while (!(w & kStValid) && spins < kSpinCap) {  // bounded: a wrong prefix beats a hung GPU
        __builtin_amdgcn_s_sleep(1);
        w = ld_status(p);
        ++spins;
      }
      if (!(w & kStValid)) atomicAdd(reinterpret_cast<unsigned long long *>(ws + kWsTimeouts), 1ull);
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
    uint64_t in_group = 0;
    if (!flat) {
      const uint64_t *mine = &status[(g << 6) + (uint32_t)lane];
      in_group = wave_total((uint32_t)lane < r ? wait_word(mine, ld_status(mine)) : 0ull);
      if (r == 63u && lane == 0) st_status(&group[g], kStValid | (in_group + total));
    }
    const uint64_t *words = flat ? status : group;
    const uint32_t nwords = flat ? tile : g;
    uint64_t sum = 0;
    for (uint32_t t0 = 0; t0 < nwords; t0 += 4 * kWave) {
      uint64_t w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t t = t0 + j * kWave + lane;
        w[j] = t < nwords ? ld_status(&words[t]) : kStValid;
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
  uint64_t pos = s_base + local;
#pragma unroll
  for (int r = 0; r < kFR; ++r) {
    const size_t q = q0 + r;
    if (q < q_end) {
      offsets[q] = pos;
      if (U) counts[q] = cnt[r];
      else if (q == q_end - 1) offsets[q_end] = pos + cnt[r];
    }
    const uint32_t *kept = reinterpret_cast<const uint32_t *>(&s_keep[r][threadIdx.x]);
    // The replay cursor walks the lane's recorded windows in segment order: `mrem` holds the bits of the current
    // window that are not consumed yet. replay(k0, k1, put) hands the ids of hits k0 .. k1-1 (consecutive calls
    // continue where the last one stopped) to put(k, id). kGather ids are fetched per step with all their loads in
    // flight together: one load per hit in a while-loop made every lane wait a full memory latency per id, which
    // was most of phase 2 when queries have ~16 hits.
    uint64_t mrem = rp[r].mask;
    uint32_t cur_al = rp[r].al, cur_rec = 1;
    bool cur_packed = rp[r].packed;
    const uint32_t *xrec = reinterpret_cast<const uint32_t *>(&s_xrec[MS ? threadIdx.x : 0]);
    auto replay = [&](uint32_t k0, uint32_t k1, auto put) {
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
            if (rp[r].kept && k + i < kKeep) ids[i] = kept[k + i];
            else if (MS) ids[i] = (pk >> i & 1u) ? v.rec[slot[i]].y : v.id[slot[i]];
            else ids[i] = rp[r].packed ? v.rec[rp[r].al + slot[i]].y : v.id[rp[r].al + slot[i]];
          }
        }
#pragma unroll
        for (uint32_t i = 0; i < kGather; ++i)
          if (k + i < k1) put(k + i, ids[i]);
      }
    };
    const bool all_replay = __all(rp[r].ok);
    const uint64_t wpos0 = __shfl((unsigned long long)pos, 0, kWave);
    const uint32_t loff = (uint32_t)(pos - wpos0);
    const uint32_t wtotal = __shfl(loff + cnt[r], kWave - 1, kWave);
    if (cap == 0) {
      // a pure count (bivx_count_dev): the offsets are all that is asked for
    } else if (S && all_replay) {
      // Rounds of consecutive lanes whose lists fit half the stage together (a list has at most kLight ids): ids
      // go to one half in slot order, every lane rank-sorts its own list into the other half, and that half is
      // streamed out coalesced.
      uint32_t *in = s_out[wave], *outb = s_out[wave] + kStage / 2;
      uint32_t first = 0;
      while (first < (uint32_t)kWave) {
        const uint32_t base = __shfl(loff, (int)first, kWave);
        const uint64_t fit = __ballot((uint32_t)lane >= first && loff + cnt[r] - base <= kStage / 2);
        const uint64_t nofit = ~fit & (~0ull << first);
        const uint32_t next = nofit ? (uint32_t)__ffsll((long long)nofit) - 1u : (uint32_t)kWave;
        const bool mine = (uint32_t)lane >= first && (uint32_t)lane < next && cnt[r] != 0;
        const uint32_t rel = loff - base;
        if (mine) replay(0u, cnt[r], [&](uint32_t k, uint32_t id) { in[rel + k] = id; });
        wave_sync_mem();
        if (mine) rank_sort_list<kFusedRankBlock>(in, outb, rel, cnt[r]);
        wave_sync_mem();
        const uint32_t nthis = __shfl(loff + cnt[r], (int)next - 1, kWave) - base;
        for (uint32_t i = lane; i < nthis; i += kWave) {
          const uint64_t p = wpos0 + base + i;
          if (p < cap) hits[p] = outb[i];
        }
        wave_sync_mem();
        first = next;
      }
    } else if (all_replay && wtotal >= kStageMin) {
      uint32_t *buf = s_out[wave];
      uint32_t kdone = 0;  // a lane's hits enter the stage in order, over one or more consecutive rounds
      for (uint32_t base = 0; base < wtotal; base += kStage) {
        if (kdone < cnt[r] && loff < base + kStage) {
          const uint32_t room = base + kStage - loff;
          const uint32_t kend = cnt[r] < room ? cnt[r] : room;
          replay(kdone, kend, [&](uint32_t k, uint32_t id) { buf[loff + k - base] = id; });
          kdone = kend;
        }
        wave_sync_mem();
        const uint32_t nthis = wtotal - base < kStage ? wtotal - base : kStage;
        for (uint32_t i = lane; i < nthis; i += kWave) {
          const uint64_t p = wpos0 + base + i;
          if (p < cap) hits[p] = buf[i];
        }
        wave_sync_mem();
      }
    } else {
      // few ids per lane (or a wavefront that holds general-path queries): every lane stores its own list
      if (rp[r].ok) {
        replay(0u, cnt[r], [&](uint32_t k, uint32_t id) {
          if (pos + k < cap) hits[pos + k] = id;
        });
        qy[r].nseg = 0;
      }
      if (!all_replay)
        (void)enumerate_hits<Mode::Fill, F, false, MS ? kRowsWide : kRowsLean>(v, segs, qy[r], hits, pos, cap, nullptr);
      if (S) {  // a wavefront with general-path queries: sort what it has just written (lists cut by `cap` stay cut)
        wave_sync_mem();
        const uint64_t e = pos + cnt[r];
        wave_sort_lists<kStage, kFusedRankBlock>(s_out[wave], pos < cap ? pos : cap, e < cap ? e : cap, hits, lane);
      }
    }
    pos += cnt[r];
  }
  // self-cleaning workspace: every tile bumps `done` when it leaves (its sweep is long over); the tile that
  // sees gridDim.x - 1 knows nobody reads the words any more and zeroes them for the next launch. Off the
  // critical path: nothing waits for this but the end of the kernel.
  // Unordered output finds the last tile with the same word, and sums the tiles' totals in it on the way (one
  // atomic carries both: departures in the high bits, ids in the low kDoneShift bits), so the last tile knows the
  // launch total without reading a word other tiles are still adding to — no fence anywhere: an agent-scope fence
  // writes back and invalidates the XCD's whole L2 on this chip, which cost more than the prefix it replaced.
  if (self_clean || U) {
    if (threadIdx.x == 0) {
      if (U) {
        const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long *>(ws + kWsDone),
                                                 (1ull << kDoneShift) | (unsigned long long)total);
        if ((old >> kDoneShift) == gridDim.x - 1) {
          s_last = 1;
          s_launch_total = (old & ((1ull << kDoneShift) - 1)) + total;  // ids reserved by this launch
        }
      } else if (atomicAdd(reinterpret_cast<unsigned int *>(ws + kWsDone), 1u) == gridDim.x - 1) {
        s_last = 1;
      }
    }
    __syncthreads();
    if (s_last) {
      if (!U)
        for (uint32_t t = threadIdx.x; t < gridDim.x; t += kFThreads) {
          status[t] = 0;
          if (t < (gridDim.x + kWave - 1) / kWave) group[t] = 0;
        }
      if (threadIdx.x == 0) {
        if (U) {  // running total over the call's launches; the reservation counter restarts after the last one
          const uint64_t sum = ws[kWsCarry] + s_launch_total;
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
  if (threadIdx.x == 0) g_stamps[(blockIdx.x % kStampTiles) * 8 + 7] = tile;
#endif
}

inline bool fits_lds(const IndexView &v) { return v.nseg <= kLdsSegs && v.nchrom <= kLdsChroms; }
inline unsigned tiles_for(size_t q) { return (unsigned)((q + kQThreads - 1) / kQThreads); }

template <Mode M>
int launch_query(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                 size_t q, const uint64_t *d_offsets, uint32_t *d_out, hipStream_t s) {
  if (q == 0) return 0;
  const dim3 grid(tiles_for(q)), block(kQThreads);
  const bool lds = fits_lds(v), flt = v.flt_kind != BIVX_FILTER_NONE;  // filter code is compiled out when unused
  if (lds && !flt)
    hipLaunchKernelGGL((k_query<M, true, false>), grid, block, 0, s, v, d_qchrom, d_qlow, d_qhigh, q, d_offsets, d_out);
  else if (lds)
    hipLaunchKernelGGL((k_query<M, true, true>), grid, block, 0, s, v, d_qchrom, d_qlow, d_qhigh, q, d_offsets, d_out);
  else if (!flt)
    hipLaunchKernelGGL((k_query<M, false, false>), grid, block, 0, s, v, d_qchrom, d_qlow, d_qhigh, q, d_offsets, d_out);
  else
    hipLaunchKernelGGL((k_query<M, false, true>), grid, block, 0, s, v, d_qchrom, d_qlow, d_qhigh, q, d_offsets, d_out);
  BIVX_HIP(hipGetLastError());
  return 0;
}

}  // namespace

int launch_fill(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                size_t q, const uint64_t *d_offsets, uint32_t *d_hits, hipStream_t s) {
  return launch_query<Mode::Fill>(v, d_qchrom, d_qlow, d_qhigh, q, d_offsets, d_hits, s);
}

int launch_any(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
               size_t q, uint32_t *d_first, hipStream_t s) {
  return launch_query<Mode::Any>(v, d_qchrom, d_qlow, d_qhigh, q, nullptr, d_first, s);
}

size_t fused_workspace_bytes(size_t q) {
  (void)q;
  return ((size_t)kFMaxTiles + kFMaxGroups + kWsStatus) * sizeof(uint64_t);
}

size_t fused_workspace_timeouts_offset() { return (size_t)kWsTimeouts * sizeof(uint64_t); }

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
  const size_t per_launch = (size_t)(unordered ? (1u << 19) : kFMaxTiles) * kFTile;
  // caller's workspace: zeroed in front of every launch (ordered output), or once per call (unordered output:
  // the running total lives in it across the call's launches)
  if (!self_clean && unordered) BIVX_HIP(hipMemsetAsync(d_ws, 0, (size_t)kWsStatus * sizeof(uint64_t), s));
  for (size_t q0 = 0; q0 < q; q0 += per_launch) {
    const size_t q1 = q0 + per_launch < q ? q0 + per_launch : q;
    const unsigned tiles = (unsigned)((q1 - q0 + kFTile - 1) / kFTile);
    if (!self_clean && !unordered)
      BIVX_HIP(hipMemsetAsync(d_ws, 0, ((size_t)tiles + kFMaxGroups + kWsStatus) * sizeof(uint64_t), s));
    const dim3 grid(tiles), block(kFThreads);
    const bool lds = fits_lds(v), flt = v.flt_kind != BIVX_FILTER_NONE;
    const int flags = (self_clean ? kFlagSelfClean : 0) | (q1 == q ? kFlagFinal : 0);
    // Ordering ids inside the kernel pays while a wavefront's 64 lists fit half its output stage (one round, all
    // lanes busy); the buffer capacity is the only bound on the hit count the host has. Denser results are
    // ordered by k_sort_hits afterwards, whose stage is eight times larger.
    const bool sort_inside = sort_ids && !unordered && cap <= (uint64_t)kFusedSortMaxAvg * q;
#define BIVX_LAUNCH_FUSED_V(L, FL, SO, MSV, UV)                                                               \
  hipLaunchKernelGGL((k_query_fused<L, FL, SO, MSV, UV>), grid, block, 0, s, v, d_qchrom, d_qlow, d_qhigh, q0, \
                     q1, d_offsets, d_hits, cap, ws, flags, d_counts, d_total)
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
#undef BIVX_LAUNCH_FUSED
#undef BIVX_LAUNCH_FUSED_V
    if (sort_ids && !sort_inside && !unordered) {
      BIVX_HIP(hipGetLastError());
      if (int rc = launch_sort_hits(d_offsets + q0, d_hits, q1 - q0, cap, s)) return rc;
    }
  }
  BIVX_HIP(hipGetLastError());
  return 0;
}

#ifdef BIVX_STAMPS
extern "C" int bivx_debug_stamps(unsigned long long *out, size_t n) {
  if (n > (size_t)kStampTiles * 8) n = (size_t)kStampTiles * 8;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

int launch_sort_hits(const uint64_t *d_offsets, uint32_t *d_hits, size_t q, uint64_t cap, hipStream_t s) {
  if (q == 0) return 0;
  hipLaunchKernelGGL(k_sort_hits, dim3(tiles_for(q)), dim3(kQThreads), 0, s, d_offsets, d_hits, q, cap);
  BIVX_HIP(hipGetLastError());
  return 0;
}

}  // namespace bivx
