// query_device.h — device-side building blocks shared by the query kernels (query.hip, query_fused.hip):
// the candidate window of a query in a segment, the lane and the wavefront-cooperative enumeration, the fused
// post-filters, and the per-query id sort. Everything here is inline device code for gfx950 (wave64).
#ifndef BIVX_QUERY_DEVICE_H_
#define BIVX_QUERY_DEVICE_H_

#include "common.h"
#include "wave_device.h"

namespace bivx {
namespace {

constexpr int kQThreads = 256;
constexpr int kQWaves = kQThreads / kWave;
constexpr uint32_t kLight = 64;      // window slots a lane reads by itself; longer windows go to the wavefront
#ifndef BIVX_TRIM
#define BIVX_TRIM 512
#endif
constexpr uint32_t kTrim = BIVX_TRIM;  // wavefront windows longer than this are first trimmed by a 64-ary search
constexpr uint32_t kRows = 4;        // rows of 64 slots (and their ids) the wavefront-cooperative path keeps in flight
constexpr uint32_t kSlabSlots = 256;  // candidate slots a wavefront stages through LDS when its 64 windows are neighbours (8 keep slots per lane; 128 with 4)
constexpr uint32_t kSlabMinLanes = 32;  // ... and at least this many of its lanes' windows fit the slab
constexpr uint32_t kLdsSegs = 128;   // descriptors staged in LDS (4 KiB) ...
constexpr uint32_t kLdsChroms = 512; // ... with the chromosomes' segment ranges (4 KiB); larger indexes read them from global

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
// bit j <-> slot al + j, al = a rounded down to 2. If `keep` is given, the ids of the first KEEP hits are
// written there (ascending slot order) as they are found.
constexpr uint32_t kKeep = 4;  // default number of kept ids (one uint4 slot per lane)

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
template <bool F, uint32_t KEEP>
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
            if (n < KEEP) keep[n] = ii[e];
            ++n;
          }
        }
      }
    }
  }
  return m;
}

template <bool F, uint32_t KEEP = kKeep>
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
      mask |= (uint64_t)packed_eval_chunk<F, KEEP>(v, w, lo, hi, qaux, al + c0, r, keep, n) << c0;
    }
  }
  return mask;
}

// The same out of the wavefront's LDS slab: slab[i] holds the record pair of slots lbase + 2i, lbase + 2i + 1
// (lbase even). Used when the 64 windows of a wavefront are neighbours in the index — a position-sorted batch — so
// that every line of records is fetched from memory once per wavefront, with coalesced 16-byte loads, instead of
// once per query (reference loop replaced: one find_overlaps per record in file order, mapper.hpp:202-236).
template <bool F>
__device__ __forceinline__ uint64_t light_mask_packed_lds(const IndexView &v, const uint4 *slab, uint32_t lbase,
                                                          const Window &w, uint32_t lo, uint32_t hi, uint32_t qaux,
                                                          uint32_t &al) {
  al = w.a & ~1u;
  uint64_t mask = 0;
  uint32_t n = 0;
#pragma unroll 1
  for (uint32_t c0 = 0; c0 < kLight; c0 += 8) {
    if (al + c0 < w.b) {
      uint4 r[4];
      const uint32_t c = al + c0;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c + 2 * j < w.b) r[j] = slab[((c - lbase) >> 1) + j];
      mask |= (uint64_t)packed_eval_chunk<F, kKeep>(v, w, lo, hi, qaux, c, r, nullptr, n) << c0;
    }
  }
  return mask;
}

// The common case of the slab, without a branch: every window of the wavefront is shorter than 32 slots, so the hit mask
// is one word. All lanes walk the same number of 8-slot chunks (the longest window's); a chunk is four 16-byte LDS
// reads whatever the window's end — what lies beyond is evaluated like the rest and masked off at the end with the
// window's own bits. Coordinates are taken relative to the window's first cell, where a record's low is its 16 stored
// bits minus the cell's (mod 65536): hit <=> low <= q.high and low + length >= q.low. The mask is built by doubling
// (m = 2m + hit, one add-with-carry per record) and reversed once.
// m = 2m + (a1 <= b1 && a2 >= b2) in three vector instructions: the two comparisons' lane masks (llvm.amdgcn.icmp,
// predicates 37 = ule, 35 = uge) are and-ed on the scalar unit and become the carry-in of an add-with-carry
__device__ __forceinline__ uint32_t shift_in_le_ge(uint32_t m, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2) {
  uint32_t r;
  const uint64_t cc = __builtin_amdgcn_uicmp(a1, b1, 37) & __builtin_amdgcn_uicmp(a2, b2, 35);
  asm("v_addc_co_u32 %0, vcc, %1, %1, %2" : "=v"(r) : "v"(m), "s"(cc) : "vcc");
  return r;
}

__device__ __forceinline__ uint32_t slab_mask32(const uint4 *slab, uint32_t lbase, const Window &w, uint32_t lo,
                                                uint32_t hi, bool nonempty) {
  const uint32_t al = w.a & ~1u;
  const uint32_t nch = wave_max(nonempty ? (w.b - al + 7u) >> 3 : 0u);
  if (nch == 0) return 0u;
  const uint4 *sp = slab + (nonempty ? (al - lbase) >> 1 : 0u);
  const uint32_t base = w.cell0_low;
  const uint32_t qh = hi - base, ql = lo > base ? lo - base : 0u;
  uint32_t m = 0;
#pragma unroll 1
  for (uint32_t c = 0; c < nch; ++c) {
    uint4 r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = sp[4 * c + j];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t rr[2] = {r[j].x, r[j].z};
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const uint32_t rl = (rr[e] - base) & 0xFFFFu;
        const uint32_t rh = rl + (rr[e] >> 16);
        m = shift_in_le_ge(m, rl, qh, rh, ql);
      }
    }
  }
  m = __brev(m) >> (32u - 8u * nch);
  const uint32_t wm = nonempty ? ((1u << (w.b - w.a)) - 1u) << (w.a - al) : 0u;
  return m & wm;
}

// The same for windows of up to 64 slots (kLight): chunks 0..3 build the low word, chunks 4..7 the high word.
__device__ __forceinline__ uint64_t slab_mask64(const uint4 *slab, uint32_t lbase, const Window &w, uint32_t lo,
                                                uint32_t hi, bool nonempty) {
  const uint32_t al = w.a & ~1u;
  const uint32_t nch = wave_max(nonempty ? (w.b - al + 7u) >> 3 : 0u);
  if (nch == 0) return 0ull;
  const uint4 *sp = slab + (nonempty ? (al - lbase) >> 1 : 0u);
  const uint32_t base = w.cell0_low;
  const uint32_t qh = hi - base, ql = lo > base ? lo - base : 0u;
  uint32_t mw[2] = {0u, 0u};
#pragma unroll
  for (uint32_t h = 0; h < 2; ++h) {
    const uint32_t c0 = 4u * h, c1 = nch < c0 + 4u ? nch : c0 + 4u;
    if (c1 <= c0) break;
    uint32_t m = 0;
#pragma unroll 1
    for (uint32_t c = c0; c < c1; ++c) {
      uint4 r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = sp[4 * c + j];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t rr[2] = {r[j].x, r[j].z};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const uint32_t rl = (rr[e] - base) & 0xFFFFu;
          m = shift_in_le_ge(m, rl, qh, rl + (rr[e] >> 16), ql);
        }
      }
    }
    mw[h] = __brev(m) >> (32u - 8u * (c1 - c0));
  }
  const uint32_t n = w.b - al;  // <= 64
  const uint64_t wm = nonempty ? ((n >= 64u ? ~0ull : (1ull << n) - 1ull) & ~(uint64_t)(w.a - al)) : 0ull;
  return ((uint64_t)mw[0] | (uint64_t)mw[1] << 32) & wm;
}

// First slot in [a, b) whose low is >= x (b if there is none), found by the whole wavefront: a 64-ary search — the
// 64 lanes probe 64 evenly spaced slots, one __ballot tells which gap holds the answer, repeat. Used to trim long
// candidate windows (many intervals starting inside one directory cell) to the slots whose low lies in
// [q.low - maxlen, q.high] before they are scanned. All 64 lanes must call it with the same arguments.
// `gran`: the slots are ordered by (low - base) >> gran only (IndexView::order_shift: by directory cell, append order
// inside a cell; 0: by low itself) — the comparison is made at that granularity, so the answer is the first slot of x's
// cell: a superset of the exact range, which is all a window has to be (every slot in it is evaluated).
__device__ __forceinline__ uint32_t wave_lower_bound_low(const uint2 *se, uint32_t a, uint32_t b, uint32_t x, int lane,
                                                         uint32_t base = 0, uint32_t gran = 0) {
  const uint32_t xk = x > base ? (x - base) >> gran : 0u;
  auto key_ge = [&](uint32_t p) { return ((se[p].x - base) >> gran) >= xk; };
  while (b - a > (uint32_t)kWave) {
    const uint32_t step = (b - a + kWave - 1) / kWave;
    const uint32_t p = a + step * (uint32_t)lane;
    const bool ge = p < b ? key_ge(p) : true;  // keys ascend inside a segment: the ballot is 0..01..1
    const uint64_t m = __ballot(ge);
    const uint32_t first = m ? (uint32_t)__ffsll((long long)m) - 1u : (uint32_t)kWave;
    const uint32_t nb = first < (uint32_t)kWave ? min(a + step * first, b) : b;
    const uint32_t na = first > 0 ? a + step * (first - 1) + 1 : a;
    a = na;
    b = nb;
  }
  const uint32_t p = a + (uint32_t)lane;
  const uint64_t m = __ballot(p < b ? key_ge(p) : true);
  const uint32_t first = m ? (uint32_t)__ffsll((long long)m) - 1u : (uint32_t)kWave;
  return min(a + first, b);
}

// orders one wavefront's LDS / global accesses: what lanes wrote before is visible to all lanes after
__device__ __forceinline__ void wave_sync_mem() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// the same for LDS traffic only: one wavefront's LDS instructions execute in order, so all it takes is that the
// compiler keeps the order too (the full form above also waits for the wavefront's outstanding global stores,
// which an output stage that is refilled in rounds must not do)
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The batch's own streams — queries in, offsets and ids out — are touched once: non-temporal, so that they do not
// push the index's lines (which random queries come back to) out of L2 and the Infinity Cache. Config 3 moves 330 MB of
// them per batch next to a 212 MB index: 0.361 -> 0.333 ms in generation order, 0.182 -> 0.170 position-sorted.
// Only for stores a wavefront makes line by line (offsets, ids out of a stage): a lane storing its own list four bytes
// at a time needs L2 to merge the pieces — non-temporal there doubled the SV-like spectra's times.
template <typename T>
__device__ __forceinline__ T stream_load(const T *p) {
  return __builtin_nontemporal_load(p);
}
template <typename T>
__device__ __forceinline__ void stream_store(T *p, T x) {
  __builtin_nontemporal_store(x, p);
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
  bool kept;      // ids of the first min(hits, KEEP) hits were written to the caller's `keep` slots
  bool packed;    // the window was read from packed records: ids sit in rec[].y
  bool lds;       // ... out of the wavefront's LDS slab, where they still are: slot s is slab2[s - lbase]
  uint32_t lbase; // first slot of the slab (wavefront-uniform)
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
//   SLAB (Count only, one segment per chromosome): `slab` is kSlabSlots * 8 bytes of LDS owned by the calling
//          wavefront; it may alias the 64 lanes' `keep` slots (a wavefront that fills the slab keeps no ids there).
template <Mode M, bool F, bool MS = false, uint32_t KEEP = kKeep, uint32_t kHeavyRows = kRows, bool SLAB = false>
__device__ __forceinline__ uint32_t enumerate_hits(const IndexView &v, const SegDesc *segs, const Query &qy,
                                                   uint32_t *hits_base, uint64_t dst_pos, uint64_t cap,
                                                   Replay *rp, uint32_t *keep = nullptr, uint32_t *xrec = nullptr,
                                                   uint4 *slab = nullptr) {
  const int lane = threadIdx.x & (kWave - 1);
  uint32_t acc = (M == Mode::Any) ? BIVX_NO_HIT : 0u;
  const uint32_t lo = qy.lo, hi = qy.hi;
  if (M == Mode::Count && rp) {
    rp->al = 0;
    rp->mask = 0;
    rp->ok = MS || qy.nseg <= 1;
    rp->kept = false;
    rp->packed = false;
    rp->lds = false;
    rp->lbase = 0;
    rp->nrec = 0;
  }
  // the segment loop is wavefront-uniform so the cooperative part may use __ballot / __shfl
  // (kernels built for indexes with one segment per chromosome make a single trip)
  for (uint32_t k = 0; MS || M != Mode::Count ? __any(k < qy.nseg) : k < 1u; ++k) {
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
    // Neighbouring windows (a position-sorted batch): the wavefront fetches the union of its windows once, with
    // coalesced 16-byte loads, into its LDS slab; the lanes then read their windows — and later the ids of their
    // hits — from there. Decided per wavefront from the windows themselves, so it needs no hint from the caller:
    // a wavefront whose windows are scattered (queries in arbitrary order) keeps the per-lane loads below.
    bool in_slab = false, slab_on = false;
    uint32_t lbase = 0;
    if (SLAB && M == Mode::Count) {
      const bool cand = nonempty && !heavy && packed;
      lbase = __builtin_amdgcn_readfirstlane(wave_min(cand ? (w.a & ~1u) : 0xFFFFFFFFu));
      in_slab = cand && w.b - lbase <= KEEP * (kWave / 2);  // the slab is the 64 lanes' keep slots: KEEP * 32 records
      slab_on = (uint32_t)__popcll(__ballot(in_slab)) >= kSlabMinLanes;
      if (slab_on) {
        const uint32_t bmax = __builtin_amdgcn_readfirstlane(wave_max(in_slab ? w.b : 0u));
        const uint32_t npairs = (bmax - lbase + 1) >> 1;  // fits the slab; rec[] carries two spare slots
        const uint4 *src = reinterpret_cast<const uint4 *>(v.rec) + (lbase >> 1);
        for (uint32_t i = (uint32_t)lane; i < npairs; i += kWave) slab[i] = src[i];
        wave_sync_lds();
      }
      in_slab = in_slab && slab_on;
      if (!MS && !F && slab_on && rp) {
        // every window of the wavefront is in the slab (hence light and packed): the branch-free evaluation
        if (!__any(nonempty && !in_slab)) {
          const bool short32 = !__any(nonempty && w.b - (w.a & ~1u) >= 32u);
          const uint64_t m = short32 ? (uint64_t)slab_mask32(slab, lbase, w, lo, hi, nonempty)
                                     : slab_mask64(slab, lbase, w, lo, hi, nonempty);
          rp->al = w.a & ~1u;
          rp->mask = m;
          rp->packed = true;
          rp->lds = true;
          rp->lbase = lbase;
          rp->nrec = 1;
          return (uint32_t)__popcll(m);
        }
      }
    }
    if (nonempty && !heavy) {
      uint32_t al;
      uint64_t mask;
      const bool want = M == Mode::Count && packed && keep != nullptr && qy.nseg == 1 && !slab_on;
      if (SLAB && in_slab) {
        mask = light_mask_packed_lds<F>(v, slab, lbase, w, lo, hi, qy.aux, al);
      } else if (packed) {
        mask = light_mask_packed<F, KEEP>(v, w, lo, hi, qy.aux, al, want ? keep : nullptr);
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
            rp->lds = SLAB && in_slab;
            rp->lbase = lbase;
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
      // (`src` is the same in every lane: a lane's register is read straight into a scalar — v_readlane — where __shfl
      // goes through the LDS crossbar and is waited for, six times per window)
      auto of_src = [&](uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane((int)x, src); };
      uint32_t ca = of_src(w.a), cb = of_src(w.b);
      const uint32_t cl = of_src(lo), ch = of_src(hi);
      const uint32_t cx = F ? of_src(qy.aux) : 0u;
      if (cb - ca > kTrim) {  // long window: trim it to the slots with low in [q.low - maxlen, q.high]
        // (an index ordered by directory cell — IndexView::order_shift — is trimmed to whole cells: the segment's base and
        // cell shift are read again from its descriptor, this path is rare)
        uint32_t cbase = 0, gran = 0;
        if (v.order_shift) {
          const SegDesc *sd = segs + of_src(qy.s0) + k;
          cbase = sd->base;
          gran = sd->shift & 31u;
        }
        ca = wave_lower_bound_low(v.se, ca, cb, of_src(xlow), lane, cbase, gran);
        // first slot whose key is beyond q.high's: at cell granularity that is the first slot of the NEXT cell
        const uint32_t nx = gran ? (((ch > cbase ? (ch - cbase) >> gran : 0u) + 1u) << gran) + cbase : ch + 1u;
        if (nx > ch) cb = wave_lower_bound_low(v.se, ca, cb, nx, lane, cbase, gran);  // (no wrap past 2^32)
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
        c = wave_sum(c);  // (a row's hits counted on its ballot instead: slower — 0.86 against 0.77 ms on tools/clustered_bench.py)
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
        const uint64_t pos0 = (uint64_t)of_src((uint32_t)(dst_pos + acc)) | (uint64_t)of_src((uint32_t)((dst_pos + acc) >> 32)) << 32;
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

// Stages the chromosomes' segment ranges and the descriptors through LDS (block-cooperative); returns the pointers to use.
template <bool LDS_DESC>
__device__ __forceinline__ void stage_descriptors(const IndexView &v, SegDesc *s_seg, uint2 *s_cs,
                                                  const SegDesc *&segs, const uint2 *&cs) {
  if (LDS_DESC) {
    const uint4 *src = reinterpret_cast<const uint4 *>(v.seg);
    uint4 *dst = reinterpret_cast<uint4 *>(s_seg);
    for (uint32_t t = threadIdx.x; t < v.nseg * 2; t += blockDim.x) dst[t] = src[t];
    for (uint32_t t = threadIdx.x; t < v.nchrom; t += blockDim.x) s_cs[t] = v.chrom_rng[t];
    segs = s_seg;
    cs = s_cs;
  } else {
    segs = v.seg;
    cs = v.chrom_rng;
  }
}

template <bool F>
__device__ __forceinline__ Query load_query(const IndexView &v, const uint2 *cs, const uint32_t *qchrom,
                                            const uint32_t *qlow, const uint32_t *qhigh, size_t q, bool valid) {
  Query qy{0u, 0u, 0u, 0u, 0u};
  if (valid) {
    qy.lo = stream_load(qlow + q);
    qy.hi = stream_load(qhigh + q);
    if (F && v.flt_qaux) qy.aux = v.flt_qaux[q];
    const uint32_t c = qchrom ? stream_load(qchrom + q) : 0u;
    if (c < v.nchrom) {
      const uint2 r = cs[c];
      qy.s0 = r.x;
      qy.nseg = r.y;
    }
  }
  return qy;
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

// One lane's list of up to N ids (N = 8, 16, 32) ordered by a sorting network on registers: in[off .. off + n) ->
// outb[off .. off + n) ascending, `outb` may be `in` (missing ids are +inf and stay behind the list's end; ids are
// distinct). 19 comparators for eight (two instructions each), 70 for sixteen (eight up, eight down, a bitonic merge), 220
// for thirty-two — against
// n * ceil(n / 8) * 16 instructions of the rank sort above, which every lane of a wavefront pays for the LONGEST list among
// the 64: 2 930 vector instructions per wavefront at config 5 (lists of 8-35 ids), 78 % of the ordering pass's time.
#define BIVX_NET_CE(i, j)                  \
  {                                        \
    const uint32_t lo_ = min(x[i], x[j]);  \
    x[j] = max(x[i], x[j]);                \
    x[i] = lo_;                            \
  }
#define BIVX_NET_SORT8(a0, a1, a2, a3, a4, a5, a6, a7) /* ascending in the order the indexes are given */                  \
  BIVX_NET_CE(a0, a1) BIVX_NET_CE(a2, a3) BIVX_NET_CE(a4, a5) BIVX_NET_CE(a6, a7)                                       \
  BIVX_NET_CE(a0, a2) BIVX_NET_CE(a1, a3) BIVX_NET_CE(a4, a6) BIVX_NET_CE(a5, a7)                                       \
  BIVX_NET_CE(a1, a2) BIVX_NET_CE(a5, a6) BIVX_NET_CE(a0, a4) BIVX_NET_CE(a3, a7)                                       \
  BIVX_NET_CE(a1, a5) BIVX_NET_CE(a2, a6) BIVX_NET_CE(a1, a4) BIVX_NET_CE(a3, a6) BIVX_NET_CE(a2, a4) BIVX_NET_CE(a3, a5) \
  BIVX_NET_CE(a3, a4)
template <uint32_t N>
__device__ __forceinline__ void net_sort_list(const uint32_t *in, uint32_t *outb, uint32_t off, uint32_t n) {
  static_assert(N == 8 || N == 16 || N == 32, "network sizes");
  uint32_t x[N];
#pragma unroll
  for (uint32_t k = 0; k < N; ++k) x[k] = k < n ? in[off + k] : 0xFFFFFFFFu;
  BIVX_NET_SORT8(0, 1, 2, 3, 4, 5, 6, 7)
  if (N >= 16) {
    BIVX_NET_SORT8(15, 14, 13, 12, 11, 10, 9, 8)
#pragma unroll
    for (uint32_t d = 8; d > 0; d >>= 1)
#pragma unroll
      for (uint32_t i = 0; i < 16u; ++i)
        if ((i & d) == 0) BIVX_NET_CE(i, i + d)
  }
  if (N >= 32) {
    // the second sixteen DESCENDING (eight up, eight down, merged with the larger of a pair at the lower index) ...
    BIVX_NET_SORT8(16, 17, 18, 19, 20, 21, 22, 23)
    BIVX_NET_SORT8(31, 30, 29, 28, 27, 26, 25, 24)
#pragma unroll
    for (uint32_t d = 8; d > 0; d >>= 1)
#pragma unroll
      for (uint32_t i = 16; i < 32u; ++i)
        if (((i - 16u) & d) == 0) BIVX_NET_CE(i + d, i)
    // ... and the bitonic merge of the thirty-two
#pragma unroll
    for (uint32_t d = 16; d > 0; d >>= 1)
#pragma unroll
      for (uint32_t i = 0; i < 32u; ++i)
        if ((i & d) == 0) BIVX_NET_CE(i, i + d)
  }
#pragma unroll
  for (uint32_t k = 0; k < N; ++k)
    if (k < n) outb[off + k] = x[k];
}
// ... and up to 64 ids with 32 registers, in place: the first thirty-two ordered and put back, the rest ordered in
// registers, one comparator step between the two halves (the smaller of a pair stays in LDS, the larger in its register:
// LDS then holds the smallest thirty-two, the registers the rest, both bitonic), and a bitonic merge of each half.
__device__ __forceinline__ void net_sort_list64(uint32_t *in, uint32_t off, uint32_t n) {
  uint32_t x[32];
  auto merge32 = [&] {
#pragma unroll
    for (uint32_t d = 16; d > 0; d >>= 1)
#pragma unroll
      for (uint32_t i = 0; i < 32u; ++i)
        if ((i & d) == 0) BIVX_NET_CE(i, i + d)
  };
  net_sort_list<32>(in, in, off, n < 32u ? n : 32u);
  if (!__any(n > 32u)) return;
  const uint32_t nb = n > 32u ? n - 32u : 0u;  // ids of the second half
#pragma unroll
  for (uint32_t k = 0; k < 32u; ++k) x[k] = k < nb ? in[off + 32u + k] : 0xFFFFFFFFu;
  BIVX_NET_SORT8(0, 1, 2, 3, 4, 5, 6, 7)
  BIVX_NET_SORT8(15, 14, 13, 12, 11, 10, 9, 8)
#pragma unroll
  for (uint32_t d = 8; d > 0; d >>= 1)
#pragma unroll
    for (uint32_t i = 0; i < 16u; ++i)
      if ((i & d) == 0) BIVX_NET_CE(i, i + d)
  BIVX_NET_SORT8(16, 17, 18, 19, 20, 21, 22, 23)
  BIVX_NET_SORT8(31, 30, 29, 28, 27, 26, 25, 24)
#pragma unroll
  for (uint32_t d = 8; d > 0; d >>= 1)
#pragma unroll
    for (uint32_t i = 16; i < 32u; ++i)
      if (((i - 16u) & d) == 0) BIVX_NET_CE(i + d, i)
  merge32();  // x ascending
  // the step between the halves: the first half's i-th against the second half's (31 - i)-th
  const uint32_t na = n < 32u ? n : 32u;
#pragma unroll
  for (uint32_t i = 0; i < 32u; ++i) {
    const uint32_t a = i < na ? in[off + i] : 0xFFFFFFFFu;
    const uint32_t lo = min(a, x[31u - i]);
    x[31u - i] = max(a, x[31u - i]);
    if (i < na) in[off + i] = lo;
  }
  merge32();
#pragma unroll
  for (uint32_t k = 0; k < 32u; ++k)
    if (k < nb) in[off + 32u + k] = x[k];
#pragma unroll
  for (uint32_t k = 0; k < 32u; ++k) x[k] = k < na ? in[off + k] : 0xFFFFFFFFu;
  merge32();
#pragma unroll
  for (uint32_t k = 0; k < 32u; ++k)
    if (k < na) in[off + k] = x[k];
}
#undef BIVX_NET_SORT8
#undef BIVX_NET_CE

// Sorts the 64 hit lists of one wavefront, hits[o0 .. o1) per lane (adjacent in memory, lane order), ascending,
// in place. `lds` is the wavefront's own stage of LDSN words. All 64 lanes must call it.
// SKIP_SORTED: look first whether every list is ascending already, and leave then.
template <uint32_t LDSN, uint32_t RB, bool SKIP_SORTED = false>
__device__ __forceinline__ void wave_sort_lists(uint32_t *lds, uint64_t o0, uint64_t o1, uint32_t *hits, int lane) {
  const uint64_t cnt = o1 - o0;
  // Fast path, the usual case: every list of the wavefront has at most 64 ids and the 64 lists fit the stage. The
  // region is loaded with coalesced reads, every lane orders its own list where it lies (a sorting network on
  // registers), and the region is streamed back coalesced.
  const uint64_t wb = __shfl((unsigned long long)o0, 0, kWave);
  const uint64_t we = __shfl((unsigned long long)o1, kWave - 1, kWave);
  if (__all(cnt <= 64u) && we - wb <= LDSN) {
    const uint32_t wtotal = (uint32_t)(we - wb);
    uint32_t *in = lds, *outb = lds;  // (every lane orders its own list where it lies)
    if (__any(cnt > 1)) {
      {  // the region in, sixteen bytes per lane and instruction (the region begins at any 4-byte boundary: gfx950's
         // unaligned access mode), the loads of a trip leaving together
        typedef uint32_t u32x4_lds __attribute__((ext_vector_type(4)));
        const uint32_t n4 = wtotal & ~3u;
        const uint32_t *src = hits + wb;
        for (uint32_t i0 = (uint32_t)lane * 4u; i0 < n4; i0 += kWave * 16u) {
          u32x4_a4 v[4];
#pragma unroll
          for (uint32_t u = 0; u < 4; ++u)
            if (i0 + u * kWave * 4u < n4) v[u] = *reinterpret_cast<const u32x4_a4 *>(src + i0 + u * kWave * 4u);
#pragma unroll
          for (uint32_t u = 0; u < 4; ++u)
            if (i0 + u * kWave * 4u < n4) {
              u32x4_lds w;
              w.x = v[u].x; w.y = v[u].y; w.z = v[u].z; w.w = v[u].w;
              *reinterpret_cast<u32x4_lds *>(in + i0 + u * kWave * 4u) = w;
            }
        }
        if (n4 + (uint32_t)lane < wtotal) in[n4 + lane] = src[n4 + lane];
      }
      wave_sync_mem();
      // Already ascending (position-sorted input has ids in index order inside a length class, so a sorted VCF
      // mostly arrives this way): nothing to rank and nothing to write back.
      if (SKIP_SORTED) {
        bool ascending = true;
        for (uint32_t i = 1; i < (uint32_t)cnt; ++i)
          ascending = ascending && in[(uint32_t)(o0 - wb) + i - 1] < in[(uint32_t)(o0 - wb) + i];
        if (__all(ascending)) return;
      }
      {
        // a sorting network on registers, sized by the longest list of the wavefront
        const uint32_t off = (uint32_t)(o0 - wb), n = (uint32_t)cnt;
        const uint32_t longest = wave_max(n);
        if (longest <= 8u)
          net_sort_list<8>(in, outb, off, n);
        else if (longest <= 16u)
          net_sort_list<16>(in, outb, off, n);
        else if (longest <= 32u)
          net_sort_list<32>(in, outb, off, n);
        else
          net_sort_list64(in, off, n);
      }
      wave_sync_mem();
      {  // ... and out the same way
        typedef uint32_t u32x4_lds __attribute__((ext_vector_type(4)));
        const uint32_t n4 = wtotal & ~3u;
        uint32_t *dst = hits + wb;
        for (uint32_t i0 = (uint32_t)lane * 4u; i0 < n4; i0 += kWave * 4u) {
          const u32x4_lds w = *reinterpret_cast<const u32x4_lds *>(outb + i0);
          u32x4_a4 v;
          v.x = w.x; v.y = w.y; v.z = w.z; v.w = w.w;
          *reinterpret_cast<u32x4_a4 *>(dst + i0) = v;
        }
        if (n4 + (uint32_t)lane < wtotal) dst[n4 + lane] = outb[n4 + lane];
      }
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

inline bool fits_lds(const IndexView &v) { return v.nseg <= kLdsSegs && v.nchrom <= kLdsChroms; }
inline unsigned tiles_for(size_t q) { return (unsigned)((q + kQThreads - 1) / kQThreads); }

}  // namespace
}  // namespace bivx

#endif  // BIVX_QUERY_DEVICE_H_
