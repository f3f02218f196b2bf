// query.hip — batched overlap queries against the built index (gfx950, wave64).
//
// Replaces IntervalTree::find_overlaps / find_overlap (reference interval_tree.hpp:290-334) for a batch.
// One lane per query. Per (chromosome, length class) segment ONE bucket-directory load gives the slot to
// start from: the first slot of the cell holding x = q.low - maxlen. The lane then walks the start-sorted
// (low, high) pairs while low <= q.high and tests q.low <= high on each; slots before x need no separate
// search because low < q.low - maxlen implies high < q.low. Walks longer than kLight slots are finished by
// the whole wavefront (coalesced loads, __ballot compaction), so a chromosome-scale query costs
// O(window / 64) wave steps instead of stalling one lane.
//
// Segment descriptors are staged through LDS. Integer compare/index work only: no MFMA anywhere.
//
// Two ways in:  k_query<Count|Fill|Any>  (two-pass API: count -> offsets scan -> fill), and
//               k_query_fused            (single pass: count, chained prefix across workgroups with a
//                                         decoupled look-back, then fill into a caller-sized buffer).
#include "common.h"

namespace bivx {
namespace {

constexpr int kQThreads = 256;
constexpr int kQWaves = kQThreads / kWave;
constexpr uint32_t kLight = 32;      // window slots a lane reads by itself; longer windows go to the wavefront
constexpr uint32_t kLdsSegs = 128;   // descriptors staged in LDS (4 KiB) ...
constexpr uint32_t kLdsChroms = 511; // ... with chrom_seg (2 KiB); larger indexes read them from global

__device__ __forceinline__ SegDesc load_seg(const SegDesc *p) {
  const uint4 *q = reinterpret_cast<const uint4 *>(p);
  const uint4 u = q[0], w = q[1];
  SegDesc d;
  d.begin = u.x; d.end = u.y; d.base = u.z; d.last = u.w;
  d.shift = w.x; d.table_off = w.y; d.maxlen = w.z; d.ncell = w.w;
  return d;
}

// Candidate window [a, b) of query [lo, hi] in segment d, straight from the bucket directory with no
// refinement: a = first slot of the cell holding x = lo - maxlen, b = one past the last slot of the cell
// holding hi. Every hit has low <= hi and low >= high - maxlen >= lo - maxlen (entries with low > high count
// as length 0 and obey the same bound), so all hits are inside; slots of the two edge cells that are not hits
// fail `low <= hi && high >= lo`, which is evaluated on every candidate anyway. The two directory loads are
// independent of each other, and so are all candidate loads once a and b are known.
__device__ __forceinline__ void seg_window(const IndexView &v, const SegDesc &d, uint32_t lo, uint32_t hi,
                                           uint32_t &a, uint32_t &b) {
  const uint32_t x = lo > d.maxlen ? lo - d.maxlen : 0u;
  if (hi < d.base || x > d.last) {
    a = b = d.end;
    return;
  }
  const uint32_t *t = v.table + d.table_off;
  const uint32_t ca = x <= d.base ? 0u : (x - d.base) >> d.shift;
  const uint32_t cb = hi >= d.last ? d.ncell : ((hi - d.base) >> d.shift) + 1u;
  a = t[ca];  // t[0] == d.begin, t[ncell] == d.end
  b = t[cb];
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

enum class Mode { Count, Fill, Any };

// Per-lane query state shared by every kernel.
struct Query {
  uint32_t lo, hi;
  uint32_t s0, nseg;  // segments [s0, s0 + nseg) of the query's chromosome
};

// What a lane remembers from a counting walk so that the fill needs no second look at (low, high):
// valid when the query touched one segment and its window fitted the lane budget.
struct Replay {
  uint32_t a2;    // even-aligned first slot of the window
  uint32_t mask;  // bit j set: slot a2 + j is a hit
  bool ok;
};

// The whole hit enumeration of one query per lane, wavefront-converged (all 64 lanes must call it).
//   Count: returns the number of hits (and fills *rp).   Any: returns the smallest hit id (BIVX_NO_HIT if none).
//   Fill:  writes hit ids to hits_base[dst_pos ..), in index order, only positions below `cap`;
//          returns the number of hits.
// Lane budget: windows of at most kLight slots are read by the lane itself, two slots per 16-byte load and
// four loads in flight; longer windows are read by the whole wavefront, one coalesced 512-byte row per step.
template <Mode M>
__device__ __forceinline__ uint32_t enumerate_hits(const IndexView &v, const SegDesc *segs, const Query &qy,
                                                   uint32_t *hits_base, uint64_t dst_pos, uint64_t cap,
                                                   Replay *rp) {
  const int lane = threadIdx.x & (kWave - 1);
  uint32_t acc = (M == Mode::Any) ? BIVX_NO_HIT : 0u;
  const uint32_t lo = qy.lo, hi = qy.hi;
  const uint4 *pairs = reinterpret_cast<const uint4 *>(v.se);
  if (M == Mode::Count && rp) {
    rp->a2 = 0;
    rp->mask = 0;
    rp->ok = qy.nseg <= 1;
  }
  // the segment loop is wavefront-uniform so the cooperative part may use __ballot / __shfl
  for (uint32_t k = 0; __any(k < qy.nseg); ++k) {
    uint32_t a = 0, b = 0;
    if (k < qy.nseg) {
      const SegDesc d = load_seg(segs + qy.s0 + k);
      seg_window(v, d, lo, hi, a, b);
    }
    const uint32_t a2 = a & ~1u;
    const bool heavy = b > a && (b - a2) > kLight;
    if (b > a && !heavy) {
      uint32_t mask = 0;
#pragma unroll
      for (uint32_t c0 = 0; c0 < kLight; c0 += 8) {
        if (a2 + c0 < b) {
          uint4 r[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t s = a2 + c0 + 2 * j;
            if (s < b) r[j] = pairs[s >> 1];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t s = a2 + c0 + 2 * j;
            if (s < b) {
              if (s >= a && r[j].x <= hi && r[j].y >= lo) mask |= 1u << (c0 + 2 * j);
              if (s + 1 < b && r[j].z <= hi && r[j].w >= lo) mask |= 1u << (c0 + 2 * j + 1);
            }
          }
        }
      }
      if (M == Mode::Count) {
        acc += (uint32_t)__popc(mask);
        if (rp) {
          rp->a2 = a2;
          rp->mask = mask;
        }
      } else {
        while (mask) {
          const uint32_t j = (uint32_t)__ffs((int)mask) - 1u;
          mask &= mask - 1;
          const uint32_t hid = v.id[a2 + j];
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
      const uint32_t ca = __shfl(a, src, kWave), cb = __shfl(b, src, kWave);
      const uint32_t cl = __shfl(lo, src, kWave), ch = __shfl(hi, src, kWave);
      if (M == Mode::Count) {
        uint32_t c = 0;
        for (uint32_t j = ca + lane; j < cb; j += kWave) {
          const uint2 e = v.se[j];
          c += (e.x <= ch && e.y >= cl) ? 1u : 0u;
        }
        c = wave_sum(c);
        if (lane == src) acc += c;
      } else if (M == Mode::Any) {
        uint32_t m = BIVX_NO_HIT;
        for (uint32_t j = ca + lane; j < cb; j += kWave) {
          const uint2 e = v.se[j];
          if (e.x <= ch && e.y >= cl) m = min(m, v.id[j]);
        }
        m = wave_min(m);
        if (lane == src) acc = min(acc, m);
      } else {
        // ballot compaction keeps ascending slot order: the output does not depend on which path ran
        const uint64_t pos0 = __shfl((unsigned long long)(dst_pos + acc), src, kWave);
        uint32_t written = 0;
        for (uint32_t j0 = ca; j0 < cb; j0 += kWave) {
          const uint32_t j = j0 + lane;
          bool hit = false;
          if (j < cb) {
            const uint2 e = v.se[j];
            hit = e.x <= ch && e.y >= cl;
          }
          const uint64_t m = __ballot(hit);
          if (hit) {
            const uint64_t p = pos0 + written + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (p < cap) hits_base[p] = v.id[j];
          }
          written += (uint32_t)__popcll(m);
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
    for (uint32_t t = threadIdx.x; t < v.nseg * 2; t += kQThreads) dst[t] = src[t];
    for (uint32_t t = threadIdx.x; t <= v.nchrom; t += kQThreads) s_cs[t] = v.chrom_seg[t];
    segs = s_seg;
    cs = s_cs;
  } else {
    segs = v.seg;
    cs = v.chrom_seg;
  }
}

__device__ __forceinline__ Query load_query(const IndexView &v, const uint32_t *cs, const uint32_t *qchrom,
                                            const uint32_t *qlow, const uint32_t *qhigh, size_t q, bool valid) {
  Query qy{0u, 0u, 0u, 0u};
  if (valid) {
    qy.lo = qlow[q];
    qy.hi = qhigh[q];
    const uint32_t c = qchrom ? qchrom[q] : 0u;
    if (c < v.nchrom) {
      qy.s0 = cs[c];
      qy.nseg = cs[c + 1] - qy.s0;
    }
  }
  return qy;
}

// ---- two-pass kernels ----------------------------------------------------------------------------------
template <Mode M, bool LDS_DESC>
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
  const Query qy = load_query(v, cs, qchrom, qlow, qhigh, q, valid);
  uint64_t pos = 0;
  if (M == Mode::Fill && valid) pos = offsets[q];
  const uint32_t r = enumerate_hits<M>(v, segs, qy, out, pos, ~0ull, nullptr);
  if (valid && M != Mode::Fill) out[q] = r;
}

// ---- single-pass kernel ----------------------------------------------------------------------------------
// ws[0] (low 32 bits): tile ticket; ws[1 + t]: status of tile t = state << 62 | value, written and polled as
// ONE 8-byte agent-scope atomic so the value needs no separate fence (per-XCD L2s are not coherent; sc1
// accesses go to memory). state 1 = tile aggregate, 2 = inclusive prefix. Tiles take tickets in launch order,
// so every predecessor of a polling tile is already resident: the chain cannot deadlock. Spins are bounded.
constexpr uint64_t kStAgg = 1ull << 62, kStInc = 2ull << 62, kStMask = 3ull << 62;
constexpr uint32_t kSpinCap = 1u << 22;

__device__ __forceinline__ uint64_t ld_status(const uint64_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_status(uint64_t *p, uint64_t w) {
  __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool LDS_DESC>
__global__ __launch_bounds__(kQThreads) void k_query_fused(IndexView v, const uint32_t *__restrict__ qchrom,
                                                           const uint32_t *__restrict__ qlow,
                                                           const uint32_t *__restrict__ qhigh, size_t nq,
                                                           uint64_t *__restrict__ offsets,
                                                           uint32_t *__restrict__ hits, uint64_t cap,
                                                           uint64_t *__restrict__ ws) {
  __shared__ SegDesc s_seg[LDS_DESC ? kLdsSegs : 1];
  __shared__ uint32_t s_cs[LDS_DESC ? kLdsChroms + 1 : 1];
  __shared__ uint32_t s_tile;
  __shared__ uint32_t s_wsum[kQWaves];
  __shared__ uint64_t s_base;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;

  if (threadIdx.x == 0) s_tile = atomicAdd(reinterpret_cast<unsigned int *>(ws), 1u);
  const SegDesc *segs;
  const uint32_t *cs;
  stage_descriptors<LDS_DESC>(v, s_seg, s_cs, segs, cs);
  __syncthreads();
  const uint32_t tile = s_tile;
  uint64_t *status = ws + 1;

  const size_t q = (size_t)tile * kQThreads + threadIdx.x;
  const bool valid = q < nq;
  const Query qy = load_query(v, cs, qchrom, qlow, qhigh, q, valid);

  // phase 1: count
  Replay rp;
  const uint32_t cnt = enumerate_hits<Mode::Count>(v, segs, qy, nullptr, 0, 0, &rp);

  // workgroup exclusive scan of the counts
  uint32_t incl = cnt;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d, kWave);
    if (lane >= d) incl += o;
  }
  if (lane == kWave - 1) s_wsum[wave] = incl;
  __syncthreads();
  uint32_t wbase = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kQWaves; ++w) {
    const uint32_t s = s_wsum[w];
    if (w < wave) wbase += s;
    total += s;
  }
  const uint32_t local = wbase + incl - cnt;

  // chained prefix across tiles: wave 0 publishes the aggregate, looks back, publishes the inclusive prefix
  if (wave == 0) {
    uint64_t excl = 0;
    if (tile == 0) {
      if (lane == 0) st_status(&status[0], kStInc | (uint64_t)total);
    } else {
      if (lane == 0) st_status(&status[tile], kStAgg | (uint64_t)total);
      int64_t look = (int64_t)tile - 1;
      uint32_t spins = 0;
      while (true) {
        const int64_t t = look - lane;
        uint64_t w = kStInc;  // tiles before 0: inclusive prefix 0
        if (t >= 0) w = ld_status(&status[t]);
        while (__any((w & kStMask) == 0) && spins < kSpinCap) {
          __builtin_amdgcn_s_sleep(2);
          if ((w & kStMask) == 0) w = ld_status(&status[t]);
          ++spins;
        }
        const uint64_t inc_mask = __ballot((w & kStMask) == kStInc);
        const uint64_t val = w & ~kStMask;
        if (inc_mask) {
          const int first = __ffsll((long long)inc_mask) - 1;
          uint64_t part = lane <= first ? val : 0ull;
#pragma unroll
          for (int d = 32; d > 0; d >>= 1) part += __shfl_xor((unsigned long long)part, d, kWave);
          excl += part;
          break;
        }
        uint64_t part = val;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) part += __shfl_xor((unsigned long long)part, d, kWave);
        excl += part;
        look -= kWave;
        if (spins >= kSpinCap) break;  // never expected; leaves a wrong prefix instead of a hung GPU
      }
      if (lane == 0) st_status(&status[tile], kStInc | (excl + total));
    }
    if (lane == 0) s_base = excl;
  }
  __syncthreads();
  const uint64_t pos = s_base + local;
  if (valid) {
    offsets[q] = pos;
    if (q == nq - 1) offsets[nq] = pos + cnt;
  }

  // phase 2: fill. A lane whose walk was recorded replays the hit mask (ids only, no second look at the
  // (low, high) pairs); the others enumerate again with the lines of phase 1 still in this CU's L1/L2.
  if (rp.ok) {
    uint32_t mask = rp.mask, k = 0;
    while (mask) {
      const uint32_t j = (uint32_t)__ffs((int)mask) - 1u;
      mask &= mask - 1;
      const uint32_t hid = v.id[rp.a2 + j];
      if (pos + k < cap) hits[pos + k] = hid;
      ++k;
    }
  }
  Query q2 = qy;
  if (rp.ok) q2.nseg = 0;
  (void)enumerate_hits<Mode::Fill>(v, segs, q2, hits, pos, cap, nullptr);
}

// ---- per-query ascending-id ordering of a CSR hit list ------------------------------------------------

constexpr uint32_t kSortLane = 24;    // <= this many hits: the owning lane insertion-sorts in place
constexpr uint32_t kSortLds = 2048;   // <= this many: the wavefront bitonic-sorts through LDS

__device__ __forceinline__ void wave_sync_mem() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

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

__global__ __launch_bounds__(kQThreads) void k_sort_hits(const uint64_t *__restrict__ offsets,
                                                         uint32_t *__restrict__ hits, size_t nq) {
  __shared__ uint32_t lds[kQWaves][kSortLds];
  const size_t q = (size_t)blockIdx.x * kQThreads + threadIdx.x;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  uint64_t o0 = 0, o1 = 0;
  if (q < nq) {
    o0 = offsets[q];
    o1 = offsets[q + 1];
  }
  const uint64_t cnt = o1 - o0;
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
    if (n <= kSortLds) {
      uint32_t *s = lds[wave];
      for (uint32_t i = lane; i < (uint32_t)n; i += kWave) s[i] = h[i];
      wave_sync_mem();
      wave_bitonic_sort<uint32_t>(s, (uint32_t)n, lane);
      for (uint32_t i = lane; i < (uint32_t)n; i += kWave) h[i] = s[i];
    } else {
      wave_bitonic_sort<uint64_t>(h, n, lane);  // very long hit lists: same network in global memory
    }
    wave_sync_mem();
  }
}

inline bool fits_lds(const IndexView &v) { return v.nseg <= kLdsSegs && v.nchrom <= kLdsChroms; }
inline unsigned tiles_for(size_t q) { return (unsigned)((q + kQThreads - 1) / kQThreads); }

template <Mode M>
int launch_query(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                 size_t q, const uint64_t *d_offsets, uint32_t *d_out, hipStream_t s) {
  if (q == 0) return 0;
  if (fits_lds(v))
    hipLaunchKernelGGL((k_query<M, true>), dim3(tiles_for(q)), dim3(kQThreads), 0, s, v, d_qchrom, d_qlow, d_qhigh,
                       q, d_offsets, d_out);
  else
    hipLaunchKernelGGL((k_query<M, false>), dim3(tiles_for(q)), dim3(kQThreads), 0, s, v, d_qchrom, d_qlow, d_qhigh,
                       q, d_offsets, d_out);
  BIVX_HIP(hipGetLastError());
  return 0;
}

}  // namespace

int launch_count(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                 size_t q, uint32_t *d_counts, hipStream_t s) {
  return launch_query<Mode::Count>(v, d_qchrom, d_qlow, d_qhigh, q, nullptr, d_counts, s);
}

int launch_fill(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                size_t q, const uint64_t *d_offsets, uint32_t *d_hits, hipStream_t s) {
  return launch_query<Mode::Fill>(v, d_qchrom, d_qlow, d_qhigh, q, d_offsets, d_hits, s);
}

int launch_any(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
               size_t q, uint32_t *d_first, hipStream_t s) {
  return launch_query<Mode::Any>(v, d_qchrom, d_qlow, d_qhigh, q, nullptr, d_first, s);
}

size_t fused_workspace_bytes(size_t q) { return ((size_t)tiles_for(q) + 2) * sizeof(uint64_t); }

int launch_query_fused(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow,
                       const uint32_t *d_qhigh, size_t q, uint64_t *d_offsets, uint32_t *d_hits, uint64_t cap,
                       void *d_ws, hipStream_t s) {
  if (q == 0) {
    BIVX_HIP(hipMemsetAsync(d_offsets, 0, sizeof(uint64_t), s));
    return 0;
  }
  BIVX_HIP(hipMemsetAsync(d_ws, 0, fused_workspace_bytes(q), s));  // ticket + tile status words
  uint64_t *ws = static_cast<uint64_t *>(d_ws);
  if (fits_lds(v))
    hipLaunchKernelGGL((k_query_fused<true>), dim3(tiles_for(q)), dim3(kQThreads), 0, s, v, d_qchrom, d_qlow,
                       d_qhigh, q, d_offsets, d_hits, cap, ws);
  else
    hipLaunchKernelGGL((k_query_fused<false>), dim3(tiles_for(q)), dim3(kQThreads), 0, s, v, d_qchrom, d_qlow,
                       d_qhigh, q, d_offsets, d_hits, cap, ws);
  BIVX_HIP(hipGetLastError());
  return 0;
}

int launch_sort_hits(const uint64_t *d_offsets, uint32_t *d_hits, size_t q, hipStream_t s) {
  if (q == 0) return 0;
  hipLaunchKernelGGL(k_sort_hits, dim3(tiles_for(q)), dim3(kQThreads), 0, s, d_offsets, d_hits, q);
  BIVX_HIP(hipGetLastError());
  return 0;
}

}  // namespace bivx
