// query.hip — batched overlap queries against the built index (gfx950, wave64).
//
// Replaces IntervalTree::find_overlaps / find_overlap (reference interval_tree.hpp:290-334) for a batch:
// one lane per query; per (chromosome, length class) segment two directory-guided searches in the
// start-sorted array bound the candidate window [a, b), and the closed-interval predicate
// q.low <= high (&& low <= q.high, implied by b) is evaluated on every candidate. Windows longer than
// kHeavy are scanned by the whole wavefront with coalesced loads and __ballot compaction.
// Integer compare/index work only: no MFMA anywhere.
#include "common.h"

namespace bivx {
namespace {

constexpr int kQThreads = 256;
constexpr uint32_t kHeavy = 96;  // candidate-window length above which the wavefront scans cooperatively

__device__ __forceinline__ SegDesc load_seg(const SegDesc *p) {
  const uint4 *q = reinterpret_cast<const uint4 *>(p);
  const uint4 u = q[0], w = q[1];
  SegDesc d;
  d.begin = u.x; d.end = u.y; d.base = u.z; d.last = u.w;
  d.shift = w.x; d.table_off = w.y; d.maxlen = w.z; d.ncell = w.w;
  return d;
}

// first slot in the segment whose low is >= x
__device__ __forceinline__ uint32_t seg_lower_bound(const IndexView &v, const SegDesc &d, uint32_t x) {
  if (x <= d.base) return d.begin;
  if (x > d.last) return d.end;
  const uint32_t cell = (x - d.base) >> d.shift;
  uint32_t a = v.table[d.table_off + cell], b = v.table[d.table_off + cell + 1];
  while (a < b) {
    const uint32_t m = (a + b) >> 1;
    if (v.se[m].x < x) a = m + 1; else b = m;
  }
  return a;
}

// first slot in the segment whose low is > x
__device__ __forceinline__ uint32_t seg_upper_bound(const IndexView &v, const SegDesc &d, uint32_t x) {
  if (x < d.base) return d.begin;
  if (x >= d.last) return d.end;
  const uint32_t cell = (x - d.base) >> d.shift;
  uint32_t a = v.table[d.table_off + cell], b = v.table[d.table_off + cell + 1];
  while (a < b) {
    const uint32_t m = (a + b) >> 1;
    if (v.se[m].x <= x) a = m + 1; else b = m;
  }
  return a;
}

// candidate window of query [lo, hi] in one segment: every hit has low <= hi and
// low >= high - maxlen >= lo - maxlen (also for low > high entries, whose length counts as 0).
__device__ __forceinline__ void seg_window(const IndexView &v, const SegDesc &d, uint32_t lo, uint32_t hi,
                                           uint32_t &a, uint32_t &b) {
  const uint32_t x = lo > d.maxlen ? lo - d.maxlen : 0u;
  a = seg_lower_bound(v, d, x);
  b = seg_upper_bound(v, d, hi);
  if (b < a) b = a;  // only for lo > hi queries
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

// One kernel body for count / fill / any so the three can never disagree on the hit set.
template <Mode M>
__global__ __launch_bounds__(kQThreads) void k_query(IndexView v, const uint32_t *__restrict__ qchrom,
                                                     const uint32_t *__restrict__ qlow,
                                                     const uint32_t *__restrict__ qhigh, size_t nq,
                                                     const uint64_t *__restrict__ offsets,
                                                     uint32_t *__restrict__ out) {
  const size_t q = (size_t)blockIdx.x * kQThreads + threadIdx.x;
  const int lane = threadIdx.x & (kWave - 1);
  const bool valid = q < nq;

  uint32_t lo = 0, hi = 0, s0 = 0, nseg = 0;
  if (valid) {
    lo = qlow[q];
    hi = qhigh[q];
    const uint32_t c = qchrom ? qchrom[q] : 0u;
    if (c < v.nchrom) {
      s0 = v.chrom_seg[c];
      nseg = v.chrom_seg[c + 1] - s0;
    }
  }

  uint32_t acc = (M == Mode::Any) ? BIVX_NO_HIT : 0u;  // count, or running minimum id
  uint32_t *dst = nullptr;                               // Fill: next output slot of this query
  if (M == Mode::Fill && valid) dst = out + offsets[q];

  // the segment loop is wavefront-uniform (all 64 lanes stay converged) so the cooperative path below
  // may use __ballot / __shfl safely
  for (uint32_t k = 0; __any(k < nseg); ++k) {
    uint32_t a = 0, b = 0;
    if (k < nseg) {
      const SegDesc d = load_seg(v.seg + s0 + k);
      seg_window(v, d, lo, hi, a, b);
    }
    const bool heavy = (b - a) > kHeavy;
    if (!heavy) {
      for (uint32_t i = a; i < b; ++i) {
        if (v.se[i].y >= lo) {
          if (M == Mode::Count) ++acc;
          if (M == Mode::Fill) *dst++ = v.id[i];
          if (M == Mode::Any) acc = min(acc, v.id[i]);
        }
      }
    }
    uint64_t hm = __ballot(heavy);
    while (hm) {
      const int src = __ffsll((long long)hm) - 1;
      hm &= hm - 1;
      const uint32_t ca = __shfl(a, src, kWave), cb = __shfl(b, src, kWave), cl = __shfl(lo, src, kWave);
      if (M == Mode::Count) {
        uint32_t c = 0;
        for (uint32_t i = ca + lane; i < cb; i += kWave) c += (v.se[i].y >= cl) ? 1u : 0u;
        c = wave_sum(c);
        if (lane == src) acc += c;
      } else if (M == Mode::Any) {
        uint32_t m = BIVX_NO_HIT;
        for (uint32_t i = ca + lane; i < cb; i += kWave)
          if (v.se[i].y >= cl) m = min(m, v.id[i]);
        m = wave_min(m);
        if (lane == src) acc = min(acc, m);
      } else {
        // ballot compaction keeps ascending slot order, so Fill's output order does not depend on
        // which path a window took
        const unsigned long long dp = __shfl((unsigned long long)(uintptr_t)dst, src, kWave);
        uint32_t *cdst = reinterpret_cast<uint32_t *>((uintptr_t)dp);
        uint32_t written = 0;
        for (uint32_t i0 = ca; i0 < cb; i0 += kWave) {
          const uint32_t i = i0 + lane;
          const bool hit = i < cb && v.se[i].y >= cl;
          const uint64_t m = __ballot(hit);
          if (hit) cdst[written + __popcll(m & ((1ull << lane) - 1ull))] = v.id[i];
          written += __popcll(m);
        }
        if (lane == src) dst += written;
      }
    }
  }

  if (valid) {
    if (M == Mode::Count) out[q] = acc;
    if (M == Mode::Any) out[q] = acc;
  }
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
  __shared__ uint32_t lds[kQThreads / kWave][kSortLds];
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

}  // namespace

int launch_count(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                 size_t q, uint32_t *d_counts, hipStream_t s) {
  if (q == 0) return 0;
  const unsigned nb = (unsigned)((q + kQThreads - 1) / kQThreads);
  hipLaunchKernelGGL(k_query<Mode::Count>, dim3(nb), dim3(kQThreads), 0, s, v, d_qchrom, d_qlow, d_qhigh, q,
                     (const uint64_t *)nullptr, d_counts);
  BIVX_HIP(hipGetLastError());
  return 0;
}

int launch_fill(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                size_t q, const uint64_t *d_offsets, uint32_t *d_hits, hipStream_t s) {
  if (q == 0) return 0;
  const unsigned nb = (unsigned)((q + kQThreads - 1) / kQThreads);
  hipLaunchKernelGGL(k_query<Mode::Fill>, dim3(nb), dim3(kQThreads), 0, s, v, d_qchrom, d_qlow, d_qhigh, q,
                     d_offsets, d_hits);
  BIVX_HIP(hipGetLastError());
  return 0;
}

int launch_any(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
               size_t q, uint32_t *d_first, hipStream_t s) {
  if (q == 0) return 0;
  const unsigned nb = (unsigned)((q + kQThreads - 1) / kQThreads);
  hipLaunchKernelGGL(k_query<Mode::Any>, dim3(nb), dim3(kQThreads), 0, s, v, d_qchrom, d_qlow, d_qhigh, q,
                     (const uint64_t *)nullptr, d_first);
  BIVX_HIP(hipGetLastError());
  return 0;
}

int launch_sort_hits(const uint64_t *d_offsets, uint32_t *d_hits, size_t q, hipStream_t s) {
  if (q == 0) return 0;
  const unsigned nb = (unsigned)((q + kQThreads - 1) / kQThreads);
  hipLaunchKernelGGL(k_sort_hits, dim3(nb), dim3(kQThreads), 0, s, d_offsets, d_hits, q);
  BIVX_HIP(hipGetLastError());
  return 0;
}

}  // namespace bivx
