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
// Kernels here: k_query<Fill|Any> (fill a CSR whose offsets are known; existence + smallest id) and k_sort_hits.
// The single-pass kernel (count, prefix across workgroups, fill) lives in query_fused.hip; the device code all of
// them share is query_device.h.
#include "query_device.h"

namespace bivx {
namespace {

// ---- two-pass kernels ----------------------------------------------------------------------------------
template <Mode M, bool LDS_DESC, bool F>
__global__ __launch_bounds__(kQThreads) void k_query(IndexView v, const uint32_t *__restrict__ qchrom,
                                                     const uint32_t *__restrict__ qlow,
                                                     const uint32_t *__restrict__ qhigh, size_t nq,
                                                     const uint64_t *__restrict__ offsets,
                                                     uint32_t *__restrict__ out) {
  __shared__ SegDesc s_seg[LDS_DESC ? kLdsSegs : 1];
  __shared__ uint2 s_cs[LDS_DESC ? kLdsChroms : 1];
  const SegDesc *segs;
  const uint2 *cs;
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

// LDSN: words of LDS per wavefront. The usual path needs the wavefront's 64 lists to fit it; the stage was 16 KB per
// wavefront whatever the lists — 8 wavefronts per CU, and the pass over config 5's 861 M ids took 7.8 ms, 0.9 TB/s. The host
// now picks the smallest of four sizes that holds 1.3 x the average 64 lists (it knows the total, or an upper bound).
template <uint32_t LDSN>
__global__ __launch_bounds__(kQThreads) void k_sort_hits(const uint64_t *__restrict__ offsets,
                                                         uint32_t *__restrict__ hits, size_t nq, uint64_t cap,
                                                         const uint32_t *__restrict__ cond, uint32_t seq) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[kQWaves][LDSN];
  // conditional form: the single-pass kernel ordered the ids itself unless one of its wavefronts said otherwise
  if (cond && __hip_atomic_load(cond, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq) return;
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
  wave_sort_lists<LDSN, kRankBlock, true>(lds[threadIdx.x >> 6], o0, o1, hits, threadIdx.x & (kWave - 1));
}

// One wavefront answers up to 64 queries completely — counts, their prefix sum, offsets, ids in index order — for the
// few-queries host calls (capi.hip, mailbox path): no workspace, no ticket, no second kernel. Two enumerations (count,
// then fill) of at most 64 windows. offsets[q] receives the total; ids beyond `cap` are not written.
template <bool LDS_DESC>
__global__ __launch_bounds__(kWave) void k_query_tiny(IndexView v, const uint32_t *__restrict__ qchrom,
                                                     const uint32_t *__restrict__ qlow,
                                                     const uint32_t *__restrict__ qhigh, uint32_t nq,
                                                     uint64_t *__restrict__ offsets, uint32_t *__restrict__ hits,
                                                     uint64_t cap) {
  __shared__ SegDesc s_seg[LDS_DESC ? kLdsSegs : 1];
  __shared__ uint2 s_cs[LDS_DESC ? kLdsChroms : 1];
  const SegDesc *segs;
  const uint2 *cs;
  stage_descriptors<false>(v, s_seg, s_cs, segs, cs);  // (64 queries: the descriptors are read where they lie)
  const uint32_t lane = threadIdx.x;
  const bool valid = lane < nq;
  const Query qy = load_query<false>(v, cs, qchrom, qlow, qhigh, lane, valid);
  // (MS = true on purpose: it makes the counting enumeration walk ALL of the chromosome's segments — the MS = false
  // form is the one-segment kernels' single trip through the segment loop, which would miss every later length class)
  const uint32_t cnt = enumerate_hits<Mode::Count, false, true>(v, segs, qy, nullptr, 0, 0, nullptr);
  // 64-bit exclusive prefix over the lanes (a single query may have more than 2^32 / 64 hits)
  uint64_t incl = cnt;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const uint64_t o = __shfl_up((unsigned long long)incl, d, kWave);
    if ((int)lane >= d) incl += o;
  }
  const uint64_t pos = incl - cnt;
  if (valid) offsets[lane] = pos;
  if (lane == kWave - 1) offsets[nq] = incl;
  if (cap != 0) (void)enumerate_hits<Mode::Fill, false>(v, segs, qy, hits, pos, cap, nullptr);
}

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

int launch_query_tiny(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                      size_t q, uint64_t *d_offsets, uint32_t *d_hits, uint64_t cap, hipStream_t s) {
  if (q == 0 || q > (size_t)kWave || v.flt_kind != BIVX_FILTER_NONE) return -1;
  hipLaunchKernelGGL(k_query_tiny<false>, dim3(1), dim3(kWave), 0, s, v, d_qchrom, d_qlow, d_qhigh, (uint32_t)q,
                     d_offsets, d_hits, cap);
  BIVX_HIP(hipGetLastError());
  return 0;
}

int launch_fill(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                size_t q, const uint64_t *d_offsets, uint32_t *d_hits, hipStream_t s) {
  return launch_query<Mode::Fill>(v, d_qchrom, d_qlow, d_qhigh, q, d_offsets, d_hits, s);
}

int launch_any(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
               size_t q, uint32_t *d_first, hipStream_t s) {
  return launch_query<Mode::Any>(v, d_qchrom, d_qlow, d_qhigh, q, nullptr, d_first, s);
}

int launch_sort_hits(const uint64_t *d_offsets, uint32_t *d_hits, size_t q, uint64_t cap, hipStream_t s,
                     const uint32_t *d_cond, uint32_t seq, uint64_t total_hint) {
  if (q == 0) return 0;
  // ids per 64 lists, by the total if the caller knows it, else by the buffer's capacity (an upper bound; ~0: unknown)
  const uint64_t known = total_hint ? total_hint : cap;
  const double per_wave = known == ~0ull ? 1e30 : 64.0 * (double)known / (double)q;
  if (per_wave * 1.3 <= 512.0)
    hipLaunchKernelGGL(k_sort_hits<512>, dim3(tiles_for(q)), dim3(kQThreads), 0, s, d_offsets, d_hits, q, cap, d_cond, seq);
  else if (per_wave * 1.3 <= 1024.0)
    hipLaunchKernelGGL(k_sort_hits<1024>, dim3(tiles_for(q)), dim3(kQThreads), 0, s, d_offsets, d_hits, q, cap, d_cond, seq);
  else if (per_wave * 1.3 <= 2048.0)
    hipLaunchKernelGGL(k_sort_hits<2048>, dim3(tiles_for(q)), dim3(kQThreads), 0, s, d_offsets, d_hits, q, cap, d_cond, seq);
  else
    hipLaunchKernelGGL(k_sort_hits<kSortLds>, dim3(tiles_for(q)), dim3(kQThreads), 0, s, d_offsets, d_hits, q, cap, d_cond, seq);
  BIVX_HIP(hipGetLastError());
  return 0;
}


}  // namespace bivx
