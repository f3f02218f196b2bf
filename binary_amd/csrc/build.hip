// build.hip — build-side kernels of the interval index (gfx950, wave64).
//
// Replaces the net effect of the reference's per-record tree insertion (IntervalTree::insert_node_impl
// interval_tree.hpp:230-260, RbTree::fix_insert rb_tree.hpp:304-344, rotations interval_tree.hpp:206-228)
// with a batch build: per-(chromosome, length-bin) statistics -> host picks length classes -> stable LSD
// radix sort of (segment, low) carrying the append-order id -> gather (low, high) -> bucket directory.
// All integer/byte work; bound by HBM streaming and scatter, never by MFMA.
#include "common.h"

namespace bivx {
namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ uint32_t len_bin(uint32_t low, uint32_t high, uint32_t &len) {
  len = high >= low ? high - low : 0u;  // low > high entries can only be hit inside [high, low]: length 0
  return len == 0 ? 0u : 32u - (uint32_t)__clz((int)len);
}

// ---- per (chromosome, length bin) statistics ---------------------------------------------------------

constexpr uint32_t kStatsLdsEntries = 1650;  // (chrom, bin) pairs privatised in LDS (50 chromosomes, 33 KB)

template <bool USE_LDS>
__global__ __launch_bounds__(kThreads) void k_bin_stats(const uint32_t *__restrict__ chrom,
                                                        const uint32_t *__restrict__ low,
                                                        const uint32_t *__restrict__ high, size_t n,
                                                        uint32_t nchrom, BinStats *__restrict__ stats) {
  __shared__ BinStats lds[USE_LDS ? kStatsLdsEntries : 1];
  const uint32_t nent = nchrom * kLenBins;
  if (USE_LDS) {
    for (uint32_t e = threadIdx.x; e < nent; e += kThreads) lds[e] = BinStats{0u, 0xFFFFFFFFu, 0u, 0u, 0u};
    __syncthreads();
  }
  BinStats *tab = USE_LDS ? lds : stats;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) {
    const uint32_t c = chrom ? chrom[i] : 0u;
    const uint32_t lo = low[i];
    const uint32_t hi = high[i];
    uint32_t len;
    const uint32_t b = len_bin(lo, hi, len);
    BinStats *e = tab + (size_t)c * kLenBins + b;
    atomicAdd(&e->count, 1u);
    if (lo > hi) atomicAdd(&e->n_inverted, 1u);
    atomicMin(&e->min_low, lo);
    atomicMax(&e->max_low, lo);
    atomicMax(&e->max_len, len);
  }
  if (USE_LDS) {
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < nent; e += kThreads) {
      const BinStats s = lds[e];
      if (s.count) {
        atomicAdd(&stats[e].count, s.count);
        atomicMin(&stats[e].min_low, s.min_low);
        atomicMax(&stats[e].max_low, s.max_low);
        atomicMax(&stats[e].max_len, s.max_len);
        if (s.n_inverted) atomicAdd(&stats[e].n_inverted, s.n_inverted);
      }
    }
  }
}

__global__ __launch_bounds__(kThreads) void k_init_stats(BinStats *stats, uint32_t nent) {
  const uint32_t e = blockIdx.x * kThreads + threadIdx.x;
  if (e < nent) stats[e] = BinStats{0u, 0xFFFFFFFFu, 0u, 0u, 0u};
}

__global__ __launch_bounds__(kThreads) void k_max_u32(const uint32_t *__restrict__ in, size_t n,
                                                      uint32_t *__restrict__ out) {
  uint32_t m = 0;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads)
    m = max(m, in[i]);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_xor(m, d, kWave));
  if ((threadIdx.x & (kWave - 1)) == 0) atomicMax(out, m);
}

// largest number of slots any directory cell holds: max over e of table[e + 1] - table[e] (a segment's entries end with
// its last slot + 1, which is where the next segment's begin: the difference across a boundary is 0)
__global__ __launch_bounds__(kThreads) void k_max_cell(const uint32_t *__restrict__ table, size_t nentries,
                                                       uint32_t *__restrict__ out) {
  uint32_t m = 0;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i + 1 < nentries; i += (size_t)gridDim.x * kThreads) {
    const uint32_t a = table[i], b = table[i + 1];
    m = max(m, b > a ? b - a : 0u);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_xor(m, d, kWave));
  if ((threadIdx.x & (kWave - 1)) == 0) atomicMax(out, m);
}

__global__ __launch_bounds__(kThreads) void k_max_u8(const uint8_t *__restrict__ in, size_t n,
                                                     uint32_t *__restrict__ out) {
  uint32_t m = 0;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads)
    m = max(m, (uint32_t)in[i]);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_xor(m, d, kWave));
  if ((threadIdx.x & (kWave - 1)) == 0) atomicMax(out, m);
}

// the (chromosome, svtype) partition id of every interval (include/bivx.h, bivx_append_typed)
__global__ __launch_bounds__(kThreads) void k_make_vchrom(const uint32_t *__restrict__ chrom,
                                                          const uint8_t *__restrict__ type, size_t n,
                                                          uint32_t ntypes, uint32_t *__restrict__ vchrom) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i < n) vchrom[i] = (chrom ? chrom[i] : 0u) * ntypes + type[i];
}

__global__ __launch_bounds__(kThreads) void k_gather_u8(const uint8_t *__restrict__ src,
                                                        const uint32_t *__restrict__ ids, size_t n, size_t n_src,
                                                        uint8_t *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i < n) out[i] = ids[i] < n_src ? (src ? src[ids[i]] : (uint8_t)0) : (uint8_t)0xFF;
}

// ---- sort keys -----------------------------------------------------------------------------------------

__global__ __launch_bounds__(kThreads) void k_make_segkeys(const uint32_t *__restrict__ chrom,
                                                           const uint32_t *__restrict__ low,
                                                           const uint32_t *__restrict__ high, size_t n,
                                                           const uint32_t *__restrict__ bin2seg,
                                                           uint32_t *__restrict__ segkey,
                                                           uint32_t *__restrict__ ids) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const uint32_t c = chrom ? chrom[i] : 0u;
  uint32_t len;
  const uint32_t b = len_bin(low[i], high[i], len);
  segkey[i] = bin2seg[(size_t)c * kLenBins + b];
  ids[i] = (uint32_t)i;
}

// ---- stable LSD radix sort, 8 bits per pass --------------------------------------------------------------
//
// A workgroup owns a tile of kTile consecutive keys; wave w owns the w-th quarter of the tile and lane l
// of round r the key at quarter + r*64 + l, so "position in tile" order is (wave, round, lane). Ranks
// inside a round come from a ballot match on the 8 digit bits, which needs no LDS atomics and is stable.

constexpr int kRadixBits = 8;
constexpr int kRadix = 1 << kRadixBits;
constexpr int kRounds = 4;
constexpr int kWaves = kThreads / kWave;
constexpr int kTile = kThreads * kRounds;  // 1024 keys (more rounds per thread cost ~20 VGPRs each: 16 took 256)

__device__ __forceinline__ uint64_t match_digit(uint32_t digit, bool valid) {
  uint64_t m = __ballot(valid);
#pragma unroll
  for (int b = 0; b < kRadixBits; ++b) {
    const bool bit = (digit >> b) & 1u;
    const uint64_t bal = __ballot(bit);
    m &= bit ? bal : ~bal;
  }
  return m;  // lanes (valid ones) holding the same digit as this lane
}

// hist[digit * nblocks + block] = number of keys with that digit in that tile
__global__ __launch_bounds__(kThreads) void k_radix_hist(const uint32_t *__restrict__ keys, size_t n, int shift,
                                                         uint32_t *__restrict__ hist, uint32_t nblocks) {
  __shared__ uint32_t cnt[kRadix];
  cnt[threadIdx.x] = 0;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * kTile;
#pragma unroll 4
  for (int r = 0; r < kRounds; ++r) {
    const size_t i = base + (size_t)r * kThreads + threadIdx.x;
    if (i < n) atomicAdd(&cnt[(keys[i] >> shift) & (kRadix - 1)], 1u);
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = cnt[threadIdx.x];
}

__global__ __launch_bounds__(kThreads) void k_radix_scatter(const uint32_t *__restrict__ keys_in,
                                                            const uint32_t *__restrict__ vals_in,
                                                            uint32_t *__restrict__ keys_out,
                                                            uint32_t *__restrict__ vals_out, size_t n, int shift,
                                                            const uint32_t *__restrict__ offs, uint32_t nblocks) {
  __shared__ uint32_t wcnt[kWaves][kRadix];  // per-wave digit counts, then per-wave running bases
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  for (int w = 0; w < kWaves; ++w) wcnt[w][threadIdx.x] = 0;
  __syncthreads();

  const size_t wbase = (size_t)blockIdx.x * kTile + (size_t)wave * (kTile / kWaves);
  uint32_t key[kRounds], val[kRounds];
  volatile uint32_t *mycnt = wcnt[wave];
  // phase A: load the wave's quarter, count digits (one leader lane per distinct digit adds the group size)
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const size_t i = wbase + (size_t)r * kWave + lane;
    const bool valid = i < n;
    key[r] = valid ? keys_in[i] : 0u;
    val[r] = valid ? vals_in[i] : 0u;
    const uint32_t dg = (key[r] >> shift) & (kRadix - 1);
    const uint64_t m = match_digit(dg, valid);
    if (valid && (m & ((1ull << lane) - 1ull)) == 0) mycnt[dg] += (uint32_t)__popcll(m);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  // thread t = digit t: turn per-wave counts into per-wave global bases
  {
    uint32_t run = offs[(size_t)threadIdx.x * nblocks + blockIdx.x];
    for (int w = 0; w < kWaves; ++w) {
      const uint32_t c = wcnt[w][threadIdx.x];
      wcnt[w][threadIdx.x] = run;
      run += c;
    }
  }
  __syncthreads();
  // phase B: replay the rounds in the same order; base[digit] advances by the group size each round
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const size_t i = wbase + (size_t)r * kWave + lane;
    const bool valid = i < n;
    const uint32_t dg = (key[r] >> shift) & (kRadix - 1);
    const uint64_t m = match_digit(dg, valid);
    const uint32_t below = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    uint32_t pos = 0;
    if (valid) pos = mycnt[dg] + below;
    __builtin_amdgcn_wave_barrier();
    if (valid && below == 0) mycnt[dg] += (uint32_t)__popcll(m);
    __builtin_amdgcn_wave_barrier();
    if (valid) {
      keys_out[pos] = key[r];
      vals_out[pos] = val[r];
    }
  }
}

// ---- gathers ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(kThreads) void k_gather_u32(const uint32_t *__restrict__ src,
                                                         const uint32_t *__restrict__ idx,
                                                         uint32_t *__restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}

__global__ __launch_bounds__(kThreads) void k_gather_se(const uint32_t *__restrict__ low,
                                                        const uint32_t *__restrict__ high,
                                                        const uint32_t *__restrict__ idx, uint2 *__restrict__ se,
                                                        size_t n) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i < n) {
    const uint32_t j = idx[i];
    se[i] = make_uint2(low[j], high[j]);
  }
}

__global__ __launch_bounds__(kThreads) void k_gather_intervals(const uint32_t *__restrict__ chrom,
                                                               const uint32_t *__restrict__ low,
                                                               const uint32_t *__restrict__ high,
                                                               const uint32_t *__restrict__ ids, size_t n,
                                                               size_t n_intervals, uint32_t *__restrict__ oc,
                                                               uint32_t *__restrict__ ol,
                                                               uint32_t *__restrict__ oh) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const uint32_t j = ids[i];
  const bool ok = j < n_intervals;
  if (oc) oc[i] = ok ? (chrom ? chrom[j] : 0u) : 0xFFFFFFFFu;
  if (ol) ol[i] = ok ? low[j] : 0xFFFFFFFFu;
  if (oh) oh[i] = ok ? high[j] : 0u;
}

// ---- bucket directory --------------------------------------------------------------------------------------
// table[d.table_off + c] = first slot in segment whose low >= d.base + (c << d.shift), for c in [0, ncell];
// entry ncell is the segment end. One thread per directory entry, a binary search each.

__global__ __launch_bounds__(kThreads) void k_build_table(const uint2 *__restrict__ se,
                                                          const SegDesc *__restrict__ seg, uint32_t nseg,
                                                          uint32_t *__restrict__ table, uint64_t nentries) {
  const uint64_t e = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
  if (e >= nentries) return;
  // segment owning entry e: last s with table_off <= e (entries of segment s: table_off .. table_off+ncell)
  uint32_t lo = 0, hi = nseg;
  while (hi - lo > 1) {
    const uint32_t m = (lo + hi) >> 1;
    if ((uint64_t)seg[m].table_off <= e) lo = m; else hi = m;
  }
  const SegDesc d = seg[lo];
  const uint32_t c = (uint32_t)(e - d.table_off);
  uint32_t a = d.begin, b = d.end;
  if (c < d.ncell) {
    const uint32_t x = d.base + (c << (d.shift & 31u));
    while (a < b) {
      const uint32_t m = (a + b) >> 1;
      if (se[m].x < x) a = m + 1; else b = m;
    }
  } else {
    a = d.end;
  }
  table[e] = a;
}

// ---- packed records ----------------------------------------------------------------------------------------
// rec[i] = ((low & 0xFFFF) | (high - low) << 16, id) for slots of kSegPacked segments ((0, id) elsewhere)
__global__ __launch_bounds__(kThreads) void k_pack_records(const uint2 *__restrict__ se,
                                                           const uint32_t *__restrict__ id,
                                                           const SegDesc *__restrict__ seg, uint32_t nseg,
                                                           uint2 *__restrict__ rec, size_t n) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  uint32_t lo = 0, hi = nseg;  // last segment with begin <= i
  while (hi - lo > 1) {
    const uint32_t m = (lo + hi) >> 1;
    if ((size_t)seg[m].begin <= i) lo = m; else hi = m;
  }
  const SegDesc d = seg[lo];
  uint32_t r = 0;
  if (d.shift & kSegPacked) {
    const uint2 e = se[i];
    r = (e.x & 0xFFFFu) | ((e.y - e.x) << 16);
  }
  rec[i] = make_uint2(r, id[i]);
}

inline unsigned grid_for(size_t n, int per_block, unsigned cap = 0) {
  size_t nb = (n + (size_t)per_block - 1) / (size_t)per_block;
  if (nb < 1) nb = 1;
  if (cap && nb > cap) nb = cap;
  return (unsigned)nb;
}

}  // namespace

int launch_max_cell(const uint32_t *d_table, size_t nentries, uint32_t *d_out, hipStream_t s) {
  BIVX_HIP(hipMemsetAsync(d_out, 0, 4, s));
  if (nentries < 2) return 0;
  hipLaunchKernelGGL(k_max_cell, dim3(grid_for(nentries, kThreads * 8, 2048)), dim3(kThreads), 0, s, d_table, nentries, d_out);
  BIVX_HIP(hipGetLastError());
  return 0;
}

int launch_max_u32(const uint32_t *d_in, size_t n, uint32_t *d_out, hipStream_t s) {
  BIVX_HIP(hipMemsetAsync(d_out, 0, sizeof(uint32_t), s));
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_max_u32, dim3(grid_for(n, kThreads * 8, 2048)), dim3(kThreads), 0, s, d_in, n, d_out);
  BIVX_HIP(hipGetLastError());
  return 0;
}

int launch_max_u8(const uint8_t *d_in, size_t n, uint32_t *d_out, hipStream_t s) {
  BIVX_HIP(hipMemsetAsync(d_out, 0, sizeof(uint32_t), s));
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_max_u8, dim3(grid_for(n, kThreads * 16, 2048)), dim3(kThreads), 0, s, d_in, n, d_out);
  BIVX_HIP(hipGetLastError());
  return 0;
}

int launch_make_vchrom(const uint32_t *d_chrom, const uint8_t *d_type, size_t n, uint32_t ntypes, uint32_t *d_vchrom,
                       hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_make_vchrom, dim3(grid_for(n, kThreads)), dim3(kThreads), 0, s, d_chrom, d_type, n, ntypes,
                     d_vchrom);
  BIVX_HIP(hipGetLastError());
  return 0;
}

int launch_gather_u8(const uint8_t *d_src, const uint32_t *d_ids, size_t n, size_t n_src, uint8_t *d_out,
                     hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_gather_u8, dim3(grid_for(n, kThreads)), dim3(kThreads), 0, s, d_src, d_ids, n, n_src, d_out);
  BIVX_HIP(hipGetLastError());
  return 0;
}

int launch_bin_stats(const uint32_t *d_chrom, const uint32_t *d_low, const uint32_t *d_high, size_t n,
                     uint32_t nchrom, BinStats *d_stats, hipStream_t s) {
  const uint32_t nent = nchrom * kLenBins;
  hipLaunchKernelGGL(k_init_stats, dim3(grid_for(nent, kThreads)), dim3(kThreads), 0, s, d_stats, nent);
  if (n) {
    const unsigned nb = grid_for(n, kThreads * 16, 2048);
    if (nent <= kStatsLdsEntries)
      hipLaunchKernelGGL(k_bin_stats<true>, dim3(nb), dim3(kThreads), 0, s, d_chrom, d_low, d_high, n, nchrom, d_stats);
    else
      hipLaunchKernelGGL(k_bin_stats<false>, dim3(nb), dim3(kThreads), 0, s, d_chrom, d_low, d_high, n, nchrom, d_stats);
  }
  BIVX_HIP(hipGetLastError());
  return 0;
}

int launch_make_segkeys(const uint32_t *d_chrom, const uint32_t *d_low, const uint32_t *d_high, size_t n,
                        const uint32_t *d_bin2seg, uint32_t *d_segkey, uint32_t *d_ids, hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_make_segkeys, dim3(grid_for(n, kThreads)), dim3(kThreads), 0, s, d_chrom, d_low, d_high, n,
                     d_bin2seg, d_segkey, d_ids);
  BIVX_HIP(hipGetLastError());
  return 0;
}

size_t radix_scratch_bytes(size_t n) {
  const size_t nblocks = (n + kTile - 1) / kTile;
  const size_t nh = nblocks * kRadix;
  // histogram + its exclusive scan (+1) + scan scratch
  return (nh + nh + 1) * sizeof(uint32_t) + 64 + scan_scratch_bytes(nh);
}

int radix_sort_pairs(uint32_t **keys, uint32_t **vals, uint32_t **keys_alt, uint32_t **vals_alt, size_t n,
                     int nbits, void *d_scratch, hipStream_t s) {
  if (n == 0 || nbits <= 0) return 0;
  const uint32_t nblocks = (uint32_t)((n + kTile - 1) / kTile);
  const size_t nh = (size_t)nblocks * kRadix;
  uint32_t *hist = static_cast<uint32_t *>(d_scratch);
  uint32_t *offs = hist + nh;
  void *scan_scr = reinterpret_cast<void *>(((uintptr_t)(offs + nh + 1) + 63) & ~(uintptr_t)63);
  for (int shift = 0; shift < nbits; shift += kRadixBits) {
    hipLaunchKernelGGL(k_radix_hist, dim3(nblocks), dim3(kThreads), 0, s, *keys, n, shift, hist, nblocks);
    BIVX_TRY(exclusive_scan_u32_u32(hist, offs, nh, scan_scr, s));
    hipLaunchKernelGGL(k_radix_scatter, dim3(nblocks), dim3(kThreads), 0, s, *keys, *vals, *keys_alt, *vals_alt, n,
                       shift, offs, nblocks);
    BIVX_HIP(hipGetLastError());
    uint32_t *t = *keys; *keys = *keys_alt; *keys_alt = t;
    t = *vals; *vals = *vals_alt; *vals_alt = t;
  }
  return 0;
}

int launch_gather_u32(const uint32_t *d_src, const uint32_t *d_idx, uint32_t *d_dst, size_t n, hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_gather_u32, dim3(grid_for(n, kThreads)), dim3(kThreads), 0, s, d_src, d_idx, d_dst, n);
  BIVX_HIP(hipGetLastError());
  return 0;
}

int launch_gather_se(const uint32_t *d_low, const uint32_t *d_high, const uint32_t *d_idx, uint2 *d_se, size_t n,
                     hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_gather_se, dim3(grid_for(n, kThreads)), dim3(kThreads), 0, s, d_low, d_high, d_idx, d_se, n);
  BIVX_HIP(hipGetLastError());
  return 0;
}

int launch_build_table(const uint2 *d_se, const SegDesc *d_seg, uint32_t nseg, uint32_t *d_table,
                       uint64_t nentries, hipStream_t s) {
  if (nentries == 0) return 0;
  hipLaunchKernelGGL(k_build_table, dim3(grid_for(nentries, kThreads)), dim3(kThreads), 0, s, d_se, d_seg, nseg,
                     d_table, nentries);
  BIVX_HIP(hipGetLastError());
  return 0;
}

int launch_pack_records(const uint2 *d_se, const uint32_t *d_id, const SegDesc *d_seg, uint32_t nseg, uint2 *d_rec,
                        size_t n, hipStream_t s) {
  if (n == 0 || nseg == 0) return 0;
  hipLaunchKernelGGL(k_pack_records, dim3(grid_for(n, kThreads)), dim3(kThreads), 0, s, d_se, d_id, d_seg, nseg, d_rec,
                     n);
  BIVX_HIP(hipGetLastError());
  return 0;
}

int launch_gather_intervals(const uint32_t *d_chrom, const uint32_t *d_low, const uint32_t *d_high,
                            const uint32_t *d_ids, size_t n, size_t n_intervals, uint32_t *d_c, uint32_t *d_l,
                            uint32_t *d_h, hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_gather_intervals, dim3(grid_for(n, kThreads)), dim3(kThreads), 0, s, d_chrom, d_low, d_high,
                     d_ids, n, n_intervals, d_c, d_l, d_h);
  BIVX_HIP(hipGetLastError());
  return 0;
}

}  // namespace bivx
