// build.hip — build-side kernels of the interval index (gfx950, wave64).
//
// Replaces the net effect of the reference's per-record tree insertion (IntervalTree::insert_node_impl
// interval_tree.hpp:230-260, RbTree::fix_insert rb_tree.hpp:304-344, rotations interval_tree.hpp:206-228)
// with a batch build: per-(chromosome, length-bin) statistics -> host picks length classes -> stable LSD
// radix sort of (segment, low) carrying the append-order id -> gather (low, high) -> bucket directory.
// All integer/byte work; bound by HBM streaming and scatter, never by MFMA.
#include "common.h"
#include "wave_device.h"

namespace bivx {
namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ uint32_t len_bin(uint32_t low, uint32_t high, uint32_t &len) {
  len = high >= low ? high - low : 0u;  // low > high entries can only be hit inside [high, low]: length 0
  return len == 0 ? 0u : 32u - (uint32_t)__clz((int)len);
}

// ---- per (chromosome, length bin) statistics ---------------------------------------------------------
// One pass over the appended columns. A wavefront takes 64 intervals at a time and aggregates them by (partition,
// length bin) BEFORE anything touches a shared table: for every distinct key among its 64 lanes (a scalar loop over
// ballots) the count is a popcount and min / max are DPP reductions, and one lane adds the result to the table. The
// first form did five LDS atomics per interval, all of a workgroup's lanes on the ~10 words of one chromosome's common
// length bins: the LDS serialises same-address atomics, so 10 M intervals took 169 us — 0.7 TB/s for a 120 MB read.

constexpr uint32_t kStatsLdsEntries = 3300;  // (partition, bin) pairs privatised in LDS (100 partitions: 66 KB)
constexpr int kStatsThreads = 512;

// the partition ("virtual chromosome") of interval i: chrom * ntypes + svtype (include/bivx.h, bivx_append_typed)
__device__ __forceinline__ uint32_t part_of(const uint32_t *__restrict__ chrom, const uint8_t *__restrict__ type,
                                            uint32_t ntypes, size_t i) {
  const uint32_t c = chrom ? chrom[i] : 0u;
  return type ? c * ntypes + type[i] : c;
}

template <bool USE_LDS>
__global__ __launch_bounds__(kStatsThreads) void k_bin_stats(const uint32_t *__restrict__ chrom,
                                                             const uint8_t *__restrict__ type, uint32_t ntypes,
                                                             const uint32_t *__restrict__ low,
                                                             const uint32_t *__restrict__ high, size_t n,
                                                             uint32_t nent, BinStats *__restrict__ stats) {
  __shared__ BinStats lds[USE_LDS ? kStatsLdsEntries : 1];
  if (USE_LDS) {
    for (uint32_t e = threadIdx.x; e < nent; e += kStatsThreads) lds[e] = BinStats{0u, 0xFFFFFFFFu, 0u, 0u, 0u};
    __syncthreads();
  }
  BinStats *tab = USE_LDS ? lds : stats;
  const uint32_t lane = threadIdx.x & (kWave - 1);
  // (wavefront-uniform trip count: the reductions below need all 64 lanes)
  for (size_t i0 = ((size_t)blockIdx.x * kStatsThreads + (threadIdx.x & ~(uint32_t)(kWave - 1))); i0 < n;
       i0 += (size_t)gridDim.x * kStatsThreads) {
    const size_t i = i0 + lane;
    const bool valid = i < n;
    uint32_t lo = 0, hi = 0, key = 0xFFFFFFFFu, len = 0;
    if (valid) {
      lo = low[i];
      hi = high[i];
      key = part_of(chrom, type, ntypes, i) * kLenBins + len_bin(lo, hi, len);
    }
    uint64_t todo = __ballot(valid);
    while (todo) {
      const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, __ffsll((long long)todo) - 1);
      const bool in = key == k;
      const uint64_t m = __ballot(in);
      const uint32_t mn = wave_min(in ? lo : 0xFFFFFFFFu), mx = wave_max(in ? lo : 0u), ml = wave_max(in ? len : 0u);
      const uint32_t ninv = (uint32_t)__popcll(__ballot(in && lo > hi));
      if (lane == 0) {
        BinStats *e = tab + k;
        atomicAdd(&e->count, (uint32_t)__popcll(m));
        atomicMin(&e->min_low, mn);
        atomicMax(&e->max_low, mx);
        atomicMax(&e->max_len, ml);
        if (ninv) atomicAdd(&e->n_inverted, ninv);
      }
      todo &= ~m;
    }
  }
  if (USE_LDS) {
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < nent; e += kStatsThreads) {
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

// workgroup maximum of one value per thread -> ONE atomic per workgroup (every wavefront of a 2 048-workgroup grid
// adding to one address, as the first form did, is 8 192 same-address device atomics of ~10 ns each: 100 us to reduce
// 40 MB)
__device__ __forceinline__ void block_max_to(uint32_t m, uint32_t *out) {
  __shared__ uint32_t s_m[kThreads / kWave];
  m = wave_max(m);
  if ((threadIdx.x & (kWave - 1)) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t r = 0;
    for (int w = 0; w < kThreads / kWave; ++w) r = max(r, s_m[w]);
    if (r) atomicMax(out, r);
  }
}

// out[0] = max chromosome id, out[1] = max svtype (type may be nullptr), in one pass
__global__ __launch_bounds__(kThreads) void k_max_chrom_type(const uint32_t *__restrict__ chrom,
                                                             const uint8_t *__restrict__ type, size_t n,
                                                             uint32_t *__restrict__ out) {
  uint32_t mc = 0, mt = 0;
  const size_t n4 = n / 4;
  const uint4 *c4 = reinterpret_cast<const uint4 *>(chrom);
  const uint32_t *t4 = reinterpret_cast<const uint32_t *>(type);
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (size_t)gridDim.x * kThreads) {
    if (chrom) {
      const uint4 v = c4[i];
      mc = max(max(mc, v.x), max(max(v.y, v.z), v.w));
    }
    if (type) {
      const uint32_t t = t4[i];
      mt = max(max(mt, t & 0xFFu), max(max((t >> 8) & 0xFFu, (t >> 16) & 0xFFu), t >> 24));
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = n4 * 4 + threadIdx.x;
    if (chrom) mc = max(mc, chrom[i]);
    if (type) mt = max(mt, (uint32_t)type[i]);
  }
  // (both reductions by every thread: block_max_to holds a barrier)
  block_max_to(mc, out);
  __syncthreads();
  block_max_to(mt, out + 1);
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
  block_max_to(m, out);
}

__global__ __launch_bounds__(kThreads) void k_gather_u8(const uint8_t *__restrict__ src,
                                                        const uint32_t *__restrict__ ids, size_t n, size_t n_src,
                                                        uint8_t *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i < n) out[i] = ids[i] < n_src ? (src ? src[ids[i]] : (uint8_t)0) : (uint8_t)0xFF;
}

// ---- sort keys -----------------------------------------------------------------------------------------
// The order the index wants is (segment, low, id). ONE 32-bit key gives it whenever the segments' coordinate spans add
// up to less than 2^32 (24 human chromosomes: 3.09 G): segment s owns the key range [keybase[s], keybase[s] + span_s]
// and an interval's key is keybase[seg] + (low - base[seg]) — a stable sort by that key is the whole job, and `low`
// comes back out of the sorted key (no gather). Otherwise (kKeyLow / kKeySegOfId) two stable sorts: by low, then by
// segment of the ids as they lie after the first.
enum : int { kKeyDense = 0, kKeyLow = 1, kKeySegOfId = 2 };

template <int MODE>
__global__ __launch_bounds__(kThreads) void k_make_keys(const uint32_t *__restrict__ chrom,
                                                        const uint8_t *__restrict__ type, uint32_t ntypes,
                                                        const uint32_t *__restrict__ low,
                                                        const uint32_t *__restrict__ high, size_t n,
                                                        const uint32_t *__restrict__ bin2seg,
                                                        const uint2 *__restrict__ segkey,  // (keybase, base) per segment
                                                        const uint32_t *__restrict__ ids,   // kKeySegOfId: sorted ids
                                                        uint32_t *__restrict__ keys) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  if (MODE == kKeyLow) {
    keys[i] = low[i];
    return;
  }
  const size_t j = MODE == kKeySegOfId ? ids[i] : i;
  uint32_t len;
  const uint32_t lo = low[j];
  const uint32_t b = len_bin(lo, high[j], len);
  const uint32_t seg = bin2seg[(size_t)part_of(chrom, type, ntypes, j) * kLenBins + b];
  if (MODE == kKeySegOfId) {
    keys[i] = seg;
  } else {
    const uint2 k = segkey[seg];
    keys[i] = k.x + (lo - k.y);
  }
}

// ---- stable LSD radix sort, 8 bits per pass --------------------------------------------------------------
//
// A workgroup (512 threads) owns a tile of 8 192 consecutive keys; wave w owns the w-th eighth of the tile and lane l
// of round r the key at eighth + r * 64 + l, so "position in tile" order is (wave, round, lane). Ranks inside a round
// come from a ballot match on the 8 digit bits, which needs no LDS atomics and is stable. The tile is then REORDERED
// by digit in LDS and leaves in that order: with 256 digits and 8 192 keys a digit's keys of one tile are ~32
// neighbours in the output, i.e. whole 128-byte lines per store instruction. (The first form scattered straight from
// registers, 1 024 keys per tile: 4-key fragments, every line of the output written in sixteen pieces — 121 us per
// pass for 160 MB.)

constexpr int kRadixBits = 8;
constexpr int kRadix = 1 << kRadixBits;
constexpr int kSortThreads = 512;
constexpr int kSortWaves = kSortThreads / kWave;       // 8
constexpr int kSortRounds = 16;                        // keys per thread
constexpr int kTile = kSortThreads * kSortRounds;      // 8 192 keys
constexpr int kWaveKeys = kTile / kSortWaves;          // 1 024

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
__global__ __launch_bounds__(kSortThreads) void k_radix_hist(const uint32_t *__restrict__ keys, size_t n, int shift,
                                                             uint32_t *__restrict__ hist, uint32_t nblocks) {
  __shared__ uint32_t cnt[kSortWaves][kRadix];  // one table per wavefront: waves do not contend with each other
  for (int w = 0; w < kSortWaves; ++w)
    if (threadIdx.x < kRadix) cnt[w][threadIdx.x] = 0;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * kTile;
  uint32_t *mine = cnt[threadIdx.x >> 6];
#pragma unroll 4
  for (int r = 0; r < kSortRounds; ++r) {
    const size_t i = base + (size_t)r * kSortThreads + threadIdx.x;
    if (i < n) atomicAdd(&mine[(keys[i] >> shift) & (kRadix - 1)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < kRadix) {
    uint32_t c = 0;
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) c += cnt[w][threadIdx.x];
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = c;
  }
}

// IOTA: the values are the keys' own indices (first pass of a sort whose values are the ids 0 .. n-1): nothing is read.
template <bool IOTA>
__global__ __launch_bounds__(kSortThreads) void k_radix_scatter(const uint32_t *__restrict__ keys_in,
                                                                const uint32_t *__restrict__ vals_in,
                                                                uint32_t *__restrict__ keys_out,
                                                                uint32_t *__restrict__ vals_out, size_t n, int shift,
                                                                const uint32_t *__restrict__ offs, uint32_t nblocks) {
  __shared__ uint32_t s_key[kTile];
  __shared__ uint32_t s_val[kTile];
  __shared__ uint32_t wcnt[kSortWaves][kRadix];  // per-wave digit counts, then the waves' first positions per digit
  __shared__ uint32_t s_goff[kRadix];            // where tile position j of digit d goes: s_goff[d] + j
  __shared__ uint32_t s_wsum[kSortWaves];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  for (int w = 0; w < kSortWaves; ++w)
    if (threadIdx.x < kRadix) wcnt[w][threadIdx.x] = 0;
  __syncthreads();

  const size_t tbase = (size_t)blockIdx.x * kTile;
  const size_t wbase = tbase + (size_t)wave * kWaveKeys;
  uint32_t key[kSortRounds], val[kSortRounds];
  uint32_t rank[kSortRounds / 2];  // two 16-bit ranks per register: rank of the key among its wave's keys of that digit
#pragma unroll
  for (int r = 0; r < kSortRounds; ++r) {
    const size_t i = wbase + (size_t)r * kWave + lane;
    key[r] = i < n ? keys_in[i] : 0u;
    val[r] = IOTA ? (uint32_t)i : (i < n ? vals_in[i] : 0u);
  }
  volatile uint32_t *mycnt = wcnt[wave];  // (other lanes of the wavefront write what this lane reads a round later)
#pragma unroll
  for (int r = 0; r < kSortRounds; ++r) {
    const size_t i = wbase + (size_t)r * kWave + lane;
    const bool valid = i < n;
    const uint32_t dg = (key[r] >> shift) & (kRadix - 1);
    const uint64_t m = match_digit(dg, valid);
    const uint32_t below = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    const uint32_t before = valid ? mycnt[dg] : 0u;  // (a wavefront's LDS operations execute in order)
    __builtin_amdgcn_wave_barrier();
    if (valid && below == 0) mycnt[dg] = before + (uint32_t)__popcll(m);  // the group's first lane
    __builtin_amdgcn_wave_barrier();
    const uint32_t rk = before + below;
    if (r & 1) rank[r >> 1] |= rk << 16;
    else rank[r >> 1] = rk;
  }
  __syncthreads();
  // thread t = digit t: the waves' counts become the waves' first positions inside the digit's run; the digits' runs
  // are laid out one after the other in the tile (an exclusive scan over the 256 digit totals)
  uint32_t total = 0;
  if (threadIdx.x < kRadix) {
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) {
      const uint32_t c = wcnt[w][threadIdx.x];
      wcnt[w][threadIdx.x] = total;
      total += c;
    }
  }
  const uint32_t incl = wave_scan_incl(total);
  if (lane == kWave - 1) s_wsum[wave] = incl;
  __syncthreads();
  if (threadIdx.x < kRadix) {
    uint32_t excl = incl - total;
    for (int w = 0; w < wave; ++w) excl += s_wsum[w];
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) wcnt[w][threadIdx.x] += excl;
    s_goff[threadIdx.x] = offs[(size_t)threadIdx.x * nblocks + blockIdx.x] - excl;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kSortRounds; ++r) {
    const size_t i = wbase + (size_t)r * kWave + lane;
    if (i < n) {
      const uint32_t dg = (key[r] >> shift) & (kRadix - 1);
      const uint32_t pos = mycnt[dg] + ((rank[r >> 1] >> ((r & 1) * 16)) & 0xFFFFu);
      s_key[pos] = key[r];
      s_val[pos] = val[r];
    }
  }
  __syncthreads();
  const uint32_t count = n - tbase < (size_t)kTile ? (uint32_t)(n - tbase) : (uint32_t)kTile;
#pragma unroll 4
  for (uint32_t j = threadIdx.x; j < count; j += kSortThreads) {
    const uint32_t k = s_key[j];
    const uint32_t out = s_goff[(k >> shift) & (kRadix - 1)] + j;
    keys_out[out] = k;
    vals_out[out] = s_val[j];
  }
}

// ---- gathers ---------------------------------------------------------------------------------------------

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

// ---- the sorted arrays -------------------------------------------------------------------------------------
// se[i] = (low, high) of sorted slot i, rec[i] = its packed record beside its id (kSegPacked segments; (0, id) elsewhere),
// from the sorted keys and ids in one pass. DENSE: low comes back out of the key (keybase / base of the slot's segment);
// `high` is the one gather — by id, i.e. in append order, and ids of neighbouring slots are scattered over their
// chromosome's part of the column. The launch is XCD-aware for it: workgroup b works on chunk (b % 8) * nchunks / 8 + b / 8,
// so each of the eight L2s (workgroups are dealt round-robin over the XCDs) walks ONE contiguous eighth of the slots and
// with it one chromosome's 3 MB of `high` at a time, instead of all eight walking all of it.
template <bool DENSE>
__global__ __launch_bounds__(kThreads) void k_finalize(const uint32_t *__restrict__ keys,
                                                       const uint32_t *__restrict__ ids,
                                                       const uint32_t *__restrict__ low,
                                                       const uint32_t *__restrict__ high,
                                                       const SegDesc *__restrict__ seg,
                                                       const uint2 *__restrict__ segkey, uint32_t nseg,
                                                       uint2 *__restrict__ se, uint2 *__restrict__ rec, size_t n,
                                                       uint32_t nchunks) {
  const uint32_t per = (nchunks + 7u) / 8u;
  const uint32_t chunk = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
  if ((blockIdx.x >> 3) >= per || chunk >= nchunks) return;
  const size_t i = (size_t)chunk * kThreads + threadIdx.x;
  if (i >= n) return;
  uint32_t lo = 0, hi = nseg;  // last segment with begin <= i
  while (hi - lo > 1) {
    const uint32_t m = (lo + hi) >> 1;
    if ((size_t)seg[m].begin <= i) lo = m; else hi = m;
  }
  const uint32_t shift = seg[lo].shift;
  const uint32_t id = ids[i];
  uint32_t l;
  if (DENSE) {
    const uint2 k = segkey[lo];
    l = k.y + (keys[i] - k.x);
  } else {
    l = low[id];
  }
  const uint32_t h = high[id];
  se[i] = make_uint2(l, h);
  rec[i] = make_uint2((shift & kSegPacked) ? ((l & 0xFFFFu) | ((h - l) << 16)) : 0u, id);
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
  hipLaunchKernelGGL(k_max_cell, dim3(grid_for(nentries, kThreads * 8, 512)), dim3(kThreads), 0, s, d_table, nentries, d_out);
  BIVX_HIP(hipGetLastError());
  return 0;
}

int launch_max_chrom_type(const uint32_t *d_chrom, const uint8_t *d_type, size_t n, uint32_t *d_out2, hipStream_t s) {
  BIVX_HIP(hipMemsetAsync(d_out2, 0, 2 * sizeof(uint32_t), s));
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_max_chrom_type, dim3(grid_for(n, kThreads * 16, 512)), dim3(kThreads), 0, s, d_chrom, d_type, n, d_out2);
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

int launch_bin_stats(const uint32_t *d_chrom, const uint8_t *d_type, uint32_t ntypes, const uint32_t *d_low,
                     const uint32_t *d_high, size_t n, uint32_t nparts, BinStats *d_stats, hipStream_t s) {
  const uint32_t nent = nparts * kLenBins;
  hipLaunchKernelGGL(k_init_stats, dim3(grid_for(nent, kThreads)), dim3(kThreads), 0, s, d_stats, nent);
  if (n) {
    const unsigned nb = grid_for(n, kStatsThreads * 8, 1024);
    if (nent <= kStatsLdsEntries)
      hipLaunchKernelGGL(k_bin_stats<true>, dim3(nb), dim3(kStatsThreads), 0, s, d_chrom, d_type, ntypes, d_low, d_high, n,
                         nent, d_stats);
    else
      hipLaunchKernelGGL(k_bin_stats<false>, dim3(nb), dim3(kStatsThreads), 0, s, d_chrom, d_type, ntypes, d_low, d_high, n,
                         nent, d_stats);
  }
  BIVX_HIP(hipGetLastError());
  return 0;
}

int launch_make_keys(int mode, const uint32_t *d_chrom, const uint8_t *d_type, uint32_t ntypes, const uint32_t *d_low,
                     const uint32_t *d_high, size_t n, const uint32_t *d_bin2seg, const uint2 *d_segkey,
                     const uint32_t *d_ids, uint32_t *d_keys, hipStream_t s) {
  if (n == 0) return 0;
  const dim3 grid(grid_for(n, kThreads)), block(kThreads);
  if (mode == kKeyDense)
    hipLaunchKernelGGL(k_make_keys<kKeyDense>, grid, block, 0, s, d_chrom, d_type, ntypes, d_low, d_high, n, d_bin2seg,
                       d_segkey, d_ids, d_keys);
  else if (mode == kKeyLow)
    hipLaunchKernelGGL(k_make_keys<kKeyLow>, grid, block, 0, s, d_chrom, d_type, ntypes, d_low, d_high, n, d_bin2seg,
                       d_segkey, d_ids, d_keys);
  else
    hipLaunchKernelGGL(k_make_keys<kKeySegOfId>, grid, block, 0, s, d_chrom, d_type, ntypes, d_low, d_high, n, d_bin2seg,
                       d_segkey, d_ids, d_keys);
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
                     int nbits, void *d_scratch, bool vals_are_iota, hipStream_t s) {
  if (n == 0) return 0;
  const uint32_t nblocks = (uint32_t)((n + kTile - 1) / kTile);
  const size_t nh = (size_t)nblocks * kRadix;
  uint32_t *hist = static_cast<uint32_t *>(d_scratch);
  uint32_t *offs = hist + nh;
  void *scan_scr = reinterpret_cast<void *>(((uintptr_t)(offs + nh + 1) + 63) & ~(uintptr_t)63);
  bool iota = vals_are_iota;
  // (at least one pass, so that the values exist in memory when they were only implied)
  for (int shift = 0; shift < nbits || iota; shift += kRadixBits) {
    hipLaunchKernelGGL(k_radix_hist, dim3(nblocks), dim3(kSortThreads), 0, s, *keys, n, shift, hist, nblocks);
    BIVX_TRY(exclusive_scan_u32_u32(hist, offs, nh, scan_scr, s));
    if (iota)
      hipLaunchKernelGGL(k_radix_scatter<true>, dim3(nblocks), dim3(kSortThreads), 0, s, *keys, *vals, *keys_alt,
                         *vals_alt, n, shift, offs, nblocks);
    else
      hipLaunchKernelGGL(k_radix_scatter<false>, dim3(nblocks), dim3(kSortThreads), 0, s, *keys, *vals, *keys_alt,
                         *vals_alt, n, shift, offs, nblocks);
    BIVX_HIP(hipGetLastError());
    iota = false;
    uint32_t *t = *keys; *keys = *keys_alt; *keys_alt = t;
    t = *vals; *vals = *vals_alt; *vals_alt = t;
  }
  return 0;
}

int launch_finalize(const uint32_t *d_keys, const uint32_t *d_ids, const uint32_t *d_low, const uint32_t *d_high,
                    const SegDesc *d_seg, const uint2 *d_segkey, uint32_t nseg, uint2 *d_se, uint2 *d_rec, size_t n,
                    hipStream_t s) {
  if (n == 0 || nseg == 0) return 0;
  const uint32_t nchunks = grid_for(n, kThreads);
  const unsigned grid = ((nchunks + 7u) / 8u) * 8u;
  if (d_keys)
    hipLaunchKernelGGL(k_finalize<true>, dim3(grid), dim3(kThreads), 0, s, d_keys, d_ids, d_low, d_high, d_seg, d_segkey, nseg,
                       d_se, d_rec, n, nchunks);
  else
    hipLaunchKernelGGL(k_finalize<false>, dim3(grid), dim3(kThreads), 0, s, d_keys, d_ids, d_low, d_high, d_seg, d_segkey,
                       nseg, d_se, d_rec, n, nchunks);
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
