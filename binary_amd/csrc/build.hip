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
// One pass over the appended columns. Per interval ONE LDS atomic (its key's count). The three extrema hardly ever
// change once a table entry has seen a few hundred intervals: every lane READS its entry (same-address LDS reads
// broadcast, they do not serialise) and only a wavefront in which some lane would improve an entry reduces those lanes'
// values (DPP) and lets one lane issue the atomics. Steps measured at 10 M intervals (a 120 MB read): five LDS atomics
// per interval, one chunk of 64 per wavefront and trip: 169 us; everything reduced per distinct key in every wavefront:
// 149; counts by a scalar loop over the wavefront's keys + the conditional extrema: 152, with four chunks in flight 103;
// one LDS atomic per interval: 72. Then the cost was per WORKGROUP, not per interval — every workgroup ends by adding its
// table to the global one, ~360 device atomics on the same 360 addresses whatever the grid: 2 048 workgroups of 512
// threads 77 us, 256 of 1 024 (one per CU, eight chunks in flight per wavefront) 46 us = 2.6 TB/s; 50 M intervals 178 us =
// 3.4 TB/s. (The BIVX_STATS_* macros are tools/ab_build.sh's knobs.)

constexpr uint32_t kStatsLdsEntries = 3300;  // (partition, bin) pairs privatised in LDS (100 partitions: 66 KB)
#ifndef BIVX_STATS_PARTS
#define BIVX_STATS_PARTS 64
#endif
constexpr uint32_t kStatsAutoParts = BIVX_STATS_PARTS;     // chromosome ids the one-pass form (no svtypes) has a table for
#ifndef BIVX_STATS_THREADS
#define BIVX_STATS_THREADS 1024
#endif
constexpr int kStatsThreads = BIVX_STATS_THREADS;
#ifndef BIVX_STATS_UNROLL
#define BIVX_STATS_UNROLL 8
#endif
#ifndef BIVX_STATS_TRIPS
#define BIVX_STATS_TRIPS 1
#endif
#ifndef BIVX_STATS_PEEL
#define BIVX_STATS_PEEL 0   // keys of a chunk counted by one ballot each before the rest count themselves (0: none — see below)
#endif
#ifndef BIVX_STATS_CAP
#define BIVX_STATS_CAP 256
#endif
constexpr int kStatsUnroll = BIVX_STATS_UNROLL;            // chunks of 64 intervals a wavefront has in flight

// the partition ("virtual chromosome") of interval i: chrom * ntypes + svtype (include/bivx.h, bivx_append_typed)
__device__ __forceinline__ uint32_t part_of(const uint32_t *__restrict__ chrom, const uint8_t *__restrict__ type,
                                            uint32_t ntypes, size_t i) {
  const uint32_t c = chrom ? chrom[i] : 0u;
  return type ? c * ntypes + type[i] : c;
}

__device__ __forceinline__ uint32_t peek(const uint32_t *p) {  // (a plain read the compiler may not keep in a register)
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// AUTO (no svtypes; USE_LDS): the number of chromosomes is not known yet. The table has nent / kLenBins rows: the host reads
// the largest chromosome id off the rows that came back non-empty, unless the kernel met an id beyond the table — then
// scal[0] holds the largest such id and the host repeats the pass with a table of the right size. One pass over the columns
// and one read-back instead of two of each.
// COPY (bivx_append*_dev): the columns are on their way INTO the index — the pass that copies them takes the statistics
// from the same read (out_* = the index's columns at the append position; chrom may be null: chromosome 0), and the
// table is the index's own, which every append adds to: the build then finds its statistics waiting.
template <bool USE_LDS, bool AUTO = false, bool COPY = false>
__global__ __launch_bounds__(kStatsThreads) void k_bin_stats(const uint32_t *__restrict__ chrom,
                                                             const uint8_t *__restrict__ type, uint32_t ntypes,
                                                             const uint32_t *__restrict__ low,
                                                             const uint32_t *__restrict__ high, size_t n,
                                                             uint32_t nent, BinStats *__restrict__ stats,
                                                             uint32_t *__restrict__ scal,
                                                             uint32_t *__restrict__ out_chrom = nullptr,
                                                             uint32_t *__restrict__ out_low = nullptr,
                                                             uint32_t *__restrict__ out_high = nullptr) {
  extern __shared__ BinStats lds[];  // nent entries (USE_LDS): a small table leaves room for more workgroups per CU
  if (USE_LDS) {
    for (uint32_t e = threadIdx.x; e < nent; e += kStatsThreads) lds[e] = BinStats{0u, 0xFFFFFFFFu, 0u, 0u, 0u};
    __syncthreads();
  }
  BinStats *tab = USE_LDS ? lds : stats;
  const uint32_t lane = threadIdx.x & (kWave - 1);
  uint32_t beyond = 0;  // AUTO: largest chromosome id this thread met that the table has no row for
  // A wavefront takes kStatsUnroll x 64 consecutive intervals per trip and has all their loads in flight before it looks
  // at the first (with one chunk per trip the kernel waited for memory: 4 wavefronts per SIMD x 768 bytes each).
  // (wavefront-uniform trip count: the ballots and reductions below need all 64 lanes)
  const size_t stride = (size_t)gridDim.x * kStatsThreads * kStatsUnroll;
  for (size_t i0 = ((size_t)blockIdx.x * kStatsThreads + (threadIdx.x & ~(uint32_t)(kWave - 1))) * kStatsUnroll; i0 < n;
       i0 += stride) {
    uint32_t lo_[kStatsUnroll], hi_[kStatsUnroll], part_[kStatsUnroll];
#pragma unroll
    for (int u = 0; u < kStatsUnroll; ++u) {
      const size_t i = i0 + (size_t)u * kWave + lane;
      lo_[u] = hi_[u] = part_[u] = 0;
      if (i < n) {
        lo_[u] = low[i];
        hi_[u] = high[i];
        part_[u] = part_of(chrom, type, ntypes, i);
      }
    }
    if (COPY) {
#pragma unroll
      for (int u = 0; u < kStatsUnroll; ++u) {
        const size_t i = i0 + (size_t)u * kWave + lane;
        if (i < n) {
          out_low[i] = lo_[u];
          out_high[i] = hi_[u];
          out_chrom[i] = part_[u];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kStatsUnroll; ++u) {
      const size_t i = i0 + (size_t)u * kWave + lane;
      bool valid = i < n;
      const uint32_t lo = lo_[u], hi = hi_[u];
      uint32_t key = 0xFFFFFFFFu, len = 0;
      if (AUTO && valid && part_[u] >= nent / kLenBins) {  // (the host repeats the statistics with a table of the right size)
        beyond = max(beyond, part_[u]);
        valid = false;
      }
      if (valid) key = part_[u] * kLenBins + len_bin(lo, hi, len);
      // counts: one LDS atomic per interval (lanes of one key serialise on its word, ~30 cycles for the commonest
      // length bin — a scalar loop over the wavefront's distinct keys cost more instructions than that)
      // counts: one LDS atomic per interval (lanes of one key serialise on its word). Counting the chunk's commonest keys
      // by a ballot each first (BIVX_STATS_PEEL rounds, one lane adds the number) is slower: 66.7 / 72.0 / 76.1 / 80.6 us
      // for 0 / 2 / 3 / 4 rounds at 10 M intervals (the copying form) — the serialised atomics are not what the pass waits for
      {
        uint64_t left = __ballot(valid);
#pragma unroll
        for (int r = 0; r < BIVX_STATS_PEEL; ++r) {
          if (left == 0) break;
          const int src = __ffsll((long long)left) - 1;
          const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, src);
          const uint64_t m = __ballot(valid && key == k) & left;
          if ((int)lane == src) atomicAdd(&tab[k].count, (uint32_t)__popcll(m));
          left &= ~m;
        }
        if ((left >> lane) & 1ull) atomicAdd(&tab[key].count, 1u);
      }
      if (valid && lo > hi) atomicAdd(&tab[key].n_inverted, 1u);
      // extrema: only lanes that would improve their entry take part
      bool better = false;
      if (valid) {
        const BinStats *e = tab + key;
        better = lo < peek(&e->min_low) || lo > peek(&e->max_low) || len > peek(&e->max_len);
      }
      uint64_t todo = __ballot(better);
      while (todo) {
        const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, __ffsll((long long)todo) - 1);
        const bool in = better && key == k;
        const uint32_t mn = wave_min(in ? lo : 0xFFFFFFFFu), mx = wave_max(in ? lo : 0u), ml = wave_max(in ? len : 0u);
        if (lane == 0) {
          atomicMin(&tab[k].min_low, mn);
          atomicMax(&tab[k].max_low, mx);
          atomicMax(&tab[k].max_len, ml);
        }
        todo &= ~__ballot(in);
      }
    }
  }
  if (AUTO && __ballot(beyond != 0)) {  // (no wavefront comes here when every id has a row: the usual case costs nothing)
    const uint32_t m = wave_max(beyond);
    if (lane == 0 && m > peek(&scal[0])) atomicMax(&scal[0], m);
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

__global__ __launch_bounds__(kThreads) void k_init_stats(BinStats *stats, uint32_t nent, uint32_t *scal) {
  const uint32_t e = blockIdx.x * kThreads + threadIdx.x;
  if (e < nent) stats[e] = BinStats{0u, 0xFFFFFFFFu, 0u, 0u, 0u};
  if (scal && e < 4) scal[e] = 0;  // (the build's maxima start at zero)
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
    // (a word that already holds as much is left alone: the atomics of a grid on one address queue up, ~10 ns each)
    if (r && r > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, r);
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

// kKeyLow also leaves every interval's segment in seg_of[] (append order), so that the second stage — kKeySegOfId:
// keys[i] = seg_of[ids[i]] — is one gather of one word (recomputing the segment from the gathered columns was four).
// The key kernel of the first sort (kKeyDense / kKeyLow) walks the sort's own tiles — kTile keys per workgroup of
// kSortThreads — and leaves the first pass's digit histogram beside the keys (hist0, digit 0 = the keys' low 8 bits): the
// first pass needs no histogram kernel of its own.
template <int MODE>
__global__ __launch_bounds__(512) void k_make_keys(const uint32_t *__restrict__ chrom,
                                                   const uint8_t *__restrict__ type, uint32_t ntypes,
                                                   const uint32_t *__restrict__ low,
                                                   const uint32_t *__restrict__ high, size_t n,
                                                   const uint32_t *__restrict__ bin2seg,
                                                   const uint2 *__restrict__ segkey,  // (keybase, base) per segment
                                                   const uint32_t *__restrict__ ids,   // kKeySegOfId: sorted ids
                                                   uint32_t *__restrict__ seg_of,      // kKeyLow: out; kKeySegOfId: in
                                                   uint32_t *__restrict__ keys, uint32_t *__restrict__ hist0,
                                                   uint32_t nblocks, int first_shift) {
  if (MODE == kKeySegOfId) {
    const size_t i = (size_t)blockIdx.x * 512 + threadIdx.x;
    if (i < n) keys[i] = seg_of[ids[i]];
    return;
  }
  __shared__ uint32_t cnt[8][256];  // one table per wavefront
  for (int w = 0; w < 8; ++w)
    if (threadIdx.x < 256) cnt[w][threadIdx.x] = 0;
  __syncthreads();
  uint32_t *mine = cnt[threadIdx.x >> 6];
  const size_t base = (size_t)blockIdx.x * 8192;
#pragma unroll 4
  for (int r = 0; r < 16; ++r) {
    const size_t i = base + (size_t)r * 512 + threadIdx.x;
    if (i < n) {
      uint32_t len;
      const uint32_t lo = low[i];
      const uint32_t b = len_bin(lo, high[i], len);
      const uint32_t seg = bin2seg[(size_t)part_of(chrom, type, ntypes, i) * kLenBins + b];
      uint32_t key = lo;
      if (MODE == kKeyLow) {
        seg_of[i] = seg;
      } else {
        const uint2 k = segkey[seg];
        key = k.x + (lo - k.y);
      }
      keys[i] = key;
      atomicAdd(&mine[(key >> first_shift) & 255u], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    uint32_t c = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) c += cnt[w][threadIdx.x];
    hist0[(size_t)threadIdx.x * nblocks + blockIdx.x] = c;
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

// offs[d][b] = number of keys with digit d in the tiles before b (one workgroup per digit walks its row of the
// histogram); totals[d] = the digit's keys in all tiles. The scatter kernel adds the digits before d itself, so one
// pass is three launches — histogram, rows, scatter — where a device-wide scan of the whole table took three by itself.
__global__ __launch_bounds__(kThreads) void k_radix_rows(const uint32_t *__restrict__ hist, uint32_t *__restrict__ offs,
                                                         uint32_t *__restrict__ totals, uint32_t nblocks) {
  __shared__ uint32_t s_w[kThreads / kWave];
  const uint32_t *row = hist + (size_t)blockIdx.x * nblocks;
  uint32_t *out = offs + (size_t)blockIdx.x * nblocks;
  const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  uint32_t carry = 0;
  for (uint32_t c = 0; c < nblocks; c += kThreads * 4) {
    const uint32_t i = c + threadIdx.x * 4;
    uint32_t v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = i + k < nblocks ? row[i + k] : 0u;
      sum += v[k];
    }
    const uint32_t incl = wave_scan_incl(sum);
    if (lane == kWave - 1) s_w[wave] = incl;
    __syncthreads();
    uint32_t run = carry + incl - sum, all = 0;
#pragma unroll
    for (uint32_t w = 0; w < kThreads / kWave; ++w) {
      const uint32_t t = s_w[w];
      if (w < wave) run += t;
      all += t;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (i + k < nblocks) out[i + k] = run;
      run += v[k];
    }
    carry += all;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// IOTA: the values are the keys' own indices (first pass of a sort whose values are the ids 0 .. n-1): nothing is read.
template <bool IOTA>
__global__ __launch_bounds__(kSortThreads) void k_radix_scatter(const uint32_t *__restrict__ keys_in,
                                                                const uint32_t *__restrict__ vals_in,
                                                                uint32_t *__restrict__ keys_out,
                                                                uint32_t *__restrict__ vals_out, size_t n, int shift,
                                                                const uint32_t *__restrict__ offs,
                                                                const uint32_t *__restrict__ totals, uint32_t nblocks) {
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
  // exclusive scan over the 256 values that threads 0 .. 255 hold (the others pass 0)
  auto excl_scan = [&](uint32_t v) {
    const uint32_t incl = wave_scan_incl(v);
    if (lane == kWave - 1) s_wsum[wave] = incl;
    __syncthreads();
    uint32_t ex = incl - v;
    for (int w = 0; w < wave; ++w) ex += s_wsum[w];
    __syncthreads();
    return ex;
  };
  const uint32_t excl = excl_scan(total);                                         // the digit's first position in the tile
  const uint32_t dbase = excl_scan(threadIdx.x < kRadix ? totals[threadIdx.x] : 0u);  // ... and in the whole output
  if (threadIdx.x < kRadix) {
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) wcnt[w][threadIdx.x] += excl;
    s_goff[threadIdx.x] = dbase + offs[(size_t)threadIdx.x * nblocks + blockIdx.x] - excl;
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

// ---- the sorted arrays and the bucket directory ----------------------------------------------------------
// se[i] = (low, high) of sorted slot i, rec[i] = its packed record beside its id (kSegPacked segments; (0, id) elsewhere),
// from the sorted keys and ids in one pass. DENSE: low comes back out of the key (keybase / base of the slot's segment);
// `high` is the one gather — by id, i.e. in append order, and ids of neighbouring slots are scattered over their
// chromosome's part of the column. The launch is XCD-aware for it: workgroup b works on chunk (b % 8) * nchunks / 8 + b / 8,
// so each of the eight L2s (workgroups are dealt round-robin over the XCDs) walks ONE contiguous eighth of the slots and
// with it one chromosome's 3 MB of `high` at a time, instead of all eight walking all of it.
//
// The same pass writes the bucket directory: table[d.table_off + c] = first slot of the segment whose low >= d.base +
// (c << d.shift), for c in [0, ncell]; entry ncell is the segment's end. Slot i, in cell c_i, is that first slot for every
// cell after its predecessor's up to its own: the entries (c_{i-1}, c_i] receive i (from entry 0 on if i opens its
// segment), and the segment's last slot also gives the entries behind its cell the segment's end. Mostly none, one or
// two entries per slot; a longer stretch of empty cells (a centromere) is written by the slot's whole wavefront, and
// one of more than kCoopMax entries goes on a short list that k_fill_gaps, launched behind this kernel, works off with
// the whole grid. (A search per entry — the first form — is a chain of ~20 dependent probes: 66 us for 3 M entries.)
constexpr uint32_t kInlineFill = 4;         // entries a slot writes by itself
constexpr uint32_t kCoopMax = 1u << 14;     // entries a wavefront writes for one of its slots

struct Gap {
  uint32_t first, count, value, pad;
};

constexpr uint32_t kFinSlots = 4;  // slots per thread, kThreads apart: their loads are all in flight together

template <bool DENSE>
__global__ __launch_bounds__(kThreads) void k_finalize(const uint32_t *__restrict__ keys,
                                                       const uint32_t *__restrict__ ids,
                                                       const uint32_t *__restrict__ low,
                                                       const uint32_t *__restrict__ high,
                                                       const SegDesc *__restrict__ seg,
                                                       const uint2 *__restrict__ segkey, uint32_t nseg,
                                                       uint2 *__restrict__ se, uint2 *__restrict__ rec,
                                                       uint32_t *__restrict__ table, Gap *__restrict__ gaps,
                                                       uint32_t *__restrict__ ngaps, uint32_t gap_cap, size_t n,
                                                       uint32_t nchunks) {
  const uint32_t per = (nchunks + 7u) / 8u;
  const uint32_t chunk = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
  if ((blockIdx.x >> 3) >= per || chunk >= nchunks) return;  // (workgroup-uniform)
  const size_t i0 = (size_t)chunk * (kThreads * kFinSlots) + threadIdx.x;
  if (chunk == 0 && threadIdx.x < 2) {  // the two spare slots behind the arrays (query lanes read pairs of slots)
    se[n + threadIdx.x] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    rec[n + threadIdx.x] = make_uint2(0u, 0u);
  }
  const uint32_t lane = threadIdx.x & (kWave - 1);
  // the segment of the chunk's first slot: the last one with begin <= that slot (the others follow by stepping)
  uint32_t s0 = 0;
  {
    const size_t first = (size_t)chunk * (kThreads * kFinSlots);
    uint32_t lo = 0, hi = nseg;
    while (hi - lo > 1) {
      const uint32_t m = (lo + hi) >> 1;
      if ((size_t)seg[m].begin <= first) lo = m; else hi = m;
    }
    s0 = lo;
  }
  // (the streams are touched once: non-temporal, so that they leave the L2 to the gathered column)
  uint32_t id[kFinSlots], key[kFinSlots], prev[kFinSlots], h[kFinSlots];
#pragma unroll
  for (uint32_t j = 0; j < kFinSlots; ++j) {
    const size_t i = i0 + (size_t)j * kThreads;
    id[j] = key[j] = prev[j] = 0;
    if (i < n) {
      id[j] = __builtin_nontemporal_load(ids + i);
      if (DENSE) {
        key[j] = __builtin_nontemporal_load(keys + i);
        prev[j] = i ? keys[i - 1] : 0u;
      } else {
        prev[j] = i ? ids[i - 1] : 0u;
      }
    }
  }
#pragma unroll
  for (uint32_t j = 0; j < kFinSlots; ++j) {
    const size_t i = i0 + (size_t)j * kThreads;
    h[j] = 0;
    if (i < n) {
      h[j] = high[id[j]];
      if (!DENSE) {
        key[j] = low[id[j]];
        prev[j] = i ? low[prev[j]] : 0u;
      }
    }
  }
  unsigned long long *se64 = reinterpret_cast<unsigned long long *>(se), *rec64 = reinterpret_cast<unsigned long long *>(rec);
#pragma unroll
  for (uint32_t j = 0; j < kFinSlots; ++j) {
    const size_t i = i0 + (size_t)j * kThreads;
    const bool valid = i < n;
    // what this slot writes into the directory: entries [e0, e0 + cnt) = i, and behind a segment's last slot [t0, t0 + tcnt) = end
    uint32_t e0 = 0, cnt = 0, t0 = 0, tcnt = 0, tval = 0;
    if (valid) {
      uint32_t si = s0;
      while (si + 1 < nseg && (size_t)seg[si + 1].begin <= i) ++si;
      const SegDesc d = seg[si];
      uint32_t l, lp;
      if (DENSE) {
        const uint2 k = segkey[si];
        l = k.y + (key[j] - k.x);
        lp = k.y + (prev[j] - k.x);
      } else {
        l = key[j];
        lp = prev[j];
      }
      const uint32_t r = (d.shift & kSegPacked) ? ((l & 0xFFFFu) | ((h[j] - l) << 16)) : 0u;
      __builtin_nontemporal_store((unsigned long long)l | (unsigned long long)h[j] << 32, se64 + i);
      __builtin_nontemporal_store((unsigned long long)r | (unsigned long long)id[j] << 32, rec64 + i);
      // directory
      const uint32_t sh = d.shift & 31u;
      const uint32_t c = (l - d.base) >> sh;
      const uint32_t first = i > d.begin ? ((lp - d.base) >> sh) + 1u : 0u;
      e0 = d.table_off + first;
      cnt = c + 1u - first;  // (0 when the slot shares its predecessor's cell)
      if (i + 1 == d.end) {
        t0 = d.table_off + c + 1u;
        tcnt = d.ncell - c;
        tval = d.end;
      }
    }
    const uint32_t val = (uint32_t)i;
    if (cnt <= kInlineFill)
      for (uint32_t c = 0; c < cnt; ++c) table[e0 + c] = val;
    if (tcnt <= kInlineFill)
      for (uint32_t c = 0; c < tcnt; ++c) table[t0 + c] = tval;
    // longer stretches: the wavefront writes them, lane by lane (every lane of the wavefront is here)
    auto cooperative = [&](bool pending, uint32_t b0, uint32_t bc, uint32_t bv) {
      uint64_t pm = __ballot(pending);
      while (pm) {
        const int src = __ffsll((long long)pm) - 1;
        pm &= pm - 1;
        const uint32_t g0 = (uint32_t)__builtin_amdgcn_readlane((int)b0, src);
        const uint32_t gc = (uint32_t)__builtin_amdgcn_readlane((int)bc, src);
        const uint32_t gv = (uint32_t)__builtin_amdgcn_readlane((int)bv, src);
        if (gc > kCoopMax) {
          if (lane == 0) {
            const uint32_t g = atomicAdd(ngaps, 1u);
            if (g < gap_cap) gaps[g] = Gap{g0, gc, gv, 0u};  // (the capacity covers every stretch there can be)
          }
        } else {
          for (uint32_t c = lane; c < gc; c += kWave) table[g0 + c] = gv;
        }
      }
    };
    cooperative(cnt > kInlineFill, e0, cnt, val);
    cooperative(tcnt > kInlineFill, t0, tcnt, tval);
  }
}

// the listed stretches of directory entries, by the whole grid
__global__ __launch_bounds__(kThreads) void k_fill_gaps(const Gap *__restrict__ gaps, const uint32_t *__restrict__ ngaps,
                                                        uint32_t gap_cap, uint32_t *__restrict__ table) {
  const uint32_t ng = min(*ngaps, gap_cap);
  const uint32_t t = blockIdx.x * kThreads + threadIdx.x, nt = gridDim.x * kThreads;
  for (uint32_t g = 0; g < ng; ++g) {
    const Gap gp = gaps[g];
    for (uint32_t c = t; c < gp.count; c += nt) table[gp.first + c] = gp.value;
  }
}

// largest number of slots any directory cell holds: max over e of table[e + 1] - table[e] (a segment's entries end with
// its last slot + 1, which is where the next segment's begin: the difference across a boundary is 0). Also writes the
// directory's three spare entries (query lanes read entries four at a time).
__global__ __launch_bounds__(kThreads) void k_max_cell(uint32_t *__restrict__ table, size_t nentries,
                                                       uint32_t *__restrict__ out) {
  uint32_t m = 0;
  auto diff = [](uint32_t a, uint32_t b) { return b > a ? b - a : 0u; };
  // four entries per 16-byte load and the one behind them (its line is the next group's), two groups in flight per thread
  // (one entry and its neighbour per trip: 18 us for the 24 MB of config 3's directory)
  const size_t n4 = nentries ? (nentries - 1) / 4 : 0;  // groups g whose entries 4g .. 4g + 4 all exist
  const uint4 *t4 = reinterpret_cast<const uint4 *>(table);
  const size_t stride = (size_t)gridDim.x * kThreads;
  for (size_t g = (size_t)blockIdx.x * kThreads + threadIdx.x; g < n4; g += 2 * stride) {
    const size_t g2 = g + stride;
    const bool two = g2 < n4;
    const uint4 v = t4[g];
    const uint32_t nx = table[4 * g + 4];
    uint4 w = make_uint4(0u, 0u, 0u, 0u);
    uint32_t nw = 0;
    if (two) {
      w = t4[g2];
      nw = table[4 * g2 + 4];
    }
    m = max(max(m, diff(v.x, v.y)), max(max(diff(v.y, v.z), diff(v.z, v.w)), diff(v.w, nx)));
    if (two) m = max(max(m, diff(w.x, w.y)), max(max(diff(w.y, w.z), diff(w.z, w.w)), diff(w.w, nw)));
  }
  if (blockIdx.x == 0)
    for (size_t i = 4 * n4 + threadIdx.x; i + 1 < nentries; i += kThreads) m = max(m, diff(table[i], table[i + 1]));
  if (blockIdx.x == 0 && threadIdx.x < 3) table[nentries + threadIdx.x] = 0xFFFFFFFFu;
  block_max_to(m, out);
}

// The index's own intervals as a batch of queries in slot order (bivx_self_overlaps_dev): (chromosome of the slot's
// segment, low, high) — position-sorted by construction.
__global__ __launch_bounds__(kThreads) void k_self_queries(const uint2 *__restrict__ se, const SegDesc *__restrict__ seg,
                                                           const uint32_t *__restrict__ seg_chrom, uint32_t nseg, size_t n,
                                                           uint32_t *__restrict__ qchrom, uint32_t *__restrict__ qlow,
                                                           uint32_t *__restrict__ qhigh) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  uint32_t lo = 0, hi = nseg;  // last segment with begin <= i
  while (hi - lo > 1) {
    const uint32_t m = (lo + hi) >> 1;
    if ((size_t)seg[m].begin <= i) lo = m; else hi = m;
  }
  const uint2 e = se[i];
  qchrom[i] = seg_chrom[lo];
  qlow[i] = e.x;
  qhigh[i] = e.y;
}

inline unsigned grid_for(size_t n, int per_block, unsigned cap = 0) {
  size_t nb = (n + (size_t)per_block - 1) / (size_t)per_block;
  if (nb < 1) nb = 1;
  if (cap && nb > cap) nb = cap;
  return (unsigned)nb;
}

}  // namespace

int launch_max_chrom_type(const uint32_t *d_chrom, const uint8_t *d_type, size_t n, uint32_t *d_out2, hipStream_t s) {
  if (n == 0) return 0;  // (the caller has zeroed d_out2)
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
  hipLaunchKernelGGL(k_init_stats, dim3(grid_for(nent, kThreads)), dim3(kThreads), 0, s, d_stats, nent, (uint32_t *)nullptr);
  if (n) {
    const unsigned nb = grid_for(n, kStatsThreads * kStatsUnroll * BIVX_STATS_TRIPS, BIVX_STATS_CAP);
    if (nent <= kStatsLdsEntries)
      hipLaunchKernelGGL((k_bin_stats<true, false>), dim3(nb), dim3(kStatsThreads), (size_t)nent * sizeof(BinStats), s, d_chrom,
                         d_type, ntypes, d_low, d_high, n, nent, d_stats, (uint32_t *)nullptr);
    else
      hipLaunchKernelGGL((k_bin_stats<false, false>), dim3(nb), dim3(kStatsThreads), 0, s, d_chrom, d_type, ntypes, d_low, d_high,
                         n, nent, d_stats, (uint32_t *)nullptr);
  }
  BIVX_HIP(hipGetLastError());
  return 0;
}

// No svtypes: statistics of the chromosomes below bin_stats_auto_parts() in one pass without knowing how many there are;
// d_scal[0] != 0: the largest chromosome id beyond the table (the statistics are then incomplete). d_stats: bin_stats_auto_parts() * kLenBins entries; d_scal[0..3] are zeroed here.
uint32_t bin_stats_auto_parts() { return kStatsAutoParts; }
int launch_bin_stats_auto(const uint32_t *d_chrom, const uint32_t *d_low, const uint32_t *d_high, size_t n, BinStats *d_stats,
                          uint32_t *d_scal, hipStream_t s) {
  const uint32_t nent = bin_stats_auto_parts() * kLenBins;
  hipLaunchKernelGGL(k_init_stats, dim3(grid_for(nent, kThreads)), dim3(kThreads), 0, s, d_stats, nent, d_scal);
  if (n) {
    const unsigned nb = grid_for(n, kStatsThreads * kStatsUnroll * BIVX_STATS_TRIPS, BIVX_STATS_CAP);
    hipLaunchKernelGGL((k_bin_stats<true, true>), dim3(nb), dim3(kStatsThreads), (size_t)nent * sizeof(BinStats), s, d_chrom,
                       (const uint8_t *)nullptr, 1u, d_low, d_high, n, nent, d_stats, d_scal);
  }
  BIVX_HIP(hipGetLastError());
  return 0;
}

// The index's own statistics table (auto form: [4 scalars in 256 bytes | bin_stats_auto_parts() x kLenBins entries]),
// emptied at creation and by bivx_clear; every untyped append adds its intervals to it — copying them into the index's
// columns by the same pass when they come from device memory (d_out_* non-null; d_chrom may be null: chromosome 0).
size_t auto_stats_bytes() { return 256 + (size_t)bin_stats_auto_parts() * kLenBins * sizeof(BinStats); }
int launch_init_auto_stats(void *d_block, hipStream_t s) {
  const uint32_t nent = bin_stats_auto_parts() * kLenBins;
  hipLaunchKernelGGL(k_init_stats, dim3(grid_for(nent, kThreads)), dim3(kThreads), 0, s,
                     reinterpret_cast<BinStats *>(static_cast<char *>(d_block) + 256), nent, static_cast<uint32_t *>(d_block));
  BIVX_HIP(hipGetLastError());
  return 0;
}
int launch_append_stats(const uint32_t *d_chrom, const uint32_t *d_low, const uint32_t *d_high, size_t n, uint32_t *d_out_chrom,
                        uint32_t *d_out_low, uint32_t *d_out_high, void *d_block, hipStream_t s) {
  if (n == 0) return 0;
  const uint32_t nent = bin_stats_auto_parts() * kLenBins;
  BinStats *d_stats = reinterpret_cast<BinStats *>(static_cast<char *>(d_block) + 256);
  uint32_t *d_scal = static_cast<uint32_t *>(d_block);
  const unsigned nb = grid_for(n, kStatsThreads * kStatsUnroll * BIVX_STATS_TRIPS, BIVX_STATS_CAP);
  if (d_out_low)
    hipLaunchKernelGGL((k_bin_stats<true, true, true>), dim3(nb), dim3(kStatsThreads), (size_t)nent * sizeof(BinStats), s, d_chrom,
                       (const uint8_t *)nullptr, 1u, d_low, d_high, n, nent, d_stats, d_scal, d_out_chrom, d_out_low, d_out_high);
  else
    hipLaunchKernelGGL((k_bin_stats<true, true, false>), dim3(nb), dim3(kStatsThreads), (size_t)nent * sizeof(BinStats), s, d_chrom,
                       (const uint8_t *)nullptr, 1u, d_low, d_high, n, nent, d_stats, d_scal, (uint32_t *)nullptr,
                       (uint32_t *)nullptr, (uint32_t *)nullptr);
  BIVX_HIP(hipGetLastError());
  return 0;
}

// d_hist0 (modes 0 and 1): the radix scratch — the first pass's histogram is left there (radix_sort_pairs: hist0_ready)
int launch_make_keys(int mode, const uint32_t *d_chrom, const uint8_t *d_type, uint32_t ntypes, const uint32_t *d_low,
                     const uint32_t *d_high, size_t n, const uint32_t *d_bin2seg, const uint2 *d_segkey,
                     const uint32_t *d_ids, uint32_t *d_seg_of, uint32_t *d_keys, void *d_hist0, hipStream_t s,
                     int first_shift) {
  if (n == 0) return 0;
  const uint32_t nblocks = (uint32_t)((n + kTile - 1) / kTile);
  uint32_t *hist0 = static_cast<uint32_t *>(d_hist0);
  if (mode == kKeyDense)
    hipLaunchKernelGGL(k_make_keys<kKeyDense>, dim3(nblocks), dim3(512), 0, s, d_chrom, d_type, ntypes, d_low, d_high, n,
                       d_bin2seg, d_segkey, d_ids, d_seg_of, d_keys, hist0, nblocks, first_shift);
  else if (mode == kKeyLow)
    hipLaunchKernelGGL(k_make_keys<kKeyLow>, dim3(nblocks), dim3(512), 0, s, d_chrom, d_type, ntypes, d_low, d_high, n,
                       d_bin2seg, d_segkey, d_ids, d_seg_of, d_keys, hist0, nblocks, first_shift);
  else
    hipLaunchKernelGGL(k_make_keys<kKeySegOfId>, dim3(grid_for(n, 512)), dim3(512), 0, s, d_chrom, d_type, ntypes, d_low,
                       d_high, n, d_bin2seg, d_segkey, d_ids, d_seg_of, d_keys, hist0, nblocks, first_shift);
  BIVX_HIP(hipGetLastError());
  return 0;
}

size_t radix_scratch_bytes(size_t n) {
  const size_t nblocks = (n + kTile - 1) / kTile;
  const size_t nh = nblocks * kRadix;
  return (nh + nh + kRadix) * sizeof(uint32_t);  // histogram, its row-wise exclusive scan, the digits' totals
}

int radix_sort_pairs(uint32_t **keys, uint32_t **vals, uint32_t **keys_alt, uint32_t **vals_alt, size_t n,
                     int nbits, void *d_scratch, bool vals_are_iota, bool hist0_ready, hipStream_t s, int first_shift) {
  if (n == 0) return 0;
  const uint32_t nblocks = (uint32_t)((n + kTile - 1) / kTile);
  const size_t nh = (size_t)nblocks * kRadix;
  uint32_t *hist = static_cast<uint32_t *>(d_scratch);
  uint32_t *offs = hist + nh;
  uint32_t *totals = offs + nh;
  bool iota = vals_are_iota;
  // (at least one pass, so that the values exist in memory when they were only implied)
  // (first_shift: the key bits below it do not take part — the build orders by directory cell, not by every bit of low)
  for (int shift = first_shift; shift < nbits || iota; shift += kRadixBits) {
    if (!(hist0_ready && shift == first_shift))  // (the key kernel left the first pass's histogram)
      hipLaunchKernelGGL(k_radix_hist, dim3(nblocks), dim3(kSortThreads), 0, s, *keys, n, shift, hist, nblocks);
    hipLaunchKernelGGL(k_radix_rows, dim3(kRadix), dim3(kThreads), 0, s, hist, offs, totals, nblocks);
    if (iota)
      hipLaunchKernelGGL(k_radix_scatter<true>, dim3(nblocks), dim3(kSortThreads), 0, s, *keys, *vals, *keys_alt,
                         *vals_alt, n, shift, offs, totals, nblocks);
    else
      hipLaunchKernelGGL(k_radix_scatter<false>, dim3(nblocks), dim3(kSortThreads), 0, s, *keys, *vals, *keys_alt,
                         *vals_alt, n, shift, offs, totals, nblocks);
    BIVX_HIP(hipGetLastError());
    iota = false;
    uint32_t *t = *keys; *keys = *keys_alt; *keys_alt = t;
    t = *vals; *vals = *vals_alt; *vals_alt = t;
  }
  return 0;
}

size_t finalize_gap_capacity(uint64_t nentries, uint32_t nseg) { return (size_t)(nentries / kCoopMax) + 2u * nseg + 64u; }
size_t finalize_gap_bytes(uint64_t nentries, uint32_t nseg) { return finalize_gap_capacity(nentries, nseg) * sizeof(Gap); }

int launch_finalize(const uint32_t *d_keys, const uint32_t *d_ids, const uint32_t *d_low, const uint32_t *d_high,
                    const SegDesc *d_seg, const uint2 *d_segkey, uint32_t nseg, uint2 *d_se, uint2 *d_rec,
                    uint32_t *d_table, uint64_t nentries, void *d_gaps, uint32_t *d_ngaps, uint32_t *d_max_cell,
                    size_t n, hipStream_t s) {
  if (n == 0 || nseg == 0) return 0;
  const uint32_t nchunks = grid_for(n, kThreads * kFinSlots);
  const unsigned grid = ((nchunks + 7u) / 8u) * 8u;
  const uint32_t cap = (uint32_t)finalize_gap_capacity(nentries, nseg);
  Gap *gaps = static_cast<Gap *>(d_gaps);
  if (d_keys)
    hipLaunchKernelGGL(k_finalize<true>, dim3(grid), dim3(kThreads), 0, s, d_keys, d_ids, d_low, d_high, d_seg, d_segkey, nseg,
                       d_se, d_rec, d_table, gaps, d_ngaps, cap, n, nchunks);
  else
    hipLaunchKernelGGL(k_finalize<false>, dim3(grid), dim3(kThreads), 0, s, d_keys, d_ids, d_low, d_high, d_seg, d_segkey,
                       nseg, d_se, d_rec, d_table, gaps, d_ngaps, cap, n, nchunks);
  hipLaunchKernelGGL(k_fill_gaps, dim3(256), dim3(kThreads), 0, s, gaps, d_ngaps, cap, d_table);
  hipLaunchKernelGGL(k_max_cell, dim3(grid_for(nentries, kThreads * 8, 512)), dim3(kThreads), 0, s, d_table, (size_t)nentries,
                     d_max_cell);
  BIVX_HIP(hipGetLastError());
  return 0;
}

int launch_self_queries(const uint2 *d_se, const SegDesc *d_seg, const uint32_t *d_seg_chrom, uint32_t nseg, size_t n,
                        uint32_t *d_qchrom, uint32_t *d_qlow, uint32_t *d_qhigh, hipStream_t s) {
  if (n == 0 || nseg == 0) return 0;
  hipLaunchKernelGGL(k_self_queries, dim3(grid_for(n, kThreads)), dim3(kThreads), 0, s, d_se, d_seg, d_seg_chrom, nseg, n,
                     d_qchrom, d_qlow, d_qhigh);
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
