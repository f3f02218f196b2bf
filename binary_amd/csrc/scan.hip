// scan.hip — device-wide exclusive prefix sum (u32 counts -> u64 offsets), hand-written for gfx950: per-query hit counts
// to CSR offsets where a call produces the counts in another order than the offsets (bivx_self_overlaps_dev).
//
// Three launches: per-tile reduce -> single-workgroup scan of the tile sums -> per-tile scan + base.
// Pure HBM streaming: reads the input twice (8 B/elem for u32) and writes the output once.
#include "common.h"

namespace bivx {
namespace {

constexpr int kScanThreads = 1024;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;  // 8192 elements per workgroup (with 2048 the single workgroup that scans
                                                      // the tile sums took 43 us of the 355 at 50 M elements)

template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v) {
  const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    T o = __shfl_up(v, d, kWave);
    if (lane >= d) v += o;
  }
  return v;
}

// exclusive scan of one value per thread across the workgroup; returns the exclusive prefix and the
// workgroup total through `total`. lds must hold (blockDim.x / 64) entries.
template <typename T>
__device__ __forceinline__ T block_exclusive_scan(T v, T *lds, T &total) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int nwave = blockDim.x >> 6;
  T incl = wave_inclusive_scan(v);
  if (lane == kWave - 1) lds[wave] = incl;
  __syncthreads();
  T wave_base = 0, tot = 0;
  for (int w = 0; w < nwave; ++w) {
    T s = lds[w];
    if (w < wave) wave_base += s;
    tot += s;
  }
  __syncthreads();
  total = tot;
  return wave_base + incl - v;
}

// HI: the values are the bits from `HI` upwards of 64-bit words (HI >= 32: bivx_self_overlaps_dev keeps a list's length above
// its 38-bit position in ONE word per id, so that the slot-order pass makes one scattered store per interval, not two)
template <int HI>
__device__ __forceinline__ uint32_t hi_of(uint32_t high_word) { return high_word >> (HI - 32); }

template <typename OutT, int HI>
__global__ __launch_bounds__(kScanThreads) void k_tile_reduce_hi(const uint64_t *__restrict__ in, size_t n, OutT *__restrict__ sums) {
  __shared__ OutT lds[kScanThreads / kWave];
  const size_t base = (size_t)blockIdx.x * kScanTile;
  OutT acc = 0;
  if (base + kScanTile <= n && (reinterpret_cast<uintptr_t>(in) & 15u) == 0) {
    const uint4 *p = reinterpret_cast<const uint4 *>(in + base);
    uint4 a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = p[k * kScanThreads + threadIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += (OutT)hi_of<HI>(a[k].y) + hi_of<HI>(a[k].w);
  } else {
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      size_t i = base + (size_t)k * kScanThreads + threadIdx.x;
      if (i < n) acc += hi_of<HI>((uint32_t)(in[i] >> 32));
    }
  }
  OutT total;
  (void)block_exclusive_scan<OutT>(acc, lds, total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

template <typename OutT>
__global__ __launch_bounds__(kScanThreads) void k_tile_reduce(const uint32_t *__restrict__ in, size_t n,
                                                              OutT *__restrict__ sums) {
  __shared__ OutT lds[kScanThreads / kWave];
  const size_t base = (size_t)blockIdx.x * kScanTile;
  OutT acc = 0;
  if (base + kScanTile <= n && (reinterpret_cast<uintptr_t>(in) & 15u) == 0) {  // a full tile of an aligned array: two 16-byte loads per thread (order is irrelevant for a sum)
    const uint4 *p = reinterpret_cast<const uint4 *>(in + base);
    const uint4 a = p[threadIdx.x], b = p[kScanThreads + threadIdx.x];
    acc = (OutT)a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
  } else {
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      size_t i = base + (size_t)k * kScanThreads + threadIdx.x;
      if (i < n) acc += in[i];
    }
  }
  OutT total;
  (void)block_exclusive_scan<OutT>(acc, lds, total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// one workgroup walks all tile sums (nb is small: n / 2048), leaves exclusive prefixes in place and the
// grand total in sums[nb].
template <typename OutT>
__global__ __launch_bounds__(1024) void k_sums_scan(OutT *__restrict__ sums, size_t nb) {
  __shared__ OutT lds[1024 / kWave];
  OutT carry = 0;
  for (size_t c = 0; c < nb; c += 1024) {
    size_t i = c + threadIdx.x;
    OutT v = i < nb ? sums[i] : 0;
    OutT total;
    OutT ex = block_exclusive_scan<OutT>(v, lds, total);
    if (i < nb) sums[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) sums[nb] = carry;
}

// HI == 0: `in_v` is an array of 32-bit values; otherwise of 64-bit words whose bits from HI upwards are the values
template <typename OutT, int HI = 0>
__global__ __launch_bounds__(kScanThreads) void k_tile_scan(const void *__restrict__ in_v, size_t n,
                                                            const OutT *__restrict__ sums, size_t nb,
                                                            OutT *__restrict__ out) {
  __shared__ OutT lds[kScanThreads / kWave];
  const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanItems;  // blocked
  uint32_t v[kScanItems];
  if (HI == 0) {
    const uint32_t *in = static_cast<const uint32_t *>(in_v);
    if (base + kScanItems <= n && (reinterpret_cast<uintptr_t>(in) & 15u) == 0) {
      const uint4 a = *reinterpret_cast<const uint4 *>(in + base);
      const uint4 b = *reinterpret_cast<const uint4 *>(in + base + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
      v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int k = 0; k < kScanItems; ++k) v[k] = (base + k < n) ? in[base + k] : 0u;
    }
  } else {
    const uint64_t *in = static_cast<const uint64_t *>(in_v);
    if (base + kScanItems <= n && (reinterpret_cast<uintptr_t>(in) & 15u) == 0) {
      const uint4 *p = reinterpret_cast<const uint4 *>(in + base);
      uint4 a[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] = p[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[2 * k] = hi_of<HI ? HI : 32>(a[k].y);
        v[2 * k + 1] = hi_of<HI ? HI : 32>(a[k].w);
      }
    } else {
#pragma unroll
      for (int k = 0; k < kScanItems; ++k) v[k] = (base + k < n) ? hi_of<HI ? HI : 32>((uint32_t)(in[base + k] >> 32)) : 0u;
    }
  }
  OutT tsum = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) tsum += v[k];
  OutT total;
  OutT run = block_exclusive_scan<OutT>(tsum, lds, total) + sums[blockIdx.x];
  if (sizeof(OutT) == 8 && base + kScanItems <= n && (reinterpret_cast<uintptr_t>(out) & 15u) == 0) {
    // a thread's eight offsets are 64 consecutive bytes: four 16-byte stores
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    u64x2 *o = reinterpret_cast<u64x2 *>(out + base);
#pragma unroll
    for (int k = 0; k < kScanItems; k += 2) {
      u64x2 w;
      w.x = (unsigned long long)run;
      w.y = (unsigned long long)(run + v[k]);
      o[k / 2] = w;
      run += (OutT)v[k] + (OutT)v[k + 1];
    }
  } else {
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      if (base + k < n) out[base + k] = run;
      run += v[k];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = sums[nb];
}

template <typename OutT>
int exclusive_scan_impl(const uint32_t *d_in, OutT *d_out, size_t n, void *d_scratch, hipStream_t s) {
  OutT *sums = static_cast<OutT *>(d_scratch);
  if (n == 0) {
    BIVX_HIP(hipMemsetAsync(d_out, 0, sizeof(OutT), s));
    return 0;
  }
  const size_t nb = (n + kScanTile - 1) / kScanTile;
  hipLaunchKernelGGL(k_tile_reduce<OutT>, dim3((unsigned)nb), dim3(kScanThreads), 0, s, d_in, n, sums);
  hipLaunchKernelGGL(k_sums_scan<OutT>, dim3(1), dim3(1024), 0, s, sums, nb);
  hipLaunchKernelGGL((k_tile_scan<OutT, 0>), dim3((unsigned)nb), dim3(kScanThreads), 0, s, (const void *)d_in, n, sums, nb, d_out);
  BIVX_HIP(hipGetLastError());
  return 0;
}

// Sums of the values per block of 256 and per tile of 8 192 (= 32 blocks) in ONE read of the words: what a kernel that takes 256
// consecutive values per workgroup needs to know where its piece of their prefix sum begins — the tile's exclusive prefix (after
// k_sums_scan) + the blocks before it inside the tile — so that it can make the prefix sum itself instead of reading it
// (bivx_self_overlaps_dev: k_permute_lines writes the offsets it computes; a separate scan pass read the words and wrote the
// offsets, and the gather read both again).
template <int HI>
__global__ __launch_bounds__(kScanThreads) void k_block_sums_hi(const uint64_t *__restrict__ in, size_t n,
                                                                uint64_t *__restrict__ sums256, uint64_t *__restrict__ sums8192) {
  static_assert(kScanThreads == 1024 && kScanItems == 8, "32 threads x 8 values = a block of 256");
  __shared__ uint64_t lds[kScanThreads / kWave];
  const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanItems;  // blocked: 32 threads = 256 values
  uint32_t mine = 0;  // (eight values below 2^26 each)
  if (base + kScanItems <= n && (reinterpret_cast<uintptr_t>(in) & 15u) == 0) {
    const uint4 *p = reinterpret_cast<const uint4 *>(in + base);
    uint4 a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = p[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) mine += hi_of<HI>(a[k].y) + hi_of<HI>(a[k].w);
  } else {
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
      if (base + k < n) mine += hi_of<HI>((uint32_t)(in[base + k] >> 32));
  }
  unsigned long long sum = mine;
#pragma unroll
  for (int d = 1; d < 32; d <<= 1) sum += __shfl_xor(sum, d, kWave);  // (the two halves of a wavefront are two blocks)
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  if ((lane & 31) == 0) {
    const size_t b = (size_t)blockIdx.x * (kScanTile / 256) + (threadIdx.x >> 5);
    if (b * 256 < n) sums256[b] = sum;
  }
  const unsigned long long both = sum + __shfl_xor(sum, 32, kWave);
  if (lane == 0) lds[wave] = both;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < kScanThreads / kWave; ++w) t += lds[w];
    sums8192[blockIdx.x] = t;
  }
}

template <int HI>
int exclusive_scan_hi_impl(const uint64_t *d_in, uint64_t *d_out, size_t n, void *d_scratch, hipStream_t s) {
  uint64_t *sums = static_cast<uint64_t *>(d_scratch);
  if (n == 0) {
    BIVX_HIP(hipMemsetAsync(d_out, 0, sizeof(uint64_t), s));
    return 0;
  }
  const size_t nb = (n + kScanTile - 1) / kScanTile;
  hipLaunchKernelGGL((k_tile_reduce_hi<uint64_t, HI>), dim3((unsigned)nb), dim3(kScanThreads), 0, s, d_in, n, sums);
  hipLaunchKernelGGL(k_sums_scan<uint64_t>, dim3(1), dim3(1024), 0, s, sums, nb);
  hipLaunchKernelGGL((k_tile_scan<uint64_t, HI>), dim3((unsigned)nb), dim3(kScanThreads), 0, s, (const void *)d_in, n, sums, nb, d_out);
  BIVX_HIP(hipGetLastError());
  return 0;
}

}  // namespace

size_t scan_scratch_bytes(size_t n) {  // [tile sums: nb + 2 | block sums: 32 per tile]
  const size_t nb = (n + kScanTile - 1) / kScanTile;
  return (nb + 2 + nb * (kScanTile / 256)) * sizeof(uint64_t);
}

// sums of the lengths (the words' bits from kSelfPosBits up) per block of 256 words, and the exclusive prefix of the sums per tile
// of 8 192 (the grand total behind the last): *tile_prefix = d_scratch, *block_sums behind it
int self_length_sums(const uint64_t *d_in, size_t n, void *d_scratch, const uint64_t **tile_prefix, const uint64_t **block_sums,
                     hipStream_t s) {
  const size_t nb = (n + kScanTile - 1) / kScanTile;
  uint64_t *t = static_cast<uint64_t *>(d_scratch), *b = t + nb + 2;
  *tile_prefix = t;
  *block_sums = b;
  if (n == 0) return 0;
  hipLaunchKernelGGL((k_block_sums_hi<kSelfPosBits>), dim3((unsigned)nb), dim3(kScanThreads), 0, s, d_in, n, b, t);
  hipLaunchKernelGGL(k_sums_scan<uint64_t>, dim3(1), dim3(1024), 0, s, t, nb);
  BIVX_HIP(hipGetLastError());
  return 0;
}

int exclusive_scan_u32_u64(const uint32_t *d_in, uint64_t *d_out, size_t n, void *d_scratch, hipStream_t s) {
  return exclusive_scan_impl<uint64_t>(d_in, d_out, n, d_scratch, s);
}

// d_out[i] = sum over j < i of (d_in[j] >> kSelfPosBits): the lengths kept above the positions (common.h)
int exclusive_scan_lengths_u64(const uint64_t *d_in, uint64_t *d_out, size_t n, void *d_scratch, hipStream_t s) {
  return exclusive_scan_hi_impl<kSelfPosBits>(d_in, d_out, n, d_scratch, s);
}

}  // namespace bivx
