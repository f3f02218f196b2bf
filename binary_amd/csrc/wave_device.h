// wave_device.h — wavefront-wide reductions and the 64-lane prefix sum on the DPP datapath (gfx950, wave64), shared by
// the query kernels (query_device.h) and the build kernels (build.hip).
#ifndef BIVX_WAVE_DEVICE_H_
#define BIVX_WAVE_DEVICE_H_

#include "common.h"

namespace bivx {
namespace {

// Wavefront reductions and the 64-lane prefix sum on the DPP datapath (gfx9 row_shr / row_bcast): six VALU instructions
// each and no LDS round trip (the __shfl forms are a ds_bpermute plus its address arithmetic per step — 6 LDS round
// trips and ~24 VALU instructions for one scan). All 64 lanes must be active. A lane without a source (row_shr across
// the start of a row, rows masked out of a row_bcast step) combines with `old`, the identity of the operation — which is
// also what lets the compiler fold the move into the arithmetic instruction (v_add_u32_dpp, v_max_u32_dpp).
template <int CTRL, int ROWS = 0xF>
__device__ __forceinline__ uint32_t dpp_from(uint32_t old, uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)x, CTRL, ROWS, 0xF, false);
}
constexpr int kDppShr1 = 0x111, kDppShr2 = 0x112, kDppShr4 = 0x114, kDppShr8 = 0x118;
constexpr int kDppBcast15 = 0x142, kDppBcast31 = 0x143;  // lane 15 of a row to the next row; lane 31 to rows 2 and 3

// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x) {
  x += dpp_from<kDppShr1>(0u, x);
  x += dpp_from<kDppShr2>(0u, x);
  x += dpp_from<kDppShr4>(0u, x);
  x += dpp_from<kDppShr8>(0u, x);
  x += dpp_from<kDppBcast15, 0xA>(0u, x);
  x += dpp_from<kDppBcast31, 0xC>(0u, x);
  return x;
}
__device__ __forceinline__ uint32_t wave_last(uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane((int)x, kWave - 1); }
// (the results below are wavefront-uniform: lane 63 of the running form holds the reduction)
__device__ __forceinline__ uint32_t wave_sum(uint32_t x) { return wave_last(wave_scan_incl(x)); }
__device__ __forceinline__ uint32_t wave_max(uint32_t x) {
  x = max(x, dpp_from<kDppShr1>(0u, x));
  x = max(x, dpp_from<kDppShr2>(0u, x));
  x = max(x, dpp_from<kDppShr4>(0u, x));
  x = max(x, dpp_from<kDppShr8>(0u, x));
  x = max(x, dpp_from<kDppBcast15, 0xA>(0u, x));
  x = max(x, dpp_from<kDppBcast31, 0xC>(0u, x));
  return wave_last(x);
}
__device__ __forceinline__ uint32_t wave_min(uint32_t x) {
  x = min(x, dpp_from<kDppShr1>(~0u, x));
  x = min(x, dpp_from<kDppShr2>(~0u, x));
  x = min(x, dpp_from<kDppShr4>(~0u, x));
  x = min(x, dpp_from<kDppShr8>(~0u, x));
  x = min(x, dpp_from<kDppBcast15, 0xA>(~0u, x));
  x = min(x, dpp_from<kDppBcast31, 0xC>(~0u, x));
  return wave_last(x);
}

}  // namespace
}  // namespace bivx

#endif  // BIVX_WAVE_DEVICE_H_
