// ub_gather.hip — microbenchmark behind DESIGN.md's choice of how a wavefront fetches its 64 candidate windows.
// 64 random 64-byte windows per wavefront round out of a table in the Infinity Cache / HBM, fetched three ways:
//   G=1  every lane fetches its own window with 4 dependent-free 16-byte loads (what round 1's kernel did)
//   G=4  4 lanes share a window, one 16-byte load each; 4 instructions cover the round's 64 windows
//   G=8  128-byte windows, 8 lanes each (one full line per window); 8 instructions cover 64 windows
// build: hipcc -O3 --offload-arch=gfx950 tools/ub_gather.hip -o /tmp/ub_gather ; run: /tmp/ub_gather [table MiB]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                          \
      return 1;                                                                      \
    }                                                                                \
  } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// WIN16: window size in 16-byte pieces (4 = 64 B, 8 = 128 B); G: lanes per window
template <int WIN16, int G>
__global__ __launch_bounds__(1024, 8) void k_gather(const uint4 *__restrict__ tab, uint32_t nwin_mask, int rounds,
                                                    uint32_t *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  uint32_t acc = 0;
  for (int r = 0; r < rounds; ++r) {
    // the round's 64 windows: window j of this wave and round
    if (G == 1) {
      const uint32_t w = mix(wave * 7919u + r * 104729u + lane) & nwin_mask;
      uint4 v[WIN16];
#pragma unroll
      for (int p = 0; p < WIN16; ++p) v[p] = tab[(size_t)w * WIN16 + p];
#pragma unroll
      for (int p = 0; p < WIN16; ++p) acc += v[p].x ^ v[p].w;
    } else {
      constexpr int kInstr = 64 / (64 / G);  // = G instructions cover 64 windows
      uint4 v[kInstr];
#pragma unroll
      for (int k = 0; k < kInstr; ++k) {
        const uint32_t j = k * (64 / G) + lane / G;  // window index within the round
        const uint32_t w = mix(wave * 7919u + r * 104729u + j) & nwin_mask;
        const int piece = lane % G;
        v[k] = piece < WIN16 ? tab[(size_t)w * WIN16 + piece] : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < kInstr; ++k) acc += v[k].x ^ v[k].w;
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int WIN16, int G>
int run(const uint4 *tab, size_t table_bytes, uint32_t *out, const char *name) {
  uint32_t nwin = 1;
  while ((size_t)nwin * 2 * WIN16 * 16 <= table_bytes) nwin *= 2;
  const int rounds = 64, blocks = 2048;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_gather<WIN16, G>), dim3(blocks), dim3(1024), 0, 0, tab, nwin - 1, rounds, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i)
    hipLaunchKernelGGL((k_gather<WIN16, G>), dim3(blocks), dim3(1024), 0, 0, tab, nwin - 1, rounds, out);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  const double windows = (double)blocks * 16 * rounds * 64;
  printf("%-44s %8.3f ms  %7.2f G windows/s  %6.2f TB/s of window bytes\n", name, ms, windows / ms / 1e6,
         windows * WIN16 * 16 / ms / 1e9);
  return 0;
}

int main(int argc, char **argv) {
  const size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 80;
  const size_t bytes = mib << 20;
  uint4 *tab = nullptr;
  uint32_t *out = nullptr;
  CK(hipMalloc((void **)&tab, bytes));
  CK(hipMalloc((void **)&out, 64));
  CK(hipMemset(tab, 1, bytes));
  printf("table %zu MiB, 64 random windows per wavefront round, 2048 x 1024 threads\n", mib);
  if (run<4, 1>(tab, bytes, out, "64-B windows, 1 lane each (4 loads per lane)")) return 1;
  if (run<4, 4>(tab, bytes, out, "64-B windows, 4 lanes each (4 instr per round)")) return 1;
  if (run<8, 1>(tab, bytes, out, "128-B windows, 1 lane each (8 loads per lane)")) return 1;
  if (run<8, 8>(tab, bytes, out, "128-B windows, 8 lanes each (8 instr per round)")) return 1;
  if (run<2, 1>(tab, bytes, out, "32-B windows, 1 lane each (2 loads per lane)")) return 1;
  if (run<2, 2>(tab, bytes, out, "32-B windows, 2 lanes each (2 instr per round)")) return 1;
  if (run<1, 1>(tab, bytes, out, "16-B windows, 1 lane each (1 load per lane)")) return 1;
  return 0;
}
