// query_pipe.hip — the pipelined form of the single-pass query kernel (gfx950, wave64), for the common case:
// one segment per chromosome, no fused filter, canonical CSR in index order, few ids per query. Same protocol and
// same output, bit for bit, as k_query_fused (query_fused.hip), which keeps every other case.
//
// Why: a tile of k_query_fused lives ~22 us (config 3, generation order) of which ~8.5 us are the counting; the rest
// is the ticket, the barrier skew of sixteen wavefronts, the sweep over the earlier tiles' totals and the way out
//  — and a CU holds two tiles, so throughput is 2048 queries per tile lifetime. Here workgroups
// are PERSISTENT (two per CU) and a tile's output is DEFERRED by one iteration:
//   iteration i:  count tile N (masks; first ids in the keep slots / slab) -> barrier A -> wave 0: publish N's total,
//                 draw the next ticket, sum the totals before tile O = the tile of iteration i-1 (published an
//                 iteration ago: its predecessors are through, the sweep finds every word valid) -> barrier B ->
//                 every wavefront: prefetch the next tile's queries; write O's offsets and stream O's ids out of the
//                 stage; lay N's ids out in the stage, back to back as they will sit in the output.
// What a tile carries across the iteration is its stage (the 2 KiB per wavefront that k_query_fused uses for output
// staging) and ONE register per lane (list offset | count); the keep slots / slab are free again for the next tile.
// The sweep's latency, the store latency and a good part of the barrier skew overlap the next tile's counting.
// A tile with a wavefront that cannot stage (more than kStage ids, a wavefront-cooperative window) gets its counts
// and offsets here — the prefix chain needs them — and is put on a list; k_fill_tiles, launched behind this kernel,
// enumerates the listed tiles' ids into place (it finds the list empty on ordinary data and returns).
// Tickets are drawn a few microseconds ahead (after barrier A), not a whole tile ahead: a ticket held by a busy
// workgroup delays that tile's published total and every later sweep waits for it (measured: 0.49 -> 0.84 ms).
#include <cstdlib>

#include "prefix_device.h"
#include "query_device.h"

namespace bivx {
namespace {

constexpr int kPThreads = 1024;
constexpr int kPWaves = kPThreads / kWave;
constexpr uint32_t kPStage = 512;   // ids per wavefront stage
constexpr uint32_t kPKeep = 8;      // ids kept per query while counting (a wavefront's 64 x 8 slots are its slab too)
constexpr uint32_t kPGather = 4;    // ids a lane re-reads per step when more than kPKeep were found outside the slab

// Diagnostic build only (-DBIVX_STAMPS): per-tile wall-clock stamps, written to a buffer no other code reads.
#ifdef BIVX_STAMPS
constexpr unsigned kPStampTiles = 1024, kPStampSlots = 8;
__device__ unsigned long long g_pstamps[kPStampTiles * kPStampSlots];
#define PSTAMP(t, k) \
  if (threadIdx.x == 0) g_pstamps[((t) % kPStampTiles) * kPStampSlots + (k)] = __builtin_amdgcn_s_memrealtime()
#else
#define PSTAMP(t, k)
#endif

struct PipeArgs {
  const uint32_t *qchrom, *qlow, *qhigh;
  size_t q_begin, q_end;
  uint64_t *offsets;
  uint32_t *hits;
  uint64_t cap;
  uint64_t *ws;
  uint32_t ntiles;
  int flags;
};

// The kernel's arguments stay where the launch put them — the kernarg segment, constant memory — and are re-read
// where they are used: a persistent loop otherwise keeps all ~50 scalar registers of pointers and sizes alive across
// every phase, and with 80 SGPRs per wavefront (what eight wavefronts per SIMD leave) the compiler answers with a
// hundred scalar spills that push the vector registers into scratch. The empty asm makes the segment's address
// opaque at every phase boundary, so nothing read through it can be held across one.
struct KernArgs {
  IndexView v;
  PipeArgs a;
};
typedef const __attribute__((address_space(4))) KernArgs *kargs_t;
__device__ __forceinline__ kargs_t fresh(kargs_t p) {
  asm volatile("" : "+s"(p));
  return p;
}

__global__ __launch_bounds__(kPThreads, 8) void k_query_pipe(IndexView v_in, PipeArgs a_in) {
  (void)v_in;
  (void)a_in;
  kargs_t ka = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
#define A(f) (fresh(ka)->a.f)
  // a view of the index made of freshly read words (only the fields a phase uses are ever loaded)
  auto view_now = [&]() {
    kargs_t p = fresh(ka);
    IndexView w;
    w.se = p->v.se;
    w.rec = p->v.rec;
    w.id = p->v.id;
    w.table = p->v.table;
    w.seg = p->v.seg;
    w.chrom_rng = p->v.chrom_rng;
    w.nchrom = p->v.nchrom;
    w.nseg = p->v.nseg;
    w.max_segs = 1;
    w.flt_kind = BIVX_FILTER_NONE;
    w.flt_dist = 0;
    w.flt_strand = 0;
    w.flt_qaux = nullptr;
    w.flt_iaux = nullptr;
    w.err = p->v.err;
    return w;
  };
  auto prefix_now = [&]() {
    kargs_t p = fresh(ka);
    PrefixCtx c;
    c.group = p->a.ws + kWsStatus;
    c.status = c.group + kFMaxGroups;
    c.ntiles = p->a.ntiles;
    const uint32_t wl = ((uint32_t)p->a.flags >> kFlagWaitShift) & 0xFFu;
    c.wait_ticks = 1ull << (wl ? wl : kWaitLog2Default);
    c.err = p->v.err;
    return c;
  };
  __shared__ SegDesc s_seg[kLdsSegs];
  __shared__ uint2 s_cs[kLdsChroms];
  // the words the wavefronts hand each other are double-buffered by the parity of the iteration: a wavefront that
  // is through with an iteration walks into the next while slower ones still finish; the two barriers of an
  // iteration are the only places where the sixteen wait for each other
  __shared__ uint32_t s_tile[2];
  __shared__ uint32_t s_wsum[2][kPWaves];
  __shared__ uint64_t s_w64[kPWaves];               // wavefront totals of a listed tile, in 64 bits
  __shared__ uint64_t s_base[2];                    // first output position of the pending tile [0], of a listed new tile [1]
  __shared__ uint4 s_keep[kPThreads * (kPKeep / 4)];   // keep slots / slabs of the tile being counted
  __shared__ uint32_t s_stage[kPWaves][kPStage];       // ids of the pending tile, laid out as in the output
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;

  if (threadIdx.x == 0) s_tile[0] = atomicAdd(reinterpret_cast<unsigned int *>(A(ws) + kWsTicket), 1u);
  const SegDesc *segs;
  const uint2 *cs;
  {
    const IndexView v0 = view_now();
    stage_descriptors<true>(v0, s_seg, s_cs, segs, cs);
  }
  __syncthreads();

  uint32_t *const kept = reinterpret_cast<uint32_t *>(&s_keep[threadIdx.x * (kPKeep / 4)]);
  uint4 *const slab = &s_keep[(threadIdx.x & ~(kWave - 1)) * (kPKeep / 4)];
  uint32_t *const stage = s_stage[wave];

  // the pending tile of this workgroup: counted and published, its ids in the stage, output deferred
  bool have_old = false;
  uint32_t o_tile = 0, o_wbase = 0, o_wtotal = 0, o_state = 0;
  auto flush_old = [&](uint64_t base) {
    kargs_t p = fresh(ka);
    const size_t q_end = p->a.q_end;
    const size_t q = p->a.q_begin + (size_t)o_tile * kPThreads + threadIdx.x;
    const uint64_t wpos0 = base + o_wbase;
    if (q < q_end) {
      uint64_t *off = p->a.offsets;
      off[q] = wpos0 + (o_state >> 16);
      if (q == q_end - 1) off[q_end] = wpos0 + (o_state >> 16) + (o_state & 0xFFFFu);
    }
    const uint64_t cap = p->a.cap;
    if (cap != 0) {
      uint32_t *hits = p->a.hits;
      for (uint32_t i = lane; i < o_wtotal; i += kWave) {
        const uint64_t pos = wpos0 + i;
        if (pos < cap) hits[pos] = stage[i];
      }
    }
  };
  auto query_of = [&](uint32_t t) {
    kargs_t p = fresh(ka);
    const size_t q = p->a.q_begin + (size_t)t * kPThreads + threadIdx.x;
    IndexView w;  // load_query reads nchrom only (no filter)
    w.nchrom = p->v.nchrom;
    w.flt_qaux = nullptr;
    return load_query<false>(w, cs, p->a.qchrom, p->a.qlow, p->a.qhigh, q, t < p->a.ntiles && q < p->a.q_end);
  };

  // the first tile's queries; later tiles' are fetched an iteration ahead, behind barrier B
  uint32_t tile = __builtin_amdgcn_readfirstlane(s_tile[0]);
  Query qy = query_of(tile);

  for (uint32_t par = 0;; par ^= 1u) {
    if (tile >= A(ntiles)) {
      // Every workgroup draws exactly one ticket beyond the batch, so none can reach ntiles + gridDim.x unless the
      // counter was not zero when the launch began (a launch that died half-way, a caller workspace that was not
      // cleared): then tiles were skipped and nothing this launch wrote can be trusted. Say so.
      if (tile >= A(ntiles) + gridDim.x && threadIdx.x == 0) raise_error(fresh(ka)->v.err, kErrWorkspace);
      if (have_old) {  // drain
        if (wave == 0) {
          const PrefixCtx pc = prefix_now();
          const uint64_t sum = tiles_before(pc, o_tile, lane);
          if (lane == 0) s_base[0] = sum + (A(q_begin) ? A(offsets)[A(q_begin)] : 0ull);
        }
        __syncthreads();
        flush_old(s_base[0]);
      }
      break;
    }

    PSTAMP(tile, 0);
    // ---- count the new tile ---------------------------------------------------------------------------------
    Replay rp;
    uint32_t cnt;
    {
      const IndexView v1 = view_now();
      cnt = enumerate_hits<Mode::Count, false, false, kPKeep, kRows, true>(v1, segs, qy, nullptr, 0, 0, &rp, kept,
                                                                         nullptr, slab);
    }
    uint32_t incl = cnt;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d, kWave);
      if (lane >= d) incl += o;
    }
    const uint32_t loff = incl - cnt;
    // 2^22 hits in one lane would overflow the 32-bit tile sums: such a tile is counted in 64 bits below
    const bool huge = cnt >= (1u << 22);
    const uint32_t wtotal = __builtin_amdgcn_readfirstlane(__shfl(incl, kWave - 1, kWave));
    // (uniform per wavefront; a pure count has no ids to lay out, only the packed state must fit)
    const bool no_ids = A(cap) == 0;
    const bool staged = no_ids ? !__any(huge) && wtotal < 65536u : !__any(huge || !rp.ok) && wtotal <= kPStage;
    if (lane == kWave - 1) s_wsum[par][wave] = incl;
    PSTAMP(tile, 1);
    const bool listed = __syncthreads_or(!staged) != 0;  // (barrier A)
    PSTAMP(tile, 2);
    uint64_t total = 0, wbase = 0;
    if (!listed) {
      uint32_t wb = 0, t32 = 0;
#pragma unroll
      for (int w = 0; w < kPWaves; ++w) {
        const uint32_t x = s_wsum[par][w];
        if (w < wave) wb += x;
        t32 += x;
      }
      total = t32;
      wbase = wb;
    } else {  // rare: the sixteen wavefront totals again, in 64 bits
      const uint64_t wt = wave_total64(cnt);
      if (lane == 0) s_w64[wave] = wt;
      __syncthreads();
      for (int w = 0; w < kPWaves; ++w) {
        const uint64_t x = s_w64[w];
        if (w < wave) wbase += x;
        total += x;
      }
    }

    // ---- publish, next ticket, sweeps ------------------------------------------------------------------------
    if (wave == 0) {
      const PrefixCtx pc = prefix_now();
      uint64_t *ws = A(ws);
      uint32_t next_tile = 0;
      if (lane == 0) next_tile = atomicAdd(reinterpret_cast<unsigned int *>(ws + kWsTicket), 1u);
      publish_tile(pc, tile, total, lane);
      const size_t qb = A(q_begin);
      const uint64_t carry = qb ? A(offsets)[qb] : 0ull;
      if (have_old) {
        const uint64_t sum = tiles_before(pc, o_tile, lane);
        if (lane == 0) s_base[0] = sum + carry;
      }
      if (listed) {
        const uint64_t sum = tiles_before(pc, tile, lane);
        if (lane == 0) {
          s_base[1] = sum + carry;
          uint32_t *todo = reinterpret_cast<uint32_t *>(pc.status + kFMaxTiles);
          todo[atomicAdd(reinterpret_cast<unsigned int *>(ws + kWsTodo), 1u)] = tile;
        }
      }
      if (lane == 0) s_tile[par ^ 1u] = next_tile;
      PSTAMP(tile, 3);
    }
    __syncthreads();  // (barrier B)
    PSTAMP(tile, 4);

    // ---- the next tile's queries leave now; the pending tile goes out; the new tile's ids are laid out -----------
    const uint32_t ntile = __builtin_amdgcn_readfirstlane(s_tile[par ^ 1u]);
    const Query nqy = query_of(ntile);
    if (have_old) flush_old(s_base[0]);
    PSTAMP(tile, 5);
    have_old = !listed;
    if (listed) {
      // counts and offsets only; k_fill_tiles writes the ids
      uint64_t lpos = loff;
      if (__any(huge)) {  // a wavefront with such a lane: its list offsets need 64 bits too
        uint64_t i64 = cnt;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
          const uint64_t o = __shfl_up((unsigned long long)i64, d, kWave);
          if (lane >= d) i64 += o;
        }
        lpos = i64 - cnt;
      }
      const uint64_t pos = s_base[1] + wbase + lpos;
      kargs_t p = fresh(ka);
      const size_t q_end = p->a.q_end;
      const size_t q = p->a.q_begin + (size_t)tile * kPThreads + threadIdx.x;
      if (q < q_end) {
        uint64_t *off = p->a.offsets;
        off[q] = pos;
        if (q == q_end - 1) off[q_end] = pos + cnt;
      }
    } else {
      o_tile = tile;
      o_wbase = (uint32_t)wbase;
      o_wtotal = wtotal;
      o_state = (loff << 16) | cnt;
      if (!no_ids) {
        wave_sync_lds();  // the stage was read by flush_old just above
        if (cnt) {
          uint64_t mrem = rp.mask;
          if (rp.lds) {  // everything is in the wavefront's slab
            const uint2 *s2 = reinterpret_cast<const uint2 *>(slab) + (rp.al - rp.lbase);
            for (uint32_t k = 0; k < cnt; ++k) {
              const uint32_t j = (uint32_t)__ffsll((long long)mrem) - 1u;
              mrem &= mrem - 1;
              stage[loff + k] = s2[j].y;
            }
          } else {
            const uint32_t nk = rp.kept ? (cnt < kPKeep ? cnt : kPKeep) : 0u;
            for (uint32_t k = 0; k < nk; ++k) {
              stage[loff + k] = kept[k];
              mrem &= mrem - 1;
            }
            if (nk < cnt) {  // the rest is re-read next to its record
              kargs_t p = fresh(ka);
              const uint2 *rec = p->v.rec;
              const uint32_t *idv = p->v.id;
              for (uint32_t k = nk; k < cnt; k += kPGather) {
                uint32_t ids[kPGather];
#pragma unroll
                for (uint32_t i = 0; i < kPGather; ++i)
                  if (k + i < cnt) {
                    const uint32_t j = (uint32_t)__ffsll((long long)mrem) - 1u;
                    mrem &= mrem - 1;
                    ids[i] = rp.packed ? rec[rp.al + j].y : idv[rp.al + j];
                  }
#pragma unroll
                for (uint32_t i = 0; i < kPGather; ++i)
                  if (k + i < cnt) stage[loff + k + i] = ids[i];
              }
            }
          }
        }
        wave_sync_lds();
      }
    }
    PSTAMP(tile, 6);
    tile = ntile;
    qy = nqy;
  }

  // self-cleaning workspace: every workgroup bumps `done` when it leaves — its tiles are written, its sweeps long
  // over, its last ticket drawn; the one that sees gridDim.x - 1 knows nobody touches the words any more and zeroes
  // them for the next launch (the list of tiles for k_fill_tiles stays: that kernel clears its count).
  if ((A(flags) & kFlagSelfClean) && wave == 0) {
    uint64_t *ws = A(ws);
    uint32_t last = 0;
    if (lane == 0) last = atomicAdd(reinterpret_cast<unsigned int *>(ws + kWsDone), 1u) == gridDim.x - 1 ? 1u : 0u;
    if (__shfl(last, 0, kWave)) {
      const uint32_t ntiles = A(ntiles);
      uint64_t *group = ws + kWsStatus, *status = group + kFMaxGroups;
      for (uint32_t t = (uint32_t)lane; t < ntiles; t += kWave) {
        status[t] = 0;
        if (t < (ntiles + kWave - 1) / kWave) group[t] = 0;
      }
      if (lane == 0) {
        ws[kWsTicket] = 0;
        ws[kWsDone] = 0;
      }
    }
  }
#undef A
}

// Fills the ids of the tiles k_query_pipe listed: offsets are in place, the enumeration is the general one
// (k_query<Fill>'s, wavefront-cooperative windows included). One work item = a quarter tile (256 queries); the grid is
// fixed and strides over the items, so the launch needs no host knowledge of the list; the last workgroup to finish
// clears the count.
__global__ __launch_bounds__(kQThreads) void k_fill_tiles(IndexView v, PipeArgs a) {
  __shared__ SegDesc s_seg[kLdsSegs];
  __shared__ uint2 s_cs[kLdsChroms];
  __shared__ uint32_t s_last;
  uint64_t *status = a.ws + kWsStatus + kFMaxGroups;
  const uint32_t *todo = reinterpret_cast<const uint32_t *>(status + kFMaxTiles);
  const uint32_t n =
      __hip_atomic_load(reinterpret_cast<const uint32_t *>(a.ws + kWsTodo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (n == 0) return;  // (nothing was listed: nobody touched the counters either)
  const SegDesc *segs;
  const uint2 *cs;
  stage_descriptors<true>(v, s_seg, s_cs, segs, cs);
  __syncthreads();
  constexpr uint32_t kParts = kPThreads / kQThreads;
  for (uint32_t w = blockIdx.x; w < n * kParts; w += gridDim.x) {
    const uint32_t tile = todo[w / kParts];
    const size_t q = a.q_begin + (size_t)tile * kPThreads + (size_t)(w % kParts) * kQThreads + threadIdx.x;
    const bool valid = q < a.q_end;
    const Query qy = load_query<false>(v, cs, a.qchrom, a.qlow, a.qhigh, q, valid);
    const uint64_t pos = valid ? a.offsets[q] : 0;
    (void)enumerate_hits<Mode::Fill, false>(v, segs, qy, a.hits, pos, a.cap, nullptr);
  }
  // every workgroup that saw a non-empty list reports; the last one clears the list for the next call
  if (threadIdx.x == 0)
    s_last = atomicAdd(reinterpret_cast<unsigned int *>(a.ws + kWsTodoDone), 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (s_last && threadIdx.x == 0) {
    a.ws[kWsTodo] = 0;
    a.ws[kWsTodoDone] = 0;
  }
}

}  // namespace

// true if the pipelined kernel handled the launch (the caller falls back to k_query_fused otherwise)
bool pipe_eligible(const IndexView &v, size_t q, uint64_t cap, bool sort_ids, bool unordered) {
  static const int mode = [] {  // BIVX_PIPE: 0 = never, 1 = when eligible (default), 2 = also for small batches (tests)
    const char *e = std::getenv("BIVX_PIPE");
    return e ? std::atoi(e) : 1;
  }();
  if (!mode || unordered || sort_ids || v.flt_kind != BIVX_FILTER_NONE || v.max_segs > 1 || !fits_lds(v)) return false;
  // batches of a few tiles per resident workgroup gain nothing from a pipeline that has to fill and drain
  // (config 2, 977 tiles: 60 us against 55 for k_query_fused)
  if (q < (size_t)4 * 512 * kPThreads && mode != 2) return false;
  // few ids per query: a wavefront's 64 lists must fit its stage (the capacity is the only bound the host has)
  return cap <= (uint64_t)6 * q;
}

int launch_query_pipe(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                      size_t q0, size_t q1, unsigned tiles, uint64_t *d_offsets, uint32_t *d_hits, uint64_t cap,
                      uint64_t *ws, int flags, hipStream_t s) {
  unsigned wgs = 512;
  {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
      wgs = 2u * (unsigned)cus;  // two workgroups of 1024 threads are resident per CU (64 VGPRs, 71 KiB of LDS)
    if (const char *e = std::getenv("BIVX_PIPE_WGS")) {  // tuning / test knob
      const long w = std::atol(e);
      if (w >= 1 && w <= 65536) wgs = (unsigned)w;
    }
  }
  PipeArgs a{d_qchrom, d_qlow, d_qhigh, q0, q1, d_offsets, d_hits, cap, ws, tiles, flags};
  hipLaunchKernelGGL(k_query_pipe, dim3(tiles < wgs ? tiles : wgs), dim3(kPThreads), 0, s, v, a);
  hipLaunchKernelGGL(k_fill_tiles, dim3(256), dim3(kQThreads), 0, s, v, a);
  BIVX_HIP(hipGetLastError());
  return 0;
}

#ifdef BIVX_STAMPS
extern "C" int bivx_debug_pstamps(unsigned long long *out, size_t n) {
  if (n > (size_t)kPStampTiles * kPStampSlots) n = (size_t)kPStampTiles * kPStampSlots;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pstamps), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace bivx
