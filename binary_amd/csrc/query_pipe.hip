// query_pipe.hip — the pipelined form of the single-pass query kernel (gfx950, wave64), for the common case:
// one segment per chromosome, no fused filter, canonical CSR in index order, few ids per query. Same protocol between
// workgroups and the same output, bit for bit, as k_query_fused (query_fused.hip), which keeps every other case.
//
// Why: a tile of k_query_fused lives ~22 us (config 3, generation order) of which 7-8 us are the counting; the rest is
// the ticket, the barrier skew of sixteen wavefronts, the sweep over the earlier tiles' totals and the way out
//  — and a CU holds two tiles, so throughput is 2048 queries per tile lifetime.
//
// Here workgroups are PERSISTENT (two per CU), their wavefronts are SPECIALISED and they never meet at a barrier:
//   * fifteen WORKER wavefronts each own 64 consecutive queries of every tile the workgroup draws (a tile is 960
//     queries). A worker counts its slice of tile N (masks; first ids in its keep slots / slab), reports its total,
//     then writes the offsets and streams the ids of its slice of the PREVIOUS tile O out of its stage (they were laid
//     out there an iteration ago, exactly as they sit in the output), lays N's ids out in the stage, and goes on to the
//     next tile. What a slice carries across the iteration is the stage (2 KiB) and ONE register per lane.
//   * one SERVICE wavefront draws the tickets (one tile ahead, so that the workers can fetch the next tile's queries
//     early), waits — in LDS — for the fifteen totals of a tile, publishes the tile's total (one 8-byte agent-scope
//     word), sums the totals of all earlier tiles (the wide sweep of query_fused.hip, the only place that waits for
//     other workgroups) and hands the tile's first output position and the workers' bases back through LDS.
// A worker only waits for words of its own workgroup that were written an iteration earlier, so the ticket's round
// trip, the barrier skew, the sweep's latency and the store latency all overlap counting. The words live in a ring of
// kRing tile slots in LDS; a slot is recycled when all fifteen workers have flushed the tile that used it.
// A slice that cannot be staged (more than kPStage ids, a wavefront-cooperative window) gets its counts and offsets
// here — the prefix chain needs them — and is put on a list; k_fill_slices, launched behind this kernel, enumerates
// the listed slices' ids into place (it finds the list empty on ordinary data and returns).
// Deadlock: a worker waits for its own service wavefront only; a service wavefront waits for workers of its own
// workgroup (which never wait for other workgroups) and for tiles with SMALLER tickets, all drawn by resident
// workgroups; the smallest unpublished tile can therefore always be finished. Waits on other workgroups are bounded
// by wall time and an expired one fails the call (prefix_device.h).
#include <atomic>
#include <cstdlib>

#include "prefix_device.h"
#include "query_device.h"

namespace bivx {
namespace {

#ifndef BIVX_PIPE_THREADS
#define BIVX_PIPE_THREADS 1024  // (experiments: 512 = seven workers and the service wavefront)
#endif
constexpr int kPThreads = BIVX_PIPE_THREADS;
constexpr int kPWaves = kPThreads / kWave;
constexpr int kWorkers = kPWaves - 1;          // wavefront kWorkers is the service wavefront
constexpr uint32_t kPTile = kWorkers * kWave;  // 960 queries
constexpr uint32_t kRing = 8;                  // tile slots in LDS
#ifndef BIVX_DEFER
#define BIVX_DEFER 2
#endif
#ifndef BIVX_WSLEEP
#define BIVX_WSLEEP 16  // s_sleep argument between two polls of a worker's wait in LDS (units of 64 cycles)
#endif
#ifndef BIVX_SSLEEP
#define BIVX_SSLEEP 2   // ... between two passes of the service wavefront that made no progress
#endif
#ifndef BIVX_COOP_DEPTH
#define BIVX_COOP_DEPTH 2   // rounds of coop_mask32 whose loads are in flight together (1: load, wait, evaluate; 2 fits 64 registers)
#endif
#ifndef BIVX_PLACE_DEPTH
#define BIVX_PLACE_DEPTH 2   // rounds of coop_place16 whose loads are in flight together
#endif
#ifndef BIVX_EXP
#define BIVX_EXP 0      // experiments (1, 2, 4: WRONG RESULTS, timing and instruction counts only): 1 no id layout, 2 no id stream-out, 4 no keep slots; 8: coop_mask32 instead of coop_place16 (same results)
#endif
constexpr uint32_t kDefer = BIVX_DEFER;  // iterations between counting a slice and writing it out (experiments: 1, 3)
static_assert(kDefer >= 1 && kDefer <= 3, "deferral depth");
constexpr uint32_t kPStage = kDefer == 2 ? 320 : (640 / kDefer) & ~3u;   // ids per wavefront stage (there are two: a slice waits two iterations for its base)
#ifndef BIVX_FILL_BLOCKS
#define BIVX_FILL_BLOCKS 256
#endif
constexpr unsigned kFillBlocks = BIVX_FILL_BLOCKS;  // workgroups (of four wavefronts) of k_fill_slices
constexpr uint32_t kMaxCellForPipe = 64;  // indexes with a fuller directory cell than this stay on k_query_fused
constexpr uint32_t kPKeep = 8;      // ids kept per query while counting (a wavefront's 64 x 8 slots are its slab too)

// Diagnostic build only (-DBIVX_STAMPS): wall-clock stamps of worker 0, written to a buffer no other code reads.
#ifdef BIVX_STAMPS
constexpr unsigned kPStampTiles = 1024, kPStampSlots = 8;
__device__ unsigned long long g_pstamps[kPStampTiles * kPStampSlots];
#define PSTAMP(t, k) \
  if (threadIdx.x == 0) g_pstamps[((t) % kPStampTiles) * kPStampSlots + (k)] = __builtin_amdgcn_s_memrealtime()
// ... and of the workgroup's life (k_query_pipe): 0 entry, 1 descriptors staged, 2 first ticket seen, 3 first queries and
// probe issued, 4 worker 0 leaves (its last slice is out), 5 the service wavefront leaves, 6 tiles the workgroup drew
__device__ unsigned long long g_wgstamps[1024 * 8];
#define WGSTAMP(k) \
  if ((threadIdx.x & 63u) == 0 && blockIdx.x < 1024) g_wgstamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime()
#define WGSTAMP_VAL(k, v) \
  if ((threadIdx.x & 63u) == 0 && blockIdx.x < 1024) g_wgstamps[blockIdx.x * 8 + (k)] = (v)
#else
#define PSTAMP(t, k)
#define WGSTAMP(k)
#define WGSTAMP_VAL(k, v)
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
  uint32_t seq;   // != 0: ids ascending inside every query
  // begin / count output (bivx_query_dev_u): `offsets` receives every query's first position only (no entry q_end),
  // `counts` its number of ids and `total_out` the batch's total
  uint32_t *counts;
  uint64_t *total_out;
  // Results wanted in ANOTHER order than the queries' (bivx_self_overlaps_dev: the queries are the index's own intervals
  // in slot order, the CSR is wanted in id order; k_query_pipe_dense and k_fill_slices only). perm[q] = the query's
  // position in the result. The ids are written as always — the batch's lists back to back, in the queries' order — but
  // instead of offsets[q] the kernel leaves src_by_id[perm[q]] = the query's number of ids << kSelfPosBits | where its list
  // begins (one scattered store per query); k_permute_lists then moves the lists into the result's order.
  const uint32_t *perm;
  uint64_t *src_by_id;
  uint32_t *unused_;
  uint32_t rec_delta;  // k_query_pipe_ms: byte distance from se[] to rec[] (one block, below 4 GB)
  uint32_t tile_q;     // queries per tile of the kernel that listed slices for k_fill_slices (960; k_query_pipe_ms: 448)
};
constexpr int kFlagSorted = 4;  // k_query_pipe_dense: the batch is known to be position-sorted (no order probe was launched)

// The kernel's arguments stay where the launch put them — the kernarg segment, constant memory — and are re-read
// where they are used: a persistent loop otherwise keeps all ~50 scalar registers of pointers and sizes alive across
// every phase, and with 80 SGPRs per wavefront (what eight wavefronts per SIMD leave) the compiler answers with a
// hundred scalar spills that push the vector registers into scratch. The empty asm makes the segment's address
// opaque wherever it is used, so nothing read through it is held longer than its use.
struct KernArgs {
  IndexView v;
  PipeArgs a;
};
typedef const __attribute__((address_space(4))) KernArgs *kargs_t;
__device__ __forceinline__ kargs_t fresh(kargs_t p) {
  asm volatile("" : "+s"(p));
  return p;
}

// One tile's words in LDS. `gen_*` carry the workgroup's iteration number + 1 of the tile the words belong to, so
// nothing has to be cleared between uses except the two counters, which the service wavefront resets while nobody
// looks (before it hands the slot's new ticket out).
struct TileSlot {
  uint32_t tile;        // the ticket (service -> workers)
  uint32_t gen_ticket;  // iteration + 1 once `tile` is valid
  uint32_t arrived;     // workers that have reported their slice's total
  uint32_t flushed;     // workers that are through with the tile (its slot may be recycled at kWorkers)
  uint32_t gen_base;    // iteration + 1 once `base` and `wbase` are valid
  uint32_t pad;
  uint64_t base;              // first output position of the tile
  uint64_t wsum[kWorkers];    // ids of each slice (workers -> service)
  uint64_t wbase[kWorkers];   // ids of the tile's earlier slices (service -> workers)
};

__device__ __forceinline__ uint32_t lds_load(const uint32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store(uint32_t *p, uint32_t x) {
  __hip_atomic_store(p, x, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// (uniform per wavefront: every lane polls the same word)
__device__ __forceinline__ void lds_wait_eq(const uint32_t *p, uint32_t x) {
  // (polls cost issue slots the other wavefronts of the SIMD could use: a wait of a few microseconds is not polled
  // every sixty nanoseconds)
  while (lds_load(p) != x) __builtin_amdgcn_s_sleep(BIVX_WSLEEP);
}

// `n` ids from a wavefront's stage (16-byte aligned) to consecutive output slots: four per lane and instruction (the
// output address is only 4-byte aligned, which gfx950's unaligned access mode allows), then the last n % 4 one by one.
__device__ __forceinline__ void stage_to_output(const uint32_t *stage, uint32_t *out, uint32_t n, uint32_t lane) {
  typedef uint32_t u32x4_a16 __attribute__((ext_vector_type(4)));
  const uint32_t n4 = n & ~3u;
  for (uint32_t i = lane * 4u; i < n4; i += kWave * 4u)
    __builtin_nontemporal_store(*reinterpret_cast<const u32x4_a16 *>(stage + i), reinterpret_cast<u32x4_a4 *>(out + i));
  if (n4 + lane < n) stream_store(out + n4 + lane, stage[n4 + lane]);
}

// A query's candidate window, worked out ahead of the counting (bucket directory probe): slots [a, b), the coordinate of
// its first cell, and what kind it is.
struct Win {
  uint32_t a, b, base;
  uint32_t fl;  // 1: the query reaches the segment.s cells (the window is non-empty if also b > a); 2: packed records decodable
};

// Hit mask of a window shorter than 32 slots, read by its own lane (queries in arbitrary order: the windows of a
// wavefront are scattered over the index). The first sixteen slots — two chunks, up to eight 16-byte loads — leave
// together, so a window that straddles two chunks costs one round trip, not two; a load is issued only by the lanes whose
// window reaches its pair (the texture-address unit's time goes by lane, 1.3 lane-loads per cycle and CU measured:
// tools/ub_gather.hip — clamped duplicates are not free). Evaluation as in slab_mask32, in ascending slot order; the
// ids of the first KEEP hits go to the lane's keep slots as they are found. Windows of 17..31 slots (a
// wavefront-uniform branch, rare) take a second round.
// `wm16`: the window's own bits among the sixteen slots starting at the even slot 2 * p0 — what a predicated-off load
// left in its registers is evaluated like the rest and masked off. The ids of the hits then go to the lane's keep
// slots in slot order without a branch: every slot's id is stored at the running position, which advances only past a
// hit (so a later store overwrites what a non-hit left), and stops at the last keep slot — which therefore holds
// garbage once KEEP or more hits were found: the caller takes KEEP - 1 ids from the slots in that case.
template <uint32_t KEEP>
__device__ __forceinline__ uint32_t eval16(const char *rb, uint32_t p0, uint32_t plast, uint32_t wm16, uint32_t base,
                                           uint32_t qh, uint32_t ql, uint32_t *keep, uint32_t &kpos) {
  uint4 r[8];
#pragma unroll
  for (uint32_t j = 0; j < 8; ++j)
#ifdef BIVX_EXP_MAXLOADS  // experiment (wrong results): at most this many record loads per lane
    if (p0 + j <= plast && j < BIVX_EXP_MAXLOADS) r[j] = *reinterpret_cast<const uint4 *>(rb + ((p0 + j) << 4));
#else
    if (p0 + j <= plast) r[j] = *reinterpret_cast<const uint4 *>(rb + ((p0 + j) << 4));
#endif
  uint32_t m = 0;
#pragma unroll
  for (uint32_t j = 0; j < 8; ++j) {
    const uint32_t rr[2] = {r[j].x, r[j].z};
#pragma unroll
    for (uint32_t k = 0; k < 2; ++k) {
      const uint32_t rl = (rr[k] - base) & 0xFFFFu;
      m = shift_in_le_ge(m, rl, qh, rl + (rr[k] >> 16), ql);
    }
  }
  m = (__brev(m) >> 16) & wm16;
#pragma unroll
  for (uint32_t j = 0; j < 8; ++j) {
    const uint32_t ii[2] = {r[j].y, r[j].w};
#pragma unroll
    for (uint32_t k = 0; k < 2; ++k) {
      keep[kpos * kWave] = ii[k];  // (slot k of lane l is word k * 64 + l of the wavefront's region: no bank conflict)
      kpos = min(kpos + ((m >> (2 * j + k)) & 1u), KEEP - 1u);
    }
  }
  return m;
}

template <uint32_t KEEP>
__device__ __forceinline__ uint32_t lanes_mask32(const uint2 *rec, const Win &w, bool nonempty, uint32_t lo, uint32_t hi,
                                                 uint32_t *keep) {
  const uint32_t al = w.a & ~1u;
  const uint32_t b = nonempty ? w.b : al;  // (an empty window: no lane-load, no hit)
  const uint32_t p0 = al >> 1, plast = ((b + 1u) >> 1) - 1u;  // p0 + j <= plast <=> slot al + 2j < b
  // (the index holds at most 2^28 records when this kernel is chosen: byte offsets fit 32 bits)
  const char *rb = reinterpret_cast<const char *>(rec);
  const uint32_t qh = hi - w.base, ql = lo > w.base ? lo - w.base : 0u;
  const uint32_t wm = ((1u << (b - al)) - 1u) & ~(w.a - al);  // bits [a - al, b - al); a - al is 0 or 1
  uint32_t kpos = 0;
  uint32_t m = 0;
  if (b > al) m = eval16<KEEP>(rb, p0, plast, wm & 0xFFFFu, w.base, qh, ql, keep, kpos);
  if (__any(b > al + 16u)) {
    if (b > al + 16u) m |= eval16<KEEP>(rb, p0 + 8u, plast, wm >> 16, w.base, qh, ql, keep, kpos) << 16;
  }
  return m;
}

// The same windows fetched by GROUPS of eight lanes (the default; -DBIVX_NO_COOP keeps lanes_mask32). What bounds the
// counting of scattered windows is the number of separate requests the vector memory pipeline has to make, not their
// bytes: with up to eight predicated 16-byte loads per lane a slice is ~350 requests of 16 bytes, and the kernel's time
// falls 0.323 -> 0.200 ms (config 3) when every lane makes at most ONE of them (wrong results, measured); one workgroup
// per CU runs as fast as two. Here the eight lanes of a group (lanes 8g .. 8g+7) fetch ONE window together in every
// round — lane p its p-th pair of records, 128 consecutive bytes per group, 8 windows per load instruction — and the
// rounds k = 0 .. 7 go through the group's own queries (round k: the window of lane 8g + k): ~12 requests per
// instruction instead of 64. The query's parameters travel to the group by ds_swizzle (the LDS crossbar, no memory):
// three words — first pair | pairs << 27 | q.low bit 16 << 31; the window's base coordinate; q.low | q.high << 16 relative
// to it (clamped to 17 / 16 bits, which the comparisons cannot tell from the full values). Both records of a lane are
// evaluated as in eval16; the two ballots carry, in byte g, the hits of group g's even / odd slots: the round's owner
// keeps its two bytes, and every lane of the group ranks its own hits among them and puts their ids into the OWNER's
// keep slots (the first KEEP hits). A group does not know where inside its first and last pair the window begins and
// ends (a is odd, b - al is odd): it reports RAW hits, and the owner — which does — trims the mask and skips what a
// false first hit put into its keep slot 0. Returns the hit mask (bit j <-> slot al + j); `kinfo` = keep slots to skip
// | usable kept ids << 1.
__device__ __forceinline__ uint32_t spread8(uint32_t x) {  // bit i -> bit 2i
  x = (x | x << 4) & 0x0F0Fu;
  x = (x | x << 2) & 0x3333u;
  return (x | x << 1) & 0x5555u;
}

template <uint32_t KEEP>
__device__ __forceinline__ uint32_t coop_mask32(const uint2 *rec, uint32_t *keep_of_wave, const Win &w, bool nonempty,
                                                uint32_t lo, uint32_t hi, uint32_t lane, uint32_t &kinfo) {
  const uint32_t al = w.a & ~1u;
  const uint32_t n = nonempty ? w.b - al : 0u;  // slots from al to the window's end, < 32
  const uint32_t npairs = (n + 1u) >> 1;
  uint32_t qh = hi - w.base, ql = lo > w.base ? lo - w.base : 0u;
  qh = qh > 0xFFFFu ? 0xFFFFu : qh;      // (a record's low is 16 bits, low + length 17)
  ql = ql > 0x1FFFFu ? 0x1FFFFu : ql;
  const uint32_t w2 = (ql & 0xFFFFu) | qh << 16;
  const char *rb = reinterpret_cast<const char *>(rec);
  const uint32_t p = lane & 7u, gsh = (lane >> 3) * 8u, below = (1u << p) - 1u;
  uint32_t *const kbase = keep_of_wave + (lane & 0x38u);  // + rank * 64 + k: slot `rank` of the round's owner
  uint32_t rawA = 0, rawB = 0;  // the owner's bytes of its own round

// A round in two halves, so that the loads of several rounds can be in flight together (BIVX_COOP_DEPTH, default 2: the
// round trips of a slice's eight rounds are what a worker's iteration mostly waits for — the kernel is bound by that
// latency chain, not by instruction issue: profiles/r04_query_kernel_experiments.txt): LOAD fetches the owner's words
// and issues the group's load, EVAL evaluates what arrived.
#define BIVX_COOP_LOAD(k, W0)                                                                                          \
  const uint32_t s0_##k = (uint32_t)__builtin_amdgcn_ds_swizzle((int)(W0), 0x18 | ((k) << 5));                        \
  const uint32_t s1_##k = (uint32_t)__builtin_amdgcn_ds_swizzle((int)w.base, 0x18 | ((k) << 5));                     \
  const uint32_t s2_##k = (uint32_t)__builtin_amdgcn_ds_swizzle((int)w2, 0x18 | ((k) << 5));                         \
  /* (the lanes whose pair lies in the window, as a lane mask straight from the comparison; what a lane that loads      \
     nothing has in its registers is evaluated like the rest and masked off by it) */                                 \
  const uint32_t np_##k = (s0_##k >> 27) & 15u;                                                                       \
  const uint64_t live_##k = __builtin_amdgcn_uicmp(p, np_##k, 36);                                                    \
  const uint32_t sql_##k = (s2_##k & 0xFFFFu) | (s0_##k >> 31) << 16;                                                 \
  uint4 r_##k;                                                                                                        \
  asm volatile("" : "=v"(r_##k.x), "=v"(r_##k.y), "=v"(r_##k.z), "=v"(r_##k.w));                                      \
  if (p < np_##k) r_##k = *reinterpret_cast<const uint4 *>(rb + (((s0_##k & 0x7FFFFFFu) + p) << 4));

#define BIVX_COOP_EVAL(k, KEEPIDS, OUTA, OUTB)                                                                        \
  {                                                                                                                   \
    const uint32_t sqh = s2_##k >> 16;                                                                                \
    const uint32_t la = (r_##k.x - s1_##k) & 0xFFFFu, lb = (r_##k.z - s1_##k) & 0xFFFFu;                              \
    const uint64_t hA = __builtin_amdgcn_uicmp(la, sqh, 37) & __builtin_amdgcn_uicmp(la + (r_##k.x >> 16), sql_##k, 35) & live_##k; \
    const uint64_t hB = __builtin_amdgcn_uicmp(lb, sqh, 37) & __builtin_amdgcn_uicmp(lb + (r_##k.z >> 16), sql_##k, 35) & live_##k; \
    const uint32_t bA = (uint32_t)(hA >> gsh) & 0xFFu, bB = (uint32_t)(hB >> gsh) & 0xFFu;                            \
    if (p == (k)) {                                                                                                   \
      OUTA = bA;                                                                                                      \
      OUTB = bB;                                                                                                      \
    }                                                                                                                 \
    if (KEEPIDS) {                                                                                                    \
      const uint32_t ra = (uint32_t)__popc(bA & below) + (uint32_t)__popc(bB & below), rbk = ra + ((bA >> p) & 1u);   \
      if (((bA >> p) & 1u) != 0u && ra < KEEP) kbase[ra * kWave + (k)] = r_##k.y;                                      \
      if (((bB >> p) & 1u) != 0u && rbk < KEEP) kbase[rbk * kWave + (k)] = r_##k.w;                                    \
    }                                                                                                                 \
  }

// the eight rounds with BIVX_COOP_DEPTH loads in flight
#if BIVX_COOP_DEPTH == 1
#define BIVX_COOP_ROUNDS(W0, KEEPIDS, OUTA, OUTB)                                                            \
  { BIVX_COOP_LOAD(0, W0) BIVX_COOP_EVAL(0, KEEPIDS, OUTA, OUTB) } { BIVX_COOP_LOAD(1, W0) BIVX_COOP_EVAL(1, KEEPIDS, OUTA, OUTB) } \
  { BIVX_COOP_LOAD(2, W0) BIVX_COOP_EVAL(2, KEEPIDS, OUTA, OUTB) } { BIVX_COOP_LOAD(3, W0) BIVX_COOP_EVAL(3, KEEPIDS, OUTA, OUTB) } \
  { BIVX_COOP_LOAD(4, W0) BIVX_COOP_EVAL(4, KEEPIDS, OUTA, OUTB) } { BIVX_COOP_LOAD(5, W0) BIVX_COOP_EVAL(5, KEEPIDS, OUTA, OUTB) } \
  { BIVX_COOP_LOAD(6, W0) BIVX_COOP_EVAL(6, KEEPIDS, OUTA, OUTB) } { BIVX_COOP_LOAD(7, W0) BIVX_COOP_EVAL(7, KEEPIDS, OUTA, OUTB) }
#elif BIVX_COOP_DEPTH == 2
#define BIVX_COOP_ROUNDS(W0, KEEPIDS, OUTA, OUTB)                                                            \
  {                                                                                                          \
    BIVX_COOP_LOAD(0, W0) BIVX_COOP_LOAD(1, W0) BIVX_COOP_EVAL(0, KEEPIDS, OUTA, OUTB)                       \
    BIVX_COOP_LOAD(2, W0) BIVX_COOP_EVAL(1, KEEPIDS, OUTA, OUTB) BIVX_COOP_LOAD(3, W0) BIVX_COOP_EVAL(2, KEEPIDS, OUTA, OUTB) \
    BIVX_COOP_LOAD(4, W0) BIVX_COOP_EVAL(3, KEEPIDS, OUTA, OUTB) BIVX_COOP_LOAD(5, W0) BIVX_COOP_EVAL(4, KEEPIDS, OUTA, OUTB) \
    BIVX_COOP_LOAD(6, W0) BIVX_COOP_EVAL(5, KEEPIDS, OUTA, OUTB) BIVX_COOP_LOAD(7, W0) BIVX_COOP_EVAL(6, KEEPIDS, OUTA, OUTB) \
    BIVX_COOP_EVAL(7, KEEPIDS, OUTA, OUTB)                                                                   \
  }
#elif BIVX_COOP_DEPTH == 3
#define BIVX_COOP_ROUNDS(W0, KEEPIDS, OUTA, OUTB)                                                            \
  {                                                                                                          \
    BIVX_COOP_LOAD(0, W0) BIVX_COOP_LOAD(1, W0) BIVX_COOP_LOAD(2, W0) BIVX_COOP_EVAL(0, KEEPIDS, OUTA, OUTB) \
    BIVX_COOP_LOAD(3, W0) BIVX_COOP_EVAL(1, KEEPIDS, OUTA, OUTB) BIVX_COOP_LOAD(4, W0) BIVX_COOP_EVAL(2, KEEPIDS, OUTA, OUTB) \
    BIVX_COOP_LOAD(5, W0) BIVX_COOP_EVAL(3, KEEPIDS, OUTA, OUTB) BIVX_COOP_LOAD(6, W0) BIVX_COOP_EVAL(4, KEEPIDS, OUTA, OUTB) \
    BIVX_COOP_LOAD(7, W0) BIVX_COOP_EVAL(5, KEEPIDS, OUTA, OUTB) BIVX_COOP_EVAL(6, KEEPIDS, OUTA, OUTB)      \
    BIVX_COOP_EVAL(7, KEEPIDS, OUTA, OUTB)                                                                   \
  }
#else  // 4
#define BIVX_COOP_ROUNDS(W0, KEEPIDS, OUTA, OUTB)                                                            \
  {                                                                                                          \
    BIVX_COOP_LOAD(0, W0) BIVX_COOP_LOAD(1, W0) BIVX_COOP_LOAD(2, W0) BIVX_COOP_LOAD(3, W0)                  \
    BIVX_COOP_EVAL(0, KEEPIDS, OUTA, OUTB) BIVX_COOP_LOAD(4, W0) BIVX_COOP_EVAL(1, KEEPIDS, OUTA, OUTB) BIVX_COOP_LOAD(5, W0) \
    BIVX_COOP_EVAL(2, KEEPIDS, OUTA, OUTB) BIVX_COOP_LOAD(6, W0) BIVX_COOP_EVAL(3, KEEPIDS, OUTA, OUTB) BIVX_COOP_LOAD(7, W0) \
    BIVX_COOP_EVAL(4, KEEPIDS, OUTA, OUTB) BIVX_COOP_EVAL(5, KEEPIDS, OUTA, OUTB)                            \
    BIVX_COOP_EVAL(6, KEEPIDS, OUTA, OUTB) BIVX_COOP_EVAL(7, KEEPIDS, OUTA, OUTB)                            \
  }
#endif

  {
    const uint32_t w0 = (al >> 1) | (npairs < 8u ? npairs : 8u) << 27 | (ql >> 16) << 31;
    constexpr bool kk = (BIVX_EXP & 4) == 0;
    BIVX_COOP_ROUNDS(w0, kk, rawA, rawB)
  }
  const uint32_t raw = spread8(rawA) | spread8(rawB) << 1;
  const uint32_t wm = ((1u << n) - 1u) & ~(w.a - al);  // the window's own bits: [a - al, n); a - al is 0 or 1
  uint32_t m = raw & wm & 0xFFFFu;
  {
    // kept ids: raw hits 0 .. min(raw hits, KEEP) - 1; the true ones among them: not a false first (slot al, a odd), not
    // a false last (slot al + n, n odd)
    const uint32_t rawcnt = (uint32_t)__popc(raw);
    const uint32_t ff = raw & (w.a - al) & 1u;
    const uint32_t fl = n < 16u ? (raw >> n) & 1u : 0u;
    const uint32_t upto = rawcnt - fl < KEEP ? rawcnt - fl : KEEP;
    kinfo = ff | (upto - ff) << 1;
  }
  if (__any(n > 16u)) {  // windows of 17 .. 31 slots (rare): their second sixteen the same way, ids are re-read later
    uint32_t rawC = 0, rawD = 0;
    const uint32_t np2 = npairs > 8u ? npairs - 8u : 0u;
    const uint32_t w0 = ((al >> 1) + 8u) | np2 << 27 | (ql >> 16) << 31;
    BIVX_COOP_ROUNDS(w0, false, rawC, rawD)
    m |= ((spread8(rawC) | spread8(rawD) << 1) << 16) & wm;
  }
#undef BIVX_COOP_ROUNDS
#undef BIVX_COOP_LOAD
#undef BIVX_COOP_EVAL
  return m;
}


// ---- scattered windows of at most sixteen slots: the ids placed where they will leave from (round 4) ------------------
// The rounds of coop_mask32 with the ROUND'S OWNERS NEIGHBOURS IN THE OUTPUT: in round k group g (lanes 8g .. 8g+7) fetches
// the window of lane 8k + g (its words come through ds_bpermute), so the hits of a round are the lists of queries 8k .. 8k+7
// back to back — in lane order, a lane's first record before its second, which IS slot order — and a hit's place in the
// slice is (hits of the rounds before) + (hits of the lower lanes in this round): two v_mbcnt pairs on the round's two hit
// masks. The lanes store their ids there at once (EXEC = the hit mask, no branch), into `out` — the wavefront's keep
// region, which a straight copy moves to the output stage when that is free — and lane 8g leaves the group's first place
// in `pos[8k + g]`: a query's offset in the slice; its count is the next query's offset minus its own. No per-owner hit
// mask, no keep-slot ranking, no prefix sum, no per-lane layout loop: coop_mask32 + scan + layout were ~330 of the ~690
// vector instructions of a slice, this is ~190 (profiles/r04_query_kernel_experiments.txt).
// A group must know which of its sixteen records are the window's: the owner's word carries the number of slots n from
// the even slot al on (0 .. 16) and whether a is odd — record A of lane p is the window's if (a odd ? p > 0 : true) and
// 2p < n, record B if 2p + 1 < n. Wavefronts with a longer window, or a query whose low end lies more than 65535 above
// its window's base (low > high queries), take coop_mask32.
__device__ __forceinline__ uint32_t lds_addr_of(const void *p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
// LDS store by the lanes of `mask` only (the mask becomes EXEC for the one instruction)
__device__ __forceinline__ void lds_store_lanes(uint64_t mask, uint32_t byte_addr, uint32_t data) {
  uint64_t saved;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tds_write_b32 %2, %3\n\ts_mov_b64 exec, %0"
               : "=&s"(saved)
               : "s"(mask), "v"(byte_addr), "v"(data)
               : "memory");
}

constexpr uint32_t kPlacePos = 432;  // words of the keep region where the 65 offsets of a slice live (behind its <= kPStage ids)
static_assert(kPStage <= kPlacePos && kPlacePos + 65u <= kWave * kPKeep, "the keep region holds a slice's ids and offsets");

// Returns the slice's number of ids; `cnt` / `loff` receive the lane's own count and offset. `out`: the wavefront's keep
// region (ids of the first kPStage places are stored, none if !store_ids).
__device__ __forceinline__ uint32_t coop_place16(const uint2 *rec, uint32_t *out, const Win &w, bool nonempty, uint32_t lo,
                                                 uint32_t hi, uint32_t lane, bool store_ids, uint32_t &cnt, uint32_t &loff) {
  const uint32_t al = w.a & ~1u;
  const uint32_t n = nonempty ? w.b - al : 0u;  // <= 16
  uint32_t qh = hi - w.base, ql = lo > w.base ? lo - w.base : 0u;  // (ql <= 0xFFFF: the caller has looked)
  qh = qh > 0xFFFFu ? 0xFFFFu : qh;
  const uint32_t w0 = (al >> 1) | n << 26 | (nonempty ? (w.a & 1u) : 0u) << 31;
  const uint32_t w2 = ql | qh << 16;
  const char *rb = reinterpret_cast<const char *>(rec);
  const uint32_t p = lane & 7u, p2 = 2u * p, p2p1 = 2u * p + 1u;
  const uint32_t from = (lane >> 3) << 2;                 // ds_bpermute address of lane (lane >> 3); + 32 k: lane 8k + (lane >> 3)
  const uint32_t out_addr = lds_addr_of(out);
  const uint32_t pos_addr = out_addr + (kPlacePos << 2) + from;   // + 32 k: word 8k + (lane >> 3)
  constexpr uint64_t kP0 = 0x0101010101010101ull;         // the groups' first lanes
  uint32_t wpos = 0;                                      // hits of the rounds so far (wavefront-uniform)

#define BIVX_PLACE_LOAD(k)                                                                                            \
  const uint32_t s0_##k = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from + 32u * (k)), (int)w0);                   \
  const uint32_t s1_##k = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from + 32u * (k)), (int)w.base);              \
  const uint32_t s2_##k = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from + 32u * (k)), (int)w2);                  \
  const uint32_t n_##k = (s0_##k >> 26) & 31u;                                                                        \
  const uint64_t inA_##k = __builtin_amdgcn_uicmp(p2, n_##k, 36), inB_##k = __builtin_amdgcn_uicmp(p2p1, n_##k, 36);  \
  /* (a odd: the group's first record is the slot before the window) */                                               \
  const uint64_t liveA_##k = inA_##k & ~(__builtin_amdgcn_uicmp(s0_##k, 0x7FFFFFFFu, 34) & kP0);                      \
  uint4 r_##k;                                                                                                        \
  asm volatile("" : "=v"(r_##k.x), "=v"(r_##k.y), "=v"(r_##k.z), "=v"(r_##k.w));                                      \
  if (p2 < n_##k) r_##k = *reinterpret_cast<const uint4 *>(rb + (((s0_##k & 0x3FFFFFFu) + p) << 4));

#define BIVX_PLACE_EVAL(k)                                                                                            \
  {                                                                                                                   \
    const uint32_t sqh = s2_##k >> 16, sql = s2_##k & 0xFFFFu;                                                        \
    const uint32_t la = (r_##k.x - s1_##k) & 0xFFFFu, lb = (r_##k.z - s1_##k) & 0xFFFFu;                              \
    const uint64_t hA = __builtin_amdgcn_uicmp(la, sqh, 37) & __builtin_amdgcn_uicmp(la + (r_##k.x >> 16), sql, 35) & liveA_##k; \
    const uint64_t hB = __builtin_amdgcn_uicmp(lb, sqh, 37) & __builtin_amdgcn_uicmp(lb + (r_##k.z >> 16), sql, 35) & inB_##k;   \
    /* hits of the lower lanes (both records of each), then the lane's own first record */                            \
    uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(hA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hA, 0u));   \
    before = __builtin_amdgcn_mbcnt_hi((uint32_t)(hB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hB, before));        \
    const uint32_t atA = out_addr + ((wpos + before) << 2);                                                           \
    const uint32_t nh = (uint32_t)__builtin_popcountll(hA) + (uint32_t)__builtin_popcountll(hB);                       \
    lds_store_lanes(kP0, pos_addr + 32u * (k), wpos + before);                                                        \
    if (store_ids && wpos + nh <= kPStage) {  /* (wavefront-uniform; a slice beyond the stage is not staged at all) */ \
      uint32_t stepA;                                                                                                 \
      asm("v_cndmask_b32_e64 %0, 0, 4, %1" : "=v"(stepA) : "s"(hA));  /* 4 where the lane's first record is a hit */    \
      const uint32_t atB = atA + stepA;                                                                               \
      lds_store_lanes(hA, atA, r_##k.y);                                                                              \
      lds_store_lanes(hB, atB, r_##k.w);                                                                              \
    }                                                                                                                 \
    wpos += nh;                                                                                                       \
  }

  {  // (BIVX_PLACE_DEPTH rounds' loads in flight: 2 by default)
#if BIVX_PLACE_DEPTH == 4
    BIVX_PLACE_LOAD(0) BIVX_PLACE_LOAD(1) BIVX_PLACE_LOAD(2) BIVX_PLACE_LOAD(3) BIVX_PLACE_EVAL(0)
    BIVX_PLACE_LOAD(4) BIVX_PLACE_EVAL(1) BIVX_PLACE_LOAD(5) BIVX_PLACE_EVAL(2)
    BIVX_PLACE_LOAD(6) BIVX_PLACE_EVAL(3) BIVX_PLACE_LOAD(7) BIVX_PLACE_EVAL(4)
    BIVX_PLACE_EVAL(5) BIVX_PLACE_EVAL(6) BIVX_PLACE_EVAL(7)
#elif BIVX_PLACE_DEPTH == 3
    BIVX_PLACE_LOAD(0) BIVX_PLACE_LOAD(1) BIVX_PLACE_LOAD(2) BIVX_PLACE_EVAL(0)
    BIVX_PLACE_LOAD(3) BIVX_PLACE_EVAL(1) BIVX_PLACE_LOAD(4) BIVX_PLACE_EVAL(2)
    BIVX_PLACE_LOAD(5) BIVX_PLACE_EVAL(3) BIVX_PLACE_LOAD(6) BIVX_PLACE_EVAL(4)
    BIVX_PLACE_LOAD(7) BIVX_PLACE_EVAL(5) BIVX_PLACE_EVAL(6) BIVX_PLACE_EVAL(7)
#else
    BIVX_PLACE_LOAD(0) BIVX_PLACE_LOAD(1) BIVX_PLACE_EVAL(0)
    BIVX_PLACE_LOAD(2) BIVX_PLACE_EVAL(1) BIVX_PLACE_LOAD(3) BIVX_PLACE_EVAL(2)
    BIVX_PLACE_LOAD(4) BIVX_PLACE_EVAL(3) BIVX_PLACE_LOAD(5) BIVX_PLACE_EVAL(4)
    BIVX_PLACE_LOAD(6) BIVX_PLACE_EVAL(5) BIVX_PLACE_LOAD(7) BIVX_PLACE_EVAL(6)
    BIVX_PLACE_EVAL(7)
#endif
  }
#undef BIVX_PLACE_LOAD
#undef BIVX_PLACE_EVAL
  if (lane == 0) out[kPlacePos + kWave] = wpos;
  wave_sync_lds();
  const uint32_t mine = out[kPlacePos + lane], next = out[kPlacePos + lane + 1u];
  loff = mine;
  cnt = next - mine;
  return wpos;
}

#define A(f) (fresh(ka)->a.f)

// The service wavefront of a pipelined workgroup (k_query_pipe and k_query_pipe_dense): tickets, published totals, the
// sweeps over the earlier tiles, the workers' bases — everything that talks to other workgroups.
template <int NW>  // worker wavefronts of the workgroup (the service wavefront is wavefront NW)
__device__ __forceinline__ void pipe_service_wave(kargs_t ka, TileSlot *s_slot, int lane) {
  // ================================ the service wavefront ================================================
  PrefixCtx pc;
  {
    kargs_t p = fresh(ka);
    pc.group = p->a.ws + kWsStatus;
    pc.status = pc.group + kFMaxGroups;
    pc.ntiles = p->a.ntiles;
    const uint32_t wl = ((uint32_t)p->a.flags >> kFlagWaitShift) & 0xFFu;
    pc.wait_ticks = 1ull << (wl ? wl : kWaitLog2Default);
    pc.err = p->v.err;
  }
  // A polled state machine over the workgroup's tiles, in the order they were drawn (iteration numbers):
  //   drawn:     tickets handed out so far. Ticket k+1 is drawn when the first worker has reported on tile k (its
  //              queries can then be fetched while the stragglers finish), never earlier: a ticket held by a busy
  //              workgroup delays that tile's published total, and every later tile's sweep waits for it.
  //   published: tiles whose total is out (needs the fifteen totals);   grouped: ... whose group word, if it is the
  //   64th tile of a group, is out (needs the group's earlier tiles);   swept: ... whose first output position the
  //   workers have (needs every earlier tile of the launch). None of the three waits inside: a sweep that finds a
  //   word missing is simply tried again on the next pass, so one slow predecessor never holds a ticket back.
  uint32_t drawn = 0, published = 0, grouped = 0, swept = 0;
  uint32_t last_tile = 0;   // ticket of the newest drawn tile
  bool drawing = true;      // no ticket beyond the batch yet
  uint64_t stuck_since = 0; // when the oldest unswept tile's sweep was first found blocked
  const bool flat = pc.ntiles <= kFlatTiles;
  for (;;) {
    bool progress = false;
    // ---- a ticket ----
    if (drawing && drawn - swept < kRing) {
      bool want = drawn == 0;
#ifdef BIVX_TICKET_EARLY   // experiment: a counting phase ahead (when that many workers have begun the tile before; NW = all)
      if (!want) want = drawn == 1 || lds_load(&s_slot[(drawn - 2) % kRing].arrived) >= (uint32_t)(BIVX_TICKET_EARLY);
#else
      if (!want) want = lds_load(&s_slot[(drawn - 1) % kRing].arrived) != 0;
#endif
      TileSlot &sl = s_slot[drawn % kRing];
      // (the slot's last user was iteration drawn - kRing: swept, but every worker must also be through with it)
      if (want && (drawn < kRing || lds_load(&sl.flushed) == (uint32_t)NW)) {
        // The ticket: ONE counter for the launch. (Round 4 tried eight sharded counters for the launch's first gridDim.x
        // tiles — the burst of 512 first draws queues on one word for up to 6 us, profiles/r04_small_batch_timeline.txt —
        // with the later draws stealing what a late workgroup had not drawn. It is unsafe: a workgroup's tiles must come in
        // ASCENDING order, because the flush of its tile k waits for a sweep over every smaller tile, and a smaller tile
        // drawn later sits behind k in the same workgroup's ring — the workers that would count it are the ones waiting.
        // With part of a launch resident (two processes on one card) exactly that happened: timeouts. One counter hands
        // every workgroup ascending tiles by construction.)
        uint32_t tile = 0;
        if (lane == 0) {
          tile = atomicAdd(reinterpret_cast<unsigned int *>(A(ws) + kWsTicket), 1u);
          sl.arrived = 0;
          sl.flushed = 0;
          sl.tile = tile;
          lds_store(&sl.gen_ticket, drawn + 1);
        }
        last_tile = __builtin_amdgcn_readfirstlane(tile);
        // Every workgroup draws exactly one ticket beyond the batch, so none can reach ntiles + gridDim.x unless
        // the counter was not zero when the launch began (a launch that died half-way, a caller workspace that was
        // not cleared): then tiles were skipped and nothing this launch wrote can be trusted. Say so.
        if (last_tile >= pc.ntiles + gridDim.x && lane == 0) raise_error(pc.err, kErrWorkspace);
        if (last_tile >= pc.ntiles) drawing = false;  // the workers see it, flush what is pending and leave
        ++drawn;
        progress = true;
      }
    }
    const uint32_t real = drawing ? drawn : drawn - 1;  // tiles of the batch among the drawn ones
    // ---- publish ----
    if (published < real) {
      TileSlot &ps = s_slot[published % kRing];
      if (lds_load(&ps.arrived) == (uint32_t)NW) {
        const uint64_t mine = lane < NW ? ps.wsum[lane] : 0ull;
        uint64_t incl = mine;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
          const uint64_t o = __shfl_up((unsigned long long)incl, d, kWave);
          if (lane >= d) incl += o;
        }
        if (lane < NW) ps.wbase[lane] = incl - mine;
        const uint64_t total = __shfl((unsigned long long)incl, NW - 1, kWave);
        if (lane == 0) {
          ps.base = total;  // parked here until the sweep replaces it with the tile's first position
          st_status(&pc.status[ps.tile], kStValid | total);
        }
        ++published;
        progress = true;
      }
    }
    // ---- the group word of a group's 64th tile ----
    if (grouped < published) {
      TileSlot &gs = s_slot[grouped % kRing];
      const uint32_t t = gs.tile;
      if (flat || (t & 63u) != 63u) {
        ++grouped;
        progress = true;
      } else {
        uint64_t in_group = 0;
        if (try_sum_in_group(pc, t, lane, in_group)) {
          if (lane == 0) st_status(&pc.group[t >> 6], kStValid | (in_group + gs.base));
          ++grouped;
          progress = true;
        }
      }
    }
    // ---- sweep ----
    if (swept < grouped) {
      TileSlot &ss = s_slot[swept % kRing];
      uint64_t sum = 0;
      bool ok = try_tiles_before(pc, ss.tile, lane, sum);
      if (!ok) {
        // bounded by wall time: a device that stopped making progress fails the call instead of hanging it
        const uint64_t now = __builtin_amdgcn_s_memrealtime();
        if (stuck_since == 0) stuck_since = now;
        else if (now - stuck_since > pc.wait_ticks) {
          if (lane == 0) raise_error(pc.err, kErrTimeout);
          ok = true;  // go on with a wrong prefix; no entry point lets the call pass as success
        }
      }
      if (ok) {
        if (lane == 0) {
          const size_t qb = A(q_begin);
          ss.base = sum + (qb ? A(offsets)[qb] : 0ull);
        }
        lds_store(&ss.gen_base, swept + 1);  // (every lane stores the same word, after lane 0's base)
        ++swept;
        stuck_since = 0;
        progress = true;
      }
    }
    if (!drawing && swept == drawn - 1) break;
    if (!progress) __builtin_amdgcn_s_sleep(BIVX_SSLEEP);
  }
  // self-cleaning workspace: every service wavefront bumps `done` when its sweeps are over and its last ticket is
  // drawn; the one that sees gridDim.x - 1 knows nobody touches the words any more and zeroes them for the next
  // launch (the list of slices for k_fill_slices stays: that kernel clears its count).
  if (A(flags) & kFlagSelfClean) {
    uint64_t *ws = A(ws);
    uint32_t last = 0;
    if (lane == 0) last = atomicAdd(reinterpret_cast<unsigned int *>(ws + kWsDone), 1u) == gridDim.x - 1 ? 1u : 0u;
    if (__shfl(last, 0, kWave)) {
      for (uint32_t t = (uint32_t)lane; t < pc.ntiles; t += kWave) {
        pc.status[t] = 0;
        if (t < (pc.ntiles + kWave - 1) / kWave) pc.group[t] = 0;
      }
      if (lane == 0) {
        ws[kWsTicket] = 0;
        ws[kWsDone] = 0;
      }
    }
  }
}

// S: every query's ids leave in ascending order (ordered by their lane while they sit in the stage).
template <bool S>
__global__ __launch_bounds__(kPThreads, 8) void k_query_pipe(IndexView v_in, PipeArgs a_in) {
  (void)v_in;
  (void)a_in;
  kargs_t ka = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
  if (threadIdx.x == 0) { WGSTAMP(0); }
  __shared__ SegDesc s_seg[kLdsSegs];
  __shared__ uint2 s_cs[kLdsChroms];
  __shared__ TileSlot s_slot[kRing];
  __shared__ uint4 s_keep[kWorkers * kWave * (kPKeep / 4)];   // keep slots / slabs of the slice being counted
  __shared__ __attribute__((aligned(16))) uint32_t s_stage[kWorkers][kDefer][kPStage];     // ids of the pending slices, laid out as in the output
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;

  {  // descriptors and the ring, once per workgroup
    kargs_t p = fresh(ka);
    const uint4 *src = reinterpret_cast<const uint4 *>(p->v.seg);
    uint4 *dst = reinterpret_cast<uint4 *>(s_seg);
    const uint32_t nseg2 = p->v.nseg * 2, nchrom = p->v.nchrom;
    for (uint32_t t = threadIdx.x; t < nseg2; t += kPThreads) dst[t] = src[t];
    const uint2 *rng = p->v.chrom_rng;
    for (uint32_t t = threadIdx.x; t < nchrom; t += kPThreads) s_cs[t] = rng[t];
    if (threadIdx.x < kRing) {
      s_slot[threadIdx.x].gen_ticket = 0;
      s_slot[threadIdx.x].gen_base = 0;
      s_slot[threadIdx.x].arrived = 0;
      s_slot[threadIdx.x].flushed = 0;
    }
  }
  __syncthreads();  // the only barrier of the kernel
  if (threadIdx.x == 0) { WGSTAMP(1); }
  const SegDesc *const segs = s_seg;
  const uint2 *const cs = s_cs;

  if (wave == kWorkers) {
    pipe_service_wave<kWorkers>(ka, s_slot, lane);
    WGSTAMP(5);
    return;
  }

  // ==================================== a worker wavefront ===================================================
  // The thread's index, taken afresh wherever an address is derived from it: the loop below is long, and values the
  // compiler computes once in front of it (lane * 16 as a 64-bit offset, the lane's LDS addresses) would otherwise be
  // held — or spilled — across all of its phases.
  auto tid = [] {
    uint32_t t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
  };
  // keep slot k of a lane: kept_slots()[k * kWave] (the wavefront's 512 words, transposed; the same words are its slab)
  auto kept_slots = [&] {
    const uint32_t t = tid();
    return reinterpret_cast<uint32_t *>(&s_keep[(t & ~(uint32_t)(kWave - 1)) * (kPKeep / 4)]) + (t & (kWave - 1));
  };
  auto slab_of_wave = [&] { return &s_keep[(tid() & ~(uint32_t)(kWave - 1)) * (kPKeep / 4)]; };

  auto query_of = [&](uint32_t t) {
    kargs_t p = fresh(ka);
    const size_t q = p->a.q_begin + (size_t)t * kPTile + threadIdx.x;
    IndexView w;  // load_query reads nchrom only (no filter)
    w.nchrom = p->v.nchrom;
    w.flt_qaux = nullptr;
    return load_query<false>(w, cs, p->a.qchrom, p->a.qlow, p->a.qhigh, q, t < p->a.ntiles && q < p->a.q_end);
  };

  // The bucket-directory probe of a query (seg_window's arithmetic, query_device.h), issued and NOT waited for: a and b
  // are the two directory words on their way; `fl` says whether the query touches the segment's cells at all (1) and
  // whether the window's packed records are decodable (2). The window is non-empty if fl & 1 and b > a.
  auto window_of = [&](const Query &q) {
    Win w{0u, 0u, 0u, 0u};
    if (q.nseg) {
      const SegDesc d = load_seg(segs + q.s0);
      const uint32_t x = q.lo > d.maxlen ? q.lo - d.maxlen : 0u;
      if (!(q.hi < d.base || x > d.last || q.hi < x)) {
        const uint32_t sh = d.shift & 31u;
        const uint32_t ca = x <= d.base ? 0u : (x - d.base) >> sh;
        const uint32_t cb = q.hi >= d.last ? d.ncell : ((q.hi - d.base) >> sh) + 1u;
        // (32-bit byte offsets: the directory has fewer entries than the index has records, at most 2^28 here)
        const char *t = reinterpret_cast<const char *>(fresh(ka)->v.table);
        w.a = *reinterpret_cast<const uint32_t *>(t + ((d.table_off + ca) << 2));
        w.b = *reinterpret_cast<const uint32_t *>(t + ((d.table_off + cb) << 2));
        w.base = d.base + (ca << sh);
        w.fl = 1u | ((d.shift & kSegPacked) != 0 && ((uint64_t)(cb - ca) << sh) <= 65536ull ? 2u : 0u);
      }
    }
    return w;
  };

  // The two pending slices: counted and reported, their ids in a stage each (or, unstaged, their counts), output
  // deferred by TWO iterations — the sweep for a tile has two counting phases to find its predecessors published
  // (with one, a workgroup that falls behind stalls everybody after it in ticket order: measured 4.5 us median,
  // 14 us p90 of waiting per iteration). `a` is the younger.
  struct Pending {
    bool have, staged;
    uint32_t tile, wtotal, state;
  };
  Pending pd_[kDefer];
#pragma unroll
  for (uint32_t k = 0; k < kDefer; ++k) pd_[k] = Pending{false, false, 0u, 0u, 0u};
  // writes the offsets of a pending slice and streams its ids; `j` is the iteration it was counted in
  auto flush = [&](const Pending &pd, uint32_t j) {
    TileSlot &os = s_slot[j % kRing];
    uint32_t *const stage = s_stage[wave][j % kDefer];
    lds_wait_eq(&os.gen_base, j + 1);
    const uint64_t wpos0 = os.base + os.wbase[wave];
    kargs_t p = fresh(ka);
    const size_t q_end = p->a.q_end;
    const size_t q = p->a.q_begin + (size_t)pd.tile * kPTile + threadIdx.x;
    uint64_t *off = p->a.offsets;
    if (pd.staged) {
      if (q < q_end) {
        stream_store(off + q, wpos0 + (pd.state >> 16));
        uint32_t *counts = p->a.counts;
        if (counts) stream_store(counts + q, pd.state & 0xFFFFu);
        if (q == q_end - 1) *(counts ? p->a.total_out : off + q_end) = wpos0 + (pd.state >> 16) + (pd.state & 0xFFFFu);
      }
      const uint64_t cap = p->a.cap;
      if (cap != 0) {
        // (the slice's first output position is the same in every lane: a scalar base, 32-bit lane offsets)
        const uint64_t wp = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)wpos0) |
                            (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(wpos0 >> 32)) << 32;
        uint32_t *const out = p->a.hits + wp;
        const uint32_t lim = cap > wp ? (cap - wp < pd.wtotal ? (uint32_t)(cap - wp) : pd.wtotal) : 0u;
        if (!(BIVX_EXP & 2)) stage_to_output(stage, out, lim, tid() & (kWave - 1));
      }
    } else {
      // an unstaged slice kept (list offset, count) per lane in its stage; its ids are k_fill_slices' business
      const uint64_t lp = (uint64_t)stage[3 * lane] | (uint64_t)stage[3 * lane + 1] << 32;
      const uint32_t c = stage[3 * lane + 2];
      if (q < q_end) {
        off[q] = wpos0 + lp;
        uint32_t *counts = p->a.counts;
        if (counts) counts[q] = c;
        if (q == q_end - 1) *(counts ? p->a.total_out : off + q_end) = wpos0 + lp + c;
      }
      if (lane == 0 && p->a.cap != 0) {  // (no buffer, no ids to fill in: k_fill_slices is not even launched then)
        uint64_t *ws = p->a.ws;
        uint32_t *todo = reinterpret_cast<uint32_t *>(ws + kWsList);
        todo[atomicAdd(reinterpret_cast<unsigned int *>(ws + kWsTodo), 1u)] = pd.tile * 16u + (uint32_t)wave;
      }
    }
    wave_sync_lds();  // the stage may be refilled now
    if (lane == 0) __hip_atomic_fetch_add(&os.flushed, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  lds_wait_eq(&s_slot[0].gen_ticket, 1u);
  if (threadIdx.x == 0) { WGSTAMP(2); }
  uint32_t tile = __builtin_amdgcn_readfirstlane(s_slot[0].tile);
  // What a lane carries from one iteration to the next: its query of the coming tile and that query's directory
  // probe, both issued an iteration ahead (the queries when the ticket comes in, the probe when they have arrived —
  // behind the flush — so that its round trip runs under the id layout).
  uint32_t qlo, qhi;
  Win wn;
  {
    const Query q0 = query_of(tile);
    qlo = q0.lo;
    qhi = q0.hi;
    wn = window_of(q0);
  }
  if (threadIdx.x == 0) { WGSTAMP(3); }

  for (uint32_t it = 0;; ++it) {
    const bool live = tile < A(ntiles);
    int path = 0;                  // how the slice was counted (wavefront-uniform), see below
    uint32_t qw_a = 0;             // paths 1 and 2: the window's first slot
    uint32_t m32 = 0, lbase = 0;   // paths 1 and 2: the hit mask (bit j <-> slot (a & ~1) + j); path 1: the slab's first slot
    uint32_t kinfo = 0;            // path 2: keep slots to skip | usable kept ids << 1 (coop_mask32)
    uint32_t cnt = 0, loff = 0, wtotal = 0;
    bool staged = false, no_ids = false;
    uint64_t lpos64 = 0;
    if (live) {
      PSTAMP(tile, 0);
      // ---- count the slice of the new tile ------------------------------------------------------------------
      // Three ways, chosen per wavefront from the windows themselves: (1) all windows short and neighbours in the
      // index (a position-sorted batch): their union goes through the LDS slab once; (2) all short, scattered (or only
      // partly neighbours): every lane reads its own; (0) long windows, unpacked segments: the general enumeration
      // counts, and the slice's ids are k_fill_slices' business (it is listed like one that overflows its stage).
      const bool nonempty = (wn.fl & 1u) != 0 && wn.b > wn.a;
      const uint32_t al = wn.a & ~1u;
      qw_a = wn.a;
      path = 0;
      if (!__any(nonempty && !((wn.fl & 2u) != 0 && wn.b - al < 32u))) {
        lbase = wave_min(nonempty ? al : 0xFFFFFFFFu);
        const bool in_slab = nonempty && wn.b - lbase <= kPKeep * (kWave / 2);
        const uint32_t nin = (uint32_t)__popcll(__ballot(in_slab)), nne = (uint32_t)__popcll(__ballot(nonempty));
        path = nin >= kSlabMinLanes && nin == nne ? 1 : 2;
        // scattered windows of at most sixteen slots (and no query that begins 64 K above its window's base: low > high
        // queries can): the ids go straight to their places (coop_place16)
        if (path == 2 && (BIVX_EXP & 8) == 0 &&
            !__any(nonempty && (wn.b - al > 16u || (qlo > wn.base && qlo - wn.base > 0xFFFFu) || (al >> 1) >= (1u << 26))))
          path = 3;
      }
      no_ids = A(cap) == 0;
      if (path == 1) {
        const uint32_t npairs = (wave_max(nonempty ? wn.b : 0u) - lbase + 1u) >> 1;  // fits the slab; rec[] carries two spare slots
        const uint4 *src = reinterpret_cast<const uint4 *>(fresh(ka)->v.rec) + (lbase >> 1);
        uint4 *const slab = slab_of_wave();
        for (uint32_t i = tid() & (kWave - 1); i < npairs; i += kWave) slab[i] = src[i];
        wave_sync_lds();
        const Window w{wn.a, wn.b, wn.base, 1u, true};
        m32 = slab_mask32(slab, lbase, w, qlo, qhi, nonempty);
        cnt = (uint32_t)__popc(m32);
      } else if (path == 3) {
        wtotal = coop_place16(fresh(ka)->v.rec, reinterpret_cast<uint32_t *>(slab_of_wave()), wn, nonempty, qlo, qhi,
                              tid() & (kWave - 1), !no_ids, cnt, loff);
      } else if (path == 2) {
#ifdef BIVX_NO_COOP
        m32 = lanes_mask32<kPKeep>(fresh(ka)->v.rec, wn, nonempty, qlo, qhi, kept_slots());
        kinfo = 0xFFFFFFFFu;
#else
        m32 = coop_mask32<kPKeep>(fresh(ka)->v.rec, reinterpret_cast<uint32_t *>(slab_of_wave()), wn, nonempty, qlo, qhi,
                                  tid() & (kWave - 1), kinfo);
#endif
        cnt = (uint32_t)__popc(m32);
      } else {
        kargs_t p = fresh(ka);
        IndexView v1;
        v1.se = p->v.se;
        v1.rec = p->v.rec;
        v1.id = p->v.id;
        v1.table = p->v.table;
        v1.seg = nullptr;
        v1.chrom_rng = nullptr;
        v1.nchrom = 0;
        v1.nseg = 0;
        v1.max_segs = 1;
        v1.flt_kind = BIVX_FILTER_NONE;
        v1.flt_dist = 0;
        v1.flt_strand = 0;
        v1.flt_qaux = nullptr;
        v1.flt_iaux = nullptr;
        v1.order_shift = p->v.order_shift;
        v1.err = nullptr;
        const Query qy = query_of(tile);  // (rare: the query is fetched again rather than carried)
        cnt = enumerate_hits<Mode::Count, false, false, kPKeep, kRows, false>(v1, segs, qy, nullptr, 0, 0, nullptr);
      }
      bool huge = false;
      if (path != 3) {  // (coop_place16 leaves the offsets and the total itself: at most sixteen ids per query there)
        const uint32_t incl = wave_scan_incl(cnt);
        loff = incl - cnt;
        // (2^22 hits in one lane would overflow the 32-bit scan: such a slice is never staged and is counted in 64 bits)
        huge = __any(cnt >= (1u << 22));
        wtotal = wave_last(incl);
      }
      staged = !huge && (no_ids ? wtotal < 65536u : path != 0 && wtotal <= kPStage);
      uint64_t wt64 = wtotal;
      lpos64 = loff;
      if (huge) {
        uint64_t i64 = cnt;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
          const uint64_t o = __shfl_up((unsigned long long)i64, d, kWave);
          if (lane >= d) i64 += o;
        }
        lpos64 = i64 - cnt;
        wt64 = __shfl((unsigned long long)i64, kWave - 1, kWave);
      }
      // report the slice's total (the service wavefront publishes the tile when all fifteen are in)
      TileSlot &sl = s_slot[it % kRing];
      if (lane == 0) {
        sl.wsum[wave] = wt64;
        __hip_atomic_fetch_add(&sl.arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      PSTAMP(tile, 1);
    }

    // ---- the next tile's ticket is out already (drawn an iteration ahead): its queries leave now -------------------
    uint32_t ntile = 0xFFFFFFFFu;
    Query nqy{0u, 0u, 0u, 0u, 0u};
    if (live) {
      TileSlot &nx = s_slot[(it + 1) % kRing];
      lds_wait_eq(&nx.gen_ticket, it + 2);
      ntile = __builtin_amdgcn_readfirstlane(nx.tile);
      nqy = query_of(ntile);
      PSTAMP(tile, 2);
    }

    // ---- the slice counted two iterations ago goes out (both pending ones when the batch is used up) -----------------
    if (pd_[kDefer - 1].have) flush(pd_[kDefer - 1], it - kDefer);
    if (!live) {
#pragma unroll
      for (uint32_t k = kDefer - 1; k-- > 0;)
        if (pd_[k].have) flush(pd_[k], it - 1 - k);
      if (threadIdx.x == 0) {
        WGSTAMP(4);
        WGSTAMP_VAL(6, (unsigned long long)it);
      }
      break;
    }
    PSTAMP(tile, 3);
    const Win nwn = window_of(nqy);  // (its two loads are consumed at the top of the next iteration)

    // ---- lay the new slice's ids out in the stage, back to back as they will sit in the output -------------------
#pragma unroll
    for (uint32_t k = kDefer - 1; k > 0; --k) pd_[k] = pd_[k - 1];
    pd_[0] = Pending{true, staged, tile, wtotal, (loff << 16) | cnt};
    uint32_t *const stage = s_stage[wave][it % kDefer];
    if (!staged) {
      stage[3 * lane] = (uint32_t)lpos64;
      stage[3 * lane + 1] = (uint32_t)(lpos64 >> 32);
      stage[3 * lane + 2] = cnt;
    } else if ((BIVX_EXP & 1) != 0) {
    } else if (!no_ids && path == 1) {  // the ids are in the wavefront's slab
      const uint2 *s2 = reinterpret_cast<const uint2 *>(slab_of_wave()) + ((qw_a & ~1u) - lbase);
      uint32_t *dst = stage + loff;
      while (m32) {
        const uint32_t j = (uint32_t)__ffs((int)m32) - 1u;
        m32 &= m32 - 1u;
        *dst++ = s2[j].y;
      }
    } else if (!no_ids && path == 3) {  // the ids lie in the keep region as they will lie in the output: a straight copy
      const uint4 *src = reinterpret_cast<const uint4 *>(slab_of_wave());
      uint4 *dst = reinterpret_cast<uint4 *>(stage);
      for (uint32_t i = tid() & (kWave - 1); i < (wtotal + 3u) >> 2; i += kWave) dst[i] = src[i];
    } else if (!no_ids && path == 2) {  // the first kPKeep ids are in the lane's keep slots; a longer list re-reads the rest
      uint32_t *dst = stage + loff;
#ifdef BIVX_NO_COOP
      const uint32_t *kslots = kept_slots();
      const uint32_t nk = cnt < kPKeep ? cnt : kPKeep - 1u;  // (the last slot is only good while it was not the limit)
#else
      const uint32_t *kslots = kept_slots() + (kinfo & 1u) * kWave;  // (a false first hit sits in slot 0)
      const uint32_t nk = cnt < (kinfo >> 1) ? cnt : kinfo >> 1;
#endif
      for (uint32_t k = 0; k < nk; ++k) {
        dst[k] = kslots[k * kWave];
        m32 &= m32 - 1u;
      }
      if (__any(m32 != 0u)) {
        const char *rb = reinterpret_cast<const char *>(fresh(ka)->v.rec);
        dst += nk;
        while (m32) {
          const uint32_t j = (uint32_t)__ffs((int)m32) - 1u;
          m32 &= m32 - 1u;
          *dst++ = *reinterpret_cast<const uint32_t *>(rb + ((((qw_a & ~1u) + j) << 3) + 4u));
        }
      }
    }
    if (S && staged && !no_ids) {
      // Every lane orders its own list where it lies (ids are distinct inside a query): up to eight ids go through a
      // sorting network in registers; a longer list is ranked into the keep slots, which are idle until the next tile
      // is counted, and copied back. Lane-local throughout — a wavefront's LDS operations execute in order, nothing to wait for.
      wave_sync_lds();  // (the slab next door may still be read by this wavefront's other lanes)
#define BIVX_CE(i, j)                      \
  {                                        \
    const uint32_t lo_ = min(x[i], x[j]);  \
    x[j] = max(x[i], x[j]);                \
    x[i] = lo_;                            \
  }
#define BIVX_SORT8(a0, a1, a2, a3, a4, a5, a6, a7) /* 19 comparators; ascending in the order the indexes are given */ \
  BIVX_CE(a0, a1) BIVX_CE(a2, a3) BIVX_CE(a4, a5) BIVX_CE(a6, a7)                                                   \
  BIVX_CE(a0, a2) BIVX_CE(a1, a3) BIVX_CE(a4, a6) BIVX_CE(a5, a7)                                                   \
  BIVX_CE(a1, a2) BIVX_CE(a5, a6) BIVX_CE(a0, a4) BIVX_CE(a3, a7)                                                   \
  BIVX_CE(a1, a5) BIVX_CE(a2, a6) BIVX_CE(a1, a4) BIVX_CE(a3, a6) BIVX_CE(a2, a4) BIVX_CE(a3, a5) BIVX_CE(a3, a4)
      if (cnt > 1 && cnt <= 8u) {
        // a sorting network on eight registers (missing ids are +inf and stay behind the list's end)
        uint32_t x[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) x[k] = k < cnt ? stage[loff + k] : 0xFFFFFFFFu;
        BIVX_SORT8(0, 1, 2, 3, 4, 5, 6, 7)
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k)
          if (k < cnt) stage[loff + k] = x[k];
      } else if (cnt > 8u && cnt <= 16u) {
        // sixteen: the first eight ascending, the last eight descending, then a bitonic merge (70 comparators; ranking
        // a list of ten in LDS costs five times the instructions)
        uint32_t x[16];
#pragma unroll
        for (uint32_t k = 0; k < 16u; ++k) x[k] = k < cnt ? stage[loff + k] : 0xFFFFFFFFu;
        BIVX_SORT8(0, 1, 2, 3, 4, 5, 6, 7)
        BIVX_SORT8(15, 14, 13, 12, 11, 10, 9, 8)
#pragma unroll
        for (uint32_t d = 8; d > 0; d >>= 1)
#pragma unroll
          for (uint32_t i = 0; i < 16u; ++i)
            if ((i & d) == 0) BIVX_CE(i, i + d)
#pragma unroll
        for (uint32_t k = 0; k < 16u; ++k)
          if (k < cnt) stage[loff + k] = x[k];
#undef BIVX_SORT8
#undef BIVX_CE
      } else if (cnt > 16u) {
        uint32_t *const tmp = reinterpret_cast<uint32_t *>(slab_of_wave());
        rank_sort_list<8>(stage, tmp, loff, cnt);
        for (uint32_t k = 0; k < cnt; ++k) stage[loff + k] = tmp[loff + k];
      }
    }
    wave_sync_lds();
    PSTAMP(tile, 4);
    tile = ntile;
    qlo = nqy.lo;
    qhi = nqy.hi;
    wn = nwn;
  }

}

// ---- the same pipeline for MANY ids per query, on position-sorted batches ----------------------------------------------
// (config 5: an index overlapped with itself in its own order, 17 ids per query — a wavefront's 64 lists are ~1100 ids
// and fit no stage.) What a pending slice keeps for two iterations is not its ids but what regenerates them: every
// lane's hit mask (64 bits), the window's first slot and the slab's bounds. When the slice's base is known the slab —
// the union of the wavefront's windows, read for counting two iterations ago and still in L2 — is fetched again
// (issued right after counting, consumed here) and the ids go from it through the stage to the output in rounds of
// kDRound. Only wavefronts whose 64 windows are neighbours, light (<= 64 slots) and packed are handled that way;
// any other slice is counted by the general enumeration and listed for k_fill_slices. A batch that is not
// position-sorted would list everything, so this kernel is launched next to k_query_fused and a device-side probe of
// the query order (k_probe_order) decides which of the two does the launch's work; the other returns at once.
constexpr uint32_t kDRound = 2 * kPStage;  // ids per round: the wavefront's whole stage

__global__ __launch_bounds__(kPThreads, 8) void k_query_pipe_dense(IndexView v_in, PipeArgs a_in) {
  (void)v_in;
  (void)a_in;
  kargs_t ka = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
  __shared__ SegDesc s_seg[kLdsSegs];
  __shared__ uint2 s_cs[kLdsChroms];
  __shared__ TileSlot s_slot[kRing];
  __shared__ uint4 s_keep[kWorkers * kWave * (kPKeep / 4)];   // the wavefronts' slabs
  __shared__ __attribute__((aligned(16))) uint32_t s_stage[kWorkers][2 * kPStage];
  {  // the order probe's verdict: this launch's sequence number if the batch is position-sorted
    kargs_t p = fresh(ka);
    const uint32_t w = __hip_atomic_load(reinterpret_cast<const uint32_t *>(p->a.ws + kWsOrder), __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
    if (w != p->a.seq && !(p->a.flags & kFlagSorted)) return;
  }
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  {
    kargs_t p = fresh(ka);
    const uint4 *src = reinterpret_cast<const uint4 *>(p->v.seg);
    uint4 *dst = reinterpret_cast<uint4 *>(s_seg);
    const uint32_t nseg2 = p->v.nseg * 2, nchrom = p->v.nchrom;
    for (uint32_t t = threadIdx.x; t < nseg2; t += kPThreads) dst[t] = src[t];
    const uint2 *rng = p->v.chrom_rng;
    for (uint32_t t = threadIdx.x; t < nchrom; t += kPThreads) s_cs[t] = rng[t];
    if (threadIdx.x < kRing) {
      s_slot[threadIdx.x].gen_ticket = 0;
      s_slot[threadIdx.x].gen_base = 0;
      s_slot[threadIdx.x].arrived = 0;
      s_slot[threadIdx.x].flushed = 0;
    }
  }
  __syncthreads();
  const SegDesc *const segs = s_seg;
  const uint2 *const cs = s_cs;
  if (wave == kWorkers) {
    pipe_service_wave<kWorkers>(ka, s_slot, lane);
    return;
  }

  auto tid = [] {
    uint32_t t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
  };
  auto slab_of_wave = [&] { return &s_keep[(tid() & ~(uint32_t)(kWave - 1)) * (kPKeep / 4)]; };
  auto query_of = [&](uint32_t t) {
    kargs_t p = fresh(ka);
    const size_t q = p->a.q_begin + (size_t)t * kPTile + threadIdx.x;
    IndexView w;
    w.nchrom = p->v.nchrom;
    w.flt_qaux = nullptr;
    return load_query<false>(w, cs, p->a.qchrom, p->a.qlow, p->a.qhigh, q, t < p->a.ntiles && q < p->a.q_end);
  };
  auto window_of = [&](const Query &q) {  // as in k_query_pipe
    Win w{0u, 0u, 0u, 0u};
    if (q.nseg) {
      const SegDesc d = load_seg(segs + q.s0);
      const uint32_t x = q.lo > d.maxlen ? q.lo - d.maxlen : 0u;
      if (!(q.hi < d.base || x > d.last || q.hi < x)) {
        const uint32_t sh = d.shift & 31u;
        const uint32_t ca = x <= d.base ? 0u : (x - d.base) >> sh;
        const uint32_t cb = q.hi >= d.last ? d.ncell : ((q.hi - d.base) >> sh) + 1u;
        const char *t = reinterpret_cast<const char *>(fresh(ka)->v.table);
        w.a = *reinterpret_cast<const uint32_t *>(t + ((d.table_off + ca) << 2));
        w.b = *reinterpret_cast<const uint32_t *>(t + ((d.table_off + cb) << 2));
        w.base = d.base + (ca << sh);
        w.fl = 1u | ((d.shift & kSegPacked) != 0 && ((uint64_t)(cb - ca) << sh) <= 65536ull ? 2u : 0u);
      }
    }
    return w;
  };

  // A pending slice: counted and reported, output deferred by two iterations. Three registers per lane:
  //   slab slice (its ids come back out of the slab): x = the lane's hit mask; st = first position within the slice
  //   << 15 | (window's first even slot - lbase) << 7 | hits — a slab slice has at most 64 x 64 ids, 256 slots;
  //   any other slice: x = first position within the slice (64 bits), st = hits; it is listed for k_fill_slices
  //   if ids are asked for.
  struct Pending {
    bool have, slab;
    uint32_t tile, wtotal, lbase, npairs;  // wavefront-uniform
    uint64_t x;
    uint32_t st;
  };
  Pending pa{false, false, 0u, 0u, 0u, 0u, 0ull, 0u}, pb = pa;

  // the first slab row of the slice about to be flushed, fetched ahead (one 16-byte row per lane covers 128 records;
  // a longer slab fetches its second row when it is needed)
  uint4 row0 = make_uint4(0u, 0u, 0u, 0u);
  auto fetch_row0 = [&](const Pending &pd) {
    const uint4 *src = reinterpret_cast<const uint4 *>(fresh(ka)->v.rec) + (pd.lbase >> 1);
    const uint32_t l = tid() & (kWave - 1);
    if (l < pd.npairs) row0 = src[l];
  };

  auto flush = [&](const Pending &pd, uint32_t j, bool have_row0) {
    TileSlot &os = s_slot[j % kRing];
    uint32_t *const stage = s_stage[wave];
    kargs_t p0 = fresh(ka);
    const uint64_t cap = p0->a.cap;
    const bool ids = pd.slab && cap != 0 && pd.wtotal != 0;
    if (ids) {
      if (!have_row0) fetch_row0(pd);
      uint4 *const slab = slab_of_wave();
      const uint32_t l = tid() & (kWave - 1);
      if (l + kWave < pd.npairs)
        slab[l + kWave] = (reinterpret_cast<const uint4 *>(p0->v.rec) + (pd.lbase >> 1))[l + kWave];
      if (l < pd.npairs) slab[l] = row0;
    }
    lds_wait_eq(&os.gen_base, j + 1);
    const uint64_t wpos0 = os.base + os.wbase[wave];
    kargs_t p = fresh(ka);
    const size_t q_end = p->a.q_end;
    const size_t q = p->a.q_begin + (size_t)pd.tile * kPTile + threadIdx.x;
    uint64_t *off = p->a.offsets;
    const uint32_t *perm = p->a.perm;
    if (pd.slab) {
      const uint32_t loff = pd.st >> 15, cnt = pd.st & 127u;
      if (q < q_end) {
        if (perm) {  // (bivx_self_overlaps_dev: count and list position go where the query's id says)
          // (ONE scattered 8-byte store per interval: the list's length above the position where it begins)
          // (a position beyond the buffer — the true total may be far beyond 2^38 — is never read: it is stored as all ones so
          // that it cannot run into the length, which the offsets are made of)
          const uint64_t pos = cap != 0 ? wpos0 + loff : 0ull;
          p->a.src_by_id[perm[q]] = (uint64_t)cnt << kSelfPosBits | (pos < kSelfPosMask ? pos : kSelfPosMask);
        } else {
          stream_store(off + q, wpos0 + loff);
          if (q == q_end - 1) off[q_end] = wpos0 + loff + cnt;
        }
      }
      if (ids) {
        wave_sync_lds();  // the slab is in place
        // (the slice's first output position is the same in every lane: a scalar base, 32-bit lane offsets)
        const uint64_t wp = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)wpos0) |
                            (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(wpos0 >> 32)) << 32;
        uint32_t *const out = p->a.hits + wp;
        const uint32_t room = cap > wp ? (cap - wp < 0xFFFFFFFFull ? (uint32_t)(cap - wp) : 0xFFFFFFFFu) : 0u;
        const uint2 *s2 = reinterpret_cast<const uint2 *>(slab_of_wave()) + ((pd.st >> 7) & 255u);
        // the mask is walked one 32-bit word at a time (find-first-set and clear-lowest are one and two instructions
        // on a word, four each on 64 bits): first the low word's hits, then the high word's
        uint32_t m0 = (uint32_t)pd.x, m1 = (uint32_t)(pd.x >> 32);
        uint32_t done = 0;  // ids this lane has laid out in earlier rounds
        const uint32_t end = loff + cnt;
        for (uint32_t r0 = 0; r0 < pd.wtotal; r0 += kDRound) {
          // a lane's ids enter the stage in order, over one or more consecutive rounds
          const uint32_t stop = end < r0 + kDRound ? end : r0 + kDRound;
          if (loff + done < stop) {
            uint32_t *dst = stage + (loff + done - r0);
            uint32_t *const dstop = stage + (stop - r0);
            while (m0 != 0u && dst < dstop) {
              const uint32_t jj = (uint32_t)__ffs((int)m0) - 1u;
              m0 &= m0 - 1u;
              *dst++ = s2[jj].y;
            }
            while (m1 != 0u && dst < dstop) {  // (only once the low word is used up: otherwise dst == dstop)
              const uint32_t jj = (uint32_t)__ffs((int)m1) - 1u;
              m1 &= m1 - 1u;
              *dst++ = s2[32u + jj].y;
            }
            done = (uint32_t)(dst - stage) + r0 - loff;
          }
          wave_sync_lds();
          const uint32_t n = pd.wtotal - r0 < kDRound ? pd.wtotal - r0 : kDRound;
          const uint32_t lim = room > r0 ? (room - r0 < n ? room - r0 : n) : 0u;
          stage_to_output(stage, out + r0, lim, tid() & (kWave - 1));
          wave_sync_lds();
        }
      }
    } else {
      if (q < q_end) {
        if (perm) {
          const uint64_t pos = cap != 0 ? wpos0 + pd.x : 0ull;
          p->a.src_by_id[perm[q]] = (uint64_t)pd.st << kSelfPosBits | (pos < kSelfPosMask ? pos : kSelfPosMask);
        } else {
          off[q] = wpos0 + pd.x;
          if (q == q_end - 1) off[q_end] = wpos0 + pd.x + pd.st;
        }
      }
      if (cap != 0 && lane == 0) {
        uint64_t *ws = p->a.ws;
        uint32_t *todo = reinterpret_cast<uint32_t *>(ws + kWsList);
        todo[atomicAdd(reinterpret_cast<unsigned int *>(ws + kWsTodo), 1u)] = pd.tile * 16u + (uint32_t)wave;
      }
    }
    if (lane == 0) __hip_atomic_fetch_add(&os.flushed, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  lds_wait_eq(&s_slot[0].gen_ticket, 1u);
  uint32_t tile = __builtin_amdgcn_readfirstlane(s_slot[0].tile);
  uint32_t qlo, qhi;
  Win wn;
  {
    const Query q0 = query_of(tile);
    qlo = q0.lo;
    qhi = q0.hi;
    wn = window_of(q0);
  }

  for (uint32_t it = 0;; ++it) {
    const bool live = tile < A(ntiles);
    bool in_slab_path = false;
    uint32_t lbase = 0, npairs = 0, al = 0, cnt = 0, loff = 0, wtotal = 0;
    uint64_t mask = 0, lpos64 = 0;
    if (live) {
      const bool nonempty = (wn.fl & 1u) != 0 && wn.b > wn.a;
      al = wn.a & ~1u;
      // every window of the wavefront light, packed and inside one slab?
      lbase = wave_min(nonempty ? al : 0xFFFFFFFFu);
      const bool fits = !nonempty || ((wn.fl & 2u) != 0 && wn.b - al <= kLight && wn.b - lbase <= kPKeep * (kWave / 2));
      in_slab_path = !__any(!fits);
      if (in_slab_path) {
        if (__any(nonempty)) {
          npairs = (wave_max(nonempty ? wn.b : 0u) - lbase + 1u) >> 1;  // fits the slab; rec[] carries two spare slots
        } else {
          lbase = 0;
        }
        const uint4 *src = reinterpret_cast<const uint4 *>(fresh(ka)->v.rec) + (lbase >> 1);
        uint4 *const slab = slab_of_wave();
        for (uint32_t i = tid() & (kWave - 1); i < npairs; i += kWave) slab[i] = src[i];
        wave_sync_lds();
        const Window w{wn.a, wn.b, wn.base, 1u, true};
        const bool short32 = !__any(nonempty && wn.b - al >= 32u);
        mask = short32 ? (uint64_t)slab_mask32(slab, lbase, w, qlo, qhi, nonempty)
                       : slab_mask64(slab, lbase, w, qlo, qhi, nonempty);
        cnt = (uint32_t)__popcll(mask);
      } else {
        kargs_t p = fresh(ka);
        IndexView v1;
        v1.se = p->v.se;
        v1.rec = p->v.rec;
        v1.id = p->v.id;
        v1.table = p->v.table;
        v1.seg = nullptr;
        v1.chrom_rng = nullptr;
        v1.nchrom = 0;
        v1.nseg = 0;
        v1.max_segs = 1;
        v1.nslots = 0;
        v1.flt_kind = BIVX_FILTER_NONE;
        v1.flt_dist = 0;
        v1.flt_strand = 0;
        v1.flt_qaux = nullptr;
        v1.flt_iaux = nullptr;
        v1.order_shift = p->v.order_shift;
        v1.err = nullptr;
        const Query qy = query_of(tile);
        cnt = enumerate_hits<Mode::Count, false, false, kPKeep, kRows, false>(v1, segs, qy, nullptr, 0, 0, nullptr);
      }
      const uint32_t incl = wave_scan_incl(cnt);
      loff = incl - cnt;
      const bool huge = __any(cnt >= (1u << 22));
      wtotal = wave_last(incl);
      uint64_t wt64 = wtotal;
      lpos64 = loff;
      if (huge) {
        uint64_t i64 = cnt;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
          const uint64_t o = __shfl_up((unsigned long long)i64, d, kWave);
          if (lane >= d) i64 += o;
        }
        lpos64 = i64 - cnt;
        wt64 = __shfl((unsigned long long)i64, kWave - 1, kWave);
      }
      TileSlot &sl = s_slot[it % kRing];
      if (lane == 0) {
        sl.wsum[wave] = wt64;
        __hip_atomic_fetch_add(&sl.arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    // the slab of the slice that goes out in this iteration is on its way while the ticket is waited for
    const bool want_row = pb.have && pb.slab && pb.wtotal != 0 && A(cap) != 0;
    if (want_row) fetch_row0(pb);

    uint32_t ntile = 0xFFFFFFFFu;
    Query nqy{0u, 0u, 0u, 0u, 0u};
    if (live) {
      TileSlot &nx = s_slot[(it + 1) % kRing];
      lds_wait_eq(&nx.gen_ticket, it + 2);
      ntile = __builtin_amdgcn_readfirstlane(nx.tile);
      nqy = query_of(ntile);
    }
    if (pb.have) flush(pb, it - 2, want_row);
    if (!live) {
      if (pa.have) flush(pa, it - 1, false);
      break;
    }
    const Win nwn = window_of(nqy);
    pb = pa;
    pa = Pending{true, in_slab_path, tile, wtotal, lbase, npairs, in_slab_path ? mask : lpos64,
                 in_slab_path ? (loff << 15) | ((al - lbase) & 255u) << 7 | cnt : cnt};
    tile = ntile;
    qlo = nqy.lo;
    qhi = nqy.hi;
    wn = nwn;
  }
}

// ---- the same pipeline for EVERYTHING ELSE: several segments per query, fused filters, many ids per query --------------
// (SV-like length spectra: two or three length classes per chromosome, 8-50 ids per query, windows of 20-150 slots, the
// longest class not packed. k_query_fused<MS> walks windows beyond 64 slots with the whole wavefront, one query after the
// other, once to count and once to fill: ~450 vector instructions per query, and that — not memory — is its time:
// 0.74 ms for 1 M queries of tools/skewed_bench.py 1e5.)
// Here every lane walks its own windows, sixteen slots per trip (eight 16-byte loads in flight), TWICE: once to count —
// nothing but the count is kept — and again, when the slice's place in the output is known two iterations later, to put
// the ids into the wavefront's stage exactly as they sit in the output, from where they leave in whole lines. A pending
// slice is two registers per lane and no LDS, so the whole LDS of a CU (one workgroup of 1 024 threads, 128 registers)
// goes to the fifteen stages: 2 432 ids each — a round holds as many consecutive lanes' lists as fit. The second walk
// finds its lines in L2 (the index of such a workload is a few MB).
// A slice with a window beyond kMsLaneMax slots, or a list longer than a stage, is counted by the general enumeration
// and listed for k_fill_slices, as in k_query_pipe.
#ifndef BIVX_MS_WAVES
#define BIVX_MS_WAVES 4  // wavefronts per SIMD the kernel is built for: 4 = 128 registers
#endif
// Workgroups of 512 threads — seven workers and the service wavefront, tiles of 448 queries — two per CU (80 KB of LDS
// each): 1 024 threads as in k_query_pipe, one workgroup per CU, ran 0.142 / 0.539 ms on tools/skewed_bench.py 1e4 / 1e5
// where this runs 0.136 / 0.493; 256 threads 0.159 / 0.505.
constexpr int kMsThreads = 512;
constexpr int kMsWorkers = kMsThreads / kWave - 1;
constexpr uint32_t kMsTile = kMsWorkers * kWave;
#ifndef BIVX_MS_DEFER
#define BIVX_MS_DEFER 2   // iterations a counted slice waits for its place in the output (1: experiment — one pending slice, twice the keep slots)
#endif
constexpr uint32_t kMsDefer = BIVX_MS_DEFER;
constexpr uint32_t kMsKeepN = 32 / kMsDefer;                     // ids a lane keeps while its slice is pending — on average: see ms_keep_word
#ifdef BIVX_MS_NOKEEP
constexpr uint32_t kMsKeepWords = 0;
#else
constexpr uint32_t kMsKeepWords = kMsKeepN * kWave;              // ... a wavefront's keep slots, one pending slice
#endif
#ifndef BIVX_MS_UNFIT_RUN
#define BIVX_MS_UNFIT_RUN 2
#endif
#ifndef BIVX_MS_BUF
#define BIVX_MS_BUF 448
#endif
constexpr uint32_t kMsBuf = BIVX_MS_BUF;      // ids a wavefront lines up per round on their way out
static_assert(kMsBuf >= 6 * kWave, "the walk's table of window words lives in the buffer");
constexpr uint32_t kMsStage = kMsDefer * kMsKeepWords + kMsBuf;  // a wavefront's LDS: the pending slices' keep slots + the buffer
#ifndef BIVX_MS_OWN
#define BIVX_MS_OWN 48
#endif
#ifndef BIVX_MS_OWN_TRIP
#define BIVX_MS_OWN_TRIP 16
#endif
constexpr uint32_t kMsOwnTrip = BIVX_MS_OWN_TRIP;                // slots of its own window a lane evaluates per trip
constexpr uint32_t kMsOwnSlots = BIVX_MS_OWN;                    // a segment whose windows all end within that many slots of their
                                                                 // first even slot: every lane evaluates its own (0: never)
constexpr uint32_t kMsGroupMax = 1024;                           // slots of the longest window a group of lanes walks

// Every lane's hits over all its segments, in index order (segment, slot), by GROUPS of eight lanes, as coop_mask32 does it
// for the short windows of k_query_pipe: in step j of a round the group fetches sixteen slots — one 128-byte line of
// records — of the window of its lane j, lane p the p-th pair; the eight steps' loads leave together, so a round is ONE
// memory round trip for 64 windows, and a window of n slots takes n / 16 rounds. (A lane walking its own window makes
// eight 16-byte requests per line, and those requests were the kernel's time: 0.23 ms to count tools/skewed_bench.py
// 1e5; a group walking one window after the other waits 8 x per segment for memory: 0.22 ms.) The owner's window and
// query travel to its group by ds_swizzle. A step's two ballots carry, in byte g, the hits of group g's even and odd
// slots; the owner adds their number up; EMIT: every lane ranks its own hits among the group's and stores their ids at
// the owner's place in the stage (`lpos`: where the lane's list begins there). Packed and plain segments take the same
// loads (8 bytes per slot from rec[] or se[]) and the same comparison, relative to the window's first cell with the
// record's 16 + 16 bits or absolute. The directory probe of the next segment is on its way while a segment is walked.
// `too_long` (counting only): some window is beyond kMsGroupMax slots, the count is meaningless.
template <int J>
__device__ __forceinline__ uint32_t of_group_lane(uint32_t x) {  // the value of lane J of this lane's group of eight
  return (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, 0x18 | (J << 5));
}

// The keep slots of a wavefront are 64 x kMsKeepN words, and two neighbouring lanes SHARE their 2 x kMsKeepN: the even lane
// fills them from the bottom, the odd lane from the top, so a slice is kept whenever no PAIR has more than 32 ids — with 7.6 ids
// per query 18 % of the slices have a lane beyond 16, 1 % a pair beyond 32, and every slice that is not kept is walked twice.
// (Where the two lists would meet, one overwrites the other: such a slice is not kept, nobody reads its slots.)
// Pair slot t of the pair (2p, 2p + 1) is word (t / 2) * 64 + 2p + t % 2.
__device__ __forceinline__ uint32_t ms_keep_word(uint32_t i, uint32_t lane) {
  const uint32_t t = (lane & 1u) ? 2u * kMsKeepN - 1u - i : i;
  return ((t >> 1) << 6) | (lane & ~1u) | (t & 1u);
}

// MODE: kMsCount — the counts alone; kMsKeep — the counts, and every lane's first ids go to its pair's keep slots (`stage`:
// the wavefront's 64 x kMsKeepN words, ms_keep_word; at most 2 x kMsKeepN per lane are written); kMsEmit — the ids go to
// stage[lpos ..) (the output itself; positions from `limit` on are not written).
constexpr int kMsCount = 0, kMsKeep = 1, kMsEmit = 2;
#ifdef BIVX_EXP_NOSTORE  // experiment (wrong results): the second walk without its stores
#define BIVX_EXP_STORE(c) ((c) && limit == 0x12345u)
#else
#define BIVX_EXP_STORE(c) (c)
#endif
template <int MODE, bool F>
__device__ __forceinline__ uint32_t group_scan(kargs_t ka, const SegDesc *segs, const Query &q, bool active, uint32_t *stage,
                                               uint32_t lpos, uint32_t limit, uint32_t lane, uint32_t *tab, bool &too_long) {
  constexpr bool EMIT = MODE != kMsCount || F;  // ids are wanted (a filter may ask for them)
  constexpr bool STORE = MODE != kMsCount;
  too_long = false;
  uint32_t acc = 0;
  const uint32_t nseg = active ? q.nseg : 0u;
  const uint32_t p = lane & 7u, gsh = lane & 0x38u, below = (1u << p) - 1u;
  IndexView fv;  // what filter_accept reads
  if (F) {
    kargs_t pp = fresh(ka);
    fv.flt_kind = pp->v.flt_kind;
    fv.flt_dist = pp->v.flt_dist;
    fv.flt_strand = pp->v.flt_strand;
    fv.flt_iaux = pp->v.flt_iaux;
  }
  struct Probe {
    uint32_t a, b, cell0;
    bool packed;
  };
  auto probe = [&](uint32_t k) {  // (seg_window's arithmetic, query_device.h; the two loads are not waited for here)
    Probe w{0u, 0u, 0u, false};
    if (k < nseg) {
      const SegDesc d = load_seg(segs + q.s0 + k);
      const uint32_t x = q.lo > d.maxlen ? q.lo - d.maxlen : 0u;
      if (!(q.hi < d.base || x > d.last || q.hi < x)) {
        const uint32_t sh = d.shift & 31u;
        const uint32_t ca = x <= d.base ? 0u : (x - d.base) >> sh;
        const uint32_t cb = q.hi >= d.last ? d.ncell : ((q.hi - d.base) >> sh) + 1u;
        const uint32_t *t = fresh(ka)->v.table + d.table_off;
        w.a = t[ca];
        w.b = t[cb];
        w.cell0 = d.base + (ca << sh);
        w.packed = (d.shift & kSegPacked) != 0 && ((uint64_t)(cb - ca) << sh) <= 65536ull;
      }
    }
    return w;
  };
  // The owners' words sit in a table in LDS (`tab`: 384 words of the wavefront's own), written once per segment: a step
  // reads its owner's entry at an address that depends on nothing but the lane — eight steps' reads leave back to back,
  // where eight broadcasts through ds_swizzle were eight waits in a row.
  //   tabA[l] = (slots | the window begins at the odd slot << 30 | packed << 31,  first even slot)
  //   tabB[l] = (coordinate the comparison is relative to, q.high, q.low relative to it, -)
  uint2 *const tabA = reinterpret_cast<uint2 *>(tab);
  uint4 *const tabB = reinterpret_cast<uint4 *>(tab + 2 * kWave);
  const char *se_b;
  const char *id_b;
  uint32_t rec_delta;
  {
    kargs_t pp = fresh(ka);
    se_b = reinterpret_cast<const char *>(pp->v.se);
    id_b = reinterpret_cast<const char *>(pp->v.id);
    rec_delta = pp->a.rec_delta;
  }
  Probe nx = probe(0u);
  for (uint32_t k = 0; __any(k < nseg); ++k) {
    const Probe w = nx;
    nx = probe(k + 1u);
    const uint32_t al = w.a & ~1u;
    const uint32_t n = w.b > w.a ? w.b - al : 0u;  // slots from the even slot al to the window's end
    if (MODE != kMsEmit && __any(n > kMsGroupMax)) {
      too_long = true;
      return 0u;
    }
    if (kMsOwnSlots != 0 && STORE && !__any(n > kMsOwnSlots)) {
      // Every window of this segment is short (the short classes of an SV-like spectrum: a handful of slots each, 9 and 15 on
      // average in tools/skewed_bench.py 1e4): each lane evaluates its own, kMsOwnTrip slots per trip — that many / 2 16-byte
      // loads in flight —, no table, no ballots, its hits in slot order straight to its list. (Eight lanes per window make a
      // round of eight steps of sixteen slots, ~45 vector instructions a step, whatever the windows hold.) Only when ids are
      // stored: what it saves is their ranking among the group's hits and the owner's place travelling through ds_swizzle —
      // a walk that only counts is faster in groups. tools/skewed_bench.py 1e4 / 1e5 / 1e6, single pass, by kMsOwnSlots:
      // 0: 0.139 / 0.495 / 2.29 ms   24: 0.128 / 0.497 / 2.32   48: 0.124 / 0.498 / 2.33   64: 0.126 / 0.568 / 2.75   128: 0.126 / 0.907 / 3.08
      const uint32_t sub = w.packed ? w.cell0 : 0u;
      const uint32_t qh = q.hi - sub, ql = q.lo > sub ? q.lo - sub : 0u;
      const uint32_t smsk = w.packed ? 0xFFFFu : 0xFFFFFFFFu;
      for (uint32_t c0 = 0; __any(c0 < n); c0 += kMsOwnTrip) {
        const uint32_t off = ((al + c0) >> 1) << 4;
        uint4 r[kMsOwnTrip / 2];
        uint2 ip[kMsOwnTrip / 2];
#pragma unroll
        for (uint32_t j = 0; j < kMsOwnTrip / 2; ++j) {
          asm volatile("" : "=v"(r[j].x), "=v"(r[j].y), "=v"(r[j].z), "=v"(r[j].w));  // (no value: masked below)
          asm volatile("" : "=v"(ip[j].x), "=v"(ip[j].y));
          if (c0 + 2u * j < n) {
            r[j] = *reinterpret_cast<const uint4 *>(se_b + (off + 16u * j + (w.packed ? rec_delta : 0u)));
            if (EMIT && !w.packed) ip[j] = *reinterpret_cast<const uint2 *>(id_b + ((off + 16u * j) >> 1));
          }
        }
#pragma unroll
        for (uint32_t j = 0; j < kMsOwnTrip / 2; ++j) {
          const uint32_t s2 = c0 + 2u * j;
          const uint32_t la = (r[j].x - sub) & smsk, lb = (r[j].z - sub) & smsk;
          const uint32_t ha = w.packed ? la + (r[j].x >> 16) : r[j].y, hb = w.packed ? lb + (r[j].z >> 16) : r[j].w;
          const uint32_t ida = w.packed ? r[j].y : ip[j].x, idb = w.packed ? r[j].w : ip[j].y;
          bool fa = s2 < n && (s2 != 0u || (w.a & 1u) == 0u) && la <= qh && ha >= ql;
          bool fb = s2 + 1u < n && lb <= qh && hb >= ql;
          if (F) {
            if (fa) fa = filter_accept(fv, q.lo, q.hi, q.aux, la + sub, ha + sub, ida);
            if (fb) fb = filter_accept(fv, q.lo, q.hi, q.aux, lb + sub, hb + sub, idb);
          }
          if (STORE) {
            const uint32_t atA = lpos + acc, atB = atA + (fa ? 1u : 0u);
            if (MODE == kMsKeep) {
              if (fa && atA < 2u * kMsKeepN) stage[ms_keep_word(atA, lane)] = ida;
              if (fb && atB < 2u * kMsKeepN) stage[ms_keep_word(atB, lane)] = idb;
            } else {
              char *const ob = reinterpret_cast<char *>(stage);
              if (BIVX_EXP_STORE(fa && atA < limit)) *reinterpret_cast<uint32_t *>(ob + (atA << 2)) = ida;
              if (BIVX_EXP_STORE(fb && atB < limit)) *reinterpret_cast<uint32_t *>(ob + (atB << 2)) = idb;
            }
          }
          acc += (fa ? 1u : 0u) + (fb ? 1u : 0u);
        }
      }
      continue;
    }
    {
      const uint32_t sub = w.packed ? w.cell0 : 0u;
      wave_sync_lds();  // (the last segment's reads are through)
      tabA[lane] = make_uint2(n | (w.a & 1u) << 30 | (w.packed ? 1u << 31 : 0u), al);
      tabB[lane] = make_uint4(sub, q.hi - sub, q.lo > sub ? q.lo - sub : 0u, 0u);
      wave_sync_lds();
    }
    for (uint32_t done = 0; __any(done < n); done += 16u) {  // a round: sixteen more slots of every window
      uint4 r[8];
      uint2 ip[8];
      uint2 was[8];  // (the eight owners' words first: their reads leave together)
#pragma unroll
      for (uint32_t u = 0; u < 8; ++u) was[u] = tabA[gsh | u];
#define BIVX_MS_LOAD(J)                                                                                        \
  {                                                                                                            \
    const uint2 wa = was[J];                                                                                   \
    const uint32_t nn = wa.x & 0x3FFFFFFFu;                                                                    \
    asm volatile("" : "=v"(r[J].x), "=v"(r[J].y), "=v"(r[J].z), "=v"(r[J].w));  /* (no value: masked below) */ \
    asm volatile("" : "=v"(ip[J].x), "=v"(ip[J].y));                                                           \
    if (done + 2u * p < nn) {                                                                                  \
      const uint32_t off = (((wa.y + done) >> 1) + p) << 4;  /* (at most 2^27 slots: 32-bit byte offsets) */    \
      r[J] = *reinterpret_cast<const uint4 *>(se_b + (off + ((int)wa.x < 0 ? rec_delta : 0u)));                \
      if (EMIT && (int)wa.x >= 0) ip[J] = *reinterpret_cast<const uint2 *>(id_b + (off >> 1));                 \
    }                                                                                                          \
  }
#define BIVX_MS_EVAL(J)                                                                                        \
  {                                                                                                            \
    const uint2 wa = tabA[gsh | (J)];                                                                          \
    const uint4 wb = tabB[gsh | (J)];                                                                          \
    const uint32_t nn = wa.x & 0x3FFFFFFFu, s = done + 2u * p;                                                 \
    const bool spk = (int)wa.x < 0;                                                                            \
    const uint32_t smsk = spk ? 0xFFFFu : 0xFFFFFFFFu;                                                         \
    const bool liveA = s < nn && (s != 0u || (wa.x & (1u << 30)) == 0u), liveB = s + 1u < nn;                  \
    const uint32_t la = (r[J].x - wb.x) & smsk, lb = (r[J].z - wb.x) & smsk;                                   \
    const uint32_t ha = spk ? la + (r[J].x >> 16) : r[J].y, hb = spk ? lb + (r[J].z >> 16) : r[J].w;           \
    const uint32_t ida = spk ? r[J].y : ip[J].x, idb = spk ? r[J].w : ip[J].y;                                 \
    uint64_t hA, hB;                                                                                           \
    if (!F) {                                                                                                  \
      hA = __builtin_amdgcn_uicmp(la, wb.y, 37) & __builtin_amdgcn_uicmp(ha, wb.z, 35) & __ballot(liveA);      \
      hB = __builtin_amdgcn_uicmp(lb, wb.y, 37) & __builtin_amdgcn_uicmp(hb, wb.z, 35) & __ballot(liveB);      \
    } else {                                                                                                   \
      const uint32_t slo = of_group_lane<J>(q.lo), shi = of_group_lane<J>(q.hi), saux = of_group_lane<J>(q.aux); \
      bool fa = liveA && la <= wb.y && ha >= wb.z, fb = liveB && lb <= wb.y && hb >= wb.z;                     \
      if (fa) fa = filter_accept(fv, slo, shi, saux, la + wb.x, ha + wb.x, ida);                               \
      if (fb) fb = filter_accept(fv, slo, shi, saux, lb + wb.x, hb + wb.x, idb);                               \
      hA = __ballot(fa);                                                                                       \
      hB = __ballot(fb);                                                                                       \
    }                                                                                                          \
    const uint32_t bA = (uint32_t)(hA >> gsh) & 0xFFu, bB = (uint32_t)(hB >> gsh) & 0xFFu;                     \
    if (STORE) {                                                                                               \
      const uint32_t spos = of_group_lane<J>(lpos + acc);                                                      \
      const uint32_t mineA = (bA >> p) & 1u, mineB = (bB >> p) & 1u;                                           \
      const uint32_t at = spos + (uint32_t)__popc(bA & below) + (uint32_t)__popc(bB & below);                  \
      if (MODE == kMsKeep) {                                                                                   \
        if (mineA && at < 2u * kMsKeepN) stage[ms_keep_word(at, gsh | (J))] = ida;                             \
        if (mineB && at + mineA < 2u * kMsKeepN) stage[ms_keep_word(at + mineA, gsh | (J))] = idb;             \
      } else { /* (a slice has fewer than 2^28 ids: 32-bit byte offsets from the slice's first output position) */ \
        char *const ob = reinterpret_cast<char *>(stage);                                                      \
        if (BIVX_EXP_STORE(mineA && at < limit)) *reinterpret_cast<uint32_t *>(ob + (at << 2)) = ida;          \
        if (BIVX_EXP_STORE(mineB && at + mineA < limit)) *reinterpret_cast<uint32_t *>(ob + ((at + mineA) << 2)) = idb; \
      }                                                                                                        \
    }                                                                                                          \
    if (p == (J)) acc += (uint32_t)__popc(bA) + (uint32_t)__popc(bB);                                          \
  }
      // (eight lines in flight; the filtered variants, which carry the query's own words too, four and four: no scratch)
      if (!F) {
        BIVX_MS_LOAD(0) BIVX_MS_LOAD(1) BIVX_MS_LOAD(2) BIVX_MS_LOAD(3)
        BIVX_MS_LOAD(4) BIVX_MS_LOAD(5) BIVX_MS_LOAD(6) BIVX_MS_LOAD(7)
        BIVX_MS_EVAL(0) BIVX_MS_EVAL(1) BIVX_MS_EVAL(2) BIVX_MS_EVAL(3)
        BIVX_MS_EVAL(4) BIVX_MS_EVAL(5) BIVX_MS_EVAL(6) BIVX_MS_EVAL(7)
      } else {
        BIVX_MS_LOAD(0) BIVX_MS_LOAD(1) BIVX_MS_LOAD(2) BIVX_MS_LOAD(3)
        BIVX_MS_EVAL(0) BIVX_MS_EVAL(1) BIVX_MS_EVAL(2) BIVX_MS_EVAL(3)
        BIVX_MS_LOAD(4) BIVX_MS_LOAD(5) BIVX_MS_LOAD(6) BIVX_MS_LOAD(7)
        BIVX_MS_EVAL(4) BIVX_MS_EVAL(5) BIVX_MS_EVAL(6) BIVX_MS_EVAL(7)
      }
#undef BIVX_MS_LOAD
#undef BIVX_MS_EVAL
    }
  }
  return acc;
}

template <bool F>
__global__ __launch_bounds__(kMsThreads, BIVX_MS_WAVES) void k_query_pipe_ms(IndexView v_in, PipeArgs a_in) {
  (void)v_in;
  (void)a_in;
  kargs_t ka = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
  __shared__ SegDesc s_seg[kLdsSegs];
  __shared__ uint2 s_cs[kLdsChroms];
  __shared__ TileSlot s_slot[kRing];
  __shared__ __attribute__((aligned(16))) uint32_t s_stage[kMsWorkers][kMsStage];
  {  // behind k_query_pipe_dense: that kernel did the launch's work if the order probe left this launch's number
    kargs_t p = fresh(ka);
    if (p->a.seq != 0 && __hip_atomic_load(reinterpret_cast<const uint32_t *>(p->a.ws + kWsOrder), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT) == p->a.seq)
      return;
  }
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  {
    kargs_t p = fresh(ka);
    const uint4 *src = reinterpret_cast<const uint4 *>(p->v.seg);
    uint4 *dst = reinterpret_cast<uint4 *>(s_seg);
    const uint32_t nseg2 = p->v.nseg * 2, nchrom = p->v.nchrom;
    for (uint32_t t = threadIdx.x; t < nseg2; t += kMsThreads) dst[t] = src[t];
    const uint2 *rng = p->v.chrom_rng;
    for (uint32_t t = threadIdx.x; t < nchrom; t += kMsThreads) s_cs[t] = rng[t];
    if (threadIdx.x < kRing) {
      s_slot[threadIdx.x].gen_ticket = 0;
      s_slot[threadIdx.x].gen_base = 0;
      s_slot[threadIdx.x].arrived = 0;
      s_slot[threadIdx.x].flushed = 0;
    }
  }
  __syncthreads();
  const SegDesc *const segs = s_seg;
  const uint2 *const cs = s_cs;
  if (wave == kMsWorkers) {
    pipe_service_wave<kMsWorkers>(ka, s_slot, lane);
    return;
  }

  auto query_of = [&](uint32_t t) {
    kargs_t p = fresh(ka);
    const size_t q = p->a.q_begin + (size_t)t * kMsTile + threadIdx.x;
    IndexView w;
    w.nchrom = p->v.nchrom;
    w.flt_qaux = p->v.flt_qaux;
    return load_query<F>(w, cs, p->a.qchrom, p->a.qlow, p->a.qhigh, q, t < p->a.ntiles && q < p->a.q_end);
  };

  // A pending slice: counted and reported, output deferred by two iterations. Per lane: where its list begins inside the
  // slice and how long it is; `kept`: every list of the slice is in its lane's keep slots (the usual case while queries
  // have a few ids each); otherwise the windows are walked again when the slice goes out.
  struct Pending {
    bool have, general, kept;
    uint32_t tile, wtotal;
    uint64_t x;
    uint32_t cnt;
  };
  Pending pa{false, false, false, 0u, 0u, 0ull, 0u}, pb = pa;

  auto flush = [&](const Pending &pd, uint32_t j) {
    TileSlot &os = s_slot[j % kRing];
    lds_wait_eq(&os.gen_base, j + 1);
    const uint64_t wpos0 = os.base + os.wbase[wave];
    kargs_t p = fresh(ka);
    const size_t q_end = p->a.q_end;
    const size_t q = p->a.q_begin + (size_t)pd.tile * kMsTile + threadIdx.x;
    const uint64_t cap = p->a.cap;
    if (q < q_end) {
      uint64_t *off = p->a.offsets;
      stream_store(off + q, wpos0 + pd.x);
      if (q == q_end - 1) off[q_end] = wpos0 + pd.x + pd.cnt;
    }
    if (cap != 0 && pd.general) {
      if (lane == 0) {
        uint64_t *ws = p->a.ws;
        uint32_t *todo = reinterpret_cast<uint32_t *>(ws + kWsList);
        todo[atomicAdd(reinterpret_cast<unsigned int *>(ws + kWsTodo), 1u)] = pd.tile * 16u + (uint32_t)wave;
      }
    } else if (cap != 0 && pd.wtotal != 0) {
      const uint32_t loff = (uint32_t)pd.x;
      const uint64_t wp = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)wpos0) |
                          (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(wpos0 >> 32)) << 32;
      uint32_t *const out = fresh(ka)->a.hits + wp;
      const uint32_t room = cap > wp ? (cap - wp < 0xFFFFFFFFull ? (uint32_t)(cap - wp) : 0xFFFFFFFFu) : 0u;
      if (pd.kept) {
        // out of the keep slots: kMsBuf ids at a time are lined up as they sit in the output and leave in whole lines
        const uint32_t *const keep = s_stage[wave] + (kMsDefer == 2 ? (j & 1u) * kMsKeepWords : 0u);
        uint32_t *const buf = s_stage[wave] + kMsDefer * kMsKeepWords;
        for (uint32_t r0 = 0; r0 < pd.wtotal; r0 += kMsBuf) {
          const uint32_t i0 = r0 > loff ? r0 - loff : 0u;                                         // the lane's ids i0 .. i1-1 are in
          const uint32_t i1 = r0 + kMsBuf < loff + pd.cnt ? (r0 + kMsBuf > loff ? r0 + kMsBuf - loff : 0u) : pd.cnt;  // this round
          for (uint32_t i = i0; i < i1; ++i) buf[loff + i - r0] = keep[ms_keep_word(i, (uint32_t)lane)];
          wave_sync_lds();
          const uint32_t nthis = pd.wtotal - r0 < kMsBuf ? pd.wtotal - r0 : kMsBuf;
          const uint32_t lim = room > r0 ? (room - r0 < nthis ? room - r0 : nthis) : 0u;
          stage_to_output(buf, out + r0, lim, (uint32_t)lane);
          wave_sync_lds();
        }
      } else {
        // some list is longer than the keep slots: the windows are walked again, every lane storing its hits straight
        // to their places — the hits of a round's sixteen slots are neighbours in the output, L2 merges the pieces of a line
        const Query qy = query_of(pd.tile);
        bool dummy;
        (void)group_scan<kMsEmit, F>(ka, segs, qy, pd.cnt != 0, out, loff, room, (uint32_t)lane,
                                     s_stage[wave] + kMsDefer * kMsKeepWords, dummy);
      }
    }
    if (lane == 0) __hip_atomic_fetch_add(&os.flushed, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  lds_wait_eq(&s_slot[0].gen_ticket, 1u);
  uint32_t tile = __builtin_amdgcn_readfirstlane(s_slot[0].tile);
  Query qy = query_of(tile);
#ifdef BIVX_MS_NOKEEP  // experiment: every slice is walked twice, no keep slots in use
  bool keep_mode = false;
#else
  bool keep_mode = true;
  uint32_t unfit_run = 0;  // slices in a row that had a list beyond the keep slots
#endif

  for (uint32_t it = 0;; ++it) {
    const bool live = tile < A(ntiles);
    // the slice counted two iterations ago goes out FIRST: its keep slots are the ones this iteration's slice fills
    if (kMsDefer == 2) {
      if (pb.have) flush(pb, it - 2);
      if (!live) {
        if (pa.have) flush(pa, it - 1);
        break;
      }
    } else {
      if (pa.have) flush(pa, it - 1);
      if (!live) break;
    }
    bool too_long;
    uint32_t cnt;
    // (ids are kept while the wavefront's last slice found room for all of them in its keep slots: with ~50 ids per query
    // every slice is walked a second time anyway and the first walk need not store anything)
    const bool no_ids = A(cap) == 0 || !keep_mode;
    if (no_ids)
      cnt = group_scan<kMsCount, F>(ka, segs, qy, true, nullptr, 0u, 0u, (uint32_t)lane, s_stage[wave] + kMsDefer * kMsKeepWords, too_long);
    else
      cnt = group_scan<kMsKeep, F>(ka, segs, qy, true, s_stage[wave] + (kMsDefer == 2 ? (it & 1u) * kMsKeepWords : 0u), 0u, 0u, (uint32_t)lane,
                                   s_stage[wave] + kMsDefer * kMsKeepWords, too_long);
    if (too_long) {
      kargs_t p = fresh(ka);
      IndexView v1;
      v1.se = p->v.se;
      v1.rec = p->v.rec;
      v1.id = p->v.id;
      v1.table = p->v.table;
      v1.seg = nullptr;
      v1.chrom_rng = nullptr;
      v1.nchrom = 0;
      v1.nseg = 0;
      v1.max_segs = p->v.max_segs;
      v1.nslots = 0;
      v1.max_cell = 0;
      v1.max_window = 0;
      v1.flt_kind = p->v.flt_kind;
      v1.flt_dist = p->v.flt_dist;
      v1.flt_strand = p->v.flt_strand;
      v1.flt_qaux = nullptr;
      v1.flt_iaux = p->v.flt_iaux;
      v1.order_shift = p->v.order_shift;
      v1.err = nullptr;
      cnt = enumerate_hits<Mode::Count, F, true, kKeep, kRows, false>(v1, segs, qy, nullptr, 0, 0, nullptr);
    }
    const uint32_t incl = wave_scan_incl(cnt);
    // (2^22 hits in one lane would overflow the 32-bit scan: such a slice is summed in 64 bits and left to k_fill_slices)
    const bool huge = __any(cnt >= (1u << 22));
    const bool general = too_long || huge;
    // (kept: no pair of neighbouring lanes has more ids than the pair's keep slots)
    const uint32_t pair_cnt = cnt + (uint32_t)__builtin_amdgcn_ds_swizzle((int)cnt, 0x041F);  // (xor 1: the neighbour's count)
    const bool fits = !general && !__any(pair_cnt > 2u * kMsKeepN);
    const bool kept = fits && !no_ids;
#ifndef BIVX_MS_NOKEEP
    // (two slices in a row with a list beyond the keep slots before the first walk stops keeping ids: with 7.6 ids per query
    // 18 % of the slices have such a list, and giving up keeping after every one of them made a third of all slices walk twice)
    if (!general) {
      unfit_run = fits ? 0u : unfit_run + 1u;
      keep_mode = unfit_run < BIVX_MS_UNFIT_RUN;
    }
#endif
    uint32_t wtotal = wave_last(incl);
    uint64_t wt64 = wtotal, lpos64 = incl - cnt;
    if (huge) {
      uint64_t i64 = cnt;
#pragma unroll
      for (int d = 1; d < kWave; d <<= 1) {
        const uint64_t o = __shfl_up((unsigned long long)i64, d, kWave);
        if (lane >= d) i64 += o;
      }
      lpos64 = i64 - cnt;
      wt64 = __shfl((unsigned long long)i64, kWave - 1, kWave);
      wtotal = 0xFFFFFFFFu;
    }
    TileSlot &sl = s_slot[it % kRing];
    if (lane == 0) {
      sl.wsum[wave] = wt64;
      __hip_atomic_fetch_add(&sl.arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // the next tile's ticket is out already (drawn an iteration ahead): its queries leave now
    TileSlot &nx = s_slot[(it + 1) % kRing];
    lds_wait_eq(&nx.gen_ticket, it + 2);
    const uint32_t ntile = __builtin_amdgcn_readfirstlane(nx.tile);
    const Query nqy = query_of(ntile);
    pb = pa;
    pa = Pending{true, general, kept, tile, wtotal, lpos64, cnt};
    tile = ntile;
    qy = nqy;
  }
}

// Is the batch in position order? 4096 neighbouring pairs, evenly spread: (chromosome, low) must not descend in more
// than 2 % of them. Leaves the launch's sequence number in ws[kWsOrder] if so, 0 otherwise.
__global__ __launch_bounds__(256) void k_probe_order(const uint32_t *__restrict__ qchrom, const uint32_t *__restrict__ qlow,
                                                      size_t q0, size_t q1, uint64_t *ws, uint32_t seq) {
  __shared__ uint32_t s_desc[4];
  const size_t n = q1 - q0;
  uint32_t desc = 0;
  for (uint32_t k = 0; k < 16; ++k) {
    const size_t i = q0 + (size_t)((unsigned __int128)(n - 1) * (threadIdx.x * 16u + k) / 4096u);
    if (i + 1 < q1) {
      const uint32_t c0 = qchrom ? qchrom[i] : 0u, c1 = qchrom ? qchrom[i + 1] : 0u;
      desc += (c0 > c1 || (c0 == c1 && qlow[i] > qlow[i + 1])) ? 1u : 0u;
    }
  }
  desc = wave_sum(desc);
  if ((threadIdx.x & (kWave - 1)) == 0) s_desc[threadIdx.x >> 6] = desc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t d = s_desc[0] + s_desc[1] + s_desc[2] + s_desc[3];
    __hip_atomic_store(reinterpret_cast<uint32_t *>(ws + kWsOrder), d * 50u <= 4096u ? seq : 0u, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  }
}
#undef A

// Fills the ids of the slices k_query_pipe listed: offsets are in place, the enumeration is the general one
// (k_query<Fill>'s, wavefront-cooperative windows included). One work item = one slice (64 queries, one wavefront);
// the grid is fixed and strides over the items, so the launch needs no host knowledge of the list; the last workgroup
// to finish clears the count.
template <bool S, bool F = false>
__global__ __launch_bounds__(kQThreads) void k_fill_slices(IndexView v, PipeArgs a) {
  __shared__ SegDesc s_seg[kLdsSegs];
  __shared__ uint2 s_cs[kLdsChroms];
  __shared__ uint32_t s_last;
  __shared__ __attribute__((aligned(16))) uint32_t s_sort[S ? kQWaves : 1][S ? kSortLds / 2 : 4];
  const uint32_t *todo = reinterpret_cast<const uint32_t *>(a.ws + kWsList);
  const uint32_t n =
      __hip_atomic_load(reinterpret_cast<const uint32_t *>(a.ws + kWsTodo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (n == 0) return;  // (nothing was listed: nobody touched the counters either)
  const SegDesc *segs;
  const uint2 *cs;
  stage_descriptors<true>(v, s_seg, s_cs, segs, cs);
  __syncthreads();
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
  for (uint32_t w = blockIdx.x * kQWaves + wave; w < n; w += gridDim.x * kQWaves) {
    const uint32_t e = todo[w];
    const size_t q = a.q_begin + (size_t)(e >> 4) * a.tile_q + (size_t)(e & 15u) * kWave + lane;
    const bool valid = q < a.q_end;
    const Query qy = load_query<F>(v, cs, a.qchrom, a.qlow, a.qhigh, q, valid);
    const uint64_t pos = !valid ? 0 : a.perm ? a.src_by_id[a.perm[q]] & kSelfPosMask : a.offsets[q];
    (void)enumerate_hits<Mode::Fill, F>(v, segs, qy, a.hits, pos, a.cap, nullptr);
    if (S) {
      // ascending ids were asked for: the slice's 64 lists are ordered here, where they were just written (asking for
      // the conditional k_sort_hits pass instead would send it over the whole batch for the sake of a few slices)
      wave_sync_mem();
      uint64_t o0 = pos, o1;
      if (valid) {
        o1 = a.offsets[q + 1];
      } else {
        o0 = o1 = a.offsets[a.q_end];
      }
      o0 = o0 < a.cap ? o0 : a.cap;
      o1 = o1 < a.cap ? o1 : a.cap;
      wave_sort_lists<kSortLds / 2, kRankBlock, false>(s_sort[wave], o0, o1, a.hits, (int)lane);
    }
  }
  // every workgroup that saw a non-empty list reports; the last one clears the list for the next call
  __syncthreads();
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
static int mode_forced() {
  const char *env = std::getenv("BIVX_PIPE");
  return env ? std::atoi(env) : 1;
}

bool pipe_eligible(const IndexView &v, size_t q, uint64_t cap, bool sort_ids, bool unordered) {
  // (begin / count output: the ordered CSR is a valid answer, and since the pipelined kernel it is the faster one —
  // config 3 0.33 against 0.35 ms; one launch only, there is no entry q_end to chain launches through)
  // Only for large batches: at 1 M queries k_query_fused<U>, whose tiles wait for nobody, takes 43 us (22 position-sorted)
  // against 48 (32) here.
  if (unordered && (sort_ids || q > (size_t)kFMaxTiles * kPTile || (q < ((size_t)4 << 20) && mode_forced() != 2))) return false;
  // BIVX_PIPE: 0 = never, 1 = when eligible (default), 2 = also for small batches (tests)
  const char *env = std::getenv("BIVX_PIPE");
  const int mode = env ? std::atoi(env) : 1;
  if (!mode || v.flt_kind != BIVX_FILTER_NONE || v.max_segs > 1 || !fits_lds(v)) return false;
  if (v.nslots > (1u << 28)) return false;  // (32-bit byte offsets into the records, lanes_mask32)
  // Positional hotspots (thousands of intervals starting inside one directory cell) make single slices take hundreds of
  // microseconds; a tile here waits for all fifteen of its slices, k_query_fused's tiles wait for nobody but their
  // predecessors' totals (tools/clustered_bench.py, zero-capacity count: 1.32 ms here, 0.78 there)
  if (v.max_cell > kMaxCellForPipe && mode != 2) return false;
  // a pipeline has to fill and drain: below about 0.7 M queries (1.4 tiles per resident workgroup) k_query_fused is
  // the faster one (0.25 M: 19.7 against 23.7 us, 0.5 M: 32 / 35, 0.75 M: 45 / 43, 1 M: 54 / 49, 1.3 M: 70 / 60)
  if (q < (size_t)768 * 1024 && mode != 2) return false;
  // few ids per query: a wavefront's 64 lists must fit its stage (the capacity is the only bound the host has)
  return cap <= (uint64_t)6 * q;
}

// Many ids per query (more than the stages hold): the regenerating form of the pipeline, for position-sorted batches
// (the kernel itself returns at once when k_probe_order found the batch unsorted; the caller launches k_query_fused
// behind it with the opposite condition).
bool pipe_dense_eligible(const IndexView &v, size_t q, uint64_t cap, bool sort_ids, bool unordered) {
  (void)sort_ids;  // (ascending ids: k_sort_hits orders the CSR afterwards, whichever kernel wrote it)
  const char *env = std::getenv("BIVX_PIPE");
  const int mode = env ? std::atoi(env) : 1;
  if (!mode || unordered || v.flt_kind != BIVX_FILTER_NONE || v.max_segs > 1 || !fits_lds(v)) return false;
  if (v.nslots > (1u << 28) || (v.max_cell > kMaxCellForPipe && mode != 2)) return false;
  if (q < (size_t)4 * 512 * kPTile && mode != 2) return false;
  return cap > (uint64_t)6 * q;
}

// What k_query_pipe leaves out — several segments per chromosome, a fused filter — in index order, canonical CSR
// (ascending ids: k_sort_hits behind it, as behind k_query_fused).
bool pipe_ms_eligible(const IndexView &v, size_t q, uint64_t cap, bool unordered) {
  const char *env = std::getenv("BIVX_PIPE");
  const int mode = env ? std::atoi(env) : 1;
  if (const char *e = std::getenv("BIVX_PIPE_MS"))  // (0: never — tests compare the two kernels)
    if (std::atoi(e) == 0) return false;
  if (!mode || unordered || !fits_lds(v)) return false;
  if (q < (size_t)768 * 1024 && mode != 2) return false;
  {  // (se[] and rec[] in one block, 32-bit byte offsets from se[] into both; ids 8 bytes per pair)
    const ptrdiff_t delta = reinterpret_cast<const char *>(v.rec) - reinterpret_cast<const char *>(v.se);
    if (v.nslots > (1u << 27) || delta < 0 || (uint64_t)delta + ((uint64_t)v.nslots + 2) * 8 > 0xFFFFFFFFull) return false;
  }
  // (positional hotspots: windows beyond kMsGroupMax slots go through the general enumeration twice here)
  if ((v.max_cell > kMsGroupMax / 4 || v.max_window > kMsGroupMax / 2) && mode != 2) return false;
  // (many ids per query on ONE length class, queries in any order: k_query_fused is the faster one — config 5 in
  // generation order 7.7 ms against 8.5 here; tests send it here with BIVX_PIPE=2)
  return v.max_segs > 1 || v.flt_kind != BIVX_FILTER_NONE || (mode == 2 && cap > (uint64_t)6 * q);
}

size_t pipe_queries_per_launch() { return (size_t)kFMaxTiles * kPTile; }
static_assert((uint64_t)kFMaxTiles * kPTile < (1ull << (64 - kSelfPosBits)),
              "bivx_self_overlaps_dev keeps a list's length (at most the launch's intervals) above kSelfPosBits of one word");
size_t pipe_ms_queries_per_launch() { return (size_t)kFMaxTiles * kMsTile; }

// Compute units of the calling thread's current device, looked up once per device and process (a launch used to ask the
// runtime every time: hipGetDevice + hipDeviceGetAttribute are microseconds on the path of a 40 us call). The environment
// knobs below stay per launch: tests switch them between calls.
static unsigned cus_of_current_device() {
  static std::atomic<unsigned> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256u;
  unsigned c = cache[dev].load(std::memory_order_relaxed);
  if (c == 0) {
    int cus = 0;
    c = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0 ? (unsigned)cus : 256u;
    cache[dev].store(c, std::memory_order_relaxed);
  }
  return c;
}

int launch_query_pipe_ms(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                         size_t q0, size_t q1, uint64_t *d_offsets, uint32_t *d_hits, uint64_t cap, uint64_t *ws,
                         int flags, uint32_t skip_seq, hipStream_t s) {
  const unsigned tiles = (unsigned)((q1 - q0 + kMsTile - 1) / kMsTile);
  unsigned wgs = 512;
  {
    wgs = 2u * cus_of_current_device();  // two workgroups of 512 threads per CU: 128 registers, 79 KB of LDS each
    if (const char *e = std::getenv("BIVX_PIPE_WGS")) {
      const long w = std::atol(e);
      if (w >= 1 && w <= 65536) wgs = (unsigned)w;
    }
  }
  // (seq: launched behind k_query_pipe_dense, which did the launch's work if the order probe left this number)
  PipeArgs a{d_qchrom, d_qlow, d_qhigh, q0, q1, d_offsets, d_hits, cap, ws, tiles, flags, skip_seq, nullptr, nullptr,
             nullptr, nullptr, nullptr, (uint32_t)(reinterpret_cast<const char *>(v.rec) - reinterpret_cast<const char *>(v.se)),
             kMsTile};
  const dim3 grid(tiles < wgs ? tiles : wgs);
  if (v.flt_kind != BIVX_FILTER_NONE) {
    hipLaunchKernelGGL(k_query_pipe_ms<true>, grid, dim3(kMsThreads), 0, s, v, a);
    if (cap != 0) hipLaunchKernelGGL((k_fill_slices<false, true>), dim3(kFillBlocks), dim3(kQThreads), 0, s, v, a);
  } else {
    hipLaunchKernelGGL(k_query_pipe_ms<false>, grid, dim3(kMsThreads), 0, s, v, a);
    if (cap != 0) hipLaunchKernelGGL((k_fill_slices<false, false>), dim3(kFillBlocks), dim3(kQThreads), 0, s, v, a);
  }
  BIVX_HIP(hipGetLastError());
  return 0;
}

int launch_query_pipe(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                      size_t q0, size_t q1, uint64_t *d_offsets, uint32_t *d_hits, uint64_t cap, uint64_t *ws,
                      int flags, uint32_t sort_seq, uint32_t *d_counts, uint64_t *d_total, hipStream_t s) {
  const unsigned tiles = (unsigned)((q1 - q0 + kPTile - 1) / kPTile);
  unsigned wgs = 512;
  {
    wgs = 2u * cus_of_current_device();  // two workgroups of 1024 threads are resident per CU (64 VGPRs, 71 KiB of LDS)
    if (const char *e = std::getenv("BIVX_PIPE_WGS")) {  // tuning / test knob
      const long w = std::atol(e);
      if (w >= 1 && w <= 65536) wgs = (unsigned)w;
    }
  }
  PipeArgs a{d_qchrom, d_qlow, d_qhigh, q0, q1, d_offsets, d_hits, cap, ws, tiles, flags, sort_seq, d_counts, d_total,
             nullptr, nullptr, nullptr, 0u, kPTile};
  if (sort_seq)
    hipLaunchKernelGGL(k_query_pipe<true>, dim3(tiles < wgs ? tiles : wgs), dim3(kPThreads), 0, s, v, a);
  else
    hipLaunchKernelGGL(k_query_pipe<false>, dim3(tiles < wgs ? tiles : wgs), dim3(kPThreads), 0, s, v, a);
  if (cap == 0) {
    // a pure count: nothing is listed, nothing to fill in (the launch, empty as it usually is, costs 4.8 us)
  } else if (sort_seq) {
    hipLaunchKernelGGL(k_fill_slices<true>, dim3(kFillBlocks), dim3(kQThreads), 0, s, v, a);
  } else {
    hipLaunchKernelGGL(k_fill_slices<false>, dim3(kFillBlocks), dim3(kQThreads), 0, s, v, a);
  }
  BIVX_HIP(hipGetLastError());
  return 0;
}

int launch_query_pipe_dense(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                            size_t q0, size_t q1, uint64_t *d_offsets, uint32_t *d_hits, uint64_t cap, uint64_t *ws,
                            int flags, uint32_t seq, hipStream_t s) {
  const unsigned tiles = (unsigned)((q1 - q0 + kPTile - 1) / kPTile);
  unsigned wgs = 512;
  {
    wgs = 2u * cus_of_current_device();
    if (const char *e = std::getenv("BIVX_PIPE_WGS")) {
      const long w = std::atol(e);
      if (w >= 1 && w <= 65536) wgs = (unsigned)w;
    }
  }
  PipeArgs a{d_qchrom, d_qlow, d_qhigh, q0, q1, d_offsets, d_hits, cap, ws, tiles, flags, seq, nullptr, nullptr,
             nullptr, nullptr, nullptr, 0u, kPTile};
  hipLaunchKernelGGL(k_probe_order, dim3(1), dim3(256), 0, s, d_qchrom, d_qlow, q0, q1, ws, seq);
  hipLaunchKernelGGL(k_query_pipe_dense, dim3(tiles < wgs ? tiles : wgs), dim3(kPThreads), 0, s, v, a);
  a.seq = 0;  // (k_fill_slices: index order)
  if (cap != 0) hipLaunchKernelGGL(k_fill_slices<false>, dim3(kFillBlocks), dim3(kQThreads), 0, s, v, a);
  BIVX_HIP(hipGetLastError());
  return 0;
}

// The index overlapped with itself (bivx_self_overlaps_dev): the queries are the index's own intervals in SLOT order —
// position-sorted by construction, which is what k_query_pipe_dense is for — and the result is wanted in id order:
// perm = the slots' ids. The kernel writes the lists back to back in slot order into d_tmp_hits and leaves, per id, the
// list's length and where it begins; offsets are a scan of the lengths; k_permute_lists gathers the lists into place.
// (Writing every list straight to its place from the slot-order pass was tried first: 50 M lists of ~68 bytes at random
// places of a 3.4 GB buffer are partial-line writes the memory side has to read-modify-write: 6.1 ms for that pass at
// config 5. Random READS of the same lists run at the gather rate, and the writes are a stream.)
bool self_overlaps_eligible(const IndexView &v, size_t n) {
  const char *env = std::getenv("BIVX_PIPE");
  const int mode = env ? std::atoi(env) : 1;
  if (!mode || v.flt_kind != BIVX_FILTER_NONE || v.max_segs > 1 || !fits_lds(v)) return false;
  if (v.nslots > (1u << 28) || v.max_cell > kMaxCellForPipe) return false;
  if (n > pipe_queries_per_launch()) return false;  // (one launch: nothing chains output positions across launches)
  return n >= (size_t)64 * kPTile || mode == 2;
}

// result list i = d_tmp[src[i] .. src[i] + (offsets[i + 1] - offsets[i])), written to d_hits[offsets[i] ..). A wavefront
// takes 64 consecutive lists — one contiguous piece of the output — and every lane one output element at a time: the
// element's list is found by bisection of the 64 list ends in LDS, its source is a gather, the stores are a stream.
__global__ __launch_bounds__(kQThreads) void k_permute_lists(const uint64_t *__restrict__ offsets,
                                                             const uint64_t *__restrict__ src, const uint32_t *__restrict__ tmp,
                                                             uint32_t *__restrict__ hits, size_t n, uint64_t cap) {
  __shared__ uint32_t s_end[kQWaves][kWave];
  __shared__ uint64_t s_src[kQWaves][kWave];
  const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const size_t i = (size_t)blockIdx.x * kQThreads + threadIdx.x;
  const uint64_t o0 = offsets[i < n ? i : n], o1 = offsets[i < n ? i + 1 : n];
  const uint64_t wb = __shfl((unsigned long long)o0, 0, kWave);  // the wavefront's piece of the output: [wb, we)
  const uint64_t we = __shfl((unsigned long long)o1, kWave - 1, kWave);
  if (we - wb > 0xFFFFFFFFull) {  // (more than 2^32 ids in 64 lists: every lane copies its own)
    if (i < n)
      for (uint64_t k = 0; k < o1 - o0; ++k)
        if (o0 + k < cap) hits[o0 + k] = tmp[(src[i] & kSelfPosMask) + k];
    return;
  }
  s_end[wave][lane] = (uint32_t)(o1 - wb);
  s_src[wave][lane] = i < n ? (src[i] & kSelfPosMask) - (o0 - wb) : 0ull;  // source of the list's first element, minus its place in the piece
  wave_sync_lds();
  const uint32_t total = (uint32_t)(we - wb);
  // every lane one output element of four consecutive rows of 64 per trip: four bisections, then four gathers in flight,
  // then four row stores (each row a 256-byte stream). (One row per trip: 3.3 ms at config 5; four consecutive
  // elements per lane moved as 16 bytes — unaligned gathers and stores — 4.9.)
  constexpr uint32_t kRowsPerTrip = 4;
  for (uint32_t e0 = lane; e0 < total; e0 += kWave * kRowsPerTrip) {
    uint32_t owner[kRowsPerTrip];
#pragma unroll
    for (uint32_t r = 0; r < kRowsPerTrip; ++r) {
      const uint32_t e = e0 + r * kWave;
      uint32_t lo = 0, hi = kWave - 1;  // first list whose end is beyond e
#pragma unroll
      for (int step = 0; step < 6; ++step) {  // (64 lists: six halvings, branch-free)
        const uint32_t m = (lo + hi) >> 1;
        const bool right = s_end[wave][m] <= e;
        lo = right ? m + 1 : lo;
        hi = right ? hi : m;
      }
      owner[r] = lo < kWave ? lo : kWave - 1;
    }
    uint32_t x[kRowsPerTrip];
    bool ok[kRowsPerTrip];
#pragma unroll
    for (uint32_t r = 0; r < kRowsPerTrip; ++r) {
      const uint32_t e = e0 + r * kWave;
      const uint64_t sp = s_src[wave][owner[r]] + e;
      ok[r] = e < total && wb + e < cap && sp < cap;
      x[r] = ok[r] ? stream_load(tmp + sp) : 0u;
    }
#pragma unroll
    for (uint32_t r = 0; r < kRowsPerTrip; ++r)
      if (ok[r]) stream_store(hits + wb + e0 + r * kWave, x[r]);
  }
}

// The same, a line at a time (the default; k_permute_lists keeps wavefronts with very long lists). Random 128-byte lines
// come out of HBM at 45-50 G per second when a group of eight lanes asks for one (tools/ub_gather.hip, tables of 1-8
// GB); the element-by-element gather above reaches 23 G. Here the wavefront's 64 lists — one contiguous piece of the
// output — are put together in LDS: in step j of a round, group g fetches one aligned 128-byte line of the list of its
// lane j (lane p the p-th 16 bytes) and every lane drops the up to four ids of its 16 bytes that belong to the list at
// their places in the piece; the eight steps' loads leave together, a list of k lines takes k rounds. The piece then
// leaves in whole lines. Pieces beyond the buffer go in runs of consecutive lists that fit.
#ifndef BIVX_PERM_BUF
#define BIVX_PERM_BUF 1536
#endif
constexpr uint32_t kPermBuf = BIVX_PERM_BUF;      // ids a wavefront puts together at a time (6 KB: 1024 / 1536 / 2048 / 3072 / 4096 -> 5.11 / 4.48 / 4.59 / 4.84 / 5.65 ms at config 5)
constexpr uint32_t kPermListMax = 1024;  // a wavefront with a longer list takes the element-wise way

// S: every list leaves in ascending order (ordered by its lane while the piece sits in LDS: no second pass over the ids)
// The kernel makes the offsets itself — the exclusive prefix sum of the lists' lengths (the words' high bits) — and writes them:
// a workgroup's 256 ids begin where its tile of 8 192 begins (tile_prefix) + the blocks of 256 before it inside the tile
// (block_sums; self_length_sums) + the wavefronts before this one. (A scan pass in front read the words and wrote the offsets,
// and this kernel read both again: 0.19 ms of 4.2 at config 5.)
template <bool S>
__global__ __launch_bounds__(kQThreads) void k_permute_lines(uint64_t *__restrict__ offsets,
                                                             const uint64_t *__restrict__ src, const uint32_t *__restrict__ tmp,
                                                             uint32_t *__restrict__ hits, size_t n, uint64_t cap,
                                                             const uint64_t *__restrict__ tile_prefix,
                                                             const uint64_t *__restrict__ block_sums) {
  static_assert(kQThreads == 256, "self_length_sums sums blocks of 256 ids");
  __shared__ __attribute__((aligned(16))) uint32_t s_buf[kQWaves][kPermBuf];
  __shared__ uint4 s_tab[kQWaves][kWave];  // per list: (source low, source high, length, place in the run)
  __shared__ uint64_t s_wtot[kQWaves];
  __shared__ uint64_t s_base;
  const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const size_t i = (size_t)blockIdx.x * kQThreads + threadIdx.x;
  const uint64_t word = i < n ? src[i] : 0ull;
  uint64_t o0, o1;
  {
    const uint64_t mylen = word >> kSelfPosBits;
    unsigned long long incl = mylen;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const unsigned long long o = __shfl_up(incl, d, kWave);
      if ((int)lane >= d) incl += o;
    }
    if (lane == kWave - 1) s_wtot[wave] = incl;
    if (wave == 0) {  // where the workgroup's ids begin
      const uint32_t b = blockIdx.x, first = b & ~31u;
      unsigned long long part = lane < (b & 31u) ? block_sums[first + lane] : 0ull;
#pragma unroll
      for (int d = 1; d < 32; d <<= 1) part += __shfl_xor(part, d, kWave);
      if (lane == 0) s_base = tile_prefix[b >> 5] + part;
    }
    __syncthreads();
    uint64_t before = s_base;
    for (uint32_t w = 0; w < wave; ++w) before += s_wtot[w];
    o1 = before + incl;
    o0 = o1 - mylen;
    if (i < n) {
      offsets[i] = o0;
      if (i == n - 1) offsets[n] = o1;
    }
  }
  const uint64_t wb = __shfl((unsigned long long)o0, 0, kWave);  // the wavefront's piece of the output: [wb, we)
  const uint64_t we = __shfl((unsigned long long)o1, kWave - 1, kWave);
  const uint64_t sp = i < n && o1 > o0 ? word & kSelfPosMask : 0ull;
  if (we - wb > 0xFFFFFFFFull || __any(o1 - o0 > kPermListMax)) {
    // (rare: the element-wise gather of k_permute_lists, one list per lane)
    if (i < n)
      for (uint64_t k = 0; k < o1 - o0; ++k)
        if (o0 + k < cap && sp + k < cap) hits[o0 + k] = tmp[sp + k];
    if (S) {
      wave_sync_mem();
      const uint64_t c0 = o0 < cap ? o0 : cap, c1 = o1 < cap ? o1 : cap;
      wave_sort_lists<kPermBuf, kRankBlock, false>(s_buf[wave], c0, c1, hits, (int)lane);
    }
    return;
  }
  const uint32_t len = (uint32_t)(o1 - o0), loff = (uint32_t)(o0 - wb);
  const uint32_t p = lane & 7u, gsh = lane & 0x38u;
  uint32_t *const buf = s_buf[wave];
  uint32_t first = 0;
  while (first < (uint32_t)kWave) {
    // a run of consecutive lists that fit the buffer together (a list fits by itself)
    const uint32_t base = __shfl(loff, (int)first, kWave);
    const uint64_t fit = __ballot(lane >= first && loff + len - base <= kPermBuf);
    const uint64_t nofit = ~fit & (~0ull << first);
    const uint32_t next = nofit ? (uint32_t)__ffsll((long long)nofit) - 1u : (uint32_t)kWave;
    const bool mine = lane >= first && lane < next && len != 0;
    // the list's first line (32 ids) and how many it spans
    const uint32_t shift = (uint32_t)sp & 31u;
    const uint32_t nlines = mine ? (shift + len + 31u) >> 5 : 0u;
    s_tab[wave][lane] = make_uint4((uint32_t)sp, (uint32_t)(sp >> 32), mine ? len : 0u, loff - base);
    wave_sync_lds();
    for (uint32_t r = 0; __any(r < nlines); ++r) {
      uint4 v[8];
#pragma unroll
      for (uint32_t j = 0; j < 8; ++j) {
        const uint4 t = s_tab[wave][gsh | j];
        const uint64_t s0 = (uint64_t)t.x | (uint64_t)t.y << 32;
        const uint32_t sh = t.x & 31u;
        // this lane's four ids: elements e0 .. e0 + 3 of the list, e0 = 32 r + 4 p - sh
        const int32_t e0 = (int32_t)(32u * r + 4u * p) - (int32_t)sh;
        asm volatile("" : "=v"(v[j].x), "=v"(v[j].y), "=v"(v[j].z), "=v"(v[j].w));
        const uint64_t at = (s0 & ~31ull) + 32u * r + 4u * p;
        if (e0 + 3 >= 0 && e0 < (int32_t)t.z && at + 3 < cap + 32u) v[j] = *reinterpret_cast<const uint4 *>(tmp + at);
      }
#pragma unroll
      for (uint32_t j = 0; j < 8; ++j) {
        const uint4 t = s_tab[wave][gsh | j];
        const uint32_t sh = t.x & 31u;
        const int32_t e0 = (int32_t)(32u * r + 4u * p) - (int32_t)sh;
        const uint32_t x[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (e0 + k >= 0 && e0 + k < (int32_t)t.z) buf[t.w + (uint32_t)(e0 + k)] = x[k];
      }
    }
    wave_sync_lds();
    if (S) {  // every lane of the run orders its own list where it lies (lists of up to kPermListMax ids)
      const uint32_t nl = mine ? len : 0u, off = loff - base, longest = wave_max(nl);
      if (longest <= 8u)
        net_sort_list<8>(buf, buf, off, nl);
      else if (longest <= 16u)
        net_sort_list<16>(buf, buf, off, nl);
      else if (longest <= 32u)
        net_sort_list<32>(buf, buf, off, nl);
      else if (longest <= 64u)
        net_sort_list64(buf, off, nl);
      else {
        // (rare: a list beyond 64 ids — the whole wavefront orders such a list, one after the other)
        net_sort_list64(buf, off, nl <= 64u ? nl : 0u);
        uint64_t todo = __ballot(nl > 64u);
        while (todo) {
          const int src = __ffsll((long long)todo) - 1;
          todo &= todo - 1;
          wave_sync_mem();
          wave_bitonic_sort<uint32_t>(buf + __shfl(off, src, kWave), __shfl(nl, src, kWave), (int)lane);
        }
      }
      wave_sync_lds();
    }
    const uint32_t nthis = __shfl(loff + len, (int)next - 1, kWave) - base;
    const uint64_t out0 = wb + base;
    const uint32_t lim = cap > out0 ? (cap - out0 < nthis ? (uint32_t)(cap - out0) : nthis) : 0u;
    {
      typedef uint32_t u32x4_a16 __attribute__((ext_vector_type(4)));
      uint32_t *const out = hits + out0;
      const uint32_t n4 = lim & ~3u;
      for (uint32_t e = lane * 4u; e < n4; e += kWave * 4u)
        __builtin_nontemporal_store(*reinterpret_cast<const u32x4_a16 *>(buf + e), reinterpret_cast<u32x4_a4 *>(out + e));
      if (n4 + lane < lim) stream_store(out + n4 + lane, buf[n4 + lane]);
    }
    wave_sync_lds();
    first = next;
  }
}

int launch_self_overlaps(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                         const uint32_t *d_perm, size_t n, uint64_t *d_src_by_id,
                         uint64_t *d_offsets_scratch, uint32_t *d_tmp_hits, uint64_t cap, uint64_t *ws, bool self_clean,
                         hipStream_t s) {
  const int flags = (self_clean ? kFlagSelfClean : 0) | kFlagFinal | kFlagSorted;
  unsigned wgs = 512;
  {
    wgs = 2u * cus_of_current_device();
    if (const char *e = std::getenv("BIVX_PIPE_WGS")) {
      const long w = std::atol(e);
      if (w >= 1 && w <= 65536) wgs = (unsigned)w;
    }
  }
  if (n > pipe_queries_per_launch()) {  // (launches would have to chain their output positions: not needed below 62 M)
    set_error("bivx_self_overlaps_dev: more than %zu intervals", pipe_queries_per_launch());
    return BIVX_E_RANGE;
  }
  const unsigned tiles = (unsigned)((n + kPTile - 1) / kPTile);
  PipeArgs a{d_qchrom, d_qlow, d_qhigh, 0, n, d_offsets_scratch, d_tmp_hits, cap, ws, tiles, flags, 1u,
             nullptr, nullptr, d_perm, d_src_by_id, nullptr, 0u, kPTile};
  hipLaunchKernelGGL(k_query_pipe_dense, dim3(tiles < wgs ? tiles : wgs), dim3(kPThreads), 0, s, v, a);
  if (cap != 0) {
    a.seq = 0;  // (k_fill_slices: index order)
    hipLaunchKernelGGL(k_fill_slices<false>, dim3(kFillBlocks), dim3(kQThreads), 0, s, v, a);
  }
  BIVX_HIP(hipGetLastError());
  return 0;
}

// *sorted: the lists left in ascending order (sort_ids asked for it and the line-wise kernel ran), else the caller orders them
int launch_permute_lists(uint64_t *d_offsets, const uint64_t *d_src, const uint32_t *d_tmp, uint32_t *d_hits, size_t n,
                         uint64_t cap, bool sort_ids, bool *sorted, void *d_scan, hipStream_t s) {
  *sorted = false;
  static const bool by_elements = [] {  // (BIVX_PERMUTE=elements: the first form, for comparison)
    const char *e = std::getenv("BIVX_PERMUTE");
    return e && e[0] == 'e';
  }();
  if (n == 0 || by_elements) {
    BIVX_TRY(exclusive_scan_lengths_u64(d_src, d_offsets, n, d_scan, s));
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_permute_lists, dim3((unsigned)((n + kQThreads - 1) / kQThreads)), dim3(kQThreads), 0, s, d_offsets,
                       d_src, d_tmp, d_hits, n, cap);
    BIVX_HIP(hipGetLastError());
    return 0;
  }
  const uint64_t *tile_prefix = nullptr, *block_sums = nullptr;
  BIVX_TRY(self_length_sums(d_src, n, d_scan, &tile_prefix, &block_sums, s));
  if (sort_ids) {
    hipLaunchKernelGGL(k_permute_lines<true>, dim3((unsigned)((n + kQThreads - 1) / kQThreads)), dim3(kQThreads), 0, s,
                       d_offsets, d_src, d_tmp, d_hits, n, cap, tile_prefix, block_sums);
    *sorted = true;
  } else {
    hipLaunchKernelGGL(k_permute_lines<false>, dim3((unsigned)((n + kQThreads - 1) / kQThreads)), dim3(kQThreads), 0, s,
                       d_offsets, d_src, d_tmp, d_hits, n, cap, tile_prefix, block_sums);
  }
  BIVX_HIP(hipGetLastError());
  return 0;
}

#ifdef BIVX_STAMPS
extern "C" int bivx_debug_wgstamps(unsigned long long *out, size_t n) {
  if (n > 1024 * 8) return -1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgstamps), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
extern "C" int bivx_debug_pstamps(unsigned long long *out, size_t n) {
  if (n > (size_t)kPStampTiles * kPStampSlots) n = (size_t)kPStampTiles * kPStampSlots;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pstamps), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace bivx
