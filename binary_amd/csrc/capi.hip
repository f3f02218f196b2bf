// capi.hip — the C ABI of libbivx.so (include/bivx.h) and the host-side build logic.
//
// Host code is plain C++ over the HIP runtime: it owns device memory, sequences kernels on a stream and
// makes the one data-dependent decision of the build (length classes per chromosome). No CPU fallback:
// every entry point needs a working gfx950 device.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "common.h"

namespace bivx {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace bivx

using namespace bivx;

struct bivx_index {
  // a handle made by bivx_create_sharded owns one ordinary index per device instead of what follows (sharded.cpp)
  bivx::ShardedState *sharded = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;
  // append-order copy (what RbTree::insert_node received, in order)
  uint32_t *d_chrom = nullptr, *d_low = nullptr, *d_high = nullptr;
  uint8_t *d_type = nullptr;  // svtype per interval (bivx_append_typed); allocated with the first typed append
  bool typed = false;         // some append carried types
  size_t n = 0, cap = 0;
  // built index. The arrays live in grow-only device blocks that a rebuild reuses (bivx_clear keeps them too): a build
  // of 10 M intervals otherwise spends as long in hipMalloc / hipFree — which synchronises the device — as in kernels.
  struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
  };
  DevBuf b_se, b_table, b_seg;  // what the pointers below point into (b_se: se[] and rec[]; b_seg: every small table)
  DevBuf b_keys[2], b_ids[2];                 // sort buffers; the ids end up in one of b_ids and stay there (d_id)
  DevBuf b_misc, b_radix, b_scalar, b_segof;  // build temporaries: statistics, key tables, histogram scratch, maxima,
                                              // every interval's segment (two-sort builds)
  // The statistics the build plans with, taken while the intervals come in (untyped appends): every append adds its
  // intervals to this table — device-side appends by the very pass that copies them into the columns — so that
  // bivx_build reads 120 MB less at 10 M intervals and an append of a few records to a large index does not pay for
  // a pass over all of it. astats_n: intervals the table covers (== n: the build uses it).
  DevBuf b_astats;
  bool astats_ready = false;                  // emptied since creation / the last bivx_clear
  size_t astats_n = 0;
  hipEvent_t astats_ev = nullptr;             // the emptying, for appends on other streams to wait for
  uint32_t *h_scalars = nullptr;              // pinned: the build's few read-backs (max ids, largest cell)
  void *h_stage = nullptr;                    // pinned, grow-only: the statistics on their way down, the small tables
  size_t h_stage_cap = 0;                     // of the index on their way up
  // bivx_append_dev copies on the CALLER's stream; the build waits for those copies on its own stream through events
  // instead of synchronising the device
  std::vector<hipEvent_t> ev_pending, ev_free;
  // the built index's own intervals as a query batch in slot order (chromosome, low, high), made by the first
  // bivx_self_overlaps_dev call after a build
  mutable DevBuf b_selfq;
  mutable bool selfq_valid = false;
  mutable std::mutex self_mutex;
  uint32_t *d_seg_chrom = nullptr;     // chromosome of every segment (behind the descriptors' sort keys in b_seg)
  bool built = false;
  uint2 *d_se = nullptr;
  uint2 *d_rec = nullptr;
  uint32_t *d_id = nullptr;
  uint32_t *d_table = nullptr;
  SegDesc *d_seg = nullptr;
  uint2 *d_chrom_rng = nullptr;        // ntypes rows of nchrom (first segment, count) pairs; row 0 = every type
  uint32_t nchrom = 0, nseg = 0, ntypes = 1;
  std::vector<uint32_t> max_segs;      // per row: most segments any one chromosome has
  uint64_t nentries = 0;
  uint32_t max_cell = 0;  // most slots in any directory cell (positional hotspots)
  uint32_t max_window = 0;  // the planner's estimate of the longest usual window (ClassPlan::max_window)
  uint32_t order_shift = 0; // IndexView::order_shift of the built index
  size_t built_n = 0;
  double build_ms = 0.0;
  // prefix workspaces of bivx_query_dev calls made without a caller workspace: one per stream (calls on one
  // stream are ordered, so they can share it); zeroed once, and every launch leaves its workspace zeroed again
  struct Workspace {
    void *p = nullptr;
    bool needs_reset = false;  // a reported error may have left protocol state behind: cleared before the next launch
    void *self_p = nullptr;    // bivx_self_overlaps_dev's counts and scan scratch on this stream
    size_t self_cap = 0;
  };
  mutable std::mutex ws_mutex;
  mutable std::unordered_map<hipStream_t, Workspace> ws_of_stream;
  // Caller streams on which device-pointer calls have read the built arrays since the last build: a rebuild (and
  // bivx_clear) overwrites those arrays on idx->stream and must come after whatever is still running there. The
  // query path pays one hash insert for it; the build waits for these streams, not for the whole device.
  mutable std::unordered_set<hipStream_t> reader_streams;
  // error block of the kernels (IndexView::err): pinned host memory mapped into the device, so that every
  // synchronising entry point can look at it without a copy
  uint32_t *h_err = nullptr;  // host address
  uint32_t *d_err = nullptr;  // the same memory as the device sees it
  mutable std::atomic<uint64_t> errors_reported{0};
  // device blocks the host-pointer entry points used for their temporaries and handed back: a call with a handful
  // of queries otherwise spends more time in hipMalloc / hipFree (which synchronises the device) than on the GPU
  mutable std::mutex cache_mutex;
  mutable std::multimap<size_t, void *> cache_free;  // block size -> block
  // pinned host blocks mapped into the device (host address, device address), for the few-queries path of
  // bivx_find_overlaps: the kernel reads the queries and writes offsets and ids straight through them
  mutable std::vector<std::pair<void *, void *>> mailboxes;
  mutable size_t cache_bytes = 0;
  // Const queries on one index from several host threads (the reference shares one tree across its pool threads,
  // mapper.cpp:127-142): every host-pointer query call runs on a LANE of its own — a stream, an error block (pinned, mapped
  // into the device) and, through ws_of_stream, a prefix workspace — so that the calls' device round trips overlap and a
  // failed kernel is reported to the call that launched it. Lane 0 is (idx->stream, h_err, d_err); further lanes are made
  // when a call finds none free, up to kMaxLanes (beyond that a call waits for one).
  struct Lane {
    hipStream_t stream = nullptr;
    uint32_t *h_err = nullptr, *d_err = nullptr;
  };
  static constexpr size_t kMaxLanes = 32;
  mutable std::mutex lane_mutex;
  mutable std::condition_variable lane_cv;
  mutable std::vector<Lane> lanes_free;
  mutable size_t lanes_made = 0;  // lane 0 included once it has been handed out for the first time
  mutable std::vector<Lane> lanes_all;  // (for teardown)
};

// the device-pointer entry points address ONE device's memory: a sharded handle has no meaning for them
#define BIVX_NOT_SHARDED(idx, who)                                                                         \
  if ((idx) && (idx)->sharded) {                                                                           \
    set_error("%s: a sharded index takes host pointers (bivx_find_overlaps, bivx_count, bivx_fill, "      \
              "bivx_any); device pointers belong to one device",                                           \
              who);                                                                                        \
    return BIVX_E_STATE;                                                                                   \
  }

namespace {

// cost of one extra segment search, in scanned-candidate units (BIVX_SEARCH_COST overrides: tuning knob)
static const double kSearchCost = [] {
  const char *e = std::getenv("BIVX_SEARCH_COST");
  const double v = e ? std::atof(e) : 16.0;
  return v > 0.0 ? v : 16.0;
}();

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = (prev == dev) || hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

#define BIVX_GUARD(idx)                                               \
  DeviceGuard bivx_guard_((idx)->device);                             \
  if (!bivx_guard_.ok) {                                              \
    set_error("hipSetDevice(%d) failed", (idx)->device);              \
    return BIVX_E_HIP;                                                \
  }

// hipMalloc that gives the device memory of parked index objects (bivx_destroy keeps up to BIVX_INDEX_POOL of them with
// their grow-only blocks, invisible to the caller) back before it reports failure: a build that fitted before an index
// was dropped must still fit afterwards.
extern "C" void bivx_release_pooled(void);
hipError_t dev_malloc(void **p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipSuccess) return e;
  (void)hipGetLastError();
  bivx_release_pooled();
  return hipMalloc(p, bytes);
}

// Temporaries of one call: everything allocated through it is handed back when it goes out of scope — to the
// index's block cache when the block is small (the caller has synchronised its stream by then), else to hipFree.
struct TempPool {
  static constexpr size_t kMaxCachedBlock = 16u << 20, kMaxCachedTotal = 256u << 20;
  const bivx_index *idx;
  hipStream_t stream;  // the stream the call's work runs on (its lane's)
  std::vector<std::pair<void *, size_t>> ptrs;
  explicit TempPool(const bivx_index *owner, hipStream_t s = nullptr) : idx(owner), stream(s ? s : (owner ? owner->stream : nullptr)) {}
  TempPool(const TempPool &) = delete;
  TempPool &operator=(const TempPool &) = delete;
  ~TempPool() {
    // every user has synchronised its stream on its way out; an error return may not have, and a recycled block
    // must not be in use: an idle stream makes this a no-op
    if (idx && stream && !ptrs.empty()) (void)hipStreamSynchronize(stream);
    for (auto &b : ptrs) {
      if (!b.first) continue;
      if (idx && b.second <= kMaxCachedBlock) {
        std::lock_guard<std::mutex> lock(idx->cache_mutex);
        if (idx->cache_bytes + b.second <= kMaxCachedTotal) {
          idx->cache_free.emplace(b.second, b.first);
          idx->cache_bytes += b.second;
          continue;
        }
      }
      (void)hipFree(b.first);
    }
  }
  template <typename T>
  int alloc(T **out, size_t count) {
    size_t bytes = 256;  // size classes: powers of two
    while (bytes < (count ? count : 1) * sizeof(T)) bytes <<= 1;
    void *p = nullptr;
    if (idx && bytes <= kMaxCachedBlock) {
      std::lock_guard<std::mutex> lock(idx->cache_mutex);
      auto it = idx->cache_free.find(bytes);
      if (it != idx->cache_free.end()) {
        p = it->second;
        idx->cache_free.erase(it);
        idx->cache_bytes -= bytes;
      }
    }
    if (!p) BIVX_HIP(dev_malloc(&p, bytes));
    ptrs.emplace_back(p, bytes);
    *out = static_cast<T *>(p);
    return 0;
  }
  void release(void *p) {  // ownership moves to the index
    for (auto &q : ptrs)
      if (q.first == p) q.first = nullptr;
  }
};

void drop_block_cache(const bivx_index *idx) {
  std::lock_guard<std::mutex> lock(idx->cache_mutex);
  for (auto &kv : idx->cache_free) (void)hipFree(kv.second);
  idx->cache_free.clear();
  idx->cache_bytes = 0;
}

// the index is no longer searchable; its device blocks stay for the next build
void free_built(bivx_index *idx) {
  idx->d_se = nullptr;
  idx->d_rec = nullptr;
  idx->d_id = nullptr;
  idx->d_table = nullptr;
  idx->d_seg = nullptr;
  idx->d_chrom_rng = nullptr;
  idx->nchrom = idx->nseg = 0;
  idx->ntypes = 1;
  idx->max_segs.clear();
  idx->nentries = 0;
  idx->built = false;
  idx->selfq_valid = false;
}

void release_build_blocks(bivx_index *idx) {
  for (bivx_index::DevBuf *b : {&idx->b_se, &idx->b_table, &idx->b_seg, &idx->b_keys[0],
                                &idx->b_keys[1], &idx->b_ids[0], &idx->b_ids[1], &idx->b_misc, &idx->b_radix, &idx->b_scalar, &idx->b_segof, &idx->b_selfq, &idx->b_astats}) {
    (void)hipFree(b->p);
    b->p = nullptr;
    b->cap = 0;
  }
}

// grow-only: a block that is large enough is handed back as it is
int ensure_block(bivx_index::DevBuf &b, size_t bytes) {
  if (bytes <= b.cap && b.p) return 0;
  (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
  const size_t want = (bytes + 255) & ~(size_t)255;
  BIVX_HIP(dev_malloc(&b.p, want ? want : 256));
  b.cap = want ? want : 256;
  return 0;
}

int ensure_stage(bivx_index *idx, size_t bytes) {
  if (bytes <= idx->h_stage_cap && idx->h_stage) return 0;
  if (idx->h_stage) (void)hipHostFree(idx->h_stage);
  idx->h_stage = nullptr;
  idx->h_stage_cap = 0;
  const size_t want = std::max<size_t>((bytes + 4095) & ~(size_t)4095, 128 << 10);
  BIVX_HIP(hipHostMalloc(&idx->h_stage, want, hipHostMallocDefault));
  idx->h_stage_cap = want;
  return 0;
}

// the copies of a bivx_append*_dev call were just enqueued on `s`: the next build must come after them
int note_device_append(bivx_index *idx, hipStream_t s) {
  hipEvent_t ev = nullptr;
  if (!idx->ev_free.empty()) {
    ev = idx->ev_free.back();
    idx->ev_free.pop_back();
  } else {
    BIVX_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  }
  if (hipEventRecord(ev, s) != hipSuccess) {  // (e.g. a stream that is being captured: fall back to a full wait)
    idx->ev_free.push_back(ev);
    BIVX_HIP(hipDeviceSynchronize());
    return 0;
  }
  idx->ev_pending.push_back(ev);
  if (idx->ev_pending.size() > 256) {  // (thousands of appends without a build: wait once instead of piling events up)
    BIVX_HIP(hipDeviceSynchronize());
    for (hipEvent_t e : idx->ev_pending) idx->ev_free.push_back(e);
    idx->ev_pending.clear();
  }
  return 0;
}

// a device-pointer call is about to read the built index on the caller's stream `s`
void note_reader(const bivx_index *idx, hipStream_t s) {
  if (s == idx->stream) return;  // (ordered with the build by the stream itself)
  std::lock_guard<std::mutex> lock(idx->ws_mutex);
  idx->reader_streams.insert(s);
}

// Before the built arrays (or the append-order columns they were made from) are overwritten: every query, fill or
// self-overlap call enqueued so far on a caller stream must have finished reading them. The streams themselves are NOT touched
// again: a caller may have destroyed one since, and the runtime does not report such a handle as invalid — it dereferences it
// (a crash in hipStreamSynchronize, found with a parked index object whose last owner's streams were gone). If any caller
// stream has read the index since the last wait, the whole device is waited for; the usual build — no device-pointer call
// since the last one — skips this.
int wait_for_readers(bivx_index *idx) {
  bool any;
  {
    std::lock_guard<std::mutex> lock(idx->ws_mutex);
    any = !idx->reader_streams.empty();
    idx->reader_streams.clear();
  }
  if (any) BIVX_HIP(hipDeviceSynchronize());
  return 0;
}

int ensure_capacity(bivx_index *idx, size_t need) {
  if (need <= idx->cap) return 0;
  if (need >= 0xFFFFFFFFull) {
    set_error("too many intervals: %zu (ids are uint32)", need);
    return BIVX_E_RANGE;
  }
  size_t nc = idx->cap ? idx->cap * 2 : 1024;
  if (nc < need) nc = need;
  if (nc > 0xFFFFFFFEull) nc = 0xFFFFFFFEull;
  uint32_t *c = nullptr, *l = nullptr, *h = nullptr;
  uint8_t *ty = nullptr;
  auto grow = [&]() -> int {
    BIVX_HIP(dev_malloc((void **)&c, nc * sizeof(uint32_t)));
    BIVX_HIP(dev_malloc((void **)&l, nc * sizeof(uint32_t)));
    BIVX_HIP(dev_malloc((void **)&h, nc * sizeof(uint32_t)));
    BIVX_HIP(dev_malloc((void **)&ty, nc));
    BIVX_HIP(hipMemsetAsync(ty, 0, nc, idx->stream));
    if (idx->n) {
      // Earlier bivx_append_dev calls copy on the CALLER's streams, which idx->stream is not ordered after: wait
      // for the whole device before the old arrays are read and freed (growth doubles, so this is rare).
      BIVX_HIP(hipDeviceSynchronize());
      for (hipEvent_t ev : idx->ev_pending) idx->ev_free.push_back(ev);  // (all of them have happened now)
      idx->ev_pending.clear();
      BIVX_HIP(hipMemcpyAsync(c, idx->d_chrom, idx->n * 4, hipMemcpyDeviceToDevice, idx->stream));
      BIVX_HIP(hipMemcpyAsync(l, idx->d_low, idx->n * 4, hipMemcpyDeviceToDevice, idx->stream));
      BIVX_HIP(hipMemcpyAsync(h, idx->d_high, idx->n * 4, hipMemcpyDeviceToDevice, idx->stream));
      BIVX_HIP(hipMemcpyAsync(ty, idx->d_type, idx->n, hipMemcpyDeviceToDevice, idx->stream));
    }
    BIVX_HIP(hipStreamSynchronize(idx->stream));
    return 0;
  };
  if (int rc = grow()) {
    (void)hipFree(c);
    (void)hipFree(l);
    (void)hipFree(h);
    (void)hipFree(ty);
    return rc;
  }
  (void)hipFree(idx->d_type);
  idx->d_type = ty;
  (void)hipFree(idx->d_chrom);
  (void)hipFree(idx->d_low);
  (void)hipFree(idx->d_high);
  idx->d_chrom = c;
  idx->d_low = l;
  idx->d_high = h;
  idx->cap = nc;
  return 0;
}

constexpr size_t kMaxIdleWorkspaces = 32;  // per-stream workspaces an index keeps across a rebuild
constexpr size_t kAppendStatsFrom = (size_t)4 << 20;  // intervals in an index's first append from which statistics ride on appends

int append_impl(bivx_index *idx, const uint32_t *chrom, const uint32_t *low, const uint32_t *high,
                const uint8_t *svtype, size_t n, hipMemcpyKind kind, hipStream_t s) {
  if (!idx || (n && (!low || !high))) {
    set_error("bivx_append: null argument");
    return BIVX_E_INVALID;
  }
  if (n == 0) return 0;
  BIVX_GUARD(idx);
  if (kind != hipMemcpyHostToDevice && s) {
    // an append inside a stream capture would record its completion event as a graph node: the build can neither wait
    // for it nor stay out of the caller's capture
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
      set_error("bivx_append_dev: the stream is being captured; appends cannot be part of a graph");
      return BIVX_E_STATE;
    }
  }
  BIVX_TRY(ensure_capacity(idx, idx->n + n));
  // statistics ride on the append while the index has no interval types (see bivx_index::b_astats)
  // (test knobs: BIVX_NO_APPEND_STATS = the build's own pass always; BIVX_APPEND_STATS_FROM = the first append's size from which)
  const bool stats_at_append = !std::getenv("BIVX_NO_APPEND_STATS");
  size_t stats_from = kAppendStatsFrom;
  if (const char *e = std::getenv("BIVX_APPEND_STATS_FROM")) stats_from = (size_t)std::strtoull(e, nullptr, 10);
  // (begun by a bulk append only: below a few million intervals the build's own pass is cheaper than the bookkeeping here —
  // 1 M intervals: append + build 194 us with the pass in the build, 201 with it here; 50 M: 2.87 / 2.76 ms)
  const bool with_stats = stats_at_append && !svtype && !idx->typed && idx->astats_n == idx->n &&
                          (idx->astats_n != 0 || (idx->n == 0 && n >= stats_from));
  if (with_stats && !idx->astats_ready) {
    BIVX_TRY(ensure_block(idx->b_astats, auto_stats_bytes()));
    if (!idx->astats_ev) BIVX_HIP(hipEventCreateWithFlags(&idx->astats_ev, hipEventDisableTiming));
    BIVX_TRY(launch_init_auto_stats(idx->b_astats.p, idx->stream));
    BIVX_HIP(hipEventRecord(idx->astats_ev, idx->stream));
    idx->astats_ready = true;
  }
  if (with_stats && s != idx->stream) BIVX_HIP(hipStreamWaitEvent(s, idx->astats_ev, 0));
  if (with_stats && kind == hipMemcpyDeviceToDevice) {
    // one pass: the columns into the index, their statistics into the table
    BIVX_TRY(launch_append_stats(chrom, low, high, n, idx->d_chrom + idx->n, idx->d_low + idx->n, idx->d_high + idx->n,
                                 idx->b_astats.p, s));
  } else {
    if (chrom)
      BIVX_HIP(hipMemcpyAsync(idx->d_chrom + idx->n, chrom, n * 4, kind, s));
    else
      BIVX_HIP(hipMemsetAsync(idx->d_chrom + idx->n, 0, n * 4, s));
    BIVX_HIP(hipMemcpyAsync(idx->d_low + idx->n, low, n * 4, kind, s));
    BIVX_HIP(hipMemcpyAsync(idx->d_high + idx->n, high, n * 4, kind, s));
    if (with_stats)
      BIVX_TRY(launch_append_stats(idx->d_chrom + idx->n, idx->d_low + idx->n, idx->d_high + idx->n, n, nullptr, nullptr, nullptr,
                                   idx->b_astats.p, s));
  }
  if (with_stats) idx->astats_n += n;
  if (svtype) {
    BIVX_HIP(hipMemcpyAsync(idx->d_type + idx->n, svtype, n, kind, s));
    idx->typed = true;
  } else if (idx->typed) {  // (a fresh allocation is zero already; slots reused after bivx_clear are not)
    BIVX_HIP(hipMemsetAsync(idx->d_type + idx->n, 0, n, s));
  }
  if (kind == hipMemcpyHostToDevice) BIVX_HIP(hipStreamSynchronize(s));  // caller may reuse its buffers
  else BIVX_TRY(note_device_append(idx, s));
  idx->n += n;
  idx->built = false;
  return 0;
}

constexpr uint32_t kCrowdedCell = 64;  // slots in one directory cell beyond which an index is ordered by every bit of low

struct ClassPlan {
  std::vector<uint32_t> bin2seg;       // nchrom * kLenBins, 0xFFFFFFFF for empty bins
  std::vector<SegDesc> segs;           // grouped by chromosome, classes by ascending length
  std::vector<uint32_t> chrom_seg;     // nchrom + 1
  uint64_t nentries = 0;
  uint32_t max_low = 0;
  // sort keys: segment s owns the key range [segkey[s].x, segkey[s].x + (last - base)] (segkey[s].y = base), so that
  // ONE 32-bit key orders by (segment, low); key_span is the sum of the ranges (> 2^32: no such key exists)
  std::vector<uint2> segkey;
  uint64_t key_span = 0;
  // The sort may leave out the key's low order_shift bits (the smallest cell shift of any segment; key ranges begin at
  // multiples of 2^order_shift): the slots then come out ordered by directory cell with append order inside a cell —
  // all a query needs, it evaluates every slot of the cells it touches — and when that saves a whole radix pass the
  // build takes it (bivx_build). 0: the segments' key ranges are packed and every bit takes part.
  uint32_t order_shift = 0;
  // slots a point query's window is expected to hold in the segment where that is most: count x (longest length + a
  // directory cell) / coordinate span — what tells k_query_pipe_ms's territory from the wavefront-cooperative walk's
  uint32_t max_window = 0;
};

// Chooses, per chromosome, how to cut the 33 length bins into classes. A class is searched with two
// directory lookups (cost kSearchCost) and scans about density * max_len candidates that cannot be hits;
// a tiny dynamic programme over the non-empty bins minimises the sum. Uniform short intervals end up in
// one class; a few chromosome-scale intervals get a class of their own instead of widening every window.
int plan_classes(const std::vector<BinStats> &st, uint32_t nchrom, size_t n_total, ClassPlan &plan) {
  // Directory density: at most cnt / slots_per_cell cells per segment, i.e. slots_per_cell .. 2 x slots_per_cell slots per
  // cell. A finer directory means fewer candidate slots per window and twice the directory: it pays once the index is
  // past what an XCD's L2 holds anyway (config 3, 10 M intervals: 0.306 -> 0.293 ms per batch with 1 instead of 2; config
  // 2, 1 M intervals: 43.8 -> 46.2 us). BIVX_SLOTS_PER_CELL overrides.
  const uint64_t slots_per_cell = [&]() -> uint64_t {
    if (const char *e = std::getenv("BIVX_SLOTS_PER_CELL")) {
      const long v = std::atol(e);
      return (uint64_t)(v < 1 ? 1 : v);
    }
    return n_total >= ((size_t)4 << 20) ? 1u : 2u;
  }();
  plan.bin2seg.assign((size_t)nchrom * kLenBins, 0xFFFFFFFFu);
  plan.chrom_seg.assign(nchrom + 1, 0);
  plan.segs.clear();
  plan.max_window = 0;
  uint64_t begin = 0, table_off = 0;
  for (uint32_t c = 0; c < nchrom; ++c) {
    plan.chrom_seg[c] = (uint32_t)plan.segs.size();
    int bins[kLenBins], m = 0;
    for (int b = 0; b < kLenBins; ++b)
      if (st[(size_t)c * kLenBins + b].count) bins[m++] = b;
    if (m == 0) continue;
    double best[kLenBins + 1];
    int cut[kLenBins + 1];
    best[0] = 0.0;
    for (int j = 1; j <= m; ++j) {
      best[j] = 1e300;
      uint64_t cnt = 0;
      uint32_t mn = 0xFFFFFFFFu, mx = 0, ml = 0;
      for (int i = j; i >= 1; --i) {  // class = bins[i-1 .. j-1]
        const BinStats &s = st[(size_t)c * kLenBins + bins[i - 1]];
        cnt += s.count;
        mn = s.min_low < mn ? s.min_low : mn;
        mx = s.max_low > mx ? s.max_low : mx;
        ml = s.max_len > ml ? s.max_len : ml;
        const double span = (double)(mx - mn) + 1.0;
        double waste = (double)cnt * ((double)ml + 1.0) / span;
        if (waste > (double)cnt) waste = (double)cnt;
        const double cost = best[i - 1] + kSearchCost + waste;
        if (cost < best[j]) {
          best[j] = cost;
          cut[j] = i - 1;
        }
      }
    }
    // walk the cuts back, then emit classes in ascending length order
    int starts[kLenBins], ns = 0;
    for (int j = m; j > 0; j = cut[j]) starts[ns++] = cut[j];
    for (int k = ns - 1; k >= 0; --k) {
      const int i0 = starts[k], i1 = (k == 0) ? m : starts[k - 1];
      SegDesc d{};
      uint64_t cnt = 0, ninv = 0;
      uint32_t mn = 0xFFFFFFFFu, mx = 0, ml = 0;
      for (int i = i0; i < i1; ++i) {
        const BinStats &s = st[(size_t)c * kLenBins + bins[i]];
        cnt += s.count;
        ninv += s.n_inverted;
        mn = s.min_low < mn ? s.min_low : mn;
        mx = s.max_low > mx ? s.max_low : mx;
        ml = s.max_len > ml ? s.max_len : ml;
        plan.bin2seg[(size_t)c * kLenBins + bins[i]] = (uint32_t)plan.segs.size();
      }
      d.begin = (uint32_t)begin;
      d.end = (uint32_t)(begin + cnt);
      d.base = mn;
      d.last = mx;
      d.maxlen = ml;
      const uint64_t span = (uint64_t)mx - mn;
      const uint64_t target = cnt / slots_per_cell > 1 ? cnt / slots_per_cell : 1;  // (two and four cells per slot: 0.315, 0.369 ms)
      uint32_t sh = 0;
      while (sh < 31 && (span >> sh) + 1 > target) ++sh;
      d.shift = sh;
      if (ml <= 0xFFFFu && ninv == 0) d.shift |= kSegPacked;
      d.ncell = (uint32_t)((span >> sh) + 1);
      {
        const double est = (double)cnt * ((double)ml + (double)(1ull << sh)) / ((double)span + 1.0);
        const uint32_t w = est >= (double)cnt ? (uint32_t)cnt : (uint32_t)est;
        if (w > plan.max_window) plan.max_window = w;
      }
      if (table_off + d.ncell + 1 > 0xFFFFFFFFull) {
        set_error("bucket directory too large");
        return BIVX_E_RANGE;
      }
      d.table_off = (uint32_t)table_off;
      table_off += (uint64_t)d.ncell + 1;
      begin += cnt;
      if (mx > plan.max_low) plan.max_low = mx;
      if (std::getenv("BIVX_PLAN_DEBUG"))  // (stderr: what the planner made of the statistics)
        std::fprintf(stderr, "[bivx plan] partition %u class %d: %llu intervals, max length %u, lows %u..%u, cell 2^%u, %u cells%s\n",
                     c, ns - 1 - k, (unsigned long long)cnt, ml, mn, mx, sh, d.ncell, (d.shift & kSegPacked) ? ", packed" : "");
      plan.segs.push_back(d);
    }
  }
  plan.chrom_seg[nchrom] = (uint32_t)plan.segs.size();
  plan.nentries = table_off;
  plan.segkey.resize(plan.segs.size());
  auto lay_keys = [&](uint32_t align_bits) {
    const uint64_t a = (1ull << align_bits) - 1ull;
    plan.key_span = 0;
    for (size_t k = 0; k < plan.segs.size(); ++k) {
      plan.key_span = (plan.key_span + a) & ~a;
      plan.segkey[k] = make_uint2((uint32_t)plan.key_span, plan.segs[k].base);  // (meaningless once the sum passes 2^32)
      plan.key_span += (uint64_t)plan.segs[k].last - plan.segs[k].base + 1u;
    }
  };
  lay_keys(0);
  plan.order_shift = 0;
  if (!plan.segs.empty() && plan.key_span <= 0xFFFFFFFFull && !std::getenv("BIVX_BUILD_FULL_SORT")) {  // (env: test knob)
    uint32_t sh = 31;
    for (const SegDesc &d : plan.segs) sh = std::min(sh, d.shift & 31u);
    auto passes = [](uint64_t span, uint32_t from) {
      int bits = 0;
      while (bits < 32 && ((span - 1) >> bits) != 0) ++bits;
      return bits > (int)from ? (bits - (int)from + 7) / 8 : 1;
    };
    const int full = passes(plan.key_span, 0);
    const uint64_t packed_span = plan.key_span;
    lay_keys(sh);
    if (sh > 0 && plan.key_span <= 0xFFFFFFFFull && passes(plan.key_span, sh) < full) {
      plan.order_shift = sh;
    } else {
      lay_keys(0);
      (void)packed_span;
    }
  }
  return 0;
}

int bits_for(uint32_t maxval) {
  int b = 0;
  while (maxval) {
    ++b;
    maxval >>= 1;
  }
  return b;
}

// the error block the calling thread's kernels report to: its lane's, inside a host-pointer query call (LaneLease)
static thread_local uint32_t *g_lane_d_err = nullptr;

IndexView view_of(const bivx_index *idx, uint32_t svtype = 0) {
  IndexView v;
  v.se = idx->d_se;
  v.rec = idx->d_rec;
  v.id = idx->d_id;
  v.table = idx->d_table;
  v.seg = idx->d_seg;
  // row of the requested interval type; a type the index does not hold selects an all-empty row
  const uint32_t row = svtype < idx->ntypes ? svtype : idx->ntypes;
  v.chrom_rng = idx->d_chrom_rng + (size_t)row * idx->nchrom;
  v.nchrom = idx->nchrom;
  v.nseg = idx->nseg;
  v.max_segs = row < idx->max_segs.size() ? idx->max_segs[row] : 0;
  v.nslots = idx->built_n < 0xFFFFFFFFull ? (uint32_t)idx->built_n : 0xFFFFFFFFu;
  v.max_cell = idx->max_cell;
  v.max_window = idx->max_window;
  v.order_shift = idx->order_shift;
  v.flt_kind = BIVX_FILTER_NONE;
  v.flt_dist = 0;
  v.flt_strand = 0;
  v.flt_qaux = nullptr;
  v.flt_iaux = nullptr;
  v.err = g_lane_d_err ? g_lane_d_err : idx->d_err;
  return v;
}

// Turns a raised error word into BIVX_E_TIMEOUT, once. Call after a synchronisation that covers the kernels in
// question. The words are reset and every index-owned workspace is cleared before its next launch.
int report_device_errors(const bivx_index *idx, const char *who, uint32_t *block = nullptr) {
  // read-and-clear in one step per word: a flag the device raises between a read and a separate clear would be lost
  uint32_t *e = block ? block : idx->h_err;
  if (__atomic_load_n(&e[kErrTimeout], __ATOMIC_RELAXED) == 0 && __atomic_load_n(&e[kErrWorkspace], __ATOMIC_RELAXED) == 0)
    return 0;
  const bool timeout = __atomic_exchange_n(&e[kErrTimeout], 0u, __ATOMIC_ACQ_REL) != 0;
  const bool dirty = __atomic_exchange_n(&e[kErrWorkspace], 0u, __ATOMIC_ACQ_REL) != 0;
  if (!timeout && !dirty) return 0;  // (another thread's report took them)
  idx->errors_reported.fetch_add(1);
  {
    std::lock_guard<std::mutex> lock(idx->ws_mutex);
    for (auto &kv : idx->ws_of_stream) kv.second.needs_reset = true;
  }
  set_error("%s: %s%s%s: the result of that call is invalid, repeat it", who,
            timeout ? "a single-pass query kernel gave up waiting for an earlier workgroup's hit total" : "",
            timeout && dirty ? "; " : "",
            dirty ? "a single-pass query kernel found its prefix workspace not zeroed" : "");
  return BIVX_E_TIMEOUT;
}

// A host-pointer query call's lane for its duration (bivx_index::Lane). While it is held the calling thread's kernels
// report to the lane's error block (view_of) and report() reads that block.
struct LaneLease {
  const bivx_index *idx;
  bivx_index::Lane lane;
  bool ok = false;
  explicit LaneLease(const bivx_index *owner) : idx(owner) {
    std::unique_lock<std::mutex> lock(idx->lane_mutex);
    for (;;) {
      if (!idx->lanes_free.empty()) {
        lane = idx->lanes_free.back();
        idx->lanes_free.pop_back();
        ok = true;
        break;
      }
      if (idx->lanes_made == 0) {  // lane 0: what a single-threaded caller has always run on
        lane.stream = idx->stream;
        lane.h_err = idx->h_err;
        lane.d_err = idx->d_err;
        idx->lanes_made = 1;
        ok = true;
        break;
      }
      if (idx->lanes_made < bivx_index::kMaxLanes) {
        bivx_index::Lane l;
        void *h = nullptr, *d = nullptr;
        if (hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) == hipSuccess &&
            hipHostMalloc(&h, kErrWords * sizeof(uint32_t), hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
          std::memset(h, 0, kErrWords * sizeof(uint32_t));
          l.h_err = static_cast<uint32_t *>(h);
          l.d_err = static_cast<uint32_t *>(d);
          idx->lanes_all.push_back(l);
          ++idx->lanes_made;
          lane = l;
          ok = true;
          break;
        }
        (void)hipGetLastError();
        if (h) (void)hipHostFree(h);
        if (l.stream) (void)hipStreamDestroy(l.stream);
        // (no more to be had: wait for one like everybody beyond kMaxLanes)
      }
      idx->lane_cv.wait(lock);
    }
    if (ok) g_lane_d_err = lane.d_err;
  }
  ~LaneLease() {
    g_lane_d_err = nullptr;
    if (!ok) return;
    {
      std::lock_guard<std::mutex> lock(idx->lane_mutex);
      idx->lanes_free.push_back(lane);
    }
    idx->lane_cv.notify_one();
  }
  int report(const char *who) const { return report_device_errors(idx, who, lane.h_err); }
  LaneLease(const LaneLease &) = delete;
  LaneLease &operator=(const LaneLease &) = delete;
};

// view with a fused post-filter; the aux pointers are DEVICE pointers here
int view_with_filter(const bivx_index *idx, const bivx_filter *f, IndexView &v) {
  if (f && f->svtype > 255u) {
    set_error("filter svtype %u out of range (0 = any, 1..255)", f->svtype);
    return BIVX_E_INVALID;
  }
  v = view_of(idx, f ? f->svtype : 0u);
  if (!f || f->kind == BIVX_FILTER_NONE) return 0;
  if (f->kind > BIVX_FILTER_SV2NL_TRA) {
    set_error("unknown filter kind %u", f->kind);
    return BIVX_E_INVALID;
  }
  if (f->kind == BIVX_FILTER_SV2NL_TRA && (!f->query_aux || !f->interval_aux)) {
    set_error("BIVX_FILTER_SV2NL_TRA needs query_aux and interval_aux");
    return BIVX_E_INVALID;
  }
  if (f->kind == BIVX_FILTER_SV2NL_INV && f->use_strand && !f->query_aux) {
    set_error("BIVX_FILTER_SV2NL_INV with use_strand needs query_aux");
    return BIVX_E_INVALID;
  }
  v.flt_kind = f->kind;
  v.flt_dist = f->max_dist;
  v.flt_strand = f->use_strand;
  v.flt_qaux = f->query_aux;
  v.flt_iaux = f->interval_aux;
  return 0;
}

int check_query_args(const bivx_index *idx, const void *qlow, const void *qhigh, size_t q, const char *who) {
  if (!idx || (q && (!qlow || !qhigh))) {
    set_error("%s: null argument", who);
    return BIVX_E_INVALID;
  }
  BIVX_NOT_SHARDED(idx, who);
  if (!idx->built) {
    set_error("%s: index not built (call bivx_build after the last append)", who);
    return BIVX_E_STATE;
  }
  return 0;
}

}  // namespace

namespace {

// Single-device index objects handed back by bivx_destroy, waiting for the next bivx_create (include/bivx.h,
// bivx_release_pooled). Never torn down at exit: the HIP runtime may be gone by then.
struct IndexPool {
  std::mutex m;
  std::vector<bivx_index *> idle;
};
IndexPool &index_pool() {
  static IndexPool *p = new IndexPool();
  return *p;
}
size_t env_size(const char *name, size_t dflt) {
  const char *e = std::getenv(name);
  if (!e) return dflt;
  const long long v = std::atoll(e);
  return v < 0 ? dflt : (size_t)v;
}

size_t device_bytes_of(const bivx_index *idx) {
  size_t b = idx->cap * 13 + idx->cache_bytes + idx->b_selfq.cap;
  for (const bivx_index::DevBuf *d : {&idx->b_se, &idx->b_table, &idx->b_seg, &idx->b_keys[0], &idx->b_keys[1], &idx->b_ids[0],
                                      &idx->b_ids[1], &idx->b_misc, &idx->b_radix, &idx->b_scalar, &idx->b_segof, &idx->b_astats})
    b += d->cap;
  for (auto &kv : idx->ws_of_stream) b += fused_workspace_bytes(0) + kv.second.self_cap;
  return b;
}

void destroy_now(bivx_index *idx) {
  DeviceGuard g(idx->device);
  if (idx->stream) (void)hipStreamSynchronize(idx->stream);
  (void)hipDeviceSynchronize();
  for (auto &kv : idx->ws_of_stream) {
    (void)hipFree(kv.second.p);
    (void)hipFree(kv.second.self_p);
  }
  if (idx->h_err) (void)hipHostFree(idx->h_err);
  for (auto &l : idx->lanes_all) {  // (lane 0 is idx->stream / h_err themselves and not in this list)
    if (l.stream) (void)hipStreamDestroy(l.stream);
    if (l.h_err) (void)hipHostFree(l.h_err);
  }
  for (auto &m : idx->mailboxes) (void)hipHostFree(m.first);
  idx->mailboxes.clear();
  drop_block_cache(idx);
  free_built(idx);
  release_build_blocks(idx);
  if (idx->h_scalars) (void)hipHostFree(idx->h_scalars);
  if (idx->h_stage) (void)hipHostFree(idx->h_stage);
  for (hipEvent_t ev : idx->ev_pending) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : idx->ev_free) (void)hipEventDestroy(ev);
  if (idx->astats_ev) (void)hipEventDestroy(idx->astats_ev);
  (void)hipFree(idx->d_chrom);
  (void)hipFree(idx->d_low);
  (void)hipFree(idx->d_high);
  (void)hipFree(idx->d_type);
  if (idx->stream) (void)hipStreamDestroy(idx->stream);
  delete idx;
}

// empties the index as bivx_clear does and parks it; false: it cannot be kept (the caller destroys it)
bool park(bivx_index *idx) {
  const size_t slots = env_size("BIVX_INDEX_POOL", 4);
  if (slots == 0 || device_bytes_of(idx) > (env_size("BIVX_INDEX_POOL_MB", 4096) << 20)) return false;
  DeviceGuard g(idx->device);
  if (!g.ok) return false;
  // nothing of the last owner's may still be running on the blocks the next owner will write
  if (hipStreamSynchronize(idx->stream) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return false;
  {
    volatile uint32_t *e = idx->h_err;
    if (e[kErrTimeout] | e[kErrWorkspace]) return false;  // (a failed launch may have left a workspace dirty)
    std::lock_guard<std::mutex> lock(idx->ws_mutex);
    for (auto &kv : idx->ws_of_stream)
      if (kv.second.needs_reset) return false;
  }
  for (hipEvent_t ev : idx->ev_pending) idx->ev_free.push_back(ev);
  idx->ev_pending.clear();
  {  // (the device was just waited for: the last owner's caller streams are done with the index, and may be gone)
    std::lock_guard<std::mutex> lock(idx->ws_mutex);
    idx->reader_streams.clear();
  }
  if (idx->typed && idx->d_type && idx->n) {
    if (hipMemsetAsync(idx->d_type, 0, idx->n, idx->stream) != hipSuccess || hipStreamSynchronize(idx->stream) != hipSuccess)
      return false;
  }
  free_built(idx);
  idx->n = 0;
  idx->typed = false;
  idx->astats_n = 0;
  idx->astats_ready = false;
  idx->built_n = 0;
  idx->build_ms = 0.0;
  idx->max_cell = idx->max_window = 0;
  idx->errors_reported.store(0);
  IndexPool &p = index_pool();
  std::lock_guard<std::mutex> lock(p.m);
  if (p.idle.size() >= slots) return false;
  p.idle.push_back(idx);
  return true;
}

}  // namespace

extern "C" {

uint32_t bivx_abi_version(void) { return BIVX_ABI_VERSION; }
const char *bivx_last_error(void) { return g_err; }

int bivx_create(bivx_index **out, int device) {
  if (!out) {
    set_error("bivx_create: null out");
    return BIVX_E_INVALID;
  }
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_error("bivx_create: no HIP device (this library has no CPU fallback)");
    return BIVX_E_HIP;
  }
  if (device < 0 || device >= ndev) {
    set_error("bivx_create: device %d out of range [0, %d)", device, ndev);
    return BIVX_E_INVALID;
  }
  {  // an index object a bivx_destroy left for this device?
    IndexPool &p = index_pool();
    std::lock_guard<std::mutex> lock(p.m);
    for (size_t k = p.idle.size(); k-- > 0;)
      if (p.idle[k]->device == device) {
        *out = p.idle[k];
        p.idle.erase(p.idle.begin() + (ptrdiff_t)k);
        return 0;
      }
  }
  bivx_index *idx = new (std::nothrow) bivx_index();
  if (!idx) {
    set_error("bivx_create: out of host memory");
    return BIVX_E_NOMEM;
  }
  idx->device = device;
  DeviceGuard g(device);
  if (!g.ok || hipStreamCreateWithFlags(&idx->stream, hipStreamNonBlocking) != hipSuccess) {
    set_error("bivx_create: cannot create a stream on device %d", device);
    delete idx;
    return BIVX_E_HIP;
  }
  void *herr = nullptr, *derr = nullptr;
  if (hipHostMalloc(&herr, kErrWords * sizeof(uint32_t), hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer(&derr, herr, 0) != hipSuccess) {
    set_error("bivx_create: cannot map the error block into device %d", device);
    if (herr) (void)hipHostFree(herr);
    (void)hipStreamDestroy(idx->stream);
    delete idx;
    return BIVX_E_HIP;
  }
  std::memset(herr, 0, kErrWords * sizeof(uint32_t));
  idx->h_err = static_cast<uint32_t *>(herr);
  idx->d_err = static_cast<uint32_t *>(derr);
  *out = idx;
  return 0;
}

int bivx_create_sharded(bivx_index **out, const int *devices, int ndev) {
  if (!out) {
    set_error("bivx_create_sharded: null out");
    return BIVX_E_INVALID;
  }
  *out = nullptr;
  ShardedState *st = nullptr;
  BIVX_TRY(sharded_create(&st, devices, ndev));
  bivx_index *idx = new (std::nothrow) bivx_index();
  if (!idx) {
    sharded_destroy(st);
    set_error("bivx_create_sharded: out of host memory");
    return BIVX_E_NOMEM;
  }
  idx->sharded = st;
  idx->device = devices[0];
  *out = idx;
  return 0;
}

int bivx_num_devices(const bivx_index *idx) { return !idx ? 0 : idx->sharded ? sharded_num_devices(idx->sharded) : 1; }

int bivx_device_of_chrom(const bivx_index *idx, uint32_t chrom) {
  if (!idx) return -1;
  return idx->sharded ? sharded_device_of_chrom(idx->sharded, chrom) : idx->device;
}

void bivx_destroy(bivx_index *idx) {
  if (!idx) return;
  if (idx->sharded) {
    sharded_destroy(idx->sharded);
    delete idx;
    return;
  }
  if (!park(idx)) destroy_now(idx);
}

void bivx_release_pooled(void) {
  std::vector<bivx_index *> all;
  {
    IndexPool &p = index_pool();
    std::lock_guard<std::mutex> lock(p.m);
    all.swap(p.idle);
  }
  for (bivx_index *idx : all) destroy_now(idx);
}

int bivx_device(const bivx_index *idx) { return idx ? idx->device : -1; }

int bivx_append(bivx_index *idx, const uint32_t *chrom, const uint32_t *low, const uint32_t *high, size_t n) {
  if (idx && idx->sharded) return sharded_append(idx->sharded, chrom, low, high, nullptr, n);
  return append_impl(idx, chrom, low, high, nullptr, n, hipMemcpyHostToDevice, idx ? idx->stream : nullptr);
}

int bivx_append_typed(bivx_index *idx, const uint32_t *chrom, const uint32_t *low, const uint32_t *high,
                      const uint8_t *svtype, size_t n) {
  if (idx && idx->sharded) return sharded_append(idx->sharded, chrom, low, high, svtype, n);
  return append_impl(idx, chrom, low, high, svtype, n, hipMemcpyHostToDevice, idx ? idx->stream : nullptr);
}

int bivx_append_dev(bivx_index *idx, const uint32_t *d_chrom, const uint32_t *d_low, const uint32_t *d_high,
                    size_t n, void *stream) {
  BIVX_NOT_SHARDED(idx, "bivx_append_dev");
  return append_impl(idx, d_chrom, d_low, d_high, nullptr, n, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream));
}

int bivx_append_typed_dev(bivx_index *idx, const uint32_t *d_chrom, const uint32_t *d_low, const uint32_t *d_high,
                          const uint8_t *d_svtype, size_t n, void *stream) {
  BIVX_NOT_SHARDED(idx, "bivx_append_typed_dev");
  return append_impl(idx, d_chrom, d_low, d_high, d_svtype, n, hipMemcpyDeviceToDevice,
                     static_cast<hipStream_t>(stream));
}

int bivx_clear(bivx_index *idx) {
  if (!idx) {
    set_error("bivx_clear: null index");
    return BIVX_E_INVALID;
  }
  if (idx->sharded) return sharded_clear(idx->sharded);
  BIVX_GUARD(idx);
  BIVX_HIP(hipStreamSynchronize(idx->stream));
  BIVX_TRY(wait_for_readers(idx));  // (the general self-overlap call reads the appended columns the next appends overwrite)
  // appends still on their way on caller streams write the columns and the statistics table the next appends start over
  for (hipEvent_t ev : idx->ev_pending) {
    BIVX_HIP(hipEventSynchronize(ev));
    idx->ev_free.push_back(ev);
  }
  idx->ev_pending.clear();
  free_built(idx);
  if (idx->typed && idx->d_type && idx->n) {
    // the slots are reused by later appends: an untyped append into them only clears type bytes while `typed` is set,
    // so bytes left here would label those intervals once a later typed append sets it again
    BIVX_HIP(hipDeviceSynchronize());  // (typed appends made with bivx_append_typed_dev ran on caller streams)
    BIVX_HIP(hipMemsetAsync(idx->d_type, 0, idx->n, idx->stream));
    BIVX_HIP(hipStreamSynchronize(idx->stream));
  }
  idx->n = 0;
  idx->typed = false;
  idx->astats_n = 0;
  idx->astats_ready = false;
  return 0;
}

int bivx_build(bivx_index *idx) {
  if (!idx) {
    set_error("bivx_build: null index");
    return BIVX_E_INVALID;
  }
  if (idx->sharded) return sharded_build(idx->sharded);
  if (idx->built && idx->built_n == idx->n) return 0;
  BIVX_GUARD(idx);
  const auto t0 = std::chrono::steady_clock::now();
  hipStream_t s = idx->stream;
  // appends made with bivx_append_dev on a caller stream must be complete before they are read: the build stream waits
  // for their events (the host does not)
  for (hipEvent_t ev : idx->ev_pending) {
    BIVX_HIP(hipStreamWaitEvent(s, ev, 0));
    idx->ev_free.push_back(ev);
  }
  idx->ev_pending.clear();
  // ... and calls that still read the arrays of the last build on caller streams must be through with them: the blocks
  // are grow-only, a rebuild of the same size overwrites them in place
  BIVX_TRY(wait_for_readers(idx));
  {
    std::lock_guard<std::mutex> lock(idx->ws_mutex);
    // on the build stream, which is synchronised before bivx_build returns (a plain hipMemset runs on the null
    // stream, which non-blocking streams do not wait for)
    // (a workspace per caller stream ever seen, 4.7 MB each: a caller that makes a stream per request would pile them up.
    // Nothing is running on them here — the readers were just waited for — so a large pile is simply dropped.)
    if (idx->ws_of_stream.size() > kMaxIdleWorkspaces) {
      for (auto &kv : idx->ws_of_stream) {
        (void)hipFree(kv.second.p);
        (void)hipFree(kv.second.self_p);
      }
      idx->ws_of_stream.clear();
    }
    for (auto &kv : idx->ws_of_stream) BIVX_HIP(hipMemsetAsync(kv.second.p, 0, fused_workspace_bytes(0), s));
  }
  free_built(idx);
  const size_t n = idx->n;
  if (!idx->h_scalars) BIVX_HIP(hipHostMalloc((void **)&idx->h_scalars, 64, hipHostMallocDefault));
  volatile uint32_t *hs = idx->h_scalars;

  // 1. largest chromosome id and svtype. Interval types (bivx_append_typed): the index is partitioned by (chromosome,
  // svtype) — the "virtual chromosome" chrom * ntypes + svtype takes the chromosome's place in everything below, so a
  // query that asks for one type walks only that type's segments and pays nothing per candidate (the svtype filter of
  // mapper.hpp:153-156, done by the layout instead of by three trees).
  // Without types the statistics pass (2.) finds the largest chromosome id itself: one pass over the columns and one
  // read-back instead of two, as long as the ids stay below bin_stats_auto_parts().
  // (b_scalar: [0] max chromosome, [1] max svtype, [2] largest cell, [3] the directory pass's list length; behind them
  // the one-pass statistics table, so that one copy brings both back)
  const uint32_t auto_ent = bin_stats_auto_parts() * kLenBins;
  const size_t auto_bytes = 256 + (size_t)auto_ent * sizeof(BinStats);
  BIVX_TRY(ensure_block(idx->b_scalar, auto_bytes));
  uint32_t *d_scalar = static_cast<uint32_t *>(idx->b_scalar.p);
  uint32_t max_chrom = 0, max_type = 0;
  bool have_stats = false;
  if (n && !idx->typed) {
    BIVX_TRY(ensure_stage(idx, auto_bytes));
    if (idx->astats_ready && idx->astats_n == n) {
      // the appends left the statistics behind (their passes are ordered before this stream by the events above)
      BIVX_HIP(hipMemsetAsync(d_scalar, 0, 16, s));  // (the build's own scalars: largest cell, the directory pass's list)
      BIVX_HIP(hipMemcpyAsync(idx->h_stage, idx->b_astats.p, auto_bytes, hipMemcpyDeviceToHost, s));
    } else {
      BIVX_TRY(launch_bin_stats_auto(idx->d_chrom, idx->d_low, idx->d_high, n,
                                     reinterpret_cast<BinStats *>(static_cast<char *>(idx->b_scalar.p) + 256), d_scalar, s));
      BIVX_HIP(hipMemcpyAsync(idx->h_stage, d_scalar, auto_bytes, hipMemcpyDeviceToHost, s));
    }
    BIVX_HIP(hipStreamSynchronize(s));
    max_chrom = static_cast<const uint32_t *>(idx->h_stage)[0];
    if (max_chrom == 0) {  // (no chromosome id beyond the pass's table: the largest is its last non-empty row)
      have_stats = true;
      const BinStats *rows = reinterpret_cast<const BinStats *>(static_cast<const char *>(idx->h_stage) + 256);
      for (uint32_t e = 0; e < auto_ent; ++e)
        if (rows[e].count) max_chrom = e / kLenBins;
    }
  } else if (n) {
    BIVX_HIP(hipMemsetAsync(d_scalar, 0, 16, s));  // the maxima start at zero
    BIVX_TRY(launch_max_chrom_type(idx->d_chrom, idx->d_type, n, d_scalar, s));
    BIVX_HIP(hipMemcpyAsync(idx->h_scalars, d_scalar, 8, hipMemcpyDeviceToHost, s));
    BIVX_HIP(hipStreamSynchronize(s));
    max_chrom = hs[0];
    max_type = hs[1];
  }
  if (n && max_chrom >= BIVX_MAX_CHROMS) {
    set_error("chromosome id %u exceeds BIVX_MAX_CHROMS", max_chrom);
    return BIVX_E_RANGE;
  }
  const uint32_t nchrom = n ? max_chrom + 1 : 0;
  const uint32_t ntypes = max_type + 1;
  if ((uint64_t)nchrom * ntypes > BIVX_MAX_CHROMS) {
    set_error("%u chromosome ids x %u interval types exceed BIVX_MAX_CHROMS", nchrom, ntypes);
    return BIVX_E_RANGE;
  }
  const uint8_t *d_type = ntypes > 1 ? idx->d_type : nullptr;  // what partitions the index besides the chromosome
  const uint32_t nvchrom = nchrom * ntypes;

  // 2. per (partition, length bin) statistics -> host
  std::vector<BinStats> st((size_t)nvchrom * kLenBins);
  ClassPlan plan;
  if (have_stats) {
    std::memcpy(st.data(), static_cast<const char *>(idx->h_stage) + 256, st.size() * sizeof(BinStats));
  } else if (n) {
    const size_t bytes = st.size() * sizeof(BinStats);
    BIVX_TRY(ensure_block(idx->b_misc, bytes));
    BinStats *d_stats = static_cast<BinStats *>(idx->b_misc.p);
    BIVX_TRY(launch_bin_stats(idx->d_chrom, d_type, ntypes, idx->d_low, idx->d_high, n, nvchrom, d_stats, s));
    const bool staged = bytes <= ((size_t)1 << 20);  // (through pinned memory: no staging copy inside the runtime)
    if (staged) BIVX_TRY(ensure_stage(idx, bytes));
    BIVX_HIP(hipMemcpyAsync(staged ? idx->h_stage : (void *)st.data(), d_stats, bytes, hipMemcpyDeviceToHost, s));
    BIVX_HIP(hipStreamSynchronize(s));
    if (staged) std::memcpy(st.data(), idx->h_stage, bytes);
  }
  // 3. length classes, segment descriptors
  BIVX_TRY(plan_classes(st, nvchrom, n, plan));
  const uint32_t nseg = (uint32_t)plan.segs.size();

  // The small tables of the index in ONE device block, put together in pinned memory and uploaded by one copy:
  //   [SegDesc x nseg | (keybase, base) x nseg | chromosome x nseg | segment ranges | (partition, bin) -> segment]
  // segment ranges per chromosome, one row per interval type: row 0 = every type (the types of a chromosome are
  // neighbours in the segment order), row t = type t, and one all-empty row behind them for types the index lacks
  const size_t nseg1 = nseg ? nseg : 1;
  const size_t nrng = (size_t)(ntypes + 1) * (nchrom ? nchrom : 1);
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t off_key = up(nseg1 * sizeof(SegDesc));
  const size_t off_chr = off_key + up(nseg1 * sizeof(uint2));
  const size_t off_rng = off_chr + up(nseg1 * sizeof(uint32_t));
  const size_t off_b2s = off_rng + up(nrng * sizeof(uint2));
  const size_t meta_bytes = off_b2s + up(plan.bin2seg.size() * sizeof(uint32_t));
  BIVX_TRY(ensure_block(idx->b_seg, meta_bytes));
  BIVX_TRY(ensure_stage(idx, meta_bytes));
  char *hm = static_cast<char *>(idx->h_stage), *dm = static_cast<char *>(idx->b_seg.p);
  idx->d_seg = reinterpret_cast<SegDesc *>(dm);
  uint2 *d_segkey = reinterpret_cast<uint2 *>(dm + off_key);
  idx->d_seg_chrom = reinterpret_cast<uint32_t *>(dm + off_chr);
  idx->d_chrom_rng = reinterpret_cast<uint2 *>(dm + off_rng);
  uint32_t *d_bin2seg = reinterpret_cast<uint32_t *>(dm + off_b2s);
  if (nseg) {
    std::memcpy(hm, plan.segs.data(), (size_t)nseg * sizeof(SegDesc));
    std::memcpy(hm + off_key, plan.segkey.data(), (size_t)nseg * sizeof(uint2));
  }
  uint32_t *seg_chrom = reinterpret_cast<uint32_t *>(hm + off_chr);
  for (uint32_t c = 0; c < nvchrom; ++c)
    for (uint32_t k = plan.chrom_seg[c]; k < plan.chrom_seg[c + 1]; ++k) seg_chrom[k] = c / ntypes;
  uint2 *rng = reinterpret_cast<uint2 *>(hm + off_rng);
  std::fill(rng, rng + nrng, make_uint2(0u, 0u));
  std::vector<uint32_t> max_segs(ntypes + 1, 0u);
  for (uint32_t c = 0; c < nchrom; ++c) {
    const uint32_t *cs = plan.chrom_seg.data() + (size_t)c * ntypes;
    rng[c] = make_uint2(cs[0], cs[ntypes] - cs[0]);
    max_segs[0] = std::max(max_segs[0], cs[ntypes] - cs[0]);
    for (uint32_t t = 1; t < ntypes; ++t) {
      rng[(size_t)t * nchrom + c] = make_uint2(cs[t], cs[t + 1] - cs[t]);
      max_segs[t] = std::max(max_segs[t], cs[t + 1] - cs[t]);
    }
  }
  if (!plan.bin2seg.empty()) std::memcpy(hm + off_b2s, plan.bin2seg.data(), plan.bin2seg.size() * sizeof(uint32_t));
  BIVX_HIP(hipMemcpyAsync(dm, hm, meta_bytes, hipMemcpyHostToDevice, s));

  uint32_t max_cell = 0, order_shift = 0;
  if (n) {
    // 4. stable sort to (segment, low, id). One 32-bit key holds (segment, low) while the segments' coordinate spans add
    // up to less than 2^32 (see ClassPlan::segkey): then ONE sort does it and `low` comes back out of the sorted key.
    // Otherwise two: by low, then by the segment of the ids as the first sort left them.
    for (int k = 0; k < 2; ++k) {
      BIVX_TRY(ensure_block(idx->b_keys[k], n * 4));
      BIVX_TRY(ensure_block(idx->b_ids[k], (n + 2) * 4));  // (two spare ids: query lanes read ids two at a time)
    }
    // (histogram scratch of the sort, then the directory pass's short list of long empty stretches)
    BIVX_TRY(ensure_block(idx->b_radix, std::max(radix_scratch_bytes(n), finalize_gap_bytes(plan.nentries, nseg))));
    const bool dense = plan.key_span <= 0xFFFFFFFFull && !std::getenv("BIVX_BUILD_TWO_STAGE");  // (env: test knob)
    // 5. sorted (low, high) pairs and packed (record, id) pairs
    // two spare slots each: query lanes read pairs two at a time (16 B), so the pair holding the last slot may reach
    // one slot past the end
    // (one block: k_query_pipe_ms addresses both arrays with 32-bit offsets from one base)
    const size_t se_bytes = ((n + 2) * sizeof(uint2) + 255) & ~(size_t)255;
    BIVX_TRY(ensure_block(idx->b_se, 2 * se_bytes));
    idx->d_se = static_cast<uint2 *>(idx->b_se.p);
    idx->d_rec = reinterpret_cast<uint2 *>(static_cast<char *>(idx->b_se.p) + se_bytes);
    // 6. ... and the bucket directory, by the same pass (+3 spare entries: query lanes read entries four at a time)
    BIVX_TRY(ensure_block(idx->b_table, ((size_t)plan.nentries + 3) * 4));
    idx->d_table = static_cast<uint32_t *>(idx->b_table.p);
    // The dense-key sort leaves out the key's low `skip` bits when the plan says that saves a radix pass
    // (ClassPlan::order_shift): the slots then come out ordered by directory cell, append order inside a cell.
    auto sort_and_finalize = [&](int skip) -> int {
      uint32_t *kA = static_cast<uint32_t *>(idx->b_keys[0].p), *kB = static_cast<uint32_t *>(idx->b_keys[1].p);
      uint32_t *vA = static_cast<uint32_t *>(idx->b_ids[0].p), *vB = static_cast<uint32_t *>(idx->b_ids[1].p);
      if (dense) {
        BIVX_TRY(launch_make_keys(kBuildKeyDense, idx->d_chrom, d_type, ntypes, idx->d_low, idx->d_high, n, d_bin2seg, d_segkey,
                                  nullptr, nullptr, kA, idx->b_radix.p, s, skip));
        BIVX_TRY(radix_sort_pairs(&kA, &vA, &kB, &vB, n, bits_for((uint32_t)(plan.key_span - 1)), idx->b_radix.p, true, true, s,
                                  skip));
      } else {
        // (every interval's segment, in append order: the second sort's keys are one gather of it)
        BIVX_TRY(ensure_block(idx->b_segof, n * 4));
        uint32_t *d_seg_of = static_cast<uint32_t *>(idx->b_segof.p);
        BIVX_TRY(launch_make_keys(kBuildKeyLow, idx->d_chrom, d_type, ntypes, idx->d_low, idx->d_high, n, d_bin2seg, d_segkey,
                                  nullptr, d_seg_of, kA, idx->b_radix.p, s));
        BIVX_TRY(radix_sort_pairs(&kA, &vA, &kB, &vB, n, bits_for(plan.max_low), idx->b_radix.p, true, true, s));
        if (nseg > 1) {
          BIVX_TRY(launch_make_keys(kBuildKeySegOfId, idx->d_chrom, d_type, ntypes, idx->d_low, idx->d_high, n, d_bin2seg,
                                    d_segkey, vA, d_seg_of, kA, nullptr, s));
          BIVX_TRY(radix_sort_pairs(&kA, &vA, &kB, &vB, n, bits_for(nseg - 1), idx->b_radix.p, false, false, s));
        }
      }
      idx->d_id = vA;  // (one of the two id blocks; it stays the index's until the next build)
      BIVX_TRY(launch_finalize(dense ? kA : nullptr, idx->d_id, idx->d_low, idx->d_high, idx->d_seg, d_segkey, nseg, idx->d_se,
                               idx->d_rec, idx->d_table, plan.nentries, idx->b_radix.p, d_scalar + 3, d_scalar + 2, n, s));
      BIVX_HIP(hipMemcpyAsync(idx->h_scalars + 2, d_scalar + 2, 4, hipMemcpyDeviceToHost, s));
      return 0;
    };
    int skip = dense ? (int)plan.order_shift : 0;
    BIVX_TRY(sort_and_finalize(skip));
    if (skip > 0) {
      // A crowded cell (positional hotspot: many intervals starting inside one cell) is what long windows get trimmed
      // inside of, by a search on low (query_device.h, wave_lower_bound_low) — that wants the cell's slots ordered by low:
      // such an index is sorted again on every bit (it is the exception; the usual index keeps a handful of slots per cell)
      BIVX_HIP(hipStreamSynchronize(s));
      if (hs[2] > kCrowdedCell) {
        BIVX_HIP(hipMemsetAsync(d_scalar + 2, 0, 8, s));  // (largest cell, the directory pass's list length)
        skip = 0;
        BIVX_TRY(sort_and_finalize(0));
      }
    }
    order_shift = (uint32_t)skip;
  }
  BIVX_HIP(hipStreamSynchronize(s));
  if (n) max_cell = hs[2];
  idx->nchrom = nchrom;
  idx->nseg = nseg;
  idx->ntypes = ntypes;
  idx->max_segs = std::move(max_segs);
  idx->nentries = plan.nentries;
  idx->max_cell = max_cell;
  idx->max_window = plan.max_window;
  idx->order_shift = order_shift;
  idx->built = true;
  idx->built_n = n;
  idx->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}

int bivx_is_built(const bivx_index *idx) {
  if (idx && idx->sharded) return sharded_is_built(idx->sharded);
  return idx && idx->built && idx->built_n == idx->n;
}
size_t bivx_size(const bivx_index *idx) { return !idx ? 0 : idx->sharded ? sharded_size(idx->sharded) : idx->n; }
uint32_t bivx_num_chroms(const bivx_index *idx) {
  return !idx ? 0 : idx->sharded ? sharded_num_chroms(idx->sharded) : idx->nchrom;
}
uint32_t bivx_num_types(const bivx_index *idx) {
  return !idx ? 0 : idx->sharded ? sharded_num_types(idx->sharded) : idx->ntypes;
}

int bivx_get_svtypes(const bivx_index *idx, const uint32_t *ids, size_t n, uint8_t *svtype_out) {
  if (!idx || (n && (!ids || !svtype_out))) {
    set_error("bivx_get_svtypes: null argument");
    return BIVX_E_INVALID;
  }
  if (n == 0) return 0;
  if (idx->sharded) return sharded_get_svtypes(idx->sharded, ids, n, svtype_out);
  BIVX_GUARD(idx);
  hipStream_t s = idx->stream;
  TempPool tmp(idx);
  uint32_t *d_ids = nullptr;
  uint8_t *d_t = nullptr;
  BIVX_TRY(tmp.alloc(&d_ids, n));
  BIVX_TRY(tmp.alloc(&d_t, n));
  BIVX_HIP(hipMemcpyAsync(d_ids, ids, n * 4, hipMemcpyHostToDevice, s));
  BIVX_TRY(launch_gather_u8(idx->typed ? idx->d_type : nullptr, d_ids, n, idx->n, d_t, s));
  BIVX_HIP(hipMemcpyAsync(svtype_out, d_t, n, hipMemcpyDeviceToHost, s));
  BIVX_HIP(hipStreamSynchronize(s));
  return 0;
}

int bivx_get_intervals(const bivx_index *idx, const uint32_t *ids, size_t n, uint32_t *chrom_out, uint32_t *low_out,
                       uint32_t *high_out) {
  if (!idx || (n && !ids)) {
    set_error("bivx_get_intervals: null argument");
    return BIVX_E_INVALID;
  }
  if (n == 0) return 0;
  if (idx->sharded) return sharded_get_intervals(idx->sharded, ids, n, chrom_out, low_out, high_out);
  BIVX_GUARD(idx);
  hipStream_t s = idx->stream;
  TempPool tmp(idx);
  uint32_t *d_ids = nullptr, *d_c = nullptr, *d_l = nullptr, *d_h = nullptr;
  BIVX_TRY(tmp.alloc(&d_ids, n));
  if (chrom_out) BIVX_TRY(tmp.alloc(&d_c, n));
  if (low_out) BIVX_TRY(tmp.alloc(&d_l, n));
  if (high_out) BIVX_TRY(tmp.alloc(&d_h, n));
  BIVX_HIP(hipMemcpyAsync(d_ids, ids, n * 4, hipMemcpyHostToDevice, s));
  BIVX_TRY(launch_gather_intervals(idx->d_chrom, idx->d_low, idx->d_high, d_ids, n, idx->n, d_c, d_l, d_h, s));
  if (chrom_out) BIVX_HIP(hipMemcpyAsync(chrom_out, d_c, n * 4, hipMemcpyDeviceToHost, s));
  if (low_out) BIVX_HIP(hipMemcpyAsync(low_out, d_l, n * 4, hipMemcpyDeviceToHost, s));
  if (high_out) BIVX_HIP(hipMemcpyAsync(high_out, d_h, n * 4, hipMemcpyDeviceToHost, s));
  BIVX_HIP(hipStreamSynchronize(s));
  return 0;
}

namespace {
int query_single_pass(const bivx_index *idx, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                      size_t q, const bivx_filter *filter, int sort_by_id, uint64_t *d_offsets, uint32_t *d_counts,
                      uint32_t *d_hit_ids, uint64_t hit_capacity, uint64_t *d_total, void *d_workspace,
                      size_t workspace_bytes, void *stream, const char *who);
}

int bivx_count_dev(const bivx_index *idx, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                   size_t q, uint64_t *d_offsets, void *stream) {
  return bivx_count_dev_f(idx, d_qchrom, d_qlow, d_qhigh, q, nullptr, d_offsets, stream);
}

int bivx_count_dev_f(const bivx_index *idx, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                     size_t q, const bivx_filter *filter, uint64_t *d_offsets, void *stream) {
  // One launch of the single-pass kernel with a zero-capacity hit buffer: it counts, chains the prefix across
  // workgroups and writes the offsets, and skips the output phase. (The counts + three-launch scan it replaces
  // took 55 us at config 2, this takes 40.)
  return query_single_pass(idx, d_qchrom, d_qlow, d_qhigh, q, filter, 0, d_offsets, nullptr, nullptr, 0, nullptr,
                           nullptr, 0, stream, "bivx_count_dev");
}

int bivx_fill_dev(const bivx_index *idx, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                  size_t q, const uint64_t *d_offsets, uint32_t *d_hit_ids, void *stream) {
  return bivx_fill_dev_f(idx, d_qchrom, d_qlow, d_qhigh, q, nullptr, d_offsets, d_hit_ids, stream);
}

int bivx_fill_dev_f(const bivx_index *idx, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                    size_t q, const bivx_filter *filter, const uint64_t *d_offsets, uint32_t *d_hit_ids, void *stream) {
  BIVX_TRY(check_query_args(idx, d_qlow, d_qhigh, q, "bivx_fill_dev"));
  IndexView view;
  BIVX_TRY(view_with_filter(idx, filter, view));
  if (q && !d_offsets) {
    set_error("bivx_fill_dev: null d_offsets");
    return BIVX_E_INVALID;
  }
  BIVX_GUARD(idx);
  note_reader(idx, static_cast<hipStream_t>(stream));
  return launch_fill(view, d_qchrom, d_qlow, d_qhigh, q, d_offsets, d_hit_ids, static_cast<hipStream_t>(stream));
}

size_t bivx_query_workspace_bytes(size_t q) { return fused_workspace_bytes(q); }

int bivx_query_dev(const bivx_index *idx, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                   size_t q, uint64_t *d_offsets, uint32_t *d_hit_ids, uint64_t hit_capacity, void *d_workspace,
                   size_t workspace_bytes, void *stream) {
  return bivx_query_dev_f(idx, d_qchrom, d_qlow, d_qhigh, q, nullptr, d_offsets, d_hit_ids, hit_capacity, d_workspace,
                          workspace_bytes, stream);
}

int bivx_query_dev_f(const bivx_index *idx, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                     size_t q, const bivx_filter *filter, uint64_t *d_offsets, uint32_t *d_hit_ids,
                     uint64_t hit_capacity, void *d_workspace, size_t workspace_bytes, void *stream) {
  return bivx_query_dev_s(idx, d_qchrom, d_qlow, d_qhigh, q, filter, 0, d_offsets, d_hit_ids, hit_capacity, d_workspace,
                          workspace_bytes, stream);
}

namespace {
// the single-pass entry points; d_counts != nullptr selects the unordered begin/count output
int query_single_pass(const bivx_index *idx, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                      size_t q, const bivx_filter *filter, int sort_by_id, uint64_t *d_offsets, uint32_t *d_counts,
                      uint32_t *d_hit_ids, uint64_t hit_capacity, uint64_t *d_total, void *d_workspace,
                      size_t workspace_bytes, void *stream, const char *who) {
  BIVX_TRY(check_query_args(idx, d_qlow, d_qhigh, q, who));
  IndexView view;
  BIVX_TRY(view_with_filter(idx, filter, view));
  if ((q && !d_offsets) || (!d_counts && !d_offsets) || (hit_capacity && !d_hit_ids)) {
    set_error("%s: null argument", who);
    return BIVX_E_INVALID;
  }
  if (d_workspace && workspace_bytes < fused_workspace_bytes(q)) {
    set_error("%s: workspace too small (%zu < %zu)", who, workspace_bytes, fused_workspace_bytes(q));
    return BIVX_E_INVALID;
  }
  BIVX_GUARD(idx);
  hipStream_t s = static_cast<hipStream_t>(stream);
  note_reader(idx, s);
  bool self_clean = false;
  if (!d_workspace) {  // the index's own per-stream workspace: no memset in front of the kernel
    std::lock_guard<std::mutex> lock(idx->ws_mutex);
    auto it = idx->ws_of_stream.find(s);
    if (it == idx->ws_of_stream.end() || it->second.needs_reset) {
      // (re)initialising a workspace is a memset that must run exactly once, before the first launch: inside a
      // stream capture it would become a node that a replay repeats (or that is never run at all)
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (s && hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
        set_error("%s: the first call on a stream (and the first after an error) allocates and clears the index's "
                  "prefix workspace and cannot be captured: make one call on this stream before the capture", who);
        return BIVX_E_STATE;
      }
      if (it == idx->ws_of_stream.end()) {
        void *p = nullptr;
        BIVX_HIP(dev_malloc(&p, fused_workspace_bytes(q)));
        bivx_index::Workspace w;
        w.p = p;
        it = idx->ws_of_stream.emplace(s, w).first;
      }
      BIVX_HIP(hipMemsetAsync(it->second.p, 0, fused_workspace_bytes(q), s));  // ordered before the kernel: same stream
      it->second.needs_reset = false;
    }
    d_workspace = it->second.p;
    self_clean = true;
  }
  return launch_query_fused(view, d_qchrom, d_qlow, d_qhigh, q, d_offsets, d_hit_ids, hit_capacity, d_workspace,
                            self_clean, sort_by_id != 0, s, d_counts, d_total);
}
}  // namespace

int bivx_query_dev_s(const bivx_index *idx, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                     size_t q, const bivx_filter *filter, int sort_by_id, uint64_t *d_offsets, uint32_t *d_hit_ids,
                     uint64_t hit_capacity, void *d_workspace, size_t workspace_bytes, void *stream) {
  return query_single_pass(idx, d_qchrom, d_qlow, d_qhigh, q, filter, sort_by_id, d_offsets, nullptr, d_hit_ids,
                           hit_capacity, nullptr, d_workspace, workspace_bytes, stream, "bivx_query_dev");
}

int bivx_query_dev_u(const bivx_index *idx, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                     size_t q, const bivx_filter *filter, uint64_t *d_begin, uint32_t *d_count, uint32_t *d_hit_ids,
                     uint64_t hit_capacity, uint64_t *d_total, void *d_workspace, size_t workspace_bytes,
                     void *stream) {
  if (!d_total || (q && !d_count)) {
    set_error("bivx_query_dev_u: null argument");
    return BIVX_E_INVALID;
  }
  uint32_t dummy = 0;
  return query_single_pass(idx, d_qchrom, d_qlow, d_qhigh, q, filter, 0, d_begin, q ? d_count : &dummy, d_hit_ids,
                           hit_capacity, d_total, d_workspace, workspace_bytes, stream, "bivx_query_dev_u");
}

int bivx_self_overlaps_dev(const bivx_index *idx, int sort_by_id, uint64_t *d_offsets, uint32_t *d_hit_ids,
                           uint64_t hit_capacity, void *stream) {
  if (!idx || !d_offsets || (hit_capacity && !d_hit_ids)) {
    set_error("bivx_self_overlaps_dev: null argument");
    return BIVX_E_INVALID;
  }
  BIVX_NOT_SHARDED(idx, "bivx_self_overlaps_dev");
  if (!idx->built || idx->built_n != idx->n) {
    set_error("bivx_self_overlaps_dev: index not built (call bivx_build after the last append)");
    return BIVX_E_STATE;
  }
  const size_t n = idx->built_n;
  const IndexView view = view_of(idx);
  if (n == 0 || !self_overlaps_eligible(view, n))  // the general path: the appended columns are the batch
    return query_single_pass(idx, idx->d_chrom, idx->d_low, idx->d_high, n, nullptr, sort_by_id, d_offsets, nullptr,
                             d_hit_ids, hit_capacity, nullptr, nullptr, 0, stream, "bivx_self_overlaps_dev");
  BIVX_GUARD(idx);
  hipStream_t s = static_cast<hipStream_t>(stream);
  note_reader(idx, s);
  // the intervals as queries in slot order, once per build (on the index's stream, finished before anybody uses them)
  {
    std::lock_guard<std::mutex> lock(idx->self_mutex);
    if (!idx->selfq_valid) {
      BIVX_TRY(ensure_block(idx->b_selfq, n * 12));
      uint32_t *q = static_cast<uint32_t *>(idx->b_selfq.p);
      BIVX_TRY(launch_self_queries(idx->d_se, idx->d_seg, idx->d_seg_chrom, idx->nseg, n, q, q + n, q + 2 * n, idx->stream));
      BIVX_HIP(hipStreamSynchronize(idx->stream));
      idx->selfq_valid = true;
    }
  }
  const uint32_t *q = static_cast<const uint32_t *>(idx->b_selfq.p);
  void *ws = nullptr, *self_p = nullptr;
  // per stream: [scan scratch | per id: list length << kSelfPosBits | where it begins, u64 x n | the lists in slot order]
  const size_t scan_bytes = (scan_scratch_bytes(n) + 255) & ~(size_t)255;
  const size_t src_bytes = (n * 8 + 255) & ~(size_t)255;
  // (+ 64 ids: k_permute_lines reads the lists in whole aligned lines, the last one may reach past the last id)
  const uint64_t tmp_cap = hit_capacity ? hit_capacity + 64 : 0;
  if (tmp_cap > kSelfPosMask) {
    set_error("bivx_self_overlaps_dev: a hit capacity beyond 2^38 ids");
    return BIVX_E_RANGE;
  }
  const size_t self_bytes = scan_bytes + src_bytes + (size_t)tmp_cap * 4;
  {
    std::lock_guard<std::mutex> lock(idx->ws_mutex);
    auto it = idx->ws_of_stream.find(s);
    const bool fresh = it == idx->ws_of_stream.end();
    if (fresh || it->second.needs_reset || it->second.self_cap < self_bytes) {
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (s && hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
        set_error("bivx_self_overlaps_dev: the first call on a stream allocates the index's workspaces and cannot be "
                  "captured: make one call on this stream before the capture");
        return BIVX_E_STATE;
      }
      if (fresh) {
        void *p = nullptr;
        BIVX_HIP(dev_malloc(&p, fused_workspace_bytes(n)));
        bivx_index::Workspace w;
        w.p = p;
        w.needs_reset = true;
        it = idx->ws_of_stream.emplace(s, w).first;
      }
      if (it->second.needs_reset) {
        BIVX_HIP(hipMemsetAsync(it->second.p, 0, fused_workspace_bytes(n), s));
        it->second.needs_reset = false;
      }
      if (it->second.self_cap < self_bytes) {
        BIVX_HIP(hipStreamSynchronize(s));  // (an earlier call on this stream may still use the old block)
        (void)hipFree(it->second.self_p);
        it->second.self_p = nullptr;
        it->second.self_cap = 0;
        BIVX_HIP(dev_malloc(&it->second.self_p, self_bytes));
        it->second.self_cap = self_bytes;
      }
    }
    ws = it->second.p;
    self_p = it->second.self_p;
  }
  char *base = static_cast<char *>(self_p);
  void *scan_scr = base;
  uint64_t *d_src = reinterpret_cast<uint64_t *>(base + scan_bytes);
  uint32_t *d_tmp = reinterpret_cast<uint32_t *>(base + scan_bytes + src_bytes);
  // one pass in slot order (lists back to back in d_tmp; per id ONE word: the list's length above where it begins), offsets =
  // a scan of the lengths, then the lists are gathered into id order
  BIVX_TRY(launch_self_overlaps(view, q, q + n, q + 2 * n, idx->d_id, n, d_src, d_offsets, hit_capacity ? d_tmp : nullptr,
                                hit_capacity, static_cast<uint64_t *>(ws), true, s));
  if (!hit_capacity) {
    BIVX_TRY(exclusive_scan_lengths_u64(d_src, d_offsets, n, scan_scr, s));
  } else {
    bool sorted = false;  // (the line-wise gather makes the offsets itself, and orders the lists while they pass through LDS)
    BIVX_TRY(launch_permute_lists(d_offsets, d_src, d_tmp, d_hit_ids, n, hit_capacity, sort_by_id != 0, &sorted, scan_scr, s));
    if (sort_by_id && !sorted) BIVX_TRY(launch_sort_hits(d_offsets, d_hit_ids, n, hit_capacity, s));
  }
  return 0;
}

int bivx_stream_status(const bivx_index *idx, void *stream) {
  if (!idx) {
    set_error("bivx_stream_status: null index");
    return BIVX_E_INVALID;
  }
  BIVX_NOT_SHARDED(idx, "bivx_stream_status");
  BIVX_GUARD(idx);
  BIVX_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  return report_device_errors(idx, "bivx_stream_status");
}

int bivx_debug_corrupt_workspace(const bivx_index *idx, void *stream) {
  if (!idx) {
    set_error("bivx_debug_corrupt_workspace: null index");
    return BIVX_E_INVALID;
  }
  BIVX_NOT_SHARDED(idx, "bivx_debug_corrupt_workspace");
  BIVX_GUARD(idx);
  hipStream_t s = static_cast<hipStream_t>(stream);
  std::lock_guard<std::mutex> lock(idx->ws_mutex);
  auto it = idx->ws_of_stream.find(s);
  if (it == idx->ws_of_stream.end()) {
    set_error("bivx_debug_corrupt_workspace: no workspace for this stream yet");
    return BIVX_E_STATE;
  }
  BIVX_HIP(hipMemsetAsync(it->second.p, 0x7F, sizeof(uint32_t), s));  // ticket word: far beyond any grid
  return 0;
}

const char *bivx_query_kernel_name(const bivx_index *idx, size_t q, uint64_t hit_capacity, int sort_by_id,
                                   const bivx_filter *filter) {
  if (!idx || idx->sharded || !idx->built) return "";
  IndexView view;
  if (view_with_filter(idx, filter, view) != 0) return "";
  if (pipe_eligible(view, q, hit_capacity, sort_by_id != 0, false)) return "k_query_pipe";
  const bool ms = pipe_ms_eligible(view, q, hit_capacity, false);
  if (pipe_dense_eligible(view, q, hit_capacity, sort_by_id != 0, false))
    return ms ? "k_query_pipe_dense|k_query_pipe_ms" : "k_query_pipe_dense|k_query_fused";
  return ms ? "k_query_pipe_ms" : "k_query_fused";
}

int bivx_sort_hits_dev(const bivx_index *idx, const uint64_t *d_offsets, uint32_t *d_hit_ids, size_t q, void *stream) {
  if (!idx || (q && !d_offsets)) {
    set_error("bivx_sort_hits_dev: null argument");
    return BIVX_E_INVALID;
  }
  BIVX_NOT_SHARDED(idx, "bivx_sort_hits_dev");
  BIVX_GUARD(idx);
  return launch_sort_hits(d_offsets, d_hit_ids, q, ~0ull, static_cast<hipStream_t>(stream));
}

int bivx_any_dev(const bivx_index *idx, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                 size_t q, uint32_t *d_first_id, void *stream) {
  BIVX_TRY(check_query_args(idx, d_qlow, d_qhigh, q, "bivx_any_dev"));
  if (q && !d_first_id) {
    set_error("bivx_any_dev: null output");
    return BIVX_E_INVALID;
  }
  BIVX_GUARD(idx);
  note_reader(idx, static_cast<hipStream_t>(stream));
  return launch_any(view_of(idx), d_qchrom, d_qlow, d_qhigh, q, d_first_id, static_cast<hipStream_t>(stream));
}

// ---- host-pointer convenience entry points -----------------------------------------------------------------

namespace {
struct DevQueries {
  uint32_t *c = nullptr, *lo = nullptr, *hi = nullptr;
};
int upload_queries(TempPool &tmp, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh, size_t q,
                   hipStream_t s, DevQueries &d) {
  BIVX_TRY(tmp.alloc(&d.lo, q));
  BIVX_TRY(tmp.alloc(&d.hi, q));
  BIVX_HIP(hipMemcpyAsync(d.lo, qlow, q * 4, hipMemcpyHostToDevice, s));
  BIVX_HIP(hipMemcpyAsync(d.hi, qhigh, q * 4, hipMemcpyHostToDevice, s));
  if (qchrom) {
    BIVX_TRY(tmp.alloc(&d.c, q));
    BIVX_HIP(hipMemcpyAsync(d.c, qchrom, q * 4, hipMemcpyHostToDevice, s));
  }
  return 0;
}
}  // namespace

namespace {
// A handful of queries (what reference-style code sends: one find_overlaps per record) in ONE device round trip:
// queries packed into one upload, one launch of the single-pass kernel into a buffer sized for kSmallBatchIds hits
// per query on average, one download of offsets and ids. A result that does not fit reports kSmallBatchOverflow
// and the caller takes the count-then-fill path.
constexpr size_t kSmallBatch = 2048, kSmallBatchIds = 32;
constexpr int kSmallBatchOverflow = 1;
// A mailbox: kMailboxBytes of pinned host memory the device can address. Taken from the index's stock (or allocated),
// given back when the call is over.
constexpr size_t kMailboxQueries = 64, kMailboxBytes = 32u << 10;
struct Mailbox {
  const bivx_index *idx;
  void *host = nullptr, *dev = nullptr;
  explicit Mailbox(const bivx_index *owner) : idx(owner) {
    {
      std::lock_guard<std::mutex> lock(idx->cache_mutex);
      if (!idx->mailboxes.empty()) {
        host = idx->mailboxes.back().first;
        dev = idx->mailboxes.back().second;
        idx->mailboxes.pop_back();
      }
    }
    if (!host) {
      if (hipHostMalloc(&host, kMailboxBytes, hipHostMallocMapped) != hipSuccess ||
          hipHostGetDevicePointer(&dev, host, 0) != hipSuccess) {
        if (host) (void)hipHostFree(host);
        host = dev = nullptr;
      }
    }
  }
  ~Mailbox() {
    if (!host) return;
    std::lock_guard<std::mutex> lock(idx->cache_mutex);
    idx->mailboxes.emplace_back(host, dev);
  }
};

// A handful of queries (the facade's per-record calls, interval_tree.hpp:306-334 under mapper.hpp:218): no copies at all.
// The queries are written into a mailbox, the single-pass kernel reads them and writes offsets and ids through the
// same mapping, and the stream is synchronised once. Returns kSmallBatchOverflow if the ids did not fit.
int find_overlaps_tiny(const bivx_index *idx, const LaneLease &lease, const uint32_t *qchrom, const uint32_t *qlow,
                       const uint32_t *qhigh, size_t q, const bivx_filter *filter, int sort_by_id, uint64_t *offsets_out,
                       uint32_t **hit_ids_out);

int find_overlaps_small(const bivx_index *idx, const LaneLease &lease, const uint32_t *qchrom, const uint32_t *qlow,
                        const uint32_t *qhigh, size_t q, const bivx_filter *filter, int sort_by_id, uint64_t *offsets_out,
                        uint32_t **hit_ids_out) {
  hipStream_t s = lease.lane.stream;
  const size_t nq_words = (qchrom ? 3 : 2) * q;
  const size_t cap = kSmallBatchIds * q + 1024;
  const size_t off_words = 2 * (q + 1);                     // u64 offsets, as u32 words
  TempPool tmp(idx, s);
  uint32_t *d_buf = nullptr;                               // [queries | pad to 8 B | offsets | ids]
  const size_t q_words = (nq_words + 1) & ~size_t(1);
  BIVX_TRY(tmp.alloc(&d_buf, q_words + off_words + cap));
  std::vector<uint32_t> h(q_words > off_words + cap ? q_words : off_words + cap);
  std::memcpy(h.data(), qlow, q * 4);
  std::memcpy(h.data() + q, qhigh, q * 4);
  if (qchrom) std::memcpy(h.data() + 2 * q, qchrom, q * 4);
  BIVX_HIP(hipMemcpyAsync(d_buf, h.data(), nq_words * 4, hipMemcpyHostToDevice, s));
  uint64_t *d_off = reinterpret_cast<uint64_t *>(d_buf + q_words);
  uint32_t *d_hits = d_buf + q_words + off_words;
  // (the filter can only be a type selection here: kind NONE, no aux arrays to upload)
  BIVX_TRY(bivx_query_dev_s(idx, qchrom ? d_buf + 2 * q : nullptr, d_buf, d_buf + q, q, filter, sort_by_id, d_off,
                            d_hits, cap, nullptr, 0, s));
  // offsets and the first ids in one copy: few queries rarely have more, and then a second copy fetches all
  const size_t have = std::min(cap, 4 * q + 128);
  BIVX_HIP(hipMemcpyAsync(h.data(), d_off, (off_words + have) * 4, hipMemcpyDeviceToHost, s));
  BIVX_HIP(hipStreamSynchronize(s));
  BIVX_TRY(lease.report("bivx_find_overlaps"));
  const uint64_t *off = reinterpret_cast<const uint64_t *>(h.data());
  const uint64_t total = off[q];
  if (total > cap) return kSmallBatchOverflow;
  std::memcpy(offsets_out, off, (q + 1) * 8);
  if (total == 0) return 0;
  uint32_t *out = static_cast<uint32_t *>(std::malloc((size_t)total * sizeof(uint32_t)));
  if (!out) {
    set_error("bivx_find_overlaps: out of host memory for %llu hit ids", (unsigned long long)total);
    return BIVX_E_NOMEM;
  }
  if (total <= have) {
    std::memcpy(out, h.data() + off_words, (size_t)total * 4);
  } else if (hipMemcpyAsync(out, d_hits, (size_t)total * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
             hipStreamSynchronize(s) != hipSuccess) {
    std::free(out);
    set_error("bivx_find_overlaps: device copy failed");
    return BIVX_E_HIP;
  }
  *hit_ids_out = out;
  return 0;
}

int find_overlaps_tiny(const bivx_index *idx, const LaneLease &lease, const uint32_t *qchrom, const uint32_t *qlow,
                       const uint32_t *qhigh, size_t q, const bivx_filter *filter, int sort_by_id, uint64_t *offsets_out,
                       uint32_t **hit_ids_out) {
  Mailbox mb(idx);
  if (!mb.host) return kSmallBatchOverflow;  // (no mapping to be had: the copying path)
  hipStream_t s = lease.lane.stream;
  // layout (u32 words): [qlow | qhigh | qchrom] (3 x 64) [pad] [offsets: 65 x u64] [ids ...]
  const size_t q_words = 3 * kMailboxQueries, off_words = 2 * (kMailboxQueries + 1);
  const size_t cap = kMailboxBytes / 4 - q_words - off_words;
  uint32_t *h = static_cast<uint32_t *>(mb.host), *d = static_cast<uint32_t *>(mb.dev);
  std::memcpy(h, qlow, q * 4);
  std::memcpy(h + kMailboxQueries, qhigh, q * 4);
  if (qchrom) std::memcpy(h + 2 * kMailboxQueries, qchrom, q * 4);
  uint64_t *d_off = reinterpret_cast<uint64_t *>(d + q_words);
  const uint64_t *h_off = reinterpret_cast<const uint64_t *>(h + q_words);
  // (one wavefront answers the whole call — k_query_tiny: no workspace, no prefix across workgroups; index order from
  // the device, and ascending ids, if asked for, are a std::sort over a few short lists below)
  IndexView view;
  BIVX_TRY(view_with_filter(idx, filter, view));  // (a type selection at most: the caller let no filter kind through)
  if (launch_query_tiny(view, qchrom ? d + 2 * kMailboxQueries : nullptr, d, d + kMailboxQueries, q, d_off,
                        d + q_words + off_words, cap, s) != 0)
    return kSmallBatchOverflow;
  BIVX_HIP(hipStreamSynchronize(s));
  BIVX_TRY(lease.report("bivx_find_overlaps"));
  const uint64_t total = h_off[q];
  if (total > cap) return kSmallBatchOverflow;
  std::memcpy(offsets_out, h_off, (q + 1) * 8);
  if (total == 0) return 0;
  uint32_t *out = static_cast<uint32_t *>(std::malloc((size_t)total * sizeof(uint32_t)));
  if (!out) {
    set_error("bivx_find_overlaps: out of host memory for %llu hit ids", (unsigned long long)total);
    return BIVX_E_NOMEM;
  }
  std::memcpy(out, h + q_words + off_words, (size_t)total * 4);
  if (sort_by_id)
    for (size_t i = 0; i < q; ++i) std::sort(out + h_off[i], out + h_off[i + 1]);
  *hit_ids_out = out;
  return 0;
}

// uploads a host-side filter's aux arrays; `dev` receives the same filter with device pointers
int upload_filter(TempPool &tmp, const bivx_index *idx, const bivx_filter *f, size_t q, hipStream_t s, bivx_filter &dev) {
  dev = bivx_filter{};
  if (!f) return 0;
  dev.svtype = f->svtype;  // a type selection travels with every kind, NONE included
  if (f->kind == BIVX_FILTER_NONE) return 0;
  dev = *f;
  dev.query_aux = dev.interval_aux = nullptr;
  if (f->query_aux && q) {
    uint32_t *d = nullptr;
    BIVX_TRY(tmp.alloc(&d, q));
    BIVX_HIP(hipMemcpyAsync(d, f->query_aux, q * 4, hipMemcpyHostToDevice, s));
    dev.query_aux = d;
  }
  if (f->interval_aux && idx->n) {
    uint32_t *d = nullptr;
    BIVX_TRY(tmp.alloc(&d, idx->n));
    BIVX_HIP(hipMemcpyAsync(d, f->interval_aux, idx->n * 4, hipMemcpyHostToDevice, s));
    dev.interval_aux = d;
  }
  return 0;
}
}  // namespace

int bivx_count(const bivx_index *idx, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh, size_t q,
               uint64_t *offsets_out) {
  return bivx_count_f(idx, qchrom, qlow, qhigh, q, nullptr, offsets_out);
}

int bivx_count_f(const bivx_index *idx, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh, size_t q,
                 const bivx_filter *filter, uint64_t *offsets_out) {
  if (idx && idx->sharded) {
    if ((q && (!qlow || !qhigh)) || !offsets_out) {
      set_error("bivx_count: null argument");
      return BIVX_E_INVALID;
    }
    return sharded_count(idx->sharded, qchrom, qlow, qhigh, q, filter, offsets_out);
  }
  BIVX_TRY(check_query_args(idx, qlow, qhigh, q, "bivx_count"));
  if (!offsets_out) {
    set_error("bivx_count: null offsets_out");
    return BIVX_E_INVALID;
  }
  if (q == 0) {
    offsets_out[0] = 0;
    return 0;
  }
  BIVX_GUARD(idx);
  LaneLease lease(idx);  // launch .. report on a lane of this call's own (bivx_index::Lane)
  hipStream_t s = lease.lane.stream;
  TempPool tmp(idx, s);
  DevQueries d;
  BIVX_TRY(upload_queries(tmp, qchrom, qlow, qhigh, q, s, d));
  uint64_t *d_off = nullptr;
  BIVX_TRY(tmp.alloc(&d_off, q + 1));
  bivx_filter dflt;
  BIVX_TRY(upload_filter(tmp, idx, filter, q, s, dflt));
  BIVX_TRY(bivx_count_dev_f(idx, d.c, d.lo, d.hi, q, &dflt, d_off, s));
  BIVX_HIP(hipMemcpyAsync(offsets_out, d_off, (q + 1) * 8, hipMemcpyDeviceToHost, s));
  BIVX_HIP(hipStreamSynchronize(s));
  return lease.report("bivx_count");
}

int bivx_fill(const bivx_index *idx, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh, size_t q,
              const uint64_t *offsets, uint32_t *hit_ids_out, int sort_by_id) {
  return bivx_fill_f(idx, qchrom, qlow, qhigh, q, nullptr, offsets, hit_ids_out, sort_by_id);
}

int bivx_fill_f(const bivx_index *idx, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh, size_t q,
                const bivx_filter *filter, const uint64_t *offsets, uint32_t *hit_ids_out, int sort_by_id) {
  if (idx && idx->sharded) {
    if (q && (!qlow || !qhigh || !offsets || (offsets[q] && !hit_ids_out))) {
      set_error("bivx_fill: null argument");
      return BIVX_E_INVALID;
    }
    return q ? sharded_fill(idx->sharded, qchrom, qlow, qhigh, q, filter, offsets, hit_ids_out, sort_by_id) : 0;
  }
  BIVX_TRY(check_query_args(idx, qlow, qhigh, q, "bivx_fill"));
  if (q == 0) return 0;
  if (!offsets) {
    set_error("bivx_fill: null offsets");
    return BIVX_E_INVALID;
  }
  const uint64_t total = offsets[q];
  if (total && !hit_ids_out) {
    set_error("bivx_fill: null hit_ids_out");
    return BIVX_E_INVALID;
  }
  BIVX_GUARD(idx);
  LaneLease lease(idx);  // launch .. report on a lane of this call's own (bivx_index::Lane)
  hipStream_t s = lease.lane.stream;
  TempPool tmp(idx, s);
  DevQueries d;
  BIVX_TRY(upload_queries(tmp, qchrom, qlow, qhigh, q, s, d));
  uint64_t *d_off = nullptr;
  uint32_t *d_hits = nullptr;
  BIVX_TRY(tmp.alloc(&d_off, q + 1));
  BIVX_TRY(tmp.alloc(&d_hits, (size_t)total));
  BIVX_HIP(hipMemcpyAsync(d_off, offsets, (q + 1) * 8, hipMemcpyHostToDevice, s));
  bivx_filter dflt;
  BIVX_TRY(upload_filter(tmp, idx, filter, q, s, dflt));
  BIVX_TRY(bivx_fill_dev_f(idx, d.c, d.lo, d.hi, q, &dflt, d_off, d_hits, s));
  if (sort_by_id) BIVX_TRY(launch_sort_hits(d_off, d_hits, q, ~0ull, s, nullptr, 0, total ? total : 1));
  if (total) BIVX_HIP(hipMemcpyAsync(hit_ids_out, d_hits, (size_t)total * 4, hipMemcpyDeviceToHost, s));
  BIVX_HIP(hipStreamSynchronize(s));
  return lease.report("bivx_fill");
}

int bivx_find_overlaps(const bivx_index *idx, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh,
                       size_t q, const bivx_filter *filter, int sort_by_id, uint64_t *offsets_out,
                       uint32_t **hit_ids_out) {
  if (idx && idx->sharded) {
    if ((q && (!qlow || !qhigh)) || !offsets_out || !hit_ids_out) {
      set_error("bivx_find_overlaps: null argument");
      return BIVX_E_INVALID;
    }
    return sharded_find_overlaps(idx->sharded, qchrom, qlow, qhigh, q, filter, sort_by_id, offsets_out, hit_ids_out);
  }
  BIVX_TRY(check_query_args(idx, qlow, qhigh, q, "bivx_find_overlaps"));
  if (!offsets_out || !hit_ids_out) {
    set_error("bivx_find_overlaps: null output");
    return BIVX_E_INVALID;
  }
  *hit_ids_out = nullptr;
  if (q == 0) {
    offsets_out[0] = 0;
    return 0;
  }
  BIVX_GUARD(idx);
  LaneLease lease(idx);  // launch .. report on a lane of this call's own (bivx_index::Lane)
  hipStream_t s = lease.lane.stream;
  if (q <= kSmallBatch && (!filter || filter->kind == BIVX_FILTER_NONE)) {
    int rc = kSmallBatchOverflow;
    if (q <= kMailboxQueries)
      rc = find_overlaps_tiny(idx, lease, qchrom, qlow, qhigh, q, filter, sort_by_id, offsets_out, hit_ids_out);
    if (rc == kSmallBatchOverflow)
      rc = find_overlaps_small(idx, lease, qchrom, qlow, qhigh, q, filter, sort_by_id, offsets_out, hit_ids_out);
    if (rc != kSmallBatchOverflow) return rc;
  }
  TempPool tmp(idx, s);
  DevQueries d;
  BIVX_TRY(upload_queries(tmp, qchrom, qlow, qhigh, q, s, d));
  bivx_filter dflt;
  BIVX_TRY(upload_filter(tmp, idx, filter, q, s, dflt));
  uint64_t *d_off = nullptr;
  BIVX_TRY(tmp.alloc(&d_off, q + 1));
  BIVX_TRY(bivx_count_dev_f(idx, d.c, d.lo, d.hi, q, &dflt, d_off, s));
  uint64_t total = 0;
  BIVX_HIP(hipMemcpyAsync(&total, d_off + q, 8, hipMemcpyDeviceToHost, s));
  BIVX_HIP(hipMemcpyAsync(offsets_out, d_off, (q + 1) * 8, hipMemcpyDeviceToHost, s));
  BIVX_HIP(hipStreamSynchronize(s));
  BIVX_TRY(lease.report("bivx_find_overlaps"));
  if (total == 0) return 0;
  uint32_t *h = static_cast<uint32_t *>(std::malloc((size_t)total * sizeof(uint32_t)));
  if (!h) {
    set_error("bivx_find_overlaps: out of host memory for %llu hit ids", (unsigned long long)total);
    return BIVX_E_NOMEM;
  }
  uint32_t *d_hits = nullptr;
  int rc = tmp.alloc(&d_hits, (size_t)total);
  if (rc == 0) rc = bivx_fill_dev_f(idx, d.c, d.lo, d.hi, q, &dflt, d_off, d_hits, s);
  if (rc == 0 && sort_by_id) rc = launch_sort_hits(d_off, d_hits, q, ~0ull, s, nullptr, 0, total ? total : 1);
  if (rc == 0 && hipMemcpyAsync(h, d_hits, (size_t)total * 4, hipMemcpyDeviceToHost, s) != hipSuccess) rc = BIVX_E_HIP;
  if (rc == 0 && hipStreamSynchronize(s) != hipSuccess) rc = BIVX_E_HIP;
  if (rc != 0) {
    if (rc == BIVX_E_HIP) set_error("bivx_find_overlaps: device copy failed");
    std::free(h);
    return rc;
  }
  *hit_ids_out = h;
  return 0;
}

void bivx_free(void *p) { std::free(p); }

int bivx_any(const bivx_index *idx, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh, size_t q,
             uint32_t *first_id_out) {
  if (idx && idx->sharded) {
    if (q && (!qlow || !qhigh || !first_id_out)) {
      set_error("bivx_any: null argument");
      return BIVX_E_INVALID;
    }
    return q ? sharded_any(idx->sharded, qchrom, qlow, qhigh, q, first_id_out) : 0;
  }
  BIVX_TRY(check_query_args(idx, qlow, qhigh, q, "bivx_any"));
  if (q == 0) return 0;
  if (!first_id_out) {
    set_error("bivx_any: null output");
    return BIVX_E_INVALID;
  }
  BIVX_GUARD(idx);
  LaneLease lease(idx);
  hipStream_t s = lease.lane.stream;
  if (q <= kMailboxQueries) {  // a handful of queries (the facade's find_overlap per record): through a mailbox, no copies
    Mailbox mb(idx);
    if (mb.host) {
      uint32_t *h = static_cast<uint32_t *>(mb.host), *d = static_cast<uint32_t *>(mb.dev);
      std::memcpy(h, qlow, q * 4);
      std::memcpy(h + kMailboxQueries, qhigh, q * 4);
      if (qchrom) std::memcpy(h + 2 * kMailboxQueries, qchrom, q * 4);
      BIVX_TRY(bivx_any_dev(idx, qchrom ? d + 2 * kMailboxQueries : nullptr, d, d + kMailboxQueries, q,
                            d + 3 * kMailboxQueries, s));
      BIVX_HIP(hipStreamSynchronize(s));
      std::memcpy(first_id_out, h + 3 * kMailboxQueries, q * 4);
      return 0;
    }
  }
  TempPool tmp(idx, s);
  DevQueries d;
  BIVX_TRY(upload_queries(tmp, qchrom, qlow, qhigh, q, s, d));
  uint32_t *d_first = nullptr;
  BIVX_TRY(tmp.alloc(&d_first, q));
  BIVX_TRY(bivx_any_dev(idx, d.c, d.lo, d.hi, q, d_first, s));
  BIVX_HIP(hipMemcpyAsync(first_id_out, d_first, q * 4, hipMemcpyDeviceToHost, s));
  BIVX_HIP(hipStreamSynchronize(s));
  return 0;
}

int bivx_query_sharded_dev(const bivx_index *idx, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh,
                           size_t q, int sort_by_id, bivx_sharded_result *out) {
  if (!idx || !out) {
    set_error("bivx_query_sharded_dev: null argument");
    return BIVX_E_INVALID;
  }
  if (!idx->sharded) {
    set_error("bivx_query_sharded_dev: not a sharded handle (bivx_create_sharded); a single-device index answers "
              "device-resident batches with bivx_query_dev_s");
    return BIVX_E_STATE;
  }
  return sharded_query_dev(idx->sharded, qchrom, qlow, qhigh, q, sort_by_id, out);
}

int bivx_get_stats(const bivx_index *idx, bivx_stats *out) {
  if (!idx || !out) {
    set_error("bivx_get_stats: null argument");
    return BIVX_E_INVALID;
  }
  if (idx->sharded) {
    sharded_stats(idx->sharded, out);
    return 0;
  }
  memset(out, 0, sizeof(*out));
  out->n_intervals = idx->n;
  out->n_chroms = idx->nchrom;
  out->n_segments = idx->nseg;
  out->n_cells = idx->nentries;
  out->staging_bytes = (uint64_t)idx->cap * 12;
  if (idx->built)
    out->index_bytes = (uint64_t)idx->built_n * 20 + idx->nentries * 4 + (uint64_t)idx->nseg * sizeof(SegDesc) +
                       ((uint64_t)idx->nchrom + 1) * 4;
  out->build_ms = idx->build_ms;
  {
    volatile uint32_t *e = idx->h_err;
    out->prefix_timeouts = idx->errors_reported.load() + ((e[kErrTimeout] | e[kErrWorkspace]) ? 1u : 0u);
  }
  return 0;
}

}  // extern "C"
