// sharded.cpp — one index over several GPUs of a node, behind the same handle and the same host-pointer entry points
// (include/bivx.h, bivx_create_sharded). Plain host C++ on top of the library's own C ABI: every shard is an ordinary
// single-device bivx_index; this file only routes.
//
// Decomposition (the reference's only one: one task per chromosome, standalone/sv2nl/include/mapper.hpp:238-246):
// whole chromosomes are assigned to devices by LPT on n_c * log2(n_c); a query goes to the device
// that holds its chromosome; nothing is exchanged between devices. An index with fewer populated chromosomes than
// devices (a plain IntervalTree has one) is replicated instead and the QUERIES are split into contiguous ranges.
// One host thread per device drives its shard; results land in the caller's host arrays in query order, with
// global (append-order) interval ids.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "common.h"

namespace bivx {

struct ShardedState {
  std::vector<int> devices;
  std::vector<bivx_index *> shard;
  // everything appended, in append order (ids are indices into these)
  std::vector<uint32_t> chrom, low, high;
  std::vector<uint8_t> type;
  bool typed = false, built = false, by_chrom = false;
  size_t built_n = 0;
  uint32_t nchrom = 0, ntypes = 1;
  std::vector<uint32_t> chrom_shard;         // by_chrom: chromosome -> shard (0xFFFFFFFF: no interval there)
  std::vector<std::vector<uint32_t>> gid;    // by_chrom: shard-local id -> global id
  double build_ms = 0.0;
  // ---- bivx_query_sharded_dev: the shards' CSRs gathered into devices[0]'s memory over RCCL ----------------------
  struct Buf {  // grow-only device block
    void *p = nullptr;
    size_t cap = 0;
  };
  struct ShardDev {
    hipStream_t stream = nullptr;   // on the shard's device: its queries, its id mapping and its side of the exchange
    Buf q, off, hits, gid, sizes;   // the shard's sub-batch, its CSR, its local -> global id table, (queries, ids) x devices
    size_t gid_n = 0;               // ids of the uploaded table (0: not uploaded since the last build)
  };
  mutable std::vector<ShardDev> dev;
  mutable std::vector<ncclComm_t> comms;  // ncclCommInitAll over the handle's devices (all different), made on first use
  mutable bool comm_tried = false;
  mutable Buf out_off, out_hits, out_rows;  // the gathered CSR and its row -> query map, on devices[0]
  mutable std::mutex dev_mutex;             // one gathered call at a time: the buffers above are the handle's
};

namespace {

// No C++ exception may cross the C ABI (include/bivx.h): everything below that allocates (std::vector, std::thread)
// runs inside this, which turns a throw into a status code and an error text.
template <typename F>
int no_throw(const char *who, F &&f) noexcept {
  try {
    return f();
  } catch (const std::bad_alloc &) {
    set_error("%s: out of host memory", who);
    return BIVX_E_NOMEM;
  } catch (const std::exception &e) {
    set_error("%s: %s", who, e.what());
    return BIVX_E_HIP;
  } catch (...) {
    set_error("%s: unknown failure", who);
    return BIVX_E_HIP;
  }
}

template <typename F>
int on_every_shard(const ShardedState *st, F &&f) {
  const size_t k = st->shard.size();
  std::vector<int> rc(k, 0);
  std::vector<std::string> err(k);
  std::vector<std::thread> th;
  th.reserve(k);
  try {
    for (size_t s = 0; s < k; ++s)
      th.emplace_back([&, s] {
        // (an exception that left a thread's body would end the process: std::terminate)
        rc[s] = no_throw("shard thread", [&] { return f(s); });
        if (rc[s] != 0) {
          try {
            err[s] = bivx_last_error();  // thread-local text: carry it over to the caller's thread
          } catch (...) {
          }
        }
      });
  } catch (...) {  // std::system_error: no more threads to be had — the ones that started must still be joined
    for (auto &t : th) t.join();
    throw;
  }
  for (auto &t : th) t.join();
  for (size_t s = 0; s < k; ++s)
    if (rc[s] != 0) {
      set_error("shard %zu (device %d): %s", s, st->devices[s], err[s].c_str());
      return rc[s];
    }
  return 0;
}

// which shard answers query i, and the query lists per shard
struct Route {
  std::vector<std::vector<uint32_t>> qs;  // per shard: indices of its queries, ascending
};

int route(const ShardedState *st, const uint32_t *qchrom, size_t q, Route &r) {
  const size_t k = st->shard.size();
  r.qs.assign(k, {});
  if (q > 0xFFFFFFFFull) {
    set_error("sharded index: more than 2^32 queries in one call");
    return BIVX_E_RANGE;
  }
  if (st->by_chrom) {
    for (size_t i = 0; i < q; ++i) {
      const uint32_t c = qchrom ? qchrom[i] : 0u;
      const uint32_t s = c < st->chrom_shard.size() ? st->chrom_shard[c] : 0xFFFFFFFFu;
      if (s != 0xFFFFFFFFu) r.qs[s].push_back((uint32_t)i);
    }
  } else {  // replicated index: contiguous query ranges
    for (size_t s = 0; s < k; ++s) {
      const size_t a = q * s / k, b = q * (s + 1) / k;
      r.qs[s].resize(b - a);
      std::iota(r.qs[s].begin(), r.qs[s].end(), (uint32_t)a);
    }
  }
  return 0;
}

struct SubBatch {
  std::vector<uint32_t> c, lo, hi, aux;
  bivx_filter flt{};
};

// a shard's part of the batch (and of the filter's per-query words; per-interval words are remapped to local ids)
void gather(const ShardedState *st, size_t s, const std::vector<uint32_t> &qs, const uint32_t *qchrom,
            const uint32_t *qlow, const uint32_t *qhigh, const bivx_filter *f, SubBatch &b,
            std::vector<uint32_t> &iaux_local) {
  const size_t m = qs.size();
  b.lo.resize(m);
  b.hi.resize(m);
  if (qchrom) b.c.resize(m);
  for (size_t j = 0; j < m; ++j) {
    b.lo[j] = qlow[qs[j]];
    b.hi[j] = qhigh[qs[j]];
    if (qchrom) b.c[j] = qchrom[qs[j]];
  }
  if (f) {
    b.flt = *f;
    if (f->query_aux) {
      b.aux.resize(m);
      for (size_t j = 0; j < m; ++j) b.aux[j] = f->query_aux[qs[j]];
      b.flt.query_aux = b.aux.data();
    }
    if (f->interval_aux && st->by_chrom) {
      const auto &g = st->gid[s];
      iaux_local.resize(g.size());
      for (size_t j = 0; j < g.size(); ++j) iaux_local[j] = f->interval_aux[g[j]];
      b.flt.interval_aux = iaux_local.data();
    }
  }
}

}  // namespace

static int create_impl(ShardedState **out, const int *devices, int ndev) {
  if (!devices || ndev < 1) {
    set_error("bivx_create_sharded: need at least one device");
    return BIVX_E_INVALID;
  }
  auto st = std::make_unique<ShardedState>();
  for (int i = 0; i < ndev; ++i) {
    bivx_index *ix = nullptr;
    const int rc = bivx_create(&ix, devices[i]);
    if (rc != 0) {
      for (auto *p : st->shard) bivx_destroy(p);
      return rc;
    }
    st->devices.push_back(devices[i]);
    st->shard.push_back(ix);
  }
  *out = st.release();
  return 0;
}

static void release_dev_state(ShardedState *st);
void sharded_destroy(ShardedState *st) {
  if (!st) return;
  try {
    release_dev_state(st);
  } catch (...) {
  }
  for (auto *p : st->shard) bivx_destroy(p);
  delete st;
}

int sharded_num_devices(const ShardedState *st) { return (int)st->shard.size(); }
int sharded_device_of_chrom(const ShardedState *st, uint32_t chrom) {
  if (!st->built) return -1;
  if (!st->by_chrom) return st->devices[0];
  if (chrom >= st->chrom_shard.size() || st->chrom_shard[chrom] == 0xFFFFFFFFu) return -1;
  return st->devices[st->chrom_shard[chrom]];
}

static int append_impl(ShardedState *st, const uint32_t *chrom, const uint32_t *low, const uint32_t *high,
                   const uint8_t *svtype, size_t n) {
  if (n && (!low || !high)) {
    set_error("bivx_append: null argument");
    return BIVX_E_INVALID;
  }
  if (st->low.size() + n >= 0xFFFFFFFFull) {
    set_error("too many intervals (ids are uint32)");
    return BIVX_E_RANGE;
  }
  const size_t old = st->low.size();
  st->low.insert(st->low.end(), low, low + n);
  st->high.insert(st->high.end(), high, high + n);
  if (chrom) st->chrom.insert(st->chrom.end(), chrom, chrom + n);
  else st->chrom.resize(old + n, 0u);
  if (svtype) {
    st->type.insert(st->type.end(), svtype, svtype + n);
    st->typed = true;
  } else {
    st->type.resize(old + n, 0);
  }
  st->built = false;
  return 0;
}

static int clear_impl(ShardedState *st) {
  st->chrom.clear();
  st->low.clear();
  st->high.clear();
  st->type.clear();
  st->typed = st->built = false;
  st->built_n = 0;
  for (auto *p : st->shard) BIVX_TRY(bivx_clear(p));
  return 0;
}

size_t sharded_size(const ShardedState *st) { return st->low.size(); }
bool sharded_is_built(const ShardedState *st) { return st->built && st->built_n == st->low.size(); }
uint32_t sharded_num_chroms(const ShardedState *st) { return st->nchrom; }
uint32_t sharded_num_types(const ShardedState *st) { return st->ntypes; }

static int build_impl(ShardedState *st) {
  if (sharded_is_built(st)) return 0;
  const size_t n = st->low.size(), k = st->shard.size();
  uint32_t max_chrom = 0, max_type = 0;
  for (size_t i = 0; i < n; ++i) {
    max_chrom = std::max(max_chrom, st->chrom[i]);
    max_type = std::max<uint32_t>(max_type, st->type[i]);
  }
  // on the id itself, before anything is indexed by it (id + 1 wraps to 0 for 0xFFFFFFFF); the single-device build
  // rejects the same input the same way (capi.hip, bivx_build)
  if (n && max_chrom >= BIVX_MAX_CHROMS) {
    set_error("chromosome id %u exceeds BIVX_MAX_CHROMS", max_chrom);
    return BIVX_E_RANGE;
  }
  const uint32_t nchrom = n ? max_chrom + 1 : 0;
  std::vector<uint64_t> cnt(nchrom, 0);
  for (size_t i = 0; i < n; ++i) ++cnt[st->chrom[i]];
  const size_t populated = (size_t)std::count_if(cnt.begin(), cnt.end(), [](uint64_t c) { return c != 0; });
  st->by_chrom = k > 1 && populated >= k;
  st->chrom_shard.assign(nchrom, 0xFFFFFFFFu);
  st->gid.assign(k, {});
  std::vector<std::vector<uint32_t>> ids(k);  // global ids per shard, ascending (append order is kept inside a shard)
  if (st->by_chrom) {
    // LPT: heaviest chromosome to the least loaded device; ties to the lower chromosome id / lower device
    std::vector<uint32_t> order;
    for (uint32_t c = 0; c < nchrom; ++c)
      if (cnt[c]) order.push_back(c);
    auto weight = [&](uint32_t c) { return (double)cnt[c] * std::log2((double)cnt[c] + 2.0); };
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return weight(a) > weight(b); });
    std::vector<double> load(k, 0.0);
    for (uint32_t c : order) {
      const size_t s = (size_t)(std::min_element(load.begin(), load.end()) - load.begin());
      st->chrom_shard[c] = (uint32_t)s;
      load[s] += weight(c);
    }
    for (size_t i = 0; i < n; ++i) ids[st->chrom_shard[st->chrom[i]]].push_back((uint32_t)i);
  } else {
    for (size_t s = 0; s < k; ++s) {
      ids[s].resize(n);
      std::iota(ids[s].begin(), ids[s].end(), 0u);
    }
  }
  const auto t0 = std::chrono::steady_clock::now();
  BIVX_TRY(on_every_shard(st, [&](size_t s) -> int {
    const auto &g = ids[s];
    std::vector<uint32_t> c(g.size()), lo(g.size()), hi(g.size());
    std::vector<uint8_t> ty(g.size());
    for (size_t j = 0; j < g.size(); ++j) {
      c[j] = st->chrom[g[j]];
      lo[j] = st->low[g[j]];
      hi[j] = st->high[g[j]];
      ty[j] = st->type[g[j]];
    }
    BIVX_TRY(bivx_clear(st->shard[s]));
    if (st->typed) BIVX_TRY(bivx_append_typed(st->shard[s], c.data(), lo.data(), hi.data(), ty.data(), g.size()));
    else BIVX_TRY(bivx_append(st->shard[s], c.data(), lo.data(), hi.data(), g.size()));
    return bivx_build(st->shard[s]);
  }));
  st->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (st->by_chrom) st->gid = std::move(ids);
  for (auto &d : st->dev) d.gid_n = 0;  // (the id tables uploaded for bivx_query_sharded_dev belong to the last build)
  st->nchrom = nchrom;
  st->ntypes = st->typed ? max_type + 1 : 1;
  st->built = true;
  st->built_n = n;
  return 0;
}

int sharded_get_intervals(const ShardedState *st, const uint32_t *ids, size_t n, uint32_t *chrom_out,
                          uint32_t *low_out, uint32_t *high_out) {
  for (size_t i = 0; i < n; ++i) {
    const bool ok = ids[i] < st->low.size();
    if (chrom_out) chrom_out[i] = ok ? st->chrom[ids[i]] : 0xFFFFFFFFu;
    if (low_out) low_out[i] = ok ? st->low[ids[i]] : 0xFFFFFFFFu;
    if (high_out) high_out[i] = ok ? st->high[ids[i]] : 0u;
  }
  return 0;
}

int sharded_get_svtypes(const ShardedState *st, const uint32_t *ids, size_t n, uint8_t *out) {
  for (size_t i = 0; i < n; ++i) out[i] = ids[i] < st->type.size() ? st->type[ids[i]] : (uint8_t)0xFF;
  return 0;
}

// the whole batch: every shard answers its queries (bivx_find_overlaps), the CSR is assembled in query order
static int find_overlaps_impl(const ShardedState *st, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh,
                          size_t q, const bivx_filter *filter, int sort_by_id, uint64_t *offsets_out,
                          uint32_t **hit_ids_out) {
  *hit_ids_out = nullptr;
  if (!sharded_is_built(st)) {
    set_error("bivx_find_overlaps: index not built (call bivx_build after the last append)");
    return BIVX_E_STATE;
  }
  const size_t k = st->shard.size();
  Route r;
  BIVX_TRY(route(st, qchrom, q, r));
  std::vector<std::vector<uint64_t>> off(k);
  std::vector<uint32_t *> hits(k, nullptr);
  struct Freer {
    std::vector<uint32_t *> &h;
    ~Freer() {
      for (auto *p : h) bivx_free(p);
    }
  } freer{hits};
  BIVX_TRY(on_every_shard(st, [&](size_t s) -> int {
    SubBatch b;
    std::vector<uint32_t> iaux;
    gather(st, s, r.qs[s], qchrom, qlow, qhigh, filter, b, iaux);
    off[s].assign(r.qs[s].size() + 1, 0);
    if (r.qs[s].empty()) return 0;
    return bivx_find_overlaps(st->shard[s], qchrom ? b.c.data() : nullptr, b.lo.data(), b.hi.data(), r.qs[s].size(),
                              filter ? &b.flt : nullptr, sort_by_id, off[s].data(), &hits[s]);
  }));
  std::fill(offsets_out, offsets_out + q + 1, 0ull);
  for (size_t s = 0; s < k; ++s)
    for (size_t j = 0; j < r.qs[s].size(); ++j) offsets_out[r.qs[s][j] + 1] = off[s][j + 1] - off[s][j];
  for (size_t i = 0; i < q; ++i) offsets_out[i + 1] += offsets_out[i];
  const uint64_t total = offsets_out[q];
  if (total == 0) return 0;
  uint32_t *out = static_cast<uint32_t *>(std::malloc((size_t)total * sizeof(uint32_t)));
  if (!out) {
    set_error("bivx_find_overlaps: out of host memory for %llu hit ids", (unsigned long long)total);
    return BIVX_E_NOMEM;
  }
  // Shard-local ids ascend with the global ones (a shard keeps append order), so a list that is ascending locally is
  // ascending globally: sort_by_id needs no second look.
  struct FreeOnThrow {  // (thread creation below may throw)
    uint32_t *p;
    ~FreeOnThrow() { std::free(p); }
  } guard{out};
  (void)on_every_shard(st, [&](size_t s) -> int {
    const uint32_t *g = st->by_chrom ? st->gid[s].data() : nullptr;
    for (size_t j = 0; j < r.qs[s].size(); ++j) {
      uint32_t *dst = out + offsets_out[r.qs[s][j]];
      const uint32_t *src = hits[s] + off[s][j];
      const size_t m = (size_t)(off[s][j + 1] - off[s][j]);
      if (g) for (size_t x = 0; x < m; ++x) dst[x] = g[src[x]];
      else std::memcpy(dst, src, m * sizeof(uint32_t));
    }
    return 0;
  });
  guard.p = nullptr;
  *hit_ids_out = out;
  return 0;
}

static int count_impl(const ShardedState *st, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh,
                  size_t q, const bivx_filter *filter, uint64_t *offsets_out) {
  if (!sharded_is_built(st)) {
    set_error("bivx_count: index not built (call bivx_build after the last append)");
    return BIVX_E_STATE;
  }
  const size_t k = st->shard.size();
  Route r;
  BIVX_TRY(route(st, qchrom, q, r));
  std::vector<std::vector<uint64_t>> off(k);
  BIVX_TRY(on_every_shard(st, [&](size_t s) -> int {
    SubBatch b;
    std::vector<uint32_t> iaux;
    gather(st, s, r.qs[s], qchrom, qlow, qhigh, filter, b, iaux);
    off[s].assign(r.qs[s].size() + 1, 0);
    if (r.qs[s].empty()) return 0;
    return bivx_count_f(st->shard[s], qchrom ? b.c.data() : nullptr, b.lo.data(), b.hi.data(), r.qs[s].size(),
                        filter ? &b.flt : nullptr, off[s].data());
  }));
  std::fill(offsets_out, offsets_out + q + 1, 0ull);
  for (size_t s = 0; s < k; ++s)
    for (size_t j = 0; j < r.qs[s].size(); ++j) offsets_out[r.qs[s][j] + 1] = off[s][j + 1] - off[s][j];
  for (size_t i = 0; i < q; ++i) offsets_out[i + 1] += offsets_out[i];
  return 0;
}

static int fill_impl(const ShardedState *st, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh,
                 size_t q, const bivx_filter *filter, const uint64_t *offsets, uint32_t *hit_ids_out,
                 int sort_by_id) {
  // the offsets are the caller's; the ids are what bivx_find_overlaps gives for the same batch
  std::vector<uint64_t> off(q + 1);
  uint32_t *hits = nullptr;
  BIVX_TRY(find_overlaps_impl(st, qchrom, qlow, qhigh, q, filter, sort_by_id, off.data(), &hits));
  int rc = 0;
  if (std::memcmp(off.data(), offsets, (q + 1) * sizeof(uint64_t)) != 0) {
    set_error("bivx_fill: offsets do not belong to this batch");
    rc = BIVX_E_INVALID;
  } else if (off[q]) {
    std::memcpy(hit_ids_out, hits, (size_t)off[q] * sizeof(uint32_t));
  }
  bivx_free(hits);
  return rc;
}

static int any_impl(const ShardedState *st, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh,
                size_t q, uint32_t *first_id_out) {
  if (!sharded_is_built(st)) {
    set_error("bivx_any: index not built (call bivx_build after the last append)");
    return BIVX_E_STATE;
  }
  Route r;
  BIVX_TRY(route(st, qchrom, q, r));
  std::fill(first_id_out, first_id_out + q, BIVX_NO_HIT);
  return on_every_shard(st, [&](size_t s) -> int {
    if (r.qs[s].empty()) return 0;
    SubBatch b;
    std::vector<uint32_t> iaux;
    gather(st, s, r.qs[s], qchrom, qlow, qhigh, nullptr, b, iaux);
    std::vector<uint32_t> first(r.qs[s].size());
    BIVX_TRY(bivx_any(st->shard[s], qchrom ? b.c.data() : nullptr, b.lo.data(), b.hi.data(), r.qs[s].size(),
                      first.data()));
    for (size_t j = 0; j < r.qs[s].size(); ++j)
      if (first[j] != BIVX_NO_HIT) first_id_out[r.qs[s][j]] = st->by_chrom ? st->gid[s][first[j]] : first[j];
    return 0;
  });
}


// ---- bivx_query_sharded_dev: every shard answers on its device, RCCL gathers the CSRs into devices[0] --------------------

namespace {

#define BIVX_NCCL(call)                                                                                      \
  do {                                                                                                       \
    ncclResult_t bivx_n_ = (call);                                                                           \
    if (bivx_n_ != ncclSuccess) {                                                                            \
      set_error("%s failed: %s (%s:%d)", #call, ncclGetErrorString(bivx_n_), __FILE__, __LINE__);            \
      return BIVX_E_COMM;                                                                                    \
    }                                                                                                        \
  } while (0)

struct OnDevice {  // the calling thread's current device for a scope
  int prev = -1;
  bool ok = false;
  explicit OnDevice(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = prev == dev || hipSetDevice(dev) == hipSuccess;
  }
  ~OnDevice() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

int grow(ShardedState::Buf &b, size_t bytes) {  // (the caller is on the block's device; nothing of the old block is in use)
  if (b.p && bytes <= b.cap) return 0;
  (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
  const size_t want = std::max<size_t>((bytes + 255) & ~(size_t)255, 256);
  BIVX_HIP(hipMalloc(&b.p, want));
  b.cap = want;
  return 0;
}

// shard-local ids -> the handle's global append-order ids
__global__ __launch_bounds__(256) void k_map_ids(uint32_t *__restrict__ hits, uint64_t n, const uint32_t *__restrict__ gid) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) hits[i] = gid[hits[i]];
}

// the gathered offsets: block r arrived as the shard's own prefix sums (from 0); it begins hdisp[r] ids into the result
__global__ __launch_bounds__(256) void k_rebase_offsets(uint64_t *__restrict__ off, const uint64_t *__restrict__ qdisp,
                                                        const uint64_t *__restrict__ hdisp, uint32_t nshards, uint64_t rows,
                                                        uint64_t total) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i == rows) off[rows] = total;
  if (i >= rows) return;
  uint32_t r = 0;
  while (r + 1 < nshards && qdisp[r + 1] <= i) ++r;
  off[i] += hdisp[r];
}

int ensure_dev_state(const ShardedState *st) {
  const size_t k = st->shard.size();
  if (st->dev.size() != k) st->dev.resize(k);
  for (size_t s = 0; s < k; ++s) {
    if (st->dev[s].stream) continue;
    OnDevice g(st->devices[s]);
    if (!g.ok) {
      set_error("hipSetDevice(%d) failed", st->devices[s]);
      return BIVX_E_HIP;
    }
    BIVX_HIP(hipStreamCreateWithFlags(&st->dev[s].stream, hipStreamNonBlocking));
  }
  if (!st->comm_tried) {
    st->comm_tried = true;
    std::vector<int> d = st->devices;
    std::sort(d.begin(), d.end());
    const bool distinct = std::adjacent_find(d.begin(), d.end()) == d.end();
    if (distinct) {  // (RCCL refuses a device twice in one communicator: such handles copy device to device)
      st->comms.assign(k, nullptr);
      const ncclResult_t e = ncclCommInitAll(st->comms.data(), (int)k, st->devices.data());
      if (e != ncclSuccess) {
        st->comms.clear();
        st->comm_tried = false;
        set_error("ncclCommInitAll over %zu devices failed: %s", k, ncclGetErrorString(e));
        return BIVX_E_COMM;
      }
    }
  }
  return 0;
}

}  // namespace

static int query_dev_impl(const ShardedState *st, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh,
                          size_t q, int sort_by_id, bivx_sharded_result *out) {
  if (!out || (q && (!qlow || !qhigh))) {
    set_error("bivx_query_sharded_dev: null argument");
    return BIVX_E_INVALID;
  }
  std::memset(out, 0, sizeof(*out));
  if (!sharded_is_built(st)) {
    set_error("bivx_query_sharded_dev: index not built (call bivx_build after the last append)");
    return BIVX_E_STATE;
  }
  std::lock_guard<std::mutex> lock(st->dev_mutex);
  BIVX_TRY(ensure_dev_state(st));
  const size_t k = st->shard.size();
  Route r;
  BIVX_TRY(route(st, qchrom, q, r));
  if (st->by_chrom) {
    // a query on a chromosome no device holds still gets its (empty) row: the first shard answers it
    std::vector<uint32_t> extra;
    for (size_t i = 0; i < q; ++i) {
      const uint32_t c = qchrom ? qchrom[i] : 0u;
      if (c >= st->chrom_shard.size() || st->chrom_shard[c] == 0xFFFFFFFFu) extra.push_back((uint32_t)i);
    }
    if (!extra.empty()) {
      std::vector<uint32_t> merged(r.qs[0].size() + extra.size());
      std::merge(r.qs[0].begin(), r.qs[0].end(), extra.begin(), extra.end(), merged.begin());
      r.qs[0].swap(merged);
    }
  }
  // 1. every shard answers its queries on its own device and stream: count, size, single pass, ids made global
  std::vector<uint64_t> nq(k, 0), nh(k, 0);
  BIVX_TRY(on_every_shard(st, [&](size_t s) -> int {
    ShardedState::ShardDev &d = st->dev[s];
    OnDevice g(st->devices[s]);
    if (!g.ok) {
      set_error("hipSetDevice(%d) failed", st->devices[s]);
      return BIVX_E_HIP;
    }
    const size_t m = r.qs[s].size();
    nq[s] = m;
    BIVX_TRY(grow(d.sizes, 2 * sizeof(uint64_t) * (k + 1)));
    BIVX_TRY(grow(d.off, (m + 1) * sizeof(uint64_t)));
    if (m == 0) {
      BIVX_HIP(hipMemsetAsync(d.off.p, 0, sizeof(uint64_t), d.stream));
      BIVX_HIP(hipStreamSynchronize(d.stream));
      return 0;
    }
    SubBatch b;
    std::vector<uint32_t> iaux;
    gather(st, s, r.qs[s], qchrom, qlow, qhigh, nullptr, b, iaux);
    BIVX_TRY(grow(d.q, 3 * m * sizeof(uint32_t)));
    uint32_t *dq = static_cast<uint32_t *>(d.q.p);
    if (qchrom) BIVX_HIP(hipMemcpyAsync(dq, b.c.data(), m * 4, hipMemcpyHostToDevice, d.stream));
    BIVX_HIP(hipMemcpyAsync(dq + m, b.lo.data(), m * 4, hipMemcpyHostToDevice, d.stream));
    BIVX_HIP(hipMemcpyAsync(dq + 2 * m, b.hi.data(), m * 4, hipMemcpyHostToDevice, d.stream));
    uint64_t *doff = static_cast<uint64_t *>(d.off.p);
    BIVX_TRY(bivx_count_dev(st->shard[s], qchrom ? dq : nullptr, dq + m, dq + 2 * m, m, doff, d.stream));
    uint64_t total = 0;
    BIVX_HIP(hipMemcpyAsync(&total, doff + m, sizeof(uint64_t), hipMemcpyDeviceToHost, d.stream));
    BIVX_HIP(hipStreamSynchronize(d.stream));  // (also: the host vectors of the sub-batch may go)
    BIVX_TRY(bivx_stream_status(st->shard[s], d.stream));
    nh[s] = total;
    if (total == 0) return 0;
    BIVX_TRY(grow(d.hits, (size_t)total * sizeof(uint32_t)));
    uint32_t *dh = static_cast<uint32_t *>(d.hits.p);
    BIVX_TRY(bivx_query_dev_s(st->shard[s], qchrom ? dq : nullptr, dq + m, dq + 2 * m, m, nullptr, sort_by_id, doff, dh, total,
                              nullptr, 0, d.stream));
    if (st->by_chrom) {  // (shard-local ids ascend with the global ones: an ordered list stays ordered)
      const auto &gl = st->gid[s];
      if (d.gid_n != gl.size() || !d.gid.p) {
        BIVX_TRY(grow(d.gid, gl.size() * sizeof(uint32_t)));
        BIVX_HIP(hipMemcpyAsync(d.gid.p, gl.data(), gl.size() * sizeof(uint32_t), hipMemcpyHostToDevice, d.stream));
        d.gid_n = gl.size();
      }
      hipLaunchKernelGGL(k_map_ids, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, d.stream, dh, total,
                         static_cast<const uint32_t *>(d.gid.p));
      BIVX_HIP(hipGetLastError());
    }
    BIVX_HIP(hipStreamSynchronize(d.stream));
    return bivx_stream_status(st->shard[s], d.stream);
  }));
  // 2. sizes: ncclAllGather of (queries, ids) per device — every device learns every block's size, as a consumer on
  // any device would need them; the host's copy of the same numbers drives the displacements below
  const bool rccl = !st->comms.empty();
  std::vector<uint64_t> qdisp(k + 1, 0), hdisp(k + 1, 0);
  for (size_t s = 0; s < k; ++s) {
    qdisp[s + 1] = qdisp[s] + nq[s];
    hdisp[s + 1] = hdisp[s] + nh[s];
  }
  const uint64_t rows = qdisp[k], total = hdisp[k];
  if (rccl) {
    for (size_t s = 0; s < k; ++s) {
      OnDevice g(st->devices[s]);
      const uint64_t mine[2] = {nq[s], nh[s]};
      BIVX_HIP(hipMemcpyAsync(st->dev[s].sizes.p, mine, sizeof(mine), hipMemcpyHostToDevice, st->dev[s].stream));
      BIVX_HIP(hipStreamSynchronize(st->dev[s].stream));  // (`mine` is a stack array)
    }
    BIVX_NCCL(ncclGroupStart());
    for (size_t s = 0; s < k; ++s) {
      OnDevice g(st->devices[s]);  // (a communicator's calls are made with its device current)
      uint64_t *sz = static_cast<uint64_t *>(st->dev[s].sizes.p);
      const ncclResult_t e = ncclAllGather(sz, sz + 2, 2, ncclUint64, st->comms[s], st->dev[s].stream);
      if (e != ncclSuccess) {
        (void)ncclGroupEnd();
        set_error("ncclAllGather failed: %s", ncclGetErrorString(e));
        return BIVX_E_COMM;
      }
    }
    BIVX_NCCL(ncclGroupEnd());
    std::vector<uint64_t> seen(2 * k);
    {
      OnDevice g(st->devices[0]);
      BIVX_HIP(hipMemcpyAsync(seen.data(), static_cast<uint64_t *>(st->dev[0].sizes.p) + 2, 2 * k * sizeof(uint64_t),
                              hipMemcpyDeviceToHost, st->dev[0].stream));
      BIVX_HIP(hipStreamSynchronize(st->dev[0].stream));
    }
    for (size_t s = 0; s < k; ++s)
      if (seen[2 * s] != nq[s] || seen[2 * s + 1] != nh[s]) {
        set_error("bivx_query_sharded_dev: the gathered sizes of device %d are (%llu, %llu), expected (%llu, %llu)",
                  st->devices[s], (unsigned long long)seen[2 * s], (unsigned long long)seen[2 * s + 1],
                  (unsigned long long)nq[s], (unsigned long long)nh[s]);
        return BIVX_E_COMM;
      }
  }
  // 3. the blocks travel to devices[0]: offsets (without their last entry) and ids of every shard, each peer over its own
  // link, all inside ONE group; the root's own block is a local copy
  OnDevice root(st->devices[0]);
  if (!root.ok) {
    set_error("hipSetDevice(%d) failed", st->devices[0]);
    return BIVX_E_HIP;
  }
  hipStream_t rs = st->dev[0].stream;
  BIVX_TRY(grow(st->out_off, (rows + 1) * sizeof(uint64_t) + 2 * (k + 1) * sizeof(uint64_t)));
  BIVX_TRY(grow(st->out_hits, std::max<uint64_t>(total, 1) * sizeof(uint32_t)));
  BIVX_TRY(grow(st->out_rows, std::max<uint64_t>(rows, 1) * sizeof(uint32_t)));
  uint64_t *o_off = static_cast<uint64_t *>(st->out_off.p);
  uint32_t *o_hits = static_cast<uint32_t *>(st->out_hits.p);
  uint64_t *o_disp = o_off + rows + 1;  // (qdisp | hdisp for the rebasing kernel, behind the offsets)
  auto local_copy = [&](size_t s) -> int {
    if (nq[s]) BIVX_HIP(hipMemcpyAsync(o_off + qdisp[s], st->dev[s].off.p, nq[s] * sizeof(uint64_t), hipMemcpyDeviceToDevice, rs));
    if (nh[s]) BIVX_HIP(hipMemcpyAsync(o_hits + hdisp[s], st->dev[s].hits.p, nh[s] * sizeof(uint32_t), hipMemcpyDeviceToDevice, rs));
    return 0;
  };
  BIVX_TRY(local_copy(0));
  if (rccl && k > 1) {
    BIVX_NCCL(ncclGroupStart());
    ncclResult_t e = ncclSuccess;
    for (size_t s = 1; s < k && e == ncclSuccess; ++s) {
      if (nq[s]) e = ncclRecv(o_off + qdisp[s], nq[s], ncclUint64, (int)s, st->comms[0], rs);
      if (e == ncclSuccess && nh[s]) e = ncclRecv(o_hits + hdisp[s], nh[s], ncclUint32, (int)s, st->comms[0], rs);
      OnDevice g(st->devices[s]);  // (the peer's sends with the peer's device current; the root's is restored behind them)
      if (e == ncclSuccess && nq[s]) e = ncclSend(st->dev[s].off.p, nq[s], ncclUint64, 0, st->comms[s], st->dev[s].stream);
      if (e == ncclSuccess && nh[s]) e = ncclSend(st->dev[s].hits.p, nh[s], ncclUint32, 0, st->comms[s], st->dev[s].stream);
    }
    if (e != ncclSuccess) {
      (void)ncclGroupEnd();
      set_error("ncclSend / ncclRecv failed: %s", ncclGetErrorString(e));
      return BIVX_E_COMM;
    }
    BIVX_NCCL(ncclGroupEnd());
  } else {
    for (size_t s = 1; s < k; ++s) BIVX_TRY(local_copy(s));  // (shards that share the root's device: nothing to send)
  }
  // 4. offsets rebased to the gathered ids, the rows' batch indices
  std::vector<uint64_t> disp(2 * (k + 1));
  std::copy(qdisp.begin(), qdisp.end(), disp.begin());
  std::copy(hdisp.begin(), hdisp.end(), disp.begin() + (k + 1));
  BIVX_HIP(hipMemcpyAsync(o_disp, disp.data(), disp.size() * sizeof(uint64_t), hipMemcpyHostToDevice, rs));
  hipLaunchKernelGGL(k_rebase_offsets, dim3((unsigned)((rows + 1 + 255) / 256)), dim3(256), 0, rs, o_off, o_disp, o_disp + (k + 1),
                     (uint32_t)k, rows, total);
  BIVX_HIP(hipGetLastError());
  std::vector<uint32_t> row_q;
  row_q.reserve(rows);
  for (size_t s = 0; s < k; ++s) row_q.insert(row_q.end(), r.qs[s].begin(), r.qs[s].end());
  if (rows) BIVX_HIP(hipMemcpyAsync(st->out_rows.p, row_q.data(), rows * sizeof(uint32_t), hipMemcpyHostToDevice, rs));
  BIVX_HIP(hipStreamSynchronize(rs));
  for (size_t s = 1; s < k; ++s) {  // the peers' sends are complete when their streams are
    OnDevice g(st->devices[s]);
    BIVX_HIP(hipStreamSynchronize(st->dev[s].stream));
  }
  out->d_offsets = o_off;
  out->d_hit_ids = o_hits;
  out->d_query_of_row = static_cast<uint32_t *>(st->out_rows.p);
  out->rows = rows;
  out->total = total;
  out->device = st->devices[0];
  out->used_rccl = rccl ? 1 : 0;
  return 0;
}

static void release_dev_state(ShardedState *st) {
  for (auto c : st->comms)
    if (c) (void)ncclCommDestroy(c);
  st->comms.clear();
  for (size_t s = 0; s < st->dev.size(); ++s) {
    OnDevice g(st->devices[s]);
    ShardedState::ShardDev &d = st->dev[s];
    if (d.stream) (void)hipStreamSynchronize(d.stream);
    for (ShardedState::Buf *b : {&d.q, &d.off, &d.hits, &d.gid, &d.sizes}) (void)hipFree(b->p);
    if (d.stream) (void)hipStreamDestroy(d.stream);
  }
  st->dev.clear();
  if (!st->devices.empty()) {
    OnDevice g(st->devices[0]);
    for (ShardedState::Buf *b : {&st->out_off, &st->out_hits, &st->out_rows}) {
      (void)hipFree(b->p);
      b->p = nullptr;
      b->cap = 0;
    }
  }
}

void sharded_stats(const ShardedState *st, bivx_stats *out) {
  std::memset(out, 0, sizeof(*out));
  out->n_intervals = st->low.size();
  out->n_chroms = st->nchrom;
  out->build_ms = st->build_ms;
  for (auto *p : st->shard) {
    bivx_stats s;
    if (bivx_get_stats(p, &s) != 0) continue;
    out->n_segments += s.n_segments;
    out->n_cells += s.n_cells;
    out->index_bytes += s.index_bytes;
    out->staging_bytes += s.staging_bytes;
    out->prefix_timeouts += s.prefix_timeouts;
  }
}


// ---- the entry points capi.hip dispatches to: the implementations above behind the exception barrier ----------------
int sharded_create(ShardedState **out, const int *devices, int ndev) {
  return no_throw("bivx_create_sharded", [&] { return create_impl(out, devices, ndev); });
}
int sharded_append(ShardedState *st, const uint32_t *chrom, const uint32_t *low, const uint32_t *high,
                   const uint8_t *svtype, size_t n) {
  const size_t old = st->low.size();
  const int rc = no_throw("bivx_append", [&] { return append_impl(st, chrom, low, high, svtype, n); });
  if (rc != 0) {  // an append that ran out of memory half-way leaves the four columns at their old, equal length
    try {
      st->low.resize(std::min(st->low.size(), old));
      st->high.resize(std::min(st->high.size(), old));
      st->chrom.resize(std::min(st->chrom.size(), old));
      st->type.resize(std::min(st->type.size(), old));
    } catch (...) {
    }
  }
  return rc;
}
int sharded_clear(ShardedState *st) {
  return no_throw("bivx_clear", [&] { return clear_impl(st); });
}
int sharded_build(ShardedState *st) {
  return no_throw("bivx_build", [&] { return build_impl(st); });
}
int sharded_find_overlaps(const ShardedState *st, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh,
                          size_t q, const bivx_filter *filter, int sort_by_id, uint64_t *offsets_out,
                          uint32_t **hit_ids_out) {
  return no_throw("bivx_find_overlaps", [&] {
    return find_overlaps_impl(st, qchrom, qlow, qhigh, q, filter, sort_by_id, offsets_out, hit_ids_out);
  });
}
int sharded_count(const ShardedState *st, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh,
                  size_t q, const bivx_filter *filter, uint64_t *offsets_out) {
  return no_throw("bivx_count", [&] { return count_impl(st, qchrom, qlow, qhigh, q, filter, offsets_out); });
}
int sharded_fill(const ShardedState *st, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh,
                 size_t q, const bivx_filter *filter, const uint64_t *offsets, uint32_t *hit_ids_out,
                 int sort_by_id) {
  return no_throw("bivx_fill", [&] {
    return fill_impl(st, qchrom, qlow, qhigh, q, filter, offsets, hit_ids_out, sort_by_id);
  });
}
int sharded_any(const ShardedState *st, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh,
                size_t q, uint32_t *first_id_out) {
  return no_throw("bivx_any", [&] { return any_impl(st, qchrom, qlow, qhigh, q, first_id_out); });
}
int sharded_query_dev(const ShardedState *st, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh,
                      size_t q, int sort_by_id, bivx_sharded_result *out) {
  return no_throw("bivx_query_sharded_dev", [&] { return query_dev_impl(st, qchrom, qlow, qhigh, q, sort_by_id, out); });
}

}  // namespace bivx
