/*
 * bivx.h — C ABI of libbivx.so: the MI355X (gfx950) interval-overlap engine that stands behind
 * ylab-hi/BINARY's IntervalTree query API (binary::algorithm::tree, library/include/binary/algorithm/).
 *
 * The reference has no FFI/plugin boundary of its own: its boundary is the C++20 template API of a
 * header-only library. This header is the boundary a maintainer binds instead; every entry point cites
 * the reference interface it replaces (paths relative to the reference root). The C++20 facade in
 * include/binary/algorithm/interval_tree.hpp is the in-tree caller; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain pointers and sizes, no C++/torch types; no exceptions cross the ABI
 *   - return 0 on success, a negative bivx_status otherwise; bivx_last_error() gives the thread-local text
 *   - interval ids are 0-based APPEND ORDER indices (the reference's insertion order, rb_tree.hpp:111-117)
 *   - intervals are closed [low, high] over uint32 keys, overlap predicate exactly
 *     q.low <= high && low <= q.high (BaseInterval::is_overlap, interval_tree.hpp:119-121);
 *     low > high is accepted and handled exactly like the reference tree handles it
 *   - chrom ids partition the index: an interval and a query only meet when their chrom ids are equal
 *     (sv2nl builds one tree per chromosome, standalone/sv2nl/include/mapper.hpp:147-162,199);
 *     pass NULL chrom arrays for a single tree
 *   - calls taking `const bivx_index*` are thread-safe against each other; mutating calls are not. Host-pointer query
 *     calls made from several threads on one index run side by side (each on a stream, an error block and a workspace of
 *     its own: the reference shares one tree across its pool threads, mapper.cpp:127-142)
 *   - bivx_build and bivx_clear wait for device-pointer calls that still read the index on caller streams (for the whole
 *     device, if there was such a call since the last build); the caller does not have to synchronise before it rebuilds,
 *     and may destroy its streams whenever their work is done
 *   - `_dev` entry points take DEVICE pointers and a hipStream_t (as void*; NULL = default stream) and
 *     never synchronise; the others take HOST pointers and return when the result is in host memory
 *   - there is no CPU fallback: without a usable gfx950 device bivx_create fails with BIVX_E_HIP
 */
#ifndef BIVX_H_
#define BIVX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bivx_index bivx_index;
typedef struct bivx_filter bivx_filter; /* fused post-filter, defined below */

typedef enum bivx_status {
  BIVX_OK = 0,
  BIVX_E_INVALID = -1, /* bad argument */
  BIVX_E_HIP = -2,     /* a HIP runtime call failed (text in bivx_last_error) */
  BIVX_E_NOMEM = -3,   /* host allocation failed */
  BIVX_E_STATE = -4,   /* e.g. query before build */
  BIVX_E_RANGE = -5,   /* too many intervals (>= 2^32-1) or chromosome ids (> BIVX_MAX_CHROMS) */
  BIVX_E_COMM = -7,    /* an RCCL call of bivx_query_sharded_dev failed (text in bivx_last_error) */
  BIVX_E_TIMEOUT = -6  /* a single-pass query kernel gave up a bounded cross-workgroup wait, or found its prefix
                          workspace in an inconsistent state: the CSR of that call is INVALID (never returned with
                          rc 0); the index stays usable, repeat the call */
} bivx_status;

#define BIVX_MAX_CHROMS 65536u
#define BIVX_NO_HIT 0xFFFFFFFFu

/* ABI version of this header: major << 16 | minor. */
#define BIVX_ABI_VERSION 0x00020003u
uint32_t bivx_abi_version(void);
const char *bivx_last_error(void);

/* ---- lifetime ---------------------------------------------------------------------------------
 * replaces: IntervalTree<Node>{} construction / destruction (interval_tree.hpp:140, rb_tree.hpp:103,109);
 * the tree owns its nodes (rb_tree.hpp:169), the handle owns all device memory of the index. */
int bivx_create(bivx_index **out, int device);
void bivx_destroy(bivx_index *idx);
int bivx_device(const bivx_index *idx);
/* The reference builds a tree per task and drops it (mapper.hpp:147-162,199). A destroyed single-device index is therefore
 * kept — emptied, with its stream and its grow-only device and pinned blocks — for the next bivx_create on the same
 * device: a handful of objects per process (BIVX_INDEX_POOL, default 4; 0 = none), none above BIVX_INDEX_POOL_MB of
 * device memory (default 4096). hipFree / hipHostFree / hipStreamDestroy are otherwise 2 ms of a 2.6 ms
 * create - fill - build - query - drop cycle of a million intervals. This frees what is kept. */
void bivx_release_pooled(void);

/* ---- several GPUs of one node behind one handle ------------------------------------------------------
 * replaces: the reference's only decomposition, one task per chromosome on a thread pool
 * (standalone/sv2nl/include/mapper.hpp:238-246, main.cpp:34-44). Whole chromosomes are assigned to the ndev devices by
 * LPT on their interval counts; a query is answered by the device that holds its chromosome; no data
 * moves between devices. An index with fewer populated chromosomes than devices (a plain IntervalTree has one) is
 * replicated on every device and the queries are split instead. The handle takes the HOST-pointer entry points
 * (bivx_append*, bivx_build, bivx_count*, bivx_fill*, bivx_find_overlaps, bivx_any, bivx_get_*, bivx_get_stats), with
 * the same ids and results as a single-device index; the `_dev` entry points address one device's memory and return
 * BIVX_E_STATE for it. One host thread per device drives its shard. devices may name a device more than once.
 * (Across PROCESSES — one rank per GPU — shard with the same LPT and gather with RCCL: binary_amd/sharding.py.) */
int bivx_create_sharded(bivx_index **out, const int *devices, int ndev);
int bivx_num_devices(const bivx_index *idx);                     /* 1 for a bivx_create handle */
int bivx_device_of_chrom(const bivx_index *idx, uint32_t chrom); /* after build; -1: no interval on that chromosome */

/* The batch answered by a sharded handle with the result left ON THE DEVICE: the final gatherv of the per-chromosome hit
 * lists (BASELINE north_star) below the C ABI.
 * replaces: the reference's tasks handing their hit lists to one writer (mapper.hpp:238-246, mapper.cpp:127-142,
 * writer.hpp:27-53) — here every device answers the queries of ITS chromosomes into its own device buffers
 * (bivx_query_dev_s per shard, ids mapped to the handle's global append-order ids on the device), then the CSRs are
 * gathered into devices[0]'s memory over RCCL: ncclCommInitAll over the handle's devices (once per handle),
 * ncclAllGather of (queries, ids) per device, one ncclGroupStart .. ncclSend / ncclRecv .. ncclGroupEnd in which every
 * peer's offsets and ids travel to device 0 over its own xGMI link, and a small kernel that rebases the offsets.
 * Queries are HOST arrays (they are routed to their chromosome's device on the host, as for bivx_find_overlaps).
 * Result (device memory of out->device, owned by the handle, valid until the next bivx_query_sharded_dev, bivx_build,
 * bivx_clear or bivx_destroy on it): rows grouped by device — d_query_of_row[r] is the batch index of the query row r
 * answers (with one device: 0, 1, 2, ...: the result is bivx_query_dev_s's CSR bit for bit); lists in index order, or
 * ascending id with sort_by_id. A handle that names a device twice (tests on a one-GPU box) has no communicator — RCCL
 * wants distinct devices — and moves the same blocks with device-to-device copies instead. A bivx_create handle
 * gets BIVX_E_STATE (bivx_query_dev_s is its call). Every nccl* failure comes back as BIVX_E_COMM. */
typedef struct bivx_sharded_result {
  uint64_t *d_offsets;      /* u64[rows + 1] */
  uint32_t *d_hit_ids;      /* u32[total], global (append-order) ids */
  uint32_t *d_query_of_row; /* u32[rows] */
  uint64_t rows, total;
  int device;               /* the device that holds the three arrays: devices[0] */
  int used_rccl;            /* 1: the blocks travelled through ncclSend / ncclRecv; 0: device-to-device copies */
} bivx_sharded_result;
int bivx_query_sharded_dev(const bivx_index *idx, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh,
                           size_t q, int sort_by_id, bivx_sharded_result *out);

/* ---- build side -------------------------------------------------------------------------------
 * replaces: RbTree::insert_node(range) rb_tree.hpp:111-117 and insert_node(Args&&...) :145-149 — appends n
 * intervals; ids continue from bivx_size(). chrom may be NULL (all chrom 0). Host pointers. */
int bivx_append(bivx_index *idx, const uint32_t *chrom, const uint32_t *low, const uint32_t *high, size_t n);
/* same, device pointers, asynchronous on `stream` */
int bivx_append_dev(bivx_index *idx, const uint32_t *d_chrom, const uint32_t *d_low, const uint32_t *d_high,
                    size_t n, void *stream);
/* replaces: the (chrom, start, end, svtype) columns of the records that become tree nodes — BaseVcfInterval
 * (library/include/binary/parser/vcf.hpp:598-639: low = record.pos, high = record.info->svend) filtered by svtype
 * before insertion (standalone/sv2nl/include/mapper.hpp:153-156). svtype[i] in 1..255 labels interval i (0 = no
 * type; bivx_append appends type 0). The built index is partitioned by (chromosome, svtype): a query whose
 * bivx_filter::svtype is t meets only intervals of type t, at no cost per candidate, so ONE index serves sv2nl's three
 * mappers where the reference builds a tree per (mapper, chromosome). Host / device pointers as for bivx_append*. */
int bivx_append_typed(bivx_index *idx, const uint32_t *chrom, const uint32_t *low, const uint32_t *high,
                      const uint8_t *svtype, size_t n);
int bivx_append_typed_dev(bivx_index *idx, const uint32_t *d_chrom, const uint32_t *d_low, const uint32_t *d_high,
                          const uint8_t *d_svtype, size_t n, void *stream);
/* drops all intervals (keeps the allocation) */
int bivx_clear(bivx_index *idx);

/* replaces the net effect of IntervalTree::insert_node_impl interval_tree.hpp:230-260 + RbTree::fix_insert
 * rb_tree.hpp:304-344 + the rotations :206-228: makes the appended set searchable. On the GPU that is a
 * per-chromosome, per-length-class start-sorted array with a bucket directory. Idempotent;
 * runs on the index's own stream and returns when the index is ready. */
int bivx_build(bivx_index *idx);
int bivx_is_built(const bivx_index *idx);

/* replaces: RbTree::size() rb_tree.hpp:173-180 (number of intervals appended) */
size_t bivx_size(const bivx_index *idx);
/* number of chromosome ids in use after build (max id + 1) */
uint32_t bivx_num_chroms(const bivx_index *idx);

/* largest svtype in the built index + 1 (1 when no interval carries a type) */
uint32_t bivx_num_types(const bivx_index *idx);
/* the svtype column by id (0xFF for ids out of range). Host pointers. */
int bivx_get_svtypes(const bivx_index *idx, const uint32_t *ids, size_t n, uint8_t *svtype_out);

/* reads back appended intervals by id (what the facade needs to materialise interval_type copies,
 * interval_tree.hpp:316). Host pointers; any output may be NULL. */
int bivx_get_intervals(const bivx_index *idx, const uint32_t *ids, size_t n, uint32_t *chrom_out,
                       uint32_t *low_out, uint32_t *high_out);

/* ---- query side: all overlaps ------------------------------------------------------------------
 * replaces: IntervalTree::find_overlaps interval_tree.hpp:161-168,306-334, batched over q queries.
 * Result is CSR: offsets[q+1] (exclusive prefix of per-query hit counts, offsets[q] == total hits H)
 * and hit_ids[H]; query i's hits are hit_ids[offsets[i] .. offsets[i+1]).
 * Hit order inside one query: deterministic "index order" straight out of bivx_fill* — ascending (length class,
 * directory cell, id), or (length class, low, id) where the build sorted on every bit of low; do not rely on which —,
 * ascending id after bivx_sort_hits* / with sort_by_id. The reference returns RB-tree pre-order, a
 * function of insertion history; as a SET the result is identical (tests compare sorted lists).
 *
 * Two calls because H is unknown a priori: count -> caller allocates hit_ids[H] -> fill. */
int bivx_count(const bivx_index *idx, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh,
               size_t q, uint64_t *offsets_out);
int bivx_fill(const bivx_index *idx, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh,
              size_t q, const uint64_t *offsets, uint32_t *hit_ids_out, int sort_by_id);

/* One call for host callers: uploads the queries once, counts, sizes the result, enumerates (ids ascending inside each
 * query when sort_by_id != 0) and downloads. *hit_ids_out is allocated by the library (NULL when there is no hit);
 * release it with bivx_free. `filter` may be NULL. Up to 2048 unfiltered queries take one device round trip (one
 * upload, one single-pass launch, one download), which is what a caller that keeps the reference's one
 * find_overlaps per record pays per call. */
int bivx_find_overlaps(const bivx_index *idx, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh,
                       size_t q, const bivx_filter *filter, int sort_by_id, uint64_t *offsets_out,
                       uint32_t **hit_ids_out);
void bivx_free(void *p);

/* device-resident variants. bivx_count_dev is one launch of the single-pass kernel with a zero-capacity hit
 * buffer (it counts, chains the prefix across workgroups and writes d_offsets[q + 1]); it needs no caller scratch
 * (ABI 2.0 dropped the workspace arguments it had ignored since 1.3). Like bivx_query_dev without a caller
 * workspace it uses the index's per-stream workspace, which the first call on a stream allocates: make that call
 * before capturing the stream into a graph (inside a capture it fails with BIVX_E_STATE). */
int bivx_count_dev(const bivx_index *idx, const uint32_t *d_qchrom, const uint32_t *d_qlow,
                   const uint32_t *d_qhigh, size_t q, uint64_t *d_offsets, void *stream);
int bivx_fill_dev(const bivx_index *idx, const uint32_t *d_qchrom, const uint32_t *d_qlow,
                  const uint32_t *d_qhigh, size_t q, const uint64_t *d_offsets, uint32_t *d_hit_ids,
                  void *stream);
/* Single pass for callers that already own a hit buffer (steady-state serving): one kernel counts, chains
 * the prefix across workgroups and fills. d_offsets[q] receives the true total H even when H >
 * hit_capacity; in that case only the first hit_capacity slots of the CSR were written and the caller
 * repeats the call (or bivx_fill_dev) with a buffer of at least H entries (with ascending ids the one list that
 * straddles the capacity holds some of its ids below it, not necessarily its smallest). d_workspace: NULL (the index keeps
 * one 4.5 MB workspace per stream, which needs no clearing between calls), or bivx_query_workspace_bytes(q)
 * bytes of caller scratch (cleared by a memset in front of every launch). */
size_t bivx_query_workspace_bytes(size_t q);
int bivx_query_dev(const bivx_index *idx, const uint32_t *d_qchrom, const uint32_t *d_qlow,
                   const uint32_t *d_qhigh, size_t q, uint64_t *d_offsets, uint32_t *d_hit_ids,
                   uint64_t hit_capacity, void *d_workspace, size_t workspace_bytes, void *stream);
/* Error state of the asynchronous entry points. The single-pass kernels (bivx_query_dev*, bivx_count_dev*) chain a
 * prefix across workgroups; a workgroup whose bounded wait for a predecessor expires (about 20 s: the device
 * stopped making progress), or that finds the prefix workspace inconsistent, raises a sticky flag instead of
 * hanging. bivx_stream_status synchronises `stream`, returns BIVX_E_TIMEOUT if a flag was raised by any call on
 * this index since the last report (that call's CSR is invalid) and clears it, else 0. The host-pointer entry
 * points (bivx_count*, bivx_fill*, bivx_find_overlaps) make the same check before they return, so they never hand
 * out such a result with rc 0 (reference analogue: none, the CPU tree cannot fail this way; SURVEY.md §5 asks
 * that every status is surfaced). */
int bivx_stream_status(const bivx_index *idx, void *stream);

/* Which kernel bivx_query_dev* runs for such a call: "k_query_pipe" (the pipelined form, for batches of about 0.8 M
 * queries and more on an index with one length class per chromosome, no post-filter, at most 6 ids per query of
 * capacity), "k_query_pipe_dense|k_query_fused" (2 M queries and more, more ids per query, index order: both are
 * launched, a device-side probe of the query order picks the first for a position-sorted batch, the second otherwise,
 * and the other returns at once) or "k_query_fused" (everything else). Same results whichever runs; the names are what
 * shows up in a rocprofv3 kernel trace (bench.py reports the one that did the work as the dominant kernel). */
const char *bivx_query_kernel_name(const bivx_index *idx, size_t q, uint64_t hit_capacity, int sort_by_id,
                                   const bivx_filter *filter);

/* sorts every query's hit list ascending by id, in place */
int bivx_sort_hits_dev(const bivx_index *idx, const uint64_t *d_offsets, uint32_t *d_hit_ids, size_t q,
                       void *stream);

/* ---- query side: existence ---------------------------------------------------------------------
 * replaces: IntervalTree::find_overlap interval_tree.hpp:152-159,290-304 (std::optional<interval_type>).
 * first_id_out[i] = smallest id among query i's hits, or BIVX_NO_HIT. The reference returns whichever
 * overlapping node its single root-to-leaf descent meets first (tree-shape dependent, and it misses
 * hits when q.low == 0 meets a null left child, interval_tree.hpp:297 with get_max(nullptr) == 0);
 * this entry point is exact on existence and canonical on choice (documented deviation). */
int bivx_any(const bivx_index *idx, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh,
             size_t q, uint32_t *first_id_out);
int bivx_any_dev(const bivx_index *idx, const uint32_t *d_qchrom, const uint32_t *d_qlow,
                 const uint32_t *d_qhigh, size_t q, uint32_t *d_first_id, void *stream);

/* ---- fused post-filters ---------------------------------------------------------------------------
 * replaces: the check_condition step that sv2nl applies to every find_overlaps hit
 * (standalone/sv2nl/include/mapper.hpp:219-223; DupMapper / InvMapper / TraMapper::check_condition,
 * source/mapper.cpp:50-79,144-156; predicates include/helper.hpp:16-46,76-82). A candidate that overlaps the
 * query must ALSO pass the filter to be counted / reported, so only surviving pairs are ever written.
 * Operands: the query interval is the validated NL record (pos, svend), the stored interval the SV record as
 * inserted. Everything is unsigned-safe |a - b| arithmetic on uint32, as in the reference.
 *   BIVX_FILTER_SV2NL_DUP  low <= q.low && high >= q.high && |q.low-low| <= d && |q.high-high| <= d
 *   BIVX_FILTER_SV2NL_INV  neither interval contains the other && both distances <= d && (use_strand == 0 ||
 *                          (q.low <= low ? s1 && !s2 : !s1 && s2)); query_aux bit0 = STRAND1 is '+', bit1 = STRAND2
 *   BIVX_FILTER_SV2NL_TRA  aux = pair_id << 1 | swapped, on both sides (pair_id: id of the record's ordered
 *                          chromosome pair; swapped: chrom > chr2, i.e. breakpoint 1 is `high`): equal pair_id
 *                          and |bp1 - bp1'| <= d and |bp2 - bp2'| <= d
 * query_aux has one word per query, interval_aux one word per interval in append order; host pointers for the
 * host entry points, device pointers for the _dev ones; either may be NULL when the kind does not read it. */
enum { BIVX_FILTER_NONE = 0, BIVX_FILTER_SV2NL_DUP = 1, BIVX_FILTER_SV2NL_INV = 2, BIVX_FILTER_SV2NL_TRA = 3 };
struct bivx_filter {
  uint32_t kind;
  uint32_t max_dist;   /* sv2nl --dis */
  uint32_t use_strand; /* sv2nl: !--short */
  uint32_t svtype;     /* 0: intervals of every type; t in 1..255: only intervals appended with svtype t (a selection
                          of index partitions, not a per-candidate test; valid with kind == BIVX_FILTER_NONE too) */
  const uint32_t *query_aux;
  const uint32_t *interval_aux;
};

/* the filtered forms of count / fill / query; filter == NULL or kind == BIVX_FILTER_NONE gives the plain calls */
int bivx_count_f(const bivx_index *idx, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh,
                 size_t q, const bivx_filter *filter, uint64_t *offsets_out);
int bivx_fill_f(const bivx_index *idx, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh,
                size_t q, const bivx_filter *filter, const uint64_t *offsets, uint32_t *hit_ids_out, int sort_by_id);
int bivx_count_dev_f(const bivx_index *idx, const uint32_t *d_qchrom, const uint32_t *d_qlow,
                     const uint32_t *d_qhigh, size_t q, const bivx_filter *filter, uint64_t *d_offsets, void *stream);
int bivx_fill_dev_f(const bivx_index *idx, const uint32_t *d_qchrom, const uint32_t *d_qlow,
                    const uint32_t *d_qhigh, size_t q, const bivx_filter *filter, const uint64_t *d_offsets,
                    uint32_t *d_hit_ids, void *stream);
int bivx_query_dev_f(const bivx_index *idx, const uint32_t *d_qchrom, const uint32_t *d_qlow,
                     const uint32_t *d_qhigh, size_t q, const bivx_filter *filter, uint64_t *d_offsets,
                     uint32_t *d_hit_ids, uint64_t hit_capacity, void *d_workspace, size_t workspace_bytes,
                     void *stream);
/* bivx_query_dev_f whose ids leave in ascending order inside every query when sort_by_id != 0 — the order
 * bivx_sort_hits_dev gives, produced by the same kernel on the ids' way out instead of by a second pass over
 * the CSR (the facade's find_overlaps order, interval_tree.hpp:306-334 as a set). */
int bivx_query_dev_s(const bivx_index *idx, const uint32_t *d_qchrom, const uint32_t *d_qlow,
                     const uint32_t *d_qhigh, size_t q, const bivx_filter *filter, int sort_by_id,
                     uint64_t *d_offsets, uint32_t *d_hit_ids, uint64_t hit_capacity, void *d_workspace,
                     size_t workspace_bytes, void *stream);

/* The single pass without the canonical CSR: for callers that only need every query's hits, not one monotone
 * offsets array (sv2nl consumes one NL record's hits at a time, mapper.hpp:205-231). A workgroup reserves the
 * output range of its 1024 queries with one atomic add and waits for nobody, so the cross-workgroup prefix —
 * about a tenth of bivx_query_dev's time at 1 M queries — disappears. (On batches of 4 M queries and more the
 * pipelined kernel's ordered CSR is the faster answer and is what this call returns: d_begin = the offsets.) Query i's hits
 * are d_hit_ids[d_begin[i] .. d_begin[i] + d_count[i]) in index order; ranges of different workgroups lie in
 * the buffer in arrival order (not reproducible between calls), the sets are exactly those of bivx_query_dev.
 * *d_total receives the number of ids reserved; when it exceeds hit_capacity only slots below the capacity were
 * written and the call is repeated with a larger buffer (hit_capacity 0 gives counts and the total only).
 * Workspace as for bivx_query_dev. */
int bivx_query_dev_u(const bivx_index *idx, const uint32_t *d_qchrom, const uint32_t *d_qlow,
                     const uint32_t *d_qhigh, size_t q, const bivx_filter *filter, uint64_t *d_begin,
                     uint32_t *d_count, uint32_t *d_hit_ids, uint64_t hit_capacity, uint64_t *d_total,
                     void *d_workspace, size_t workspace_bytes, void *stream);

/* The index overlapped with itself: query i is appended interval i, i.e. the CSR bivx_query_dev_s gives for the
 * appended columns as the batch (d_offsets[n + 1], n = bivx_size; query i's own id is among its hits) — the
 * whole-genome self-overlap of one SV call set (BASELINE config 5; reference pattern: build the tree from a file's
 * records and run find_overlaps for each of the same records, mapper.hpp:199-218 with sv == nl). The library answers
 * in the index's OWN order, where neighbouring queries read neighbouring records (every line of the index is fetched
 * once per wavefront instead of once per query), and then gathers the lists into id order; no query arrays are read.
 * Same offsets and the same ids as the general call, list by list in the same order. It keeps a per-stream scratch of
 * 8 n + 4 hit_capacity bytes (hit_capacity below 2^38, else BIVX_E_RANGE). d_offsets[n] is the true total even when it
 * exceeds hit_capacity — the contents of d_hit_ids are then unspecified and the call is repeated with a buffer of at
 * least the total (hit_capacity 0 gives the offsets only). Indexes the fast path does not cover (several length classes
 * per chromosome, positional hotspots, fewer than 61 440 or more than 62.9 M intervals) take the general call, with its
 * semantics. */
int bivx_self_overlaps_dev(const bivx_index *idx, int sort_by_id, uint64_t *d_offsets, uint32_t *d_hit_ids,
                           uint64_t hit_capacity, void *stream);

/* Testing hook (tests/test_gpu_errors.py): overwrites the ticket word of the prefix workspace the index keeps for
 * `stream`, as a launch that died half-way would leave it, so that the next single-pass call on that stream finds
 * it inconsistent and the error path above can be exercised. Not for production use. */
int bivx_debug_corrupt_workspace(const bivx_index *idx, void *stream);

/* ---- introspection (bench / DESIGN.md numbers) -------------------------------------------------- */
typedef struct bivx_stats {
  uint64_t n_intervals;
  uint32_t n_chroms;
  uint32_t n_segments;      /* (chromosome, length class) pairs */
  uint64_t n_cells;         /* bucket directory entries */
  uint64_t index_bytes;     /* device bytes of the built index (sorted arrays + directory) */
  uint64_t staging_bytes;   /* device bytes of the append-order copy */
  double build_ms;          /* wall time of the last bivx_build */
  uint64_t prefix_timeouts; /* calls on this index reported with BIVX_E_TIMEOUT since it was created (plus one while a
                               raised flag waits to be reported); bivx_build does not reset it */
} bivx_stats;
int bivx_get_stats(const bivx_index *idx, bivx_stats *out);

#ifdef __cplusplus
}
#endif
#endif /* BIVX_H_ */
