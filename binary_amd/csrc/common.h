// common.h — shared declarations of libbivx (gfx950 only; no CUDA path, no CPU fallback).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/bivx.h"

namespace bivx {

void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

#define BIVX_HIP(call)                                                                              \
  do {                                                                                              \
    hipError_t bivx_e_ = (call);                                                                    \
    if (bivx_e_ != hipSuccess) {                                                                    \
      ::bivx::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(bivx_e_), __FILE__,       \
                        __LINE__);                                                                  \
      return BIVX_E_HIP;                                                                            \
    }                                                                                               \
  } while (0)

#define BIVX_TRY(call)              \
  do {                              \
    int bivx_rc_ = (call);          \
    if (bivx_rc_ != 0) return bivx_rc_; \
  } while (0)

constexpr int kWave = 64;          // gfx950 wavefront
constexpr int kLenBins = 33;       // bin 0: len == 0 (incl. low > high); bin b: len in [2^(b-1), 2^b)

// One (chromosome, length class) segment of the sorted arrays plus its bucket directory.
struct SegDesc {        // 32 B, read as two dwordx4
  uint32_t begin, end;  // element range [begin, end) in se[] / id[]
  uint32_t base, last;  // min / max `low` in the segment
  uint32_t shift;       // bits 0-4: directory cell of coordinate x is (x - base) >> shift; bit 8: kSegPacked
  uint32_t table_off;   // first directory entry; the segment owns ncell + 1 entries
  uint32_t maxlen;      // max over the segment of (high >= low ? high - low : 0)
  uint32_t ncell;
};
static_assert(sizeof(SegDesc) == 32, "SegDesc layout");
// Segment also has packed records rec[i] = ((low & 0xFFFF) | (high - low) << 16, id): needs maxlen <= 65535 and no
// low > high entry; a query decodes them when its window's cells cover at most 65536 coordinates.
constexpr uint32_t kSegPacked = 1u << 8;

// Device view of a built index, passed to kernels by value.
struct IndexView {
  const uint2 *se;            // (low, high) sorted by (segment, low, id)
  const uint2 *rec;           // packed (record, id) pairs (meaningful in kSegPacked segments only)
  const uint32_t *id;         // append-order id of each sorted slot
  const uint32_t *table;      // bucket directories, all segments back to back
  const SegDesc *seg;         // nseg descriptors, grouped by chromosome
  // segments of chromosome c as (first, count), for the interval type this call asks for: the index keeps one such
  // row per svtype (row 0: every type), so selecting a type costs the kernels nothing (bivx_filter::svtype)
  const uint2 *chrom_rng;
  uint32_t nchrom;
  uint32_t nseg;
  uint32_t max_segs;          // most segments any one chromosome has (for the selected type)
  uint32_t nslots;            // sorted slots (= intervals built) — saturates at 2^32 - 1
  uint32_t max_cell;          // most slots any directory cell holds (positional hotspots make this large)
  uint32_t max_window;        // slots a query's window is EXPECTED to hold in the segment where that is most (the planner's estimate)
  // Slots are ordered by (segment, (low - segment base) >> order_shift, id): 0 = by every bit of low; otherwise by directory
  // cell (order_shift <= every segment's cell shift) with append order inside a cell — what a query needs, since it
  // evaluates every slot of the cells it touches; the build then sorts one radix pass less (capi.hip, bivx_build)
  uint32_t order_shift;
  // optional post-filter fused into the enumeration (bivx_filter): a candidate must pass it as well
  uint32_t flt_kind;          // BIVX_FILTER_*
  uint32_t flt_dist;
  uint32_t flt_strand;
  const uint32_t *flt_qaux;   // per query
  const uint32_t *flt_iaux;   // per interval, append order
  // the index's error block: host memory mapped into the device; kernels only ever store 1 into its words
  uint32_t *err;
};
constexpr uint32_t kErrTimeout = 0;    // a bounded cross-workgroup wait of the single-pass kernel expired
constexpr uint32_t kErrWorkspace = 1;  // the prefix workspace was not zero when a launch began
constexpr uint32_t kErrWords = 16;     // size of the block (one cache line)

// per (chromosome, length bin) statistics gathered before the sort
struct BinStats {
  uint32_t count;
  uint32_t min_low;
  uint32_t max_low;
  uint32_t max_len;
  uint32_t n_inverted;  // entries with low > high
};

// ---- sharded.cpp: one index over several devices (bivx_create_sharded); capi.hip dispatches to these ---------
struct ShardedState;
int sharded_create(ShardedState **out, const int *devices, int ndev);
void sharded_destroy(ShardedState *st);
int sharded_num_devices(const ShardedState *st);
int sharded_device_of_chrom(const ShardedState *st, uint32_t chrom);
int sharded_append(ShardedState *st, const uint32_t *chrom, const uint32_t *low, const uint32_t *high,
                   const uint8_t *svtype, size_t n);
int sharded_clear(ShardedState *st);
size_t sharded_size(const ShardedState *st);
bool sharded_is_built(const ShardedState *st);
uint32_t sharded_num_chroms(const ShardedState *st);
uint32_t sharded_num_types(const ShardedState *st);
int sharded_build(ShardedState *st);
int sharded_get_intervals(const ShardedState *st, const uint32_t *ids, size_t n, uint32_t *chrom_out,
                          uint32_t *low_out, uint32_t *high_out);
int sharded_get_svtypes(const ShardedState *st, const uint32_t *ids, size_t n, uint8_t *out);
int sharded_find_overlaps(const ShardedState *st, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh,
                          size_t q, const bivx_filter *filter, int sort_by_id, uint64_t *offsets_out,
                          uint32_t **hit_ids_out);
int sharded_count(const ShardedState *st, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh,
                  size_t q, const bivx_filter *filter, uint64_t *offsets_out);
int sharded_fill(const ShardedState *st, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh,
                 size_t q, const bivx_filter *filter, const uint64_t *offsets, uint32_t *hit_ids_out,
                 int sort_by_id);
int sharded_any(const ShardedState *st, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh,
                size_t q, uint32_t *first_id_out);
void sharded_stats(const ShardedState *st, bivx_stats *out);
// the gathered device-resident CSR (bivx_query_sharded_dev)
int sharded_query_dev(const ShardedState *st, const uint32_t *qchrom, const uint32_t *qlow, const uint32_t *qhigh,
                      size_t q, int sort_by_id, bivx_sharded_result *out);

// ---- scan.hip ---------------------------------------------------------------------------------------
// out[0..n] = exclusive prefix sums of in[0..n), out[n] = total. scratch: scan_scratch_bytes(n).
size_t scan_scratch_bytes(size_t n);
int exclusive_scan_u32_u64(const uint32_t *d_in, uint64_t *d_out, size_t n, void *d_scratch, hipStream_t s);
// bivx_self_overlaps_dev keeps, per id, ONE word: the list's length above the 38-bit position where it begins in the scratch
// (2^38 ids are 1.1 TB; a launch has fewer than 2^26 queries, so a length fits the 26 bits above)
constexpr int kSelfPosBits = 38;
constexpr uint64_t kSelfPosMask = (1ull << kSelfPosBits) - 1ull;
int exclusive_scan_lengths_u64(const uint64_t *d_in, uint64_t *d_out, size_t n, void *d_scratch, hipStream_t s);
// the same lengths summed per block of 256 ids (block_sums) and, per tile of 8 192 ids, the exclusive prefix of the tile sums
// (tile_prefix; the grand total behind the last tile), both inside d_scratch (scan_scratch_bytes): what k_permute_lines needs to
// make the offsets itself
int self_length_sums(const uint64_t *d_in, size_t n, void *d_scratch, const uint64_t **tile_prefix, const uint64_t **block_sums,
                     hipStream_t s);

// ---- build.hip --------------------------------------------------------------------------------------
// the partition of interval i is chrom[i] * ntypes + type[i] (type == nullptr: chrom[i])
int launch_bin_stats(const uint32_t *d_chrom, const uint8_t *d_type, uint32_t ntypes, const uint32_t *d_low,
                     const uint32_t *d_high, size_t n, uint32_t nparts, BinStats *d_stats, hipStream_t s);
// without svtypes: the statistics of the chromosomes below bin_stats_auto_parts() without knowing their number (the host
// reads it off the non-empty rows); d_scal[0] != 0 afterwards: the largest id BEYOND the table — the two-step form must run
uint32_t bin_stats_auto_parts();
int launch_bin_stats_auto(const uint32_t *d_chrom, const uint32_t *d_low, const uint32_t *d_high, size_t n, BinStats *d_stats,
                          uint32_t *d_scal, hipStream_t s);
// the index's own auto-form table ([4 scalars in 256 bytes | the entries]): emptied by launch_init_auto_stats, added to by
// every untyped append — by the pass that copies the columns into the index when they come from device memory (d_out_*)
size_t auto_stats_bytes();
int launch_init_auto_stats(void *d_block, hipStream_t s);
int launch_append_stats(const uint32_t *d_chrom, const uint32_t *d_low, const uint32_t *d_high, size_t n, uint32_t *d_out_chrom,
                        uint32_t *d_out_low, uint32_t *d_out_high, void *d_block, hipStream_t s);
// d_out2[0] = max chromosome id, d_out2[1] = max svtype (d_type may be nullptr); atomic maxima into words the caller zeroed
int launch_max_chrom_type(const uint32_t *d_chrom, const uint8_t *d_type, size_t n, uint32_t *d_out2, hipStream_t s);
// out[i] = src ? src[ids[i]] : 0 for ids[i] < n_src, else 0xFF
int launch_gather_u8(const uint8_t *d_src, const uint32_t *d_ids, size_t n, size_t n_src, uint8_t *d_out,
                     hipStream_t s);
// sort keys. mode 0 (dense): keys[i] = segkey[seg(i)].x + (low[i] - segkey[seg(i)].y) — (segment, low) in one word;
// mode 1: keys[i] = low[i] and d_seg_of[i] = seg(i); mode 2: keys[i] = d_seg_of[ids[i]] (second stage of the two-stage sort)
enum : int { kBuildKeyDense = 0, kBuildKeyLow = 1, kBuildKeySegOfId = 2 };
int launch_make_keys(int mode, const uint32_t *d_chrom, const uint8_t *d_type, uint32_t ntypes, const uint32_t *d_low,
                     const uint32_t *d_high, size_t n, const uint32_t *d_bin2seg, const uint2 *d_segkey,
                     const uint32_t *d_ids, uint32_t *d_seg_of, uint32_t *d_keys, void *d_hist0, hipStream_t s,
                     int first_shift = 0);
// stable LSD radix sort of (key, val) pairs on key bits [0, nbits); result ends in (*keys, *vals)
// (the pointers are swapped with the alt buffers as passes ping-pong). scratch: radix_scratch_bytes(n).
// vals_are_iota: the values are 0 .. n-1 and need not exist in memory yet (the first pass writes them).
// hist0_ready: launch_make_keys left the first pass's histogram in d_scratch.
size_t radix_scratch_bytes(size_t n);
int radix_sort_pairs(uint32_t **keys, uint32_t **vals, uint32_t **keys_alt, uint32_t **vals_alt, size_t n,
                     int nbits, void *d_scratch, bool vals_are_iota, bool hist0_ready, hipStream_t s, int first_shift = 0);
// se[], rec[] and the bucket directory (with its three spare entries) from the sorted ids (and the sorted dense keys;
// d_keys == nullptr: low is gathered by id). d_gaps: finalize_gap_bytes() of scratch; *d_ngaps and *d_max_cell must be
// zero; *d_max_cell receives the largest number of slots any directory cell holds.
size_t finalize_gap_bytes(uint64_t nentries, uint32_t nseg);
int launch_finalize(const uint32_t *d_keys, const uint32_t *d_ids, const uint32_t *d_low, const uint32_t *d_high,
                    const SegDesc *d_seg, const uint2 *d_segkey, uint32_t nseg, uint2 *d_se, uint2 *d_rec,
                    uint32_t *d_table, uint64_t nentries, void *d_gaps, uint32_t *d_ngaps, uint32_t *d_max_cell,
                    size_t n, hipStream_t s);
int launch_gather_intervals(const uint32_t *d_chrom, const uint32_t *d_low, const uint32_t *d_high,
                            const uint32_t *d_ids, size_t n, size_t n_intervals, uint32_t *d_c, uint32_t *d_l,
                            uint32_t *d_h, hipStream_t s);
// the built index's intervals as queries in slot order: (chromosome of the slot's segment, low, high)
int launch_self_queries(const uint2 *d_se, const SegDesc *d_seg, const uint32_t *d_seg_chrom, uint32_t nseg, size_t n,
                        uint32_t *d_qchrom, uint32_t *d_qlow, uint32_t *d_qhigh, hipStream_t s);

// ---- query.hip, query_fused.hip ---------------------------------------------------------------------------
int launch_query_tiny(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                      size_t q, uint64_t *d_offsets, uint32_t *d_hits, uint64_t cap, hipStream_t s);
int launch_fill(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                size_t q, const uint64_t *d_offsets, uint32_t *d_hits, hipStream_t s);
int launch_any(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
               size_t q, uint32_t *d_first, hipStream_t s);
// d_cond != nullptr: the pass runs only if *d_cond == seq (the single-pass kernel asks for it that way).
// total_hint: the number of ids if the caller knows it (0: `cap` is taken as an upper bound; the kernel's LDS stage, and
// with it how many wavefronts a CU holds, is sized by the average list)
int launch_sort_hits(const uint64_t *d_offsets, uint32_t *d_hits, size_t q, uint64_t cap, hipStream_t s,
                     const uint32_t *d_cond = nullptr, uint32_t seq = 0, uint64_t total_hint = 0);
// single pass: offsets[q+1] and hits (slots below cap only) in one kernel; ws: fused_workspace_bytes(q)
size_t fused_workspace_bytes(size_t q);
int launch_query_fused(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow,
                       const uint32_t *d_qhigh, size_t q, uint64_t *d_offsets, uint32_t *d_hits, uint64_t cap,
                       void *d_ws, bool self_clean, bool sort_ids, hipStream_t s, uint32_t *d_counts = nullptr,
                       uint64_t *d_total = nullptr);  // d_counts != nullptr: unordered begin/count output

// query_pipe.hip: the pipelined single-pass kernel for the common case; launch_query_fused dispatches to it
bool pipe_eligible(const IndexView &v, size_t q, uint64_t cap, bool sort_ids, bool unordered);
size_t pipe_queries_per_launch();
size_t pipe_ms_queries_per_launch();  // (k_query_pipe_ms has tiles of its own size)
bool pipe_dense_eligible(const IndexView &v, size_t q, uint64_t cap, bool sort_ids, bool unordered);
int launch_query_pipe_dense(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                            size_t q0, size_t q1, uint64_t *d_offsets, uint32_t *d_hits, uint64_t cap, uint64_t *ws,
                            int flags, uint32_t seq, hipStream_t s);
// ... and for everything else that is large: several segments per chromosome, fused filters, many ids per query
// (skip_seq != 0: launched behind k_query_pipe_dense, returns if that kernel took the launch)
bool pipe_ms_eligible(const IndexView &v, size_t q, uint64_t cap, bool unordered);
int launch_query_pipe_ms(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                         size_t q0, size_t q1, uint64_t *d_offsets, uint32_t *d_hits, uint64_t cap, uint64_t *ws,
                         int flags, uint32_t skip_seq, hipStream_t s);
// The index overlapped with itself (queries = its intervals in slot order, results wanted in id order: d_perm = the slots'
// ids): k_query_pipe_dense writes the lists in slot order into d_tmp_hits and leaves d_src_by_id[id] = a list's length <<
// kSelfPosBits | where it begins (cap == 0: the lengths only); launch_permute_lists then gathers list i to d_hits[offsets[i]].
bool self_overlaps_eligible(const IndexView &v, size_t n);
int launch_self_overlaps(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                         const uint32_t *d_perm, size_t n, uint64_t *d_src_by_id,
                         uint64_t *d_offsets_scratch, uint32_t *d_tmp_hits, uint64_t cap, uint64_t *ws, bool self_clean,
                         hipStream_t s);
// also leaves d_offsets (n + 1): the exclusive prefix sum of the lists' lengths (d_scan: scan_scratch_bytes(n))
int launch_permute_lists(uint64_t *d_offsets, const uint64_t *d_src, const uint32_t *d_tmp, uint32_t *d_hits, size_t n,
                         uint64_t cap, bool sort_ids, bool *sorted, void *d_scan, hipStream_t s);
int launch_query_pipe(const IndexView &v, const uint32_t *d_qchrom, const uint32_t *d_qlow, const uint32_t *d_qhigh,
                      size_t q0, size_t q1, uint64_t *d_offsets, uint32_t *d_hits, uint64_t cap, uint64_t *ws,
                      int flags, uint32_t sort_seq, uint32_t *d_counts, uint64_t *d_total, hipStream_t s);

}  // namespace bivx
