/*
 * oracle/ivtree.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's interval-overlap algorithm: the augmented
 * red-black interval tree of ylab-hi/BINARY
 *   library/include/binary/algorithm/rb_tree.hpp        (RbTree: insert, fix-up, rotations, delete)
 *   library/include/binary/algorithm/interval_tree.hpp  (IntervalTree: max augmentation, find_overlap(s))
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this. The product
 * (binary_amd/, include/) never links, imports or calls it.
 *
 * Parity pin status: PINNED by the reference's own known-answer tests
 * (test/source/test_algorithm/test_interval_tree.cpp:87-155, test_rb_tree.cpp:128-283) and by the
 * reference outputs recorded in SURVEY.md §8c; see tests/test_oracle_pins.py. The reference headers
 * themselves are unbuildable in this image (they include spdlog, which is absent and may not be
 * stubbed), so there is no oracle/_ref build.
 *
 * Keys are uint32_t (the reference's UIntInterval, interval_tree.hpp:136, the type sv2nl uses). Signed
 * int32 keys (IntInterval, :135) are handled by callers through the order-preserving bias x ^ 0x80000000;
 * get_max(nullptr) = numeric_limits<key>::lowest() (interval_tree.hpp:263-267) is 0 in both pictures.
 *
 * Nodes live in one arena; node index == insertion index (0-based), -1 == nullptr.
 */
#ifndef ORACLE_IVTREE_H_
#define ORACLE_IVTREE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ivt_tree ivt_tree;

ivt_tree *ivt_create(void);
void ivt_destroy(ivt_tree *t);

/* RbTree::insert_node(Args&&...) rb_tree.hpp:145-149 -> IntervalTree::insert_node_impl
 * interval_tree.hpp:230-260 -> RbTree::fix_insert rb_tree.hpp:304-344. Returns the node index. */
int32_t ivt_insert(ivt_tree *t, uint32_t low, uint32_t high);
/* RbTree::insert_node(range) rb_tree.hpp:111-117: inserts in array order. */
void ivt_insert_many(ivt_tree *t, const uint32_t *low, const uint32_t *high, size_t n);

/* RbTree::size() rb_tree.hpp:173-180 (recursive count of reachable nodes), empty() :182-184. */
size_t ivt_size(const ivt_tree *t);
/* RbTree::root() rb_tree.hpp:186-188; -1 when empty. */
int32_t ivt_root(const ivt_tree *t);

/* node field accessors (IntervalNode members, interval_tree.hpp:96-102) */
uint32_t ivt_low(const ivt_tree *t, int32_t n);
uint32_t ivt_high(const ivt_tree *t, int32_t n);
uint32_t ivt_max(const ivt_tree *t, int32_t n);
int32_t ivt_left(const ivt_tree *t, int32_t n);
int32_t ivt_right(const ivt_tree *t, int32_t n);
int32_t ivt_parent(const ivt_tree *t, int32_t n);
int ivt_is_red(const ivt_tree *t, int32_t n);

/* minimum/maximum/successor/predecessor rb_tree.hpp:206-253; search :559-589 (by key == low). */
int32_t ivt_minimum(const ivt_tree *t, int32_t n);
int32_t ivt_maximum(const ivt_tree *t, int32_t n);
int32_t ivt_successor(const ivt_tree *t, int32_t n);
int32_t ivt_predecessor(const ivt_tree *t, int32_t n);
int32_t ivt_search(const ivt_tree *t, uint32_t key);

/* The black-height check the reference tests use (test_interval_tree.cpp:18-29): black height of the
 * subtree, or -1 if two sibling subtrees disagree. */
int ivt_black_height(const ivt_tree *t, int32_t n);
/* Checks the `max` augmentation of every node against a recomputation; 1 = consistent. */
int ivt_check_max(const ivt_tree *t);

/* Pre-order walk (node, left, right); writes node indices, returns how many. */
size_t ivt_preorder(const ivt_tree *t, int32_t *out, size_t cap);

/* IntervalTree::find_overlap interval_tree.hpp:290-304 — one descent, first overlapping node, with the
 * reference's null-left rule (goes left iff q.low <= get_max(left), and get_max(nullptr) == 0). -1 if none. */
int32_t ivt_find_overlap(const ivt_tree *t, uint32_t qlow, uint32_t qhigh);

/* IntervalTree::find_overlaps interval_tree.hpp:306-334 — all overlapping nodes in the reference's
 * pre-order with both prunes. Writes up to cap node indices, returns the total number of hits. */
size_t ivt_find_overlaps(const ivt_tree *t, uint32_t qlow, uint32_t qhigh, int32_t *out, size_t cap);

/* Batched find_overlaps over q queries on nthreads threads (const queries on a built tree are
 * thread-safe in the reference: mapper.cpp:130-141). counts[i] = hits of query i. If hits != NULL,
 * offsets (q+1 entries, exclusive prefix of counts) must have been filled by a previous call with
 * hits == NULL, and hits receives each query's node indices in pre-order at offsets[i]. Returns total. */
uint64_t ivt_find_overlaps_batch(const ivt_tree *t, const uint32_t *qlow, const uint32_t *qhigh, size_t q,
                                 int nthreads, uint32_t *counts, const uint64_t *offsets, int32_t *hits);

/* Brute force over all nodes with BaseInterval::is_overlap (interval_tree.hpp:119-121), ascending
 * insertion index; the tree-free definition of the hit set. */
size_t ivt_brute_overlaps(const uint32_t *low, const uint32_t *high, size_t n, uint32_t qlow,
                          uint32_t qhigh, int32_t *out, size_t cap);

/* RbTree::delete_node(raw_pointer) rb_tree.hpp:506-557 + fix_delete :430-495, as the base class does it
 * (keys only: the reference never defined IntervalTree::delete_node, interval_tree.hpp:28,148, so `max`
 * is NOT repaired). After a delete, node indices no longer equal insertion indices. */
void ivt_delete(ivt_tree *t, int32_t n);

#ifdef __cplusplus
}
#endif
#endif /* ORACLE_IVTREE_H_ */
