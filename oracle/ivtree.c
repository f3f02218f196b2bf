/*
 * oracle/ivtree.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see ivtree.h for the usage rule and the
 * parity-pin status). CPU restatement of ylab-hi/BINARY's red-black interval tree; every function
 * cites the reference lines it follows (paths under library/include/binary/algorithm/).
 */
#include "ivtree.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define NIL (-1)

typedef struct {
  uint32_t low, high; /* interval (BaseInterval, interval_tree.hpp:130-131) */
  uint32_t max;       /* max high in subtree (IntervalNode::max, :97) */
  int32_t left, right, parent;
  uint8_t red; /* Color (rb_tree.hpp:39); default-constructed nodes are Black (:70) */
} node_t;

struct ivt_tree {
  node_t *nd;
  size_t n, cap;
  int32_t root;
  /* the reference's nil_ sentinel (rb_tree.hpp:170): only its parent/colour are ever used, by delete */
  int32_t nil_parent;
  uint8_t nil_red;
  int ub; /* set when the reference would have dereferenced nullptr (delete of a black left leaf) */
};

ivt_tree *ivt_create(void) {
  ivt_tree *t = (ivt_tree *)calloc(1, sizeof(*t));
  if (t) {
    t->root = NIL;
    t->nil_parent = NIL;
  }
  return t;
}

void ivt_destroy(ivt_tree *t) {
  if (!t) return;
  free(t->nd);
  free(t);
}

static inline int is_red(const ivt_tree *t, int32_t n) { /* check_is_red rb_tree.hpp:190-193 */
  return n != NIL && t->nd[n].red;
}
static inline uint32_t get_max(const ivt_tree *t, int32_t n) { /* interval_tree.hpp:262-268 */
  return n == NIL ? 0u : t->nd[n].max;
}
static inline uint32_t max3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t m = a > b ? a : b;
  return m > c ? m : c;
}

/* RbTree::left_rotate rb_tree.hpp:255-279 */
static void base_left_rotate(ivt_tree *t, int32_t x) {
  node_t *nd = t->nd;
  int32_t y = nd[x].right;
  nd[x].right = nd[y].left;
  if (nd[x].right != NIL) nd[nd[x].right].parent = x;
  nd[y].parent = nd[x].parent;
  if (nd[x].parent == NIL)
    t->root = y;
  else if (x == nd[nd[x].parent].left)
    nd[nd[x].parent].left = y;
  else
    nd[nd[x].parent].right = y;
  nd[y].left = x;
  nd[x].parent = y;
}

/* RbTree::right_rotate rb_tree.hpp:281-302 */
static void base_right_rotate(ivt_tree *t, int32_t x) {
  node_t *nd = t->nd;
  int32_t y = nd[x].left;
  nd[x].left = nd[y].right;
  if (nd[x].left != NIL) nd[nd[x].left].parent = x;
  nd[y].parent = nd[x].parent;
  if (nd[x].parent == NIL)
    t->root = y;
  else if (x == nd[nd[x].parent].left)
    nd[nd[x].parent].left = y;
  else
    nd[nd[x].parent].right = y;
  nd[y].right = x;
  nd[x].parent = y;
}

/* IntervalTree::left_rotate / right_rotate interval_tree.hpp:206-228: base rotation, then
 * parent.max = max(parent.max, node.max[old]) (:212), node.max = max(high, L.max, R.max) (:213). */
static void iv_fix_after_rotate(ivt_tree *t, int32_t x) {
  node_t *nd = t->nd;
  int32_t p = nd[x].parent;
  if (nd[p].max < nd[x].max) nd[p].max = nd[x].max;
  nd[x].max = max3(nd[x].high, get_max(t, nd[x].left), get_max(t, nd[x].right));
}
static void left_rotate(ivt_tree *t, int32_t x) {
  base_left_rotate(t, x);
  iv_fix_after_rotate(t, x);
}
static void right_rotate(ivt_tree *t, int32_t x) {
  base_right_rotate(t, x);
  iv_fix_after_rotate(t, x);
}

/* RbTree::fix_insert rb_tree.hpp:304-344 */
static void fix_insert(ivt_tree *t, int32_t z) {
  node_t *nd = t->nd;
  while (is_red(t, nd[z].parent)) {
    int32_t p = nd[z].parent, g = nd[p].parent;
    if (p == nd[g].left) {
      int32_t y = nd[g].right;
      if (is_red(t, y)) { /* case 1 */
        nd[p].red = 0;
        nd[y].red = 0;
        nd[g].red = 1;
        z = g;
      } else {
        if (z == nd[p].right) { /* case 2 */
          z = p;
          left_rotate(t, z);
        }
        /* case 3 */
        nd[nd[z].parent].red = 0;
        nd[nd[nd[z].parent].parent].red = 1;
        right_rotate(t, nd[nd[z].parent].parent);
      }
    } else {
      int32_t y = nd[g].left;
      if (is_red(t, y)) {
        nd[p].red = 0;
        nd[y].red = 0;
        nd[g].red = 1;
        z = g;
      } else {
        if (z == nd[p].left) {
          z = p;
          right_rotate(t, z);
        }
        nd[nd[z].parent].red = 0;
        nd[nd[nd[z].parent].parent].red = 1;
        left_rotate(t, nd[nd[z].parent].parent);
      }
    }
  }
  nd[t->root].red = 0;
}

int32_t ivt_insert(ivt_tree *t, uint32_t low, uint32_t high) {
  if (t->n == t->cap) {
    size_t nc = t->cap ? t->cap * 2 : 1024;
    node_t *p = (node_t *)realloc(t->nd, nc * sizeof(node_t));
    if (!p) return NIL;
    t->nd = p;
    t->cap = nc;
  }
  int32_t z = (int32_t)t->n++;
  node_t *nd = t->nd;
  /* IntervalNode ctor interval_tree.hpp:66-69: max = high, key = low */
  nd[z].low = low;
  nd[z].high = high;
  nd[z].max = high;
  nd[z].left = nd[z].right = nd[z].parent = NIL;
  nd[z].red = 0;

  /* IntervalTree::insert_node_impl interval_tree.hpp:230-260 */
  int32_t x = t->root, y = NIL;
  while (x != NIL) {
    y = x;
    if (nd[x].max < nd[z].max) nd[x].max = nd[z].max; /* :240 */
    x = (low < nd[x].low) ? nd[x].left : nd[x].right; /* ties go right :242-246 */
  }
  nd[z].parent = y;
  if (y == NIL)
    t->root = z;
  else if (low < nd[y].low)
    nd[y].left = z;
  else
    nd[y].right = z;
  nd[z].red = 1;
  fix_insert(t, z);
  return z;
}

void ivt_insert_many(ivt_tree *t, const uint32_t *low, const uint32_t *high, size_t n) {
  for (size_t i = 0; i < n; ++i) ivt_insert(t, low[i], high[i]);
}

size_t ivt_size(const ivt_tree *t) { /* rb_tree.hpp:173-180, iteratively */
  size_t cnt = 0;
  int32_t stack[160];
  int sp = 0;
  if (t->root != NIL) stack[sp++] = t->root;
  while (sp) {
    int32_t n = stack[--sp];
    ++cnt;
    if (t->nd[n].right != NIL) stack[sp++] = t->nd[n].right;
    if (t->nd[n].left != NIL) stack[sp++] = t->nd[n].left;
  }
  return cnt;
}

int32_t ivt_root(const ivt_tree *t) { return t->root; }
uint32_t ivt_low(const ivt_tree *t, int32_t n) { return t->nd[n].low; }
uint32_t ivt_high(const ivt_tree *t, int32_t n) { return t->nd[n].high; }
uint32_t ivt_max(const ivt_tree *t, int32_t n) { return t->nd[n].max; }
int32_t ivt_left(const ivt_tree *t, int32_t n) { return t->nd[n].left; }
int32_t ivt_right(const ivt_tree *t, int32_t n) { return t->nd[n].right; }
int32_t ivt_parent(const ivt_tree *t, int32_t n) { return t->nd[n].parent; }
int ivt_is_red(const ivt_tree *t, int32_t n) { return t->nd[n].red; }

int32_t ivt_minimum(const ivt_tree *t, int32_t n) { /* rb_tree.hpp:206-212 */
  while (t->nd[n].left != NIL) n = t->nd[n].left;
  return n;
}
int32_t ivt_maximum(const ivt_tree *t, int32_t n) { /* rb_tree.hpp:214-220 */
  while (t->nd[n].right != NIL) n = t->nd[n].right;
  return n;
}
int32_t ivt_successor(const ivt_tree *t, int32_t n) { /* rb_tree.hpp:222-237 */
  if (t->nd[n].right != NIL) return ivt_minimum(t, t->nd[n].right);
  int32_t p = t->nd[n].parent;
  while (p != NIL && t->nd[p].right == n) {
    n = p;
    p = t->nd[p].parent;
  }
  return p;
}
int32_t ivt_predecessor(const ivt_tree *t, int32_t n) { /* rb_tree.hpp:239-253 */
  if (t->nd[n].left != NIL) return ivt_maximum(t, t->nd[n].left);
  int32_t p = t->nd[n].parent;
  while (p != NIL && t->nd[p].left == n) {
    n = p;
    p = t->nd[p].parent;
  }
  return p;
}
int32_t ivt_search(const ivt_tree *t, uint32_t key) { /* rb_tree.hpp:559-589 */
  int32_t n = t->root;
  while (n != NIL) {
    if (key == t->nd[n].low) return n;
    n = key < t->nd[n].low ? t->nd[n].left : t->nd[n].right;
  }
  return NIL;
}

int ivt_black_height(const ivt_tree *t, int32_t n) { /* test_interval_tree.cpp:18-29 */
  if (n == NIL) return 0;
  int l = ivt_black_height(t, t->nd[n].left);
  int r = ivt_black_height(t, t->nd[n].right);
  if (l < 0 || r < 0 || l != r) return -1;
  return l + (t->nd[n].red ? 0 : 1);
}

static int check_max_rec(const ivt_tree *t, int32_t n, uint32_t *out) {
  if (n == NIL) {
    *out = 0;
    return 1;
  }
  uint32_t l, r;
  if (!check_max_rec(t, t->nd[n].left, &l)) return 0;
  if (!check_max_rec(t, t->nd[n].right, &r)) return 0;
  *out = max3(t->nd[n].high, l, r);
  return *out == t->nd[n].max;
}
int ivt_check_max(const ivt_tree *t) {
  uint32_t m;
  return check_max_rec(t, t->root, &m);
}

size_t ivt_preorder(const ivt_tree *t, int32_t *out, size_t cap) {
  size_t cnt = 0;
  int32_t stack[160];
  int sp = 0;
  if (t->root != NIL) stack[sp++] = t->root;
  while (sp) {
    int32_t n = stack[--sp];
    if (cnt < cap) out[cnt] = n;
    ++cnt;
    if (t->nd[n].right != NIL) stack[sp++] = t->nd[n].right;
    if (t->nd[n].left != NIL) stack[sp++] = t->nd[n].left;
  }
  return cnt;
}

/* BaseInterval::is_overlap interval_tree.hpp:119-121: closed intervals. */
static inline int overlap(uint32_t qlow, uint32_t qhigh, uint32_t low, uint32_t high) {
  return qlow <= high && low <= qhigh;
}

int32_t ivt_find_overlap(const ivt_tree *t, uint32_t qlow, uint32_t qhigh) { /* interval_tree.hpp:290-304 */
  int32_t x = t->root;
  while (x != NIL) {
    const node_t *nd = &t->nd[x];
    if (overlap(qlow, qhigh, nd->low, nd->high)) return x;
    /* :297 — note get_max(nullptr) == 0, so a query with low == 0 walks into a null left child */
    x = (qlow <= get_max(t, nd->left)) ? nd->left : nd->right;
  }
  return NIL;
}

size_t ivt_find_overlaps(const ivt_tree *t, uint32_t qlow, uint32_t qhigh, int32_t *out,
                         size_t cap) { /* interval_tree.hpp:306-334 */
  size_t cnt = 0;
  int32_t stack[160];
  int sp = 0;
  if (t->root != NIL) stack[sp++] = t->root;
  while (sp) {
    int32_t n = stack[--sp];
    const node_t *nd = &t->nd[n];
    if (overlap(qlow, qhigh, nd->low, nd->high)) { /* :315-317 */
      if (cnt < cap) out[cnt] = n;
      ++cnt;
    }
    /* recursion order is left (:319-321) then right (:323-325): push right first. A prune that lets
     * the recursion enter a null child (get_max(nullptr) == 0 >= q.low == 0) is a no-op there. */
    if (nd->right != NIL && qhigh >= nd->low && qlow <= t->nd[nd->right].max) stack[sp++] = nd->right;
    if (nd->left != NIL && qlow <= t->nd[nd->left].max) stack[sp++] = nd->left;
  }
  return cnt;
}

typedef struct {
  const ivt_tree *t;
  const uint32_t *qlow, *qhigh;
  size_t begin, end;
  uint32_t *counts;
  const uint64_t *offsets;
  int32_t *hits;
  uint64_t total;
} batch_arg;

static void *batch_worker(void *p) {
  batch_arg *a = (batch_arg *)p;
  uint64_t total = 0;
  for (size_t i = a->begin; i < a->end; ++i) {
    size_t c;
    if (a->hits) {
      size_t cap = (size_t)(a->offsets[i + 1] - a->offsets[i]);
      c = ivt_find_overlaps(a->t, a->qlow[i], a->qhigh[i], a->hits + a->offsets[i], cap);
    } else {
      c = ivt_find_overlaps(a->t, a->qlow[i], a->qhigh[i], NULL, 0);
    }
    if (a->counts) a->counts[i] = (uint32_t)c;
    total += c;
  }
  a->total = total;
  return NULL;
}

uint64_t ivt_find_overlaps_batch(const ivt_tree *t, const uint32_t *qlow, const uint32_t *qhigh, size_t q,
                                 int nthreads, uint32_t *counts, const uint64_t *offsets, int32_t *hits) {
  if (nthreads < 1) nthreads = 1;
  if ((size_t)nthreads > q && q > 0) nthreads = (int)q;
  batch_arg *args = (batch_arg *)calloc((size_t)nthreads, sizeof(batch_arg));
  pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
  uint64_t total = 0;
  for (int k = 0; k < nthreads; ++k) {
    args[k].t = t;
    args[k].qlow = qlow;
    args[k].qhigh = qhigh;
    args[k].begin = q * (size_t)k / (size_t)nthreads;
    args[k].end = q * (size_t)(k + 1) / (size_t)nthreads;
    args[k].counts = counts;
    args[k].offsets = offsets;
    args[k].hits = hits;
    if (nthreads == 1)
      batch_worker(&args[k]);
    else
      pthread_create(&th[k], NULL, batch_worker, &args[k]);
  }
  for (int k = 0; k < nthreads; ++k) {
    if (nthreads > 1) pthread_join(th[k], NULL);
    total += args[k].total;
  }
  free(args);
  free(th);
  return total;
}

size_t ivt_brute_overlaps(const uint32_t *low, const uint32_t *high, size_t n, uint32_t qlow,
                          uint32_t qhigh, int32_t *out, size_t cap) {
  size_t cnt = 0;
  for (size_t i = 0; i < n; ++i)
    if (overlap(qlow, qhigh, low[i], high[i])) {
      if (cnt < cap) out[cnt] = (int32_t)i;
      ++cnt;
    }
  return cnt;
}

/* ---- delete: RbTree only (the reference never repairs `max` on delete) ------------------------- */

/* RbTree::transplant rb_tree.hpp:413-428: unlink `target` (its slot is simply abandoned in the arena),
 * put `source` in its place, return target's parent. */
static int32_t transplant(ivt_tree *t, int32_t target, int32_t source) {
  node_t *nd = t->nd;
  int32_t ret = nd[target].parent;
  if (source != NIL) nd[source].parent = ret;
  if (ret == NIL)
    t->root = source;
  else if (target == nd[ret].left)
    nd[ret].left = source;
  else
    nd[ret].right = source;
  return ret;
}

/* rb_tree.hpp:430-495. `x` may be the nil sentinel, encoded as NIL with t->nil_parent set. */
static void fix_delete(ivt_tree *t, int32_t x, int x_is_left) {
  node_t *nd = t->nd;
#define PARENT(v) ((v) == NIL ? t->nil_parent : nd[(v)].parent)
#define BLACK(v) (!is_red(t, (v)))
  while (x != t->root && BLACK(x) && !(x == NIL && t->nil_parent == NIL)) {
    int32_t p = PARENT(x);
    /* check_is_left_child_when_delete rb_tree.hpp:497-504 */
    int left_child = (x == NIL) ? x_is_left : (x == nd[p].left);
    if (left_child) {
      int32_t w = nd[p].right;
      if (w == NIL) { /* reference: w->is_red() on nullptr */
        t->ub = 1;
        return;
      }
      if (nd[w].red) { /* case 1 */
        nd[w].red = 0;
        nd[p].red = 1;
        base_left_rotate(t, p);
        w = nd[p].right;
        if (w == NIL) {
          t->ub = 1;
          return;
        }
      }
      if (BLACK(nd[w].left) && BLACK(nd[w].right)) { /* case 2 */
        nd[w].red = 1;
        x = p;
      } else {
        if (BLACK(nd[w].right)) { /* case 3 */
          nd[nd[w].left].red = 0;
          nd[w].red = 1;
          base_right_rotate(t, w);
          w = nd[p].right;
        }
        /* case 4 */
        nd[w].red = nd[p].red;
        nd[p].red = 0;
        nd[nd[w].right].red = 0;
        base_left_rotate(t, p);
        x = t->root;
      }
    } else {
      int32_t w = nd[p].left;
      if (w == NIL) {
        t->ub = 1;
        return;
      }
      if (nd[w].red) {
        nd[w].red = 0;
        nd[p].red = 1;
        base_right_rotate(t, p);
        w = nd[p].left;
        if (w == NIL) {
          t->ub = 1;
          return;
        }
      }
      if (BLACK(nd[w].left) && BLACK(nd[w].right)) {
        nd[w].red = 1;
        x = p;
      } else {
        if (BLACK(nd[w].left)) {
          nd[nd[w].right].red = 0;
          nd[w].red = 1;
          base_left_rotate(t, w);
          w = nd[p].left;
        }
        nd[w].red = nd[p].red;
        nd[p].red = 0;
        nd[nd[w].left].red = 0;
        base_right_rotate(t, p);
        x = t->root;
      }
    }
  }
  if (x != NIL) nd[x].red = 0; /* :494 (on the nil sentinel this only recolours the sentinel) */
#undef PARENT
#undef BLACK
}

void ivt_delete(ivt_tree *t, int32_t node) { /* rb_tree.hpp:506-553 */
  if (node == NIL) return;
  node_t *nd = t->nd;
  int32_t y = node, x = NIL, x_parent = NIL;
  int x_is_left = 0;
  int y_black = !nd[y].red;
  if (nd[node].left == NIL) {
    x = nd[node].right;
    x_parent = transplant(t, y, x);
  } else if (nd[node].right == NIL) {
    x = nd[node].left;
    x_is_left = 1;
    x_parent = transplant(t, y, x);
  } else {
    y = ivt_minimum(t, nd[node].right);
    y_black = !nd[y].red;
    x = nd[y].right;
    if (y != nd[node].right) {
      /* IntervalNode::copy_key interval_tree.hpp:77-81: key, max and interval move into `node` */
      nd[node].low = nd[y].low;
      nd[node].high = nd[y].high;
      nd[node].max = nd[y].max;
      x_parent = transplant(t, y, x);
      x_is_left = 1;
    } else {
      int32_t node_left = nd[node].left;
      nd[y].red = nd[node].red;
      transplant(t, node, y);
      nd[y].left = node_left;
      nd[node_left].parent = y;
      nd[y].right = x;
      x_parent = y;
    }
  }
  if (y_black && x_parent != NIL) {
    if (x == NIL) t->nil_parent = x_parent;
    fix_delete(t, x, x_is_left);
    t->nil_parent = NIL;
  }
}
