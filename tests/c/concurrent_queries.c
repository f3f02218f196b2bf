/* Const queries on ONE index from several host threads, through the C ABI (include/bivx.h) — the reference's pool
 * threads sharing one tree (standalone/sv2nl/source/mapper.cpp:127-142: const find_overlaps on a shared_ptr tree).
 * Builds an index, answers NTHREADS x NCALLS single-query bivx_find_overlaps calls first on one thread, then with the
 * calls dealt to NTHREADS threads; every answer must be the serial one, and the threaded pass should take a fraction of
 * the serial time (every call runs on a lane of its own: stream, error block, workspace). Prints one JSON line.
 * usage: concurrent_queries [threads=8] [calls per thread=2000] */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "bivx.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(uint32_t n) {
  rng_state += 0x9E3779B97F4A7C15ull;
  uint64_t z = rng_state;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(((z >> 32) * n) >> 32);
}
static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

typedef struct {
  const bivx_index *idx;
  const uint32_t *qlo, *qhi;
  size_t first, count;
  uint64_t *n_hits;      /* per call */
  uint64_t *sum_ids;     /* per call: sum of the ids (ascending lists: the order is fixed) */
  int errors;
} task_t;

static void *run(void *arg) {
  task_t *t = (task_t *)arg;
  for (size_t k = t->first; k < t->first + t->count; ++k) {
    uint64_t off[2] = {0, 0};
    uint32_t *hits = NULL;
    if (bivx_find_overlaps(t->idx, NULL, t->qlo + k, t->qhi + k, 1, NULL, 1, off, &hits) != 0) {
      ++t->errors;
      continue;
    }
    uint64_t s = 0;
    for (uint64_t j = 0; j < off[1]; ++j) s += (uint64_t)hits[j] * (j + 1);
    t->n_hits[k] = off[1];
    t->sum_ids[k] = s;
    bivx_free(hits);
  }
  return NULL;
}

int main(int argc, char **argv) {
  const int nthreads = argc > 1 ? atoi(argv[1]) : 8;
  const size_t ncalls = argc > 2 ? (size_t)atol(argv[2]) : 2000;
  const size_t n = 200000, total = (size_t)nthreads * ncalls;
  bivx_index *idx = NULL;
  if (bivx_create(&idx, 0) != 0) {
    printf("no GPU: %s\n", bivx_last_error());
    return 3;
  }
  uint32_t *lo = malloc(n * 4), *hi = malloc(n * 4), *qlo = malloc(total * 4), *qhi = malloc(total * 4);
  uint64_t *h1 = calloc(total, 8), *s1 = calloc(total, 8), *h2 = calloc(total, 8), *s2 = calloc(total, 8);
  for (size_t i = 0; i < n; ++i) {
    lo[i] = rnd(50000000u);
    hi[i] = lo[i] + 1u + rnd(1000u);
  }
  for (size_t i = 0; i < total; ++i) {
    qlo[i] = rnd(50000000u);
    qhi[i] = qlo[i] + rnd(2000u);
  }
  if (bivx_append(idx, NULL, lo, hi, n) != 0 || bivx_build(idx) != 0) {
    printf("build failed: %s\n", bivx_last_error());
    return 1;
  }
  task_t warm = {idx, qlo, qhi, 0, 64, h1, s1, 0};
  run(&warm);
  task_t serial = {idx, qlo, qhi, 0, total, h1, s1, 0};
  double t0 = now_s();
  run(&serial);
  const double t_serial = now_s() - t0;
  pthread_t *th = malloc(sizeof(pthread_t) * (size_t)nthreads);
  task_t *tk = malloc(sizeof(task_t) * (size_t)nthreads);
  t0 = now_s();
  for (int t = 0; t < nthreads; ++t) {
    tk[t] = (task_t){idx, qlo, qhi, (size_t)t * ncalls, ncalls, h2, s2, 0};
    pthread_create(&th[t], NULL, run, &tk[t]);
  }
  int errors = serial.errors;
  for (int t = 0; t < nthreads; ++t) {
    pthread_join(th[t], NULL);
    errors += tk[t].errors;
  }
  const double t_threads = now_s() - t0;
  size_t differ = 0;
  uint64_t hits = 0;
  for (size_t k = 0; k < total; ++k) {
    differ += (h1[k] != h2[k] || s1[k] != s2[k]) ? 1u : 0u;
    hits += h1[k];
  }
  printf("{\"threads\": %d, \"calls\": %zu, \"hits\": %llu, \"serial_us_per_call\": %.2f, \"threaded_us_per_call\": %.2f, "
         "\"wall_ratio\": %.3f, \"errors\": %d, \"answers_that_differ\": %zu}\n",
         nthreads, total, (unsigned long long)hits, t_serial / (double)total * 1e6, t_threads / (double)total * 1e6,
         t_threads / t_serial, errors, differ);
  bivx_destroy(idx);
  return (errors || differ) ? 1 : 0;
}
