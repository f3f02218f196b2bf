/* tests/c/capi_example.c — the C ABI used from plain C11 (no C++, no Python): the reference's 10-interval fixture
 * (test/source/test_algorithm/test_interval_tree.cpp:88-137) through bivx_append / build / count / fill / any.
 * Compiled by the CPU suite (the header must be valid C), run by the GPU suite.
 *   exit code 0 = all checks passed; 3 = no usable GPU (bivx_create failed) */
#include <bivx.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(c)                                               \
  do {                                                         \
    if (!(c)) {                                                \
      printf("FAIL line %d: %s (%s)\n", __LINE__, #c, bivx_last_error()); \
      return 1;                                                \
    }                                                          \
  } while (0)

int main(void) {
  const uint32_t low[10] = {16, 8, 5, 0, 6, 15, 25, 17, 19, 26};
  const uint32_t high[10] = {21, 9, 8, 3, 10, 23, 30, 19, 20, 26};
  const uint32_t qlow[3] = {7, 15, 100}, qhigh[3] = {25, 25, 111};
  uint64_t offsets[4];
  uint32_t hits[16], first[3];
  bivx_index *ix = NULL;
  bivx_stats st;

  CHECK(bivx_abi_version() == BIVX_ABI_VERSION);
  if (bivx_create(&ix, 0) != BIVX_OK) {
    printf("no GPU: %s\n", bivx_last_error());
    return 3;
  }
  CHECK(bivx_count(ix, NULL, qlow, qhigh, 3, offsets) == BIVX_E_STATE); /* query before build */
  CHECK(bivx_append(ix, NULL, low, high, 10) == BIVX_OK);
  CHECK(bivx_size(ix) == 10);
  CHECK(bivx_build(ix) == BIVX_OK && bivx_is_built(ix));
  CHECK(bivx_count(ix, NULL, qlow, qhigh, 3, offsets) == BIVX_OK);
  CHECK(offsets[0] == 0 && offsets[1] == 8 && offsets[2] == 13 && offsets[3] == 13); /* 8, 5 and 0 hits */
  CHECK(bivx_fill(ix, NULL, qlow, qhigh, 3, offsets, hits, 1) == BIVX_OK);
  { /* (15,25) overlaps ids 0 [16,21], 5 [15,23], 6 [25,30], 7 [17,19], 8 [19,20] */
    const uint32_t want[5] = {0, 5, 6, 7, 8};
    for (int i = 0; i < 5; ++i) CHECK(hits[8 + i] == want[i]);
  }
  CHECK(bivx_any(ix, NULL, qlow, qhigh, 3, first) == BIVX_OK);
  CHECK(first[0] == 0 && first[1] == 0 && first[2] == BIVX_NO_HIT);
  CHECK(bivx_get_stats(ix, &st) == BIVX_OK && st.n_intervals == 10 && st.n_segments >= 1);
  CHECK(bivx_append(ix, NULL, low, high, 0) == BIVX_OK);
  CHECK(bivx_append(ix, NULL, NULL, high, 3) == BIVX_E_INVALID);
  bivx_destroy(ix);
  printf("capi_example: ok\n");
  return 0;
}
