/* Sanitizer driver for the CPU twin (2048_q-learning_amd/csrc/q2048_host.cpp, the C ABI of include/q2048.h on host
 * memory): its threaded entry points on ONE shared table -- the Hogwild rollout (keys claimed by compare-and-swap,
 * Q values written with relaxed 4-byte atomics), the deterministic step, import / export -- with 4 threads
 * (Q2048_HOST_THREADS).  Built and run by tests/sanitize.sh with -fsanitize=thread (and =address,undefined); exits 0
 * when no row is lost (occupied slots == rows created), the deterministic step with 4 threads equals the one with 1
 * thread bit for bit, an export / import round trip returns every row, and a rollout with the key set closed
 * (Q2048_FLAG_NO_NEW_ROWS, visit rows through a row cache) creates none -- fused, then the deterministic step on the
 * same records (4 threads == 1 thread there too, row cache included). */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "q2048.h"

#define CHECK(x) do { int e_ = (x); if (e_ != 0) { fprintf(stderr, "%s -> %s\n", #x, q2048_strerror(e_)); return 2; } } while (0)

static void *zalloc(size_t bytes) {
  void *p = aligned_alloc(256, (bytes + 255) & ~(size_t)255);
  memset(p, 0, (bytes + 255) & ~(size_t)255);
  return p;
}

int main(void) {
  enum { B = 8192, CAP = 18, STEPS = 24 };
  int bad = 0;
  int64_t rows_fused = 0;
  for (int n = 4; n <= 5; ++n) {
    const size_t cells = (size_t)n * n;
    /* 1. the Hogwild rollout, 4 threads, one table */
    setenv("Q2048_HOST_THREADS", "4", 1);
    uint8_t *boards = zalloc(B * cells);
    q2048_aux *aux = zalloc(B * sizeof(q2048_aux));
    q2048_slot *table = zalloc(sizeof(q2048_slot) << CAP);
    int64_t si[Q2048_NSTAT_I] = {0}, count = 0;
    double sf[Q2048_NSTAT_F] = {0};
    uint32_t status = 0;
    CHECK(q2048_env_init(boards, aux, B, n, 3, 0, NULL));
    CHECK(q2048_fused_rollout(boards, aux, table, CAP, B, n, STEPS, 0.5, 0.1, 0.99, 3, 0, 0, 0, si, sf, &status, NULL));
    CHECK(q2048_table_count(table, CAP, &count, NULL));
    bad |= count != si[Q2048_ST_INSERTS] || si[Q2048_ST_DROPS] != 0 || si[Q2048_ST_STEPS] != (int64_t)B * STEPS;
    rows_fused += count;
    /* 2. export -> import into a second table, threaded: every row arrives */
    uint64_t *keys = zalloc((size_t)count * 16);
    float *q = zalloc((size_t)count * 16);
    int64_t got = 0, count2 = 0;
    CHECK(q2048_table_export(table, CAP, keys, q, count, n == 4 ? 1 : 2, &got, NULL));
    q2048_slot *table2 = zalloc(sizeof(q2048_slot) << (CAP + 1));
    CHECK(q2048_table_import(table2, CAP + 1, keys, q, got, n == 4 ? 1 : 2, &status, NULL));
    CHECK(q2048_table_count(table2, CAP + 1, &count2, NULL));
    bad |= got != count || count2 != count || (status & Q2048_STATUS_TABLE_FULL);
    /* 3. the deterministic step: 4 threads == 1 thread, bit for bit (boards, aux, the whole table as a set of rows) */
    uint8_t *b1 = zalloc(B * cells), *b4 = zalloc(B * cells), ws[512];
    q2048_aux *a1 = zalloc(B * sizeof(q2048_aux)), *a4 = zalloc(B * sizeof(q2048_aux));
    q2048_slot *t1 = zalloc(sizeof(q2048_slot) << CAP), *t4 = zalloc(sizeof(q2048_slot) << CAP);
    void *w = (void *)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    int64_t s1[Q2048_NSTAT_I] = {0}, s4[Q2048_NSTAT_I] = {0};
    CHECK(q2048_env_init(b1, a1, B, n, 5, 0, NULL));
    CHECK(q2048_env_init(b4, a4, B, n, 5, 0, NULL));
    setenv("Q2048_HOST_THREADS", "1", 1);
    CHECK(q2048_det_rollout(b1, a1, t1, CAP, B, n, STEPS, 0.3, 0.1, 0.99, 5, 0, 0, 0, s1, NULL, &status, w, 256, NULL));
    setenv("Q2048_HOST_THREADS", "4", 1);
    CHECK(q2048_det_rollout(b4, a4, t4, CAP, B, n, STEPS, 0.3, 0.1, 0.99, 5, 0, 0, 0, s4, NULL, &status, w, 256, NULL));
    bad |= memcmp(b1, b4, B * cells) != 0 || memcmp(a1, a4, B * sizeof(q2048_aux)) != 0 || memcmp(s1, s4, sizeof s1) != 0;
    /* (slots differ with the order in which racing threads claimed them: compare the rows through the table itself) */
    int64_t c1 = 0, c4 = 0, g1 = 0;
    CHECK(q2048_table_count(t1, CAP, &c1, NULL));
    CHECK(q2048_table_count(t4, CAP, &c4, NULL));
    uint64_t *k1 = zalloc((size_t)c1 * 16);
    float *q1 = zalloc((size_t)c1 * 16);
    CHECK(q2048_table_export(t1, CAP, k1, q1, c1, n == 4 ? 1 : 2, &g1, NULL));
    CHECK(q2048_table_import(t4, CAP, k1, q1, 0, n == 4 ? 1 : 2, &status, NULL));   /* (argument path with rows = 0) */
    bad |= c1 != c4;
    if (n == 4) {                                       /* every row of the 1-thread table, looked up in the 4-thread one */
      uint8_t *kb = zalloc((size_t)c1 * 16);
      float *qo = zalloc((size_t)c1 * 16);
      uint8_t *found = zalloc((size_t)c1);
      for (int64_t r = 0; r < c1; ++r)
        for (int c = 0; c < 16; ++c) kb[16 * r + c] = (uint8_t)((k1[r] >> (4 * c)) & 15u);
      CHECK(q2048_q_lookup(t4, CAP, kb, c1, 4, 0, 0, qo, found, &status, NULL));
      for (int64_t r = 0; r < c1; ++r) bad |= !found[r] || memcmp(qo + 4 * r, q1 + 4 * r, 16) != 0;
      free(kb); free(qo); free(found);
    }
    /* 4. the closed key set (Q2048_FLAG_NO_NEW_ROWS), 4 threads, the table of leg 1: visit rows travel through the row
     * cache across two launches; no row is created, every step either updates a row or counts a drop, no TABLE_FULL */
    {
      void *cache = zalloc((size_t)B * q2048_sizeof_rowcache(n));
      q2048_rollout_opts o;
      memset(&o, 0, sizeof o);
      o.size = (uint32_t)sizeof o;
      o.row_cache = cache;
      int64_t sz[Q2048_NSTAT_I] = {0}, after = 0;
      uint32_t st2 = 0;
      CHECK(q2048_fused_rollout_opts(boards, aux, table, CAP, B, n, STEPS / 2, 0.1, 0.1, 0.99, 3, 0, STEPS,
                                     Q2048_FLAG_NO_NEW_ROWS, sz, NULL, &st2, &o, NULL));
      if (n == 4) CHECK(q2048_table_summarise(table, CAP, NULL));   /* line summaries: written by 4 threads, then passed */
      CHECK(q2048_fused_rollout_opts(boards, aux, table, CAP, B, n, STEPS / 2, 0.1, 0.1, 0.99, 3, 0, STEPS + STEPS / 2,
                                     Q2048_FLAG_NO_NEW_ROWS | Q2048_FLAG_TD_CAS | (n == 4 ? Q2048_FLAG_LINE_SUMMARY : 0u),
                                     sz, NULL, &st2, &o, NULL));
      /* ... and on into the deterministic step, whose visit rows live in the same records: 4 threads == 1 thread */
      uint8_t *bd1 = zalloc(B * cells), *bd4 = zalloc(B * cells);
      q2048_aux *ad1 = zalloc(B * sizeof(q2048_aux)), *ad4 = zalloc(B * sizeof(q2048_aux));
      void *c1 = zalloc((size_t)B * q2048_sizeof_rowcache(n)), *c4 = zalloc((size_t)B * q2048_sizeof_rowcache(n));
      int64_t d1[Q2048_NSTAT_I] = {0}, d4[Q2048_NSTAT_I] = {0};
      memcpy(bd1, boards, B * cells); memcpy(bd4, boards, B * cells);
      memcpy(ad1, aux, B * sizeof(q2048_aux)); memcpy(ad4, aux, B * sizeof(q2048_aux));
      memcpy(c1, cache, (size_t)B * q2048_sizeof_rowcache(n)); memcpy(c4, cache, (size_t)B * q2048_sizeof_rowcache(n));
      /* (both runs on `table` itself, restored in between: a record carries a tag of the table's address) */
      q2048_slot *keep = zalloc(sizeof(q2048_slot) << CAP), *td1 = zalloc(sizeof(q2048_slot) << CAP);
      memcpy(keep, table, sizeof(q2048_slot) << CAP);
      setenv("Q2048_HOST_THREADS", "1", 1);
      CHECK(q2048_det_rollout_cached(bd1, ad1, table, CAP, B, n, 6, 0.05, 0.1, 0.99, 3, 0, 2 * STEPS, Q2048_FLAG_NO_NEW_ROWS,
                                     d1, NULL, &st2, w, 256, c1, NULL));
      memcpy(td1, table, sizeof(q2048_slot) << CAP);
      memcpy(table, keep, sizeof(q2048_slot) << CAP);
      setenv("Q2048_HOST_THREADS", "4", 1);
      CHECK(q2048_det_rollout_cached(bd4, ad4, table, CAP, B, n, 6, 0.05, 0.1, 0.99, 3, 0, 2 * STEPS, Q2048_FLAG_NO_NEW_ROWS,
                                     d4, NULL, &st2, w, 256, c4, NULL));
      bad |= memcmp(bd1, bd4, B * cells) != 0 || memcmp(ad1, ad4, B * sizeof(q2048_aux)) != 0 || memcmp(d1, d4, sizeof d1) != 0 ||
             memcmp(c1, c4, (size_t)B * q2048_sizeof_rowcache(n)) != 0 || memcmp(td1, table, sizeof(q2048_slot) << CAP) != 0 || d1[Q2048_ST_INSERTS] != 0 || d1[Q2048_ST_DROPS] <= 0;
      free(bd1); free(bd4); free(ad1); free(ad4); free(c1); free(c4); free(td1);
      memcpy(table, keep, sizeof(q2048_slot) << CAP);
      free(keep);
      CHECK(q2048_table_count(table, CAP, &after, NULL));
      bad |= after != count || sz[Q2048_ST_INSERTS] != 0 || sz[Q2048_ST_DROPS] <= 0 || sz[Q2048_ST_DROPS] > (int64_t)B * STEPS ||
             sz[Q2048_ST_STEPS] != (int64_t)B * STEPS || (st2 & Q2048_STATUS_TABLE_FULL);
      free(cache);
    }
    free(boards); free(aux); free(table); free(keys); free(q); free(table2);
    free(b1); free(b4); free(a1); free(a4); free(t1); free(t4); free(k1); free(q1);
  }
  printf("host twin driver: %d envs x %d steps, 4x4 and 5x5, 4 threads on one table, %lld rows: %s\n", B, STEPS,
         (long long)rows_fused, bad ? "MISMATCH" : "no row lost, deterministic step independent of the thread count");
  return bad;
}
