/* examples/rollout_host.c -- the C ABI of include/q2048.h driven from plain C: no Python, no
 * torch.  Build:  gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude
 *   examples/rollout_host.c -L2048_q-learning_amd/csrc -lq2048_hip -L/opt/rocm/lib -lamdhip64  It runs the fused loop of Agent/main.py:91-101 for B envs and prints the
 * statistics vector, which tests/test_gpu_parity.py compares with the Python host's.
 *
 *   usage: rollout_host <boards> <steps> <seed> <cap_log2> <eps> [steps per launch, default = steps]
 *                       [grow before launch k, default / -1 = never]
 *                       [close the key set before launch k, default = never]
 *
 * With more than one launch the rollouts go through q2048_fused_rollout_opts: the row every env
 * carries passes from launch to launch through a row cache (hipMalloc'd, zero-filled), and the
 * statistics are read from a host-side mirror (hipHostMalloc) that the last block of every launch
 * writes -- after the stream has been waited for, with no device-to-host copy.
 *
 * With the seventh argument the table is one that may GROW (q2048_table_reserve), like the reference's
 * defaultdict (Agent/main.py:16), off the caller's critical path: the fourfold table starts being mapped by the
 * library's host thread before the first launch (q2048_table_grow_begin), the move of the rows is queued on the
 * stream between launch k - 1 and launch k (q2048_table_grow_commit: returns the new table at once), and the
 * check and the hand-over of the old table happen after the last launch (q2048_table_grow_finish).  The
 * statistics it prints are those of the run on a fixed table.
 *
 * With the eighth argument the table CLOSES ITS KEY SET before launch k -- what a table that cannot grow any more
 * does (the reference's dict has no limit): from then on every launch carries Q2048_FLAG_NO_NEW_ROWS
 * (rows that exist keep learning, a state without a row reads as zeros and lives in the env's visit row -- the row
 * cache carries it from launch to launch --, its updates are dropped and counted), and, the table being 4x4, the
 * LINE SUMMARIES are written once (q2048_table_summarise) and Q2048_FLAG_LINE_SUMMARY makes a lookup of an absent
 * state one request.
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "q2048.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_Q(x) do { int e_ = (x); if (e_ != Q2048_OK) { \
  fprintf(stderr, "q2048 error: %s\n", q2048_strerror(e_)); return 3; } } while (0)

int main(int argc, char **argv) {
  const int64_t B = argc > 1 ? atoll(argv[1]) : 4096;
  const int64_t steps = argc > 2 ? atoll(argv[2]) : 64;
  const uint64_t seed = argc > 3 ? strtoull(argv[3], NULL, 10) : 0;
  const int cap_log2 = argc > 4 ? atoi(argv[4]) : 22;
  const double eps = argc > 5 ? atof(argv[5]) : 0.95;
  const int64_t per_launch = argc > 6 && atoll(argv[6]) > 0 ? atoll(argv[6]) : steps;
  const int64_t grow_at = argc > 7 ? atoll(argv[7]) : -1;
  const int64_t close_at = argc > 8 ? atoll(argv[8]) : -1;
  int cap_now = cap_log2;

  uint8_t *boards; q2048_aux *aux; q2048_slot *table; int64_t *stats_i; double *stats_f; uint32_t *status;
  CHECK_HIP(hipMalloc((void **)&boards, (size_t)B * 16));
  CHECK_HIP(hipMalloc((void **)&aux, (size_t)B * sizeof(q2048_aux)));
  /* the table: any zero-filled device memory will do (hipMalloc + hipMemset); the library's own
   * allocator maps it from 2 MiB physical chunks, which this memory system serves 15-20 % faster
   * under scattered writes (include/q2048.h, q2048_table_alloc) */
  if (grow_at >= 0) CHECK_Q(q2048_table_reserve(cap_log2, cap_log2 + 2, 0, &table));
  else CHECK_Q(q2048_table_alloc(cap_log2, 0, &table));
  CHECK_HIP(hipMalloc((void **)&stats_i, sizeof(int64_t) * Q2048_NSTAT_I));
  CHECK_HIP(hipMalloc((void **)&stats_f, sizeof(double) * Q2048_NSTAT_F));
  CHECK_HIP(hipMalloc((void **)&status, sizeof(uint32_t)));
  CHECK_HIP(hipMemset(stats_i, 0, sizeof(int64_t) * Q2048_NSTAT_I));
  CHECK_HIP(hipMemset(stats_f, 0, sizeof(double) * Q2048_NSTAT_F));
  CHECK_HIP(hipMemset(status, 0, sizeof(uint32_t)));

  CHECK_Q(q2048_env_init(boards, aux, B, 4, seed, 0, NULL));                    /* Game2048_env() x B */
  int64_t si[Q2048_NSTAT_I]; double sf[Q2048_NSTAT_F]; uint32_t st; int64_t rows = 0, *d_rows;
  if (per_launch >= steps) {
    CHECK_Q(q2048_fused_rollout(boards, aux, table, cap_log2, B, 4, steps, eps, 0.1, 0.99, seed, 0, 0, 0,
                                stats_i, stats_f, status, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(si, stats_i, sizeof si, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(sf, stats_f, sizeof sf, hipMemcpyDeviceToHost));
  } else {
    void *cache; uint32_t *ticket; uint64_t *mirror;
    const size_t rec = q2048_sizeof_rowcache(4);
    CHECK_HIP(hipMalloc(&cache, (size_t)B * rec));
    CHECK_HIP(hipMemset(cache, 0, (size_t)B * rec));                            /* zero = empty records */
    CHECK_HIP(hipMalloc((void **)&ticket, 2 * sizeof(uint32_t)));
    CHECK_HIP(hipMemset(ticket, 0, 2 * sizeof(uint32_t)));
    CHECK_HIP(hipHostMalloc((void **)&mirror, Q2048_MIRROR_WORDS * sizeof(uint64_t), hipHostMallocDefault));
    memset(mirror, 0, Q2048_MIRROR_WORDS * sizeof(uint64_t));
    q2048_rollout_opts opts;
    memset(&opts, 0, sizeof opts);
    opts.size = (uint32_t)sizeof opts;
    opts.row_cache = cache; opts.stats_mirror = mirror; opts.mirror_ticket = ticket;
    uint64_t launches = 0;
    uint32_t flags = 0;
    q2048_growth *growth = NULL;
    int64_t moved = -1;
    if (grow_at >= 0) CHECK_Q(q2048_table_grow_begin(table, cap_now, cap_now + 2, &growth));   /* returns at once */
    for (int64_t done = 0; done < steps; done += per_launch, ++launches) {
      const int64_t k = steps - done < per_launch ? steps - done : per_launch;
      if (growth != NULL && (int64_t)launches == grow_at) {
        /* the rows move between two launches, on the stream: nothing here waits for them.  Slots change with
         * the table, so the row cache is emptied (the contract; the records' table tag is only a safety net) */
        CHECK_Q(q2048_table_grow_commit(growth, 1, 0, &table, NULL));
        CHECK_HIP(hipMemsetAsync(cache, 0, (size_t)B * q2048_sizeof_rowcache(4), NULL));
        cap_now += 2;
      }
      if ((int64_t)launches == close_at) {
        /* the key set closes here: nothing creates a row from now on, so the summaries written now stay true */
        CHECK_Q(q2048_table_summarise(table, cap_now, NULL));
        flags = Q2048_FLAG_NO_NEW_ROWS | Q2048_FLAG_LINE_SUMMARY;
      }
      CHECK_Q(q2048_fused_rollout_opts(boards, aux, table, cap_now, B, 4, k, eps, 0.1, 0.99, seed, 0,
                                       (uint32_t)done, flags, stats_i, stats_f, status, &opts, NULL));
    }
    if (growth != NULL) {
      if (cap_now == cap_log2) CHECK_Q(q2048_table_grow_abort(growth));          /* never committed */
      else CHECK_Q(q2048_table_grow_finish(growth, &moved));                     /* waits for the move, checks it */
    }
    CHECK_HIP(hipDeviceSynchronize());                    /* the one wait; the statistics are already here */
    if (mirror[Q2048_MIRROR_SEQ] != launches) { fprintf(stderr, "stale statistics mirror\n"); return 4; }
    memcpy(si, mirror, sizeof si);
    memcpy(sf, mirror + Q2048_NSTAT_I, sizeof sf);
    CHECK_HIP(hipHostFree(mirror));
  }
  CHECK_HIP(hipMemcpy(&st, status, sizeof st, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMalloc((void **)&d_rows, sizeof(int64_t)));
  CHECK_HIP(hipMemset(d_rows, 0, sizeof(int64_t)));
  CHECK_Q(q2048_table_count(table, cap_now, d_rows, NULL));                     /* len(q_table) */
  CHECK_HIP(hipMemcpy(&rows, d_rows, sizeof rows, hipMemcpyDeviceToHost));
  uint8_t first[16];
  CHECK_HIP(hipMemcpy(first, boards, 16, hipMemcpyDeviceToHost));

  printf("{\"capacity_log2\": %d, ", cap_now);
  printf("\"steps\": %lld, \"episodes\": %lld, \"valid_moves\": %lld, \"score_sum\": %lld, "
         "\"inserts\": %lld, \"drops\": %lld, \"explored\": %lld, \"rows\": %lld, \"status\": %u, "
         "\"return_sum\": %.17g, \"board0\": [", (long long)si[Q2048_ST_STEPS],
         (long long)si[Q2048_ST_EPISODES], (long long)si[Q2048_ST_VALID], (long long)si[Q2048_ST_SCORE],
         (long long)si[Q2048_ST_INSERTS], (long long)si[Q2048_ST_DROPS], (long long)si[Q2048_ST_EXPLORE],
         (long long)rows, st, sf[Q2048_SF_RETURN]);
  for (int c = 0; c < 16; ++c) printf("%d%s", first[c], c < 15 ? ", " : "]}\n");
  CHECK_Q(q2048_table_free(table));
  return 0;
}
