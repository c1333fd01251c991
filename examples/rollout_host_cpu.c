/* examples/rollout_host_cpu.c -- the SAME C ABI (include/q2048.h) on host memory: this program is
 * examples/rollout_host.c with malloc in place of hipMalloc, linked against libq2048_host.so (the CPU twin,
 * 2048_q-learning_amd/csrc/q2048_host.cpp) instead of libq2048_hip.so.  No GPU, no HIP runtime, no Python.
 *   gcc -std=c11 -Iinclude examples/rollout_host_cpu.c -L2048_q-learning_amd/csrc -lq2048_host -o rollout_host_cpu
 *   usage: rollout_host_cpu <boards> <steps> <seed> <cap_log2> <eps> [steps per call, default = steps]
 * It runs the loop of Agent/main.py:91-101 for B envs and prints the statistics, which
 * tests/test_host_twin.py compares with the Python host's run on device "cpu" (Q2048_HOST_THREADS=1: the run is
 * then sequential and the comparison exact).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "q2048.h"

#define CHECK_Q(x) do { int e_ = (x); if (e_ != Q2048_OK) { \
  fprintf(stderr, "q2048 error: %s\n", q2048_strerror(e_)); return 3; } } while (0)

static void *zalloc(size_t bytes) {                       /* 16-byte alignment is all the ABI asks for */
  void *p = aligned_alloc(128, (bytes + 127) & ~(size_t)127);
  if (p != NULL) memset(p, 0, (bytes + 127) & ~(size_t)127);
  return p;
}

int main(int argc, char **argv) {
  const int64_t B = argc > 1 ? atoll(argv[1]) : 4096;
  const int64_t steps = argc > 2 ? atoll(argv[2]) : 64;
  const uint64_t seed = argc > 3 ? strtoull(argv[3], NULL, 10) : 0;
  const int cap_log2 = argc > 4 ? atoi(argv[4]) : 22;
  const double eps = argc > 5 ? atof(argv[5]) : 0.95;
  const int64_t per_call = argc > 6 && atoll(argv[6]) > 0 ? atoll(argv[6]) : steps;

  uint8_t *boards = zalloc((size_t)B * 16);
  q2048_aux *aux = zalloc((size_t)B * sizeof(q2048_aux));
  q2048_slot *table = zalloc(sizeof(q2048_slot) << cap_log2);       /* zero-filled = empty: a device table, byte for byte */
  int64_t si[Q2048_NSTAT_I] = {0}, rows = 0;
  double sf[Q2048_NSTAT_F] = {0};
  uint32_t status = 0;
  if (!boards || !aux || !table) return 2;

  CHECK_Q(q2048_env_init(boards, aux, B, 4, seed, 0, NULL));                    /* Game2048_env() x B */
  for (int64_t done = 0; done < steps; done += per_call) {
    const int64_t k = steps - done < per_call ? steps - done : per_call;
    CHECK_Q(q2048_fused_rollout(boards, aux, table, cap_log2, B, 4, k, eps, 0.1, 0.99, seed, 0, (uint32_t)done, 0,
                                si, sf, &status, NULL));                        /* complete when it returns */
  }
  CHECK_Q(q2048_table_count(table, cap_log2, &rows, NULL));                     /* len(q_table) */
  printf("{\"steps\": %lld, \"episodes\": %lld, \"valid_moves\": %lld, \"score_sum\": %lld, "
         "\"inserts\": %lld, \"drops\": %lld, \"explored\": %lld, \"rows\": %lld, \"status\": %u, "
         "\"return_sum\": %.17g, \"board0\": [", (long long)si[Q2048_ST_STEPS],
         (long long)si[Q2048_ST_EPISODES], (long long)si[Q2048_ST_VALID], (long long)si[Q2048_ST_SCORE],
         (long long)si[Q2048_ST_INSERTS], (long long)si[Q2048_ST_DROPS], (long long)si[Q2048_ST_EXPLORE],
         (long long)rows, status, sf[Q2048_SF_RETURN]);
  for (int c = 0; c < 16; ++c) printf("%d%s", boards[c], c < 15 ? ", " : "]}\n");
  free(boards); free(aux); free(table);
  return 0;
}
