/* ThreadSanitizer driver for the oracle's only multi-threaded function, orc_rollout_mt (T pthreads
 * over contiguous env ranges, one private agent each): test infrastructure for the CPU baseline that
 * bench.py times.  Built and run by tests/sanitize.sh with -fsanitize=thread; exits 0 when the
 * threaded run equals the single-threaded one (same boards, same statistics, same table sizes). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/q2048_oracle.h"

int main(void) {
  enum { B = 512, T = 4, STEPS = 60 };
  orc_env_t *e1 = calloc(B, sizeof *e1), *e2 = calloc(B, sizeof *e2);
  orc_agent_t *a1[T], *a2[T];
  int64_t s1[ORC_ST_NI] = {0}, s2[ORC_ST_NI] = {0};
  double f1[ORC_SF_NF] = {0}, f2[ORC_SF_NF] = {0};
  orc_envs_init(e1, B, 4, 7, 100);
  orc_envs_init(e2, B, 4, 7, 100);
  for (int k = 0; k < T; ++k) {
    a1[k] = orc_agent_new(100, 4, 0.1, 0.99, 0.3, 0.01, 4);
    a2[k] = orc_agent_new(100, 4, 0.1, 0.99, 0.3, 0.01, 4);
  }
  orc_rollout_mt(e1, B, a1, T, STEPS, 7, 100, 0, s1, f1);           /* four threads */
  for (int k = 0; k < T; ++k) {                                      /* the same ranges, one after the other */
    const int64_t lo = (int64_t)B * k / T, hi = (int64_t)B * (k + 1) / T;
    orc_rollout(e2 + lo, hi - lo, a2[k], STEPS, 7, 100 + (uint64_t)lo, 0, NULL, s2, f2, NULL, NULL, NULL);
  }
  int bad = memcmp(e1, e2, B * sizeof *e1) != 0 || memcmp(s1, s2, sizeof s1) != 0;
  for (int k = 0; k < T; ++k) {
    bad |= orc_agent_size(a1[k]) != orc_agent_size(a2[k]);
    orc_agent_free(a1[k]);
    orc_agent_free(a2[k]);
  }
  printf("tsan driver: %d envs x %d steps on %d threads, %lld env-steps, %lld episodes: %s\n", B, STEPS, T,
         (long long)s1[ORC_ST_STEPS], (long long)s1[ORC_ST_EPISODES], bad ? "MISMATCH" : "threaded == sequential");
  free(e1);
  free(e2);
  return bad;
}
