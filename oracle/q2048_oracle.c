/*
 * q2048_oracle.c -- CPU ORACLE (test infrastructure, see q2048_oracle.h).
 *
 * Restates, in plain C, the algorithm of the reference's tabular path.  Every function
 * names the reference lines it follows (paths relative to /root/reference/QLearningBase).
 * Pinned by the fixtures under tests/golden/, generated from the reference itself by
 * tests/golden/generate_golden.py ("parity pinned": tests/test_oracle_golden.py).
 */
#include "q2048_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ===================================================================================
 * Counter RNG: Philox4x32-10 (Salmon et al., SC'11; Random123 reference constants).
 * The reference has no counterpart: it draws from global MT19937 streams
 * (environment/Game2048_env.py:19-20, Agent/main.py:35-36).
 * =================================================================================== */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
    uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += PHILOX_W0;
    k1 += PHILOX_W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_draws(uint64_t seed, uint64_t env_id, uint32_t ctr, uint32_t stream, uint32_t out[4]) {
  uint32_t c[4] = {(uint32_t)env_id, (uint32_t)(env_id >> 32), ctr, stream};
  uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  orc_philox4x32_10(c, k, out);
}

/* Draw contract.  The reference decisions and the call each one replaces:
 *   random.random() < eps            (Agent/main.py:35)   uniform(x0) < eps
 *   random.randint(0, 3)             (Agent/main.py:36)   x1 >> 30
 *   np.random.randint(0, n_empty)    (Game2048_env.py:19) floor(x2 * n_empty / 2^32)
 *   np.random.random() < 0.9         (Game2048_env.py:20) uniform(x3) < 0.9          */
double orc_draw_uniform(uint32_t x) { return (double)x * (1.0 / 4294967296.0); }
int orc_draw_action(uint32_t x) { return (int)(x >> 30); }
int orc_draw_index(uint32_t x, int n) { return (int)(((uint64_t)x * (uint64_t)n) >> 32); }
int orc_draw_is_four(uint32_t x) { return !(orc_draw_uniform(x) < 0.9); }

/* ===================================================================================
 * Game core
 * =================================================================================== */

/* Game2048.move_left, one row (environment/Game2048_env.py:25-44): drop zeros, merge equal
 * neighbours once left-to-right with a skip flag, pad with zeros; moved if anything merged
 * or the row changed.  Tiles are log2, so "value*2" is +1 and the score term is 2^(v+1). */
int orc_move_left_line(uint8_t *line, int n, int64_t *score) {
  uint8_t nz[ORC_MAXN], merged[ORC_MAXN];
  int cnt = 0, m = 0, moved = 0, skip = 0;
  for (int i = 0; i < n; ++i)
    if (line[i] != 0) nz[cnt++] = line[i];            /* :26 */
  for (int i = 0; i < cnt; ++i) {                      /* :29 */
    if (skip) { skip = 0; continue; }                  /* :31-33 */
    if (i + 1 < cnt && nz[i] == nz[i + 1]) {           /* :34 */
      merged[m++] = (uint8_t)(nz[i] + 1);              /* :35 */
      *score += (int64_t)1 << (nz[i] + 1);             /* :36 */
      skip = 1;                                        /* :37 */
      moved = 1;                                       /* :38 */
    } else {
      merged[m++] = nz[i];                             /* :40 */
    }
  }
  while (m < n) merged[m++] = 0;                       /* :41 */
  if (memcmp(line, merged, (size_t)n) != 0) moved = 1; /* :42-43 */
  memcpy(line, merged, (size_t)n);                     /* :44 */
  return moved;
}

/* Game2048.rotate_board (:48-49): np.rot90, counter-clockwise: new[i][j] = old[j][n-1-i] */
void orc_rotate_ccw(uint8_t *board, int n) {
  uint8_t t[ORC_MAXCELLS];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) t[i * n + j] = board[j * n + (n - 1 - i)];
  memcpy(board, t, (size_t)(n * n));
}

/* Game2048.move without the spawn (:51-60): rotate `action` times, move_left every row,
 * rotate back (-action mod 4) times.  The boundary rejects actions outside 0..3 (the
 * reference would silently mis-rotate, :56-60), so the oracle does too. */
int orc_move(uint8_t *board, int n, int action, int64_t *score) {
  if (action < 0 || action > 3) return -1;
  int moved = 0;
  for (int r = 0; r < action; ++r) orc_rotate_ccw(board, n);          /* :56-57 */
  for (int i = 0; i < n; ++i)                                          /* :58 */
    if (orc_move_left_line(board + i * n, n, score)) moved = 1;
  for (int r = 0; r < ((4 - action) % 4); ++r) orc_rotate_ccw(board, n); /* :59-60 */
  return moved;
}

int orc_count_empty(const uint8_t *board, int n) {
  int c = 0;
  for (int i = 0; i < n * n; ++i) c += (board[i] == 0);
  return c;
}

/* Game2048.add_number (:16-20) with the two decisions given: k-th empty cell in row-major
 * order (np.where order, :17) receives 2 (log2 1) or 4 (log2 2). */
void orc_add_number_at(uint8_t *board, int n, int k, int is_four) {
  for (int i = 0; i < n * n; ++i) {
    if (board[i] == 0) {
      if (k == 0) { board[i] = is_four ? 2 : 1; return; }
      --k;
    }
  }
}

/* add_number from raw draws; returns 0 if the board is full (no-op, :18) */
int orc_add_number(uint8_t *board, int n, uint32_t draw_pos, uint32_t draw_val) {
  int e = orc_count_empty(board, n);
  if (e == 0) return 0;
  orc_add_number_at(board, n, orc_draw_index(draw_pos, e), orc_draw_is_four(draw_val));
  return 1;
}

/* Game2048.is_game_over (:65-75): not over while any cell is empty; otherwise over iff none
 * of the four trial moves changes the board.  (The reference's trial move also spawns and
 * is then undone, :70-73; that has no effect on the answer.) */
int orc_is_game_over(const uint8_t *board, int n) {
  for (int i = 0; i < n * n; ++i)
    if (board[i] == 0) return 0;                      /* :67-68 */
  for (int a = 0; a < 4; ++a) {                       /* :69 */
    uint8_t t[ORC_MAXCELLS];
    int64_t s = 0;
    memcpy(t, board, (size_t)(n * n));                /* :70 */
    if (orc_move(t, n, a, &s)) return 0;              /* :71-74 */
  }
  return 1;                                           /* :75 */
}

int orc_max_log2(const uint8_t *board, int n) {
  int m = 0;
  for (int i = 0; i < n * n; ++i)
    if (board[i] > m) m = board[i];
  return m;
}

/* ===================================================================================
 * Env
 * =================================================================================== */

/* Game2048_env.update_and_normalize (:197-205) */
double orc_update_and_normalize(double reward) {
  if (reward >= 0) return fmin(log2(reward + 1), 10);      /* :200-201 */
  return -fmin(log2(fabs(reward - 1)), 10);                /* :203 */
}

/* Game2048_env.calculate_reward (:136-184).  max_number is a power of two, so the reference
 * quantities are current_level = log2(max_number) = L and previous_max = 2^prev. */
double orc_calculate_reward(orc_env_t *e, int64_t score, int valid, int game_over,
                            int max_log2) {
  double reward = 0;
  int L = max_log2 < 1 ? 1 : max_log2;                     /* :141 max(2, max_number) */
  double current_level = (double)L;                        /* :144 */
  double bonus_progress = 0;                               /* :147 */
  if (L > e->previous_max_log2) {                          /* :148 */
    bonus_progress = (current_level - (double)e->previous_max_log2) *
                     pow(current_level, 1.2);              /* :149 */
    e->previous_max_log2 = L;                              /* :150 */
  }
  if (!valid) {                                            /* :152 */
    if (game_over) {                                       /* :154 */
      if (L == 9 || L == 10 || L == 11) {                  /* :156 in [512, 1024, 2048] */
        reward = bonus_progress + pow(current_level, 1.2); /* :158 */
      } else {
        reward -= log2((double)(((int64_t)1 << L) + 1));   /* :160 */
      }
    } else {
      reward -= 0.1 * current_level;                       /* :164 */
    }
  } else {
    reward = (double)score;                                /* :168 */
    if (bonus_progress > 0) reward += bonus_progress;      /* :171-172 */
    else if (bonus_progress == 0) reward += current_level * 0.05; /* :173-174 */
    if (L >= 9) reward += pow(current_level, 1.2) * 2;     /* :176-177 */
  }
  return orc_update_and_normalize(reward);                 /* :181 */
}

static void orc_new_game(orc_env_t *e, const uint32_t draws[4]) {
  /* Game2048.__init__ (:11-14): empty board, two spawns */
  memset(e->board, 0, sizeof e->board);
  orc_add_number(e->board, e->n, draws[0], draws[1]);
  orc_add_number(e->board, e->n, draws[2], draws[3]);
}

/* Game2048_env.__init__ (:81-95) */
void orc_env_init(orc_env_t *e, int n, const uint32_t draws[4]) {
  memset(e, 0, sizeof *e);
  e->n = n;
  orc_new_game(e, draws);                  /* :83 */
  e->score = 0;                            /* :84 */
  e->move_score = 0;                       /* :85 */
  e->previous_max_log2 = 1;                /* :87 previous_max = 2 */
  e->consecutive_action = ORC_NO_ACTION;   /* :92 */
  e->consecutive_count = 0;                /* :93 */
  e->last_consecutive_penalty = -1;        /* :95 */
  e->episode_return = 0;
  e->episode = 0;
}

/* Game2048_env.reset (:187-191): new game, score = 0, nothing else */
void orc_env_reset(orc_env_t *e, const uint32_t draws[4]) {
  orc_new_game(e, draws);                  /* :189 */
  e->score = 0;                            /* :190 */
  e->episode_return = 0;                   /* Agent/main.py:84 total_reward = 0 */
}

/* Game2048_env.step (:97-129) */
int orc_env_step(orc_env_t *e, int action, uint32_t draw_pos, uint32_t draw_val,
                 double *reward, int *done, int *max_log2) {
  int64_t score = 0;
  int valid = orc_move(e->board, e->n, action, &score);            /* :98 -> move :51-60 */
  if (valid < 0) return -1;
  if (valid) orc_add_number(e->board, e->n, draw_pos, draw_val);   /* :61-62 */
  int game_over = orc_is_game_over(e->board, e->n);                /* :99 */
  int mx = orc_max_log2(e->board, e->n);                           /* :100 */
  e->move_score = score;                                           /* :103 */
  e->score += score;                                               /* :104 */
  int d = 0;                                                       /* :105 */
  double r = orc_calculate_reward(e, score, valid, game_over, mx); /* :107 */
  if (action == e->consecutive_action) {                           /* :110 */
    e->consecutive_count += 1;                                     /* :111 */
  } else {
    e->consecutive_action = action;                                /* :113 */
    e->consecutive_count = 1;                                      /* :114 */
    e->last_consecutive_penalty = -1;                              /* :115 */
  }
  if (!valid && game_over) d = 1;                                  /* :117-118 */
  if (e->consecutive_count > 10) {                                 /* :121 */
    if (e->consecutive_count > 100) d = 1;                         /* :122-123 */
    double penalty = fmax(e->last_consecutive_penalty * 1.1, -10); /* :124 */
    e->last_consecutive_penalty = penalty;                         /* :125 */
    r += penalty;                                                  /* :127 */
  }
  *reward = r;
  *done = d;
  *max_log2 = mx;
  return valid;
}

/* --- second env profile: Deep_QLearning/environment/Game2048_nopenalty_env.py ------------
 * The DQN path's copy of the env.  Differences from Game2048_env.step, all restated here:
 *   - Game2048.move works on a COPY, `moved_board` (:58); `board` itself is only replaced by
 *     the caller (Deep_QLearning/main_dir/mainDQL_CNN_step2.py:237 `env.game.board =
 *     next_state`), which this restatement folds into the step: on return e->board holds
 *     moved_board.
 *   - is_game_over (:68-78) looks at `board`, i.e. the position BEFORE the move.  With an empty
 *     cell it is False (:70-71).  On a full board it calls move(action) -- not a trial -- for
 *     action 0..3 (:72-74); the first one that moves something leaves ITS result (with its own
 *     spawn, two further draws) in moved_board and returns False (:75-77): the board the step
 *     returns is then that move's, not the chosen action's.  If none moves, moved_board ends as
 *     a copy of the unchanged board and the game is over.
 *   - reward = calculate_reward2 (:122-138): -10 for an invalid move while the game is not over,
 *     otherwise the merge score of the CHOSEN action; no shaping, no normalisation, no stall
 *     rule.  done = game_over (:117-118).  max_number is taken from moved_board (:109).
 * over_pos / over_val are the two draws of the spawn inside is_game_over's move. */
int orc_env_step_dqn(orc_env_t *e, int action, uint32_t draw_pos, uint32_t draw_val,
                     uint32_t over_pos, uint32_t over_val, double *reward, int *done,
                     int *max_log2) {
  const int n = e->n;
  uint8_t moved[ORC_MAXCELLS];
  int64_t score = 0;
  memcpy(moved, e->board, ORC_MAXCELLS);                            /* :58 */
  int valid = orc_move(moved, n, action, &score);                   /* :107 -> :53-63 */
  if (valid < 0) return -1;
  if (valid) orc_add_number(moved, n, draw_pos, draw_val);          /* :64-65 */
  int game_over = 0;                                                /* :108 -> :68-78 */
  if (orc_count_empty(e->board, n) == 0) {                          /* :70 */
    game_over = 1;
    for (int a = 0; a < 4; ++a) {                                   /* :72 */
      int64_t s2 = 0;
      memcpy(moved, e->board, ORC_MAXCELLS);                        /* :58 again: moved_board is rebuilt */
      if (orc_move(moved, n, a, &s2) > 0) {                         /* :74 */
        orc_add_number(moved, n, over_pos, over_val);               /* :64-65 */
        game_over = 0;                                              /* :75-77 */
        break;
      }
    }
  }
  const int mx = orc_max_log2(moved, n);                            /* :109 */
  e->move_score = score;                                            /* :111 */
  e->score += score;                                                /* :112 */
  double r;
  if (!valid && !game_over) r = -10;                                /* :125-126 */
  else r = (double)score;                                           /* :127-128 */
  memcpy(e->board, moved, ORC_MAXCELLS);      /* mainDQL_CNN_step2.py:237 env.game.board = next_state */
  *reward = r;
  *done = game_over;                                                /* :117-118 */
  *max_log2 = mx;
  return valid;
}

/* SURVEY 7.8 opt-in (not the reference's behaviour): a reset that also restores the shaping
 * state Game2048_env.__init__ sets (:87, :92-95), which Game2048_env.reset (:187-191) leaves
 * untouched. */
void orc_env_reset_shaping(orc_env_t *e) {
  e->previous_max_log2 = 1;
  e->consecutive_action = ORC_NO_ACTION;
  e->consecutive_count = 0;
  e->last_consecutive_penalty = -1;
}

int orc_sizeof_env(void) { return (int)sizeof(orc_env_t); }

/* ===================================================================================
 * Agent: dict{state -> double[4]} (Agent/main.py:16) as an open-addressed map keyed by
 * the board bytes; grows by doubling, so it never drops (like the dict).
 * =================================================================================== */
typedef struct {
  uint8_t key[ORC_MAXCELLS];
  double q[4];
  uint8_t used;
} orc_row_t;

struct orc_qtable {
  orc_row_t *rows;
  int64_t cap, size;
};

static uint64_t orc_hash(const uint8_t *k) {
  uint64_t h = 1469598103934665603ull; /* FNV-1a over the padded 32-byte key */
  for (int i = 0; i < ORC_MAXCELLS; ++i) { h ^= k[i]; h *= 1099511628211ull; }
  h ^= h >> 29;
  return h;
}

static orc_qtable_t *orc_qtable_new(int64_t cap) {
  orc_qtable_t *t = (orc_qtable_t *)malloc(sizeof *t);
  t->cap = cap; t->size = 0;
  t->rows = (orc_row_t *)calloc((size_t)cap, sizeof(orc_row_t));
  return t;
}

static orc_row_t *orc_qtable_find(const orc_qtable_t *t, const uint8_t *key) {
  int64_t i = (int64_t)(orc_hash(key) & (uint64_t)(t->cap - 1));
  while (t->rows[i].used) {
    if (memcmp(t->rows[i].key, key, ORC_MAXCELLS) == 0) return &t->rows[i];
    i = (i + 1) & (t->cap - 1);
  }
  return NULL;
}

static void orc_qtable_grow(orc_qtable_t *t) {
  orc_row_t *old = t->rows;
  int64_t oc = t->cap;
  t->cap *= 2;
  t->rows = (orc_row_t *)calloc((size_t)t->cap, sizeof(orc_row_t));
  for (int64_t j = 0; j < oc; ++j) {
    if (!old[j].used) continue;
    int64_t i = (int64_t)(orc_hash(old[j].key) & (uint64_t)(t->cap - 1));
    while (t->rows[i].used) i = (i + 1) & (t->cap - 1);
    t->rows[i] = old[j];
  }
  free(old);
}

/* defaultdict lookup: inserts a zero row when absent (Agent/main.py:16) */
static orc_row_t *orc_qtable_get(orc_qtable_t *t, const uint8_t *key, int *inserted) {
  if (inserted) *inserted = 0;
  orc_row_t *r = orc_qtable_find(t, key);
  if (r) return r;
  if ((t->size + 1) * 2 > t->cap) orc_qtable_grow(t);
  int64_t i = (int64_t)(orc_hash(key) & (uint64_t)(t->cap - 1));
  while (t->rows[i].used) i = (i + 1) & (t->cap - 1);
  memcpy(t->rows[i].key, key, ORC_MAXCELLS);
  memset(t->rows[i].q, 0, sizeof t->rows[i].q);
  t->rows[i].used = 1;
  t->size += 1;
  if (inserted) *inserted = 1;
  return &t->rows[i];
}

static void orc_key_of(const orc_agent_t *a, const uint8_t *board, uint8_t key[ORC_MAXCELLS]) {
  memset(key, 0, ORC_MAXCELLS);
  memcpy(key, board, (size_t)(a->n * a->n)); /* tuple(map(tuple, state)), Agent/main.py:82 */
}

/* QLearningAgent.__init__ (Agent/main.py:15-32) */
orc_agent_t *orc_agent_new(double total_epochs, int action_space, double lr, double gamma,
                           double eps, double eps_min, int n) {
  orc_agent_t *a = (orc_agent_t *)calloc(1, sizeof *a);
  a->q = orc_qtable_new(1024);
  a->lr = lr;                                              /* :17 */
  a->gamma = gamma;                                        /* :18 */
  a->epsilon = eps;                                        /* :19 */
  a->epsilon_min = eps_min;                                /* :20 */
  a->action_space = action_space;                          /* :21 */
  a->total_epochs = total_epochs;                          /* :22 */
  a->n = n;
  a->first_decay_limit = total_epochs * 0.30;              /* :25 */
  a->second_decay_limit = total_epochs * 0.60;             /* :26 */
  a->third_decay_limit = total_epochs * 0.80;              /* :27 */
  a->slow_decay_1 = (eps - (eps_min * 1.5)) / a->first_decay_limit;                /* :30 */
  a->fast_decay = ((eps - eps_min) - (eps_min * 1.5)) /
                  (a->second_decay_limit - a->first_decay_limit);                  /* :31 */
  a->slow_decay_2 = (eps_min * 1.1 - eps_min) /
                    (a->third_decay_limit - a->second_decay_limit);                /* :32 */
  return a;
}

/* pre-size the dict (timing runs only: avoids rehash + first-touch page faults in the loop) */
void orc_agent_reserve(orc_agent_t *a, int64_t rows) {
  while (a->q->cap < rows * 2) orc_qtable_grow(a->q);
  for (int64_t i = 0; i < a->q->cap; i += 64) ((volatile uint8_t *)&a->q->rows[i])[0] = 0;
}

void orc_agent_free(orc_agent_t *a) {
  if (!a) return;
  free(a->q->rows);
  free(a->q);
  free(a);
}

static int orc_argmax4(const double *q, int n) { /* np.argmax: first maximum */
  int b = 0;
  for (int i = 1; i < n; ++i)
    if (q[i] > q[b]) b = i;
  return b;
}

/* q_table[state]: the defaultdict's lookup (creates the row, Agent/main.py:16) -- or, with the closed key
 * set (orc_agent_t.frozen; not in the reference), the row if it exists and else a zero row that is nobody's */
static const double *orc_row_of(orc_agent_t *a, const uint8_t *key, const orc_visit_t *visit) {
  static const double zero_row[4] = {0.0, 0.0, 0.0, 0.0};
  if (!a->frozen) return orc_qtable_get(a->q, key, NULL)->q;
  const orc_row_t *r = orc_qtable_find(a->q, key);
  if (r) return r->q;
  if (visit && visit->valid && memcmp(visit->key, key, ORC_MAXCELLS) == 0) return visit->q;   /* the env's visit row */
  return zero_row;
}

/* QLearningAgent.choose_action (:34-38); `visit`: the env's visit row (closed key set) or NULL */
static int orc_choose(orc_agent_t *a, const uint8_t *board, uint32_t draw_eps, uint32_t draw_act,
                      int *explored, const orc_visit_t *visit) {
  if (orc_draw_uniform(draw_eps) < a->epsilon) {           /* :35 */
    if (explored) *explored = 1;
    return orc_draw_action(draw_act);                      /* :36 */
  }
  if (explored) *explored = 0;
  uint8_t key[ORC_MAXCELLS];
  orc_key_of(a, board, key);
  return orc_argmax4(orc_row_of(a, key, visit), a->action_space);          /* :38 */
}
int orc_agent_choose(orc_agent_t *a, const uint8_t *board, uint32_t draw_eps,
                     uint32_t draw_act, int *explored) {
  return orc_choose(a, board, draw_eps, draw_act, explored, NULL);
}

/* QLearningAgent.update_q_value (:40-43); `visit`: the env's visit row (closed key set) or NULL */
static void orc_update(orc_agent_t *a, const uint8_t *s, int action, double reward, const uint8_t *s2,
                       int done, orc_visit_t *visit) {
  uint8_t k1[ORC_MAXCELLS], k2[ORC_MAXCELLS];
  orc_key_of(a, s, k1);
  orc_key_of(a, s2, k2);
  const double *rn = orc_row_of(a, k2, visit);
  int best_next = orc_argmax4(rn, a->action_space);                        /* :41 */
  double qn = rn[best_next];
  double target = reward + (a->gamma * qn * (double)(1 - (done ? 1 : 0))); /* :42 */
  if (a->frozen && !orc_qtable_find(a->q, k1)) {   /* closed key set: no row -- the table is not touched, the update counts as dropped */
    a->drops += 1;
    if (visit) {                                   /* ... and lands in the env's visit row, which ends when the env moves on */
      if (!(visit->valid && memcmp(visit->key, k1, ORC_MAXCELLS) == 0)) {
        memset(visit, 0, sizeof *visit);
        memcpy(visit->key, k1, ORC_MAXCELLS);
        visit->valid = 1;
      }
      visit->q[action] += a->lr * (target - visit->q[action]);             /* :43, on the fresh row */
      if (a->storage_f32) visit->q[action] = (double)(float)visit->q[action];
      if (done || memcmp(k1, k2, ORC_MAXCELLS) != 0) visit->valid = 0;
    }
    return;
  }
  if (visit) visit->valid = 0;                     /* (a state with a row has no visit row) */
  orc_row_t *rs = orc_qtable_get(a->q, k1, NULL); /* may grow: rn is dead from here */
  rs->q[action] += a->lr * (target - rs->q[action]);                       /* :43 */
  if (a->storage_f32) rs->q[action] = (double)(float)rs->q[action];        /* float32 table (option) */
}
void orc_agent_update(orc_agent_t *a, const uint8_t *s, int action, double reward,
                      const uint8_t *s2, int done) {
  orc_update(a, s, action, reward, s2, done, NULL);
}

/* QLearningAgent.decay_exploration (:45-57) */
void orc_agent_decay(orc_agent_t *a, double current_epoch) {
  if (current_epoch < a->first_decay_limit)
    a->epsilon = fmax(a->epsilon_min * 1.5, a->epsilon - a->slow_decay_1); /* :48 */
  else if (current_epoch < a->second_decay_limit)
    a->epsilon = fmax(a->epsilon_min * 1.1, a->epsilon - a->fast_decay);   /* :51 */
  else if (current_epoch < a->third_decay_limit)
    a->epsilon = fmax(a->epsilon_min, a->epsilon - a->slow_decay_2);       /* :54 */
  else
    a->epsilon = a->epsilon_min;                                           /* :57 */
}

int orc_agent_q(const orc_agent_t *a, const uint8_t *board, double out[4]) {
  uint8_t key[ORC_MAXCELLS];
  orc_key_of(a, board, key);
  orc_row_t *r = orc_qtable_find(a->q, key);
  for (int i = 0; i < 4; ++i) out[i] = r ? r->q[i] : 0.0;
  return r != NULL;
}

int64_t orc_agent_size(const orc_agent_t *a) { return a->q->size; }
void orc_agent_set_storage_f32(orc_agent_t *a, int on) { a->storage_f32 = on ? 1 : 0; }
void orc_agent_set_frozen(orc_agent_t *a, int on) { a->frozen = on ? 1 : 0; }
int64_t orc_agent_drops(const orc_agent_t *a) { return a->drops; }

int64_t orc_agent_dump(const orc_agent_t *a, uint8_t *keys, double *vals, int64_t max_rows) {
  int64_t w = 0;
  for (int64_t i = 0; i < a->q->cap && w < max_rows; ++i) {
    if (!a->q->rows[i].used) continue;
    memcpy(keys + w * ORC_MAXCELLS, a->q->rows[i].key, ORC_MAXCELLS);
    memcpy(vals + w * 4, a->q->rows[i].q, 4 * sizeof(double));
    ++w;
  }
  return w;
}

/* ===================================================================================
 * Batched driver: Agent/main.py:80-109 for B envs, lane-sequential within a step.
 * =================================================================================== */
void orc_envs_init(orc_env_t *envs, int64_t B, int n, uint64_t seed, uint64_t env_id0) {
  for (int64_t i = 0; i < B; ++i) {
    uint32_t d[4];
    orc_draws(seed, env_id0 + (uint64_t)i, 0u, ORC_STREAM_RESET, d);
    orc_env_init(&envs[i], n, d);
  }
}

void orc_rollout(orc_env_t *envs, int64_t B, orc_agent_t *agent, int64_t steps,
                 uint64_t seed, uint64_t env_id0, uint32_t ctr0, const uint8_t *actions,
                 int64_t *stats_i, double *stats_f, uint8_t *out_actions,
                 double *out_reward, uint8_t *out_done) {
  orc_rollout_ex(envs, B, agent, steps, seed, env_id0, ctr0, actions, stats_i, stats_f,
                 out_actions, out_reward, out_done, 0);
}

void orc_rollout_ex(orc_env_t *envs, int64_t B, orc_agent_t *agent, int64_t steps,
                    uint64_t seed, uint64_t env_id0, uint32_t ctr0, const uint8_t *actions,
                    int64_t *stats_i, double *stats_f, uint8_t *out_actions,
                    double *out_reward, uint8_t *out_done, int env_flags) {
  if (env_flags & ORC_ENV_NEW_VISITS)
    for (int64_t i = 0; i < B; ++i) envs[i].visit.valid = 0;
  for (int64_t t = 0; t < steps; ++t) {
    for (int64_t i = 0; i < B; ++i) {
      orc_env_t *e = &envs[i];
      uint64_t id = env_id0 + (uint64_t)i;
      uint32_t x[4];
      orc_draws(seed, id, ctr0 + (uint32_t)t, ORC_STREAM_STEP, x);
      uint8_t s[ORC_MAXCELLS];
      memcpy(s, e->board, ORC_MAXCELLS);
      int explored = 0, a;
      if (agent) a = orc_choose(agent, s, x[0], x[1], &explored, &e->visit);   /* main.py:92 */
      else if (actions) a = actions[t * B + i];
      else { a = orc_draw_action(x[1]); explored = 1; }                   /* random play */
      double r; int done, mx;
      int64_t size0 = agent ? agent->q->size : 0, drops0 = agent ? agent->drops : 0;
      int valid;
      if (env_flags & ORC_ENV_DQN) {
        uint32_t y[4];
        orc_draws(seed, id, ctr0 + (uint32_t)t, ORC_STREAM_OVER, y);
        valid = orc_env_step_dqn(e, a, x[2], x[3], y[0], y[1], &r, &done, &mx);
      } else {
        valid = orc_env_step(e, a, x[2], x[3], &r, &done, &mx);            /* :93 */
      }
      if (agent && agent->storage_f32) r = (double)(float)r; /* the device hands rewards over as float32 */
      if (agent) orc_update(agent, s, a, r, e->board, done, &e->visit);    /* :99 */
      e->episode_return += r;                                              /* :101 */
      if (out_actions) out_actions[t * B + i] = (uint8_t)a;
      if (out_reward) out_reward[t * B + i] = r;
      if (out_done) out_done[t * B + i] = (uint8_t)done;
      if (stats_i) {
        stats_i[ORC_ST_STEPS] += 1;
        stats_i[ORC_ST_VALID] += (valid > 0);
        stats_i[ORC_ST_EXPLORE] += explored;
        if (agent) stats_i[ORC_ST_INSERTS] += agent->q->size - size0;
        if (agent) stats_i[ORC_ST_DROPS] += agent->drops - drops0;
      }
      if (stats_f) stats_f[ORC_SF_REWARD] += r;
      if (done) {
        if (stats_i) {
          stats_i[ORC_ST_EPISODES] += 1;
          stats_i[ORC_ST_SCORE] += e->score;
          stats_i[ORC_ST_HIST0 + (mx > 22 ? 22 : mx)] += 1;   /* slots 8..30; 31 is the device's CAS-fallback counter */
        }
        if (stats_f) {
          stats_f[ORC_SF_RETURN] += e->episode_return;
          stats_f[ORC_SF_RETURN_SQ] += e->episode_return * e->episode_return;
        }
        uint32_t d[4];
        e->episode += 1;
        orc_draws(seed, id, e->episode, ORC_STREAM_RESET, d);
        orc_env_reset(e, d);                                               /* :81 */
        if (env_flags & ORC_ENV_RESET_SHAPING) orc_env_reset_shaping(e);
      }
    }
  }
}

void orc_rollout_sync(orc_env_t *envs, int64_t B, orc_agent_t *agent, int64_t steps,
                      uint64_t seed, uint64_t env_id0, uint32_t ctr0, int64_t *stats_i,
                      double *stats_f) {
  uint8_t *s_all = (uint8_t *)malloc((size_t)B * ORC_MAXCELLS);
  int *act = (int *)malloc((size_t)B * sizeof(int));
  double *target = (double *)malloc((size_t)B * sizeof(double));
  for (int64_t t = 0; t < steps; ++t) {
    for (int64_t i = 0; i < B; ++i) {                      /* phase 1: no table writes */
      orc_env_t *e = &envs[i];
      const uint64_t id = env_id0 + (uint64_t)i;
      uint32_t x[4];
      orc_draws(seed, id, ctr0 + (uint32_t)t, ORC_STREAM_STEP, x);
      uint8_t *s = s_all + i * ORC_MAXCELLS;
      memcpy(s, e->board, ORC_MAXCELLS);
      int explored = 0;
      act[i] = orc_choose(agent, s, x[0], x[1], &explored, &e->visit);       /* main.py:92 */
      double r; int done, mx;
      const int valid = orc_env_step(e, act[i], x[2], x[3], &r, &done, &mx);   /* :93 */
      uint8_t k1[ORC_MAXCELLS], k2[ORC_MAXCELLS];
      orc_key_of(agent, s, k1);
      orc_key_of(agent, e->board, k2);
      const double *rn = orc_row_of(agent, k2, &e->visit);                     /* :41 */
      const double qn = rn[orc_argmax4(rn, agent->action_space)];
      const double rf = (double)(float)r;  /* the device hands rewards over as float32 */
      target[i] = rf + (agent->gamma * qn * (double)(1 - (done ? 1 : 0)));     /* :42 */
      /* closed key set: a state without a row learns in the env's visit row, as in orc_update -- private to the env,
       * so it needs no ordering and is applied here; phase 2 counts the update as dropped */
      if (agent->frozen && !orc_qtable_find(agent->q, k1)) {
        orc_visit_t *v = &e->visit;
        if (!(v->valid && memcmp(v->key, k1, ORC_MAXCELLS) == 0)) {
          memset(v, 0, sizeof *v);
          memcpy(v->key, k1, ORC_MAXCELLS);
          v->valid = 1;
        }
        v->q[act[i]] += agent->lr * (target[i] - v->q[act[i]]);
        if (agent->storage_f32) v->q[act[i]] = (double)(float)v->q[act[i]];
        if (done || memcmp(k1, k2, ORC_MAXCELLS) != 0) v->valid = 0;
      } else {
        e->visit.valid = 0;
      }
      e->episode_return += rf;
      if (stats_i) {
        stats_i[ORC_ST_STEPS] += 1; stats_i[ORC_ST_VALID] += (valid > 0);
        stats_i[ORC_ST_EXPLORE] += explored;
      }
      if (stats_f) stats_f[ORC_SF_REWARD] += rf;
      if (done) {
        if (stats_i) {
          stats_i[ORC_ST_EPISODES] += 1; stats_i[ORC_ST_SCORE] += e->score;
          stats_i[ORC_ST_HIST0 + (mx > 22 ? 22 : mx)] += 1;   /* slots 8..30; 31 is the device's CAS-fallback counter */
        }
        if (stats_f) {
          stats_f[ORC_SF_RETURN] += e->episode_return;
          stats_f[ORC_SF_RETURN_SQ] += e->episode_return * e->episode_return;
        }
        uint32_t d[4];
        e->episode += 1;
        orc_draws(seed, id, e->episode, ORC_STREAM_RESET, d);
        orc_env_reset(e, d);
      }
    }
    for (int64_t i = 0; i < B; ++i) {                      /* phase 2: updates in env order */
      uint8_t k1[ORC_MAXCELLS];
      orc_key_of(agent, s_all + i * ORC_MAXCELLS, k1);
      if (agent->frozen && !orc_qtable_find(agent->q, k1)) {                  /* closed key set: dropped */
        agent->drops += 1;
        if (stats_i) stats_i[ORC_ST_DROPS] += 1;
        continue;
      }
      orc_row_t *rs = orc_qtable_get(agent->q, k1, NULL);
      rs->q[act[i]] += agent->lr * (target[i] - rs->q[act[i]]);               /* :43 */
    }
    if (agent->storage_f32)                                /* one rounding per touched entry and step */
      for (int64_t i = 0; i < B; ++i) {
        uint8_t k1[ORC_MAXCELLS];
        orc_key_of(agent, s_all + i * ORC_MAXCELLS, k1);
        orc_row_t *rs = orc_qtable_find(agent->q, k1);
        if (rs) rs->q[act[i]] = (double)(float)rs->q[act[i]];
      }
  }
  free(s_all); free(act); free(target);
}

typedef struct {
  char pad0[128]; /* a thread bumps si / sf every step: keep neighbours' counters off its cache lines */
  orc_env_t *envs; int64_t B; orc_agent_t *agent; int64_t steps;
  uint64_t seed, env_id0; uint32_t ctr0;
  int64_t si[ORC_ST_NI]; double sf[ORC_SF_NF];
  char pad1[128];
} orc_job_t;

static void *orc_job_run(void *p) {
  orc_job_t *j = (orc_job_t *)p;
  orc_rollout(j->envs, j->B, j->agent, j->steps, j->seed, j->env_id0, j->ctr0, NULL, j->si,
              j->sf, NULL, NULL, NULL);
  return NULL;
}

void orc_rollout_mt(orc_env_t *envs, int64_t B, orc_agent_t **agents, int T, int64_t steps,
                    uint64_t seed, uint64_t env_id0, uint32_t ctr0, int64_t *stats_i,
                    double *stats_f) {
  orc_job_t *jobs = (orc_job_t *)calloc((size_t)T, sizeof *jobs);
  pthread_t *th = (pthread_t *)calloc((size_t)T, sizeof *th);
  for (int k = 0; k < T; ++k) {
    int64_t lo = B * k / T, hi = B * (k + 1) / T;
    jobs[k].envs = envs + lo; jobs[k].B = hi - lo; jobs[k].agent = agents ? agents[k] : NULL;
    jobs[k].steps = steps; jobs[k].seed = seed; jobs[k].env_id0 = env_id0 + (uint64_t)lo;
    jobs[k].ctr0 = ctr0;
    pthread_create(&th[k], NULL, orc_job_run, &jobs[k]);
  }
  for (int k = 0; k < T; ++k) {
    pthread_join(th[k], NULL);
    if (stats_i) for (int q = 0; q < ORC_ST_NI; ++q) stats_i[q] += jobs[k].si[q];
    if (stats_f) for (int q = 0; q < ORC_SF_NF; ++q) stats_f[q] += jobs[k].sf[q];
  }
  free(jobs);
  free(th);
}

/* ===================================================================================
 * Row-tuple linear Q (BASELINE configs[1]) -- the build's own learner, see the header.
 * =================================================================================== */
#define ORC_RT_ROWS 4
#define ORC_RT_IDX 65536

float *orc_rt_new(void) { return (float *)calloc((size_t)ORC_RT_ROWS * ORC_RT_IDX * 4, sizeof(float)); }
void orc_rt_free(float *w) { free(w); }

static uint32_t orc_rt_index(const uint8_t *board, int r) {
  const uint8_t *p = board + 4 * r;
  return (uint32_t)(p[0] & 15) | ((uint32_t)(p[1] & 15) << 4) | ((uint32_t)(p[2] & 15) << 8) |
         ((uint32_t)(p[3] & 15) << 12);
}
static float *orc_rt_entry(const float *w, int r, uint32_t idx) {
  return (float *)w + (((size_t)r * ORC_RT_IDX + idx) << 2);
}

void orc_rt_q(const float *w, const uint8_t *board, float out[4]) {
  const float *e0 = orc_rt_entry(w, 0, orc_rt_index(board, 0));
  const float *e1 = orc_rt_entry(w, 1, orc_rt_index(board, 1));
  const float *e2 = orc_rt_entry(w, 2, orc_rt_index(board, 2));
  const float *e3 = orc_rt_entry(w, 3, orc_rt_index(board, 3));
  for (int a = 0; a < 4; ++a) out[a] = (e0[a] + e1[a]) + (e2[a] + e3[a]);
}

int orc_rt_choose(const float *w, const uint8_t *board, double eps, uint32_t draw_eps,
                  uint32_t draw_act, int *explored) {
  if (orc_draw_uniform(draw_eps) < eps) {
    if (explored) *explored = 1;
    return orc_draw_action(draw_act);
  }
  if (explored) *explored = 0;
  float q[4];
  orc_rt_q(w, board, q);
  int b = 0;
  for (int a = 1; a < 4; ++a)
    if (q[a] > q[b]) b = a;
  return b;
}

void orc_rt_update(float *w, const uint8_t *s, int action, float reward, const uint8_t *s2,
                   int done, double lr, double gamma) {
  float qn[4], qs[4];
  orc_rt_q(w, s2, qn);
  orc_rt_q(w, s, qs);
  float mx = qn[0] > qn[1] ? qn[0] : qn[1];
  float m2 = qn[2] > qn[3] ? qn[2] : qn[3];
  mx = mx > m2 ? mx : m2;
  const double target = (double)reward + (gamma * (double)mx * (done ? 0.0 : 1.0));
  const float d = (float)((lr * 0.25) * (target - (double)qs[action]));
  for (int r = 0; r < ORC_RT_ROWS; ++r) orc_rt_entry(w, r, orc_rt_index(s, r))[action] += d;
}

void orc_rt_rollout(orc_env_t *envs, int64_t B, float *w, int64_t steps, double eps, double lr,
                    double gamma, uint64_t seed, uint64_t env_id0, uint32_t ctr0,
                    int64_t *stats_i, double *stats_f) {
  for (int64_t t = 0; t < steps; ++t) {
    for (int64_t i = 0; i < B; ++i) {
      orc_env_t *e = &envs[i];
      const uint64_t id = env_id0 + (uint64_t)i;
      uint32_t x[4];
      orc_draws(seed, id, ctr0 + (uint32_t)t, ORC_STREAM_STEP, x);
      uint8_t s[ORC_MAXCELLS];
      memcpy(s, e->board, ORC_MAXCELLS);
      int explored = 0;
      const int a = orc_rt_choose(w, s, eps, x[0], x[1], &explored);
      double r; int done, mx;
      const int valid = orc_env_step(e, a, x[2], x[3], &r, &done, &mx);
      orc_rt_update(w, s, a, (float)r, e->board, done, lr, gamma);
      e->episode_return += (double)(float)r;
      if (stats_i) {
        stats_i[ORC_ST_STEPS] += 1;
        stats_i[ORC_ST_VALID] += (valid > 0);
        stats_i[ORC_ST_EXPLORE] += explored;
      }
      if (stats_f) stats_f[ORC_SF_REWARD] += (double)(float)r;
      if (done) {
        if (stats_i) {
          stats_i[ORC_ST_EPISODES] += 1;
          stats_i[ORC_ST_SCORE] += e->score;
          stats_i[ORC_ST_HIST0 + (mx > 22 ? 22 : mx)] += 1;   /* slots 8..30; 31 is the device's CAS-fallback counter */
        }
        if (stats_f) {
          stats_f[ORC_SF_RETURN] += e->episode_return;
          stats_f[ORC_SF_RETURN_SQ] += e->episode_return * e->episode_return;
        }
        uint32_t d[4];
        e->episode += 1;
        orc_draws(seed, id, e->episode, ORC_STREAM_RESET, d);
        orc_env_reset(e, d);
      }
    }
  }
}
