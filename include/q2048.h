/*
 * q2048.h -- C ABI of libq2048_hip.so: the MI355X (gfx950) hot path of the batched 2048
 * tabular Q-learning loop.
 *
 * This is the drop-in boundary for the path of Rocco9999/2048_Q-Learning named by
 * BASELINE.json: QLearningBase/environment/Game2048_env.py (Game2048_env.step/reset) and
 * QLearningBase/Agent/main.py (QLearningAgent.choose_action/update_q_value).  The reference is
 * pure Python and has no FFI; these are the entry points a binding for that path needs
 * (INTEGRATION.md shows the ctypes stub).  Every function below names the reference
 * interface it replaces.
 *
 * Conventions
 *   - Plain pointers and sizes only.  Every data pointer is a DEVICE pointer (HIP); the caller
 *     owns all memory -- the library allocates nothing, with one exception the caller asks for by
 *     name: q2048_table_alloc / _reserve / _grow* / _trim / _free (a Q-table mapped from small physical
 *     chunks, which may grow; _alloc, _reserve, _grow and _free are host-synchronous, _grow_begin and
 *     _grow_commit return at once).  `stream` is a hipStream_t (NULL = default stream).  All other
 *     calls are asynchronous and stream-ordered: no host synchronisation inside.
 *   - The same ABI exists for HOST memory: libq2048_host.so (csrc/q2048_host.cpp), compiled from the same
 *     per-lane arithmetic and table layout -- every pointer is then a host pointer, `stream` is ignored and every
 *     call is complete when it returns; the chunk allocator returns Q2048_ERR_UNSUPPORTED there.  It is a
 *     device of its own ("cpu"), loaded only when asked for by name, never a fallback.
 *   - boards   uint8_t[B][16]: 4x4 board, row-major, log2 tiles (0 empty, k = tile 2^k), the
 *              device image of the reference's np.int64[4,4] raw-value board
 *              (Game2048_env.py:12).  16-byte aligned.
 *   - aux      q2048_aux[B]: per-env state that Game2048_env keeps in Python attributes
 *              (Game2048_env.py:81-95).  16-byte aligned.
 *   - table    q2048_slot[1 << cap_log2]: open-addressed hash Q-table, the device form of
 *              `defaultdict(lambda: np.zeros(4))` (Agent/main.py:16).  Zero-filled = empty.
 *              16-byte aligned; 128-byte alignment makes every group of four slots one memory
 *              line, which is what the bucketised probe sequence is built for.
 *   - RNG      counter based: draws = Philox4x32-10(key = seed, counter = (global env id,
 *              step counter ctr, stream)); lane i has global env id env_id0 + i, so results do
 *              not depend on how a batch is sharded over GPUs.
 *   - return   0 on success, a negative Q2048_ERR_* otherwise (argument errors are detected
 *              on the host before anything is launched).  Data-dependent conditions are
 *              reported through the device `status` word (Q2048_STATUS_* bits, OR-ed).
 *   - n        board side, 4 or 5 (other values return Q2048_ERR_UNSUPPORTED).  The reference
 *              hard-codes 4 (Game2048_env.py:12,41); 5 is the same algorithm on uint8[B][25]
 *              boards (unpadded) with a 125-bit state key held in `key` + `reserved`.
 */
#ifndef Q2048_H
#define Q2048_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define Q2048_ABI_VERSION 7 /* 3: flag bits outside the ABI are refused (Q2048_ERR_FLAGS), Q2048_ST_CAS_FALLBACK,
                               deterministic step sorted by a hash of (state, action)
                               4: q2048_fused_rollout_opts (row cache + statistics mirror of the fused
                               rollout), q2048_table_grow, q2048_table_alloc verifies its zero fill
                               5: growth off the caller's critical path (q2048_table_grow_begin / _poll /
                               _commit / _finish / _abort)
                               6: Q2048_FLAG_NO_NEW_ROWS (a table that has stopped taking new rows)
                               7: Q2048_FLAG_LINE_SUMMARY + q2048_table_summarise, q2048_det_rollout_cached,
                               q2048_rowcache_rebind */

/* return codes */
#define Q2048_OK 0
#define Q2048_ERR_NULL (-1)        /* a required pointer is NULL */
#define Q2048_ERR_SIZE (-2)        /* negative batch / steps, bad cap_log2 */
#define Q2048_ERR_ALIGN (-3)       /* boards / aux / table not 16-byte aligned */
#define Q2048_ERR_UNSUPPORTED (-4) /* board side other than 4 or 5 */
#define Q2048_ERR_LAUNCH (-5)      /* the HIP runtime refused the launch */
#define Q2048_ERR_RANGE (-6)       /* a scalar is outside its domain (eps, lr, gamma) */
#define Q2048_ERR_FLAGS (-7)       /* flag bits outside the ABI, or flags the entry point refuses */
#define Q2048_ERR_ALLOC (-8)       /* q2048_table_alloc: memory could not be reserved / created / mapped */
#define Q2048_ERR_VERIFY (-9)      /* a table failed its self-check: a freshly mapped table did not read back as
                                      zeros, or q2048_table_grow did not find every row in the new table */
#define Q2048_ERR_BUSY (-10)       /* the table already takes part in a growth */
#define Q2048_PENDING 1            /* q2048_table_grow_poll: still working (not an error) */

/* bits of the device status word */
#define Q2048_STATUS_BAD_ACTION 1u    /* an action outside 0..3 was passed (lane left untouched) */
#define Q2048_STATUS_TILE_OVERFLOW 2u /* a tile above 2^15 does not fit the 64-bit state key */
#define Q2048_STATUS_TABLE_FULL 4u    /* an update found no slot within the probe limit (dropped) */
#define Q2048_STATUS_DEEP_ROW 8u      /* q2048_table_import placed a row deeper on its probe sequence than the learning
                                         paths look (2^10 slots; bulk moves and q2048_q_lookup take 2^14): lookup and
                                         export find it, choose / update / the rollouts read it as absent.  Only at
                                         loads above ~0.93: import into a larger table */

/* flags */
#define Q2048_FLAG_INDEPENDENT 1u /* every env owns private Q rows (key salted by its global id) */
#define Q2048_FLAG_SINGLE_ENV 2u  /* q_lookup: every board belongs to env `env_id0` (not env_id0 + i) */
#define Q2048_FLAG_TD_CAS 4u      /* TD update by a compare-and-swap loop: concurrent updates of one
                                     (s, a) serialise instead of "last writer wins" -- for up to 16
                                     attempts per update; an entry contended beyond that takes the
                                     update as a plain store, like the default mode, and the event
                                     is counted in Q2048_ST_CAS_FALLBACK: the guarantee is bounded,
                                     and the statistics say by how much.  Identical to the default
                                     whenever no two lanes share (s, a) */

#define Q2048_FLAG_ENV_DQN 8u      /* env step = the DQN path's env instead of Game2048_env.step:
                                     Deep_QLearning/environment/Game2048_nopenalty_env.py:106-138 --
                                     reward = calculate_reward2 (-10 for an invalid move while the game
                                     is not over, else the move's merge score; :122-138), done =
                                     game_over (:117-118), is_game_over evaluated on the board from
                                     BEFORE the move and, on a full board, replacing the result by its
                                     own first legal move (:68-78); the caller's write-back
                                     `env.game.board = next_state` (main_dir/mainDQL_CNN_step2.py:237)
                                     is part of the step.  No shaping state is read or written. */
#define Q2048_FLAG_RESET_SHAPING 16u /* a reset also restores previous_max and the consecutive-action
                                     state to their constructor values (Game2048_env.py:87,92-95).
                                     NOT the reference's behaviour (its reset, :187-191, keeps them:
                                     a lane ended by the >100-repeats rule ends again on its next
                                     repeat); opt-in fix of SURVEY 7.8 */
#define Q2048_FLAG_PLAY_ONLY 32u   /* fused rollout without a learner: the table is neither read nor
                                     written, every Q row reads as zeros.  With eps = 1 this is
                                     uniformly random play (input synthesis for benchmarks, the
                                     arithmetic floor of the kernel) */

#define Q2048_FLAG_NO_LEARN 64u    /* fused rollout that evaluates a trained table: rows are looked up
                                     (epsilon-greedy over the stored values), nothing is created or
                                     written.  What the `evaluate.py` the reference's README lists
                                     (README.md:52, no such file in the repository) has to do */

#define Q2048_FLAG_NO_NEW_ROWS 128u /* the table's key set is CLOSED for this call (a table that cannot
                                     grow any more "stops inserting and counts drops").  The reference's q_table is a
                                     defaultdict (Agent/main.py:16) that creates a row at every lookup of an unseen
                                     state (:38, :41, :43) and grows until the host swaps; a device table has a last
                                     capacity, and filling it to the brim makes every probe of an absent state walk
                                     hundreds of slots (12.9 ms per 2^20-board step at load 1.0 against 48 us).  With
                                     this flag
                                       - a state that has a row is read and updated exactly as without it;
                                       - a state without one reads as the zero row the defaultdict would have created
                                         (same greedy action, same bootstrap value 0), and NO row is created for it;
                                       - an update of Q[s][a] whose state has no row is dropped and counted in
                                         Q2048_ST_DROPS; Q2048_STATUS_TABLE_FULL is NOT raised (the caller asked);
                                       - VISIT ROWS: the dropped update is not lost on the env that made it.  The zero
                                         row the env read is ITS fresh row for as long as it stays in that state (the
                                         move was invalid): the update lands there, the env's next choose / update in
                                         the same state see it, and the row ends when the env moves on or its episode
                                         ends -- what the defaultdict's fresh row does within one visit (a negative
                                         reward for action 0 sends argmax on to action 1; without it a greedy env
                                         repeats an invalid action 0 until the >100-repeats rule ends the episode).
                                         The fused rollout keeps it in the lane's registers and hands it across
                                         launches through the row cache (a record without a slot); q_choose_cached /
                                         q_update_cached likewise, and q2048_det_rollout_cached keeps it in the row
                                         cache from step to step (one record format: the three paths hand visit rows to
                                         one another).  Without a row cache a visit row ends with the call -- in
                                         q2048_det_rollout, whose step keeps nothing per env, with the step.
                                     q2048_q_update / _cached, q2048_q_choose_cached, q2048_fused_rollout*,
                                     q2048_det_rollout(_cached); ignored by the entry points that neither create rows nor read
                                     the cache (q_choose, lookup, env, NO_LEARN, PLAY_ONLY).
                                     The host decides when: BatchedQLearningAgent(freeze_load=0.5) sets it on every
                                     launch once a table at its largest capacity holds that share of rows */

#define Q2048_FLAG_LINE_SUMMARY (1u << 24) /* with Q2048_FLAG_NO_NEW_ROWS, 4x4 tables: the table carries LINE SUMMARIES
                                     (q2048_table_summarise ran after its last row was created) and the fused rollout
                                     may decide a lookup from them.  A 4x4 slot has 8 spare bytes (`reserved`: the second
                                     key word of 5x5); a summary is four 16-bit fingerprints, one per slot of the
                                     128-byte line (0 = empty), the same word in all four slots.  The lookup reads the
                                     second half of the first slot of its sequence -- {q2, q3, summary}: ONE request --
                                     and knows where the sequence ends: at the first slot, in its order, that is empty
                                     (absent: done) or carries the key's fingerprint (that slot's head, one more
                                     request, settles it).  The slot-by-slot probe sends 2.35 requests per lookup on a
                                     table at load 0.5 and the step takes 47.9 us per 2^20 boards; with summaries
                                     1.25 and 34.8 us (28.7 on 64-step launches).  Same results, fewer requests.
                                     CONTRACT: the summaries describe the key set as it was when q2048_table_summarise
                                     ran.  Creating a row afterwards (any call without Q2048_FLAG_NO_NEW_ROWS, an import)
                                     makes them stale, and a stale summary HIDES rows: summarise again before the flag is
                                     passed again.  Ignored by every entry point but q2048_fused_rollout*, and there for
                                     5x5 tables, evaluation and play-only launches (they probe slot by slot; the extra
                                     word never disturbs them: 4x4 lookups, export with key_words 1, count, import and the
                                     growth's move neither read nor write it).
                                     (bits 8..23 belong to the measurement build's experiment switches) */

/* per-env state, Game2048_env.__init__ (Game2048_env.py:81-95) + episode bookkeeping */
typedef struct q2048_aux {
  int32_t score;       /* env.score (:84); zeroed by reset (:190) */
  float ep_return;     /* running total_reward of Agent/main.py:84,101 */
  uint8_t prev_max;    /* log2(env.previous_max) (:87); NOT reset (:187-191) */
  uint8_t cons_action; /* env.consecutive_action (:92), 0xFF = None; NOT reset */
  uint16_t cons_count; /* env.consecutive_count (:93), saturates at 60000; NOT reset */
  uint32_t episode;    /* number of resets so far (counter of the reset draws) */
} q2048_aux;

/* one row of the Q-table: q_table[state] -> 4 floats (Agent/main.py:16) */
typedef struct q2048_slot {
  uint64_t key;      /* 4x4: 16 log2 nibbles, cell 0 in the low nibble.  5x5: bits 0..62 of the 125-bit
                        key (25 cells x 5 bits) | bit 63.  0 = empty slot */
  float q[4];        /* Q(s, a), a = 0 left, 1 up, 2 right, 3 down */
  uint64_t reserved; /* 4x4: 0 (keeps rows 32-byte aligned: a row never straddles a 64-byte line).
                        5x5: bits 63..124 of the key | bit 63, published right after `key` */
} q2048_slot;

/* one finished episode = one row of the reference's debug_log.csv (Agent/main.py:59-62,71-76:
 * Episode, Action, Q-Values, Reward, Total-Reward, Max Value), written by q2048_fused_rollout_log */
typedef struct q2048_episode {
  uint64_t env_id;    /* global env id */
  uint32_t episode;   /* index of the finished episode within that env ("Episode") */
  uint8_t action;     /* last action ("Action") */
  uint8_t max_log2;   /* log2 of the max tile of the final board ("Max Value" = 2^max_log2) */
  uint16_t steps_lo;  /* low 16 bits of the global step counter at which the episode ended */
  float reward;       /* last reward ("Reward") */
  float total_return; /* sum of the episode's rewards ("Total-Reward") */
  int32_t score;      /* env.score at the end */
  float q[4];         /* post-update Q row of the last state ("Q-Values", main.py:96 aliases the row) */
  uint32_t reserved;
} q2048_episode;

/* indices of the statistics vectors (device int64[Q2048_NSTAT_I], double[Q2048_NSTAT_F]);
 * kernels ADD to them, the caller zeroes them */
enum {
  Q2048_ST_STEPS = 0,    /* env steps executed */
  Q2048_ST_EPISODES = 1, /* episodes finished */
  Q2048_ST_VALID = 2,    /* board-changing moves */
  Q2048_ST_SCORE = 3,    /* sum of env.score over finished episodes */
  Q2048_ST_INSERTS = 4,  /* Q rows created */
  Q2048_ST_DROPS = 5,    /* updates dropped: no slot within the probe limit, or (Q2048_FLAG_NO_NEW_ROWS) no row */
  Q2048_ST_EXPLORE = 6,  /* epsilon branch taken */
  Q2048_ST_CAS_RETRY = 7,/* TD compare-and-swap retries (same (s,a) updated concurrently) */
  Q2048_ST_HIST0 = 8,    /* max-tile histogram of finished episodes, log2 0..22 (saturating) */
  Q2048_ST_CAS_FALLBACK = 31, /* Q2048_FLAG_TD_CAS updates that lost 16 races in a row and were
                                 written as a plain store instead */
  Q2048_NSTAT_I = 32
};
enum { Q2048_SF_RETURN = 0, Q2048_SF_RETURN_SQ = 1, Q2048_SF_REWARD = 2, Q2048_NSTAT_F = 4 };

int q2048_abi_version(void);

/* Diagnostic, host-synchronous: lanes of the current device that ever gave up waiting for the
 * second key word of a 5x5 row under creation (the wait is bounded so that a protocol error can
 * never hang the device).  Expected value: 0, always; the tests assert it. */
int q2048_claim_timeouts(uint64_t *count_host);
const char *q2048_strerror(int code);
size_t q2048_sizeof_aux(void);  /* 16 */
size_t q2048_sizeof_slot(void); /* 32 */

/* Game2048_env() constructor for B envs (Game2048_env.py:81-95 -> Game2048.__init__ :11-14):
 * empty board + two spawns from the reset draws of episode 0, aux = initial values. */
int q2048_env_init(uint8_t *boards, q2048_aux *aux, int64_t B, int n, uint64_t seed,
                   uint64_t env_id0, void *stream);

/* Game2048_env.reset() (Game2048_env.py:187-191) for the lanes with mask[i] != 0 (all lanes
 * when mask == NULL): new board with two spawns, score = 0; previous_max and the
 * consecutive-action state persist, as in the reference. */
int q2048_env_reset(uint8_t *boards, q2048_aux *aux, const uint8_t *mask, int64_t B, int n,
                    uint64_t seed, uint64_t env_id0, void *stream);

/* q2048_env_reset with flags (Q2048_FLAG_RESET_SHAPING; other bits ignored). */
int q2048_env_reset_ex(uint8_t *boards, q2048_aux *aux, const uint8_t *mask, int64_t B, int n,
                       uint64_t seed, uint64_t env_id0, uint32_t flags, void *stream);

/* Game2048_env.step(action) (Game2048_env.py:97-129) for B envs: move (:51-63), game-over
 * probe (:65-75), shaped reward (:136-184, :197-205), stall rule (:110-127).
 * Outputs per lane: reward (float32 of the reference's float), done, max tile as log2
 * (reference `info` = 2^max_log2).  An action outside 0..3 leaves its lane untouched and
 * sets Q2048_STATUS_BAD_ACTION (the reference would silently mis-rotate, :56-60). */
int q2048_env_step(uint8_t *boards, q2048_aux *aux, const uint8_t *actions, int64_t B, int n,
                   uint64_t seed, uint64_t env_id0, uint32_t ctr, float *reward, uint8_t *done,
                   uint8_t *max_log2, uint32_t *status, void *stream);

/* q2048_env_step with the two spawn decisions' raw draws given per lane instead of derived from
 * (seed, id, ctr): draw_pos replaces np.random.randint(0, n_empty) (Game2048_env.py:19),
 * draw_val replaces np.random.random() < 0.9 (:20).  Draw injection is how parity with the
 * reference is pinned (tests/golden); training uses q2048_env_step. */
int q2048_env_step_draws(uint8_t *boards, q2048_aux *aux, const uint8_t *actions,
                         const uint32_t *draw_pos, const uint32_t *draw_val, int64_t B, int n,
                         float *reward, uint8_t *done, uint8_t *max_log2, uint32_t *status,
                         void *stream);

/* q2048_env_step with an env profile: flags = Q2048_FLAG_ENV_DQN selects the DQN path's step
 * (Game2048_nopenalty_env.py:106-138, see the flag); 0 = q2048_env_step.  draws4 = NULL: draws
 * derived from (seed, id, ctr) -- the chosen move's spawn from words 2, 3 of stream 0 as always,
 * the spawn inside is_game_over's move from words 0, 1 of stream 2.  draws4 = uint32[B][4]:
 * injected draws {pos, val, over_pos, over_val} per env (parity with the reference; seed /
 * env_id0 / ctr are then unused). */
int q2048_env_step_ex(uint8_t *boards, q2048_aux *aux, const uint8_t *actions, int64_t B, int n,
                      uint64_t seed, uint64_t env_id0, uint32_t ctr, uint32_t flags,
                      const uint32_t *draws4, float *reward, uint8_t *done, uint8_t *max_log2,
                      uint32_t *status, void *stream);

/* q2048_env_step_ex (draws derived from (seed, id, ctr)) reading boards_in and writing boards_out:
 * the same buffer (in place), or two buffers that do not overlap -- then the state before the step
 * stays intact for update_q_value(state, ...) and the batched loop of Agent/main.py:92-100 needs no
 * board copy (the host side ping-pongs two buffers).  A lane with a rejected action copies its
 * board unchanged.  max_tile (may be NULL): int32[B], the reference's `info` -- the raw max tile
 * (Game2048_env.py:100,129) -- written next to its log2. */
int q2048_env_step_to(const uint8_t *boards_in, uint8_t *boards_out, q2048_aux *aux,
                      const uint8_t *actions, int64_t B, int n, uint64_t seed, uint64_t env_id0,
                      uint32_t ctr, uint32_t flags, float *reward, uint8_t *done, uint8_t *max_log2,
                      int32_t *max_tile, uint32_t *status, void *stream);

/* QLearningAgent.choose_action(state) (Agent/main.py:34-38) for B states: epsilon test and
 * random action from the step draws, else first-maximum argmax of the row (zeros if absent;
 * a lookup never inserts -- value-equivalent to the defaultdict). */
int q2048_q_choose(const q2048_slot *table, int cap_log2, const uint8_t *boards, int64_t B,
                   int n, double eps, uint64_t seed, uint64_t env_id0, uint32_t ctr,
                   uint32_t flags, uint8_t *actions, uint32_t *status, void *stream);

/* q2048_q_choose with injected draws: draw_eps replaces random.random() (Agent/main.py:35),
 * draw_act replaces random.randint(0, 3) (:36). */
int q2048_q_choose_draws(const q2048_slot *table, int cap_log2, const uint8_t *boards,
                         const uint32_t *draw_eps, const uint32_t *draw_act, int64_t B, int n,
                         double eps, uint64_t env_id0, uint32_t flags, uint8_t *actions,
                         uint32_t *status, void *stream);

/* QLearningAgent.update_q_value(state, action, reward, next_state, done)
 * (Agent/main.py:40-43) for B transitions.  Rows for next_state and state are created when
 * absent, as the reference's defaultdict does (:41-43).  Q[s][a] is written with one 4-byte
 * store: lanes that update the same (s, a) concurrently race and the last writer wins
 * (Q2048_FLAG_TD_CAS serialises them with a compare-and-swap loop instead).  With one lane, or
 * lanes on disjoint states, this is exactly the reference's sequential update. */
int q2048_q_update(q2048_slot *table, int cap_log2, const uint8_t *boards_s,
                   const uint8_t *actions, const float *reward, const uint8_t *boards_s2,
                   const uint8_t *done, int64_t B, int n, double lr, double gamma,
                   uint64_t env_id0, uint32_t flags, int64_t *stats_i, uint32_t *status,
                   void *stream);

/* Row cache of the 4-call API: q2048_q_update / q2048_q_choose with a caller-owned device buffer of
 * B records of q2048_sizeof_rowcache(n) bytes (32 for n = 4, 48 for n = 5; 16-byte aligned;
 * zero-filled = empty; NULL = the plain entry points).  q_update leaves in record i the row env i
 * read as next_state -- key, slot, the four values, with its own write folded in when the move was
 * invalid -- and the next q_choose / q_update of the same env use it instead of probing when the
 * board they are given has that key: in the loop of Agent/main.py:92-100 `state` IS the previous
 * `next_state` unless an episode began, so an update costs one scattered row read instead of two
 * and a greedy choose none.  This is what the fused rollout carries in registers, handed over
 * through HBM as a stream; like it, a cached row does not see what OTHER envs wrote to it since.
 * Any calling pattern is correct: a record is used only when its key is the key of the board passed in AND it
 * was left by a call on this very table.  The contract for that second half: THE CALLER ZERO-FILLS THE CACHE whenever
 * the table behind it changes -- a growth's commit (slots change), another table, a table rewritten in place
 * (zero-filled, imported into, updated through an entry point without the cache).  As a safety net every record also
 * carries a 24-bit tag of the table's address and capacity, so that records of another table miss instead of steering
 * writes through stale slot indices -- with probability 1 - 2^-24 per pair of tables: a net, not the contract. */
size_t q2048_sizeof_rowcache(int n);
/* A cache's VISIT ROWS (Q2048_FLAG_NO_NEW_ROWS) follow their table's rows into another allocation -- a checkpoint of
 * a learner with a closed key set, restored: the host saves the cache's bytes beside the table's rows, the old
 * table's address and capacity (two numbers; `from_table` is never dereferenced), restores the rows into `to_table`
 * (any capacity that holds them) and calls this on the restored bytes.  Every record without a slot that a call on
 * `from_table` left becomes one of `to_table`; every other record is emptied (a slot index of the old allocation
 * means nothing in the new one).  The resumed run then equals the uninterrupted one step for step; without it the
 * envs that were in a state without a row at the checkpoint start that visit again from the zero row.  The two
 * tables must hold the SAME KEY SET (a visit row stands in for a row that does not exist). */
int q2048_rowcache_rebind(void *row_cache, int64_t B, int n, const q2048_slot *from_table, int from_cap_log2,
                          const q2048_slot *to_table, int to_cap_log2, void *stream);
int q2048_q_choose_cached(const q2048_slot *table, int cap_log2, const uint8_t *boards, int64_t B,
                          int n, double eps, uint64_t seed, uint64_t env_id0, uint32_t ctr,
                          uint32_t flags, const void *row_cache, uint8_t *actions,
                          uint32_t *status, void *stream);
int q2048_q_update_cached(q2048_slot *table, int cap_log2, const uint8_t *boards_s,
                          const uint8_t *actions, const float *reward, const uint8_t *boards_s2,
                          const uint8_t *done, int64_t B, int n, double lr, double gamma,
                          uint64_t env_id0, uint32_t flags, void *row_cache, int64_t *stats_i,
                          uint32_t *status, void *stream);

/* agent.q_table[state] (Agent/main.py:16,96) for B states: q_out[B][4] (zeros if absent),
 * found[B] (may be NULL). */
int q2048_q_lookup(const q2048_slot *table, int cap_log2, const uint8_t *boards, int64_t B,
                   int n, uint64_t env_id0, uint32_t flags, float *q_out, uint8_t *found,
                   uint32_t *status, void *stream);

/* The loop body of Agent/main.py:91-101 + the reset of :81, `steps` times for B envs in ONE
 * launch: choose -> step -> update -> accumulate -> (on done) statistics and reset.  Boards,
 * aux and the Q row of the current state stay in registers between steps.  Step t uses the
 * draws of counter ctr0 + t.  Bit-identical to calling q_choose / env_step / q_update /
 * env_reset(done) `steps` times whenever no two lanes share a state.
 * flags: Q2048_FLAG_INDEPENDENT, _TD_CAS, _ENV_DQN, _RESET_SHAPING, _PLAY_ONLY, _NO_LEARN, _NO_NEW_ROWS; any
 * bit outside the Q2048_FLAG_* set is Q2048_ERR_FLAGS (all entry points). */
int q2048_fused_rollout(uint8_t *boards, q2048_aux *aux, q2048_slot *table, int cap_log2,
                        int64_t B, int n, int64_t steps, double eps, double lr, double gamma,
                        uint64_t seed, uint64_t env_id0, uint32_t ctr0, uint32_t flags,
                        int64_t *stats_i, double *stats_f, uint32_t *status, void *stream);

/* q2048_fused_rollout that also appends one q2048_episode record per finished episode to
 * log[0 .. log_capacity) (log_count is a device uint64 cursor the caller zeroes; records beyond
 * the capacity are counted but not written).  This is the device form of log_debug_info
 * (Agent/main.py:59-62, called at :103-105). */
int q2048_fused_rollout_log(uint8_t *boards, q2048_aux *aux, q2048_slot *table, int cap_log2,
                            int64_t B, int n, int64_t steps, double eps, double lr, double gamma,
                            uint64_t seed, uint64_t env_id0, uint32_t ctr0, uint32_t flags,
                            int64_t *stats_i, double *stats_f, uint32_t *status,
                            q2048_episode *log, int64_t log_capacity, uint64_t *log_count,
                            void *stream);

/* q2048_fused_rollout with optional extras, all caller-owned, any of them NULL (opts == NULL is
 * q2048_fused_rollout itself).  The launch boundary cuts the loop of Agent/main.py:91-101 between two
 * iterations; the extras are what makes a cut cheap:
 *   log, log_capacity, log_count   as q2048_fused_rollout_log.
 *   row_cache   the row cache of the 4-call API (above: q2048_sizeof_rowcache(n) bytes per env, 16-byte
 *               aligned, zero-filled = empty), shared with it.  A launch ENDS by leaving in record i the row
 *               env i carries in registers -- key, the four values, slot -- and STARTS from that record when
 *               its key is the key of the board it is given (a coalesced 32-byte read) instead of probing
 *               the table (a scattered 128-byte request per lane: 21.5 us of every launch at 2^20 boards).
 *               K steps in launches of S with the cache are the K-step launch: a row is carried across the
 *               cut exactly as across a step.  Any calling pattern is correct (a record is used only on a
 *               key match); like a row carried in registers, a record does not see what OTHER envs wrote to
 *               its row since.  Q2048_FLAG_PLAY_ONLY launches neither read nor write it.
 *   stats_mirror, mirror_ticket   stats_mirror: (Q2048_NSTAT_I + Q2048_NSTAT_F + 1) 8-byte words the HOST
 *               can read while the device writes them -- pinned host memory mapped into the device's
 *               address space (hipHostMalloc; any 8-byte aligned pointer the device can store to works).
 *               The last block of the launch to finish copies stats_i then stats_f (both required then), as
 *               they stand after the whole launch's additions, into words 0..35 and writes the number of
 *               mirrored launches so far into word Q2048_MIRROR_SEQ: a caller that waits for the launch
 *               anyway (hipStreamSynchronize) reads its statistics from host memory with no copy queued
 *               behind the kernel.  mirror_ticket: device uint32[2], zero before its first use, private to
 *               one stream of launches (two launches that share it must not overlap).
 * `size` = sizeof(q2048_rollout_opts): a caller built against another layout is refused (Q2048_ERR_SIZE). */
#define Q2048_MIRROR_SEQ 36 /* = Q2048_NSTAT_I + Q2048_NSTAT_F */
#define Q2048_MIRROR_WORDS 37
typedef struct q2048_rollout_opts {
  uint32_t size;
  uint32_t reserved;
  q2048_episode *log;
  int64_t log_capacity;
  uint64_t *log_count;
  void *row_cache;
  void *stats_mirror;
  uint32_t *mirror_ticket;
} q2048_rollout_opts;
int q2048_fused_rollout_opts(uint8_t *boards, q2048_aux *aux, q2048_slot *table, int cap_log2,
                             int64_t B, int n, int64_t steps, double eps, double lr, double gamma,
                             uint64_t seed, uint64_t env_id0, uint32_t ctr0, uint32_t flags,
                             int64_t *stats_i, double *stats_f, uint32_t *status,
                             const q2048_rollout_opts *opts, void *stream);

/* Deterministic mode: reproducible shared-table training at any B (the default rollout is
 * lock-free and depends on scheduling wherever lanes share a state).  Per step:
 *   phase 1  every env chooses on, steps, and reads max Q(s') from the table as it is at the
 *            START of the step (phase 1 writes no Q value; it creates the rows of s and s' like
 *            update_q_value does, Agent/main.py:41-43) and emits its update: (row slot, action)
 *            and the TD target reward + gamma * max Q(s') * (1 - done) (:42);
 *   sort     the updates are sorted on the device, stably, by a 16-bit hash of (state, action):
 *            the updates of a group end up in one short run, in env order;
 *   phase 2  every group applies Q[s][a] += lr * (target - Q[s][a]) (:43) update by update in env
 *            order, in the reference's double arithmetic (no fused multiply-add), and rounds to
 *            float32 once per group and step.
 * The result equals the reference agent fed the step's transitions in env order against the
 * step-start table with its rows held as float32, for any B, and is a function of the inputs
 * alone: the sort key depends on the state and the action, never on which slot a racing insert
 * won, and every group is folded by the same sequence of operations whatever its size.  `steps`
 * steps per call, draws of counter ctr0 + t.  flags: Q2048_FLAG_INDEPENDENT, _ENV_DQN,
 * _RESET_SHAPING, _NO_NEW_ROWS; _TD_CAS is ignored; _NO_LEARN and _PLAY_ONLY are refused (Q2048_ERR_FLAGS: there
 * is no evaluation form of this step, and learning anyway would be the wrong answer).
 * `workspace`: caller-owned device scratch of q2048_det_workspace_bytes(B, cap_log2) bytes
 * (about 36 bytes per env; host arithmetic, needs no device), 256-byte aligned (B < 2^31).  It
 * holds no state between calls. */
int64_t q2048_det_workspace_bytes(int64_t B, int cap_log2);
int q2048_det_rollout(uint8_t *boards, q2048_aux *aux, q2048_slot *table, int cap_log2, int64_t B,
                      int n, int64_t steps, double eps, double lr, double gamma, uint64_t seed,
                      uint64_t env_id0, uint32_t ctr0, uint32_t flags, int64_t *stats_i,
                      double *stats_f, uint32_t *status, void *workspace, int64_t workspace_bytes,
                      void *stream);
/* ... with the envs' row cache (B records of q2048_sizeof_rowcache(n) bytes, 16-byte aligned, the one the fused
 * rollout and the _cached calls take; NULL = q2048_det_rollout).  Used with Q2048_FLAG_NO_NEW_ROWS only, for VISIT
 * ROWS: an env in a state without a row reads its visit row in phase 1, and while it stays there its update --
 * dropped from the table and counted as before -- lands in that row (float32, the fused rollout's arithmetic; it is
 * one env's, so it needs no ordering and the result stays a function of the inputs alone).  Every record the call
 * touches is left either a visit row or empty: records of table rows left by a fused launch are cleared, never
 * used.  Without the flag the cache is neither read nor written. */
int q2048_det_rollout_cached(uint8_t *boards, q2048_aux *aux, q2048_slot *table, int cap_log2, int64_t B,
                             int n, int64_t steps, double eps, double lr, double gamma, uint64_t seed,
                             uint64_t env_id0, uint32_t ctr0, uint32_t flags, int64_t *stats_i,
                             double *stats_f, uint32_t *status, void *workspace, int64_t workspace_bytes,
                             void *row_cache, void *stream);

/* Table allocation from small physical chunks -- the ONLY entry points that allocate (everything
 * else works on caller-owned memory, however it was obtained; a table from hipMalloc / a framework
 * allocator is as valid).  Why it exists: the rate at which this memory system takes scattered
 * 4-byte stores and atomics depends on how the table's memory was obtained.  A table mapped from
 * 2 MiB physical chunks (HIP virtual-memory API: hipMemCreate + hipMemMap into one reserved range)
 * takes them 15-20 % faster than a hipMalloc of the same 8-32 GiB (load + compare-and-swap + store
 * per lane-step: 46.2 against 55.4 us per 2^20 on an 8 GiB table, 51.2 with 64 MiB chunks;
 * profiles/r03_requests/vmm_*_8GiB.txt; loads do not care) -- as fast as a table that spans 128 GiB.
 * (Round 4, six fresh allocations per chunk size: 2, 8, 32 and 64 MiB chunks overlap completely,
 * profiles/r04_requests/chunk_size_six_draws.txt -- the effect is mapped memory against hipMalloc and
 * which allocation one got, not the chunk size.)
 *
 * q2048_table_alloc reserves, creates, maps and zero-fills 2^cap_log2 slots (chunk_bytes = 0: 2 MiB;
 * else a power of two that is a multiple of the allocation granularity) on the current device, and VERIFIES the zero fill
 * with one streaming count of the table before it returns (Q2048_ERR_VERIFY otherwise: a slot that
 * wrongly looks occupied is how rows would get lost silently).  Host-synchronous.
 *
 * A table that grows -- the reference's defaultdict (Agent/main.py:16) has no capacity:
 * q2048_table_reserve maps a table of 2^cap_log2 slots that may grow up to 2^max_cap_log2 (and allocates the
 * family's 64 bytes of scratch and its private stream: nothing below calls hipMalloc / hipFree).  Growing =
 * mapping the table of capacity 2^new_cap_log2 (cap_log2 < new_cap_log2 <= max_cap_log2) onto fresh chunks in
 * an address range of its own, moving every row over in one streaming pass (the rows keep their values; their
 * slots change: zero-fill any row cache), and releasing the old table.  The table's ADDRESS changes.
 *
 * Off the caller's critical path (the reference's dict grows without stopping the loop, Agent/main.py:16,91-101):
 *   q2048_table_grow_begin(table, cap_log2, new_cap_log2, &g)   returns at once; ONE host thread of the library
 *       (started by the first call, joined at exit) reserves, creates, maps, zero-fills and verifies the new table
 *       on the family's own stream while the caller keeps launching rollouts on the old one: 31 ms for 32 GiB,
 *       119 ms for 128 GiB next to a stream of launches that slow by 16 % meanwhile
 *       (profiles/r05_vmm_cost_small_chunks.txt).
 *   q2048_table_grow_poll(g)      Q2048_OK: the next call (commit, or finish after a commit) will not block;
 *       Q2048_PENDING: it would; < 0: the preparation failed (commit returns the same code and ends the growth).
 *   q2048_table_grow_wait(g, &ms)   blocks until the preparation is over (a caller that prefers to wait before its
 *       clock starts rather than at the commit) and returns its outcome; ms (host double, may be NULL) = what the
 *       host thread spent on it.
 *   q2048_table_grow_commit(g, key_words, flags, &bigger, stream)   waits for the preparation if need be, then
 *       enqueues the move (k_table_rehash, 18 G rows/s) on `stream`, behind whatever the caller queued on the old
 *       table, and returns the new table WITHOUT waiting for it: every launch queued on `stream` from here on
 *       takes *bigger.  flags: Q2048_GROW_VERIFY_COUNT also counts the new table's rows behind the move (one more
 *       streaming pass, 21 ms per 128 GiB).  Errors: Q2048_ERR_NULL / _SIZE / _FLAGS (arguments) and Q2048_ERR_BUSY
 *       (another growth of the family is committed and not yet finished, or another thread is committing this very
 *       growth) change nothing -- g is STILL VALID: retry after the other growth's finish, or q2048_table_grow_abort(g).
 *       Any other error (the preparation failed: its code; Q2048_ERR_LAUNCH) ends the growth: the old table is intact
 *       and still the caller's, the prepared table has been released, g is gone.
 *   q2048_table_grow_finish(g, &rows)   waits for the move, checks it -- every occupied slot of the old table found
 *       its place (and, with VERIFY_COUNT, the new table holds exactly that many rows): Q2048_ERR_VERIFY otherwise,
 *       both tables then stay live and g is gone; Q2048_ERR_LAUNCH when the wait itself failed: nothing is known
 *       about the move, g stays valid (retry, or free the tables: q2048_table_free resolves the growth) -- and
 *       RETIRES the old table: it is the library's from here on (do not free or
 *       touch it) and its memory is kept until a mapping on the device finds no room, the family's last live table
 *       is freed, or q2048_table_trim(live table) asks for it -- because memory a process releases is wiped by the
 *       driver at ~40 GB/s before it is handed out again, and the next mapping would wait for that: hipMemCreate of
 *       128 GiB takes 30 ms on memory that has been free for a while and 3-4 s right after 128 GiB were released,
 *       by this or by the previous process (profiles/r05_vmm_wipe.txt).  rows (host int64, may be NULL) = rows
 *       moved = the occupied slots of the old table: compare it with the rows the kernels reported
 *       (Q2048_ST_INSERTS) and the old table is verified end to end.
 *   q2048_table_grow_abort(g)     before the commit: the prepared table is released, the old one untouched.
 * One growth per table at a time (Q2048_ERR_BUSY); the next growth of the NEW table may begin right after the
 * commit, its commit only after the previous finish.  q2048_table_grow(...) is begin + commit(VERIFY_COUNT) +
 * finish + the release of the old table in one host-synchronous call.  While both tables exist the device holds both.
 *
 * Chunk sizes: a table of a family that can grow is cut into as few chunks as 32 MiB allows (2 MiB up to 2 GiB,
 * bytes / 1024 up to 32 GiB, 32 MiB beyond: 4096 chunks for 128 GiB).  hipMemMap / hipMemUnmap / hipMemSetAccess
 * cost 5-16 us per chunk (more, the more chunks a process holds: 2 MiB chunks for 128 GiB took 13.8 s,
 * profiles/r04_growth_phases_2MiB_chunks.txt); what hipMemCreate costs depends on the state of the device's free
 * memory, not on the chunk: 4-30 us per chunk on memory that has been free for a while, seconds per table (at any
 * chunk size from 8 MiB to 1 GiB) when the driver hands out memory that was released moments before, by this or
 * by the previous process, and is still being wiped (profiles/r05_vmm_cost_by_chunk.txt,
 * r05_vmm_cost_small_chunks.txt, r05_vmm_wipe.txt: 3.2-3.9 s right after, 30 ms twelve seconds later) -- which is
 * what round 4's 1.9 s for the step to 128 GiB was.  q2048_table_alloc keeps the chunk size it is asked for.
 * The caller decides when to grow: between launches, when rows created / capacity passes its load limit (the
 * rollout's step slows from 46.7 to 66.3 us as the load goes from 0.12 to 0.54, profiles/r03_load_curve.jsonl).
 *
 * q2048_table_free unmaps a table's chunks and releases their physical memory (it synchronises the
 * table's device first, whatever the calling thread's current device is; a growth the table takes part in is
 * resolved first: aborted if uncommitted, finished if committed).  THE ADDRESS RANGE STAYS
 * RESERVED until the process ends and no part of it is ever mapped a second time -- a range that is
 * freed, reserved again and mapped onto new chunks serves stale translations on ROCm 7.2: tables on a re-used
 * range lose 0.01-1.3 % of the rows written to them, with every HIP call returning success.  Library-free
 * reproducer: tools/va_reuse_repro.hip (152 lines of HIP; profiles/r05_va_reuse_repro.txt: 0 of 8 tables lose
 * rows on fresh ranges, 4 of 8 on a re-used one, 1 of 8 with a device synchronize after the free).  A process
 * therefore accumulates reserved address space -- 2^-12 of its 47-bit space per 32 GiB table -- not memory.
 * For the same reason tables are reserved in a PRIVATE REGION of the address space (16 TiB upward, one range per table
 * from a process-wide cursor; the runtime's own allocations grow down from the top): after a large hipFree the next
 * un-hinted hipMemAddressReserve returns exactly the freed address (tools/va_hint_probe.hip), so a table reserved the
 * ordinary way could sit on a range some other allocator had mapped before.  And room is asked for (hipMemGetInfo)
 * before a single chunk is created: Q2048_ERR_ALLOC comes back at once when the device cannot hold the table, because
 * mapping chunk by chunk until hipMemCreate fails left this stack in a state in which the process's next large mapping
 * faulted the GPU (profiles/r06_va_reuse_fault.txt).
 * All of these are thread-safe. */
#define Q2048_GROW_VERIFY_COUNT 1u
typedef struct q2048_growth q2048_growth; /* opaque: a growth between begin and finish / abort */
int q2048_table_alloc(int cap_log2, size_t chunk_bytes, q2048_slot **table_out);
int q2048_table_reserve(int cap_log2, int max_cap_log2, size_t chunk_bytes, q2048_slot **table_out);
int q2048_table_grow_begin(q2048_slot *table, int cap_log2, int new_cap_log2, q2048_growth **growth_out);
int q2048_table_grow_poll(q2048_growth *growth);
int q2048_table_grow_wait(q2048_growth *growth, double *prepare_ms);
int q2048_table_grow_commit(q2048_growth *growth, int key_words, uint32_t flags, q2048_slot **table_out,
                            void *stream);
int q2048_table_grow_finish(q2048_growth *growth, int64_t *rows_moved);
int q2048_table_grow_abort(q2048_growth *growth);
int q2048_table_grow(q2048_slot *table, int cap_log2, int new_cap_log2, int key_words,
                     q2048_slot **table_out, int64_t *rows_moved, void *stream);
int q2048_table_trim(q2048_slot *table); /* releases the retired predecessors of this table's family now */
int q2048_table_free(q2048_slot *table);

/* Placement probe (no reference counterpart): `lanes` lanes each issue `steps` scattered
 * device-scope atomic ORs of 0 on key words of the table -- the write-side request pattern of
 * the rollout, leaving every byte of the table as it was, so it may run on a live table.  The
 * caller times it to choose among candidate allocations (the scattered write / atomic rate
 * depends on where in device memory a table lies; reads do not). */
int q2048_table_probe(q2048_slot *table, int cap_log2, int64_t lanes, int steps, uint64_t seed,
                      void *stream);

/* Writes the LINE SUMMARIES of a 4x4 table whose key set is closed (Q2048_FLAG_LINE_SUMMARY above): one streaming pass
 * over the table (reads every key, writes every slot's `reserved` word; 19.7 ms per 32 GiB), stream-ordered.  Run it
 * after the last row was created and before the first launch that carries the flag; run it again if rows were created
 * since.  Never on a 5x5 table (its `reserved` words are key words). */
int q2048_table_summarise(q2048_slot *table, int cap_log2, void *stream);

/* len(agent.q_table): adds the number of occupied slots to *count (device int64). */
int q2048_table_count(const q2048_slot *table, int cap_log2, int64_t *count, void *stream);

/* Export occupied rows (for conversion to the reference's dict{state -> 4 floats},
 * Agent/main.py:16): writes up to max_rows (key, q[4]) pairs in unspecified order and adds the
 * number of occupied slots to *count (rows beyond max_rows are counted, not written).
 * key_words = 1 (4x4: keys_out[max_rows]) or 2 (5x5: keys_out[max_rows][2] = key, reserved). */
int q2048_table_export(const q2048_slot *table, int cap_log2, uint64_t *keys_out, float *q_out,
                       int64_t max_rows, int key_words, int64_t *count, void *stream);

/* Legal-move mask: bit a of mask_out[i] is set iff action a would change board i -- the trial-move
 * loop of Deep_QLearning/main_dir/mainDQL_CNN_step2.py:168-174 (`env.game.move(action,
 * trial=True)`), the reference's way of listing legal moves.  No board is modified. */
int q2048_legal_moves(const uint8_t *boards, int64_t B, int n, uint8_t *mask_out, void *stream);

/* One-hot state encoder of the reference's DQN front-end (Deep_QLearning/main_dir/
 * Dqn8TestNOPERCNN.py:271-277: one_hot(log2(tile), depth 16) transposed to [16, 4, 4]):
 * out[i][c][r][col] = 1 if the log2 tile at (r, col) equals c (empty cells are channel 0, tiles
 * of 2^16 and above encode as all zeros, as tf.one_hot does), else 0.  4x4 boards;
 * out is float32 (dtype = 0) or bfloat16 (dtype = 1), 16-byte aligned, [B][16][4][4]. */
int q2048_encode_onehot(const uint8_t *boards, int64_t B, int dtype, void *out, void *stream);

/* ---- row-tuple linear Q: BASELINE configs[1], "flat-array Q over row-tuple features" ----------
 * NOT the reference's learner (its Q is keyed by the whole board, Agent/main.py:82) but the same
 * loop with a different table: weights = float[4][65536][4] (16-byte aligned, zero = untrained),
 * Q(s,a) = sum over rows r of weights[r][idx_r(s)][a], idx_r = the row's four log2 nibbles.
 * Same epsilon-greedy / TD target as Agent/main.py:34-43; the error is spread evenly over the
 * four weights, each written as (value read + delta) -- concurrent lanes race per weight and the
 * last writer wins (summing all lanes' deltas would scale the step size by the batch size).
 * 4x4 boards only. */
int q2048_rt_choose(const float *weights, const uint8_t *boards, int64_t B, double eps,
                    uint64_t seed, uint64_t env_id0, uint32_t ctr, uint8_t *actions, void *stream);
int q2048_rt_lookup(const float *weights, const uint8_t *boards, int64_t B, float *q_out,
                    void *stream);
int q2048_rt_update(float *weights, const uint8_t *boards_s, const uint8_t *actions,
                    const float *reward, const uint8_t *boards_s2, const uint8_t *done, int64_t B,
                    double lr, double gamma, uint32_t *status, void *stream);
int q2048_rt_fused_rollout(uint8_t *boards, q2048_aux *aux, float *weights, int64_t B,
                           int64_t steps, double eps, double lr, double gamma, uint64_t seed,
                           uint64_t env_id0, uint32_t ctr0, int64_t *stats_i, double *stats_f,
                           uint32_t *status, void *stream);

/* Inverse of q2048_table_export (resume / load a table trained elsewhere): inserts `rows`
 * (key, q[4]) pairs into the table, key_words as above.  A key already present has its row
 * overwritten; a row that finds no slot within the probe limit sets Q2048_STATUS_TABLE_FULL. */
int q2048_table_import(q2048_slot *table, int cap_log2, const uint64_t *keys, const float *q,
                       int64_t rows, int key_words, uint32_t *status, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* Q2048_H */
