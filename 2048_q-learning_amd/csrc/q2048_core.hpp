// q2048_core.hpp -- per-lane arithmetic of the batched 2048 Q-learning step (4x4 board).
//
// One board lives in one lane: 16 x uint8 log2 tiles = four 32-bit row words in VGPRs
// (word r = row r, byte c = column c; 0 = empty, k = tile 2^k).  Everything here is
// register-only integer/fp arithmetic with no memory access, written once and compiled
//   * by hipcc for gfx950 (the kernels in q2048_kernels.hip), and
//   * by g++ for tests/hostcheck, which checks these exact functions against the CPU oracle
//     exhaustively before anything runs on a GPU.
// It is NOT a CPU fallback: nothing in the product calls the host instantiation.
//
// What each function replaces in the reference (paths under QLearningBase/):
//   move()            environment/Game2048_env.py:22-63  (rotate/move_left/rotate back)
//   spawn()           :16-20   add_number
//   game_over()       :65-75   closed form of the four trial moves
//   env_step()        :97-129  step + calculate_reward :136-184 + update_and_normalize :197-205
//   reset_board()     :11-14, :187-191
//   pack_key()        Agent/main.py:82,94  tuple(map(tuple, state))
//   eps_greedy()      Agent/main.py:34-38
//   td_value()        Agent/main.py:40-43
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define Q_HD __host__ __device__ __forceinline__
#else
#define Q_HD inline
#endif

#include "q2048_luts.inc"

namespace q2048 {

// ------------------------------------------------------------------------------------------
// counter RNG: Philox4x32-10, counter = (env_id lo, env_id hi, step counter, stream),
// key = (seed lo, seed hi).  One call serves one env step:
//   x[0] epsilon test, x[1] random action, x[2] spawn cell, x[3] spawn value.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kStreamStep = 0u, kStreamReset = 1u, kStreamOver = 2u;

struct Draws { uint32_t x0, x1, x2, x3; };

// (Q2048_PHILOX_ROUNDS: measurement builds only -- tools/archive/sessions/r04_philox7.sh times the rollouts with
// Philox4x32-7, the smallest variant that passes BigCrush, against the 10 rounds of the draw contract.)
#ifndef Q2048_PHILOX_ROUNDS
#define Q2048_PHILOX_ROUNDS 10
#endif
Q_HD Draws philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                         uint32_t k1) {
#pragma unroll
  for (int r = 0; r < Q2048_PHILOX_ROUNDS; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return Draws{c0, c1, c2, c3};
}

Q_HD Draws draws(uint64_t seed, uint64_t env_id, uint32_t ctr, uint32_t stream) {
  return philox4x32_10((uint32_t)env_id, (uint32_t)(env_id >> 32), ctr, stream, (uint32_t)seed,
                       (uint32_t)(seed >> 32));
}

// draw -> decision (DESIGN.md "draw contract"; the reference call each replaces is cited there)
Q_HD double draw_uniform(uint32_t x) { return (double)x * (1.0 / 4294967296.0); }
Q_HD int draw_action(uint32_t x) { return (int)(x >> 30); }
Q_HD uint32_t draw_index(uint32_t x, uint32_t n) { return (uint32_t)(((uint64_t)x * n) >> 32); }
// not (random() < 0.9) (Game2048_env.py:20): x / 2^32 < 0.9 (the double nearest 0.9) holds exactly for
// x <= 3865470566 (0.9 * 2^32 = 3865470566.4), so the test is an integer compare
Q_HD bool draw_is_four(uint32_t x) { return x >= 3865470567u; }

// ------------------------------------------------------------------------------------------
// SWAR helpers on four tile bytes per word.  Every tile byte is <= 0x7f (log2 <= 17 on 4x4).
// ------------------------------------------------------------------------------------------
Q_HD uint32_t nz80(uint32_t x) { return (x + 0x7f7f7f7fu) & 0x80808080u; }       // 0x80 per byte != 0
Q_HD uint32_t z80(uint32_t x) { return ~(x + 0x7f7f7f7fu) & 0x80808080u; }       // 0x80 per byte == 0
Q_HD uint32_t fill80(uint32_t m) { return (m << 1) - (m >> 7); }                 // 0x80 -> 0xff
// 0x80 -> 0x7f: as good a select mask as 0xff between words whose bytes are all <= 0x7f (bit 7 is
// 0 on both sides), one instruction less
Q_HD uint32_t fill7f(uint32_t m) { return m - (m >> 7); }
Q_HD uint32_t bsel(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }

struct Board { uint32_t r0, r1, r2, r3; };

Q_HD bool operator==(const Board& a, const Board& b) {
  return ((a.r0 ^ b.r0) | (a.r1 ^ b.r1) | (a.r2 ^ b.r2) | (a.r3 ^ b.r3)) == 0;
}
Q_HD void clear(Board& b) { b = Board{0u, 0u, 0u, 0u}; }

// 4x4 byte transpose: out word j = column j (byte i = row i)
Q_HD Board transpose(const Board& b) {
  const uint32_t a0 = (b.r0 & 0x00ff00ffu) | ((b.r1 & 0x00ff00ffu) << 8);
  const uint32_t a1 = ((b.r0 >> 8) & 0x00ff00ffu) | (b.r1 & 0xff00ff00u);
  const uint32_t b0 = (b.r2 & 0x00ff00ffu) | ((b.r3 & 0x00ff00ffu) << 8);
  const uint32_t b1 = ((b.r2 >> 8) & 0x00ff00ffu) | (b.r3 & 0xff00ff00u);
  return Board{(a0 & 0xffffu) | (b0 << 16), (a1 & 0xffffu) | (b1 << 16),
               (a0 >> 16) | (b0 & 0xffff0000u), (a1 >> 16) | (b1 & 0xffff0000u)};
}

// sum over the four bytes b of (b ? 2^b : 0); merged tiles are >= 2 so bit 0 never counts
Q_HD uint32_t pow2_sum(uint32_t v) {
  return (((1u << (v & 0xffu)) & ~1u) + ((1u << ((v >> 8) & 0xffu)) & ~1u)) +
         (((1u << ((v >> 16) & 0xffu)) & ~1u) + ((1u << (v >> 24)) & ~1u));
}

// Slide/merge four lines at once.  c_j holds cell j (counted from the edge the tiles move
// to) of each of the four lines, one line per byte.  Restates move_left
// (Game2048_env.py:25-44): compress, then one left-to-right merge pass without cascade.
Q_HD uint32_t slide_lines(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3) {
  uint32_t z;
  // compress (:26): pull cells towards c0 while the head cell is empty
#pragma unroll
  for (int rep = 0; rep < 3; ++rep) {
    z = fill7f(z80(c0));
    c0 = bsel(z, c1, c0); c1 = bsel(z, c2, c1); c2 = bsel(z, c3, c2); c3 &= ~z;
  }
#pragma unroll
  for (int rep = 0; rep < 2; ++rep) {
    z = fill7f(z80(c1));
    c1 = bsel(z, c2, c1); c2 = bsel(z, c3, c2); c3 &= ~z;
  }
  z = fill7f(z80(c2));
  c2 = bsel(z, c3, c2); c3 &= ~z;
  // merge (:29-40): pair (0,1), then the next unmerged pair, never re-merging a result
  uint32_t e = z80(c0 ^ c1) & nz80(c0);
  uint32_t m = fill7f(e);
  c0 += e >> 7;
  const uint32_t s_a = c0 & m;
  c1 = bsel(m, c2, c1); c2 = bsel(m, c3, c2); c3 &= ~m;
  e = z80(c1 ^ c2) & nz80(c1);
  m = fill7f(e);
  c1 += e >> 7;
  const uint32_t s_b = c1 & m;
  c2 = bsel(m, c3, c2); c3 &= ~m;
  e = z80(c2 ^ c3) & nz80(c2);
  m = fill7f(e);
  c2 += e >> 7;
  c3 &= ~m;
  const uint32_t s_c = c2 & m;  // a line that merged at (2,3) merged nowhere else
  return pow2_sum(s_a | s_c) + pow2_sum(s_b);  // score += merged value (:36)
}

// Game2048.move without the spawn (:51-60).  action: 0 left, 1 up, 2 right, 3 down (:54).
// Direction-agnostic: pick the four "cell j of every line" words by select, no branch.
Q_HD bool move(Board& b, int action, uint32_t& score) {
  const bool horiz = (action & 1) == 0;  // lines are rows: take columns as cell words
  const bool rev = (action & 2) != 0;    // right / down: cell 0 is the far edge
  const Board t = transpose(b);
  const uint32_t p0 = horiz ? t.r0 : b.r0, p1 = horiz ? t.r1 : b.r1;
  const uint32_t p2 = horiz ? t.r2 : b.r2, p3 = horiz ? t.r3 : b.r3;
  uint32_t c0 = rev ? p3 : p0, c1 = rev ? p2 : p1, c2 = rev ? p1 : p2, c3 = rev ? p0 : p3;
  const uint32_t o0 = c0, o1 = c1, o2 = c2, o3 = c3;
  score = slide_lines(c0, c1, c2, c3);
  const bool moved = ((c0 ^ o0) | (c1 ^ o1) | (c2 ^ o2) | (c3 ^ o3)) != 0;  // :38,42-43
  const Board q{rev ? c3 : c0, rev ? c2 : c1, rev ? c1 : c2, rev ? c0 : c3};
  const Board qt = transpose(q);
  b.r0 = horiz ? qt.r0 : q.r0; b.r1 = horiz ? qt.r1 : q.r1;
  b.r2 = horiz ? qt.r2 : q.r2; b.r3 = horiz ? qt.r3 : q.r3;
  return moved;
}

// 16-bit mask of empty cells, bit = row-major cell index (np.where order, :17)
Q_HD uint32_t movemask(uint32_t m80) { return (((m80 >> 7) * 0x01020408u) >> 24) & 0xfu; }
Q_HD uint32_t empty_mask(const Board& b) {
  return movemask(z80(b.r0)) | (movemask(z80(b.r1)) << 4) | (movemask(z80(b.r2)) << 8) |
         (movemask(z80(b.r3)) << 12);
}

Q_HD uint32_t popc(uint32_t x) { return (uint32_t)__builtin_popcount(x); }

// index of the k-th (0-based) set bit of a 16-bit mask, k < popcount(mask)
Q_HD uint32_t kth_set_bit16(uint32_t mask, uint32_t k) {
  uint32_t pos = 0, c;
  bool ge;
  c = popc(mask & 0xffu); ge = k >= c; k -= ge ? c : 0u; pos += ge ? 8u : 0u; mask >>= ge ? 8u : 0u;
  c = popc(mask & 0xfu);  ge = k >= c; k -= ge ? c : 0u; pos += ge ? 4u : 0u; mask >>= ge ? 4u : 0u;
  c = popc(mask & 0x3u);  ge = k >= c; k -= ge ? c : 0u; pos += ge ? 2u : 0u; mask >>= ge ? 2u : 0u;
  c = mask & 1u;          ge = k >= c; pos += ge ? 1u : 0u;
  return pos;
}

Q_HD void put_cell(Board& b, uint32_t pos, uint32_t v) {
  const uint32_t w = v << ((pos & 3u) * 8u), row = pos >> 2;
  b.r0 |= row == 0u ? w : 0u; b.r1 |= row == 1u ? w : 0u;
  b.r2 |= row == 2u ? w : 0u; b.r3 |= row == 3u ? w : 0u;
}

// Game2048.add_number (:16-20): k-th empty cell in row-major order gets 2 (p=.9) or 4
Q_HD void spawn(Board& b, uint32_t draw_pos, uint32_t draw_val) {
  const uint32_t em = empty_mask(b), n = popc(em);
  if (n == 0u) return;  // :18
  put_cell(b, kth_set_bit16(em, draw_index(draw_pos, n)), draw_is_four(draw_val) ? 2u : 1u);
}

// Game2048.is_game_over (:65-75) in closed form: full board and no equal neighbours.  On a
// full board a trial move changes something iff some line holds an equal adjacent pair.
Q_HD bool game_over(const Board& b) {
  const uint32_t empties = z80(b.r0) | z80(b.r1) | z80(b.r2) | z80(b.r3);
  const uint32_t vert = z80(b.r0 ^ b.r1) | z80(b.r1 ^ b.r2) | z80(b.r2 ^ b.r3);
  const uint32_t horz = (z80(b.r0 ^ (b.r0 >> 8)) | z80(b.r1 ^ (b.r1 >> 8)) |
                         z80(b.r2 ^ (b.r2 >> 8)) | z80(b.r3 ^ (b.r3 >> 8))) & 0x00808080u;
  return (empties | vert | horz) == 0u;
}

Q_HD uint32_t bytemax(uint32_t a, uint32_t b) {  // per-byte max, bytes <= 0x7f
  const uint32_t m = fill7f(((a | 0x80808080u) - b) & 0x80808080u);
  return bsel(m, a, b);
}
Q_HD uint32_t max_log2(const Board& b) {  // np.max(board) (:100), as log2
  uint32_t x = bytemax(bytemax(b.r0, b.r1), bytemax(b.r2, b.r3));
  x = bytemax(x, x >> 16);
  x = bytemax(x, x >> 8);
  return x & 0xffu;
}

// ------------------------------------------------------------------------------------------
// per-env shaping state (Game2048_env.__init__, :81-95) -- 16 bytes, one dwordx4 per lane
// ------------------------------------------------------------------------------------------
struct Aux {
  int32_t score;        // env.score (:84), reset per episode (:190)
  float ep_return;      // total_reward of Agent/main.py:84,101
  uint8_t prev_max;     // log2(previous_max) (:87), survives reset()
  uint8_t cons_action;  // consecutive_action (:92), 0xFF = None, survives reset()
  uint16_t cons_count;  // consecutive_count (:93), saturating, survives reset()
  uint32_t episode;     // resets so far = counter of the reset draws
};
constexpr uint8_t kNoAction = 0xFFu;
constexpr uint32_t kConsCountSat = 60000u;  // every count > 100 behaves alike (:121-125)

Q_HD Aux aux_init() { return Aux{0, 0.0f, 1, kNoAction, 0, 0u}; }

Q_HD double lut_pow12(uint32_t L) { constexpr double t[32] = Q2048_POW12; return t[L & 31u]; }
Q_HD double lut_log2p1(uint32_t L) { constexpr double t[32] = Q2048_LOG2P1; return t[L & 31u]; }
Q_HD double lut_stall(uint32_t k) { constexpr double t[32] = Q2048_STALL; return t[k < 31u ? k : 31u]; }

Q_HD uint64_t f64_bits(double d) { union { double d; uint64_t u; } c; c.d = d; return c.u; }
Q_HD double bits_f64(uint64_t u) { union { double d; uint64_t u; } c; c.u = u; return c.d; }

// log2(x) for finite x >= 1, relative error < 2^-46 (checked against libm in
// tests/test_core_host.py): ~20 instructions instead of the math library's ~110.  The reward is
// stored as float32, so this is >20 bits more than the rounding it feeds.
//   x = m * 2^e, m in [1, 2);  j = top 6 mantissa bits;  r = m * inv[j] - 1, |r| <= 2^-7;
//   log2(x) = e + lg[j] + r * (a1 + r * (a2 + ... + r * a6)),  lg[j] = -log2(inv[j]) exactly.
Q_HD double log2_ge1(double x) {
  constexpr double inv[64] = Q2048_LOG2_INV;
  constexpr double lg[64] = Q2048_LOG2_LG;
  constexpr double a[6] = Q2048_LOG2_COEF;
  const uint64_t u = f64_bits(x);
  const int e = (int)(u >> 52) - 1023;
  const uint32_t j = (uint32_t)(u >> 46) & 63u;
  const double m = bits_f64((u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
  const double r = fma(m, inv[j], -1.0);
  double p = fma(r, a[5], a[4]);
  p = fma(r, p, a[3]);
  p = fma(r, p, a[2]);
  p = fma(r, p, a[1]);
  p = fma(r, p, a[0]);
  return fma(r, p, (double)e + lg[j]);
}

// update_and_normalize (:197-205); both log2 arguments are >= 1
// One logarithm serves both signs: |r - 1| for r < 0 is |r| + 1, the same double (rounding is
// symmetric), and r + 1 is |r| + 1 for r >= 0 (-0.0 included).
Q_HD double normalize_reward(double r) {
  const double v = fmin(log2_ge1(fabs(r) + 1.0), 10.0);
  return r >= 0 ? v : -v;
}

// calculate_reward (:136-184); L = log2(max tile), prev = log2(previous_max), both integers
Q_HD double calculate_reward(uint32_t score, bool valid, bool over, uint32_t L, uint8_t& prev) {
  L = L < 1u ? 1u : L;                                   // :141
  const double cl = (double)L;                           // :144
  double bonus = 0.0, r;
  if (L > prev) {                                        // :148
    bonus = (cl - (double)prev) * lut_pow12(L);          // :149
    prev = (uint8_t)L;                                   // :150
  }
  if (!valid) {
    if (over) {
      if (L >= 9u && L <= 11u) r = bonus + lut_pow12(L); // :156-158
      else r = 0.0 - lut_log2p1(L);                      // :160
    } else {
      r = 0.0 - 0.1 * cl;                                // :164
    }
  } else {
    r = (double)score;                                   // :168
    if (bonus > 0) r += bonus;                           // :171-172
    else r += cl * 0.05;                                 // :173-174
    if (L >= 9u) r += lut_pow12(L) * 2;                  // :176-177
  }
  return normalize_reward(r);                            // :181
}

struct StepOut {
  float reward;      // reward rounded to f32 (the TD update consumes exactly this value)
  double reward64;   // before rounding, for parity reports
  uint32_t score;    // merge score of this move
  uint8_t done, max_log2, valid;
};

// the spawn of a valid move (Game2048_env.py:61-62), then is_game_over (:99); the 5x5 board has its
// own version that counts the empty cells once for both
template <class BoardT>
Q_HD bool spawn_then_over(BoardT& b, bool valid, uint32_t x_pos, uint32_t x_val) {
  if (valid) spawn(b, x_pos, x_val);
  return game_over(b);
}

// Game2048_env.step (:97-129) for one lane.  x_pos/x_val are the spawn draws.  BoardT is the
// 4x4 Board above or the 5x5 Board5 of q2048_core5.hpp (same functions, overloaded).
template <class BoardT>
Q_HD StepOut env_step(BoardT& b, Aux& a, int action, uint32_t x_pos, uint32_t x_val) {
  StepOut o;
  uint32_t score;
  const bool valid = move(b, action, score);                            // :98
  const bool over = spawn_then_over(b, valid, x_pos, x_val);            // :61-62, :99
  const uint32_t mx = max_log2(b);                                      // :100
  a.score += (int32_t)score;                                            // :104
  double r = calculate_reward(score, valid, over, mx, a.prev_max);      // :107
  uint32_t cnt = a.cons_count;
  if ((uint8_t)action == a.cons_action) {                               // :110
    cnt = cnt + 1u < kConsCountSat ? cnt + 1u : kConsCountSat;          // :111
  } else {
    a.cons_action = (uint8_t)action;                                    // :113
    cnt = 1u;                                                           // :114-115
  }
  a.cons_count = (uint16_t)cnt;
  bool done = !valid && over;                                           // :117-118
  if (cnt > 10u) {                                                      // :121
    if (cnt > 100u) done = true;                                        // :122-123
    r += lut_stall(cnt - 10u);                                          // :124-127
  }
  o.reward64 = r;
  o.reward = (float)r;
  a.ep_return += o.reward;
  o.score = score;
  o.done = done;
  o.max_log2 = (uint8_t)mx;
  o.valid = valid;
  return o;
}

// Env profiles (template parameter ENV of the kernels, from Q2048_FLAG_ENV_DQN / _RESET_SHAPING / _PLAY_ONLY):
//   kEnvDqn           step of the DQN path's env, Deep_QLearning/environment/
//                     Game2048_nopenalty_env.py:106-138, instead of Game2048_env.step
//   kEnvResetShaping  resets also restore previous_max and the consecutive-action state
//                     (SURVEY 7.8 opt-in; Game2048_env.reset, :187-191, leaves them alone)
//   kEnvPlayOnly      (fused rollout only) no learner: the table is neither read nor written
constexpr int kEnvDqn = 1, kEnvResetShaping = 2, kEnvPlayOnly = 4;

// Game2048_nopenalty_env.step (:106-120) with the caller's `env.game.board = next_state`
// (mainDQL_CNN_step2.py:237) folded in: on return b is moved_board.
//   - move works on a copy (:58); is_game_over (:68-78) inspects the board from BEFORE the move:
//     False if it has an empty cell (:70-71); on a full board it makes real moves 0..3 (:72-74) and
//     the first that changes something leaves ITS result, spawn included (draws y_pos, y_val), in
//     moved_board (:75-77) -- that, not the chosen action's move, is what the step returns;
//   - reward = calculate_reward2 (:122-138): -10 for an invalid move while not over, else the
//     chosen move's merge score; done = game_over (:117-118); max tile of moved_board (:109).
template <class BoardT>
Q_HD StepOut env_step_dqn(BoardT& b, Aux& a, int action, uint32_t x_pos, uint32_t x_val,
                          uint32_t y_pos, uint32_t y_val) {
  StepOut o;
  const BoardT pre = b;
  uint32_t score;
  const bool valid = move(b, action, score);                            // :107 -> :53-63
  if (valid) spawn(b, x_pos, x_val);                                    // :64-65
  bool over = false;
  if (empty_mask(pre) == 0u) {                                          // :70
    over = true;
    for (int act = 0; act < 4 && over; ++act) {                         // :72
      BoardT t = pre;                                                   // :58
      uint32_t s2;
      if (move(t, act, s2)) {                                           // :74
        spawn(t, y_pos, y_val);                                         // :64-65
        b = t;
        over = false;                                                   // :75-77
      }
    }
    if (over) b = pre;                      // nothing moved: moved_board is a copy of the board
  }
  const uint32_t mx = max_log2(b);                                      // :109
  a.score += (int32_t)score;                                            // :112
  const double r = (!valid && !over) ? -10.0 : (double)score;           // :125-128
  o.reward64 = r;
  o.reward = (float)r;
  a.ep_return += o.reward;
  o.score = score;
  o.done = over;                                                        // :117-118
  o.max_log2 = (uint8_t)mx;
  o.valid = valid;
  return o;
}

// the step of profile ENV; (y_pos, y_val) are only used by kEnvDqn
template <int ENV, class BoardT>
Q_HD StepOut env_step_profile(BoardT& b, Aux& a, int action, uint32_t x_pos, uint32_t x_val,
                              uint32_t y_pos, uint32_t y_val) {
  if constexpr ((ENV & kEnvDqn) != 0) return env_step_dqn(b, a, action, x_pos, x_val, y_pos, y_val);
  else return env_step(b, a, action, x_pos, x_val);
}

// Game2048.__init__ (:11-14) / Game2048_env.reset (:187-191): empty board, two spawns,
// score = 0.  previous_max and the consecutive-action state are NOT reset (:187-191).
template <class BoardT>
Q_HD void reset_board(BoardT& b, Aux& a, const Draws& d) {
  clear(b);
  spawn(b, d.x0, d.x1);
  spawn(b, d.x2, d.x3);
  a.score = 0;
  a.ep_return = 0.0f;
}

// env construction (Game2048_env.__init__, :81-95) and the reset that follows a finished
// episode (Agent/main.py:81): the spawn draws are keyed by (env id, episode index)
template <class BoardT>
Q_HD void init_env(BoardT& b, Aux& a, uint64_t seed, uint64_t env_id) {
  a = aux_init();
  reset_board(b, a, draws(seed, env_id, 0u, kStreamReset));
}
template <class BoardT>
Q_HD void begin_episode(BoardT& b, Aux& a, uint64_t seed, uint64_t env_id,
                        bool reset_shaping = false) {
  a.episode += 1u;
  reset_board(b, a, draws(seed, env_id, a.episode, kStreamReset));
  if (reset_shaping) {            // what Game2048_env.__init__ sets (:87, :92-93)
    a.prev_max = 1;
    a.cons_action = kNoAction;
    a.cons_count = 0;
  }
}

// register <-> memory images (little endian): board = 16 bytes, aux = 16 bytes
struct Words4 { uint32_t w0, w1, w2, w3; };
Q_HD uint32_t f32_bits(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
Q_HD float bits_f32(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }
Q_HD Words4 aux_to_words(const Aux& a) {
  return Words4{(uint32_t)a.score, f32_bits(a.ep_return),
                (uint32_t)a.prev_max | ((uint32_t)a.cons_action << 8) | ((uint32_t)a.cons_count << 16),
                a.episode};
}
Q_HD Aux words_to_aux(const Words4& w) {
  return Aux{(int32_t)w.w0, bits_f32(w.w1), (uint8_t)(w.w2 & 0xffu), (uint8_t)((w.w2 >> 8) & 0xffu),
             (uint16_t)(w.w2 >> 16), w.w3};
}

// ------------------------------------------------------------------------------------------
// agent arithmetic
// ------------------------------------------------------------------------------------------
// state key (Agent/main.py:82): 16 log2 nibbles, cell 0 in the low nibble.  Tiles above 2^15
// do not fit a nibble; `overflow` reports them (the low nibble is used, states alias).
Q_HD uint32_t pack_row(uint32_t x) {
  x &= 0x0f0f0f0fu;
  x = (x | (x >> 4)) & 0x00ff00ffu;
  return (x | (x >> 8)) & 0xffffu;
}
Q_HD uint64_t pack_key(const Board& b, bool& overflow) {
  overflow = ((b.r0 | b.r1 | b.r2 | b.r3) & 0xf0f0f0f0u) != 0u;
  return (uint64_t)(pack_row(b.r0) | (pack_row(b.r1) << 16)) |
         ((uint64_t)(pack_row(b.r2) | (pack_row(b.r3) << 16)) << 32);
}
Q_HD uint32_t unpack_row(uint32_t k) {
  k &= 0xffffu;
  k = (k | (k << 8)) & 0x00ff00ffu;
  return (k | (k << 4)) & 0x0f0f0f0fu;
}
Q_HD Board unpack_key(uint64_t key) {
  return Board{unpack_row((uint32_t)key), unpack_row((uint32_t)(key >> 16)),
               unpack_row((uint32_t)(key >> 32)), unpack_row((uint32_t)(key >> 48))};
}

Q_HD uint64_t mix64(uint64_t h) {
  h *= 0x9E3779B97F4A7C15ull;
  h ^= h >> 29;
  h *= 0xBF58476D1CE4E5B9ull;
  h ^= h >> 32;
  return h;
}
// independent-learners mode: every env owns private rows of the shared table
Q_HD uint64_t lane_salt(uint64_t env_id) { return mix64(env_id + 0x2048ull) | 1ull; }

// np.argmax: first maximum (Agent/main.py:38,41)
Q_HD int argmax4(float q0, float q1, float q2, float q3) {
  int b = 0;
  float m = q0;
  if (q1 > m) { m = q1; b = 1; }
  if (q2 > m) { m = q2; b = 2; }
  if (q3 > m) { m = q3; b = 3; }
  return b;
}
Q_HD float max4(float q0, float q1, float q2, float q3) { return fmaxf(fmaxf(q0, q1), fmaxf(q2, q3)); }

// choose_action (Agent/main.py:34-38)
Q_HD int eps_greedy(double eps, uint32_t x_eps, uint32_t x_act, float q0, float q1, float q2,
                    float q3, bool& explored) {
  explored = draw_uniform(x_eps) < eps;                       // :35
  return explored ? draw_action(x_act) : argmax4(q0, q1, q2, q3);  // :36 / :38
}

// update_q_value (Agent/main.py:41-43) in the reference's own double arithmetic: no fused
// multiply-add (CPython rounds the product and the sum separately, and so does the oracle), so the
// device's doubles equal the oracle's bit for bit and only the final float32 store differs from the
// reference's float64 dict.
#if defined(__clang__)
#define Q2048_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define Q2048_NO_CONTRACT
#endif
Q_HD double td_target(float reward, float max_q_next, bool done, double gamma) {
  Q2048_NO_CONTRACT
  const double bootstrap = gamma * (double)max_q_next;
  return (double)reward + bootstrap * (done ? 0.0 : 1.0);                                     // :42
}
Q_HD double td_fold(double q, double target, double lr) {
  Q2048_NO_CONTRACT
  const double step = lr * (target - q);
  return q + step;                                                                            // :43
}
// returns the new Q[s][a] given the current one
Q_HD float td_value(float q_sa, float reward, float max_q_next, bool done, double lr,
                    double gamma) {
  return (float)td_fold((double)q_sa, td_target(reward, max_q_next, done, gamma), lr);
}

// ------------------------------------------------------------------------------------------
// row-tuple linear Q (BASELINE configs[1]: "flat-array Q over row-tuple features").  NOT the
// reference's learner (its Q is keyed by the whole board, Agent/main.py:82): Q(s,a) is the sum
// over the four rows r of W[r][idx_r(s)][a] with idx_r = pack_row(row r); same epsilon-greedy
// and TD target as Agent/main.py:34-43, the error spread evenly over the four weights.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kRtIdx = 65536u;  // entries per row table; W is float[4][65536][4]
Q_HD float rt_sum(float e0, float e1, float e2, float e3) { return (e0 + e1) + (e2 + e3); }
Q_HD float rt_delta(float q_sa, float reward, float max_next, bool done, double lr, double gamma) {
  const double target = (double)reward + (gamma * (double)max_next * (done ? 0.0 : 1.0));
  return (float)((lr * 0.25) * (target - (double)q_sa));
}

}  // namespace q2048
