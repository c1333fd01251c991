// tests/hostcheck/hostcheck.cpp -- TEST HARNESS, not product code.
// Instantiates the per-lane arithmetic of 2048_q-learning_amd/csrc/q2048_core.hpp on the host
// (g++) so that tests/test_core_host.py can check the very functions the HIP kernels inline
// against the CPU oracle, exhaustively, in the GPU-less build container.
#include <cstring>

#include "q2048_core5.hpp"

using namespace q2048;

static Board load_board(const uint8_t* p) {
  Board b;
  std::memcpy(&b, p, 16);
  return b;
}
static void store_board(uint8_t* p, const Board& b) { std::memcpy(p, &b, 16); }
static Aux load_aux(const uint8_t* p) {
  Words4 w;
  std::memcpy(&w, p, 16);
  return words_to_aux(w);
}
static void store_aux(uint8_t* p, const Aux& a) {
  Words4 w = aux_to_words(a);
  std::memcpy(p, &w, 16);
}

template <int ENV, class BoardT, class Load, class Store>
static void rollout_profile(uint8_t* boards, uint8_t* aux, int64_t n, int64_t steps, uint64_t seed,
                            uint64_t env_id0, uint32_t ctr0, const uint8_t* actions, float* reward,
                            uint8_t* done, int stride, Load load, Store store) {
  for (int64_t t = 0; t < steps; ++t)
    for (int64_t i = 0; i < n; ++i) {
      BoardT b = load(boards + stride * i);
      Aux a = load_aux(aux + 16 * i);
      const uint64_t id = env_id0 + (uint64_t)i;
      const Draws x = draws(seed, id, ctr0 + (uint32_t)t, kStreamStep);
      Draws y{0u, 0u, 0u, 0u};
      if (ENV & kEnvDqn) y = draws(seed, id, ctr0 + (uint32_t)t, kStreamOver);
      StepOut o = env_step_profile<ENV>(b, a, actions[t * n + i], x.x2, x.x3, y.x0, y.x1);
      reward[t * n + i] = o.reward; done[t * n + i] = o.done;
      if (o.done) begin_episode(b, a, seed, id, (ENV & kEnvResetShaping) != 0);
      store(boards + stride * i, b);
      store_aux(aux + 16 * i, a);
    }
}


extern "C" {

void hc_philox(const uint32_t* c, const uint32_t* k, uint32_t* out) {
  Draws d = philox4x32_10(c[0], c[1], c[2], c[3], k[0], k[1]);
  out[0] = d.x0; out[1] = d.x1; out[2] = d.x2; out[3] = d.x3;
}

void hc_draws(uint64_t seed, uint64_t env_id, uint32_t ctr, uint32_t stream, uint32_t* out) {
  Draws d = draws(seed, env_id, ctr, stream);
  out[0] = d.x0; out[1] = d.x1; out[2] = d.x2; out[3] = d.x3;
}

void hc_move(const uint8_t* in, const uint8_t* actions, int64_t n, uint8_t* out, uint32_t* score,
             uint8_t* moved) {
  for (int64_t i = 0; i < n; ++i) {
    Board b = load_board(in + 16 * i);
    uint32_t s;
    moved[i] = move(b, actions[i], s);
    score[i] = s;
    store_board(out + 16 * i, b);
  }
}

void hc_spawn(const uint8_t* in, const uint32_t* xpos, const uint32_t* xval, int64_t n, uint8_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    Board b = load_board(in + 16 * i);
    spawn(b, xpos[i], xval[i]);
    store_board(out + 16 * i, b);
  }
}

void hc_board_props(const uint8_t* in, int64_t n, uint8_t* over, uint8_t* mx, uint16_t* empties,
                    uint64_t* keys, uint8_t* overflow) {
  for (int64_t i = 0; i < n; ++i) {
    Board b = load_board(in + 16 * i);
    over[i] = game_over(b);
    mx[i] = (uint8_t)max_log2(b);
    empties[i] = (uint16_t)empty_mask(b);
    bool ov;
    keys[i] = pack_key(b, ov);
    overflow[i] = ov;
  }
}

void hc_unpack_keys(const uint64_t* keys, int64_t n, uint8_t* out) {
  for (int64_t i = 0; i < n; ++i) store_board(out + 16 * i, unpack_key(keys[i]));
}

void hc_kth_set_bit(const uint16_t* mask, const uint8_t* k, int64_t n, uint8_t* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = (uint8_t)kth_set_bit16(mask[i], k[i]);
}

void hc_env_step(uint8_t* boards, uint8_t* aux, const uint8_t* actions, const uint32_t* xpos,
                 const uint32_t* xval, int64_t n, float* reward, double* reward64, uint8_t* done,
                 uint8_t* mx, uint8_t* valid, uint32_t* score) {
  for (int64_t i = 0; i < n; ++i) {
    Board b = load_board(boards + 16 * i);
    Aux a = load_aux(aux + 16 * i);
    StepOut o = env_step(b, a, actions[i], xpos[i], xval[i]);
    store_board(boards + 16 * i, b);
    store_aux(aux + 16 * i, a);
    reward[i] = o.reward; reward64[i] = o.reward64; done[i] = o.done; mx[i] = o.max_log2;
    valid[i] = o.valid; score[i] = o.score;
  }
}

void hc_init_envs(uint8_t* boards, uint8_t* aux, int64_t n, uint64_t seed, uint64_t env_id0) {
  for (int64_t i = 0; i < n; ++i) {
    Board b; Aux a;
    init_env(b, a, seed, env_id0 + (uint64_t)i);
    store_board(boards + 16 * i, b);
    store_aux(aux + 16 * i, a);
  }
}

// env-only rollout with given actions [steps, n], auto-reset on done (kernel sequencing)
void hc_rollout_env(uint8_t* boards, uint8_t* aux, int64_t n, int64_t steps, uint64_t seed,
                    uint64_t env_id0, uint32_t ctr0, const uint8_t* actions, float* reward,
                    uint8_t* done) {
  for (int64_t t = 0; t < steps; ++t)
    for (int64_t i = 0; i < n; ++i) {
      Board b = load_board(boards + 16 * i);
      Aux a = load_aux(aux + 16 * i);
      const uint64_t id = env_id0 + (uint64_t)i;
      Draws x = draws(seed, id, ctr0 + (uint32_t)t, kStreamStep);
      StepOut o = env_step(b, a, actions[t * n + i], x.x2, x.x3);
      reward[t * n + i] = o.reward; done[t * n + i] = o.done;
      if (o.done) begin_episode(b, a, seed, id);
      store_board(boards + 16 * i, b);
      store_aux(aux + 16 * i, a);
    }
}

void hc_eps_greedy(const double* eps, const uint32_t* x0, const uint32_t* x1, const float* q,
                   int64_t n, uint8_t* action, uint8_t* explored) {
  for (int64_t i = 0; i < n; ++i) {
    bool e;
    action[i] = (uint8_t)eps_greedy(eps[i], x0[i], x1[i], q[4 * i], q[4 * i + 1], q[4 * i + 2],
                                    q[4 * i + 3], e);
    explored[i] = e;
  }
}

void hc_td(const float* q_sa, const float* reward, const float* q_next, const uint8_t* done,
           const double* lr, const double* gamma, int64_t n, float* out) {
  for (int64_t i = 0; i < n; ++i)
    out[i] = td_value(q_sa[i], reward[i],
                      max4(q_next[4 * i], q_next[4 * i + 1], q_next[4 * i + 2], q_next[4 * i + 3]),
                      done[i] != 0, lr[i], gamma[i]);
}

void hc_luts(double* pow12, double* log2p1, double* stall) {
  for (uint32_t i = 0; i < 32; ++i) {
    pow12[i] = lut_pow12(i); log2p1[i] = lut_log2p1(i); stall[i] = lut_stall(i);
  }
}

void hc_log2_ge1(const double* x, int64_t n, double* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = log2_ge1(x[i]);
}

// ---- 5x5 geometry (q2048_core5.hpp): boards are uint8[25] ----
void hc5_move(const uint8_t* in, const uint8_t* actions, int64_t n, uint8_t* out, uint32_t* score,
              uint8_t* moved) {
  for (int64_t i = 0; i < n; ++i) {
    Board5 b = board5_from_bytes(in + 25 * i);
    uint32_t s;
    moved[i] = move(b, actions[i], s);
    score[i] = s;
    board5_to_bytes(b, out + 25 * i);
  }
}

void hc5_spawn(const uint8_t* in, const uint32_t* xpos, const uint32_t* xval, int64_t n, uint8_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    Board5 b = board5_from_bytes(in + 25 * i);
    spawn(b, xpos[i], xval[i]);
    board5_to_bytes(b, out + 25 * i);
  }
}

void hc5_board_props(const uint8_t* in, int64_t n, uint8_t* over, uint8_t* mx, uint32_t* empties,
                     uint64_t* keys, uint8_t* roundtrip) {
  for (int64_t i = 0; i < n; ++i) {
    Board5 b = board5_from_bytes(in + 25 * i);
    over[i] = game_over(b);
    mx[i] = (uint8_t)max_log2(b);
    empties[i] = empty_mask(b);
    Key5 k = pack_key(b);
    keys[2 * i] = k.k0; keys[2 * i + 1] = k.k1;
    board5_to_bytes(unpack_key(k), roundtrip + 25 * i);
  }
}

void hc5_init_envs(uint8_t* boards, uint8_t* aux, int64_t n, uint64_t seed, uint64_t env_id0) {
  for (int64_t i = 0; i < n; ++i) {
    Board5 b; Aux a;
    init_env(b, a, seed, env_id0 + (uint64_t)i);
    board5_to_bytes(b, boards + 25 * i);
    store_aux(aux + 16 * i, a);
  }
}

void hc5_rollout_env(uint8_t* boards, uint8_t* aux, int64_t n, int64_t steps, uint64_t seed,
                     uint64_t env_id0, uint32_t ctr0, const uint8_t* actions, float* reward,
                     uint8_t* done) {
  for (int64_t t = 0; t < steps; ++t)
    for (int64_t i = 0; i < n; ++i) {
      Board5 b = board5_from_bytes(boards + 25 * i);
      Aux a = load_aux(aux + 16 * i);
      const uint64_t id = env_id0 + (uint64_t)i;
      Draws x = draws(seed, id, ctr0 + (uint32_t)t, kStreamStep);
      StepOut o = env_step(b, a, actions[t * n + i], x.x2, x.x3);
      reward[t * n + i] = o.reward; done[t * n + i] = o.done;
      if (o.done) begin_episode(b, a, seed, id);
      board5_to_bytes(b, boards + 25 * i);
      store_aux(aux + 16 * i, a);
    }
}

// env profiles: the DQN path's step (draws4 = pos, val, over_pos, over_val per env) and rollouts
// with a profile (ENV bits of q2048_core.hpp: 1 = DQN step, 2 = resets restore the shaping state)
void hc_env_step_dqn(uint8_t* boards, uint8_t* aux, const uint8_t* actions, const uint32_t* draws4,
                     int64_t n, int side, float* reward, uint8_t* done, uint8_t* mx, uint8_t* valid) {
  for (int64_t i = 0; i < n; ++i) {
    Aux a = load_aux(aux + 16 * i);
    const uint32_t* d = draws4 + 4 * i;
    StepOut o;
    if (side == 4) {
      Board b = load_board(boards + 16 * i);
      o = env_step_dqn(b, a, actions[i], d[0], d[1], d[2], d[3]);
      store_board(boards + 16 * i, b);
    } else {
      Board5 b = board5_from_bytes(boards + 25 * i);
      o = env_step_dqn(b, a, actions[i], d[0], d[1], d[2], d[3]);
      board5_to_bytes(b, boards + 25 * i);
    }
    store_aux(aux + 16 * i, a);
    reward[i] = o.reward; done[i] = o.done; mx[i] = o.max_log2; valid[i] = o.valid;
  }
}

void hc_rollout_env_profile(uint8_t* boards, uint8_t* aux, int64_t n, int side, int env,
                            int64_t steps, uint64_t seed, uint64_t env_id0, uint32_t ctr0,
                            const uint8_t* actions, float* reward, uint8_t* done) {
  auto l4 = [](const uint8_t* p) { return load_board(p); };
  auto s4 = [](uint8_t* p, const Board& b) { store_board(p, b); };
  auto l5 = [](const uint8_t* p) { return board5_from_bytes(p); };
  auto s5 = [](uint8_t* p, const Board5& b) { board5_to_bytes(b, p); };
#define HC_CASE(E)                                                                                  \
  case E:                                                                                           \
    if (side == 4) rollout_profile<E, Board>(boards, aux, n, steps, seed, env_id0, ctr0, actions,   \
                                             reward, done, 16, l4, s4);                             \
    else rollout_profile<E, Board5>(boards, aux, n, steps, seed, env_id0, ctr0, actions, reward,    \
                                    done, 25, l5, s5);                                              \
    break;
  switch (env & 3) { HC_CASE(0) HC_CASE(1) HC_CASE(2) HC_CASE(3) }
#undef HC_CASE
}

void hc_kth_set_bit32(const uint32_t* mask, const uint8_t* k, int64_t n, uint8_t* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = (uint8_t)kth_set_bit32(mask[i], k[i]);
}

uint64_t hc_mix64(uint64_t x) { return mix64(x); }
uint64_t hc_lane_salt(uint64_t id) { return lane_salt(id); }
int hc_sizeof_aux(void) { return (int)sizeof(Aux); }

}  // extern "C"
