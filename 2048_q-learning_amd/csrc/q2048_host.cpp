// q2048_host.cpp -- libq2048_host.so: the CPU twin of libq2048_hip.so.
//
// The SAME C ABI (include/q2048.h: same names, argument meaning, error codes) on HOST memory, built from
// the same per-lane arithmetic the HIP kernels inline (q2048_core.hpp / q2048_core5.hpp: SWAR slide, spawn,
// closed-form game-over, reward, Philox draws, epsilon-greedy, TD fold) and the same table (32-byte slots,
// the same hash, the same bucketised probe sequence): a table trained here IS a device table, byte for
// byte, and the other way round.  Envs are split into contiguous ranges over std::threads (one thread for
// small batches); rows are claimed with a compare-and-swap on the key word and Q values written with
// 4-byte stores, as on the device -- so B = 1 (and private rows, and the deterministic step at any B) is
// the reference's sequential loop (Agent/main.py:80-109), and a shared table with several threads is the
// same Hogwild learner the fused kernel is.
//
// It is an explicit DEVICE ("cpu": `train.py --device cpu`, `BatchedGame2048Env(device="cpu")`), never a
// fallback: the package loads it only when asked for by name, and nothing of oracle/ is linked or called.
// `stream` arguments are ignored (every call is complete when it returns).  Not here: the chunked device
// allocator (q2048_table_alloc / _reserve / _grow*: Q2048_ERR_UNSUPPORTED -- host tables are the caller's
// plain memory) and the row cache as an optimisation (rows that exist are read from the table every time; the
// records the device would leave are written as EMPTY ones).  The one thing the cache carries that is not an
// optimisation IS here: the visit row of a state without a row under Q2048_FLAG_NO_NEW_ROWS (a ROWLESS record).
// Threads: Q2048_HOST_THREADS (default: the hardware's, at most one per 2048 envs), read at every call.
//
//   g++ -O3 -std=c++17 -fPIC -shared -pthread -I include -I 2048_q-learning_amd/csrc \
//       -o 2048_q-learning_amd/csrc/libq2048_host.so 2048_q-learning_amd/csrc/q2048_host.cpp
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <unordered_map>
#include <vector>

#include "q2048.h"
#include "q2048_core5.hpp"

namespace {
using namespace q2048;
using u64 = unsigned long long;

// as the device: the limits that make a probe of a FULL table end (bulk moves and lookups / the learning paths)
constexpr uint32_t kMaxProbe = 1u << 14, kRolloutProbe = 1u << 10;
constexpr int kMaxCas = 16;
constexpr int kMaxAwait = 1 << 20;

static_assert(sizeof(q2048_aux) == 16 && sizeof(q2048_slot) == 32 && sizeof(q2048_episode) == 48, "ABI layout");
static_assert(sizeof(Aux) == sizeof(q2048_aux), "core/ABI aux mismatch");

// ---- geometry -------------------------------------------------------------------------------------------------
template <int N> struct Geo;
template <> struct Geo<4> { using BoardT = Board; struct Key { u64 k0; }; static constexpr int kCells = 16; };
template <> struct Geo<5> { using BoardT = Board5; struct Key { u64 k0, k1; }; static constexpr int kCells = 25; };

inline void load_board(const uint8_t* boards, int64_t i, Board& b) { std::memcpy(&b, boards + 16 * i, 16); }
inline void load_board(const uint8_t* boards, int64_t i, Board5& b) { b = board5_from_bytes(boards + 25 * i); }
inline void store_board(uint8_t* boards, int64_t i, const Board& b) { std::memcpy(boards + 16 * i, &b, 16); }
inline void store_board(uint8_t* boards, int64_t i, const Board5& b) { board5_to_bytes(b, boards + 25 * i); }
inline Aux ld_aux(const q2048_aux* aux, int64_t i) {
  Words4 w;
  std::memcpy(&w, aux + i, 16);
  return words_to_aux(w);
}
inline void st_aux(q2048_aux* aux, int64_t i, const Aux& a) {
  const Words4 w = aux_to_words(a);
  std::memcpy(aux + i, &w, 16);
}
inline void status_or(uint32_t* status, uint32_t bits) { __atomic_fetch_or(status, bits, __ATOMIC_RELAXED); }

inline Geo<4>::Key state_key(const Board& b, u64 salt, uint32_t* status) {
  bool ov;
  u64 k = pack_key(b, ov) ^ salt;
  if (ov) status_or(status, Q2048_STATUS_TILE_OVERFLOW);
  return Geo<4>::Key{k == 0ull ? 1ull : k};
}
inline Geo<5>::Key state_key(const Board5& b, u64 salt, uint32_t*) {
  const Key5 k = pack_key(b);
  return Geo<5>::Key{k.k0 ^ (salt & 0x7fffffffffffffffull), k.k1 ^ (mix64(salt) & 0x3fffffffffffffffull)};
}
inline bool key_eq(const Geo<4>::Key& a, const Geo<4>::Key& b) { return a.k0 == b.k0; }
inline bool key_eq(const Geo<5>::Key& a, const Geo<5>::Key& b) { return a.k0 == b.k0 && a.k1 == b.k1; }
inline u64 key_hash(const Geo<4>::Key& k) { return mix64(k.k0); }
inline u64 key_hash(const Geo<5>::Key& k) { return mix64(k.k0 ^ (k.k1 * 0x9E3779B97F4A7C15ull)); }

// ---- the table: the device's layout, hash and probe sequence ------------------------------------------------
struct Seq { u64 line0, lmask; uint32_t off; };
inline Seq seq_of(u64 hash, u64 mask) { return Seq{(hash & mask) >> 2, mask >> 2, (uint32_t)hash & 3u}; }
inline u64 seq_slot(const Seq& s, uint32_t p) {
  return (((s.line0 + (u64)(p >> 2)) & s.lmask) << 2) | (u64)((s.off + p) & 3u);
}
inline uint32_t seq_pos(const Seq& s, u64 slot) {
  return ((uint32_t)(((slot >> 2) - s.line0) & s.lmask) << 2) | (((uint32_t)slot - s.off) & 3u);
}
inline uint32_t probe_limit(u64 mask, uint32_t maxp) { return mask >= (u64)maxp ? maxp : (uint32_t)mask + 1u; }

struct Row { float q0, q1, q2, q3; };
inline float row_get(const Row& r, int a) { return a == 0 ? r.q0 : a == 1 ? r.q1 : a == 2 ? r.q2 : r.q3; }
inline void row_set(Row& r, int a, float v) { (a == 0 ? r.q0 : a == 1 ? r.q1 : a == 2 ? r.q2 : r.q3) = v; }

inline u64 ld_u64(const uint64_t* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
inline float ld_f32(const float* p) {
  const uint32_t u = __atomic_load_n(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED);
  return bits_f32(u);
}
inline void st_f32(float* p, float v) { __atomic_store_n(reinterpret_cast<uint32_t*>(p), f32_bits(v), __ATOMIC_RELAXED); }
inline Row ld_row(const q2048_slot* s) { return Row{ld_f32(&s->q[0]), ld_f32(&s->q[1]), ld_f32(&s->q[2]), ld_f32(&s->q[3])}; }

unsigned long long g_claim_timeouts = 0;
inline u64 await_second(const q2048_slot* s) {   // the owner of the first key word publishes the second at once
  for (int spin = 0; spin < kMaxAwait; ++spin) {
    const u64 hi = ld_u64(&s->reserved);
    if (hi != 0ull) return hi;
    if ((spin & 1023) == 1023) std::this_thread::yield();
  }
  __atomic_fetch_add(&g_claim_timeouts, 1ull, __ATOMIC_RELAXED);
  return 0ull;
}
inline bool slot_is(const q2048_slot*, const Geo<4>::Key&) { return true; }
inline bool slot_is(const q2048_slot* s, const Geo<5>::Key& key) {
  u64 hi = ld_u64(&s->reserved);
  if (hi == 0ull) hi = await_second(s);
  return hi == key.k1;
}
inline void publish(q2048_slot*, const Geo<4>::Key&) {}
inline void publish(q2048_slot* s, const Geo<5>::Key& key) { __atomic_store_n(&s->reserved, key.k1, __ATOMIC_RELEASE); }

constexpr int64_t kNoSlot = INT64_MIN;
// slot index (>= 0) when present, else ~h (h = the empty slot that ended the probe) or kNoSlot (probe limit)
template <class Key>
inline int64_t probe_find(const q2048_slot* table, u64 mask, const Key& key, Row& row, uint32_t maxp = kRolloutProbe) {
  const Seq sq = seq_of(key_hash(key), mask);
  row = Row{0.f, 0.f, 0.f, 0.f};
  for (uint32_t p = 0, lim = probe_limit(mask, maxp); p < lim; ++p) {
    const u64 i = seq_slot(sq, p);
    const u64 k = ld_u64(&table[i].key);
    if (k == 0ull) return ~(int64_t)i;
    if (k == key.k0 && slot_is(&table[i], key)) { row = ld_row(&table[i]); return (int64_t)i; }
  }
  return kNoSlot;
}
// find-or-create from slot `start` of the key's sequence on
template <class Key>
inline int64_t probe_insert(q2048_slot* table, u64 mask, const Key& key, u64 start, bool& inserted,
                            uint32_t maxp = kRolloutProbe) {
  const Seq sq = seq_of(key_hash(key), mask);
  inserted = false;
  u64 i = start & mask;
  for (uint32_t p = seq_pos(sq, i), lim = probe_limit(mask, maxp); p < lim; i = seq_slot(sq, ++p)) {
    uint64_t k = ld_u64(&table[i].key);
    if (k == 0ull) {
      uint64_t expect = 0ull;
      if (__atomic_compare_exchange_n(&table[i].key, &expect, (uint64_t)key.k0, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
        publish(&table[i], key);
        inserted = true;
        return (int64_t)i;
      }
      k = expect;
    }
    if (k == key.k0 && slot_is(&table[i], key)) return (int64_t)i;
  }
  return kNoSlot;
}
// find, then create where the probe ended: the defaultdict's q_table[state] (Agent/main.py:16,41-43)
// (`create` = false: Q2048_FLAG_NO_NEW_ROWS -- the key set is closed, an absent state stays absent and reads as zeros)
template <class Key>
inline int64_t find_or_create(q2048_slot* table, u64 mask, const Key& key, Row& row, bool& inserted, bool create = true) {
  inserted = false;
  int64_t slot = probe_find(table, mask, key, row);
  if (slot < 0 && slot != kNoSlot && create) {
    slot = probe_insert(table, mask, key, (u64)~slot, inserted);
    if (slot >= 0 && !inserted) row = ld_row(&table[slot]);   // another thread created it meanwhile
  }
  return slot;
}

struct TdCounters { uint64_t retries = 0, fallbacks = 0; };
// update_q_value on one entry (Agent/main.py:43): one store, or (Q2048_FLAG_TD_CAS) a bounded compare-and-swap loop
inline float td_update(q2048_slot* slot, int a, float guess, float reward, float max_next, bool done, double lr,
                       double gamma, bool cas, TdCounters& c) {
  float nq = td_value(guess, reward, max_next, done, lr, gamma);
  if (!cas) { st_f32(&slot->q[a], nq); return nq; }
  uint32_t* addr = reinterpret_cast<uint32_t*>(&slot->q[a]);
  uint32_t expect = f32_bits(guess);
  for (int it = 0; it < kMaxCas; ++it) {
    if (__atomic_compare_exchange_n(addr, &expect, f32_bits(nq), false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return nq;
    ++c.retries;
    nq = td_value(bits_f32(expect), reward, max_next, done, lr, gamma);
  }
  ++c.fallbacks;
  st_f32(&slot->q[a], nq);
  return nq;
}

// ---- visit rows (Q2048_FLAG_NO_NEW_ROWS): the device's rowless row-cache records, byte for byte ---------------------
// With the key set closed a state without a row reads as the zero row the defaultdict would have created, and while the
// env STAYS in it (invalid moves) that fresh row learns as the defaultdict's would (Agent/main.py:43); it is never part
// of the table.  Inside a call it is the lane's carried row; between calls it travels through the caller's row cache
// as a record whose slot field is all ones.  Every other record the device would write is written empty here.
template <int N> struct RowCacheRec;
template <> struct RowCacheRec<4> { uint64_t key; float q[4]; uint64_t slot; };
template <> struct RowCacheRec<5> { uint64_t key; float q[4]; uint64_t key_hi; uint64_t slot; uint64_t pad; };
static_assert(sizeof(RowCacheRec<4>) == 32 && sizeof(RowCacheRec<5>) == 48, "ABI layout");
constexpr u64 kCacheSlotMask = (1ull << 40) - 1ull;
inline u64 cache_tag(const q2048_slot* table, u64 mask) {
  return (mix64((u64)reinterpret_cast<uintptr_t>(table) ^ (mask * 0x9E3779B97F4A7C15ull)) >> 40) << 40;
}
inline bool rec_key_is(const RowCacheRec<4>& r, const Geo<4>::Key& k) { return r.key == k.k0; }
inline bool rec_key_is(const RowCacheRec<5>& r, const Geo<5>::Key& k) { return r.key == k.k0 && r.key_hi == k.k1; }
inline void rec_set_key(RowCacheRec<4>& r, const Geo<4>::Key& k) { r.key = k.k0; }
inline void rec_set_key(RowCacheRec<5>& r, const Geo<5>::Key& k) { r.key = k.k0; r.key_hi = k.k1; r.pad = 0; }
template <int N>
inline bool visit_get(const void* cache, int64_t i, const q2048_slot* table, u64 mask, const typename Geo<N>::Key& key, Row& row) {
  if (cache == nullptr) return false;
  const RowCacheRec<N>& r = static_cast<const RowCacheRec<N>*>(cache)[i];
  if (!rec_key_is(r, key) || r.slot != (kCacheSlotMask | cache_tag(table, mask))) return false;
  row = Row{r.q[0], r.q[1], r.q[2], r.q[3]};
  return true;
}
template <int N>
inline void visit_put(void* cache, int64_t i, const q2048_slot* table, u64 mask, const typename Geo<N>::Key& key, const Row& row,
                      bool rowless) {
  if (cache == nullptr) return;
  RowCacheRec<N>& r = static_cast<RowCacheRec<N>*>(cache)[i];
  std::memset(&r, 0, sizeof r);                    // an empty record (this library never reads a row through the cache)
  if (!rowless) return;
  rec_set_key(r, key);
  r.q[0] = row.q0; r.q[1] = row.q1; r.q[2] = row.q2; r.q[3] = row.q3;
  r.slot = kCacheSlotMask | cache_tag(table, mask);
}

// q2048_rowcache_rebind: visit rows follow their table's rows into another allocation; every other record is emptied
template <int N>
void rowcache_rebind_n(void* row_cache, int64_t B, u64 tag_from, u64 tag_to) {
  RowCacheRec<N>* c = static_cast<RowCacheRec<N>*>(row_cache);
  for (int64_t i = 0; i < B; ++i) {
    if (c[i].key != 0ull && c[i].slot == (kCacheSlotMask | tag_from)) c[i].slot = kCacheSlotMask | tag_to;
    else std::memset(&c[i], 0, sizeof c[i]);
  }
}

// ---- threads ---------------------------------------------------------------------------------------------------
int threads_for(int64_t B) {
  int T = 0;
  if (const char* e = std::getenv("Q2048_HOST_THREADS")) T = std::atoi(e);
  if (T <= 0) T = (int)std::thread::hardware_concurrency();
  if (T <= 0) T = 1;
  const int64_t by_work = (B + 2047) / 2048;
  if ((int64_t)T > by_work) T = (int)by_work;
  return T < 1 ? 1 : T;
}
// f(lo, hi, t) over contiguous ranges of [0, B); returns the number of ranges
template <class F>
int parallel_ranges(int64_t B, F f) {
  const int T = threads_for(B);
  if (T == 1) { f((int64_t)0, B, 0); return 1; }
  std::vector<std::thread> th;
  th.reserve((size_t)T);
  for (int t = 0; t < T; ++t) {
    const int64_t lo = B * t / T, hi = B * (t + 1) / T;
    th.emplace_back([=] { f(lo, hi, t); });
  }
  for (auto& x : th) x.join();
  return T;
}

struct Stats {
  uint64_t i[Q2048_NSTAT_I] = {};
  double f[Q2048_NSTAT_F] = {};
  void episode(const Aux& a, uint32_t max_l2) {
    i[Q2048_ST_SCORE] += (uint64_t)(int64_t)a.score;
    i[Q2048_ST_HIST0 + (max_l2 > 22u ? 22u : max_l2)] += 1;
    const double ret = (double)a.ep_return;
    f[Q2048_SF_RETURN] += ret;
    f[Q2048_SF_RETURN_SQ] += ret * ret;
  }
};
void stats_merge(const std::vector<Stats>& parts, int used, int64_t* gi, double* gf) {   // in range order: a fixed sum
  for (int t = 0; t < used; ++t) {
    if (gi != nullptr) for (int k = 0; k < Q2048_NSTAT_I; ++k) gi[k] += (int64_t)parts[(size_t)t].i[k];
    if (gf != nullptr) for (int k = 0; k < Q2048_NSTAT_F; ++k) gf[k] += parts[(size_t)t].f[k];
  }
}

// ---- argument checks: the device library's, word for word ---------------------------------------------------
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int check_batch(int64_t B, int n) {
  if (n != 4 && n != 5) return Q2048_ERR_UNSUPPORTED;
  if (B < 0 || B > (int64_t)0x7fffffff * 256) return Q2048_ERR_SIZE;
  return Q2048_OK;
}
inline int check_table(const void* table, int cap_log2) {
  if (table == nullptr) return Q2048_ERR_NULL;
  if (cap_log2 < 4 || cap_log2 > 40) return Q2048_ERR_SIZE;
  if (!aligned16(table)) return Q2048_ERR_ALIGN;
  return Q2048_OK;
}
constexpr uint32_t kAbiFlags = Q2048_FLAG_INDEPENDENT | Q2048_FLAG_SINGLE_ENV | Q2048_FLAG_TD_CAS | Q2048_FLAG_ENV_DQN |
                               Q2048_FLAG_RESET_SHAPING | Q2048_FLAG_PLAY_ONLY | Q2048_FLAG_NO_LEARN |
                               Q2048_FLAG_NO_NEW_ROWS | Q2048_FLAG_LINE_SUMMARY;   // (the last one: accepted, not used --
                                                                                   // this library probes slot by slot; same results)
inline int check_flags(uint32_t flags, uint32_t refused = 0u) {
  return ((flags & ~kAbiFlags) || (flags & refused)) ? Q2048_ERR_FLAGS : Q2048_OK;
}
inline int env_bits(uint32_t flags) {
  return ((flags & Q2048_FLAG_ENV_DQN) ? kEnvDqn : 0) | ((flags & Q2048_FLAG_RESET_SHAPING) ? kEnvResetShaping : 0);
}
template <class BoardT>
inline StepOut env_step_any(int env, BoardT& b, Aux& a, int act, uint32_t x_pos, uint32_t x_val, uint32_t y_pos, uint32_t y_val) {
  return (env & kEnvDqn) ? env_step_profile<kEnvDqn>(b, a, act, x_pos, x_val, y_pos, y_val)
                         : env_step_profile<0>(b, a, act, x_pos, x_val, y_pos, y_val);
}

// ---- env ---------------------------------------------------------------------------------------------------------
template <int N>
void env_init_impl(uint8_t* boards, q2048_aux* aux, int64_t B, uint64_t seed, uint64_t env_id0) {
  parallel_ranges(B, [=](int64_t lo, int64_t hi, int) {
    for (int64_t i = lo; i < hi; ++i) {
      typename Geo<N>::BoardT b;
      Aux a;
      init_env(b, a, seed, env_id0 + (uint64_t)i);
      store_board(boards, i, b);
      st_aux(aux, i, a);
    }
  });
}
template <int N>
void env_reset_impl(uint8_t* boards, q2048_aux* aux, const uint8_t* mask, int64_t B, uint64_t seed, uint64_t env_id0,
                    uint32_t flags) {
  parallel_ranges(B, [=](int64_t lo, int64_t hi, int) {
    for (int64_t i = lo; i < hi; ++i) {
      if (mask != nullptr && mask[i] == 0) continue;
      typename Geo<N>::BoardT b;
      load_board(boards, i, b);
      Aux a = ld_aux(aux, i);
      begin_episode(b, a, seed, env_id0 + (uint64_t)i, (flags & Q2048_FLAG_RESET_SHAPING) != 0);
      store_board(boards, i, b);
      st_aux(aux, i, a);
    }
  });
}
template <int N>
void env_step_impl_n(const uint8_t* boards_in, uint8_t* boards, q2048_aux* aux, const uint8_t* actions, int64_t B,
                     uint64_t seed, uint64_t env_id0, uint32_t ctr, uint32_t flags, float* reward, uint8_t* done,
                     uint8_t* max_l2, int32_t* max_tile, uint32_t* status, const uint32_t* draw_pos,
                     const uint32_t* draw_val, const uint32_t* draw_opos, const uint32_t* draw_oval, int stride) {
  const int env = env_bits(flags);
  parallel_ranges(B, [=](int64_t lo, int64_t hi, int) {
    for (int64_t i = lo; i < hi; ++i) {
      typename Geo<N>::BoardT b;
      load_board(boards_in, i, b);
      const int act = actions[i];
      if (act > 3) {   // rejected, never masked (Game2048_env.py:56-60 would mis-rotate)
        status_or(status, Q2048_STATUS_BAD_ACTION);
        reward[i] = 0.f; done[i] = 0; max_l2[i] = 0;
        if (max_tile != nullptr) max_tile[i] = 0;
        if (boards != boards_in) store_board(boards, i, b);
        continue;
      }
      Aux a = ld_aux(aux, i);
      Draws x{0u, 0u, 0u, 0u}, y{0u, 0u, 0u, 0u};
      if (draw_pos != nullptr) {
        x.x2 = draw_pos[i * stride]; x.x3 = draw_val[i * stride];
        if (env & kEnvDqn) { y.x0 = draw_opos[i * stride]; y.x1 = draw_oval[i * stride]; }
      } else {
        x = draws(seed, env_id0 + (uint64_t)i, ctr, kStreamStep);
        if (env & kEnvDqn) y = draws(seed, env_id0 + (uint64_t)i, ctr, kStreamOver);
      }
      const StepOut o = env_step_any(env, b, a, act, x.x2, x.x3, y.x0, y.x1);
      store_board(boards, i, b);
      st_aux(aux, i, a);
      reward[i] = o.reward; done[i] = o.done; max_l2[i] = o.max_log2;
      if (max_tile != nullptr) max_tile[i] = o.max_log2 ? (int32_t)(1u << o.max_log2) : 0;
    }
  });
}
int env_step_impl(const uint8_t* boards_in, uint8_t* boards, q2048_aux* aux, const uint8_t* actions, int64_t B, int n,
                  uint64_t seed, uint64_t env_id0, uint32_t ctr, uint32_t flags, float* reward, uint8_t* done,
                  uint8_t* max_l2, int32_t* max_tile, uint32_t* status, const uint32_t* draw_pos,
                  const uint32_t* draw_val, const uint32_t* draw_opos, const uint32_t* draw_oval, int stride) {
  if (int e = check_batch(B, n)) return e;
  if (int e = check_flags(flags)) return e;
  if (!boards_in || !boards || !aux || !actions || !reward || !done || !max_l2 || !status) return Q2048_ERR_NULL;
  if (!aligned16(boards_in) || !aligned16(boards) || !aligned16(aux)) return Q2048_ERR_ALIGN;
  if (boards_in != boards) {
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(boards_in), a1 = reinterpret_cast<uintptr_t>(boards);
    const uintptr_t len = (uintptr_t)B * (uintptr_t)(n * n);
    if (a0 < a1 + len && a1 < a0 + len) return Q2048_ERR_ALIGN;
  }
  if (B == 0) return Q2048_OK;
  if (n == 4) env_step_impl_n<4>(boards_in, boards, aux, actions, B, seed, env_id0, ctr, flags, reward, done, max_l2,
                                 max_tile, status, draw_pos, draw_val, draw_opos, draw_oval, stride);
  else env_step_impl_n<5>(boards_in, boards, aux, actions, B, seed, env_id0, ctr, flags, reward, done, max_l2,
                          max_tile, status, draw_pos, draw_val, draw_opos, draw_oval, stride);
  return Q2048_OK;
}

// ---- agent -------------------------------------------------------------------------------------------------------
template <int N>
void q_choose_impl_n(const q2048_slot* table, u64 mask, const uint8_t* boards, int64_t B, double eps, uint64_t seed,
                     uint64_t env_id0, uint32_t ctr, uint32_t flags, uint8_t* actions, uint32_t* status,
                     const uint32_t* draw_eps, const uint32_t* draw_act, const void* cache) {
  const bool frozen = (flags & Q2048_FLAG_NO_NEW_ROWS) != 0;
  parallel_ranges(B, [=](int64_t lo, int64_t hi, int) {
    for (int64_t i = lo; i < hi; ++i) {
      typename Geo<N>::BoardT b;
      load_board(boards, i, b);
      const uint64_t id = env_id0 + (uint64_t)i;
      const u64 salt = (flags & Q2048_FLAG_INDEPENDENT) ? lane_salt(id) : 0ull;
      Draws x;
      if (draw_eps != nullptr) { x.x0 = draw_eps[i]; x.x1 = draw_act[i]; }
      else x = draws(seed, id, ctr, kStreamStep);
      int act;
      if (draw_uniform(x.x0) < eps) act = draw_action(x.x1);      // Agent/main.py:35-36: no table access when exploring
      else {
        Row r;
        const auto key = state_key(b, salt, status);
        if (probe_find(table, mask, key, r) < 0 && frozen) visit_get<N>(cache, i, table, mask, key, r);   // the env's visit row
        act = argmax4(r.q0, r.q1, r.q2, r.q3);
      }
      actions[i] = (uint8_t)act;
    }
  });
}
int q_choose_impl(const q2048_slot* table, int cap_log2, const uint8_t* boards, int64_t B, int n, double eps,
                  uint64_t seed, uint64_t env_id0, uint32_t ctr, uint32_t flags, const void* row_cache,
                  uint8_t* actions, uint32_t* status, const uint32_t* draw_eps, const uint32_t* draw_act) {
  if (int e = check_batch(B, n)) return e;
  if (int e = check_flags(flags)) return e;
  if (int e = check_table(table, cap_log2)) return e;
  if (!boards || !actions || !status) return Q2048_ERR_NULL;
  if (!aligned16(boards) || !aligned16(row_cache)) return Q2048_ERR_ALIGN;
  if (!(eps >= 0.0 && eps <= 1.0)) return Q2048_ERR_RANGE;
  if (B == 0) return Q2048_OK;
  const u64 mask = (1ull << cap_log2) - 1ull;
  if (n == 4) q_choose_impl_n<4>(table, mask, boards, B, eps, seed, env_id0, ctr, flags, actions, status, draw_eps, draw_act, row_cache);
  else q_choose_impl_n<5>(table, mask, boards, B, eps, seed, env_id0, ctr, flags, actions, status, draw_eps, draw_act, row_cache);
  return Q2048_OK;
}

template <int N>
void q_update_impl_n(q2048_slot* table, u64 mask, const uint8_t* s, const uint8_t* actions, const float* reward,
                     const uint8_t* s2, const uint8_t* done, int64_t B, double lr, double gamma, uint64_t env_id0,
                     uint32_t flags, int64_t* stats_i, uint32_t* status, void* cache) {
  const int T = threads_for(B);
  std::vector<Stats> parts((size_t)T);
  Stats* sp = parts.data();
  const bool cas = (flags & Q2048_FLAG_TD_CAS) != 0, create = (flags & Q2048_FLAG_NO_NEW_ROWS) == 0;
  const int used = parallel_ranges(B, [=](int64_t lo, int64_t hi, int t) {
    Stats& st = sp[t];
    TdCounters tdc;
    for (int64_t i = lo; i < hi; ++i) {
      const int act = actions[i];
      if (act > 3) { status_or(status, Q2048_STATUS_BAD_ACTION); continue; }
      typename Geo<N>::BoardT b_s, b_n;
      load_board(s, i, b_s);
      load_board(s2, i, b_n);
      const u64 salt = (flags & Q2048_FLAG_INDEPENDENT) ? lane_salt(env_id0 + (uint64_t)i) : 0ull;
      const auto key_s = state_key(b_s, salt, status);
      const auto key_n = state_key(b_n, salt, status);
      Row rs, rn;
      bool ins_s = false, ins_n = false;
      const int64_t slot = find_or_create(table, mask, key_s, rs, ins_s, create);     // q_table[state] (:43)
      if (slot < 0 && !create) visit_get<N>(cache, i, table, mask, key_s, rs);       // closed key set: the env's visit row
      rn = rs;
      const bool same = key_eq(key_n, key_s);
      int64_t slot_n = slot;
      if (!same) slot_n = find_or_create(table, mask, key_n, rn, ins_n, create);     // q_table[next_state] (:41)
      st.i[Q2048_ST_INSERTS] += (uint64_t)ins_s + (uint64_t)ins_n;
      const float max_next = max4(rn.q0, rn.q1, rn.q2, rn.q3);
      if (slot >= 0) {
        td_update(&table[slot], act, row_get(rs, act), reward[i], max_next, done[i] != 0, lr, gamma, cas, tdc);
      } else {
        st.i[Q2048_ST_DROPS] += 1;
        if (create) status_or(status, Q2048_STATUS_TABLE_FULL);   // (closed key set: the caller's policy)
        else if (same) row_set(rn, act, td_value(row_get(rs, act), reward[i], max_next, done[i] != 0, lr, gamma));
      }
      visit_put<N>(cache, i, table, mask, key_n, rn, !create && slot_n < 0);
    }
    st.i[Q2048_ST_CAS_RETRY] += tdc.retries;
    st.i[Q2048_ST_CAS_FALLBACK] += tdc.fallbacks;
  });
  stats_merge(parts, used, stats_i, nullptr);
}

template <int N>
void q_lookup_impl_n(const q2048_slot* table, u64 mask, const uint8_t* boards, int64_t B, uint64_t env_id0,
                     uint32_t flags, float* q_out, uint8_t* found, uint32_t* status) {
  parallel_ranges(B, [=](int64_t lo, int64_t hi, int) {
    for (int64_t i = lo; i < hi; ++i) {
      typename Geo<N>::BoardT b;
      load_board(boards, i, b);
      const uint64_t id = (flags & Q2048_FLAG_SINGLE_ENV) ? env_id0 : env_id0 + (uint64_t)i;
      const u64 salt = (flags & Q2048_FLAG_INDEPENDENT) ? lane_salt(id) : 0ull;
      Row r;
      const int64_t slot = probe_find(table, mask, state_key(b, salt, status), r, kMaxProbe);
      q_out[4 * i] = r.q0; q_out[4 * i + 1] = r.q1; q_out[4 * i + 2] = r.q2; q_out[4 * i + 3] = r.q3;
      if (found != nullptr) found[i] = slot >= 0;
    }
  });
}

// ---- the fused rollout: Agent/main.py:91-101 + the reset of :81, `steps` times per env ----------------------------
// A thread walks its range in GROUPS of kGroup envs, step by step: first every env of the group chooses, steps
// and asks for the cache line its next state's row starts on (a prefetch), then every env does its table work.
// One env at a time, every probe of s' is a DRAM miss the core sits through (~100 ns against ~100 ns of
// arithmetic per step); sixteen at a time the misses overlap.  Per env nothing changes -- the same calls in the
// same order on the same draws -- so B = 1 and private rows are the reference's loop as before; on a shared table
// the interleaving of envs is a different schedule of the same Hogwild learner.
constexpr int kGroup = 16;
template <int N>
void fused_rollout_n(uint8_t* boards, q2048_aux* aux, q2048_slot* table, u64 mask, int64_t B, int steps, double eps,
                     double lr, double gamma, uint64_t seed, uint64_t env_id0, uint32_t ctr0, uint32_t flags,
                     int64_t* stats_i, double* stats_f, uint32_t* status, q2048_episode* log, int64_t log_cap,
                     uint64_t* log_count, void* cache) {
  using BoardT = typename Geo<N>::BoardT;
  using Key = typename Geo<N>::Key;
  struct Lane {
    BoardT b; Aux a; Key key_s, key_n; Row q; int64_t slot_s; u64 salt; uint64_t id; double reward_sum;
    StepOut o; int act; bool explored, same;
  };
  const int env = env_bits(flags);
  const bool play_only = (flags & Q2048_FLAG_PLAY_ONLY) != 0, no_learn = (flags & Q2048_FLAG_NO_LEARN) != 0;
  const bool learns = !play_only && !no_learn, frozen = (flags & Q2048_FLAG_NO_NEW_ROWS) != 0;
  const bool creates = learns && !frozen, cas = (flags & Q2048_FLAG_TD_CAS) != 0;
  const int T = threads_for(B);
  std::vector<Stats> parts((size_t)T);
  Stats* sp = parts.data();
  const int used = parallel_ranges(B, [=](int64_t lo, int64_t hi, int tid) {
    Stats& st = sp[tid];
    TdCounters tdc;
    bool any_drop = false;
    Lane lane[kGroup];
    for (int64_t g0 = lo; g0 < hi; g0 += kGroup) {
      const int n = (int)(hi - g0 < kGroup ? hi - g0 : kGroup);
      for (int l = 0; l < n; ++l) {
        Lane& L = lane[l];
        const int64_t i = g0 + l;
        L.id = env_id0 + (uint64_t)i;
        L.salt = (flags & Q2048_FLAG_INDEPENDENT) ? lane_salt(L.id) : 0ull;
        load_board(boards, i, L.b);
        L.a = ld_aux(aux, i);
        L.key_s = state_key(L.b, L.salt, status);
        // the row of the current state: read when the state is reached and carried (as the kernel carries it in
        // registers); created at its first update (the defaultdict creates q_table[state] at :43)
        L.q = Row{0.f, 0.f, 0.f, 0.f};
        L.slot_s = play_only ? kNoSlot : probe_find(table, mask, L.key_s, L.q);
        if (frozen && learns && L.slot_s < 0) visit_get<N>(cache, i, table, mask, L.key_s, L.q);   // the visit row goes on
        L.reward_sum = 0.0;
      }
      for (int t = 0; t < steps; ++t) {
        for (int l = 0; l < n; ++l) {                    // choose, step, ask for s'
          Lane& L = lane[l];
          const Draws x = draws(seed, L.id, ctr0 + (uint32_t)t, kStreamStep);
          Draws y{0u, 0u, 0u, 0u};
          if (env & kEnvDqn) y = draws(seed, L.id, ctr0 + (uint32_t)t, kStreamOver);
          L.act = eps_greedy(eps, x.x0, x.x1, L.q.q0, L.q.q1, L.q.q2, L.q.q3, L.explored);       // :92
          L.o = env_step_any(env, L.b, L.a, L.act, x.x2, x.x3, y.x0, y.x1);                      // :93
          L.key_n = state_key(L.b, L.salt, status);                                              // :94
          L.same = key_eq(L.key_n, L.key_s);
          if (!L.same && !play_only) __builtin_prefetch(&table[key_hash(L.key_n) & mask], 1, 1);
        }
        for (int l = 0; l < n; ++l) {                    // the table: q_table[next_state], the TD write, the reset
          Lane& L = lane[l];
          const StepOut& o = L.o;
          const int act = L.act;
          bool ins_s = false, ins_n = false;
          if (L.slot_s < 0 && L.slot_s != kNoSlot && creates) L.slot_s = probe_insert(table, mask, L.key_s, (u64)~L.slot_s, ins_s);
          Row qn = L.q;                                                                          // q_table[next_state] (:41)
          int64_t slot_n = L.slot_s;
          if (!L.same) {
            if (play_only) { qn = Row{0.f, 0.f, 0.f, 0.f}; slot_n = kNoSlot; }
            else if (creates) slot_n = find_or_create(table, mask, L.key_n, qn, ins_n);
            else slot_n = probe_find(table, mask, L.key_n, qn);
          }
          const float max_next = max4(qn.q0, qn.q1, qn.q2, qn.q3);
          const bool updated = L.slot_s >= 0;
          float nq = 0.f;
          if (updated) {                                                                         // :43, :99
            if (no_learn) nq = td_value(row_get(L.q, act), o.reward, max_next, o.done != 0, lr, gamma);
            else nq = td_update(&table[L.slot_s], act, row_get(L.q, act), o.reward, max_next, o.done != 0, lr, gamma, cas, tdc);
          } else if (frozen && learns) {   // closed key set, no row: the update lands in the visit row (the carried L.q)
            nq = td_value(row_get(L.q, act), o.reward, max_next, o.done != 0, lr, gamma);
          }
          st.i[Q2048_ST_VALID] += o.valid != 0;
          st.i[Q2048_ST_EXPLORE] += L.explored;
          st.i[Q2048_ST_INSERTS] += (uint64_t)ins_s + (uint64_t)ins_n;
          if (!updated && learns) { st.i[Q2048_ST_DROPS] += 1; any_drop = any_drop || !frozen; }
          L.reward_sum += (double)o.reward;
          if (o.done) {                                                                          // :103
            st.i[Q2048_ST_EPISODES] += 1;
            st.episode(L.a, o.max_log2);
            if (log != nullptr) {                                                                // :59-62, :105
              const uint64_t at = __atomic_fetch_add(log_count, (uint64_t)1, __ATOMIC_RELAXED);
              if ((int64_t)at < log_cap) {
                Row ql = L.q;
                if ((updated || (frozen && learns)) && !no_learn) row_set(ql, act, nq);
                q2048_episode rec;
                rec.env_id = L.id; rec.episode = L.a.episode; rec.action = (uint8_t)act;
                rec.max_log2 = o.max_log2; rec.steps_lo = (uint16_t)(ctr0 + (uint32_t)t);
                rec.reward = o.reward; rec.total_return = L.a.ep_return; rec.score = L.a.score;
                rec.q[0] = ql.q0; rec.q[1] = ql.q1; rec.q[2] = ql.q2; rec.q[3] = ql.q3;
                rec.reserved = 0u;
                log[at] = rec;
              }
            }
            begin_episode(L.b, L.a, seed, L.id, (env & kEnvResetShaping) != 0);                  // :81
            L.key_s = state_key(L.b, L.salt, status);
            L.q = Row{0.f, 0.f, 0.f, 0.f};
            L.slot_s = play_only ? kNoSlot : probe_find(table, mask, L.key_s, L.q);
          } else if (L.same) {           // invalid move: same state, its row just changed (:100)
            if ((updated || (frozen && learns)) && !no_learn) row_set(L.q, act, nq);
            else if (!updated) L.slot_s = kNoSlot;
          } else {
            L.key_s = L.key_n; L.slot_s = slot_n; L.q = qn;                                      // :100
          }
        }
      }
      for (int l = 0; l < n; ++l) {
        st.i[Q2048_ST_STEPS] += (uint64_t)steps;
        st.f[Q2048_SF_REWARD] += lane[l].reward_sum;
        store_board(boards, g0 + l, lane[l].b);
        st_aux(aux, g0 + l, lane[l].a);
        if (!play_only) visit_put<N>(cache, g0 + l, table, mask, lane[l].key_s, lane[l].q, frozen && learns && lane[l].slot_s < 0);
      }
    }
    st.i[Q2048_ST_CAS_RETRY] += tdc.retries;
    st.i[Q2048_ST_CAS_FALLBACK] += tdc.fallbacks;
    if (any_drop) status_or(status, Q2048_STATUS_TABLE_FULL);
  });
  stats_merge(parts, used, stats_i, stats_f);
}

// ---- the deterministic step: phase 1 over every env against the step-start table, then each (row, action) group
// folds its updates in env order in double and rounds once (include/q2048.h "Deterministic mode") ----------------
template <int N>
void det_rollout_n(uint8_t* boards, q2048_aux* aux, q2048_slot* table, u64 mask, int64_t B, int64_t steps, double eps,
                   double lr, double gamma, uint64_t seed, uint64_t env_id0, uint32_t ctr0, uint32_t flags,
                   int64_t* stats_i, double* stats_f, uint32_t* status, void* row_cache) {
  const int env = env_bits(flags);
  const bool create = (flags & Q2048_FLAG_NO_NEW_ROWS) == 0;
  void* const cache = create ? nullptr : row_cache;   // visit rows: closed key set only
  std::vector<int64_t> cell((size_t)B);          // slot * 4 + action, or -1 (dropped)
  std::vector<double> target((size_t)B);
  int64_t* cp = cell.data();
  double* tp = target.data();
  const int T = threads_for(B);
  for (int64_t t = 0; t < steps; ++t) {
    std::vector<Stats> parts((size_t)T);
    Stats* sp = parts.data();
    const uint32_t ctr = ctr0 + (uint32_t)t;
    const int used = parallel_ranges(B, [=](int64_t lo, int64_t hi, int tid) {
      Stats& st = sp[tid];
      for (int64_t i = lo; i < hi; ++i) {
        const uint64_t id = env_id0 + (uint64_t)i;
        const u64 salt = (flags & Q2048_FLAG_INDEPENDENT) ? lane_salt(id) : 0ull;
        typename Geo<N>::BoardT b;
        load_board(boards, i, b);
        Aux a = ld_aux(aux, i);
        const auto key_s = state_key(b, salt, status);
        const Draws x = draws(seed, id, ctr, kStreamStep);
        Draws y{0u, 0u, 0u, 0u};
        if (env & kEnvDqn) y = draws(seed, id, ctr, kStreamOver);
        Row q, qn;
        bool ins_s = false, ins_n = false, explored;
        const int64_t slot_s = find_or_create(table, mask, key_s, q, ins_s, create);
        const bool dropped = slot_s < 0;
        if (dropped) { q = Row{0.f, 0.f, 0.f, 0.f}; if (create || slot_s == kNoSlot) status_or(status, Q2048_STATUS_TABLE_FULL); }
        if (dropped) visit_get<N>(cache, i, table, mask, key_s, q);                          // the env's visit row
        const int act = eps_greedy(eps, x.x0, x.x1, q.q0, q.q1, q.q2, q.q3, explored);       // :92
        const StepOut o = env_step_any(env, b, a, act, x.x2, x.x3, y.x0, y.x1);              // :93
        const auto key_n = state_key(b, salt, status);
        find_or_create(table, mask, key_n, qn, ins_n, create);                               // :41
        if (cache != nullptr) {
          // a visit row is one env's: its update needs no ordering and is applied here, not by phase 2
          const bool stays = dropped && key_eq(key_n, key_s);
          if (stays) qn = q;
          Row v = q;
          if (stays && !o.done)
            row_set(v, act, td_value(row_get(q, act), o.reward, max4(qn.q0, qn.q1, qn.q2, qn.q3), false, lr, gamma));
          visit_put<N>(cache, i, table, mask, key_s, v, stays && !o.done);
        }
        cp[i] = dropped ? -1 : slot_s * 4 + act;
        tp[i] = td_target(o.reward, max4(qn.q0, qn.q1, qn.q2, qn.q3), o.done != 0, gamma);   // :42
        st.i[Q2048_ST_STEPS] += 1;
        st.i[Q2048_ST_VALID] += o.valid != 0;
        st.i[Q2048_ST_EXPLORE] += explored;
        st.i[Q2048_ST_INSERTS] += (uint64_t)ins_s + (uint64_t)ins_n;
        st.i[Q2048_ST_DROPS] += dropped;
        st.f[Q2048_SF_REWARD] += (double)o.reward;
        if (o.done) {
          st.i[Q2048_ST_EPISODES] += 1;
          st.episode(a, o.max_log2);
          begin_episode(b, a, seed, id, (env & kEnvResetShaping) != 0);
        }
        store_board(boards, i, b);
        st_aux(aux, i, a);
      }
    });
    stats_merge(parts, used, stats_i, stats_f);
    // phase 2: env order, one running double per (row, action), one rounding per group and step (:43)
    std::unordered_map<int64_t, double> run;
    run.reserve((size_t)B);
    for (int64_t i = 0; i < B; ++i) {
      if (cp[i] < 0) continue;
      auto it = run.find(cp[i]);
      if (it == run.end()) it = run.emplace(cp[i], (double)table[cp[i] >> 2].q[cp[i] & 3]).first;
      it->second = td_fold(it->second, tp[i], lr);
    }
    for (const auto& kv : run) table[kv.first >> 2].q[kv.first & 3] = (float)kv.second;
  }
}

// ---- row-tuple linear Q (BASELINE configs[1]): W = float[4][65536][4] --------------------------------------------
struct RtRows { Row e[4]; };
inline float* rt_entry(float* w, int r, uint32_t idx) { return w + (((size_t)r * kRtIdx + idx) << 2); }
inline RtRows rt_gather(const float* w, const Board& b) {
  const uint32_t idx[4] = {pack_row(b.r0), pack_row(b.r1), pack_row(b.r2), pack_row(b.r3)};
  RtRows e;
  for (int r = 0; r < 4; ++r) {
    const float* p = rt_entry(const_cast<float*>(w), r, idx[r]);
    e.e[r] = Row{ld_f32(p), ld_f32(p + 1), ld_f32(p + 2), ld_f32(p + 3)};
  }
  return e;
}
inline Row rt_q(const RtRows& e) {
  return Row{rt_sum(e.e[0].q0, e.e[1].q0, e.e[2].q0, e.e[3].q0), rt_sum(e.e[0].q1, e.e[1].q1, e.e[2].q1, e.e[3].q1),
             rt_sum(e.e[0].q2, e.e[1].q2, e.e[2].q2, e.e[3].q2), rt_sum(e.e[0].q3, e.e[1].q3, e.e[2].q3, e.e[3].q3)};
}
inline void rt_scatter(float* w, const Board& b, const RtRows& e, int act, float d) {
  const uint32_t idx[4] = {pack_row(b.r0), pack_row(b.r1), pack_row(b.r2), pack_row(b.r3)};
  for (int r = 0; r < 4; ++r) st_f32(rt_entry(w, r, idx[r]) + act, row_get(e.e[r], act) + d);
}
}  // namespace

extern "C" {

int q2048_abi_version(void) { return Q2048_ABI_VERSION; }

int q2048_claim_timeouts(uint64_t* count_host) {
  if (count_host == nullptr) return Q2048_ERR_NULL;
  *count_host = (uint64_t)__atomic_load_n(&g_claim_timeouts, __ATOMIC_RELAXED);
  return Q2048_OK;
}

const char* q2048_strerror(int code) {
  switch (code) {
    case Q2048_OK: return "ok";
    case Q2048_ERR_NULL: return "a required pointer is NULL";
    case Q2048_ERR_SIZE: return "size out of range (batch, steps, cap_log2 or key_words)";
    case Q2048_ERR_ALIGN: return "boards/aux/table must be 16-byte aligned";
    case Q2048_ERR_UNSUPPORTED: return "unsupported here (board side other than 4 or 5; device-only entry point on the host library)";
    case Q2048_ERR_LAUNCH: return "HIP launch failed";
    case Q2048_ERR_RANGE: return "scalar out of range (eps in [0,1], lr and gamma finite)";
    case Q2048_ERR_FLAGS: return "flag bits this entry point does not take";
    case Q2048_ERR_ALLOC: return "device memory could not be reserved, created or mapped";
    case Q2048_ERR_VERIFY: return "a table failed its self-check";
    case Q2048_ERR_BUSY: return "the table already takes part in a growth";
    case Q2048_PENDING: return "still working (not an error)";
    default: return "unknown error";
  }
}
size_t q2048_sizeof_aux(void) { return sizeof(q2048_aux); }
size_t q2048_sizeof_slot(void) { return sizeof(q2048_slot); }
size_t q2048_sizeof_rowcache(int n) { return n == 4 ? 32 : n == 5 ? 48 : 0; }
int q2048_rowcache_rebind(void* row_cache, int64_t B, int n, const q2048_slot* from_table, int from_cap_log2,
                          const q2048_slot* to_table, int to_cap_log2, void*) {
  if (int e = check_batch(B, n)) return e;
  if (int e = check_table(from_table, from_cap_log2)) return e;   // (an address of the past: never dereferenced)
  if (int e = check_table(to_table, to_cap_log2)) return e;
  if (row_cache == nullptr) return Q2048_ERR_NULL;
  if (!aligned16(row_cache)) return Q2048_ERR_ALIGN;
  const u64 tag_from = cache_tag(from_table, (1ull << from_cap_log2) - 1ull), tag_to = cache_tag(to_table, (1ull << to_cap_log2) - 1ull);
  if (n == 4) rowcache_rebind_n<4>(row_cache, B, tag_from, tag_to);
  else rowcache_rebind_n<5>(row_cache, B, tag_from, tag_to);
  return Q2048_OK;
}

int q2048_env_init(uint8_t* boards, q2048_aux* aux, int64_t B, int n, uint64_t seed, uint64_t env_id0, void*) {
  if (int e = check_batch(B, n)) return e;
  if (boards == nullptr || aux == nullptr) return Q2048_ERR_NULL;
  if (!aligned16(boards) || !aligned16(aux)) return Q2048_ERR_ALIGN;
  if (B == 0) return Q2048_OK;
  if (n == 4) env_init_impl<4>(boards, aux, B, seed, env_id0); else env_init_impl<5>(boards, aux, B, seed, env_id0);
  return Q2048_OK;
}
int q2048_env_reset_ex(uint8_t* boards, q2048_aux* aux, const uint8_t* mask, int64_t B, int n, uint64_t seed,
                       uint64_t env_id0, uint32_t flags, void*) {
  if (int e = check_batch(B, n)) return e;
  if (int e = check_flags(flags)) return e;
  if (boards == nullptr || aux == nullptr) return Q2048_ERR_NULL;
  if (!aligned16(boards) || !aligned16(aux)) return Q2048_ERR_ALIGN;
  if (B == 0) return Q2048_OK;
  if (n == 4) env_reset_impl<4>(boards, aux, mask, B, seed, env_id0, flags);
  else env_reset_impl<5>(boards, aux, mask, B, seed, env_id0, flags);
  return Q2048_OK;
}
int q2048_env_reset(uint8_t* boards, q2048_aux* aux, const uint8_t* mask, int64_t B, int n, uint64_t seed,
                    uint64_t env_id0, void* stream) {
  return q2048_env_reset_ex(boards, aux, mask, B, n, seed, env_id0, 0u, stream);
}
int q2048_env_step(uint8_t* boards, q2048_aux* aux, const uint8_t* actions, int64_t B, int n, uint64_t seed,
                   uint64_t env_id0, uint32_t ctr, float* reward, uint8_t* done, uint8_t* max_log2, uint32_t* status,
                   void*) {
  return env_step_impl(boards, boards, aux, actions, B, n, seed, env_id0, ctr, 0u, reward, done, max_log2, nullptr,
                       status, nullptr, nullptr, nullptr, nullptr, 1);
}
int q2048_env_step_ex(uint8_t* boards, q2048_aux* aux, const uint8_t* actions, int64_t B, int n, uint64_t seed,
                      uint64_t env_id0, uint32_t ctr, uint32_t flags, const uint32_t* draws4, float* reward,
                      uint8_t* done, uint8_t* max_log2, uint32_t* status, void*) {
  if (draws4 == nullptr)
    return env_step_impl(boards, boards, aux, actions, B, n, seed, env_id0, ctr, flags, reward, done, max_log2,
                         nullptr, status, nullptr, nullptr, nullptr, nullptr, 1);
  return env_step_impl(boards, boards, aux, actions, B, n, 0, 0, 0, flags, reward, done, max_log2, nullptr, status,
                       draws4, draws4 + 1, draws4 + 2, draws4 + 3, 4);
}
int q2048_env_step_to(const uint8_t* boards_in, uint8_t* boards_out, q2048_aux* aux, const uint8_t* actions, int64_t B,
                      int n, uint64_t seed, uint64_t env_id0, uint32_t ctr, uint32_t flags, float* reward,
                      uint8_t* done, uint8_t* max_log2, int32_t* max_tile, uint32_t* status, void*) {
  return env_step_impl(boards_in, boards_out, aux, actions, B, n, seed, env_id0, ctr, flags, reward, done, max_log2,
                       max_tile, status, nullptr, nullptr, nullptr, nullptr, 1);
}
int q2048_env_step_draws(uint8_t* boards, q2048_aux* aux, const uint8_t* actions, const uint32_t* draw_pos,
                         const uint32_t* draw_val, int64_t B, int n, float* reward, uint8_t* done, uint8_t* max_log2,
                         uint32_t* status, void*) {
  if (!draw_pos || !draw_val) return Q2048_ERR_NULL;
  return env_step_impl(boards, boards, aux, actions, B, n, 0, 0, 0, 0u, reward, done, max_log2, nullptr, status,
                       draw_pos, draw_val, nullptr, nullptr, 1);
}

int q2048_q_choose(const q2048_slot* table, int cap_log2, const uint8_t* boards, int64_t B, int n, double eps,
                   uint64_t seed, uint64_t env_id0, uint32_t ctr, uint32_t flags, uint8_t* actions, uint32_t* status,
                   void*) {
  return q_choose_impl(table, cap_log2, boards, B, n, eps, seed, env_id0, ctr, flags, nullptr, actions, status,
                       nullptr, nullptr);
}
int q2048_q_choose_cached(const q2048_slot* table, int cap_log2, const uint8_t* boards, int64_t B, int n, double eps,
                          uint64_t seed, uint64_t env_id0, uint32_t ctr, uint32_t flags, const void* row_cache,
                          uint8_t* actions, uint32_t* status, void*) {
  return q_choose_impl(table, cap_log2, boards, B, n, eps, seed, env_id0, ctr, flags, row_cache, actions, status,
                       nullptr, nullptr);
}
int q2048_q_choose_draws(const q2048_slot* table, int cap_log2, const uint8_t* boards, const uint32_t* draw_eps,
                         const uint32_t* draw_act, int64_t B, int n, double eps, uint64_t env_id0, uint32_t flags,
                         uint8_t* actions, uint32_t* status, void*) {
  if (!draw_eps || !draw_act) return Q2048_ERR_NULL;
  return q_choose_impl(table, cap_log2, boards, B, n, eps, 0, env_id0, 0, flags, nullptr, actions, status, draw_eps,
                       draw_act);
}
int q2048_q_update_cached(q2048_slot* table, int cap_log2, const uint8_t* boards_s, const uint8_t* actions,
                          const float* reward, const uint8_t* boards_s2, const uint8_t* done, int64_t B, int n,
                          double lr, double gamma, uint64_t env_id0, uint32_t flags, void* row_cache,
                          int64_t* stats_i, uint32_t* status, void*) {
  if (int e = check_batch(B, n)) return e;
  if (int e = check_flags(flags)) return e;
  if (int e = check_table(table, cap_log2)) return e;
  if (!boards_s || !actions || !reward || !boards_s2 || !done || !status) return Q2048_ERR_NULL;
  if (!aligned16(boards_s) || !aligned16(boards_s2) || !aligned16(row_cache)) return Q2048_ERR_ALIGN;
  if (!(lr == lr) || !(gamma == gamma)) return Q2048_ERR_RANGE;
  if (B == 0) return Q2048_OK;
  const u64 mask = (1ull << cap_log2) - 1ull;
  if (n == 4) q_update_impl_n<4>(table, mask, boards_s, actions, reward, boards_s2, done, B, lr, gamma, env_id0, flags, stats_i, status, row_cache);
  else q_update_impl_n<5>(table, mask, boards_s, actions, reward, boards_s2, done, B, lr, gamma, env_id0, flags, stats_i, status, row_cache);
  return Q2048_OK;
}
int q2048_q_update(q2048_slot* table, int cap_log2, const uint8_t* boards_s, const uint8_t* actions,
                   const float* reward, const uint8_t* boards_s2, const uint8_t* done, int64_t B, int n, double lr,
                   double gamma, uint64_t env_id0, uint32_t flags, int64_t* stats_i, uint32_t* status, void* stream) {
  return q2048_q_update_cached(table, cap_log2, boards_s, actions, reward, boards_s2, done, B, n, lr, gamma, env_id0,
                               flags, nullptr, stats_i, status, stream);
}
int q2048_q_lookup(const q2048_slot* table, int cap_log2, const uint8_t* boards, int64_t B, int n, uint64_t env_id0,
                   uint32_t flags, float* q_out, uint8_t* found, uint32_t* status, void*) {
  if (int e = check_batch(B, n)) return e;
  if (int e = check_flags(flags)) return e;
  if (int e = check_table(table, cap_log2)) return e;
  if (!boards || !q_out || !status) return Q2048_ERR_NULL;
  if (!aligned16(boards) || !aligned16(q_out)) return Q2048_ERR_ALIGN;
  if (B == 0) return Q2048_OK;
  const u64 mask = (1ull << cap_log2) - 1ull;
  if (n == 4) q_lookup_impl_n<4>(table, mask, boards, B, env_id0, flags, q_out, found, status);
  else q_lookup_impl_n<5>(table, mask, boards, B, env_id0, flags, q_out, found, status);
  return Q2048_OK;
}

int q2048_fused_rollout_opts(uint8_t* boards, q2048_aux* aux, q2048_slot* table, int cap_log2, int64_t B, int n,
                             int64_t steps, double eps, double lr, double gamma, uint64_t seed, uint64_t env_id0,
                             uint32_t ctr0, uint32_t flags, int64_t* stats_i, double* stats_f, uint32_t* status,
                             const q2048_rollout_opts* opts, void*) {
  q2048_rollout_opts o = {};
  if (opts != nullptr) {
    if (opts->size != sizeof(q2048_rollout_opts)) return Q2048_ERR_SIZE;
    o = *opts;
  }
  if (int e = check_batch(B, n)) return e;
  if (int e = check_flags(flags)) return e;
  if (o.log != nullptr && (o.log_count == nullptr || o.log_capacity < 0)) return Q2048_ERR_NULL;
  if (o.log != nullptr && !aligned16(o.log)) return Q2048_ERR_ALIGN;
  if (o.row_cache != nullptr && !aligned16(o.row_cache)) return Q2048_ERR_ALIGN;
  if (o.stats_mirror != nullptr && (o.mirror_ticket == nullptr || stats_i == nullptr || stats_f == nullptr))
    return Q2048_ERR_NULL;
  if (o.stats_mirror != nullptr && (reinterpret_cast<uintptr_t>(o.stats_mirror) & 7u)) return Q2048_ERR_ALIGN;
  if (int e = check_table(table, cap_log2)) return e;
  if (!boards || !aux || !status) return Q2048_ERR_NULL;
  if (!aligned16(boards) || !aligned16(aux)) return Q2048_ERR_ALIGN;
  if (steps < 0 || steps > (1 << 30)) return Q2048_ERR_SIZE;
  if (!(eps >= 0.0 && eps <= 1.0) || !(lr == lr) || !(gamma == gamma)) return Q2048_ERR_RANGE;
  if (B == 0 || steps == 0) return Q2048_OK;
  const u64 mask = (1ull << cap_log2) - 1ull;
  if (n == 4) fused_rollout_n<4>(boards, aux, table, mask, B, (int)steps, eps, lr, gamma, seed, env_id0, ctr0, flags,
                                 stats_i, stats_f, status, o.log, o.log_capacity, o.log_count, o.row_cache);
  else fused_rollout_n<5>(boards, aux, table, mask, B, (int)steps, eps, lr, gamma, seed, env_id0, ctr0, flags, stats_i,
                          stats_f, status, o.log, o.log_capacity, o.log_count, o.row_cache);
  if (o.stats_mirror != nullptr) {               // the statistics as they stand after this call, and its number
    uint64_t* m = static_cast<uint64_t*>(o.stats_mirror);
    std::memcpy(m, stats_i, sizeof(int64_t) * Q2048_NSTAT_I);
    std::memcpy(m + Q2048_NSTAT_I, stats_f, sizeof(double) * Q2048_NSTAT_F);
    m[Q2048_MIRROR_SEQ] = ++o.mirror_ticket[1];
  }
  return Q2048_OK;
}
int q2048_fused_rollout(uint8_t* boards, q2048_aux* aux, q2048_slot* table, int cap_log2, int64_t B, int n,
                        int64_t steps, double eps, double lr, double gamma, uint64_t seed, uint64_t env_id0,
                        uint32_t ctr0, uint32_t flags, int64_t* stats_i, double* stats_f, uint32_t* status,
                        void* stream) {
  return q2048_fused_rollout_opts(boards, aux, table, cap_log2, B, n, steps, eps, lr, gamma, seed, env_id0, ctr0,
                                  flags, stats_i, stats_f, status, nullptr, stream);
}
int q2048_fused_rollout_log(uint8_t* boards, q2048_aux* aux, q2048_slot* table, int cap_log2, int64_t B, int n,
                            int64_t steps, double eps, double lr, double gamma, uint64_t seed, uint64_t env_id0,
                            uint32_t ctr0, uint32_t flags, int64_t* stats_i, double* stats_f, uint32_t* status,
                            q2048_episode* log, int64_t log_capacity, uint64_t* log_count, void* stream) {
  q2048_rollout_opts o = {};
  o.size = (uint32_t)sizeof(o);
  o.log = log; o.log_capacity = log_capacity; o.log_count = log_count;
  return q2048_fused_rollout_opts(boards, aux, table, cap_log2, B, n, steps, eps, lr, gamma, seed, env_id0, ctr0,
                                  flags, stats_i, stats_f, status, &o, stream);
}

int64_t q2048_det_workspace_bytes(int64_t B, int cap_log2) {
  if (B < 0 || B > 0x7fffffffll || cap_log2 < 4 || cap_log2 > 40) return Q2048_ERR_SIZE;
  return 256;                                     // the host step keeps its scratch itself
}
int q2048_det_rollout(uint8_t* boards, q2048_aux* aux, q2048_slot* table, int cap_log2, int64_t B, int n,
                      int64_t steps, double eps, double lr, double gamma, uint64_t seed, uint64_t env_id0,
                      uint32_t ctr0, uint32_t flags, int64_t* stats_i, double* stats_f, uint32_t* status,
                      void* workspace, int64_t workspace_bytes, void* stream) {
  return q2048_det_rollout_cached(boards, aux, table, cap_log2, B, n, steps, eps, lr, gamma, seed, env_id0, ctr0, flags,
                                  stats_i, stats_f, status, workspace, workspace_bytes, nullptr, stream);
}
int q2048_det_rollout_cached(uint8_t* boards, q2048_aux* aux, q2048_slot* table, int cap_log2, int64_t B, int n,
                             int64_t steps, double eps, double lr, double gamma, uint64_t seed, uint64_t env_id0,
                             uint32_t ctr0, uint32_t flags, int64_t* stats_i, double* stats_f, uint32_t* status,
                             void* workspace, int64_t workspace_bytes, void* row_cache, void*) {
  if (int e = check_batch(B, n)) return e;
  if (row_cache != nullptr && !aligned16(row_cache)) return Q2048_ERR_ALIGN;
  if (int e = check_flags(flags, Q2048_FLAG_NO_LEARN | Q2048_FLAG_PLAY_ONLY)) return e;
  if (B > 0x7fffffffll) return Q2048_ERR_SIZE;
  if (int e = check_table(table, cap_log2)) return e;
  if (!boards || !aux || !status || !workspace) return Q2048_ERR_NULL;
  if (!aligned16(boards) || !aligned16(aux) || (reinterpret_cast<uintptr_t>(workspace) & 255u)) return Q2048_ERR_ALIGN;
  if (steps < 0 || steps > (1 << 30) || workspace_bytes < 256) return Q2048_ERR_SIZE;
  if (!(eps >= 0.0 && eps <= 1.0) || !(lr == lr) || !(gamma == gamma)) return Q2048_ERR_RANGE;
  if (B == 0 || steps == 0) return Q2048_OK;
  const u64 mask = (1ull << cap_log2) - 1ull;
  if (n == 4) det_rollout_n<4>(boards, aux, table, mask, B, steps, eps, lr, gamma, seed, env_id0, ctr0, flags, stats_i, stats_f, status, row_cache);
  else det_rollout_n<5>(boards, aux, table, mask, B, steps, eps, lr, gamma, seed, env_id0, ctr0, flags, stats_i, stats_f, status, row_cache);
  return Q2048_OK;
}

// the chunked DEVICE allocator has no host form: host tables are the caller's plain memory
int q2048_table_alloc(int, size_t, q2048_slot** out) { if (out) *out = nullptr; return Q2048_ERR_UNSUPPORTED; }
int q2048_table_reserve(int, int, size_t, q2048_slot** out) { if (out) *out = nullptr; return Q2048_ERR_UNSUPPORTED; }
int q2048_table_grow_begin(q2048_slot*, int, int, q2048_growth** out) { if (out) *out = nullptr; return Q2048_ERR_UNSUPPORTED; }
int q2048_table_grow_poll(q2048_growth*) { return Q2048_ERR_UNSUPPORTED; }
int q2048_table_grow_wait(q2048_growth*, double*) { return Q2048_ERR_UNSUPPORTED; }
int q2048_table_grow_commit(q2048_growth*, int, uint32_t, q2048_slot** out, void*) { if (out) *out = nullptr; return Q2048_ERR_UNSUPPORTED; }
int q2048_table_grow_finish(q2048_growth*, int64_t*) { return Q2048_ERR_UNSUPPORTED; }
int q2048_table_grow_abort(q2048_growth*) { return Q2048_ERR_UNSUPPORTED; }
int q2048_table_grow(q2048_slot*, int, int, int, q2048_slot** out, int64_t*, void*) { if (out) *out = nullptr; return Q2048_ERR_UNSUPPORTED; }
int q2048_table_trim(q2048_slot*) { return Q2048_ERR_UNSUPPORTED; }
int q2048_table_free(q2048_slot* table) { return table == nullptr ? Q2048_OK : Q2048_ERR_UNSUPPORTED; }

int q2048_table_probe(q2048_slot* table, int cap_log2, int64_t lanes, int steps, uint64_t, void*) {
  if (int e = check_table(table, cap_log2)) return e;
  if (lanes < 0 || lanes > ((int64_t)1 << 30) || steps < 0 || steps > (1 << 16)) return Q2048_ERR_SIZE;
  return Q2048_OK;                                // placement is a property of device memory: nothing to probe here
}

int q2048_table_export(const q2048_slot* table, int cap_log2, uint64_t* keys_out, float* q_out, int64_t max_rows,
                       int key_words, int64_t* count, void*) {
  if (int e = check_table(table, cap_log2)) return e;
  if (count == nullptr) return Q2048_ERR_NULL;
  if ((keys_out == nullptr) != (q_out == nullptr) || max_rows < 0) return Q2048_ERR_SIZE;
  if (key_words != 1 && key_words != 2) return Q2048_ERR_SIZE;
  if (q_out != nullptr && !aligned16(q_out)) return Q2048_ERR_ALIGN;
  const u64 cap = 1ull << cap_log2;
  int64_t at = *count;                            // adds to *count, as the device does
  for (u64 i = 0; i < cap; ++i) {
    if (table[i].key == 0ull) continue;
    if (keys_out != nullptr && at < max_rows) {
      keys_out[at * key_words] = table[i].key;
      if (key_words == 2) keys_out[at * 2 + 1] = table[i].reserved;
      std::memcpy(q_out + 4 * at, table[i].q, 16);
    }
    ++at;
  }
  *count = at;
  return Q2048_OK;
}
// the device's line summaries, byte for byte (q2048_kernels.hip: summary_fp, k_table_summarise): four 16-bit
// fingerprints per 128-byte line, 0 = empty, the same word in the `reserved` field of the line's four slots
int q2048_table_summarise(q2048_slot* table, int cap_log2, void*) {
  if (int e = check_table(table, cap_log2)) return e;
  const int64_t lines = (int64_t)((1ull << cap_log2) >> 2);
  parallel_ranges(lines, [=](int64_t lo, int64_t hi, int) {
    for (int64_t l = lo; l < hi; ++l) {
      q2048_slot* s = table + (l << 2);
      u64 sum = 0ull;
      for (int r = 0; r < 4; ++r)
        if (s[r].key != 0ull) sum |= (((mix64(s[r].key) >> 48) & 0xffffull) | 1ull) << (16 * r);
      for (int r = 0; r < 4; ++r) s[r].reserved = sum;
    }
  });
  return Q2048_OK;
}
int q2048_table_count(const q2048_slot* table, int cap_log2, int64_t* count, void* stream) {
  return q2048_table_export(table, cap_log2, nullptr, nullptr, 0, 1, count, stream);
}
int q2048_table_import(q2048_slot* table, int cap_log2, const uint64_t* keys, const float* q, int64_t rows,
                       int key_words, uint32_t* status, void*) {
  if (int e = check_table(table, cap_log2)) return e;
  if (!keys || !q || !status) return Q2048_ERR_NULL;
  if (rows < 0 || (key_words != 1 && key_words != 2)) return Q2048_ERR_SIZE;
  if (!aligned16(q)) return Q2048_ERR_ALIGN;
  const u64 mask = (1ull << cap_log2) - 1ull;
  parallel_ranges(rows, [=](int64_t lo, int64_t hi, int) {
    for (int64_t i = lo; i < hi; ++i) {
      bool inserted;
      int64_t slot;
      u64 hash;
      if (key_words == 1) {
        const Geo<4>::Key key{(u64)keys[i]};
        hash = key_hash(key);
        slot = probe_insert(table, mask, key, hash & mask, inserted, kMaxProbe);
      } else {
        const Geo<5>::Key key{(u64)keys[2 * i], (u64)keys[2 * i + 1]};
        hash = key_hash(key);
        slot = probe_insert(table, mask, key, hash & mask, inserted, kMaxProbe);
      }
      if (slot < 0) { status_or(status, Q2048_STATUS_TABLE_FULL); continue; }
      // beyond the learning paths' probe limit: lookup / export find the row, choose / update / rollouts do not
      if (seq_pos(seq_of(hash, mask), (u64)slot) >= probe_limit(mask, kRolloutProbe)) status_or(status, Q2048_STATUS_DEEP_ROW);
      for (int a = 0; a < 4; ++a) st_f32(&table[slot].q[a], q[4 * i + a]);
    }
  });
  return Q2048_OK;
}

int q2048_legal_moves(const uint8_t* boards, int64_t B, int n, uint8_t* mask_out, void*) {
  if (int e = check_batch(B, n)) return e;
  if (!boards || !mask_out) return Q2048_ERR_NULL;
  if (!aligned16(boards)) return Q2048_ERR_ALIGN;
  parallel_ranges(B, [=](int64_t lo, int64_t hi, int) {
    for (int64_t i = lo; i < hi; ++i) {
      uint32_t m = 0, score;
      if (n == 4) { Board b; load_board(boards, i, b); for (int a = 0; a < 4; ++a) { Board t = b; m |= (uint32_t)move(t, a, score) << a; } }
      else { Board5 b; load_board(boards, i, b); for (int a = 0; a < 4; ++a) { Board5 t = b; m |= (uint32_t)move(t, a, score) << a; } }
      mask_out[i] = (uint8_t)m;
    }
  });
  return Q2048_OK;
}
int q2048_encode_onehot(const uint8_t* boards, int64_t B, int dtype, void* out, void*) {
  if (int e = check_batch(B, 4)) return e;
  if (B > (int64_t)0x7fffffff * 4) return Q2048_ERR_SIZE;            // the device's limit (64 threads per board)
  if (!boards || !out) return Q2048_ERR_NULL;
  if (!aligned16(boards) || !aligned16(out)) return Q2048_ERR_ALIGN;
  if (dtype != 0 && dtype != 1) return Q2048_ERR_RANGE;
  parallel_ranges(B, [=](int64_t lo, int64_t hi, int) {
    for (int64_t b = lo; b < hi; ++b)
      for (int c = 0; c < 16; ++c)
        for (int cell = 0; cell < 16; ++cell) {
          const bool hit = boards[16 * b + cell] == c;       // out[b][c][r][col], cell = 4 r + col
          if (dtype == 0) static_cast<float*>(out)[(b * 16 + c) * 16 + cell] = hit ? 1.f : 0.f;
          else static_cast<uint16_t*>(out)[(b * 16 + c) * 16 + cell] = hit ? 0x3F80u : 0u;
        }
  });
  return Q2048_OK;
}

int q2048_rt_choose(const float* weights, const uint8_t* boards, int64_t B, double eps, uint64_t seed, uint64_t env_id0,
                    uint32_t ctr, uint8_t* actions, void*) {
  if (int e = check_batch(B, 4)) return e;
  if (!weights || !boards || !actions) return Q2048_ERR_NULL;
  if (!aligned16(weights) || !aligned16(boards)) return Q2048_ERR_ALIGN;
  if (!(eps >= 0.0 && eps <= 1.0)) return Q2048_ERR_RANGE;
  parallel_ranges(B, [=](int64_t lo, int64_t hi, int) {
    for (int64_t i = lo; i < hi; ++i) {
      Board b;
      load_board(boards, i, b);
      const Draws x = draws(seed, env_id0 + (uint64_t)i, ctr, kStreamStep);
      int act;
      if (draw_uniform(x.x0) < eps) act = draw_action(x.x1);
      else { const Row q = rt_q(rt_gather(weights, b)); act = argmax4(q.q0, q.q1, q.q2, q.q3); }
      actions[i] = (uint8_t)act;
    }
  });
  return Q2048_OK;
}
int q2048_rt_lookup(const float* weights, const uint8_t* boards, int64_t B, float* q_out, void*) {
  if (int e = check_batch(B, 4)) return e;
  if (!weights || !boards || !q_out) return Q2048_ERR_NULL;
  if (!aligned16(weights) || !aligned16(boards) || !aligned16(q_out)) return Q2048_ERR_ALIGN;
  parallel_ranges(B, [=](int64_t lo, int64_t hi, int) {
    for (int64_t i = lo; i < hi; ++i) {
      Board b;
      load_board(boards, i, b);
      const Row q = rt_q(rt_gather(weights, b));
      q_out[4 * i] = q.q0; q_out[4 * i + 1] = q.q1; q_out[4 * i + 2] = q.q2; q_out[4 * i + 3] = q.q3;
    }
  });
  return Q2048_OK;
}
int q2048_rt_update(float* weights, const uint8_t* boards_s, const uint8_t* actions, const float* reward,
                    const uint8_t* boards_s2, const uint8_t* done, int64_t B, double lr, double gamma, uint32_t* status,
                    void*) {
  if (int e = check_batch(B, 4)) return e;
  if (!weights || !boards_s || !actions || !reward || !boards_s2 || !done || !status) return Q2048_ERR_NULL;
  if (!aligned16(weights) || !aligned16(boards_s) || !aligned16(boards_s2)) return Q2048_ERR_ALIGN;
  if (!(lr == lr) || !(gamma == gamma)) return Q2048_ERR_RANGE;
  parallel_ranges(B, [=](int64_t lo, int64_t hi, int) {
    for (int64_t i = lo; i < hi; ++i) {
      const int act = actions[i];
      if (act > 3) { status_or(status, Q2048_STATUS_BAD_ACTION); continue; }
      Board b_s, b_n;
      load_board(boards_s, i, b_s);
      load_board(boards_s2, i, b_n);
      const Row qn = rt_q(rt_gather(weights, b_n));
      const RtRows es = rt_gather(weights, b_s);
      const float d = rt_delta(row_get(rt_q(es), act), reward[i], max4(qn.q0, qn.q1, qn.q2, qn.q3), done[i] != 0, lr, gamma);
      rt_scatter(weights, b_s, es, act, d);
    }
  });
  return Q2048_OK;
}
int q2048_rt_fused_rollout(uint8_t* boards, q2048_aux* aux, float* weights, int64_t B, int64_t steps, double eps,
                           double lr, double gamma, uint64_t seed, uint64_t env_id0, uint32_t ctr0, int64_t* stats_i,
                           double* stats_f, uint32_t* status, void*) {
  if (int e = check_batch(B, 4)) return e;
  if (!boards || !aux || !weights || !status) return Q2048_ERR_NULL;
  if (!aligned16(boards) || !aligned16(aux) || !aligned16(weights)) return Q2048_ERR_ALIGN;
  if (steps < 0 || steps > (1 << 30)) return Q2048_ERR_SIZE;
  if (!(eps >= 0.0 && eps <= 1.0) || !(lr == lr) || !(gamma == gamma)) return Q2048_ERR_RANGE;
  if (B == 0 || steps == 0) return Q2048_OK;
  const int T = threads_for(B);
  std::vector<Stats> parts((size_t)T);
  Stats* sp = parts.data();
  const int used = parallel_ranges(B, [=](int64_t lo, int64_t hi, int tid) {
    Stats& st = sp[tid];
    for (int64_t i = lo; i < hi; ++i) {
      const uint64_t id = env_id0 + (uint64_t)i;
      Board b;
      load_board(boards, i, b);
      Aux a = ld_aux(aux, i);
      double reward_sum = 0.0;
      RtRows es = rt_gather(weights, b);
      for (int64_t t = 0; t < steps; ++t) {
        const Draws x = draws(seed, id, ctr0 + (uint32_t)t, kStreamStep);
        const Board s = b;
        const Row q = rt_q(es);
        bool explored;
        const int act = eps_greedy(eps, x.x0, x.x1, q.q0, q.q1, q.q2, q.q3, explored);
        const StepOut o = env_step(b, a, act, x.x2, x.x3);
        RtRows en = rt_gather(weights, b);
        const Row qn = rt_q(en);
        const float d = rt_delta(row_get(q, act), o.reward, max4(qn.q0, qn.q1, qn.q2, qn.q3), o.done != 0, lr, gamma);
        rt_scatter(weights, s, es, act, d);
        st.i[Q2048_ST_VALID] += o.valid != 0;
        st.i[Q2048_ST_EXPLORE] += explored;
        reward_sum += (double)o.reward;
        if (o.done) {
          st.i[Q2048_ST_EPISODES] += 1;
          st.episode(a, o.max_log2);
          begin_episode(b, a, seed, id);
          es = rt_gather(weights, b);
        } else {
          const uint32_t ib[4] = {pack_row(b.r0), pack_row(b.r1), pack_row(b.r2), pack_row(b.r3)};
          const uint32_t is[4] = {pack_row(s.r0), pack_row(s.r1), pack_row(s.r2), pack_row(s.r3)};
          for (int r = 0; r < 4; ++r)
            if (ib[r] == is[r]) row_set(en.e[r], act, row_get(es.e[r], act) + d);
          es = en;
        }
      }
      st.i[Q2048_ST_STEPS] += (uint64_t)steps;
      st.f[Q2048_SF_REWARD] += reward_sum;
      store_board(boards, i, b);
      st_aux(aux, i, a);
    }
  });
  stats_merge(parts, used, stats_i, stats_f);
  return Q2048_OK;
}

}  // extern "C"
