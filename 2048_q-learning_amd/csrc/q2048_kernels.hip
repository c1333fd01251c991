// q2048_kernels.hip -- gfx950 kernels + the C ABI of include/q2048.h.
//
// Layout in HBM (all caller-owned):
//   boards  uint8[B][n*n]  4x4: one dwordx4 per lane, perfectly coalesced (1 KiB per wave access)
//                          5x5: 25 B per board, unpadded; a block moves its 6400 contiguous
//                          bytes with 16-byte accesses through LDS and each lane picks its 25
//   aux     16 B per env   one dwordx4 per lane
//   table   32 B slots     {u64 key, f32 q[4], u64 key_hi}; random access, one 128-B line (four
//                          slots) per probe; the probe sequence visits the line's other three
//                          slots before it moves to the next line (`Seq`)
// One board per lane.  Boards, aux and the carried Q row live in VGPRs for a whole launch;
// per-step boolean statistics are wave ballots accumulated in SGPRs, rare per-episode
// statistics go through LDS, and each block ends with one global atomic per statistic.
// Every kernel is written once over the board geometry (template <int N>, N = 4 or 5).
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "q2048.h"
#include "q2048_core5.hpp"

// Measurement variants (write-mode / ablation / sort-width / boards-per-thread bits in flags 8..23)
// exist only in builds with -DQ2048_EXPERIMENTS (tools/variants/libq2048_hip_exp.so, used by tools/
// and by the tests that force crowded sort runs).  The shipped library takes the ABI flags of
// q2048.h and nothing else: any other bit is Q2048_ERR_FLAGS.
#ifdef Q2048_EXPERIMENTS
#define Q2048_XBITS(flags, shift, mask) (((flags) >> (shift)) & (mask))
#else
#define Q2048_XBITS(flags, shift, mask) 0u
#endif

namespace {
using namespace q2048;

constexpr int kBlock = 256;
// Probe limits: beyond them a lookup reads "absent" and an update drops (counted; status TABLE_FULL).  They are
// what makes a probe of a FULL table end, not a load policy: at load 0.93 the longest cluster of a 2^22-slot
// table is already thousands of slots (ln n / (a - 1 - ln a)), and round 3's limit of 256 made "full" mean
// "load ~0.85" (a racing import at load 0.93 dropped rows).  Bulk moves of rows (import, the rehash of a growth)
// and q2048_q_lookup take kMaxProbe = 2^14 slots: they must place and find every row of a table the caller sized.
// The learning paths (rollouts, choose, update) take kRolloutProbe = 2^10: a lane of a rollout probes twice per
// step, and on a fixed table that has filled up 2^14 dependent 16-byte loads per probe made a launch orders of
// magnitude slower before TABLE_FULL became visible; 2^10 bounds that at ~load 0.93, where an
// absent key's expected probe is 100 slots already.  (Or the whole table if smaller.)
constexpr uint32_t kMaxProbe = 1u << 14, kRolloutProbe = 1u << 10;
constexpr int kMaxCas = 16;      // TD compare-and-swap attempts before the update is simply stored

static_assert(sizeof(q2048_aux) == 16 && sizeof(q2048_slot) == 32, "ABI layout");
static_assert(sizeof(q2048_episode) == 48, "ABI layout");
static_assert(offsetof(q2048_slot, q) == 8 && offsetof(q2048_slot, reserved) == 24, "ABI layout");
static_assert(sizeof(Aux) == sizeof(q2048_aux), "core/ABI aux mismatch");
static_assert(Q2048_MIRROR_SEQ == Q2048_NSTAT_I + Q2048_NSTAT_F && Q2048_MIRROR_WORDS == Q2048_MIRROR_SEQ + 1, "ABI layout");

using u64 = unsigned long long;

// ---------------------------------------------------------------------------------------------
// geometry: board type, state key, HBM image
// ---------------------------------------------------------------------------------------------
template <int N> struct Geo;
template <> struct Geo<4> {
  using BoardT = Board;
  struct Key { u64 k0; };
};
template <> struct Geo<5> {
  using BoardT = Board5;
  struct Key { u64 k0, k1; };
};

// LDS staging of 5x5 boards, PER WAVE: the 64 boards of a wave are 1600 contiguous bytes of HBM
// (100 x 16 B, 16-byte aligned: 1600 = 100 * 16 and a block starts at a multiple of 6400), moved
// with 16-byte accesses through the wave's own 1600-byte slice of LDS.  A wave is its own
// synchronisation domain -- its LDS operations execute in program order -- so there is no
// __syncthreads anywhere on this path (round 2 staged per block, with two barriers per direction,
// and the single-step 5x5 kernel ran at 0.40 of the roofline).  Empty for 4x4.
template <int N, int WAVES = kBlock / 64> struct Stage { };
template <int WAVES> struct Stage<5, WAVES> { uint4 v[WAVES][100]; };
typedef __attribute__((address_space(3))) uint8_t lds_u8;   // a byte in LDS (volatile accesses through a generic pointer would be flat_*)
__device__ __forceinline__ void wave_lds_sync() {      // this wave's LDS writes are done before its next LDS reads
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Every lane of a wave calls these together (lanes with i >= B get an all-zero board and store nothing).
template <int WAVES>
__device__ __forceinline__ Board load_board(const uint8_t* boards, int64_t i, int64_t B, Stage<4, WAVES>&) {
  if (i >= B) return Board{0u, 0u, 0u, 0u};
  const uint4 v = reinterpret_cast<const uint4*>(boards)[i];
  return Board{v.x, v.y, v.z, v.w};
}
template <int WAVES>
__device__ __forceinline__ void store_board(uint8_t* boards, int64_t i, int64_t B, const Board& b,
                                            Stage<4, WAVES>&) {
  if (i < B) reinterpret_cast<uint4*>(boards)[i] = make_uint4(b.r0, b.r1, b.r2, b.r3);
}
template <int WAVES>
__device__ __forceinline__ Board5 load_board(const uint8_t* boards, int64_t i, int64_t B, Stage<5, WAVES>& st) {
  const int w = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
  const int64_t base = i - lane;                        // first env of the wave
  const int64_t left = B - base;
  Board5 b;
  clear(b);
  if (left <= 0) return b;                              // the whole wave is past the batch
  const int bytes = (int)(left < 64 ? left : 64) * 25;
  const uint8_t* src = boards + base * 25;              // 16-byte aligned (see Stage<5>)
  uint4* lds = st.v[w];
  uint8_t* l8 = reinterpret_cast<uint8_t*>(lds);
  for (int c = lane; c * 16 < bytes; c += 64) {
    if (c * 16 + 16 <= bytes) {
      uint4 v = reinterpret_cast<const uint4*>(src)[c];
      v.x &= 0x1f1f1f1fu; v.y &= 0x1f1f1f1fu; v.z &= 0x1f1f1f1fu; v.w &= 0x1f1f1f1fu;   // log2 tile <= 31
      lds[c] = v;
    } else for (int k = c * 16; k < bytes; ++k) l8[k] = src[k] & 31u;  // ragged tail of the batch's last wave
  }
  wave_lds_sync();
  // One LDS byte read and one v_lshl_or per cell (the LDS pipe is idle otherwise; taking the 25
  // bytes as seven words and pulling the fields out of them costs the VALU, which bounds the 5x5
  // env step, twice the instructions).  volatile: the compiler would merge the reads into words.
  if (i < B) {
    const volatile lds_u8* mine = (const volatile lds_u8*)(l8 + lane * 25);
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      uint32_t w = mine[5 * r];
#pragma unroll
      for (int c = 1; c < 5; ++c) w |= (uint32_t)mine[5 * r + c] << (6 * c);
      b.r[r] = w;
    }
  }
  wave_lds_sync();                                      // the slice is written again by store_board
  return b;
}
template <int WAVES>
__device__ __forceinline__ void store_board(uint8_t* boards, int64_t i, int64_t B, const Board5& b,
                                            Stage<5, WAVES>& st) {
  const int w = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
  const int64_t base = i - lane;
  const int64_t left = B - base;
  if (left <= 0) return;
  const int bytes = (int)(left < 64 ? left : 64) * 25;
  uint8_t* dst = boards + base * 25;
  uint4* lds = st.v[w];
  uint8_t* l8 = reinterpret_cast<uint8_t*>(lds);
  if (i < B) {                                          // one v_bfe and one LDS byte write per cell
    volatile lds_u8* mine = (volatile lds_u8*)(l8 + lane * 25);
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
      for (int c = 0; c < 5; ++c) mine[5 * r + c] = (uint8_t)(field5(b.r[r], c) & 31u);
  }
  wave_lds_sync();
  for (int c = lane; c * 16 < bytes; c += 64) {
    if (c * 16 + 16 <= bytes) reinterpret_cast<uint4*>(dst)[c] = lds[c];
    else for (int k = c * 16; k < bytes; ++k) dst[k] = l8[k];
  }
  wave_lds_sync();
}

__device__ __forceinline__ Aux ld_aux(const q2048_aux* aux, int64_t i) {
  const uint4 v = reinterpret_cast<const uint4*>(aux)[i];
  return words_to_aux(Words4{v.x, v.y, v.z, v.w});
}
__device__ __forceinline__ void st_aux(q2048_aux* aux, int64_t i, const Aux& a) {
  const Words4 w = aux_to_words(a);
  reinterpret_cast<uint4*>(aux)[i] = make_uint4(w.w0, w.w1, w.w2, w.w3);
}

// state keys.  Independent mode salts the key with the env id (private rows per env).
__device__ __forceinline__ Geo<4>::Key state_key(const Board& b, u64 salt, uint32_t* status) {
  bool ov;
  u64 k = pack_key(b, ov) ^ salt;
  if (ov) atomicOr(status, Q2048_STATUS_TILE_OVERFLOW);
  return Geo<4>::Key{k == 0ull ? 1ull : k};  // 0 marks an empty slot
}
__device__ __forceinline__ Geo<5>::Key state_key(const Board5& b, u64 salt, uint32_t*) {
  const Key5 k = pack_key(b);  // both words carry bit 63, so neither is ever 0
  return Geo<5>::Key{k.k0 ^ (salt & 0x7fffffffffffffffull),
                     k.k1 ^ (mix64(salt) & 0x3fffffffffffffffull)};
}
__device__ __forceinline__ bool key_eq(const Geo<4>::Key& a, const Geo<4>::Key& b) { return a.k0 == b.k0; }
__device__ __forceinline__ bool key_eq(const Geo<5>::Key& a, const Geo<5>::Key& b) {
  return a.k0 == b.k0 && a.k1 == b.k1;
}
// 64 hash bits of a key: the low ones choose the home slot, the top 16 the deterministic mode's
// sort bucket (a function of the state alone -- not of where its row ended up)
__device__ __forceinline__ u64 key_hash(const Geo<4>::Key& k) { return mix64(k.k0); }
__device__ __forceinline__ u64 key_hash(const Geo<5>::Key& k) {
  return mix64(k.k0 ^ (k.k1 * 0x9E3779B97F4A7C15ull));
}
template <class Key>
__device__ __forceinline__ u64 key_home(const Key& k, u64 mask) { return key_hash(k) & mask; }

// Probe sequence of a key: BUCKETISED.  A miss moves the whole 128-byte line (4 slots) the probed
// slot lies in, and what a lookup costs is the number of lines it touches, so a
// collision is resolved inside the line already fetched: the sequence starts at the home slot
// (hash & mask), visits the other three slots of that line cyclically -- L2 hits -- and only then
// moves on to the next line, same order.  Plain linear probing left the line at the first collision
// of every key whose home is a line's last slot.  (Slots are 32 B; with a 128-byte aligned table
// a group of four is exactly one line.  Any 16-byte aligned table works.)
struct Seq { u64 line0, lmask; uint32_t off; };
__device__ __forceinline__ Seq seq_of(u64 hash, u64 mask) {
  return Seq{(hash & mask) >> 2, mask >> 2, (uint32_t)hash & 3u};
}
__device__ __forceinline__ u64 seq_slot(const Seq& s, uint32_t p) {   // p-th slot of the sequence
  return (((s.line0 + (u64)(p >> 2)) & s.lmask) << 2) | (u64)((s.off + p) & 3u);
}
__device__ __forceinline__ uint32_t seq_pos(const Seq& s, u64 slot) { // inverse, for a slot on the sequence
  return ((uint32_t)(((slot >> 2) - s.line0) & s.lmask) << 2) | (((uint32_t)slot - s.off) & 3u);
}

__device__ __forceinline__ uint32_t probe_limit(u64 mask, uint32_t maxp) {
  return mask >= (u64)maxp ? maxp : (uint32_t)mask + 1u;
}

// ---------------------------------------------------------------------------------------------
// hash table.  Readers use agent-scope relaxed loads (they bypass the per-CU L1, which other
// CUs' atomics never refresh); rows are claimed with a device-scope compare-and-swap on the key
// word.  Keys are written once (0 -> key) and never change, so a stale read can only miss a
// brand-new row, which reads as the zero row it still is for the reader.  5x5 keys take two
// words: one compare-and-swap claims the first, its owner publishes the second (see `confirm`).
// ---------------------------------------------------------------------------------------------
struct Row { float q0, q1, q2, q3; };

__device__ __forceinline__ u64 ld_u64(const void* p) {
  return __hip_atomic_load(const_cast<u64*>(reinterpret_cast<const u64*>(p)), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float row_get(const Row& r, int a) {
  return a == 0 ? r.q0 : a == 1 ? r.q1 : a == 2 ? r.q2 : r.q3;
}
__device__ __forceinline__ void row_set(Row& r, int a, float v) {
  r.q0 = a == 0 ? v : r.q0; r.q1 = a == 1 ? v : r.q1;
  r.q2 = a == 2 ? v : r.q2; r.q3 = a == 3 ? v : r.q3;
}

// `confirm`: the slot's first key word equals key.k0 (this lane just set it, or found it so).
// Does the slot hold `key`?  4x4: yes.  5x5 keys have a second word, and a claim is ONE
// compare-and-swap (scattered device-scope atomics are the scarcest resource of the path: two per
// new state made the 5x5 rollout 1.5x slower): the lane that won the first word publishes the
// second with a write-through store at once; a lane that finds the first word equal and the
// second still 0 waits for it.  Waiting is safe here because
//   - a lane is waited for only from the moment its compare-and-swap has succeeded, and its
//     store is the next thing its wave issues: ahead of the wait of any wave-mate that lost the
//     same compare-and-swap (`wave_barrier` keeps the compiler from moving the two apart), and
//     the owner itself waits for nobody in between -- so there is no cycle of waits;
//   - the wait polls with a memory-side atomic (an L2-served load could show a stale 0 forever);
//   - it is bounded: a lane that gives up counts in `g_claim_timeouts` (tests: always 0) and
//     treats the slot as someone else's.
// The second word goes 0 -> value exactly once, written only by the owner of the first.
__device__ unsigned long long g_claim_timeouts;
constexpr int kMaxAwait = 1 << 16;

__device__ __forceinline__ u64 await_second(q2048_slot* s) {
  u64* p = reinterpret_cast<u64*>(&s->reserved);
  for (int spin = 0; spin < kMaxAwait; ++spin) {
    const u64 hi = atomicCAS(p, 0ull, 0ull);  // reads at the memory side; writes nothing new
    if (hi != 0ull) return hi;
    __builtin_amdgcn_s_sleep(4);
  }
  atomicAdd(&g_claim_timeouts, 1ull);
  return 0ull;
}
__device__ __forceinline__ bool confirm(q2048_slot*, const Geo<4>::Key&, bool won, bool& completed) {
  completed = won;
  return true;
}
__device__ __forceinline__ bool confirm(q2048_slot* s, const Geo<5>::Key& key, bool won, bool& completed) {
  completed = won;
  if (won)
    __hip_atomic_store(reinterpret_cast<u64*>(&s->reserved), key.k1, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  __builtin_amdgcn_wave_barrier();
  if (won) return true;
  u64 hi = ld_u64(&s->reserved);
  if (hi == 0ull) hi = await_second(s);
  return hi == key.k1;
}

// Lookup.  Returns the slot index (>= 0) when the key is present; otherwise ~h (< 0) where h is
// the empty slot that ended the probe -- the place an insert of this key would claim -- or
// kNoSlot when the probe limit was hit.  `created` stays false (lookups create nothing).
constexpr int64_t kNoSlot = INT64_MIN;
// The cost of the table is the NUMBER of scattered requests a lane issues, whatever line they
// hit, so the probe reads {key, q0, q1} with ONE
// 16-byte load and the other half of the slot only on a key match.  The load carries sc1 like
// the agent-scope 8-byte loads it replaces (bypasses the per-CU L1, keeps the default L2 /
// Infinity Cache policy: a non-temporal load made the next step's claim of the same line twice
// as slow).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 ld16_agent(const void* p) {  // waited for in place
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ int64_t probe_find(const q2048_slot* table, u64 mask,
                                              const Geo<4>::Key& key, Row& row, bool& created,
                                              uint32_t maxp = kRolloutProbe) {
  const Seq sq = seq_of(key_hash(key), mask);
  created = false;
  row = Row{0.f, 0.f, 0.f, 0.f};
  for (uint32_t p = 0, lim = probe_limit(mask, maxp); p < lim; ++p) {
    const u64 i = seq_slot(sq, p);
    const u32x4 v = ld16_agent(&table[i]);
    const u64 k = (u64)v.x | ((u64)v.y << 32);
    if (k == key.k0) {
      const u64 hi = ld_u64(&table[i].q[2]);
      row = Row{bits_f32(v.z), bits_f32(v.w), bits_f32((uint32_t)hi), bits_f32((uint32_t)(hi >> 32))};
      return (int64_t)i;
    }
    if (k == 0ull) return ~(int64_t)i;
  }
  return kNoSlot;
}

// 5x5: {key, q0, q1} first; on a first-word match {q2, q3, second key word} -- two requests for
// a hit instead of four (key, second word, two row halves).  A slot whose second word is still 0
// is being created by its owner: wait for the word (`confirm`).
__device__ __forceinline__ int64_t probe_find(const q2048_slot* table, u64 mask,
                                              const Geo<5>::Key& key, Row& row, bool& created,
                                              uint32_t maxp = kRolloutProbe) {
  const Seq sq = seq_of(key_hash(key), mask);
  created = false;
  row = Row{0.f, 0.f, 0.f, 0.f};
  for (uint32_t p = 0, lim = probe_limit(mask, maxp); p < lim; ++p) {
    const u64 i = seq_slot(sq, p);
    const u32x4 a = ld16_agent(&table[i]);
    const u64 k = (u64)a.x | ((u64)a.y << 32);
    if (k == 0ull) return ~(int64_t)i;
    if (k == key.k0) {
      const u32x4 b = ld16_agent(&table[i].q[2]);
      u64 hi = (u64)b.z | ((u64)b.w << 32);
      if (hi == 0ull) hi = await_second(const_cast<q2048_slot*>(&table[i]));  // being created right now
      if (hi == key.k1) {
        row = Row{bits_f32(a.z), bits_f32(a.w), bits_f32(b.x), bits_f32(b.y)};
        return (int64_t)i;
      }
    }
  }
  return kNoSlot;
}

// ---------------------------------------------------------------------------------------------
// LINE SUMMARIES (4x4 tables with a closed key set: Q2048_FLAG_LINE_SUMMARY after q2048_table_summarise).
// What a lookup costs is the number of requests its lane sends to the L2, hits included: on a table at load 0.5 the
// slot-by-slot probe of an ABSENT state -- most lookups of a run that has outgrown its table -- sends 2.35 (one miss,
// then the line's next slots until one is empty) and the step takes 47.9 us; with ONE request per lookup it takes
// 27.1 (profiles/r06_one_request.jsonl: the arithmetic alone is 20.9).  A 4x4 slot has 8 bytes to spare (`reserved`:
// the second key word of 5x5).  Once the key set is closed they can hold what the probe wants to know about the whole
// LINE: four 16-bit fingerprints, one per slot of the 128-byte line (0 = empty), the same word in all four slots.  The
// probe then reads the second half of the slot its sequence enters the line at -- {q2, q3, summary}: one request --
// and knows where the sequence ends: at the first slot, in its order, that is empty (absent: done) or carries the
// key's fingerprint (one more request, that slot's head, settles it; a false match, 2^-15 per occupied slot, costs
// that request and nothing else).  Written by one streaming pass when the key set closes; meaningless as soon as a row
// is created (the caller's contract, include/q2048.h).
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ u64 summary_fp(u64 hash) { return ((hash >> 48) & 0xffffull) | 1ull; }   // never 0
__global__ __launch_bounds__(kBlock) void k_table_summarise(q2048_slot* table, u64 lines) {
  for (u64 l = (u64)blockIdx.x * kBlock + threadIdx.x; l < lines; l += (u64)gridDim.x * kBlock) {
    q2048_slot* s = table + (l << 2);
    u64 sum = 0ull;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const u64 k = s[r].key;
      if (k != 0ull) sum |= summary_fp(mix64(k)) << (16 * r);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r].reserved = sum;
  }
}
__device__ __forceinline__ int64_t probe_find_summary(const q2048_slot* table, u64 mask, const Geo<4>::Key& key,
                                                      Row& row, bool& created, uint32_t maxp = kRolloutProbe) {
  const u64 hash = key_hash(key);
  const Seq sq = seq_of(hash, mask);
  const u64 fp = summary_fp(hash);
  created = false;
  row = Row{0.f, 0.f, 0.f, 0.f};
  for (uint32_t p = 0, lim = probe_limit(mask, maxp); p < lim; p += 4u) {
    const u64 base = ((sq.line0 + (u64)(p >> 2)) & sq.lmask) << 2;
    const u32x4 v = ld16_agent(&table[base | (u64)sq.off].q[2]);          // {q2, q3, summary} of the slot the sequence enters at
    const u64 sum = (u64)v.z | ((u64)v.w << 32);
    // bit r: slot r of the line is empty / carries the key's fingerprint
    uint32_t e = 0u, m = 0u;
#pragma unroll
    for (uint32_t r = 0; r < 4u; ++r) {
      const u64 f = (sum >> (16u * r)) & 0xffffull;
      e |= (uint32_t)(f == 0ull) << r;
      m |= (uint32_t)(f == fp) << r;
    }
    // the line's slots in the sequence's order (it enters at `off`): position j <-> slot (off + j) & 3
    uint32_t ev = (((e | m) | ((e | m) << 4)) >> sq.off) & 15u;
    while (ev != 0u) {
      const uint32_t j = (uint32_t)__builtin_ctz(ev), r = (sq.off + j) & 3u;
      const u64 i = base | (u64)r;
      if ((e >> r) & 1u) return ~(int64_t)i;                                // the sequence ends here: absent
      const u32x4 h = ld16_agent(&table[i]);                                // a candidate: its head settles it
      if (((u64)h.x | ((u64)h.y << 32)) == key.k0) {
        u64 hi = (u64)v.x | ((u64)v.y << 32);
        if (j != 0u) hi = ld_u64(&table[i].q[2]);
        row = Row{bits_f32(h.z), bits_f32(h.w), bits_f32((uint32_t)hi), bits_f32((uint32_t)(hi >> 32))};
        return (int64_t)i;
      }
      ev &= ev - 1u;                                                        // another key with this fingerprint
    }
  }
  return kNoSlot;
}
template <bool SUMMARY, class Key>
__device__ __forceinline__ int64_t probe_find_as(const q2048_slot* table, u64 mask, const Key& key, Row& row, bool& created) {
  if constexpr (SUMMARY && sizeof(Key) == sizeof(u64)) return probe_find_summary(table, mask, key, row, created);
  else return probe_find(table, mask, key, row, created);
}

// Find-or-create starting at slot `start` of the key's sequence (the hint of a failed probe_find,
// or the home slot).  The first access is the claiming compare-and-swap itself: the slot was empty
// a moment ago.  Returns the slot index or kNoSlot (probe limit: the caller drops the update).
template <class Key>
__device__ __forceinline__ int64_t probe_insert(q2048_slot* table, u64 mask, const Key& key, u64 start,
                                                bool& inserted, uint32_t maxp = kRolloutProbe) {
  const Seq sq = seq_of(key_hash(key), mask);
  u64 i = start & mask;
  inserted = false;
  u64 k = 0ull;                                      // the hinted slot: straight to the compare-and-swap
  for (uint32_t p = seq_pos(sq, i), lim = probe_limit(mask, maxp); p < lim; i = seq_slot(sq, ++p), k = ld_u64(&table[i].key)) {
    if (k == 0ull) k = atomicCAS(reinterpret_cast<u64*>(&table[i].key), 0ull, key.k0);
    if ((k == 0ull || k == key.k0) && confirm(&table[i], key, k == 0ull, inserted)) return (int64_t)i;
  }
  inserted = false;
  return kNoSlot;
}

// A row claim in flight (4x4).  The compare-and-swap is issued when the probe finds the state
// absent and its result is consumed one step later, when s' has become s, so the round trip
// hides behind the next step's arithmetic.  5x5 claims in place: the owner has to publish the
// second key word right after the first (the shorter that window, the fewer lanes ever wait),
// and its rollout is bound by request throughput, not latency (pipelining measured +-0).
struct Claim { u64 ret; u64 at; bool active; };

__device__ __forceinline__ int64_t claim_issue(q2048_slot* table, u64, int64_t slot,
                                               const Geo<4>::Key& key, Claim& c, bool&) {
  c.active = slot < 0 && slot != kNoSlot;
  if (c.active) {
    c.at = (u64)~slot;
    c.ret = atomicCAS(reinterpret_cast<u64*>(&table[c.at].key), 0ull, key.k0);
  }
  return slot;
}
__device__ __forceinline__ int64_t claim_issue(q2048_slot* table, u64 mask, int64_t slot,
                                               const Geo<5>::Key& key, Claim& c, bool& inserted) {
  c.active = false;
  if (slot < 0 && slot != kNoSlot) return probe_insert(table, mask, key, (u64)~slot, inserted);
  return slot;
}
template <class Key>
__device__ __forceinline__ int64_t claim_resolve(q2048_slot* table, u64 mask, const Key& key, Claim& c,
                                                 int64_t slot, bool& inserted) {
  if (!c.active) return slot;
  c.active = false;
  if (c.ret == 0ull) { inserted = true; return (int64_t)c.at; }
  if (c.ret == key.k0) return (int64_t)c.at;
  const Seq sq = seq_of(key_hash(key), mask);                    // another key took the slot: go on
  return probe_insert(table, mask, key, seq_slot(sq, seq_pos(sq, c.at) + 1u), inserted);
}

// update_q_value on one entry (Agent/main.py:43) against its CURRENT value.  `guess` is the
// value this lane last saw.  Write modes:
//   STORE (default)  one 4-byte store of the new value: when several lanes update the same
//          (s, a) at the same time the last writer wins (never a torn value).
//   CAS (Q2048_FLAG_TD_CAS)  compare-and-swap loop: a failed swap returns the live value and
//          the update is recomputed from it, so concurrent updates of one (s, a) serialise -- for
//          up to kMaxCas attempts; an entry contended beyond that takes the update as a plain
//          store, as the default mode would (last writer wins), and the event is counted
//          (Q2048_ST_CAS_FALLBACK): strict mode's guarantee is bounded, and a run can tell by how much.
//   NONE   nothing is written (Q2048_FLAG_NO_LEARN).
//   The other modes are measurement variants (Q2048_EXPERIMENTS builds only).
enum : uint32_t { kTdStorePlain = 0, kTdCas = 1, kTdStoreSc1 = 2, kTdStoreNt = 3, kTdNone = 4,
                  kTdAdd = 6 };
struct TdCounters { uint32_t retries, fallbacks; };
__device__ __forceinline__ float td_update(q2048_slot* slot, int a, float guess, float reward,
                                           float max_next, bool done, double lr, double gamma,
                                           TdCounters& ctrs, uint32_t mode, int max_cas = kMaxCas) {
  unsigned int* addr = reinterpret_cast<unsigned int*>(&slot->q[a]);
  unsigned int expect = f32_bits(guess);
  float nq = td_value(guess, reward, max_next, done, lr, gamma);
  if (mode == kTdStorePlain) { *addr = f32_bits(nq); return nq; }
  if (mode == kTdNone) return nq;
#ifdef Q2048_EXPERIMENTS
  if (mode == kTdStoreSc1) {
    __hip_atomic_store(addr, f32_bits(nq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return nq;
  }
  if (mode == kTdStoreNt) { __builtin_nontemporal_store(f32_bits(nq), addr); return nq; }
  if (mode == kTdAdd) { atomicAdd(&slot->q[a], nq - guess); return nq; }
#endif
  for (int it = 0; it < max_cas; ++it) {
    const unsigned int prev = atomicCAS(addr, expect, f32_bits(nq));
    if (prev == expect) return nq;
    ++ctrs.retries;
    expect = prev;
    nq = td_value(bits_f32(expect), reward, max_next, done, lr, gamma);
  }
  // kMaxCas lost races in a row: the entry is being rewritten every few hundred nanoseconds.
  // Spinning on costs far more than exactness here is worth -- unbounded, a few such entries (the
  // opening states at low epsilon) stretched every wave's step 4x (181 -> 71 us per 1 Mi boards at
  // eps = 0.01, profiles/r02_strict_td_retries.jsonl) -- so the update is written as the default
  // mode writes it, and counted.  (An atomic add of the increment instead -- no sample lost -- was
  // tried in round 3: the same hot entries then serialise at the memory side, 84 -> 267 us per step
  // from reset at eps = 0.01, profiles/r03_batch_sweep.jsonl history.)
  ++ctrs.fallbacks;
  *addr = f32_bits(nq);
  return nq;
}
// Deferred TD writes: while a lane stays in one state (invalid moves: the board did not change)
// its updates of that row live in registers (`q`) and `pend` has one bit per action written;
// they reach the table when the lane leaves the state, finishes the episode or the launch ends.
// One 4-byte store per entry, as td_update's STORE mode: only the moment changes.
__device__ __forceinline__ void flush_pending(q2048_slot* slot, const Row& q, uint32_t& pend) {
  unsigned int* w = reinterpret_cast<unsigned int*>(slot->q);
  if (pend & 1u) w[0] = f32_bits(q.q0);
  if (pend & 2u) w[1] = f32_bits(q.q1);
  if (pend & 4u) w[2] = f32_bits(q.q2);
  if (pend & 8u) w[3] = f32_bits(q.q3);
  pend = 0u;
}
__device__ __forceinline__ uint32_t td_mode_of(uint32_t flags) {
  const uint32_t x = Q2048_XBITS(flags, 8, 15u);  // measurement variants (experiment builds)
  return x ? x : ((flags & Q2048_FLAG_TD_CAS) ? kTdCas : kTdStorePlain);
}

__device__ __forceinline__ uint32_t wave_count(bool pred) {
  return (uint32_t)__popcll(__ballot(pred));
}
__device__ __forceinline__ bool wave_leader() {  // first active lane of the wave
  const u64 active = __ballot(true);
  return (u64)(threadIdx.x & 63) == (u64)__ffsll((long long)active) - 1ull;
}

// block-level statistics staging (LDS) and flush
struct BlockStats {
  u64 i[Q2048_NSTAT_I];
  double f[Q2048_NSTAT_F];
};
__device__ __forceinline__ void stats_clear(BlockStats& s) {
  if (threadIdx.x < Q2048_NSTAT_I) s.i[threadIdx.x] = 0ull;
  if (threadIdx.x < Q2048_NSTAT_F) s.f[threadIdx.x] = 0.0;
  __syncthreads();
}
__device__ __forceinline__ void stats_flush(BlockStats& s, int64_t* gi, double* gf) {
  __syncthreads();
  if (gi != nullptr && threadIdx.x < Q2048_NSTAT_I && s.i[threadIdx.x] != 0ull)
    atomicAdd(reinterpret_cast<u64*>(gi) + threadIdx.x, s.i[threadIdx.x]);
  if (gf != nullptr && threadIdx.x < Q2048_NSTAT_F && s.f[threadIdx.x] != 0.0)
    atomicAdd(gf + threadIdx.x, s.f[threadIdx.x]);
}
// Striped statistics (deterministic step).  A block's flush is one global atomic per non-zero
// statistic, and with 4096 blocks per one-step launch all of them queue on the same few addresses:
// at 1 Mi boards that serial tail was 30 of the step's 189 us (statistics pointers NULL: 159 us,
// tools/archive/exp_stats_cost.py).  So phase 1 adds into one of kStatStripes copies chosen by block index
// (64 atomics per address and launch instead of 4096), kept in the caller's workspace, and one small
// launch at the end of the call folds the copies into the caller's vectors.
constexpr int kStatStripes = 64;
struct StatStripe { u64 i[Q2048_NSTAT_I]; double f[Q2048_NSTAT_F]; };
__device__ __forceinline__ void stats_flush_striped(BlockStats& s, StatStripe* stripes) {
  __syncthreads();
  if (stripes == nullptr) return;
  StatStripe& mine = stripes[blockIdx.x % kStatStripes];
  if (threadIdx.x < Q2048_NSTAT_I && s.i[threadIdx.x] != 0ull) atomicAdd(&mine.i[threadIdx.x], s.i[threadIdx.x]);
  if (threadIdx.x < Q2048_NSTAT_F && s.f[threadIdx.x] != 0.0) atomicAdd(&mine.f[threadIdx.x], s.f[threadIdx.x]);
}
__global__ __launch_bounds__(64) void k_stats_fold(const StatStripe* stripes, int64_t* gi, double* gf) {
  const int t = (int)threadIdx.x;
  if (t < Q2048_NSTAT_I && gi != nullptr) {
    u64 sum = 0ull;
    for (int k = 0; k < kStatStripes; ++k) sum += stripes[k].i[t];
    if (sum) atomicAdd(reinterpret_cast<u64*>(gi) + t, sum);
  } else if (t >= Q2048_NSTAT_I && t < Q2048_NSTAT_I + Q2048_NSTAT_F && gf != nullptr) {
    double sum = 0.0;                                     // stripes in index order: a fixed summation order
    for (int k = 0; k < kStatStripes; ++k) sum += stripes[k].f[t - Q2048_NSTAT_I];
    if (sum != 0.0) atomicAdd(gf + (t - Q2048_NSTAT_I), sum);
  }
}
// Statistics mirror (optional, fused rollout): the LAST block of a launch to finish copies the two
// statistics vectors, as they stand after every block's flush, into `mirror` -- a buffer the HOST can
// read (pinned host memory mapped into the device's address space) -- so a caller that waits for the
// launch anyway needs no device-to-host copy behind it (a 288-byte copy queued on the stream cost the
// bench's 1 ms region 15-18 us, profiles/r03_region_overhead.txt).  "Last" is decided by a ticket: every
// block, once its own statistics atomics have been PERFORMED, takes a number; the block that draws
// gridDim.x - 1 knows all the others are in, reads the vectors where the atomics were performed
// (agent-scope loads) and writes them out with system-scope stores, followed by the number of mirrored
// launches so far (ticket[1], mirror[Q2048_MIRROR_SEQ]: how the host tells a fresh mirror from the last
// one), and resets the ticket for the next launch.  ticket = device uint32[2], zero before its first use.
//
// No fences.  An agent-scope release fence on this chip writes back every dirty line of the XCD's L2, and
// this kernel's L2s are full of freshly written Q values: `__threadfence()` per block made a 20-step launch
// 2.3 ms instead of 0.95 (gpurun_out/r04a).  Order comes from the atomics themselves: the flush uses the
// RETURNING form and waits for the answers -- a device-scope atomic has been performed at the memory side
// when its old value is back -- before the block's ticket atomic is issued, to the same coherence point.
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void stats_flush_acked(BlockStats& s, int64_t* gi, double* gf) {
  __syncthreads();
  if (threadIdx.x < Q2048_NSTAT_I && s.i[threadIdx.x] != 0ull) {
    const u64 old = atomicAdd(reinterpret_cast<u64*>(gi) + threadIdx.x, s.i[threadIdx.x]);
    asm volatile("" :: "v"(old));                           // the returning form, and its answer waited for
  }
  if (threadIdx.x < Q2048_NSTAT_F && s.f[threadIdx.x] != 0.0) {
    const double old = atomicAdd(gf + threadIdx.x, s.f[threadIdx.x]);
    asm volatile("" :: "v"(old));
  }
  wait_vm();
}
__device__ __forceinline__ void stats_mirror(const int64_t* gi, const double* gf, u64* mirror, uint32_t* ticket) {
  __shared__ uint32_t my_ticket;
  __syncthreads();                                        // every thread's statistics atomics have answered
  if (threadIdx.x == 0)
    my_ticket = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (my_ticket != gridDim.x - 1u) return;
  const int t = (int)threadIdx.x;
  if (t < Q2048_NSTAT_I + Q2048_NSTAT_F) {
    const u64* src = t < Q2048_NSTAT_I ? reinterpret_cast<const u64*>(gi) + t
                                       : reinterpret_cast<const u64*>(gf) + (t - Q2048_NSTAT_I);
    const u64 v = __hip_atomic_load(const_cast<u64*>(src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(mirror + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  wait_vm();                                              // the values are out before the launch number
  __syncthreads();
  if (t == 0) {
    const uint32_t launches = __hip_atomic_load(ticket + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    __hip_atomic_store(ticket + 1, launches, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(mirror + Q2048_MIRROR_SEQ, (u64)launches, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ void episode_stats(BlockStats& s, const Aux& a, uint32_t max_l2) {
  atomicAdd(&s.i[Q2048_ST_SCORE], (u64)(int64_t)a.score);
  atomicAdd(&s.i[Q2048_ST_HIST0 + (max_l2 > 22u ? 22u : max_l2)], 1ull);
  const double ret = (double)a.ep_return;
  atomicAdd(&s.f[Q2048_SF_RETURN], ret);
  atomicAdd(&s.f[Q2048_SF_RETURN_SQ], ret * ret);
}

// ---------------------------------------------------------------------------------------------
// env kernels
// ---------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(kBlock) void k_env_init(uint8_t* boards, q2048_aux* aux, int64_t B,
                                                     uint64_t seed, uint64_t env_id0) {
  __shared__ Stage<N> st;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  typename Geo<N>::BoardT b;
  Aux a;
  init_env(b, a, seed, env_id0 + (uint64_t)i);
  store_board(boards, i, B, b, st);
  if (i < B) st_aux(aux, i, a);
}

// Masked reset: a lane whose mask byte is 0 touches nothing (a 1 Mi-board call after a step resets
// under 1 % of the lanes: the launch reads the mask and little else).  5x5 boards move through the
// wave's LDS slice, so there a wave leaves together when none of its lanes is masked.
template <int N>
__global__ __launch_bounds__(kBlock) void k_env_reset(uint8_t* boards, q2048_aux* aux,
                                                      const uint8_t* mask, int64_t B, uint64_t seed,
                                                      uint64_t env_id0, uint32_t flags) {
  __shared__ Stage<N> st;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool mine = i < B && (mask == nullptr || mask[i] != 0);
  if constexpr (N == 4) { if (!mine) return; }
  else { if (__ballot(mine) == 0ull) return; }          // 5x5 boards move through the wave's LDS slice
  auto b = load_board(boards, i, B, st);
  if (mine) {
    Aux a = ld_aux(aux, i);
    begin_episode(b, a, seed, env_id0 + (uint64_t)i, (flags & Q2048_FLAG_RESET_SHAPING) != 0);
    st_aux(aux, i, a);
  }
  store_board(boards, i, B, b, st);
}

// ENV: env profile bits (kEnvDqn: the DQN path's step).  draw_pos != nullptr: injected draws,
// element i * draw_stride of draw_pos / draw_val (and, for kEnvDqn, draw_opos / draw_oval: the
// spawn inside is_game_over's move).
// boards_in / boards_out: the same buffer (in place) or two (the state before the step stays
// intact for update_q_value: the batched loop needs no board copy).  max_tile (may be NULL): the
// reference's `info`, the raw max tile (Game2048_env.py:100,129), next to its log2.
template <int N, int ENV>
__global__ __launch_bounds__(kBlock) void k_env_step(const uint8_t* boards_in, uint8_t* boards, q2048_aux* aux,
                                                     const uint8_t* actions, int64_t B, uint64_t seed,
                                                     uint64_t env_id0, uint32_t ctr, float* reward,
                                                     uint8_t* done, uint8_t* max_l2, int32_t* max_tile,
                                                     uint32_t* status,
                                                     const uint32_t* draw_pos, const uint32_t* draw_val,
                                                     const uint32_t* draw_opos, const uint32_t* draw_oval,
                                                     int draw_stride) {
  __shared__ Stage<N> st;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  auto b = load_board(boards_in, i, B, st);
  if (i < B) {
    const int act = actions[i];
    if (act > 3) {  // rejected, never masked (Game2048_env.py:56-60 would mis-rotate)
      atomicOr(status, Q2048_STATUS_BAD_ACTION);
      reward[i] = 0.f; done[i] = 0; max_l2[i] = 0;
      if (max_tile != nullptr) max_tile[i] = 0;
    } else {
      Aux a = ld_aux(aux, i);
      Draws x, y{0u, 0u, 0u, 0u};
      if (draw_pos != nullptr) {  // injected (parity tests)
        x.x2 = draw_pos[i * draw_stride]; x.x3 = draw_val[i * draw_stride];
        if constexpr ((ENV & kEnvDqn) != 0) { y.x0 = draw_opos[i * draw_stride]; y.x1 = draw_oval[i * draw_stride]; }
      } else {
        x = draws(seed, env_id0 + (uint64_t)i, ctr, kStreamStep);
        if constexpr ((ENV & kEnvDqn) != 0) y = draws(seed, env_id0 + (uint64_t)i, ctr, kStreamOver);
      }
      const StepOut o = env_step_profile<ENV>(b, a, act, x.x2, x.x3, y.x0, y.x1);
      st_aux(aux, i, a);
      reward[i] = o.reward; done[i] = o.done; max_l2[i] = o.max_log2;
      if (max_tile != nullptr) max_tile[i] = o.max_log2 ? (int32_t)(1u << o.max_log2) : 0;
    }
  }
  store_board(boards, i, B, b, st);
}

#ifdef Q2048_EXPERIMENTS
// The 4x4 single-step kernel of the 4-call API for more than one board per thread (experiment
// bits of q2048_env_step_ex): no LDS staging, and a thread requests board, aux and action of its
// next board before it computes the current one.  Measured slower than one board per thread
// (profiles/r02_env_step.jsonl: 14.6 / 15.2 / 16.1 / 19.1 us per 1 Mi boards for 1 / 2 / 4 / 8), as
// did starting every other block 1-8 Ki cycles late (r02_env_step_stagger.jsonl): k_env_step4
// below is the default.
template <int ENV>
__global__ __launch_bounds__(kBlock) void k_env_step4_pipelined(
    uint8_t* boards, q2048_aux* aux, const uint8_t* actions, int64_t B, uint64_t seed,
    uint64_t env_id0, uint32_t ctr, float* reward, uint8_t* done, uint8_t* max_l2, uint32_t* status) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= B) return;
  uint4 bv = reinterpret_cast<const uint4*>(boards)[i];
  uint4 av = reinterpret_cast<const uint4*>(aux)[i];
  uint32_t act = actions[i];
  for (;;) {
    const int64_t nxt = i + stride;
    uint4 bn = bv, an = av;
    uint32_t actn = 0;
    if (nxt < B) {                                        // in flight during the arithmetic below
      bn = reinterpret_cast<const uint4*>(boards)[nxt];
      an = reinterpret_cast<const uint4*>(aux)[nxt];
      actn = actions[nxt];
    }
    if (act > 3u) {  // rejected, never masked (Game2048_env.py:56-60 would mis-rotate)
      atomicOr(status, Q2048_STATUS_BAD_ACTION);
      reward[i] = 0.f; done[i] = 0; max_l2[i] = 0;
    } else {
      Board b{bv.x, bv.y, bv.z, bv.w};
      Aux a = words_to_aux(Words4{av.x, av.y, av.z, av.w});
      const Draws x = draws(seed, env_id0 + (uint64_t)i, ctr, kStreamStep);
      Draws y{0u, 0u, 0u, 0u};
      if constexpr ((ENV & kEnvDqn) != 0) y = draws(seed, env_id0 + (uint64_t)i, ctr, kStreamOver);
      const StepOut o = env_step_profile<ENV>(b, a, (int)act, x.x2, x.x3, y.x0, y.x1);
      reinterpret_cast<uint4*>(boards)[i] = make_uint4(b.r0, b.r1, b.r2, b.r3);
      st_aux(aux, i, a);
      reward[i] = o.reward; done[i] = o.done; max_l2[i] = o.max_log2;
    }
    if (nxt >= B) break;
    i = nxt; bv = bn; av = an; act = actn;
  }
}
#endif  // Q2048_EXPERIMENTS

// The 4x4 step of the 4-call API: one board per thread in straight-line code (37 VGPRs, 8 waves
// per SIMD; more boards per thread with register prefetch measured slower, profiles/r02_env_step.jsonl).  14.7 us per 1 Mi boards = 5.0 TB/s of the 70 B/step, 99.6 us per 8 Mi
// boards = 5.9 TB/s (94 % of the 6.29 TB/s copy ceiling); a kernel that only moves the same bytes
// takes 12.07 / 103.5 us (tools/archive/exp_stream_floor.hip).
template <int ENV>
__global__ __launch_bounds__(kBlock) void k_env_step4(
    const uint8_t* boards_in, uint8_t* boards, q2048_aux* aux, const uint8_t* actions, int64_t B,
    uint64_t seed, uint64_t env_id0, uint32_t ctr, float* reward, uint8_t* done, uint8_t* max_l2,
    int32_t* max_tile, uint32_t* status) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= B) return;
  const uint4 bv = reinterpret_cast<const uint4*>(boards_in)[i];
  const uint4 av = reinterpret_cast<const uint4*>(aux)[i];
  const uint32_t act = actions[i];
  // The three loads go out together: testing the action first would put its round trip in front
  // of the board's and the aux record's (the compiler sinks loads below a branch that does not
  // need them).  A bad action's lane computes a step it then discards.
  // The draws need nothing from memory: they are computed while the loads are in flight (the
  // scheduler would otherwise start with the slide, i.e. with a wait -- at the start of a launch
  // every wave of the chip sits in that wait at once).
  Draws x = draws(seed, env_id0 + (uint64_t)i, ctr, kStreamStep);
  Draws y{0u, 0u, 0u, 0u};
  if constexpr ((ENV & kEnvDqn) != 0) y = draws(seed, env_id0 + (uint64_t)i, ctr, kStreamOver);
  asm volatile("" : "+v"(x.x2), "+v"(x.x3), "+v"(y.x0), "+v"(y.x1));   // here, not sunk into the spawn's branch
  __builtin_amdgcn_sched_barrier(0);
  Board b{bv.x, bv.y, bv.z, bv.w};
  Aux a = words_to_aux(Words4{av.x, av.y, av.z, av.w});
  const StepOut o = env_step_profile<ENV>(b, a, (int)(act & 3u), x.x2, x.x3, y.x0, y.x1);
  if (act > 3u) {  // rejected, never masked (Game2048_env.py:56-60 would mis-rotate); board and aux stay
    atomicOr(status, Q2048_STATUS_BAD_ACTION);
    reward[i] = 0.f; done[i] = 0; max_l2[i] = 0;
    if (max_tile != nullptr) max_tile[i] = 0;
    if (boards != boards_in) reinterpret_cast<uint4*>(boards)[i] = bv;
    return;
  }
  reinterpret_cast<uint4*>(boards)[i] = make_uint4(b.r0, b.r1, b.r2, b.r3);
  st_aux(aux, i, a);
  reward[i] = o.reward; done[i] = o.done; max_l2[i] = o.max_log2;
  if (max_tile != nullptr) max_tile[i] = o.max_log2 ? (int32_t)(1u << o.max_log2) : 0;
}

// legal-move mask (mainDQL_CNN_step2.py:168-174): four trial moves per lane, nothing stored back
template <int N>
__global__ __launch_bounds__(kBlock) void k_legal_moves(const uint8_t* boards, int64_t B, uint8_t* mask_out) {
  __shared__ Stage<N> st;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const auto b = load_board(boards, i, B, st);
  if (i >= B) return;
  uint32_t m = 0;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    auto t = b;
    uint32_t score;
    m |= (uint32_t)move(t, a, score) << a;
  }
  mask_out[i] = (uint8_t)m;
}

// one-hot encoder (Dqn8TestNOPERCNN.py:271-277): thread = (board, channel, row) -> 4 outputs;
// the 1 KiB (f32) / 512 B (bf16) image of a board is written by 64 consecutive threads
template <bool BF16>
__global__ __launch_bounds__(kBlock) void k_encode_onehot(const uint8_t* boards, int64_t B, void* out) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= B * 64) return;
  const int64_t b = t >> 6;
  const uint32_t c = (uint32_t)(t >> 2) & 15u, r = (uint32_t)t & 3u;
  const uint32_t w = reinterpret_cast<const uint32_t*>(boards)[b * 4 + r];  // row r: 4 tile bytes
  const bool h0 = (w & 0xffu) == c, h1 = ((w >> 8) & 0xffu) == c, h2 = ((w >> 16) & 0xffu) == c,
             h3 = (w >> 24) == c;
  if constexpr (BF16) {  // bf16 1.0 = 0x3F80
    reinterpret_cast<uint2*>(out)[t] = make_uint2((h0 ? 0x3F80u : 0u) | (h1 ? 0x3F800000u : 0u),
                                                  (h2 ? 0x3F80u : 0u) | (h3 ? 0x3F800000u : 0u));
  } else {
    reinterpret_cast<float4*>(out)[t] = make_float4(h0 ? 1.f : 0.f, h1 ? 1.f : 0.f, h2 ? 1.f : 0.f,
                                                    h3 ? 1.f : 0.f);
  }
}

// ---------------------------------------------------------------------------------------------
// agent kernels
// ---------------------------------------------------------------------------------------------
// Row cache of the 4-call API (optional, caller-owned, one record per env): what the fused rollout
// carries in registers from one step to the next -- the row this env read as s' in its last
// update, with its slot -- handed from one q_update call to the next q_choose / q_update through
// HBM as a coalesced stream.  s of step t + 1 is s' of step t unless an episode began, so the
// update then needs ONE scattered row read (s') instead of two, and a greedy choose none.  A
// record is used only when its key equals the key of the board actually passed in (0 = empty), so
// any calling pattern is correct; like the register-carried row it does not see what OTHER envs
// wrote to that row since.
template <int N> struct RowCache;
template <> struct RowCache<4> { u64 key; float q[4]; u64 slot; };                      // 32 B
template <> struct RowCache<5> { u64 key; float q[4]; u64 key_hi; u64 slot; u64 pad; }; // 48 B
static_assert(sizeof(RowCache<4>) == 32 && sizeof(RowCache<5>) == 48, "ABI layout");

// A record also carries, in the 24 bits above its 40-bit slot index, a TAG of the table it was read from (a hash
// of the table's address and capacity): a record left by a launch on another table -- the table has grown, or the
// caller switched tables and did not zero the cache -- never matches, so its slot index is never used against the
// wrong table (records outlive launches since round 4 and were trusted on a key match alone).  A table
// rewritten IN PLACE (zero-filled, imported into) keeps its tag: the caller zero-fills the cache then, as before.
__host__ __device__ __forceinline__ u64 cache_tag(const q2048_slot* table, u64 mask) {
  return (mix64((u64)reinterpret_cast<uintptr_t>(table) ^ (mask * 0x9E3779B97F4A7C15ull)) >> 40) << 40;
}
constexpr u64 kCacheSlotMask = (1ull << 40) - 1ull;
// VISIT ROWS (Q2048_FLAG_NO_NEW_ROWS).  With the key set closed a state without a row reads as the zero row the
// defaultdict would have created -- and while the env STAYS in that state (invalid moves) that fresh row learns as the
// defaultdict's would (Agent/main.py:43): the lane keeps it in the registers that otherwise carry the table's row, so
// that the first invalid move's negative reward sends argmax on to the next action instead of repeating action 0 until
// the stall rule ends the episode.  It is never part of the table and ends when the env leaves the state.  Across a
// launch boundary it travels like any carried row, through the row cache, as a ROWLESS record: slot field all ones
// (no table of 2^40 slots exists).  Only calls that carry the flag write or accept such records.
constexpr u64 kCacheRowless = kCacheSlotMask;
__device__ __forceinline__ bool cache_slot(u64 s, bool rowless_ok, int64_t& slot) {
  const u64 v = s & kCacheSlotMask;
  if (v == kCacheRowless) { slot = INT64_MIN; return rowless_ok; }   // (INT64_MIN = kNoSlot, defined below)
  slot = (int64_t)v;
  return true;
}
__device__ __forceinline__ bool cache_get(const RowCache<4>* c, int64_t i, const Geo<4>::Key& key, u64 tag, Row& r,
                                          int64_t& slot, bool rowless_ok = false) {
  const uint4* p = reinterpret_cast<const uint4*>(c + i);
  const uint4 a = p[0], b = p[1];
  const u64 s = (u64)b.z | ((u64)b.w << 32);
  int64_t at;
  if (((u64)a.x | ((u64)a.y << 32)) != key.k0 || (s & ~kCacheSlotMask) != tag || !cache_slot(s, rowless_ok, at)) return false;
  r = Row{bits_f32(a.z), bits_f32(a.w), bits_f32(b.x), bits_f32(b.y)};
  slot = at;
  return true;
}
__device__ __forceinline__ bool cache_get(const RowCache<5>* c, int64_t i, const Geo<5>::Key& key, u64 tag, Row& r,
                                          int64_t& slot, bool rowless_ok = false) {
  const uint4* p = reinterpret_cast<const uint4*>(c + i);
  const uint4 a = p[0], b = p[1], d = p[2];
  const u64 s = (u64)d.x | ((u64)d.y << 32);
  int64_t at;
  if (((u64)a.x | ((u64)a.y << 32)) != key.k0 || ((u64)b.z | ((u64)b.w << 32)) != key.k1 || (s & ~kCacheSlotMask) != tag ||
      !cache_slot(s, rowless_ok, at))
    return false;
  r = Row{bits_f32(a.z), bits_f32(a.w), bits_f32(b.x), bits_f32(b.y)};
  slot = at;
  return true;
}
__device__ __forceinline__ void cache_put(RowCache<4>* c, int64_t i, const Geo<4>::Key& key, u64 tag, const Row& r,
                                          int64_t slot, bool rowless = false) {
  const u64 k = (slot >= 0 || rowless) ? key.k0 : 0ull;   // no row and no visit row: nothing to remember
  const u64 s = (slot >= 0 ? ((u64)slot & kCacheSlotMask) : kCacheRowless) | tag;
  uint4* p = reinterpret_cast<uint4*>(c + i);
  p[0] = make_uint4((uint32_t)k, (uint32_t)(k >> 32), f32_bits(r.q0), f32_bits(r.q1));
  p[1] = make_uint4(f32_bits(r.q2), f32_bits(r.q3), (uint32_t)s, (uint32_t)(s >> 32));
}
__device__ __forceinline__ void cache_put(RowCache<5>* c, int64_t i, const Geo<5>::Key& key, u64 tag, const Row& r,
                                          int64_t slot, bool rowless = false) {
  const u64 k = (slot >= 0 || rowless) ? key.k0 : 0ull;
  const u64 s = (slot >= 0 ? ((u64)slot & kCacheSlotMask) : kCacheRowless) | tag;
  uint4* p = reinterpret_cast<uint4*>(c + i);
  p[0] = make_uint4((uint32_t)k, (uint32_t)(k >> 32), f32_bits(r.q0), f32_bits(r.q1));
  p[1] = make_uint4(f32_bits(r.q2), f32_bits(r.q3), (uint32_t)key.k1, (uint32_t)(key.k1 >> 32));
  p[2] = make_uint4((uint32_t)s, (uint32_t)(s >> 32), 0u, 0u);
}

// q2048_rowcache_rebind: the visit rows of a cache follow their table's ROWS into another allocation (a checkpoint
// restored): a record without a slot that carries the old table's tag gets the new table's; every other record (the
// slot indices of the old allocation mean nothing in the new one) is emptied.
template <int N>
__global__ __launch_bounds__(kBlock) void k_rowcache_rebind(RowCache<N>* cache, int64_t B, u64 tag_from, u64 tag_to) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= B) return;
  RowCache<N> r = cache[i];
  if (r.key != 0ull && r.slot == (kCacheRowless | tag_from)) {
    r.slot = kCacheRowless | tag_to;
  } else {
    r = RowCache<N>{};
  }
  cache[i] = r;
}

template <int N>
__global__ __launch_bounds__(kBlock) void k_q_choose(const q2048_slot* table, u64 mask,
                                                     const uint8_t* boards, int64_t B, double eps,
                                                     uint64_t seed, uint64_t env_id0, uint32_t ctr,
                                                     uint32_t flags, const RowCache<N>* cache,
                                                     uint8_t* actions, uint32_t* status,
                                                     const uint32_t* draw_eps, const uint32_t* draw_act) {
  __shared__ Stage<N> st;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const auto b = load_board(boards, i, B, st);
  if (i >= B) return;
  const uint64_t id = env_id0 + (uint64_t)i;
  const u64 salt = (flags & Q2048_FLAG_INDEPENDENT) ? lane_salt(id) : 0ull;
  Draws x;
  if (draw_eps != nullptr) { x.x0 = draw_eps[i]; x.x1 = draw_act[i]; }  // injected (parity tests)
  else x = draws(seed, id, ctr, kStreamStep);
  int act;
  if (draw_uniform(x.x0) < eps) {  // Agent/main.py:35-36: no table access when exploring
    act = draw_action(x.x1);
  } else {
    Row r;
    bool made;
    int64_t slot;
    const auto key = state_key(b, salt, status);
    // (closed key set: a rowless record is this env's visit row -- the fresh row of a state without one, as far as
    // its own invalid moves have taught it)
    if (cache == nullptr || !cache_get(cache, i, key, cache_tag(table, mask), r, slot, (flags & Q2048_FLAG_NO_NEW_ROWS) != 0u))
      probe_find(table, mask, key, r, made);
    act = argmax4(r.q0, r.q1, r.q2, r.q3);
  }
  actions[i] = (uint8_t)act;
}

template <int N>
__global__ __launch_bounds__(kBlock) void k_q_lookup(const q2048_slot* table, u64 mask,
                                                     const uint8_t* boards, int64_t B, uint64_t env_id0,
                                                     uint32_t flags, float* q_out, uint8_t* found,
                                                     uint32_t* status) {
  __shared__ Stage<N> st;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const auto b = load_board(boards, i, B, st);
  if (i >= B) return;
  const uint64_t id = (flags & Q2048_FLAG_SINGLE_ENV) ? env_id0 : env_id0 + (uint64_t)i;
  const u64 salt = (flags & Q2048_FLAG_INDEPENDENT) ? lane_salt(id) : 0ull;
  Row r;
  bool made;
  const int64_t slot = probe_find(table, mask, state_key(b, salt, status), r, made, kMaxProbe);
  reinterpret_cast<float4*>(q_out)[i] = make_float4(r.q0, r.q1, r.q2, r.q3);
  if (found != nullptr) found[i] = slot >= 0;
}

// Workgroups of kUpdateBlock = 1024 lanes: every block ends with one global atomic per non-zero
// statistic, all on the same few addresses, and in a one-step launch those same-address atomics are
// a serial tail -- 4096 blocks of 256 cost the launch 15 of its 85 us at 1 Mi boards (statistics
// pointer NULL: 101 against 116 us per 4-call step, tools/archive/exp_unfused.py UNFUSED_NO_STATS); a
// quarter as many blocks, a quarter of the tail.  (The fused kernel amortises its flush over K steps:
// no measurable cost there.)
constexpr int kUpdateBlock = 1024;
// FROZEN (Q2048_FLAG_NO_NEW_ROWS) is a template parameter: as a run-time flag around `claim_issue` it cost the learning
// instantiation 7 % (45.8 against 42.6 us per 1 Mi-board call, round-5 tree against this one on one box,
// profiles/r06_four_call_ab.txt) -- the claim's compare-and-swap lost its overlap with the TD write
template <int N, bool FROZEN>
__global__ __launch_bounds__(kUpdateBlock) void k_q_update(q2048_slot* table, u64 mask, const uint8_t* s,
                                                     const uint8_t* actions, const float* reward,
                                                     const uint8_t* s2, const uint8_t* done, int64_t B,
                                                     double lr, double gamma, uint64_t env_id0,
                                                     uint32_t flags, RowCache<N>* cache, int64_t* stats_i,
                                                     uint32_t* status) {
  __shared__ BlockStats bs;
  __shared__ Stage<N, kUpdateBlock / 64> st;
  stats_clear(bs);
  const int64_t i = (int64_t)blockIdx.x * kUpdateBlock + threadIdx.x;
  const auto b_s = load_board(s, i, B, st);
  const auto b_n = load_board(s2, i, B, st);
  if (i < B) {
    const int act = actions[i];
    // every coalesced input is requested here, together: the probes below are `asm volatile` with a
    // memory clobber, and a load left after them would wait for its own round trip behind theirs
    const float rew = reward[i];
    const bool is_done = done[i] != 0;
    bool ins_n = false, ins_s = false, dropped = false;
    TdCounters tdc{0u, 0u};
    // Q2048_FLAG_NO_NEW_ROWS: the key set is closed -- absent states read as zeros, nothing is claimed, an update
    // of a state without a row is dropped and counted (the caller's policy: no TABLE_FULL)
    constexpr bool frozen = FROZEN;
    if (act > 3) {
      atomicOr(status, Q2048_STATUS_BAD_ACTION);
    } else {
      const u64 salt = (flags & Q2048_FLAG_INDEPENDENT) ? lane_salt(env_id0 + (uint64_t)i) : 0ull;
      const auto key_s = state_key(b_s, salt, status);
      const auto key_n = state_key(b_n, salt, status);
      const bool same = key_eq(key_n, key_s);
      // q_table[state] (Agent/main.py:43): the row this env carried over from its last update, or a
      // probe; the defaultdict creates the row when absent, so do we
      Row rs;
      int64_t slot = kNoSlot;
      if (cache == nullptr || !cache_get(cache, i, key_s, cache_tag(table, mask), rs, slot, frozen)) {
        slot = probe_find(table, mask, key_s, rs, ins_s);
        if (slot < 0 && slot != kNoSlot && !frozen) slot = probe_insert(table, mask, key_s, (u64)~slot, ins_s);
      }
      // q_table[next_state] (:41), created when absent as well; an invalid move stays on the row of s
      // The claim of an absent s' is issued here and its answer read after the TD write of s, which
      // does not depend on it: one round trip less on the lane's chain (4x4; 5x5 claims in place).
      Row rn = rs;
      int64_t slot_n = slot;
      Claim claim{0ull, 0ull, false};
      if (!same) {
        slot_n = probe_find(table, mask, key_n, rn, ins_n);
        if constexpr (!frozen) slot_n = claim_issue(table, mask, slot_n, key_n, claim, ins_n);
      }
      const float max_next = max4(rn.q0, rn.q1, rn.q2, rn.q3);
      if (slot >= 0) {
        const float nq = td_update(&table[slot], act, row_get(rs, act), rew, max_next, is_done,
                                   lr, gamma, tdc, td_mode_of(flags));
        if (same) row_set(rn, act, nq);                  // the row it stays on just changed (:100)
      } else {
        dropped = true;
        if (!frozen) atomicOr(status, Q2048_STATUS_TABLE_FULL);
        else if (same)   // closed key set: the update lands in the env's visit row (the record below), not in the table
          row_set(rn, act, td_value(row_get(rs, act), rew, max_next, is_done, lr, gamma));
      }
      bool ins_c = false;
      slot_n = claim_resolve(table, mask, key_n, claim, slot_n, ins_c);
      ins_n = ins_n || ins_c;
      if (cache != nullptr) cache_put(cache, i, key_n, cache_tag(table, mask), rn, slot_n, frozen);
    }
    const uint32_t n_ins = wave_count(ins_n) + wave_count(ins_s), n_drop = wave_count(dropped);
    if (tdc.retries) atomicAdd(&bs.i[Q2048_ST_CAS_RETRY], (u64)tdc.retries);
    if (tdc.fallbacks) atomicAdd(&bs.i[Q2048_ST_CAS_FALLBACK], (u64)tdc.fallbacks);
    if (wave_leader()) {
      if (n_ins) atomicAdd(&bs.i[Q2048_ST_INSERTS], (u64)n_ins);
      if (n_drop) atomicAdd(&bs.i[Q2048_ST_DROPS], (u64)n_drop);
    }
  }
  stats_flush(bs, stats_i, nullptr);
}

// ---------------------------------------------------------------------------------------------
// fused rollout: Agent/main.py:91-101 + reset (:81), `steps` times per lane in one launch.
//
// Per step and lane the table sees: one probe of the next state (a read), at most one row claim
// (compare-and-swap on the key word, only for a state reached for the first time) and one
// 4-byte write of Q[s][a].  On 4x4 the claim of s' is issued as soon as the probe finds it absent
// and consumed one step later, when s' has become s.  Rows appear exactly when the
// reference's defaultdict creates them (q_table[next_state] / q_table[state] in
// update_q_value, Agent/main.py:41-43).
// ---------------------------------------------------------------------------------------------
// Waves per SIMD the register allocator must leave room for.  4x4: 6 (<= 80 VGPRs, 9 dwords of
// scratch, two 8-byte reloads per step) measures 42.4 us per 1 Mi-board step against 43.0 / 43.3 at
// 4 / 5 waves; 7 and 8 spill more than the extra waves hide (49.1, 55.3).  5x5: 5 (96 VGPRs, no
// scratch in the step loop) measures 1 % over 6 (profiles/r03_launch_cost.txt).
#ifdef Q2048_FUSED_MIN_WAVES                       // measurement builds: one value for both geometries
#define Q2048_FUSED_WAVES(N) Q2048_FUSED_MIN_WAVES
#else
#define Q2048_FUSED_WAVES(N) ((N) == 4 ? 6 : 5)
#endif
// MODE: what the launch does with the table -- a template parameter, so the step loop carries no
// run-time mode tests (kModeLearn: plain-store TD; kModeCas: Q2048_FLAG_TD_CAS; kModeEval:
// Q2048_FLAG_NO_LEARN; play-only is an ENV bit).  Experiment builds select write modes at run time.
// kModeFrozen (a bit, with kModeLearn or kModeCas): Q2048_FLAG_NO_NEW_ROWS -- the key set is closed: no claim is ever
// issued (the Claim pipeline, the insert of an episode's opening state and of a terminal state compile away), a
// state without a row reads as zeros and its update is dropped and counted.
constexpr int kModeLearn = 0, kModeCas = 1, kModeEval = 2, kModeFrozen = 4, kModeSummary = 8;   // (kModeSummary: with kModeFrozen, 4x4)
// Lanes per workgroup of the fused rollout: 512 for batches that fill the chip more than twice over at that
// size (>= 786 432 boards), 256 below.  Nine alternating pairs of the driver's command at 1 Mi boards: 47.2 us
// per step against 48.3 (8 of 9 pairs; 5x5: 64.4 against 66.4; profiles/r04_block512_*.txt) -- half as many
// blocks end a launch with their statistics atomics on one cache line; at 65 536 boards 512-lane workgroups
// would leave half of the CUs empty (8.0 against 5.6 us per step).  64- and 128-lane workgroups cost more than
// they give at every size (profiles/r04_block_size_intercept.txt).
constexpr int kFusedBlockBig = 512, kFusedBlockSmall = kBlock;
constexpr int64_t kFusedBigBatch = 786432;
#ifdef Q2048_EXPERIMENTS
// Measurement builds only (tools/archive/exp_timeline.py): when set (q2048_debug_timeline), every block of a fused
// launch leaves four 100 MHz wall-clock stamps -- in, first step done, last step done, out -- and where it
// ran (HW_ID: wave / SIMD / CU / SH / SE; XCC_ID) in g_timeline[blockIdx.x * 8 ..]: where in a launch the
// chip is not full (ramp, rounds, drain), and whether some part of it is slower than the rest.
__device__ unsigned long long* g_timeline;
#define Q2048_STAMP(k) do { if (g_timeline != nullptr && threadIdx.x == 0) {                       \
    g_timeline[(size_t)blockIdx.x * 8 + (k)] = wall_clock64();                                     \
    if ((k) == 0) {                                                                                \
      g_timeline[(size_t)blockIdx.x * 8 + 4] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);  /* HW_REG_HW_ID */ \
      g_timeline[(size_t)blockIdx.x * 8 + 5] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20); /* HW_REG_XCC_ID */ \
    } } } while (0)
#else
#define Q2048_STAMP(k) do { } while (0)
#endif
template <int N, int ENV, int MODE, int BLOCK>
__global__ __launch_bounds__(BLOCK, Q2048_FUSED_WAVES(N)) void k_fused_rollout(
    uint8_t* boards, q2048_aux* aux, q2048_slot* table, u64 mask, int64_t B, int steps, double eps,
    double lr, double gamma, uint64_t seed, uint64_t env_id0, uint32_t ctr0, uint32_t flags,
    int64_t* stats_i, double* stats_f, uint32_t* status, q2048_episode* log, int64_t log_cap,
    u64* log_count, void* row_cache, u64* mirror, uint32_t* ticket) {
  RowCache<N>* const cache = static_cast<RowCache<N>*>(row_cache);
  __shared__ BlockStats bs;
  __shared__ Stage<N, BLOCK / 64> st;
  Q2048_STAMP(0);
  stats_clear(bs);
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  auto b = load_board(boards, i, B, st);
  if (i < B) {
    const uint64_t id = env_id0 + (uint64_t)i;
    const u64 salt = (flags & Q2048_FLAG_INDEPENDENT) ? lane_salt(id) : 0ull;
    // Q2048_FLAG_PLAY_ONLY (ENV bit kEnvPlayOnly: its own instantiation, so profiles tell the
    // learner-less launches from the learning ones): the table is never touched -- every row reads
    // as zeros, nothing is created or written.
    // Q2048_FLAG_NO_LEARN (MODE kModeEval): evaluation of a trained table -- rows are read
    // (epsilon-greedy over the stored values), nothing is created or written
    constexpr bool play_only = (ENV & kEnvPlayOnly) != 0;
    constexpr bool no_learn = MODE == kModeEval;
    constexpr bool frozen = (MODE & kModeFrozen) != 0;
    constexpr bool summary = (MODE & kModeSummary) != 0 && N == 4;   // line summaries: Q2048_FLAG_LINE_SUMMARY
#ifdef Q2048_EXPERIMENTS   // bits 8..11 write mode, 12 no row creation, 13 no next-state probe, 14 no deferral, 16..23 CAS attempts
    const uint32_t td_mode = no_learn ? (uint32_t)kTdNone : td_mode_of(flags);
    const bool x_noclaim = ((flags >> 12) & 1u) || play_only || no_learn || frozen, x_noprobe = ((flags >> 13) & 1u) || play_only;
    const bool may_defer = td_mode == kTdStorePlain && !((flags >> 14) & 1u);
    const int max_cas = ((flags >> 16) & 0xffu) ? (int)((flags >> 16) & 0xffu) : kMaxCas;
#else
    constexpr uint32_t td_mode = no_learn ? kTdNone : ((MODE & kModeCas) ? kTdCas : kTdStorePlain);
    constexpr bool x_noclaim = play_only || no_learn || frozen, x_noprobe = play_only;
    constexpr bool may_defer = td_mode == kTdStorePlain;
    constexpr int max_cas = kMaxCas;
#endif
    Aux a = ld_aux(aux, i);
    auto key_s = state_key(b, salt, status);
    Row q{0.f, 0.f, 0.f, 0.f};
    bool made0 = false;
    // The row the launch starts in: what this env's last launch (or update_q_value) left in the row
    // cache when the board is still the one it left -- a coalesced 32-byte read -- else a probe (one
    // scattered 128-byte request per lane: 21.5 us of every launch at 1 Mi boards)
    int64_t slot_s = kNoSlot;
    if (!play_only && (cache == nullptr || !cache_get(cache, i, key_s, cache_tag(table, mask), q, slot_s, frozen)))
      slot_s = probe_find_as<summary>(table, mask, key_s, q, made0);
    Claim claim{0ull, 0ull, false};
    // wave-uniform counters (ballots) and rare per-lane ones
    uint32_t n_valid = 0, n_explore = 0, n_done = 0, n_insert = wave_count(made0), n_drop = 0;
    uint32_t pend = 0;
    TdCounters tdc{0u, 0u};
    double reward_sum = 0.0;

    for (int t = 0; t < steps; ++t) {
      const Draws x = draws(seed, id, ctr0 + (uint32_t)t, kStreamStep);
      Draws y{0u, 0u, 0u, 0u};
      if constexpr ((ENV & kEnvDqn) != 0) y = draws(seed, id, ctr0 + (uint32_t)t, kStreamOver);
      bool explored;
      const int act = eps_greedy(eps, x.x0, x.x1, q.q0, q.q1, q.q2, q.q3, explored);  // main.py:92
      const StepOut o = env_step_profile<ENV>(b, a, act, x.x2, x.x3, y.x0, y.x1);      // :93
      const auto key_n = state_key(b, salt, status);                                   // :94
      const bool same = key_eq(key_n, key_s);
      // the row of s: claimed one step ago (in flight since), or now if s opened the episode/launch
      bool ins_s = false, ins_n = false;
      slot_s = claim_resolve(table, mask, key_s, claim, slot_s, ins_s);
      if (slot_s < 0 && slot_s != kNoSlot && !x_noclaim)   // s opened the episode / launch
        slot_s = probe_insert(table, mask, key_s, (u64)~slot_s, ins_s);
      // q_table[next_state] (:41)
      Row qn = q;
      int64_t slot_n = slot_s;
      if (!same) {
        if (x_noprobe) { qn = Row{0.f, 0.f, 0.f, 0.f}; slot_n = ~(int64_t)key_home(key_n, mask); }
        else slot_n = probe_find_as<summary>(table, mask, key_n, qn, ins_n);
      }
      const float max_next = max4(qn.q0, qn.q1, qn.q2, qn.q3);
      float nq = 0.f;
      const bool updated = slot_s >= 0;
      if (updated) {                                                                   // :43, :99
        if (same && !o.done && may_defer) {   // the lane stays on this row: keep the write back
          nq = td_value(row_get(q, act), o.reward, max_next, false, lr, gamma);
          pend |= 1u << act;
        } else {
          nq = td_update(&table[slot_s], act, row_get(q, act), o.reward, max_next, o.done != 0, lr,
                         gamma, tdc, td_mode, max_cas);
          pend &= ~(1u << act);
          if (pend) flush_pending(&table[slot_s], q, pend);
        }
      } else if (frozen) {   // closed key set, no row: the update lands in the visit row `q` (registers) if the env stays
        nq = td_value(row_get(q, act), o.reward, max_next, o.done != 0, lr, gamma);
      }
      if (o.done) {                                                                    // :103
        // the terminal state's row exists in the reference too (looked up at :41)
        if (!same && slot_n < 0 && slot_n != kNoSlot && !x_noclaim)
          probe_insert(table, mask, key_n, (u64)~slot_n, ins_n);
        episode_stats(bs, a, o.max_log2);
        if (log != nullptr) {                                                          // :59-62, :105
          const u64 at = atomicAdd(log_count, 1ull);
          if ((int64_t)at < log_cap) {
            Row ql = q;
            if ((updated || frozen) && !no_learn) row_set(ql, act, nq);   // the logged row is the live one (:96), post-update
            q2048_episode rec;
            rec.env_id = id; rec.episode = a.episode; rec.action = (uint8_t)act;
            rec.max_log2 = o.max_log2; rec.steps_lo = (uint16_t)(ctr0 + (uint32_t)t);
            rec.reward = o.reward; rec.total_return = a.ep_return; rec.score = a.score;
            rec.q[0] = ql.q0; rec.q[1] = ql.q1; rec.q[2] = ql.q2; rec.q[3] = ql.q3;
            rec.reserved = 0u;
            log[at] = rec;
          }
        }
        begin_episode(b, a, seed, id, (ENV & kEnvResetShaping) != 0);                  // :81
        key_s = state_key(b, salt, status);
        bool made = false;
        q = Row{0.f, 0.f, 0.f, 0.f};
        slot_s = play_only ? kNoSlot : probe_find_as<summary>(table, mask, key_s, q, made);
        ins_n = ins_n || made;
      } else if (same) {            // invalid move: same state, its row just changed (:100)
        if ((updated || frozen) && !no_learn) row_set(q, act, nq);   // (evaluation: the stored row stays as it is)
        else if (!updated) slot_s = kNoSlot;             // dropped: do not retry the claim with a stale hint
      } else {
        key_s = key_n; slot_s = slot_n; q = qn;                                        // :100
        if (!x_noclaim) {
          bool made = false;
          slot_s = claim_issue(table, mask, slot_s, key_s, claim, made);
          ins_n = ins_n || made;
        }
      }
      n_valid += wave_count(o.valid != 0);
      n_explore += wave_count(explored);
      n_insert += wave_count(ins_s) + wave_count(ins_n);
      n_drop += wave_count(!updated && !play_only && !no_learn);
      n_done += wave_count(o.done != 0);
      reward_sum += (double)o.reward;
      if (t == 0) Q2048_STAMP(1);
    }
    Q2048_STAMP(2);
    if (pend && slot_s >= 0) flush_pending(&table[slot_s], q, pend);
    bool ins_last = false;  // the claim issued by the last step (its row belongs to the dict too)
    slot_s = claim_resolve(table, mask, key_s, claim, slot_s, ins_last);
    n_insert += wave_count(ins_last);
    st_aux(aux, i, a);
    // hand the carried row to this env's next launch / choose_action / update_q_value (a state whose
    // row does not exist yet -- an episode began on the last step -- leaves an empty record)
    if (!play_only && cache != nullptr) cache_put(cache, i, key_s, cache_tag(table, mask), q, slot_s, frozen);

    if (n_drop && !frozen) atomicOr(status, Q2048_STATUS_TABLE_FULL);   // (frozen: dropping is the caller's policy)
    if (tdc.retries) atomicAdd(&bs.i[Q2048_ST_CAS_RETRY], (u64)tdc.retries);
    if (tdc.fallbacks) atomicAdd(&bs.i[Q2048_ST_CAS_FALLBACK], (u64)tdc.fallbacks);
    atomicAdd(&bs.f[Q2048_SF_REWARD], reward_sum);
    const uint32_t n_active = wave_count(true);
    if (wave_leader()) {
      atomicAdd(&bs.i[Q2048_ST_STEPS], (u64)n_active * (u64)steps);
      atomicAdd(&bs.i[Q2048_ST_VALID], (u64)n_valid);
      atomicAdd(&bs.i[Q2048_ST_EXPLORE], (u64)n_explore);
      atomicAdd(&bs.i[Q2048_ST_EPISODES], (u64)n_done);
      atomicAdd(&bs.i[Q2048_ST_INSERTS], (u64)n_insert);
      atomicAdd(&bs.i[Q2048_ST_DROPS], (u64)n_drop);
    }
  }
  store_board(boards, i, B, b, st);
  if (mirror == nullptr) {                                // (uniform over the grid)
    stats_flush(bs, stats_i, stats_f);
  } else {
    stats_flush_acked(bs, stats_i, stats_f);
    stats_mirror(stats_i, stats_f, mirror, ticket);
  }
  Q2048_STAMP(3);
}

// ---------------------------------------------------------------------------------------------
// deterministic mode.  One step = phase 1 (every env acts on the table as it is at the start of
// the step and emits where its update goes and its TD target), a stable radix sort of the updates
// by a 16-bit hash of (state, action) -- two 8-bit passes of the radix partition below -- and
// phase 2 (each (slot, action) group applies its updates in env order, one formula, one rounding).
// The sort key is a function of the state and the action alone and the fold is the same sequence of
// double operations on every path, so nothing in the result depends on how lanes are scheduled --
// not even through which slot a racing insert happened to win.
//
// An update travels as one 64-bit word + its double target:
//   bits 0..1 action | bits 2..43 row slot | bits 44..59 hash16(state, action) | 62 dropped | 63 applied
// ---------------------------------------------------------------------------------------------
constexpr int kDetHashShift = 44, kDetHashBits = 16;
constexpr u64 kDetSlotMask = (1ull << kDetHashShift) - 1ull;
constexpr u64 kDetDrop = 1ull << 62, kDetDone = 1ull << 63;
constexpr uint32_t kNoCarry = 0xffffffffu;
// the four values of a slot whose index is known: one 16-byte agent-scope request
__device__ __forceinline__ Row ld_row(const q2048_slot* s) {
  const u32x4 v = ld16_agent(&s->q[0]);
  return Row{bits_f32(v.x), bits_f32(v.y), bits_f32(v.z), bits_f32(v.w)};
}
// 4x4: 8 waves per SIMD (64 VGPRs, 3 dwords of scratch) measures 165.0 us per 1 Mi-board step against
// 167.9 at the unconstrained 65 VGPRs / 7 waves; 5x5 would spill 8-17 dwords for it and keeps 7
// VISITS (q2048_det_rollout_cached with the key set closed): the envs' VISIT ROWS -- the two-phase step keeps no row in
// registers from one step to the next, so an env's visit row lives in its row-cache record (a record without a slot,
// the fused rollout's and the 4-call API's format: the three paths hand visit rows to one another).  Its own
// instantiation: in the learning step the extra live values cost the 5x5 body 13 dwords of scratch.
template <int N, int ENV, bool VISITS>
__device__ __forceinline__ void det_phase1_body(
    uint8_t* boards, q2048_aux* aux, q2048_slot* table, u64 mask, int64_t B, double eps, double gamma,
    uint64_t seed, uint64_t env_id0, uint32_t ctr, uint32_t flags, u64* group_out, double* target_out,
    uint32_t* carry, int use_carry, StatStripe* stat_stripes, uint32_t* gs0, int n_gs, uint32_t* status,
    RowCache<N>* cache, double lr) {
  __shared__ BlockStats bs;
  __shared__ Stage<N> st;
  stats_clear(bs);
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  for (int64_t idx = i; idx < (int64_t)n_gs; idx += (int64_t)gridDim.x * kBlock) gs0[idx] = 0u;   // the first pass's group sums
  auto b = load_board(boards, i, B, st);
  if (i < B) {
    const uint64_t id = env_id0 + (uint64_t)i;
    const u64 salt = (flags & Q2048_FLAG_INDEPENDENT) ? lane_salt(id) : 0ull;
    Aux a = ld_aux(aux, i);
    const auto key_s = state_key(b, salt, status);
    const u64 hash_s = key_hash(key_s);
    const Draws x = draws(seed, id, ctr, kStreamStep);
    Draws y{0u, 0u, 0u, 0u};
    if constexpr ((ENV & kEnvDqn) != 0) y = draws(seed, id, ctr, kStreamOver);
    Row q{0.f, 0.f, 0.f, 0.f}, qn;
    bool ins_s = false, ins_n = false, dropped = false;
    const bool frozen = (flags & Q2048_FLAG_NO_NEW_ROWS) != 0u;   // closed key set: nothing is created, absent = dropped
    // s is the s' of the step before unless an episode began in between: its slot was found then
    // (rows never move), so there is no probe -- and the row itself is needed by greedy lanes only
    // (at epsilon 0.95: one scattered request per step less for 19 lanes of 20)
    const uint32_t held = use_carry ? carry[i] : kNoCarry;
    constexpr bool visits = VISITS;
    int64_t slot_s;
    if (held != kNoCarry) {
      slot_s = (int64_t)held;
      if (!(draw_uniform(x.x0) < eps)) q = ld_row(&table[slot_s]);
    } else {
      slot_s = probe_find(table, mask, key_s, q, ins_s);
      if (slot_s < 0 && slot_s != kNoSlot && !frozen) slot_s = probe_insert(table, mask, key_s, (u64)~slot_s, ins_s);
      if (slot_s < 0) { dropped = true; if (!frozen || slot_s == kNoSlot) atomicOr(status, Q2048_STATUS_TABLE_FULL); }
      if constexpr (visits) {
        if (slot_s < 0) {                                // no row: the env's visit row, as far as it has one
          Row v;
          int64_t at;
          if (cache_get(cache, i, key_s, cache_tag(table, mask), v, at, true) && at == kNoSlot) q = v;
        }
      }
    }
    bool explored;
    const int act = eps_greedy(eps, x.x0, x.x1, q.q0, q.q1, q.q2, q.q3, explored);     // main.py:92
    const StepOut o = env_step_profile<ENV>(b, a, act, x.x2, x.x3, y.x0, y.x1);         // :93
    const auto key_n = state_key(b, salt, status);
    int64_t slot_n = probe_find(table, mask, key_n, qn, ins_n);                         // :41
    if (slot_n < 0 && slot_n != kNoSlot && !frozen) slot_n = probe_insert(table, mask, key_n, (u64)~slot_n, ins_n);
    if constexpr (visits) {
      // closed key set: an env that stays in a state without a row (an invalid move) bootstraps from, and updates,
      // its own visit row -- private to the env, so it is applied here and not by the grouped phase 2; every other
      // env leaves an empty record (also what clears the records a fused launch left: `use_carry` == 0 is the
      // call's first step)
      const bool same = key_eq(key_n, key_s);
      if (dropped && same) qn = q;
      if (dropped && same && !o.done) {
        Row v = q;
        row_set(v, act, td_value(row_get(q, act), o.reward, max4(qn.q0, qn.q1, qn.q2, qn.q3), false, lr, gamma));
        cache_put(cache, i, key_s, cache_tag(table, mask), v, kNoSlot, true);
      } else if (dropped || !use_carry) {
        cache_put(cache, i, key_s, cache_tag(table, mask), q, kNoSlot, false);
      }
    }
    carry[i] = (!o.done && slot_n >= 0 && (u64)slot_n < (u64)kNoCarry) ? (uint32_t)slot_n : kNoCarry;
    // the group of this update: (slot of s, action), sorted by a hash of (s, action).  The sort is
    // stable and this array is in env order, so env order survives without an index
    const u64 h16 = ((hash_s >> 48) ^ ((u64)act * 0x5555ull)) & 0xffffull;
    group_out[i] = dropped ? kDetDrop : ((h16 << kDetHashShift) | ((u64)slot_s << 2) | (u64)act);
    target_out[i] = td_target(o.reward, max4(qn.q0, qn.q1, qn.q2, qn.q3), o.done != 0, gamma);   // :42
    if (o.done) {
      episode_stats(bs, a, o.max_log2);
      begin_episode(b, a, seed, id, (ENV & kEnvResetShaping) != 0);
    }
    st_aux(aux, i, a);
    const uint32_t n_valid = wave_count(o.valid != 0), n_explore = wave_count(explored),
                   n_done = wave_count(o.done != 0), n_ins = wave_count(ins_s) + wave_count(ins_n),
                   n_drop = wave_count(dropped), n_active = wave_count(true);
    double wave_reward = (double)o.reward;             // one LDS atomic per wave, not 64 on one address
    const bool full_wave = __ballot(true) == ~0ull;    // (the batch's ragged last wave adds lane by lane)
    if (full_wave) {
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) wave_reward += __shfl_xor(wave_reward, d);
    } else {
      atomicAdd(&bs.f[Q2048_SF_REWARD], wave_reward);
    }
    if (wave_leader()) {
      if (full_wave) atomicAdd(&bs.f[Q2048_SF_REWARD], wave_reward);
      atomicAdd(&bs.i[Q2048_ST_STEPS], (u64)n_active);
      atomicAdd(&bs.i[Q2048_ST_VALID], (u64)n_valid);
      atomicAdd(&bs.i[Q2048_ST_EXPLORE], (u64)n_explore);
      atomicAdd(&bs.i[Q2048_ST_EPISODES], (u64)n_done);
      atomicAdd(&bs.i[Q2048_ST_INSERTS], (u64)n_ins);
      atomicAdd(&bs.i[Q2048_ST_DROPS], (u64)n_drop);
    }
  }
  store_board(boards, i, B, b, st);
  stats_flush_striped(bs, stat_stripes);
}

template <int N, int ENV>
__global__ __launch_bounds__(kBlock, N == 4 ? 8 : 7) void k_det_phase1(
    uint8_t* boards, q2048_aux* aux, q2048_slot* table, u64 mask, int64_t B, double eps, double gamma,
    uint64_t seed, uint64_t env_id0, uint32_t ctr, uint32_t flags, u64* group_out, double* target_out,
    uint32_t* carry, int use_carry, StatStripe* stat_stripes, uint32_t* gs0, int n_gs, uint32_t* status) {
  det_phase1_body<N, ENV, false>(boards, aux, table, mask, B, eps, gamma, seed, env_id0, ctr, flags, group_out, target_out,
                                 carry, use_carry, stat_stripes, gs0, n_gs, status, nullptr, 0.0);
}
template <int N, int ENV>
__global__ __launch_bounds__(kBlock, 7) void k_det_phase1_visits(
    uint8_t* boards, q2048_aux* aux, q2048_slot* table, u64 mask, int64_t B, double eps, double gamma,
    uint64_t seed, uint64_t env_id0, uint32_t ctr, uint32_t flags, u64* group_out, double* target_out,
    uint32_t* carry, int use_carry, StatStripe* stat_stripes, uint32_t* gs0, int n_gs, uint32_t* status,
    void* row_cache, double lr) {
  det_phase1_body<N, ENV, true>(boards, aux, table, mask, B, eps, gamma, seed, env_id0, ctr, flags, group_out, target_out,
                                carry, use_carry, stat_stripes, gs0, n_gs, status, static_cast<RowCache<N>*>(row_cache), lr);
}

// The sort: least-significant-digit radix passes of 8 bits over (group, target) pairs, each pass a
// stable partition in three launches.  Tiles of kSortTile pairs, one workgroup each:
//   k_sort_count    per-tile digit histogram -> cnt[digit][tile]
//   k_sort_scan     one workgroup per digit: exclusive scan of its row over the tiles (in place)
//                   and the row total
//   k_sort_scatter  ranks the tile's pairs stably (a wave walks its 512 pairs 64 at a time; the
//                   lanes of a round that hold the same digit find each other with eight ballots),
//                   orders them by digit in LDS and writes every digit's pairs of the tile as one
//                   contiguous piece (8 pairs = 128 bytes on average) at
//                   [pairs of smaller digits] + [same digit, earlier tiles].
// A library sort (rocPRIM onesweep) took 57 us for the two passes over 2^20 pairs -- eleven
// launches, five of them fills of its look-back state; this one takes six launches and no fills,
// and its workspace is sized without a device.
constexpr int kSortTile = 2048;                    // 1 Mi-board step: 194 us (1024: 203, 4096: 197)
constexpr int kSortRounds = kSortTile / kBlock;     // pairs per thread
constexpr int kSortWaves = kBlock / 64;
static_assert(kBlock == 256, "one thread per digit value");

__device__ __forceinline__ u64 match_digit(uint32_t d, bool active) {
  u64 m = __ballot(active);
#pragma unroll
  for (int bit = 0; bit < 8; ++bit) {
    const bool one = ((d >> bit) & 1u) != 0u;
    const u64 v = __ballot(one);
    m &= one ? v : ~v;
  }
  return m;                                          // the active lanes that hold digit d
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t x = __shfl_up(v, d);
    if (lane >= (uint32_t)d) v += x;
  }
  return v;
}
// exclusive scan over the block's 256 values; `ws` = kSortWaves words of LDS
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, volatile uint32_t* ws) {
  const uint32_t incl = wave_incl_scan(v);
  const uint32_t w = threadIdx.x >> 6;
  if ((threadIdx.x & 63u) == 63u) ws[w] = incl;
  __syncthreads();
  uint32_t off = 0;
  for (uint32_t k = 0; k < w; ++k) off += ws[k];
  __syncthreads();
  return off + incl - v;
}

// Group sums (batches of up to 8 Mi updates): besides cnt[digit][tile] the tiles of a group of `gsz`
// add their counts into gs[digit][group] -- 16-32 atomics per address -- and the scatter kernel adds up
// the groups before its own and the tiles before it inside its group itself (<= 2 sqrt(T) cached
// loads per digit): the separate scan launch (k_sort_scan, ~5 us of which most is the launch) is
// gone.  `gs_next` is the other pass's buffer, zeroed here for the count kernel that follows.
__global__ __launch_bounds__(kBlock) void k_sort_count(const u64* keys, int64_t B, int shift,
                                                       uint32_t dmask, uint32_t* cnt, int64_t T,
                                                       uint32_t* gs, uint32_t* gs_next, int n_groups, int gsz) {
  __shared__ uint32_t h[kBlock];
  h[threadIdx.x] = 0u;
  if (gs_next != nullptr)
    for (int idx = (int)(blockIdx.x * kBlock + threadIdx.x); idx < n_groups * kBlock; idx += (int)(gridDim.x * kBlock))
      gs_next[idx] = 0u;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kSortTile;
#pragma unroll
  for (int r = 0; r < kSortRounds; ++r) {
    const int64_t idx = base + r * kBlock + threadIdx.x;
    if (idx < B) atomicAdd(&h[(uint32_t)(keys[idx] >> shift) & dmask], 1u);
  }
  __syncthreads();
  if (gs == nullptr) {                       // k_sort_scan's layout: one row of T tile counts per digit
    cnt[(int64_t)threadIdx.x * T + blockIdx.x] = h[threadIdx.x];
  } else {                                   // [tile][digit] and [group][digit]: every access of this kernel
    cnt[(int64_t)blockIdx.x * kBlock + threadIdx.x] = h[threadIdx.x];       // and of the scatter is coalesced
    if (h[threadIdx.x] != 0u)
      atomicAdd(&gs[(int)(blockIdx.x / (unsigned)gsz) * kBlock + (int)threadIdx.x], h[threadIdx.x]);
  }
}

__global__ __launch_bounds__(kBlock) void k_sort_scan(uint32_t* cnt, int64_t T, uint32_t* total) {
  __shared__ uint32_t ws[kSortWaves];
  uint32_t* row = cnt + (int64_t)blockIdx.x * T;
  uint32_t carry = 0u;
  for (int64_t t0 = 0; t0 < T; t0 += kBlock) {
    const int64_t t = t0 + threadIdx.x;
    const uint32_t v = t < T ? row[t] : 0u;
    const uint32_t ex = block_excl_scan(v, ws);
    if (t < T) row[t] = carry + ex;
    if (threadIdx.x == kBlock - 1) ws[0] = ex + v;   // the chunk's sum (block_excl_scan ended on a barrier)
    __syncthreads();
    carry += ws[0];
    __syncthreads();
  }
  if (threadIdx.x == 0) total[blockIdx.x] = carry;
}

__global__ __launch_bounds__(kBlock) void k_sort_scatter(const u64* kin, const u64* vin, u64* kout,
                                                         u64* vout, int64_t B, int shift, uint32_t dmask,
                                                         const uint32_t* tile_before,
                                                         const uint32_t* total, int64_t T,
                                                         const uint32_t* gs, int n_groups, int gsz) {
  __shared__ u64 sk[kSortTile];
  __shared__ u64 sv[kSortTile];
  __shared__ uint32_t wcount[kSortWaves][kBlock];   // per wave and digit: pairs seen so far -> pairs of earlier waves
  __shared__ uint32_t first[kBlock];                // where the digit's pairs start within the tile
  __shared__ uint32_t dest[kBlock];                 // global index of the digit's piece minus first[]
  __shared__ uint32_t ws[kSortWaves];
  volatile uint32_t(*wc)[kBlock] = wcount;
  const uint32_t tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
#pragma unroll
  for (int k = 0; k < kSortWaves; ++k) wc[k][tid] = 0u;
  __syncthreads();
  // wave w owns pairs [w * 512, (w + 1) * 512) of the tile, round r lane l the pair r * 64 + l of them
  const int64_t base = (int64_t)blockIdx.x * kSortTile + (int64_t)w * (kSortTile / kSortWaves);
  u64 k[kSortRounds], v[kSortRounds];
  uint32_t meta[kSortRounds];                        // digit << 16 | rank among the wave's pairs of the digit
#pragma unroll
  for (int r = 0; r < kSortRounds; ++r) {
    const int64_t idx = base + r * 64 + lane;
    k[r] = idx < B ? kin[idx] : 0ull;
    v[r] = idx < B ? vin[idx] : 0ull;
  }
  // thread = digit: pairs of this digit in earlier tiles, and in all tiles (requested here, used after
  // the ranking rounds).  tile_before is k_sort_scan's output (gs == nullptr) or the raw counts
  uint32_t digit_before, digit_total;
  if (gs != nullptr) {
    const int grp = (int)(blockIdx.x / (unsigned)gsz);
    uint32_t acc = 0u, tot = 0u;
#pragma unroll 8                                         // independent loads: eight in flight, not one at a time
    for (int g = 0; g < n_groups; ++g) { const uint32_t x = gs[g * kBlock + (int)tid]; tot += x; acc += g < grp ? x : 0u; }
    const uint32_t* c = tile_before + ((int64_t)grp * gsz) * kBlock + tid;
    const int n_before = (int)((int64_t)blockIdx.x - (int64_t)grp * gsz);
#pragma unroll 8
    for (int t = 0; t < n_before; ++t) acc += c[(int64_t)t * kBlock];
    digit_before = acc; digit_total = tot;
  } else {
    digit_before = tile_before[(int64_t)tid * T + blockIdx.x];
    digit_total = total[tid];
  }
#pragma unroll
  for (int r = 0; r < kSortRounds; ++r) {
    const bool act = base + r * 64 + lane < B;
    const uint32_t d = (uint32_t)(k[r] >> shift) & dmask;
    const u64 m = match_digit(d, act);
    const uint32_t before = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    const uint32_t seen = wc[w][d];
    if (act && before == 0u) wc[w][d] = seen + (uint32_t)__popcll(m);
    meta[r] = (d << 16) | (seen + before);
  }
  __syncthreads();
  {                                                  // thread = digit
    const uint32_t c0 = wc[0][tid], c1 = wc[1][tid], c2 = wc[2][tid], c3 = wc[3][tid];
    wc[0][tid] = 0u; wc[1][tid] = c0; wc[2][tid] = c0 + c1; wc[3][tid] = c0 + c1 + c2;
    const uint32_t in_tile = block_excl_scan(c0 + c1 + c2 + c3, ws);
    const uint32_t smaller = block_excl_scan(digit_total, ws);
    first[tid] = in_tile;
    dest[tid] = smaller + digit_before - in_tile;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kSortRounds; ++r)
    if (base + r * 64 + lane < B) {
      const uint32_t d = meta[r] >> 16;
      const uint32_t at = first[d] + wc[w][d] + (meta[r] & 0xffffu);
      sk[at] = k[r];
      sv[at] = v[r];
    }
  __syncthreads();
  const int64_t left = B - (int64_t)blockIdx.x * kSortTile;
  const uint32_t n_tile = left < kSortTile ? (uint32_t)left : (uint32_t)kSortTile;
  for (uint32_t j = tid; j < n_tile; j += kBlock) {
    const u64 kk = sk[j];
    const uint32_t at = dest[(uint32_t)(kk >> shift) & dmask] + j;
    kout[at] = kk;
    vout[at] = sv[j];
  }
}

// Phase 2.  The updates arrive sorted -- stably, so env order survives -- by hash16(state, action)
// only (two radix passes instead of six over the whole word): 2^20 updates fall into ~65 000 runs
// of ~16, and inside a run the handful of distinct groups are told apart by comparing the full
// words.  Agent/main.py:43 applied update by update in env order, in double precision, rounded to
// float32 once per group and step -- td_fold(), the same operations in the same order on both paths
// below (for a group of one that is exactly td_value()).  One kernel, k_det_apply:
//   runs of up to kDetRun updates: the first update of every group in the run folds the group's
//     updates in env order and writes the cell (one lane per update scans its run; ~700 000
//     independent read-modify-writes in flight);
//   longer runs -- the states many envs share, e.g. right after a reset -- are folded by the WAVE of
//     their first update once its lanes are done with their own short runs (fold_long_run: one LANE
//     per group, the wave walks the run 64 updates at a time, broadcasts each update in order, and
//     the lane that owns its group folds it in; up to 64 groups per sweep of the run, more take
//     further sweeps).  Round 2 listed the long runs and folded them in a second launch; a launch
//     costs ~5 us here whether or not there is a long run, and mid-game there is none.
// kDetDone marks an update as applied (the sorted array is scratch).
constexpr int kDetRun = 64;
__device__ __forceinline__ float* det_cell(q2048_slot* table, u64 g) {
  return &table[(g & kDetSlotMask) >> 2].q[g & 3ull];
}
__device__ __forceinline__ u64 readlane64(u64 v, int lane) {
  return (u64)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane) |
         ((u64)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane) << 32);
}
// the whole wave (all 64 lanes active) folds the run that starts at `start` (wave-uniform)
__device__ __forceinline__ void fold_long_run(q2048_slot* table, u64* group, const double* target, int64_t B,
                                              double lr, u64 run_mask, int64_t start, int lane) {
  const u64 g0 = group[start];
  int64_t len = 0;                                     // wave-uniform search for the run's end
  for (;;) {
    const int64_t pos = start + len + (int64_t)lane;
    const u64 same = __ballot(pos < B && ((group[pos] ^ g0) & run_mask) == 0ull);
    if (same != ~0ull) { len += (int64_t)(__ffsll((long long)~same) - 1); break; }
    len += 64;
  }
  for (;;) {                                           // one sweep: the first <= 64 open groups
    // (a) which groups: in order of first appearance, lane k owns the k-th
    u64 mine = 0ull;                                   // 0 = none (owned words carry the kDetDone tag)
    int n_owned = 0;
    bool more = false;                                 // open groups beyond the 64 of this sweep
    for (int64_t c = 0; c < len && !more; c += 64) {
      const bool valid = c + lane < len;
      const u64 g = valid ? group[start + c + lane] : kDetDone;
      u64 open = __ballot((g & (kDetDone | kDetDrop)) == 0ull);
      while (open != 0ull) {
        const int k = __ffsll((long long)open) - 1;
        open &= open - 1ull;
        const u64 gk = readlane64(g, k) | kDetDone;    // tagged: never equals the "none" value 0
        if (__ballot(mine == gk) != 0ull) continue;
        if (n_owned == 64) { more = true; break; }
        if (lane == n_owned) mine = gk;
        ++n_owned;
      }
    }
    if (n_owned == 0) break;
    // (b) every owner reads its cell (one round trip for the whole sweep), then (c) the run is
    // walked again: each open update is broadcast in order and folded by its group's lane
    float* cell = det_cell(table, mine);
    double q = lane < n_owned ? (double)*cell : 0.0;
    for (int64_t c = 0; c < len; c += 64) {
      const bool valid = c + lane < len;
      const u64 g = valid ? group[start + c + lane] : kDetDone;
      const double t = valid ? target[start + c + lane] : 0.0;
      u64 open = __ballot((g & (kDetDone | kDetDrop)) == 0ull);
      bool taken = false;
      while (open != 0ull) {
        const int k = __ffsll((long long)open) - 1;
        open &= open - 1ull;
        const u64 gk = readlane64(g, k) | kDetDone;
        const double tk = __longlong_as_double((long long)readlane64((u64)__double_as_longlong(t), k));
        const bool own = mine == gk;
        if (own) q = td_fold(q, tk, lr);               // Agent/main.py:43, env order
        if (__ballot(own) != 0ull && lane == k) taken = true;
      }
      if (taken) group[start + c + lane] = g | kDetDone;
    }
    if (lane < n_owned) *cell = (float)q;
    if (!more) break;
  }
}
__global__ __launch_bounds__(kBlock) void k_det_apply(q2048_slot* table, u64* group,
                                                      const double* target, int64_t B, double lr,
                                                      u64 run_mask) {
  // the block's 256 sorted words and kDetRun neighbours on either side, staged once: every
  // update scans its run (tens of words) and that traffic belongs in LDS, not in the L1
  __shared__ u64 tile[kBlock + 2 * kDetRun];
  const int64_t base = (int64_t)blockIdx.x * kBlock;
  for (int t = threadIdx.x; t < kBlock + 2 * kDetRun; t += kBlock) {
    const int64_t at = base - kDetRun + t;
    tile[t] = (at >= 0 && at < B) ? group[at] : 0ull;
  }
  __syncthreads();
  const int64_t j = base + threadIdx.x;
  const bool live = j < B;                               // (no early return: the wave stays whole for the long runs)
  const int64_t jj = live ? j : B - 1;
  const u64* w = tile + kDetRun + (live ? (int)threadIdx.x : (int)(B - 1 - base));   // w[k] = group[jj + k], |k| <= kDetRun
  const u64 g = w[0];
  // the cell and this update's own target are requested NOW, before the run is scanned: the scattered
  // read is the long pole of the kernel and the scan (tens of LDS reads) hides under it.  Every update
  // asks, also the few that turn out to be folded by an earlier one or dropped (slot bits 0 then: a
  // valid address)
  float* cell = det_cell(table, g);
  const float cell0 = *cell;
  const double target0 = target[jj];
  // is an earlier update of the run in the same group (then that one folds this one in), and how
  // far does the run go
  bool first = true;
  int back = 0, fwd = 0;
  while (back < kDetRun && jj - back - 1 >= 0 && ((w[-back - 1] ^ g) & run_mask) == 0ull) {
    first = first && w[-back - 1] != g;
    ++back;
  }
  while (fwd < kDetRun && jj + fwd + 1 < B && ((w[fwd + 1] ^ g) & run_mask) == 0ull) ++fwd;
  const bool long_run = back + fwd + 1 > kDetRun;        // every update of a long run sees that
  if (live && !long_run && first && (g & kDetDrop) == 0ull) {
    double q = td_fold((double)cell0, target0, lr);                        // Agent/main.py:43, env order
    for (int k = 1; k <= fwd; ++k)
      if (w[k] == g) q = td_fold(q, target[j + k], lr);
    *cell = (float)q;
  }
  // long runs whose first update sits in this wave: the wave folds them, one after the other
  const int lane = (int)(threadIdx.x & 63u);
  u64 heads = __ballot(live && long_run && back == 0);
  while (heads != 0ull) {
    const int k = __ffsll((long long)heads) - 1;
    heads &= heads - 1ull;
    fold_long_run(table, group, target, B, lr, run_mask, (int64_t)readlane64((u64)j, k), lane);
  }
}

// ---------------------------------------------------------------------------------------------
// row-tuple linear Q (BASELINE configs[1]), 4x4 only.  W = float[4][65536][4] (4 MiB: lives in
// L2 / Infinity Cache).  Reads are agent-scope loads of whole 16-byte entries (one request per
// entry; the fused loop gathers each state once, as s', and carries it into the next step: 4
// loads + 4 stores per env step instead of 16 + 4).  A write stores
// old + d with the value the lane gathered (last writer wins): hot row entries are shared by
// most of the batch, and summing every lane's delta (atomic add) multiplies the step size by the
// number of concurrent lanes and diverges.  With one lane this is the sequential learner.
// ---------------------------------------------------------------------------------------------
struct RtRows { Row e0, e1, e2, e3; };
__device__ __forceinline__ const float* rt_addr(const float* w, uint32_t r, uint32_t idx) {
  return w + (((size_t)r * kRtIdx + idx) << 2);
}
__device__ __forceinline__ Row rt_row(const u32x4& v) {
  return Row{bits_f32(v.x), bits_f32(v.y), bits_f32(v.z), bits_f32(v.w)};
}
// the four entries of a board: four 16-byte agent-scope loads in flight together, one wait
__device__ __forceinline__ RtRows rt_gather(const float* w, const Board& b) {
  u32x4 v0, v1, v2, v3;
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
      "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
      : "v"(rt_addr(w, 0, pack_row(b.r0))), "v"(rt_addr(w, 1, pack_row(b.r1))),
        "v"(rt_addr(w, 2, pack_row(b.r2))), "v"(rt_addr(w, 3, pack_row(b.r3)))
      : "memory");
  return RtRows{rt_row(v0), rt_row(v1), rt_row(v2), rt_row(v3)};
}
__device__ __forceinline__ Row rt_q(const RtRows& e) {
  return Row{rt_sum(e.e0.q0, e.e1.q0, e.e2.q0, e.e3.q0), rt_sum(e.e0.q1, e.e1.q1, e.e2.q1, e.e3.q1),
             rt_sum(e.e0.q2, e.e1.q2, e.e2.q2, e.e3.q2), rt_sum(e.e0.q3, e.e1.q3, e.e2.q3, e.e3.q3)};
}
__device__ __forceinline__ void rt_scatter(float* w, const Board& b, const RtRows& e, int act, float d) {
  float* base = w + act;
  base[((size_t)0 * kRtIdx + pack_row(b.r0)) << 2] = row_get(e.e0, act) + d;
  base[((size_t)1 * kRtIdx + pack_row(b.r1)) << 2] = row_get(e.e1, act) + d;
  base[((size_t)2 * kRtIdx + pack_row(b.r2)) << 2] = row_get(e.e2, act) + d;
  base[((size_t)3 * kRtIdx + pack_row(b.r3)) << 2] = row_get(e.e3, act) + d;
}

__global__ __launch_bounds__(kBlock) void k_rt_choose(const float* w, const uint8_t* boards, int64_t B,
                                                      double eps, uint64_t seed, uint64_t env_id0,
                                                      uint32_t ctr, uint8_t* actions) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= B) return;
  Stage<4> st;
  const Board b = load_board(boards, i, B, st);
  const Draws x = draws(seed, env_id0 + (uint64_t)i, ctr, kStreamStep);
  int act;
  if (draw_uniform(x.x0) < eps) act = draw_action(x.x1);
  else { const Row q = rt_q(rt_gather(w, b)); act = argmax4(q.q0, q.q1, q.q2, q.q3); }
  actions[i] = (uint8_t)act;
}

__global__ __launch_bounds__(kBlock) void k_rt_lookup(const float* w, const uint8_t* boards, int64_t B,
                                                      float* q_out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= B) return;
  Stage<4> st;
  const Row q = rt_q(rt_gather(w, load_board(boards, i, B, st)));
  reinterpret_cast<float4*>(q_out)[i] = make_float4(q.q0, q.q1, q.q2, q.q3);
}

__global__ __launch_bounds__(kBlock) void k_rt_update(float* w, const uint8_t* s, const uint8_t* actions,
                                                      const float* reward, const uint8_t* s2,
                                                      const uint8_t* done, int64_t B, double lr,
                                                      double gamma, uint32_t* status) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= B) return;
  Stage<4> st;
  const int act = actions[i];
  if (act > 3) { atomicOr(status, Q2048_STATUS_BAD_ACTION); return; }
  const Board b_s = load_board(s, i, B, st), b_n = load_board(s2, i, B, st);
  const Row qn = rt_q(rt_gather(w, b_n));
  const RtRows es = rt_gather(w, b_s);
  const float d = rt_delta(row_get(rt_q(es), act), reward[i], max4(qn.q0, qn.q1, qn.q2, qn.q3),
                           done[i] != 0, lr, gamma);
  rt_scatter(w, b_s, es, act, d);
}

__global__ __launch_bounds__(kBlock) void k_rt_fused_rollout(
    uint8_t* boards, q2048_aux* aux, float* w, int64_t B, int steps, double eps, double lr,
    double gamma, uint64_t seed, uint64_t env_id0, uint32_t ctr0, int64_t* stats_i, double* stats_f,
    uint32_t* /*status: nothing data-dependent can go wrong on this path*/) {
  __shared__ BlockStats bs;
  stats_clear(bs);
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < B) {
    Stage<4> st;
    const uint64_t id = env_id0 + (uint64_t)i;
    Board b = load_board(boards, i, B, st);
    Aux a = ld_aux(aux, i);
    uint32_t n_valid = 0, n_explore = 0, n_done = 0;
    double reward_sum = 0.0;
    RtRows es = rt_gather(w, b);   // entries of the current state, carried from step to step
    for (int t = 0; t < steps; ++t) {
      const Draws x = draws(seed, id, ctr0 + (uint32_t)t, kStreamStep);
      const Board s = b;
      const Row q = rt_q(es);
      bool explored;
      const int act = eps_greedy(eps, x.x0, x.x1, q.q0, q.q1, q.q2, q.q3, explored);
      const StepOut o = env_step(b, a, act, x.x2, x.x3);
      RtRows en = rt_gather(w, b);
      const Row qn = rt_q(en);
      const float d = rt_delta(row_get(q, act), o.reward, max4(qn.q0, qn.q1, qn.q2, qn.q3),
                               o.done != 0, lr, gamma);
      rt_scatter(w, s, es, act, d);
      n_valid += wave_count(o.valid != 0);
      n_explore += wave_count(explored);
      n_done += wave_count(o.done != 0);
      reward_sum += (double)o.reward;
      if (o.done) {
        episode_stats(bs, a, o.max_log2);
        begin_episode(b, a, seed, id);
        es = rt_gather(w, b);
      } else {
        // s' becomes s: its entries were gathered before the write above, so a row that did not
        // move (same index in the same position) takes the value just written
        if (pack_row(b.r0) == pack_row(s.r0)) row_set(en.e0, act, row_get(es.e0, act) + d);
        if (pack_row(b.r1) == pack_row(s.r1)) row_set(en.e1, act, row_get(es.e1, act) + d);
        if (pack_row(b.r2) == pack_row(s.r2)) row_set(en.e2, act, row_get(es.e2, act) + d);
        if (pack_row(b.r3) == pack_row(s.r3)) row_set(en.e3, act, row_get(es.e3, act) + d);
        es = en;
      }
    }
    store_board(boards, i, B, b, st);
    st_aux(aux, i, a);
    atomicAdd(&bs.f[Q2048_SF_REWARD], reward_sum);
    const uint32_t n_active = wave_count(true);
    if (wave_leader()) {
      atomicAdd(&bs.i[Q2048_ST_STEPS], (u64)n_active * (u64)steps);
      atomicAdd(&bs.i[Q2048_ST_VALID], (u64)n_valid);
      atomicAdd(&bs.i[Q2048_ST_EXPLORE], (u64)n_explore);
      atomicAdd(&bs.i[Q2048_ST_EPISODES], (u64)n_done);
    }
  }
  stats_flush(bs, stats_i, stats_f);
}

// ---------------------------------------------------------------------------------------------
// table utilities
// ---------------------------------------------------------------------------------------------
// Streams the whole table once, in tiles of 16 Ki slots per block: a lane reads the first 16 bytes
// ({key, q0, q1}) of 64 slots, four in flight at a time, consecutive lanes on consecutive slots,
// and keeps one occupancy bit per slot.  Counting: the per-lane totals are reduced per block and
// the block issues ONE global atomic at the very end (same-address atomics run at ~10^8/s: one
// per occupied wave made the first version of this scan 10x slower than the memory system).
// Exporting: per tile the block reserves its output range with one atomic, an LDS scan gives
// every lane its offset, and only the occupied slots are read again for their second half.
// Non-temporal loads: a scan must not evict the rows a rollout is working on.
constexpr int kTilePerLane = 64;                        // slots per lane per tile (one mask word)
constexpr u64 kTileSlots = (u64)kBlock * kTilePerLane;  // 16 Ki slots = 512 KiB of table
__global__ __launch_bounds__(kBlock) void k_table_export(const q2048_slot* table, u64 cap,
                                                         u64* keys_out, float* q_out, int64_t max_rows,
                                                         int key_words, u64* count) {
  __shared__ uint32_t wave_sum[kBlock / 64];
  __shared__ u64 tile_base;
  const u32x4* t16 = reinterpret_cast<const u32x4*>(table);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  u64 total = 0ull;                                     // count-only: this lane's occupied slots
  for (u64 tile = (u64)blockIdx.x * kTileSlots; tile < cap; tile += (u64)gridDim.x * kTileSlots) {
    u64 occ = 0ull;
#pragma unroll 1
    for (int j = 0; j < kTilePerLane; j += 4) {
      u32x4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const u64 i = tile + (u64)(j + k) * kBlock + threadIdx.x;
        v[k] = u32x4{0u, 0u, 0u, 0u};
        if (i < cap) v[k] = __builtin_nontemporal_load(&t16[2ull * i]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) occ |= (u64)((v[k].x | v[k].y) != 0u) << (j + k);
    }
    const uint32_t mine = (uint32_t)__popcll(occ);
    if (keys_out == nullptr) { total += mine; continue; }
    // block-wide exclusive scan of `mine`: wave prefix by shuffles, wave totals through LDS
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = __shfl_up(incl, d);
      if (lane >= (uint32_t)d) incl += up;
    }
    if (lane == 63u) wave_sum[wave] = incl;
    __syncthreads();
    uint32_t before = 0, block_total = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) {
      before += (uint32_t)w < wave ? wave_sum[w] : 0u;
      block_total += wave_sum[w];
    }
    if (threadIdx.x == 0) tile_base = block_total ? atomicAdd(count, (u64)block_total) : 0ull;
    __syncthreads();
    u64 at = tile_base + before + (incl - mine);
    while (occ != 0ull) {
      const int j = __ffsll((long long)occ) - 1;
      occ &= occ - 1ull;
      if ((int64_t)at < max_rows) {
        const u64 i = tile + (u64)j * kBlock + threadIdx.x;
        const u32x4 a = __builtin_nontemporal_load(&t16[2ull * i]);          // {key, q0, q1}
        const u32x4 w = __builtin_nontemporal_load(&t16[2ull * i + 1ull]);   // {q2, q3, second key word}
        keys_out[at * (u64)key_words] = (u64)a.x | ((u64)a.y << 32);
        if (key_words == 2) keys_out[at * 2ull + 1ull] = (u64)w.z | ((u64)w.w << 32);
        reinterpret_cast<u32x4*>(q_out)[at] = u32x4{a.z, a.w, w.x, w.y};
      }
      ++at;
    }
    __syncthreads();                                    // wave_sum / tile_base are reused
  }
  if (keys_out == nullptr) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) total += __shfl_xor(total, d);
    __shared__ u64 wave_total[kBlock / 64];
    if (lane == 0u) wave_total[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
      u64 sum = 0ull;
#pragma unroll
      for (int w = 0; w < kBlock / 64; ++w) sum += wave_total[w];
      if (sum) atomicAdd(count, sum);
    }
  }
}

template <int WORDS>
__global__ __launch_bounds__(kBlock) void k_table_import(q2048_slot* table, u64 mask, const u64* keys,
                                                         const float* q, int64_t rows, uint32_t* status) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= rows) return;
  typename Geo<WORDS == 1 ? 4 : 5>::Key key;
  key.k0 = keys[i * WORDS];
  if constexpr (WORDS == 2) key.k1 = keys[i * 2 + 1];
  bool inserted;
  const int64_t slot = probe_insert(table, mask, key, key_home(key, mask), inserted, kMaxProbe);
  if (slot < 0) { atomicOr(status, Q2048_STATUS_TABLE_FULL); return; }
  // placed beyond the learning paths' probe limit (only at loads above ~0.93): q_lookup / export find the row,
  // choose / update / the rollouts read it as absent -- the caller is told
  if (seq_pos(seq_of(key_hash(key), mask), (u64)slot) >= probe_limit(mask, kRolloutProbe)) atomicOr(status, Q2048_STATUS_DEEP_ROW);
  const float4 v = reinterpret_cast<const float4*>(q)[i];
  table[slot].q[0] = v.x; table[slot].q[1] = v.y; table[slot].q[2] = v.z; table[slot].q[3] = v.w;
}

// Growth (q2048_table_grow): every occupied row of `old_t` is inserted into `new_t` (zero-filled, larger)
// in ONE streaming pass -- consecutive lanes read consecutive slots (32 B each, non-temporal), an
// occupied one claims its slot in the new table with the find-or-create of the rollout (the first
// access is the claiming compare-and-swap at the key's new home) and writes its four values.  Keys of
// one table are distinct, so no two lanes ever insert the same key; counters[0] += rows moved,
// counters[1] += rows that found no slot or found their key already there (expected: 0; the host
// refuses the new table otherwise).
template <int WORDS>
__global__ __launch_bounds__(kBlock) void k_table_rehash(const q2048_slot* old_t, u64 old_cap, q2048_slot* new_t,
                                                         u64 new_mask, u64* counters) {
  const u32x4* t16 = reinterpret_cast<const u32x4*>(old_t);
  u64 moved = 0ull, failed = 0ull;
  for (u64 i = (u64)blockIdx.x * kBlock + threadIdx.x; i < old_cap; i += (u64)gridDim.x * kBlock) {
    const u32x4 a = __builtin_nontemporal_load(&t16[2ull * i]);          // {key, q0, q1}
    const u64 k = (u64)a.x | ((u64)a.y << 32);
    if (k == 0ull) continue;
    const u32x4 w = __builtin_nontemporal_load(&t16[2ull * i + 1ull]);   // {q2, q3, second key word}
    typename Geo<WORDS == 1 ? 4 : 5>::Key key;
    key.k0 = k;
    if constexpr (WORDS == 2) key.k1 = (u64)w.z | ((u64)w.w << 32);
    bool inserted;
    const int64_t slot = probe_insert(new_t, new_mask, key, key_home(key, new_mask), inserted, kMaxProbe);
    if (slot < 0 || !inserted) { ++failed; continue; }
    uint2* q = reinterpret_cast<uint2*>(new_t[slot].q);                  // 8-byte aligned (offset 8 of a 32-B slot)
    q[0] = make_uint2(a.z, a.w);
    q[1] = make_uint2(w.x, w.y);
    ++moved;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { moved += __shfl_xor(moved, d); failed += __shfl_xor(failed, d); }
  __shared__ u64 wm[kBlock / 64], wf[kBlock / 64];
  if ((threadIdx.x & 63u) == 0u) { wm[threadIdx.x >> 6] = moved; wf[threadIdx.x >> 6] = failed; }
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 m = 0ull, f = 0ull;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) { m += wm[w]; f += wf[w]; }
    if (m) atomicAdd(&counters[0], m);
    if (f) atomicAdd(&counters[1], f);
  }
}

// Placement probe: `steps` scattered device-scope atomic ORs of 0 per lane into key words chosen
// like the rollout chooses rows -- the table's write-side request pattern with no effect on its
// contents (x | 0 == x).  The host times it: where in device memory a table lies moves the
// scattered write / atomic rate by ~20 % (DESIGN.md 4 "table placement"), reads not at all.
// `zero` is a run-time 0 so that the operation stays an atomic.
__global__ __launch_bounds__(kBlock) void k_table_probe(q2048_slot* table, u64 mask, int64_t lanes,
                                                        int steps, uint64_t seed, uint32_t zero) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= lanes) return;
  u64 x = mix64(seed ^ ((u64)i << 20));
  for (int t = 0; t < steps; ++t) {
    x = mix64(x + (u64)t + 1ull);
    __hip_atomic_fetch_or(reinterpret_cast<uint32_t*>(&table[x & mask].key), zero, __ATOMIC_RELAXED,
                          __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---------------------------------------------------------------------------------------------
// host side of the ABI
// ---------------------------------------------------------------------------------------------
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline unsigned grid_for(int64_t B) { return (unsigned)((B + kBlock - 1) / kBlock); }
inline int launch_status() { return hipGetLastError() == hipSuccess ? Q2048_OK : Q2048_ERR_LAUNCH; }
inline int check_batch(int64_t B, int n) {
  if (n != 4 && n != 5) return Q2048_ERR_UNSUPPORTED;
  // one block per 256 envs and HIP caps grid.x at 2^31 - 1 blocks
  if (B < 0 || B > (int64_t)0x7fffffff * kBlock) return Q2048_ERR_SIZE;
  return Q2048_OK;
}
inline int check_table(const void* table, int cap_log2) {
  if (table == nullptr) return Q2048_ERR_NULL;
  if (cap_log2 < 4 || cap_log2 > 40) return Q2048_ERR_SIZE;
  if (!aligned16(table)) return Q2048_ERR_ALIGN;
  return Q2048_OK;
}
// launches kernel<4> or kernel<5>
#define Q2048_LAUNCH(kernel, n, B, stream, ...)                                                   \
  do {                                                                                            \
    if ((n) == 4)                                                                                 \
      hipLaunchKernelGGL(kernel<4>, dim3(grid_for(B)), dim3(kBlock), 0, (hipStream_t)(stream),    \
                         __VA_ARGS__);                                                            \
    else                                                                                          \
      hipLaunchKernelGGL(kernel<5>, dim3(grid_for(B)), dim3(kBlock), 0, (hipStream_t)(stream),    \
                         __VA_ARGS__);                                                            \
  } while (0)
// launches kernel<4, ENV> or kernel<5, ENV>, ENV = the env profile bits of `flags`
#define Q2048_LAUNCH_ENV_CASE(kernel, E, n, B, stream, ...)                                       \
  case E:                                                                                         \
    if ((n) == 4)                                                                                 \
      hipLaunchKernelGGL((kernel<4, E>), dim3(grid_for(B)), dim3(kBlock), 0,                      \
                         (hipStream_t)(stream), __VA_ARGS__);                                     \
    else                                                                                          \
      hipLaunchKernelGGL((kernel<5, E>), dim3(grid_for(B)), dim3(kBlock), 0,                      \
                         (hipStream_t)(stream), __VA_ARGS__);                                     \
    break;
#define Q2048_LAUNCH_ENV(kernel, flags, n, B, stream, ...)                                        \
  switch (env_bits(flags) & 3) {                                                                  \
    Q2048_LAUNCH_ENV_CASE(kernel, 0, n, B, stream, __VA_ARGS__)                                   \
    Q2048_LAUNCH_ENV_CASE(kernel, 1, n, B, stream, __VA_ARGS__)                                   \
    Q2048_LAUNCH_ENV_CASE(kernel, 2, n, B, stream, __VA_ARGS__)                                   \
    Q2048_LAUNCH_ENV_CASE(kernel, 3, n, B, stream, __VA_ARGS__)                                   \
  }
// the fused rollout: ENV with the play-only bit, MODE = what the launch does with the table
#define Q2048_LAUNCH_FUSED_ONE(NN, E, M, BLK, B, stream, ...)                                      \
  hipLaunchKernelGGL((k_fused_rollout<NN, E, M, BLK>), dim3((unsigned)(((B) + (BLK) - 1) / (BLK))), dim3(BLK), 0, \
                     (hipStream_t)(stream), __VA_ARGS__)
#define Q2048_LAUNCH_FUSED_CASE(E, M, n, B, stream, ...)                                          \
  case (E) * 16 + (M):                                                                            \
    if ((n) == 4) {                                                                               \
      if ((B) >= kFusedBigBatch) Q2048_LAUNCH_FUSED_ONE(4, E, M, kFusedBlockBig, B, stream, __VA_ARGS__);     \
      else Q2048_LAUNCH_FUSED_ONE(4, E, M, kFusedBlockSmall, B, stream, __VA_ARGS__);             \
    } else {                                                                                      \
      if ((B) >= kFusedBigBatch) Q2048_LAUNCH_FUSED_ONE(5, E, M, kFusedBlockBig, B, stream, __VA_ARGS__);     \
      else Q2048_LAUNCH_FUSED_ONE(5, E, M, kFusedBlockSmall, B, stream, __VA_ARGS__);             \
    }                                                                                             \
    break;
#define Q2048_LAUNCH_FUSED_CASE4(E, M, B, stream, ...)   /* 4x4 only (fused_mode never asks for it on 5x5) */ \
  case (E) * 16 + (M):                                                                            \
    if ((B) >= kFusedBigBatch) Q2048_LAUNCH_FUSED_ONE(4, E, M, kFusedBlockBig, B, stream, __VA_ARGS__);       \
    else Q2048_LAUNCH_FUSED_ONE(4, E, M, kFusedBlockSmall, B, stream, __VA_ARGS__);               \
    break;
#define Q2048_LAUNCH_FUSED_ENV(E, n, B, stream, ...)                                              \
  Q2048_LAUNCH_FUSED_CASE(E, kModeLearn, n, B, stream, __VA_ARGS__)                               \
  Q2048_LAUNCH_FUSED_CASE(E, kModeCas, n, B, stream, __VA_ARGS__)                                 \
  Q2048_LAUNCH_FUSED_CASE(E, kModeEval, n, B, stream, __VA_ARGS__)                                \
  Q2048_LAUNCH_FUSED_CASE(E, kModeFrozen, n, B, stream, __VA_ARGS__)                              \
  Q2048_LAUNCH_FUSED_CASE(E, kModeFrozen + kModeCas, n, B, stream, __VA_ARGS__)                   \
  Q2048_LAUNCH_FUSED_CASE4(E, kModeFrozen + kModeSummary, B, stream, __VA_ARGS__)                 \
  Q2048_LAUNCH_FUSED_CASE4(E, kModeFrozen + kModeSummary + kModeCas, B, stream, __VA_ARGS__)
#define Q2048_LAUNCH_FUSED(flags, n, B, stream, ...)                                              \
  switch (env_bits(flags) * 16 + fused_mode(flags, n)) {                                          \
    Q2048_LAUNCH_FUSED_ENV(0, n, B, stream, __VA_ARGS__)                                          \
    Q2048_LAUNCH_FUSED_ENV(1, n, B, stream, __VA_ARGS__)                                          \
    Q2048_LAUNCH_FUSED_ENV(2, n, B, stream, __VA_ARGS__)                                          \
    Q2048_LAUNCH_FUSED_ENV(3, n, B, stream, __VA_ARGS__)                                          \
    Q2048_LAUNCH_FUSED_CASE(4, kModeLearn, n, B, stream, __VA_ARGS__)                             \
    Q2048_LAUNCH_FUSED_CASE(5, kModeLearn, n, B, stream, __VA_ARGS__)                             \
    Q2048_LAUNCH_FUSED_CASE(6, kModeLearn, n, B, stream, __VA_ARGS__)                             \
    Q2048_LAUNCH_FUSED_CASE(7, kModeLearn, n, B, stream, __VA_ARGS__)                             \
  }
inline int env_bits(uint32_t flags) {
  return ((flags & Q2048_FLAG_ENV_DQN) ? kEnvDqn : 0) | ((flags & Q2048_FLAG_RESET_SHAPING) ? kEnvResetShaping : 0) |
         ((flags & Q2048_FLAG_PLAY_ONLY) ? kEnvPlayOnly : 0);
}
inline int fused_mode(uint32_t flags, int n) {   // play-only launches touch no table: one instantiation
  if (flags & Q2048_FLAG_PLAY_ONLY) return kModeLearn;
  if (flags & Q2048_FLAG_NO_LEARN) return kModeEval;   // (evaluation creates nothing anyway)
  const bool frozen = (flags & Q2048_FLAG_NO_NEW_ROWS) != 0u;
  return ((flags & Q2048_FLAG_TD_CAS) ? kModeCas : kModeLearn) | (frozen ? kModeFrozen : 0) |
         ((frozen && n == 4 && (flags & Q2048_FLAG_LINE_SUMMARY)) ? kModeSummary : 0);
}
// flag bits outside the ABI are an argument error (experiment builds also take bits 8..23)
constexpr uint32_t kAbiFlags = Q2048_FLAG_INDEPENDENT | Q2048_FLAG_SINGLE_ENV | Q2048_FLAG_TD_CAS |
                               Q2048_FLAG_ENV_DQN | Q2048_FLAG_RESET_SHAPING | Q2048_FLAG_PLAY_ONLY |
                               Q2048_FLAG_NO_LEARN | Q2048_FLAG_NO_NEW_ROWS | Q2048_FLAG_LINE_SUMMARY;
inline int check_flags(uint32_t flags, uint32_t refused = 0u) {
  uint32_t allowed = kAbiFlags;
#ifdef Q2048_EXPERIMENTS
  allowed |= 0x00ffff00u;
#endif
  return ((flags & ~allowed) || (flags & refused)) ? Q2048_ERR_FLAGS : Q2048_OK;
}
}  // namespace

extern "C" {

int q2048_abi_version(void) { return Q2048_ABI_VERSION; }

#ifdef Q2048_EXPERIMENTS
int q2048_debug_timeline(unsigned long long* stamps) {   // device uint64[blocks][8], or NULL to switch off
  return hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &stamps, sizeof(stamps)) == hipSuccess ? Q2048_OK : Q2048_ERR_LAUNCH;
}
#endif

int q2048_claim_timeouts(uint64_t* count_host) {
  if (count_host == nullptr) return Q2048_ERR_NULL;
  unsigned long long v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_claim_timeouts), sizeof(v)) != hipSuccess) return Q2048_ERR_LAUNCH;
  *count_host = (uint64_t)v;
  return Q2048_OK;
}
size_t q2048_sizeof_aux(void) { return sizeof(q2048_aux); }
size_t q2048_sizeof_slot(void) { return sizeof(q2048_slot); }

const char* q2048_strerror(int code) {
  switch (code) {
    case Q2048_OK: return "ok";
    case Q2048_ERR_NULL: return "a required pointer is NULL";
    case Q2048_ERR_SIZE: return "size out of range (batch, steps, cap_log2 or key_words)";
    case Q2048_ERR_ALIGN: return "boards/aux/table must be 16-byte aligned";
    case Q2048_ERR_UNSUPPORTED: return "unsupported board side (n must be 4 or 5)";
    case Q2048_ERR_LAUNCH: return "HIP launch failed";
    case Q2048_ERR_RANGE: return "scalar out of range (eps in [0,1], lr and gamma finite)";
    case Q2048_ERR_FLAGS: return "flag bits this entry point does not take";
    case Q2048_ERR_ALLOC: return "device memory could not be reserved, created or mapped";
    case Q2048_ERR_VERIFY: return "a table failed its self-check (a fresh table not all zeros, or rows lost while growing)";
    case Q2048_ERR_BUSY: return "the table already takes part in a growth (finish or abort that one first)";
    case Q2048_PENDING: return "still working (not an error)";
    default: return "unknown error";
  }
}

int q2048_env_init(uint8_t* boards, q2048_aux* aux, int64_t B, int n, uint64_t seed,
                   uint64_t env_id0, void* stream) {
  if (int e = check_batch(B, n)) return e;
  if (boards == nullptr || aux == nullptr) return Q2048_ERR_NULL;
  if (!aligned16(boards) || !aligned16(aux)) return Q2048_ERR_ALIGN;
  if (B == 0) return Q2048_OK;
  Q2048_LAUNCH(k_env_init, n, B, stream, boards, aux, B, seed, env_id0);
  return launch_status();
}

int q2048_env_reset_ex(uint8_t* boards, q2048_aux* aux, const uint8_t* mask, int64_t B, int n,
                       uint64_t seed, uint64_t env_id0, uint32_t flags, void* stream) {
  if (int e = check_batch(B, n)) return e;
  if (int e = check_flags(flags)) return e;
  if (boards == nullptr || aux == nullptr) return Q2048_ERR_NULL;
  if (!aligned16(boards) || !aligned16(aux)) return Q2048_ERR_ALIGN;
  if (B == 0) return Q2048_OK;
  Q2048_LAUNCH(k_env_reset, n, B, stream, boards, aux, mask, B, seed, env_id0, flags);
  return launch_status();
}

int q2048_env_reset(uint8_t* boards, q2048_aux* aux, const uint8_t* mask, int64_t B, int n,
                    uint64_t seed, uint64_t env_id0, void* stream) {
  return q2048_env_reset_ex(boards, aux, mask, B, n, seed, env_id0, 0u, stream);
}

static int env_step_impl(const uint8_t* boards_in, uint8_t* boards, q2048_aux* aux, const uint8_t* actions,
                         int64_t B, int n, uint64_t seed, uint64_t env_id0, uint32_t ctr, uint32_t flags,
                         float* reward, uint8_t* done, uint8_t* max_log2, int32_t* max_tile,
                         uint32_t* status, const uint32_t* draw_pos, const uint32_t* draw_val,
                         const uint32_t* draw_opos, const uint32_t* draw_oval, int draw_stride,
                         void* stream) {
  if (int e = check_batch(B, n)) return e;
  if (int e = check_flags(flags)) return e;
  if (!boards_in || !boards || !aux || !actions || !reward || !done || !max_log2 || !status) return Q2048_ERR_NULL;
  if (!aligned16(boards_in) || !aligned16(boards) || !aligned16(aux)) return Q2048_ERR_ALIGN;
  if (boards_in != boards) {       // two buffers must not overlap (lanes of one block read and write at different times)
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(boards_in), a1 = reinterpret_cast<uintptr_t>(boards);
    const uintptr_t len = (uintptr_t)B * (uintptr_t)(n * n);
    if (a0 < a1 + len && a1 < a0 + len) return Q2048_ERR_ALIGN;
  }
  if (B == 0) return Q2048_OK;
  if (n == 4 && draw_pos == nullptr) {
    unsigned blocks = grid_for(B);
#ifdef Q2048_EXPERIMENTS   // bits 8..11: boards per thread (k_env_step4_pipelined, in place only)
    const int64_t per_thread = ((flags >> 8) & 15u) ? (int64_t)((flags >> 8) & 15u) : 1;
    if (per_thread > 1 && boards_in == boards && max_tile == nullptr) {
      blocks = (unsigned)((B + kBlock * per_thread - 1) / (kBlock * per_thread));
      if (flags & Q2048_FLAG_ENV_DQN)
        hipLaunchKernelGGL(k_env_step4_pipelined<kEnvDqn>, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream,
                           boards, aux, actions, B, seed, env_id0, ctr, reward, done, max_log2, status);
      else
        hipLaunchKernelGGL(k_env_step4_pipelined<0>, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream,
                           boards, aux, actions, B, seed, env_id0, ctr, reward, done, max_log2, status);
      return launch_status();
    }
#endif
    if (flags & Q2048_FLAG_ENV_DQN)
      hipLaunchKernelGGL(k_env_step4<kEnvDqn>, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, boards_in,
                         boards, aux, actions, B, seed, env_id0, ctr, reward, done, max_log2, max_tile, status);
    else
      hipLaunchKernelGGL(k_env_step4<0>, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, boards_in,
                         boards, aux, actions, B, seed, env_id0, ctr, reward, done, max_log2, max_tile, status);
    return launch_status();
  }
  Q2048_LAUNCH_ENV(k_env_step, flags & Q2048_FLAG_ENV_DQN, n, B, stream, boards_in, boards, aux, actions, B,
                   seed, env_id0, ctr, reward, done, max_log2, max_tile, status, draw_pos, draw_val,
                   draw_opos, draw_oval, draw_stride);
  return launch_status();
}

int q2048_env_step(uint8_t* boards, q2048_aux* aux, const uint8_t* actions, int64_t B, int n,
                   uint64_t seed, uint64_t env_id0, uint32_t ctr, float* reward, uint8_t* done,
                   uint8_t* max_log2, uint32_t* status, void* stream) {
  return env_step_impl(boards, boards, aux, actions, B, n, seed, env_id0, ctr, 0u, reward, done, max_log2,
                       nullptr, status, nullptr, nullptr, nullptr, nullptr, 1, stream);
}

int q2048_env_step_ex(uint8_t* boards, q2048_aux* aux, const uint8_t* actions, int64_t B, int n,
                      uint64_t seed, uint64_t env_id0, uint32_t ctr, uint32_t flags,
                      const uint32_t* draws4, float* reward, uint8_t* done, uint8_t* max_log2,
                      uint32_t* status, void* stream) {
  if (draws4 == nullptr)
    return env_step_impl(boards, boards, aux, actions, B, n, seed, env_id0, ctr, flags, reward, done, max_log2,
                         nullptr, status, nullptr, nullptr, nullptr, nullptr, 1, stream);
  return env_step_impl(boards, boards, aux, actions, B, n, 0, 0, 0, flags, reward, done, max_log2, nullptr, status,
                       draws4, draws4 + 1, draws4 + 2, draws4 + 3, 4, stream);
}

int q2048_env_step_to(const uint8_t* boards_in, uint8_t* boards_out, q2048_aux* aux, const uint8_t* actions,
                      int64_t B, int n, uint64_t seed, uint64_t env_id0, uint32_t ctr, uint32_t flags,
                      float* reward, uint8_t* done, uint8_t* max_log2, int32_t* max_tile, uint32_t* status,
                      void* stream) {
  return env_step_impl(boards_in, boards_out, aux, actions, B, n, seed, env_id0, ctr, flags, reward, done,
                       max_log2, max_tile, status, nullptr, nullptr, nullptr, nullptr, 1, stream);
}

int q2048_env_step_draws(uint8_t* boards, q2048_aux* aux, const uint8_t* actions,
                         const uint32_t* draw_pos, const uint32_t* draw_val, int64_t B, int n,
                         float* reward, uint8_t* done, uint8_t* max_log2, uint32_t* status,
                         void* stream) {
  if (!draw_pos || !draw_val) return Q2048_ERR_NULL;
  return env_step_impl(boards, boards, aux, actions, B, n, 0, 0, 0, 0u, reward, done, max_log2, nullptr, status,
                       draw_pos, draw_val, nullptr, nullptr, 1, stream);
}

static int q_choose_impl(const q2048_slot* table, int cap_log2, const uint8_t* boards, int64_t B,
                         int n, double eps, uint64_t seed, uint64_t env_id0, uint32_t ctr,
                         uint32_t flags, const void* row_cache, uint8_t* actions, uint32_t* status,
                         const uint32_t* draw_eps, const uint32_t* draw_act, void* stream) {
  if (int e = check_batch(B, n)) return e;
  if (int e = check_flags(flags)) return e;
  if (int e = check_table(table, cap_log2)) return e;
  if (!boards || !actions || !status) return Q2048_ERR_NULL;
  if (!aligned16(boards) || !aligned16(row_cache)) return Q2048_ERR_ALIGN;
  if (!(eps >= 0.0 && eps <= 1.0)) return Q2048_ERR_RANGE;
  if (B == 0) return Q2048_OK;
  const u64 mask = (1ull << cap_log2) - 1ull;
  if (n == 4)
    hipLaunchKernelGGL(k_q_choose<4>, dim3(grid_for(B)), dim3(kBlock), 0, (hipStream_t)stream, table, mask,
                       boards, B, eps, seed, env_id0, ctr, flags, static_cast<const RowCache<4>*>(row_cache),
                       actions, status, draw_eps, draw_act);
  else
    hipLaunchKernelGGL(k_q_choose<5>, dim3(grid_for(B)), dim3(kBlock), 0, (hipStream_t)stream, table, mask,
                       boards, B, eps, seed, env_id0, ctr, flags, static_cast<const RowCache<5>*>(row_cache),
                       actions, status, draw_eps, draw_act);
  return launch_status();
}

int q2048_q_choose(const q2048_slot* table, int cap_log2, const uint8_t* boards, int64_t B, int n,
                   double eps, uint64_t seed, uint64_t env_id0, uint32_t ctr, uint32_t flags,
                   uint8_t* actions, uint32_t* status, void* stream) {
  return q_choose_impl(table, cap_log2, boards, B, n, eps, seed, env_id0, ctr, flags, nullptr, actions,
                       status, nullptr, nullptr, stream);
}

int q2048_q_choose_cached(const q2048_slot* table, int cap_log2, const uint8_t* boards, int64_t B, int n,
                          double eps, uint64_t seed, uint64_t env_id0, uint32_t ctr, uint32_t flags,
                          const void* row_cache, uint8_t* actions, uint32_t* status, void* stream) {
  return q_choose_impl(table, cap_log2, boards, B, n, eps, seed, env_id0, ctr, flags, row_cache, actions,
                       status, nullptr, nullptr, stream);
}

int q2048_q_choose_draws(const q2048_slot* table, int cap_log2, const uint8_t* boards,
                         const uint32_t* draw_eps, const uint32_t* draw_act, int64_t B, int n,
                         double eps, uint64_t env_id0, uint32_t flags, uint8_t* actions,
                         uint32_t* status, void* stream) {
  if (!draw_eps || !draw_act) return Q2048_ERR_NULL;
  return q_choose_impl(table, cap_log2, boards, B, n, eps, 0, env_id0, 0, flags, nullptr, actions, status,
                       draw_eps, draw_act, stream);
}

size_t q2048_sizeof_rowcache(int n) { return n == 4 ? sizeof(RowCache<4>) : n == 5 ? sizeof(RowCache<5>) : 0; }

int q2048_rowcache_rebind(void* row_cache, int64_t B, int n, const q2048_slot* from_table, int from_cap_log2,
                          const q2048_slot* to_table, int to_cap_log2, void* stream) {
  if (int e = check_batch(B, n)) return e;
  if (int e = check_table(from_table, from_cap_log2)) return e;   // (an address of the past: never dereferenced)
  if (int e = check_table(to_table, to_cap_log2)) return e;
  if (row_cache == nullptr) return Q2048_ERR_NULL;
  if (!aligned16(row_cache)) return Q2048_ERR_ALIGN;
  if (B == 0) return Q2048_OK;
  const u64 tag_from = cache_tag(from_table, (1ull << from_cap_log2) - 1ull), tag_to = cache_tag(to_table, (1ull << to_cap_log2) - 1ull);
  if (n == 4)
    hipLaunchKernelGGL(k_rowcache_rebind<4>, dim3(grid_for(B)), dim3(kBlock), 0, (hipStream_t)stream,
                       static_cast<RowCache<4>*>(row_cache), B, tag_from, tag_to);
  else
    hipLaunchKernelGGL(k_rowcache_rebind<5>, dim3(grid_for(B)), dim3(kBlock), 0, (hipStream_t)stream,
                       static_cast<RowCache<5>*>(row_cache), B, tag_from, tag_to);
  return launch_status();
}

int q2048_q_update_cached(q2048_slot* table, int cap_log2, const uint8_t* boards_s, const uint8_t* actions,
                          const float* reward, const uint8_t* boards_s2, const uint8_t* done, int64_t B,
                          int n, double lr, double gamma, uint64_t env_id0, uint32_t flags,
                          void* row_cache, int64_t* stats_i, uint32_t* status, void* stream) {
  if (int e = check_batch(B, n)) return e;
  if (int e = check_flags(flags)) return e;
  if (int e = check_table(table, cap_log2)) return e;
  if (!boards_s || !actions || !reward || !boards_s2 || !done || !status) return Q2048_ERR_NULL;
  if (!aligned16(boards_s) || !aligned16(boards_s2) || !aligned16(row_cache)) return Q2048_ERR_ALIGN;
  if (!(lr == lr) || !(gamma == gamma)) return Q2048_ERR_RANGE;
  if (B == 0) return Q2048_OK;
  const u64 mask = (1ull << cap_log2) - 1ull;
  const dim3 grid((unsigned)((B + kUpdateBlock - 1) / kUpdateBlock)), block(kUpdateBlock);
  const hipStream_t s = (hipStream_t)stream;
#define Q2048_LAUNCH_UPDATE(NN, FR)                                                                                   \
  hipLaunchKernelGGL((k_q_update<NN, FR>), grid, block, 0, s, table, mask, boards_s, actions, reward, boards_s2, done, \
                     B, lr, gamma, env_id0, flags, static_cast<RowCache<NN>*>(row_cache), stats_i, status)
  if (flags & Q2048_FLAG_NO_NEW_ROWS) { if (n == 4) Q2048_LAUNCH_UPDATE(4, true); else Q2048_LAUNCH_UPDATE(5, true); }
  else { if (n == 4) Q2048_LAUNCH_UPDATE(4, false); else Q2048_LAUNCH_UPDATE(5, false); }
#undef Q2048_LAUNCH_UPDATE
  return launch_status();
}

int q2048_q_update(q2048_slot* table, int cap_log2, const uint8_t* boards_s, const uint8_t* actions,
                   const float* reward, const uint8_t* boards_s2, const uint8_t* done, int64_t B,
                   int n, double lr, double gamma, uint64_t env_id0, uint32_t flags,
                   int64_t* stats_i, uint32_t* status, void* stream) {
  return q2048_q_update_cached(table, cap_log2, boards_s, actions, reward, boards_s2, done, B, n, lr, gamma,
                               env_id0, flags, nullptr, stats_i, status, stream);
}

int q2048_q_lookup(const q2048_slot* table, int cap_log2, const uint8_t* boards, int64_t B, int n,
                   uint64_t env_id0, uint32_t flags, float* q_out, uint8_t* found, uint32_t* status,
                   void* stream) {
  if (int e = check_batch(B, n)) return e;
  if (int e = check_flags(flags)) return e;
  if (int e = check_table(table, cap_log2)) return e;
  if (!boards || !q_out || !status) return Q2048_ERR_NULL;
  if (!aligned16(boards) || !aligned16(q_out)) return Q2048_ERR_ALIGN;
  if (B == 0) return Q2048_OK;
  Q2048_LAUNCH(k_q_lookup, n, B, stream, table, (u64)((1ull << cap_log2) - 1ull), boards, B, env_id0,
               flags, q_out, found, status);
  return launch_status();
}

int q2048_fused_rollout(uint8_t* boards, q2048_aux* aux, q2048_slot* table, int cap_log2, int64_t B,
                        int n, int64_t steps, double eps, double lr, double gamma, uint64_t seed,
                        uint64_t env_id0, uint32_t ctr0, uint32_t flags, int64_t* stats_i,
                        double* stats_f, uint32_t* status, void* stream) {
  return q2048_fused_rollout_opts(boards, aux, table, cap_log2, B, n, steps, eps, lr, gamma, seed,
                                  env_id0, ctr0, flags, stats_i, stats_f, status, nullptr, stream);
}

int q2048_fused_rollout_log(uint8_t* boards, q2048_aux* aux, q2048_slot* table, int cap_log2,
                            int64_t B, int n, int64_t steps, double eps, double lr, double gamma,
                            uint64_t seed, uint64_t env_id0, uint32_t ctr0, uint32_t flags,
                            int64_t* stats_i, double* stats_f, uint32_t* status, q2048_episode* log,
                            int64_t log_capacity, uint64_t* log_count, void* stream) {
  q2048_rollout_opts o = {};
  o.size = (uint32_t)sizeof(o);
  o.log = log; o.log_capacity = log_capacity; o.log_count = log_count;
  return q2048_fused_rollout_opts(boards, aux, table, cap_log2, B, n, steps, eps, lr, gamma, seed,
                                  env_id0, ctr0, flags, stats_i, stats_f, status, &o, stream);
}

int q2048_fused_rollout_opts(uint8_t* boards, q2048_aux* aux, q2048_slot* table, int cap_log2,
                             int64_t B, int n, int64_t steps, double eps, double lr, double gamma,
                             uint64_t seed, uint64_t env_id0, uint32_t ctr0, uint32_t flags,
                             int64_t* stats_i, double* stats_f, uint32_t* status,
                             const q2048_rollout_opts* opts, void* stream) {
  q2048_rollout_opts o = {};
  if (opts != nullptr) {
    if (opts->size != sizeof(q2048_rollout_opts)) return Q2048_ERR_SIZE;   // a caller built against another header
    o = *opts;
  }
  if (int e = check_batch(B, n)) return e;
  if (int e = check_flags(flags)) return e;
  if (o.log != nullptr && (o.log_count == nullptr || o.log_capacity < 0)) return Q2048_ERR_NULL;
  if (o.log != nullptr && !aligned16(o.log)) return Q2048_ERR_ALIGN;
  if (o.row_cache != nullptr && !aligned16(o.row_cache)) return Q2048_ERR_ALIGN;
  // the mirror is a copy of both vectors, taken by the launch's last block: it needs both, and its ticket
  if (o.stats_mirror != nullptr && (o.mirror_ticket == nullptr || stats_i == nullptr || stats_f == nullptr))
    return Q2048_ERR_NULL;
  if (o.stats_mirror != nullptr && (reinterpret_cast<uintptr_t>(o.stats_mirror) & 7u)) return Q2048_ERR_ALIGN;
  if (int e = check_table(table, cap_log2)) return e;
  if (!boards || !aux || !status) return Q2048_ERR_NULL;
  if (!aligned16(boards) || !aligned16(aux)) return Q2048_ERR_ALIGN;
  if (steps < 0 || steps > (1 << 30)) return Q2048_ERR_SIZE;
  if (!(eps >= 0.0 && eps <= 1.0) || !(lr == lr) || !(gamma == gamma)) return Q2048_ERR_RANGE;
  if (B == 0 || steps == 0) return Q2048_OK;
  Q2048_LAUNCH_FUSED(flags, n, B, stream, boards, aux, table,
                     (u64)((1ull << cap_log2) - 1ull), B, (int)steps, eps, lr, gamma, seed, env_id0, ctr0,
                     flags, stats_i, stats_f, status, o.log, o.log_capacity, reinterpret_cast<u64*>(o.log_count),
                     o.row_cache, reinterpret_cast<u64*>(o.stats_mirror), o.mirror_ticket);
  return launch_status();
}

// workspace of q2048_det_rollout: double-buffered (group, target) pairs, the slot every env holds
// for its next step, the list of long groups, the sort's per-tile digit counts and digit totals;
// every part 256-byte aligned
struct DetLayout { size_t group[2], target[2], carry, cnt, total_cnt, stripes, gs[2], total; int64_t tiles; int gsz, n_groups; };
static int det_layout(int64_t B, int, DetLayout& L) {
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t at = 0;
  for (int k = 0; k < 2; ++k) { L.group[k] = at; at += up((size_t)B * 8); }
  for (int k = 0; k < 2; ++k) { L.target[k] = at; at += up((size_t)B * 8); }
  L.carry = at; at += up((size_t)B * 4);
  L.tiles = (B + kSortTile - 1) / kSortTile;
  L.cnt = at; at += up((size_t)L.tiles * kBlock * 4);
  L.total_cnt = at; at += up((size_t)kBlock * 4);
  L.stripes = at; at += up(sizeof(StatStripe) * kStatStripes);
  // group sums of the partition (see k_sort_count): groups of gsz = 2^ceil(log2(sqrt(T))) tiles; batches
  // beyond 8 Mi updates (T > 4096) keep the separate scan launch
  L.gsz = 1;
  while ((int64_t)L.gsz * L.gsz < L.tiles) L.gsz <<= 1;
  L.n_groups = L.tiles <= 4096 ? (int)((L.tiles + L.gsz - 1) / L.gsz) : 0;
  for (int k = 0; k < 2; ++k) { L.gs[k] = at; at += up((size_t)L.n_groups * kBlock * 4); }
  L.total = at;
  return Q2048_OK;
}

int64_t q2048_det_workspace_bytes(int64_t B, int cap_log2) {
  if (B < 0 || B > 0x7fffffffll || cap_log2 < 4 || cap_log2 > 40) return Q2048_ERR_SIZE;
  DetLayout L;
  if (int e = det_layout(B > 0 ? B : 1, cap_log2, L)) return e;
  return (int64_t)L.total;
}

int q2048_det_rollout(uint8_t* boards, q2048_aux* aux, q2048_slot* table, int cap_log2, int64_t B, int n,
                      int64_t steps, double eps, double lr, double gamma, uint64_t seed,
                      uint64_t env_id0, uint32_t ctr0, uint32_t flags, int64_t* stats_i,
                      double* stats_f, uint32_t* status, void* workspace, int64_t workspace_bytes,
                      void* stream) {
  return q2048_det_rollout_cached(boards, aux, table, cap_log2, B, n, steps, eps, lr, gamma, seed, env_id0, ctr0, flags,
                                  stats_i, stats_f, status, workspace, workspace_bytes, nullptr, stream);
}

int q2048_det_rollout_cached(uint8_t* boards, q2048_aux* aux, q2048_slot* table, int cap_log2, int64_t B, int n,
                             int64_t steps, double eps, double lr, double gamma, uint64_t seed,
                             uint64_t env_id0, uint32_t ctr0, uint32_t flags, int64_t* stats_i,
                             double* stats_f, uint32_t* status, void* workspace, int64_t workspace_bytes,
                             void* row_cache, void* stream) {
  if (int e = check_batch(B, n)) return e;
  if (row_cache != nullptr && !aligned16(row_cache)) return Q2048_ERR_ALIGN;
  // no learner-less or evaluation form of the deterministic step: refuse rather than learn anyway
  if (int e = check_flags(flags, Q2048_FLAG_NO_LEARN | Q2048_FLAG_PLAY_ONLY)) return e;
  if (B > 0x7fffffffll) return Q2048_ERR_SIZE;                       // one sort of at most 2^31 updates
  if (int e = check_table(table, cap_log2)) return e;
  static_assert(40 + 2 <= kDetHashShift, "slot << 2 | action fits below the hash field");
  if (!boards || !aux || !status || !workspace) return Q2048_ERR_NULL;
  if (!aligned16(boards) || !aligned16(aux) || (reinterpret_cast<uintptr_t>(workspace) & 255u)) return Q2048_ERR_ALIGN;
  if (steps < 0 || steps > (1 << 30)) return Q2048_ERR_SIZE;
  if (!(eps >= 0.0 && eps <= 1.0) || !(lr == lr) || !(gamma == gamma)) return Q2048_ERR_RANGE;
  if (B == 0 || steps == 0) return Q2048_OK;
  DetLayout L;
  if (int e = det_layout(B, cap_log2, L)) return e;
  if (workspace_bytes < (int64_t)L.total) return Q2048_ERR_SIZE;
  char* ws = static_cast<char*>(workspace);
  u64* group[2] = {reinterpret_cast<u64*>(ws + L.group[0]), reinterpret_cast<u64*>(ws + L.group[1])};
  double* target[2] = {reinterpret_cast<double*>(ws + L.target[0]), reinterpret_cast<double*>(ws + L.target[1])};
  uint32_t* carry = reinterpret_cast<uint32_t*>(ws + L.carry);
  uint32_t* cnt = reinterpret_cast<uint32_t*>(ws + L.cnt);
  uint32_t* total_cnt = reinterpret_cast<uint32_t*>(ws + L.total_cnt);
  uint32_t* gs[2] = {reinterpret_cast<uint32_t*>(ws + L.gs[0]), reinterpret_cast<uint32_t*>(ws + L.gs[1])};
  const hipStream_t s = (hipStream_t)stream;
  // statistics: striped copies in the workspace, folded into the caller's vectors once per call
  StatStripe* stripes = (stats_i != nullptr || stats_f != nullptr) ? reinterpret_cast<StatStripe*>(ws + L.stripes) : nullptr;
  if (stripes != nullptr && hipMemsetAsync(stripes, 0, sizeof(StatStripe) * kStatStripes, s) != hipSuccess)
    return Q2048_ERR_LAUNCH;
  const u64 mask = (1ull << cap_log2) - 1ull;
  // sort range: the 16 hash bits of the update word (comment at k_det_apply).  Experiment builds:
  // bits 8..13 of flags = k sorts by the low k hash bits only (crowded runs), 63 by the whole word
  int sort_lo = kDetHashShift, sort_hi = kDetHashShift + kDetHashBits;
  if (const int xb = (int)Q2048_XBITS(flags, 8, 63u)) {
    if (xb < kDetHashBits) sort_hi = sort_lo + xb;
    else if (xb == 63) sort_lo = 0;
  }
  const u64 run_mask = ((1ull << sort_hi) - 1ull) & ~((1ull << sort_lo) - 1ull);
  if (Q2048_XBITS(flags, 14, 1u)) L.n_groups = 0;   // experiment builds: the partition with its scan launch (the path of batches > 8 Mi)
  for (int64_t t = 0; t < steps; ++t) {
    if (row_cache != nullptr && (flags & Q2048_FLAG_NO_NEW_ROWS) != 0u) {
      Q2048_LAUNCH_ENV(k_det_phase1_visits, flags, n, B, s, boards, aux, table, mask, B, eps, gamma, seed, env_id0,
                       ctr0 + (uint32_t)t, flags, group[0], target[0], carry, (int)(t > 0), stripes, gs[0],
                       L.n_groups * kBlock, status, row_cache, lr);
    } else {
      Q2048_LAUNCH_ENV(k_det_phase1, flags, n, B, s, boards, aux, table, mask, B, eps, gamma, seed, env_id0,
                       ctr0 + (uint32_t)t, flags, group[0], target[0], carry, (int)(t > 0), stripes, gs[0],
                       L.n_groups * kBlock, status);
    }
    int cur = 0;                                     // which buffer holds the pairs
    for (int lo = sort_lo, pass = 0; lo < sort_hi; lo += 8, cur ^= 1, ++pass) {
      const uint32_t dmask = sort_hi - lo >= 8 ? 255u : (1u << (sort_hi - lo)) - 1u;
      uint32_t* gs_now = L.n_groups ? gs[pass & 1] : nullptr;
      hipLaunchKernelGGL(k_sort_count, dim3((unsigned)L.tiles), dim3(kBlock), 0, s, group[cur], B, lo, dmask,
                         cnt, L.tiles, gs_now, L.n_groups ? gs[(pass + 1) & 1] : nullptr, L.n_groups, L.gsz);
      if (gs_now == nullptr)
        hipLaunchKernelGGL(k_sort_scan, dim3(kBlock), dim3(kBlock), 0, s, cnt, L.tiles, total_cnt);
      hipLaunchKernelGGL(k_sort_scatter, dim3((unsigned)L.tiles), dim3(kBlock), 0, s, group[cur],
                         reinterpret_cast<const u64*>(target[cur]), group[cur ^ 1],
                         reinterpret_cast<u64*>(target[cur ^ 1]), B, lo, dmask, cnt, total_cnt, L.tiles,
                         gs_now, L.n_groups, L.gsz);
    }
    hipLaunchKernelGGL(k_det_apply, dim3(grid_for(B)), dim3(kBlock), 0, s, table, group[cur], target[cur], B, lr,
                       run_mask);
    if (int e = launch_status()) return e;
  }
  if (stripes != nullptr)
    hipLaunchKernelGGL(k_stats_fold, dim3(1), dim3(64), 0, s, stripes, stats_i, stats_f);
  return launch_status();
}

int q2048_legal_moves(const uint8_t* boards, int64_t B, int n, uint8_t* mask_out, void* stream) {
  if (int e = check_batch(B, n)) return e;
  if (!boards || !mask_out) return Q2048_ERR_NULL;
  if (!aligned16(boards)) return Q2048_ERR_ALIGN;
  if (B == 0) return Q2048_OK;
  Q2048_LAUNCH(k_legal_moves, n, B, stream, boards, B, mask_out);
  return launch_status();
}

int q2048_encode_onehot(const uint8_t* boards, int64_t B, int dtype, void* out, void* stream) {
  if (int e = check_batch(B, 4)) return e;
  if (B > (int64_t)0x7fffffff * (kBlock / 64)) return Q2048_ERR_SIZE;   // 64 threads per board
  if (!boards || !out) return Q2048_ERR_NULL;
  if (!aligned16(boards) || !aligned16(out)) return Q2048_ERR_ALIGN;
  if (dtype != 0 && dtype != 1) return Q2048_ERR_RANGE;
  if (B == 0) return Q2048_OK;
  const unsigned grid = (unsigned)((B * 64 + kBlock - 1) / kBlock);
  if (dtype == 0)
    hipLaunchKernelGGL(k_encode_onehot<false>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, boards, B, out);
  else
    hipLaunchKernelGGL(k_encode_onehot<true>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, boards, B, out);
  return launch_status();
}

int q2048_rt_choose(const float* weights, const uint8_t* boards, int64_t B, double eps, uint64_t seed,
                    uint64_t env_id0, uint32_t ctr, uint8_t* actions, void* stream) {
  if (int e = check_batch(B, 4)) return e;
  if (!weights || !boards || !actions) return Q2048_ERR_NULL;
  if (!aligned16(weights) || !aligned16(boards)) return Q2048_ERR_ALIGN;
  if (!(eps >= 0.0 && eps <= 1.0)) return Q2048_ERR_RANGE;
  if (B == 0) return Q2048_OK;
  hipLaunchKernelGGL(k_rt_choose, dim3(grid_for(B)), dim3(kBlock), 0, (hipStream_t)stream, weights,
                     boards, B, eps, seed, env_id0, ctr, actions);
  return launch_status();
}

int q2048_rt_lookup(const float* weights, const uint8_t* boards, int64_t B, float* q_out, void* stream) {
  if (int e = check_batch(B, 4)) return e;
  if (!weights || !boards || !q_out) return Q2048_ERR_NULL;
  if (!aligned16(weights) || !aligned16(boards) || !aligned16(q_out)) return Q2048_ERR_ALIGN;
  if (B == 0) return Q2048_OK;
  hipLaunchKernelGGL(k_rt_lookup, dim3(grid_for(B)), dim3(kBlock), 0, (hipStream_t)stream, weights,
                     boards, B, q_out);
  return launch_status();
}

int q2048_rt_update(float* weights, const uint8_t* boards_s, const uint8_t* actions, const float* reward,
                    const uint8_t* boards_s2, const uint8_t* done, int64_t B, double lr, double gamma,
                    uint32_t* status, void* stream) {
  if (int e = check_batch(B, 4)) return e;
  if (!weights || !boards_s || !actions || !reward || !boards_s2 || !done || !status) return Q2048_ERR_NULL;
  if (!aligned16(weights) || !aligned16(boards_s) || !aligned16(boards_s2)) return Q2048_ERR_ALIGN;
  if (!(lr == lr) || !(gamma == gamma)) return Q2048_ERR_RANGE;
  if (B == 0) return Q2048_OK;
  hipLaunchKernelGGL(k_rt_update, dim3(grid_for(B)), dim3(kBlock), 0, (hipStream_t)stream, weights,
                     boards_s, actions, reward, boards_s2, done, B, lr, gamma, status);
  return launch_status();
}

int q2048_rt_fused_rollout(uint8_t* boards, q2048_aux* aux, float* weights, int64_t B, int64_t steps,
                           double eps, double lr, double gamma, uint64_t seed, uint64_t env_id0,
                           uint32_t ctr0, int64_t* stats_i, double* stats_f, uint32_t* status,
                           void* stream) {
  if (int e = check_batch(B, 4)) return e;
  if (!boards || !aux || !weights || !status) return Q2048_ERR_NULL;
  if (!aligned16(boards) || !aligned16(aux) || !aligned16(weights)) return Q2048_ERR_ALIGN;
  if (steps < 0 || steps > (1 << 30)) return Q2048_ERR_SIZE;
  if (!(eps >= 0.0 && eps <= 1.0) || !(lr == lr) || !(gamma == gamma)) return Q2048_ERR_RANGE;
  if (B == 0 || steps == 0) return Q2048_OK;
  hipLaunchKernelGGL(k_rt_fused_rollout, dim3(grid_for(B)), dim3(kBlock), 0, (hipStream_t)stream,
                     boards, aux, weights, B, (int)steps, eps, lr, gamma, seed, env_id0, ctr0, stats_i,
                     stats_f, status);
  return launch_status();
}

int q2048_table_import(q2048_slot* table, int cap_log2, const uint64_t* keys, const float* q,
                       int64_t rows, int key_words, uint32_t* status, void* stream) {
  if (int e = check_table(table, cap_log2)) return e;
  if (!keys || !q || !status) return Q2048_ERR_NULL;
  if (rows < 0 || (key_words != 1 && key_words != 2)) return Q2048_ERR_SIZE;
  if (!aligned16(q)) return Q2048_ERR_ALIGN;
  if (rows == 0) return Q2048_OK;
  const u64 mask = (1ull << cap_log2) - 1ull;
  if (key_words == 1)
    hipLaunchKernelGGL(k_table_import<1>, dim3(grid_for(rows)), dim3(kBlock), 0, (hipStream_t)stream,
                       table, mask, reinterpret_cast<const u64*>(keys), q, rows, status);
  else
    hipLaunchKernelGGL(k_table_import<2>, dim3(grid_for(rows)), dim3(kBlock), 0, (hipStream_t)stream,
                       table, mask, reinterpret_cast<const u64*>(keys), q, rows, status);
  return launch_status();
}

int q2048_table_probe(q2048_slot* table, int cap_log2, int64_t lanes, int steps, uint64_t seed,
                      void* stream) {
  if (int e = check_table(table, cap_log2)) return e;
  if (lanes < 0 || lanes > ((int64_t)1 << 30) || steps < 0 || steps > (1 << 16)) return Q2048_ERR_SIZE;
  if (lanes == 0 || steps == 0) return Q2048_OK;
  hipLaunchKernelGGL(k_table_probe, dim3(grid_for(lanes)), dim3(kBlock), 0, (hipStream_t)stream, table,
                     (1ull << cap_log2) - 1ull, lanes, steps, seed, 0u);
  return launch_status();
}

// Table allocation from small physical chunks (HIP virtual-memory API).  The only entry points that
// allocate; everything else works on caller-owned memory, wherever it came from.
//
// A FAMILY is a table that may grow: q2048_table_reserve maps the first capacity, q2048_table_grow maps the
// next one onto fresh physical chunks IN AN ADDRESS RANGE OF ITS OWN, moves the rows over and releases the
// smaller table's chunks -- so every address is mapped at most ONCE in the life of the process and no range is
// ever freed after it was mapped.  That is the rule the round-3 trap taught (below); it also means a table's
// address changes when it grows.  (One reservation per table, starting exactly at the table: tables placed at
// offsets inside one large reservation mapped, but hipMemSetAccess refused them -- invalid value -- whenever
// their chunks were larger than 2 MiB and the reservation had not come back aligned to them.)
namespace {
// (scratch: 64 B of device memory + 64 B of pinned host memory per family for the counters of a growth, and a
// stream of its own -- allocated once, so that no growth ever calls hipMalloc / hipFree, which synchronise the device)
struct Family { size_t chunk; int dev, cap0_log2, max_log2; u64* dev_scratch; unsigned long long* host_scratch; hipStream_t stream; };
struct ChunkedTable { size_t bytes, chunk; int cap_log2; Family* fam; std::vector<hipMemGenericAllocationHandle_t> handles; bool retired = false; };
std::mutex g_tables_mutex;
std::map<void*, ChunkedTable> g_tables;
std::vector<Family*> g_families;                         // kept for the life of the process

// Chunk size of the table of capacity 2^cap_log2 in a family that can grow: its bytes / 1024, at least the family's
// own size (2 MiB) and AT MOST 32 MiB.  What the virtual-memory calls cost, measured call by call
// (tools/exp_vmm_cost.hip, profiles/r05_vmm_cost_by_chunk.txt, profiles/r05_vmm_cost_small_chunks.txt):
//   hipMemCreate   ~4 us per chunk up to 32 MiB -- and 1.7 ms per 64 MiB chunk, 7-29 ms per 256 MiB chunk, 46 ms per
//                  1 GiB chunk the first time: physically contiguous memory beyond 32 MiB is searched for, not found.
//                  That, not the number of chunks, was round 4's 1.9 s for the step to 128 GiB (2048 chunks of 64 MiB);
//   hipMemMap      5-13 us per chunk, hipMemUnmap 16 us, hipMemSetAccess 5 us -- growing slowly with the number of chunks
//                  a process holds (2 MiB chunks for 128 GiB = 65 536 of them: 13.8 s, profiles/r04_growth_phases_2MiB_chunks.txt).
// So a table is cut into as few chunks as 32 MiB allows: 1024 up to 32 GiB, 2048 for 64 GiB, 4096 for 128 GiB.
// The chunk size itself does not decide how fast a table takes scattered writes (six fresh 16 and 32 GiB tables each
// from 2, 8, 32 and 64 MiB chunks span 44-56 us per 2^20 load + CAS + store with every size's range inside the others',
// profiles/r04_requests/chunk_size_six_draws.txt).  (A family of one -- q2048_table_alloc, the fixed tables the bench
// runs on -- keeps the chunk size it asked for.)
constexpr size_t kMaxChunks = 1024, kBigChunk = (size_t)32 << 20;
size_t chunk_of(const Family& f, int cap_log2) {
  size_t c = f.chunk;
  if (f.max_log2 > f.cap0_log2)
    while (((sizeof(q2048_slot) << cap_log2) + c - 1) / c > kMaxChunks && c < kBigChunk) c <<= 1;
  return c;
}
// An address range of `bytes` that starts at a multiple of `align` and has never been mapped -- BY ANYBODY.
//
// Tables live in a PRIVATE REGION of the address space, handed out once each by a process-wide cursor (16 TiB upward;
// the runtime's own allocations -- hipMalloc, and the host's mmap -- grow down from the top of the 47-bit space).  Why:
// q2048_table_free keeps its ranges reserved for ever because a range that is mapped a second time serves stale
// translations on this stack (below) -- but the same happens with a range ANOTHER allocator has used: after a large
// hipFree the next un-hinted hipMemAddressReserve returns exactly the freed address (tools/va_hint_probe.hip,
// profiles/r06_va_hint_probe.jsonl: 64 and 200 GiB, both times): a table reserved the ordinary way right after a
// framework freed a large tensor sits on a range that has been mapped before.  An address hint in the private region
// is honoured and a chunk mapped there works (same probe), so no table ever shares a virtual address with anything
// that was mapped before it, whoever mapped it.
// When the hint is not honoured (another user of the region, an exotic address-space layout) the reservation falls back
// to what the runtime offers: the alignment argument is asked for and checked; when the runtime returns less (it honours
// the granularity only, on this stack) a range one `align` larger is reserved to find room, given back UNMAPPED (no
// translation of it ever existed) and reserved again at the aligned address inside it.
// (the measurement build is a second library that tests load into the same process: a region of its own)
#ifdef Q2048_EXPERIMENTS
constexpr uintptr_t kPrivateVaBase = (uintptr_t)0x380000000000ull, kPrivateVaEnd = (uintptr_t)0x600000000000ull;
#else
constexpr uintptr_t kPrivateVaBase = (uintptr_t)0x100000000000ull, kPrivateVaEnd = (uintptr_t)0x380000000000ull;
#endif
std::mutex g_va_mutex;
uintptr_t g_va_cursor = kPrivateVaBase;
void* reserve_aligned(size_t bytes, size_t align) {
  for (int attempt = 0; attempt < 4; ++attempt) {
    uintptr_t want;
    {
      std::lock_guard<std::mutex> lock(g_va_mutex);
      want = (g_va_cursor + align - 1) / align * align;
      if (want + bytes > kPrivateVaEnd) break;
      g_va_cursor = want + bytes + ((uintptr_t)1 << 30);            // (1 GiB of nothing between two tables)
    }
    void* va = nullptr;
    if (hipMemAddressReserve(&va, bytes, align, reinterpret_cast<void*>(want), 0) != hipSuccess) continue;
    if (reinterpret_cast<uintptr_t>(va) == want) return va;
    (void)hipMemAddressFree(va, bytes);                             // somewhere else: never mapped, given back
  }
  void* va = nullptr;
  if (hipMemAddressReserve(&va, bytes, align, nullptr, 0) != hipSuccess) return nullptr;
  if (reinterpret_cast<uintptr_t>(va) % align == 0) return va;
  if (hipMemAddressFree(va, bytes) != hipSuccess) return nullptr;
  void* probe = nullptr;
  if (hipMemAddressReserve(&probe, bytes + align, align, nullptr, 0) != hipSuccess) return nullptr;
  const uintptr_t want = (reinterpret_cast<uintptr_t>(probe) + align - 1) / align * align;
  if (hipMemAddressFree(probe, bytes + align) != hipSuccess) return nullptr;
  va = nullptr;
  if (hipMemAddressReserve(&va, bytes, align, reinterpret_cast<void*>(want), 0) != hipSuccess) return nullptr;
  if (reinterpret_cast<uintptr_t>(va) % align == 0) return va;
  (void)hipMemAddressFree(va, bytes);
  return nullptr;
}
// Freeing: every chunk is unmapped (one by one, as it was mapped) and its physical memory released;
// the ADDRESS RANGE is kept reserved for the life of the process and never handed out again.  On
// this stack (ROCm 7.2) a range that is freed, reserved again and mapped onto new physical chunks
// keeps serving some accesses through translations of its previous life: the third table of a
// process lost 1-15 % of its rows (inserts != occupied slots, 5x5 claims timing out;
// tools/archive/chunk_debug.py reproduces it in seconds, and with fresh addresses every table is exact).
// Virtual addresses are not scarce (a 32 GiB table uses 2^-12 of a 47-bit space).
int release_chunks(void* va, size_t chunk, std::vector<hipMemGenericAllocationHandle_t>& handles, size_t mapped) {
  int bad = 0;
  for (size_t k = 0; k < mapped; ++k)
    bad += hipMemUnmap(static_cast<char*>(va) + k * chunk, chunk) != hipSuccess;
  for (auto& h : handles) bad += hipMemRelease(h) != hipSuccess;
  handles.clear();
  return bad;
}
// RAII: the calling thread's current device while a table of another device is worked on
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) (void)hipSetDevice(dev); else prev = -1;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
// occupied slots of a table, counted on `stream` and waited for (scratch: 8 B of device memory, 8 B of pinned host
// memory -- a family's own, so that nothing here calls hipMalloc / hipFree, which synchronise the whole device)
int count_rows_on(const q2048_slot* table, int cap_log2, u64* dev8, unsigned long long* host8, hipStream_t stream,
                  uint64_t* rows) {
  if (hipMemsetAsync(dev8, 0, 8, stream) != hipSuccess) return Q2048_ERR_LAUNCH;
  if (int rc = q2048_table_count(table, cap_log2, reinterpret_cast<int64_t*>(dev8), stream)) return rc;
  if (hipMemcpyAsync(host8, dev8, 8, hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipStreamSynchronize(stream) != hipSuccess)
    return Q2048_ERR_LAUNCH;
  *rows = (uint64_t)*host8;
  return Q2048_OK;
}
// reserves, maps (from chunks of `chunk` bytes), zero-fills and verifies the table of capacity 2^cap_log2 of a
// family, all of it on the family's own stream (never the caller's: this runs next to the caller's launches when a
// growth is prepared); registers the table
int map_table_with(Family* f, int cap_log2, size_t chunk, q2048_slot** out) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = f->dev;
  ChunkedTable t;
  t.cap_log2 = cap_log2;
  t.fam = f;
  t.chunk = chunk;
  t.bytes = ((sizeof(q2048_slot) << cap_log2) + chunk - 1) / chunk * chunk;
  // Room is asked for BEFORE anything is created: a table is never mapped chunk by chunk into a device that cannot hold
  // it.  Walking the device into exhaustion is not a clean failure on this stack -- hipMemCreate did return an error
  // when 12 GiB were left for a 64 GiB table, and every chunk was given back, but the NEXT large mapping of the
  // process (after the memory had been freed again) ended in "Memory access fault by GPU ... Reason: Unknown" during
  // its zero fill, twice in two runs (profiles/r06_va_reuse_fault.txt).  256 MiB of headroom for the runtime's own needs.
  {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && t.bytes + ((size_t)256 << 20) > free_b) return Q2048_ERR_ALLOC;
  }
  char* va = static_cast<char*>(reserve_aligned(t.bytes, chunk));
  if (va == nullptr) return Q2048_ERR_ALLOC;
  const size_t n = t.bytes / t.chunk;
  t.handles.reserve(n);
  size_t mapped = 0;
#ifdef Q2048_EXPERIMENTS
#define Q2048_MAP_FAIL(what, e) fprintf(stderr, "[q2048 debug] map 2^%d slots, chunk %zu, chunk %zu of %zu: %s failed (%d)\n", cap_log2, t.chunk, k, n, what, (int)(e))
#else
#define Q2048_MAP_FAIL(what, e) do { } while (0)
#endif
  size_t k = 0;
  for (; k < n; ++k) {
    hipMemGenericAllocationHandle_t h;
    if (hipError_t e = hipMemCreate(&h, t.chunk, &prop, 0)) {
      Q2048_MAP_FAIL("hipMemCreate", e);
      release_chunks(va, t.chunk, t.handles, mapped);
      if (mapped == 0) (void)hipMemAddressFree(va, t.bytes);   // nothing was ever mapped here: the range can go back
      return Q2048_ERR_ALLOC;
    }
    t.handles.push_back(h);
    if (hipError_t e = hipMemMap(va + k * t.chunk, t.chunk, 0, h, 0)) {
      Q2048_MAP_FAIL("hipMemMap", e);
      release_chunks(va, t.chunk, t.handles, mapped);
      return Q2048_ERR_ALLOC;
    }
    ++mapped;
  }
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  if (hipError_t e = hipMemSetAccess(va, t.bytes, &acc, 1)) { Q2048_MAP_FAIL("hipMemSetAccess", e); release_chunks(va, t.chunk, t.handles, mapped); return Q2048_ERR_ALLOC; }
  if (hipError_t e = hipMemsetAsync(va, 0, t.bytes, f->stream)) { Q2048_MAP_FAIL("hipMemsetAsync", e); release_chunks(va, t.chunk, t.handles, mapped); return Q2048_ERR_ALLOC; }
  // Silent row loss is the worst failure this library can have, and a table that does not read back as
  // zeros is how it would start (a slot that looks occupied swallows a key's probe sequence; stale
  // translations of a re-used range do exactly that, above): one streaming count of the fresh table, ~5 ms per 32 GiB.
  uint64_t rows = 0;
  int rc = count_rows_on(reinterpret_cast<q2048_slot*>(va), cap_log2, f->dev_scratch + 4, f->host_scratch + 4, f->stream, &rows);
  if (rc == Q2048_OK && rows != 0) rc = Q2048_ERR_VERIFY;
  if (rc != Q2048_OK) { release_chunks(va, t.chunk, t.handles, mapped); return rc; }
  std::lock_guard<std::mutex> lock(g_tables_mutex);
  g_tables.emplace(va, std::move(t));
  *out = reinterpret_cast<q2048_slot*>(va);
  return Q2048_OK;
}
int release_retired(int dev, Family* only);
// the table of a family: from at most kMaxChunks chunks if the stack will have it, else from the family's own;
// when the device has no room, once more after the tables that growths have retired on it were given back
int map_table(Family* f, int cap_log2, q2048_slot** out) {
  const size_t chunk = chunk_of(*f, cap_log2);
  int rc = map_table_with(f, cap_log2, chunk, out);
  if (rc == Q2048_ERR_ALLOC && chunk != f->chunk) rc = map_table_with(f, cap_log2, f->chunk, out);
  if (rc == Q2048_ERR_ALLOC && release_retired(f->dev, nullptr) > 0) {
    rc = map_table_with(f, cap_log2, chunk, out);
    if (rc == Q2048_ERR_ALLOC && chunk != f->chunk) rc = map_table_with(f, cap_log2, f->chunk, out);
  }
  return rc;
}
// takes a table out of the registry and gives its chunks back (the address range stays reserved, above).
// `quiesced`: the caller knows that nothing on the device uses the table any more (a growth's event)
int unregister_and_release(q2048_slot* table, bool quiesced) {
  ChunkedTable t;
  {
    std::lock_guard<std::mutex> lock(g_tables_mutex);
    auto it = g_tables.find(table);
    if (it == g_tables.end()) return Q2048_ERR_NULL;      // not one of q2048_table_reserve's
    t = std::move(it->second);
    g_tables.erase(it);
  }
  DeviceGuard guard(t.fam->dev);                          // the table's device, whatever the caller's current one is
  if (!quiesced && hipDeviceSynchronize() != hipSuccess) return Q2048_ERR_LAUNCH;
#ifdef Q2048_EXPERIMENTS
  // Measurement builds only (tools/archive/chunk_debug.py): the round-3 free path that handed the address range
  // back, to look for what goes wrong when a range is re-used.  Q2048_DEBUG_VA_FREE = 1: unmap, release,
  // hipMemAddressFree; 2: the same and a device synchronize after it; 3: hipMemAddressFree only after
  // every chunk's hipMemRelease returned success, return codes printed.  Families of one table only.
  // (tools/va_reuse_repro.hip shows the same loss with no library at all.)
  if (const char* mode = getenv("Q2048_DEBUG_VA_FREE")) {
    if (t.fam->cap0_log2 == t.fam->max_log2) {
      const int m = atoi(mode);
      int bad_unmap = 0, bad_release = 0;
      for (size_t k = 0; k < t.handles.size(); ++k)
        bad_unmap += hipMemUnmap(reinterpret_cast<char*>(table) + k * t.chunk, t.chunk) != hipSuccess;
      for (auto& h : t.handles) bad_release += hipMemRelease(h) != hipSuccess;
      const hipError_t fr = hipMemAddressFree(table, t.bytes);
      const hipError_t sy = m >= 2 ? hipDeviceSynchronize() : hipSuccess;
      if (m >= 3)
        fprintf(stderr, "[q2048 debug] free %p: %zu chunks, unmap errors %d, release errors %d, AddressFree %d, sync %d\n",
                (void*)table, t.handles.size(), bad_unmap, bad_release, (int)fr, (int)sy);
      return (bad_unmap || bad_release || fr != hipSuccess) ? Q2048_ERR_ALLOC : Q2048_OK;
    }
  }
#endif
  return release_chunks(table, t.chunk, t.handles, t.handles.size()) ? Q2048_ERR_ALLOC : Q2048_OK;
}

// RETIRED tables.  A growth that has moved its rows does not give the old table's memory back: memory released
// by a process is wiped by the driver before it is handed out again, at ~40 GB/s, and an allocation made while that
// goes on waits for it -- hipMemCreate of a 128 GiB table takes 30 ms on memory that has been free for a while and
// 3-4 s right after 128 GiB were released (profiles/r05_vmm_wipe.txt), which is exactly what the NEXT growth of a
// run would meet.  So the old table stays mapped and allocated (nothing reads or writes it; a stale pointer into it
// hits dead memory, not a fault) until (a) a mapping on its device finds no room, (b) the last live table of its
// family is freed, or (c) the caller asks (q2048_table_trim).  With fourfold steps the retired tables of a family
// add up to a third of the live one.  Returns the number of tables released.
int release_retired(int dev, Family* only) {
  std::vector<q2048_slot*> gone;
  {
    std::lock_guard<std::mutex> lock(g_tables_mutex);
    for (auto& kv : g_tables)
      if (kv.second.retired && kv.second.fam->dev == dev && (only == nullptr || kv.second.fam == only))
        gone.push_back(static_cast<q2048_slot*>(kv.first));
  }
  int n = 0;
  for (q2048_slot* t : gone) n += unregister_and_release(t, n > 0) == Q2048_OK;   // one device synchronize is enough
  return n;
}

// ONE host thread per process for the virtual-memory calls that a growth takes off the caller's critical path
// (mapping the next table, un-mapping the previous one): tasks run one after the other, so the driver never sees
// two of these call sequences at once.  Started by the first q2048_table_grow_begin; joined by an atexit handler
// registered at that moment -- i.e. after the HIP runtime's own, so it runs before the runtime is torn down.
class Worker {
 public:
  void post(std::function<void()> fn) {
    std::unique_lock<std::mutex> lock(m_);
    if (!started_) {
      started_ = true;
      th_ = std::thread([this] { run(); });
      std::atexit([] { worker().stop(); });
    }
    q_.push_back(std::move(fn));
    cv_.notify_all();
  }
  void drain() {                                          // until every posted task has run
    std::unique_lock<std::mutex> lock(m_);
    cv_.wait(lock, [this] { return q_.empty() && !busy_; });
  }
  void stop() {
    {
      std::unique_lock<std::mutex> lock(m_);
      if (!started_ || stop_) return;
      stop_ = true;
      cv_.notify_all();
    }
    th_.join();
  }
  static Worker& worker() { static Worker* w = new Worker; return *w; }   // never destroyed

 private:
  void run() {
    std::unique_lock<std::mutex> lock(m_);
    for (;;) {
      cv_.wait(lock, [this] { return stop_ || !q_.empty(); });
      if (q_.empty()) return;                              // stop requested and nothing left to do
      std::function<void()> fn = std::move(q_.front());
      q_.pop_front();
      busy_ = true;
      lock.unlock();
      fn();
      lock.lock();
      busy_ = false;
      cv_.notify_all();
    }
  }
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<std::function<void()>> q_;
  std::thread th_;
  bool started_ = false, stop_ = false, busy_ = false;
};
}  // namespace

// A growth in progress (q2048_table_grow_begin .. _finish / _abort).  `state` is guarded by g_growth_mutex.
struct q2048_growth {
  q2048_slot* old_t = nullptr;
  q2048_slot* bigger = nullptr;
  int old_log2 = 0, new_log2 = 0;
  Family* fam = nullptr;
  int prepared = Q2048_PENDING;      // Q2048_PENDING while the worker maps the new table, then Q2048_OK or an error
  double prepare_ms = 0.0;           // what the worker spent on it (reserve + create + map + zero-fill + verify)
  bool committed = false, committing = false, verify_count = false;   // committing: a q2048_table_grow_commit is at work on it
  hipEvent_t moved = nullptr;        // recorded behind the rehash (and the counters' copy) on the caller's stream
};

namespace {
std::mutex g_growth_mutex;
std::condition_variable g_growth_cv;
std::vector<q2048_growth*> g_growths;                    // begun and neither finished nor aborted

q2048_growth* growth_of(const q2048_slot* table) {        // the growth a table takes part in (g_growth_mutex held)
  for (q2048_growth* g : g_growths)
    if (g->old_t == table || (g->bigger == table && g->committed)) return g;
  return nullptr;
}
bool growth_is_live(const q2048_growth* g) {              // (g_growth_mutex held)
  for (q2048_growth* x : g_growths) if (x == g) return true;
  return false;
}
void growth_forget(q2048_growth* g) {                     // (g_growth_mutex held)
  for (size_t k = 0; k < g_growths.size(); ++k)
    if (g_growths[k] == g) { g_growths.erase(g_growths.begin() + (long)k); break; }
  if (g->moved != nullptr) (void)hipEventDestroy(g->moved);
  delete g;
}
int wait_prepared(q2048_growth* g) {                      // blocks until the worker is done with the new table
  std::unique_lock<std::mutex> lock(g_growth_mutex);
  g_growth_cv.wait(lock, [g] { return g->prepared != Q2048_PENDING; });
  return g->prepared;
}
}  // namespace

int q2048_table_reserve(int cap_log2, int max_cap_log2, size_t chunk_bytes, q2048_slot** table_out) {
  if (table_out == nullptr) return Q2048_ERR_NULL;
  *table_out = nullptr;
  if (cap_log2 < 4 || cap_log2 > 40 || max_cap_log2 < cap_log2 || max_cap_log2 > 40) return Q2048_ERR_SIZE;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return Q2048_ERR_LAUNCH;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0)
    return Q2048_ERR_ALLOC;
  Family* f = new Family{chunk_bytes ? chunk_bytes : ((size_t)2 << 20), dev, cap_log2, max_cap_log2, nullptr, nullptr, nullptr};
  if (f->chunk % gran != 0 || (f->chunk & (f->chunk - 1)) != 0) { delete f; return Q2048_ERR_SIZE; }
  int rc = Q2048_OK;
  if (hipMalloc(&f->dev_scratch, 64) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&f->host_scratch), 64, hipHostMallocDefault) != hipSuccess ||
      hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking) != hipSuccess)
    rc = Q2048_ERR_ALLOC;
  if (rc == Q2048_OK) rc = map_table(f, cap_log2, table_out);
  if (rc != Q2048_OK) {
    if (f->stream != nullptr) (void)hipStreamDestroy(f->stream);
    if (f->host_scratch != nullptr) (void)hipHostFree(f->host_scratch);
    if (f->dev_scratch != nullptr) (void)hipFree(f->dev_scratch);
    delete f;
    return rc;
  }
  std::lock_guard<std::mutex> lock(g_tables_mutex);
  g_families.push_back(f);
  return rc;
}

int q2048_table_alloc(int cap_log2, size_t chunk_bytes, q2048_slot** table_out) {
  return q2048_table_reserve(cap_log2, cap_log2, chunk_bytes, table_out);
}

int q2048_table_grow_begin(q2048_slot* table, int cap_log2, int new_cap_log2, q2048_growth** growth_out) {
  if (table == nullptr || growth_out == nullptr) return Q2048_ERR_NULL;
  *growth_out = nullptr;
  Family* f = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_tables_mutex);
    auto it = g_tables.find(table);
    if (it == g_tables.end()) return Q2048_ERR_NULL;      // not one of q2048_table_reserve's
    if (it->second.cap_log2 != cap_log2) return Q2048_ERR_SIZE;
    f = it->second.fam;
  }
  if (new_cap_log2 <= cap_log2 || new_cap_log2 > f->max_log2) return Q2048_ERR_SIZE;
  q2048_growth* g = new q2048_growth;
  g->old_t = table;
  g->old_log2 = cap_log2;
  g->new_log2 = new_cap_log2;
  g->fam = f;
  {
    std::lock_guard<std::mutex> lock(g_growth_mutex);
    for (q2048_growth* x : g_growths)
      if (x->old_t == table) { delete g; return Q2048_ERR_BUSY; }   // one growth per table
    g_growths.push_back(g);
  }
  Worker::worker().post([g] {
    DeviceGuard guard(g->fam->dev);
    q2048_slot* bigger = nullptr;
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const int rc = map_table(g->fam, g->new_log2, &bigger);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    std::lock_guard<std::mutex> lock(g_growth_mutex);
    g->bigger = bigger;
    g->prepare_ms = (double)(t1.tv_sec - t0.tv_sec) * 1e3 + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-6;
    g->prepared = rc;
    g_growth_cv.notify_all();
  });
  *growth_out = g;
  return Q2048_OK;
}

int q2048_table_grow_poll(q2048_growth* g) {
  if (g == nullptr) return Q2048_ERR_NULL;
  std::lock_guard<std::mutex> lock(g_growth_mutex);
  if (!growth_is_live(g)) return Q2048_ERR_NULL;
  if (!g->committed) return g->prepared;
  const hipError_t e = hipEventQuery(g->moved);
  return e == hipSuccess ? Q2048_OK : (e == hipErrorNotReady ? Q2048_PENDING : Q2048_ERR_LAUNCH);
}

int q2048_table_grow_wait(q2048_growth* g, double* prepare_ms) {
  if (g == nullptr) return Q2048_ERR_NULL;
  {
    std::lock_guard<std::mutex> lock(g_growth_mutex);
    if (!growth_is_live(g)) return Q2048_ERR_NULL;
  }
  const int rc = wait_prepared(g);
  if (prepare_ms != nullptr) *prepare_ms = g->prepare_ms;
  return rc;
}

int q2048_table_grow_commit(q2048_growth* g, int key_words, uint32_t flags, q2048_slot** table_out, void* stream) {
  if (g == nullptr || table_out == nullptr) return Q2048_ERR_NULL;
  *table_out = nullptr;
  if (key_words != 1 && key_words != 2) return Q2048_ERR_SIZE;
  if (flags & ~(uint32_t)Q2048_GROW_VERIFY_COUNT) return Q2048_ERR_FLAGS;
  {
    // ONE critical section decides who commits: the growth must be live and untouched, no other growth of its family
    // may be committed-and-unfinished (the family's counters are in use until that one is finished) or being
    // committed right now, and the winner marks `committing` before the lock is released -- two host threads that
    // commit the same growth, or two growths of one family, cannot both get past this point.  The errors up to here
    // (NULL, SIZE, FLAGS, BUSY) leave `g` valid: retry later, or q2048_table_grow_abort.
    std::lock_guard<std::mutex> lock(g_growth_mutex);
    if (!growth_is_live(g) || g->committed) return Q2048_ERR_NULL;
    if (g->committing) return Q2048_ERR_BUSY;
    for (q2048_growth* x : g_growths)
      if (x != g && x->fam == g->fam && (x->committed || x->committing)) return Q2048_ERR_BUSY;
    g->committing = true;
  }
  int rc = wait_prepared(g);
  Family* f = g->fam;
  DeviceGuard guard(f->dev);
  hipStream_t s = (hipStream_t)stream;
  if (rc == Q2048_OK && hipEventCreateWithFlags(&g->moved, hipEventDisableTiming) != hipSuccess) rc = Q2048_ERR_LAUNCH;
  if (rc == Q2048_OK && hipMemsetAsync(f->dev_scratch, 0, 32, s) != hipSuccess) rc = Q2048_ERR_LAUNCH;
  if (rc == Q2048_OK) {
    // ordered on `stream` behind whatever still works on the old table; 8 blocks of 4 waves per CU
    const u64 cap = 1ull << g->old_log2, mask = (1ull << g->new_log2) - 1ull;
    const u64 want = (cap + kBlock - 1) / kBlock;
    const unsigned blocks = (unsigned)(want < 2048 ? want : 2048);
    if (key_words == 1)
      hipLaunchKernelGGL(k_table_rehash<1>, dim3(blocks), dim3(kBlock), 0, s, g->old_t, cap, g->bigger, mask, f->dev_scratch);
    else
      hipLaunchKernelGGL(k_table_rehash<2>, dim3(blocks), dim3(kBlock), 0, s, g->old_t, cap, g->bigger, mask, f->dev_scratch);
    rc = launch_status();
  }
  g->verify_count = (flags & Q2048_GROW_VERIFY_COUNT) != 0u;
  if (rc == Q2048_OK && g->verify_count)
    rc = q2048_table_count(g->bigger, g->new_log2, reinterpret_cast<int64_t*>(f->dev_scratch + 2), s);
  if (rc == Q2048_OK && (hipMemcpyAsync(f->host_scratch, f->dev_scratch, 32, hipMemcpyDeviceToHost, s) != hipSuccess ||
                         hipEventRecord(g->moved, s) != hipSuccess))
    rc = Q2048_ERR_LAUNCH;
  if (rc != Q2048_OK) {                                   // the old table is intact and stays the caller's
    if (g->bigger != nullptr) {
      (void)hipStreamSynchronize(s);
      (void)unregister_and_release(g->bigger, false);
    }
    std::lock_guard<std::mutex> lock(g_growth_mutex);
    growth_forget(g);
    return rc;
  }
  std::lock_guard<std::mutex> lock(g_growth_mutex);
  g->committed = true;
  g->committing = false;
  *table_out = g->bigger;
  return Q2048_OK;
}

int q2048_table_grow_finish(q2048_growth* g, int64_t* rows_moved) {
  if (g == nullptr) return Q2048_ERR_NULL;
  {
    std::lock_guard<std::mutex> lock(g_growth_mutex);
    if (!growth_is_live(g) || !g->committed) return Q2048_ERR_NULL;
  }
  Family* f = g->fam;
  int rc = Q2048_OK;
  {
    DeviceGuard guard(f->dev);
    // a wait that fails says nothing about the move: the growth stays registered and both tables stay live -- the
    // caller retries, or frees the NEW table (q2048_table_free finishes what it can and releases it) and carries on
    // with the old one at its own risk; forgetting `g` here would leave nobody knowing which table holds the rows
    if (hipEventSynchronize(g->moved) != hipSuccess) return Q2048_ERR_LAUNCH;
  }
  const unsigned long long moved = f->host_scratch[0], failed = f->host_scratch[1], counted = f->host_scratch[2];
  // every occupied slot of the old table must have found its place, and (when asked for) the new table must hold
  // exactly those rows
  if (rc == Q2048_OK && (failed != 0ull || (g->verify_count && counted != moved))) rc = Q2048_ERR_VERIFY;
  if (rows_moved != nullptr) *rows_moved = (int64_t)moved;
  q2048_slot* old_t = g->old_t;
  {
    std::lock_guard<std::mutex> lock(g_growth_mutex);
    growth_forget(g);
  }
  // nothing on the device reads the old table any more (the event): it is RETIRED -- kept until its memory is
  // needed (release_retired).  On Q2048_ERR_VERIFY both tables stay live.
  if (rc == Q2048_OK) {
    std::lock_guard<std::mutex> lock(g_tables_mutex);
    auto it = g_tables.find(old_t);
    if (it != g_tables.end()) it->second.retired = true;
  }
  return rc;
}

int q2048_table_grow_abort(q2048_growth* g) {
  if (g == nullptr) return Q2048_OK;
  {
    std::lock_guard<std::mutex> lock(g_growth_mutex);
    if (!growth_is_live(g) || g->committed) return Q2048_ERR_NULL;
  }
  const int rc = wait_prepared(g);
  q2048_slot* bigger = g->bigger;
  {
    std::lock_guard<std::mutex> lock(g_growth_mutex);
    growth_forget(g);
  }
  if (rc == Q2048_OK && bigger != nullptr) return unregister_and_release(bigger, true);   // never used by any kernel
  return Q2048_OK;
}

int q2048_table_grow(q2048_slot* table, int cap_log2, int new_cap_log2, int key_words, q2048_slot** table_out,
                     int64_t* rows_moved, void* stream) {
  if (table == nullptr || table_out == nullptr) return Q2048_ERR_NULL;
  *table_out = nullptr;
  if (key_words != 1 && key_words != 2) return Q2048_ERR_SIZE;
  q2048_growth* g = nullptr;
  if (int e = q2048_table_grow_begin(table, cap_log2, new_cap_log2, &g)) return e;
  q2048_slot* bigger = nullptr;
  if (int e = q2048_table_grow_commit(g, key_words, Q2048_GROW_VERIFY_COUNT, &bigger, stream)) {
    // BUSY (another growth of the family is committed and unfinished) leaves `g` valid and the prepared table mapped:
    // this host-synchronous form owns `g`, so it gives both back (any other error has ended the growth already)
    if (e == Q2048_ERR_BUSY) (void)q2048_table_grow_abort(g);
    return e;
  }
  if (int e = q2048_table_grow_finish(g, rows_moved)) {   // the old table is intact and stays the caller's
    (void)unregister_and_release(bigger, false);
    return e;
  }
  (void)unregister_and_release(table, false);             // host-synchronous form: the old table's memory is back
  *table_out = bigger;
  return Q2048_OK;
}

int q2048_table_trim(q2048_slot* table) {
  if (table == nullptr) return Q2048_ERR_NULL;
  Family* f = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_tables_mutex);
    auto it = g_tables.find(table);
    if (it == g_tables.end()) return Q2048_ERR_NULL;
    f = it->second.fam;
  }
  (void)release_retired(f->dev, f);
  return Q2048_OK;
}

int q2048_table_free(q2048_slot* table) {
  if (table == nullptr) return Q2048_OK;
  // a table that takes part in a growth: the growth is resolved first (an unprepared / uncommitted one is
  // aborted, a committed one finished), so that a caller tearing down in any order frees everything once
  for (;;) {
    q2048_growth* g = nullptr;
    bool committed = false;
    {
      std::lock_guard<std::mutex> lock(g_growth_mutex);
      g = growth_of(table);
      if (g != nullptr) committed = g->committed;
    }
    if (g == nullptr) break;
    if (!committed) (void)q2048_table_grow_abort(g);
    else if (q2048_table_grow_finish(g, nullptr) == Q2048_ERR_LAUNCH) {   // the device is gone: forget the growth, free what there is
      std::lock_guard<std::mutex> lock(g_growth_mutex);
      if (growth_is_live(g)) growth_forget(g);
    }
  }
  Worker::worker().drain();
  Family* f = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_tables_mutex);
    auto it = g_tables.find(table);
    if (it == g_tables.end()) return Q2048_ERR_NULL;      // not one of q2048_table_reserve's
    f = it->second.fam;
  }
  const int rc = unregister_and_release(table, false);
  // the family's last live table is gone: its retired predecessors go with it
  bool live = false;
  {
    std::lock_guard<std::mutex> lock(g_tables_mutex);
    for (auto& kv : g_tables) live = live || (kv.second.fam == f && !kv.second.retired);
  }
  if (!live) {
    (void)release_retired(f->dev, f);
    // nothing of the family is left (no table, and therefore no growth): its stream and scratch go too
    bool gone = true;
    {
      std::lock_guard<std::mutex> lock(g_tables_mutex);
      for (auto& kv : g_tables) gone = gone && kv.second.fam != f;
      if (gone)
        for (size_t k = 0; k < g_families.size(); ++k)
          if (g_families[k] == f) { g_families.erase(g_families.begin() + (long)k); break; }
    }
    if (gone) {
      DeviceGuard guard(f->dev);
      if (f->stream != nullptr) (void)hipStreamDestroy(f->stream);
      if (f->host_scratch != nullptr) (void)hipHostFree(f->host_scratch);
      if (f->dev_scratch != nullptr) (void)hipFree(f->dev_scratch);
      delete f;
    }
  }
  return rc;
}

int q2048_table_summarise(q2048_slot* table, int cap_log2, void* stream) {
  if (int e = check_table(table, cap_log2)) return e;
  const u64 lines = (1ull << cap_log2) >> 2;
  const u64 blocks = (lines + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(k_table_summarise, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(kBlock), 0,
                     (hipStream_t)stream, table, lines);
  return launch_status();
}

int q2048_table_count(const q2048_slot* table, int cap_log2, int64_t* count, void* stream) {
  return q2048_table_export(table, cap_log2, nullptr, nullptr, 0, 1, count, stream);
}

int q2048_table_export(const q2048_slot* table, int cap_log2, uint64_t* keys_out, float* q_out,
                       int64_t max_rows, int key_words, int64_t* count, void* stream) {
  if (int e = check_table(table, cap_log2)) return e;
  if (count == nullptr) return Q2048_ERR_NULL;
  if ((keys_out == nullptr) != (q_out == nullptr) || max_rows < 0) return Q2048_ERR_SIZE;
  if (key_words != 1 && key_words != 2) return Q2048_ERR_SIZE;
  if (q_out != nullptr && !aligned16(q_out)) return Q2048_ERR_ALIGN;
  const u64 cap = 1ull << cap_log2;
  const u64 blocks = (cap + kTileSlots - 1) / kTileSlots;
  // 8 blocks of 4 waves on each of the 256 CUs: the whole device streams
  hipLaunchKernelGGL(k_table_export, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(kBlock), 0,
                     (hipStream_t)stream, table, cap, reinterpret_cast<u64*>(keys_out), q_out,
                     max_rows, key_words, reinterpret_cast<u64*>(count));
  return launch_status();
}

}  // extern "C"
