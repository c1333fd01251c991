// q2048_core5.hpp -- per-lane arithmetic of the 5x5 board variant (BASELINE configs[4]).
//
// The reference hard-codes a 4x4 board (environment/Game2048_env.py:12,41); 5x5 is the same
// algorithm with n = 5 (the oracle restates every function for general n and is pinned to the
// reference at n = 4).  A board is five 32-bit row words, cell c of a row in the 6-bit field at
// bit 6c: 5 value bits (log2 tile <= 31) + 1 guard bit for the SWAR compares.  The functions
// mirror q2048_core.hpp one for one (same names, overloaded on the board type), so the env step
// and the kernels are written once over both geometries.
#pragma once
#include "q2048_core.hpp"

namespace q2048 {

struct Board5 { uint32_t r[5]; };

constexpr uint32_t k5One = 0x01041041u;   // bit 0 of each of the five 6-bit fields
constexpr uint32_t k5Low = k5One * 31u;   // 0x1f per field
constexpr uint32_t k5High = k5One << 5;   // guard bit per field
constexpr uint32_t k5All = k5One * 63u;

Q_HD uint32_t nz5(uint32_t x) { return (x + k5Low) & k5High; }    // guard bit set per field != 0
Q_HD uint32_t z5(uint32_t x) { return ~(x + k5Low) & k5High; }    // guard bit set per field == 0
// guard bit -> 0x1f: selects between fields whose guard bits are both 0 need no more (as fill7f)
Q_HD uint32_t fill5(uint32_t m) { return m - (m >> 5); }
Q_HD uint32_t field5(uint32_t w, int c) { return (w >> (6 * c)) & 63u; }

Q_HD bool operator==(const Board5& a, const Board5& b) {
  uint32_t d = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) d |= a.r[i] ^ b.r[i];
  return d == 0;
}
Q_HD void clear(Board5& b) {
#pragma unroll
  for (int i = 0; i < 5; ++i) b.r[i] = 0u;
}

// (a & m) | (b & ~m) for a mask m that is a compile-time constant: on the device as ONE v_bfi_b32
// (the compiler otherwise folds a constant mask into shift / and / or: three instructions a cell)
Q_HD uint32_t bsel_const(uint32_t m, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t d;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "s"(m), "v"(a), "v"(b));
  return d;
#else
  return bsel(m, a, b);
#endif
}

// the same for a per-lane mask of all ones or all zeros: the word-wise selects of move().  (Written
// as `cond ? p[4 - j] : p[j]` the compiler turns them into a dynamically indexed array, four
// v_cndmask a word.)
Q_HD uint32_t bsel_lane(uint32_t m, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t d;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(m), "v"(a), "v"(b));
  return d;
#else
  return bsel(m, a, b);
#endif
}

// out word j = column j (field i = row i): one shift and one bit-select per off-diagonal cell
Q_HD Board5 transpose(const Board5& b) {
  Board5 t;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    uint32_t w = b.r[j] & (63u << (6 * j));
#pragma unroll
    for (int i = 0; i < 5; ++i)
      if (i != j) w = bsel_const(63u << (6 * i), i < j ? b.r[i] >> (6 * (j - i)) : b.r[i] << (6 * (i - j)), w);
    t.r[j] = w;
  }
  return t;
}

// sum over the five fields f of 2^f (an empty field counts 1: the caller takes those off again)
Q_HD uint32_t pow2_sum5_raw(uint32_t v) {
  uint32_t s = 0;
#pragma unroll
  for (int c = 0; c < 5; ++c) s += 1u << field5(v, c);
  return s;
}

// move_left (Game2048_env.py:25-44) on five lines at once; c[j] = cell j of every line
Q_HD uint32_t slide_lines(uint32_t (&c)[5]) {
  // compress (:26): empty cells bubble to the far end, tiles keep their order (ten neighbour swaps)
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
    for (int k = 0; k < 4 - pass; ++k) {
      const uint32_t z = fill5(z5(c[k]));
      c[k] |= c[k + 1] & z;
      c[k + 1] &= ~z;
    }
  }
  // merge (:29-40): pairs left to right, a merged tile never merges again.  A line merges at most
  // twice, and a merge at t = 3 (cells 3 and 4 equal and not empty: nothing moved up before) is its
  // only one, so the merged tiles of t = 0 and t = 3 share one word.
  uint32_t merged[4], n = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const uint32_t e = ~((c[t] ^ c[t + 1]) + k5Low) & (c[t] + k5Low) & k5High;
    const uint32_t m = fill5(e);
    c[t] += e >> 5;
    merged[t] = c[t] & m;  // :36
    n += popc(e);
#pragma unroll
    for (int k = t + 1; k < 4; ++k) c[k] = bsel(m, c[k + 1], c[k]);
    c[4] &= ~m;
  }
  return pow2_sum5_raw(merged[0] | merged[3]) + pow2_sum5_raw(merged[1]) + pow2_sum5_raw(merged[2]) - (15u - n);
}

// Game2048.move without the spawn (:51-60); 0 left, 1 up, 2 right, 3 down (:54)
Q_HD bool move(Board5& b, int action, uint32_t& score) {
  const uint32_t horiz = (action & 1) == 0 ? ~0u : 0u, rev = (action & 2) != 0 ? ~0u : 0u;
  const Board5 t = transpose(b);
  uint32_t p[5], c[5], o[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) p[j] = bsel_lane(horiz, t.r[j], b.r[j]);
#pragma unroll
  for (int j = 0; j < 5; ++j) { c[j] = j == 2 ? p[2] : bsel_lane(rev, p[4 - j], p[j]); o[j] = c[j]; }
  score = slide_lines(c);
  uint32_t diff = 0;
#pragma unroll
  for (int j = 0; j < 5; ++j) diff |= c[j] ^ o[j];
  Board5 q;
#pragma unroll
  for (int j = 0; j < 5; ++j) q.r[j] = j == 2 ? c[2] : bsel_lane(rev, c[4 - j], c[j]);
  const Board5 qt = transpose(q);
#pragma unroll
  for (int j = 0; j < 5; ++j) b.r[j] = bsel_lane(horiz, qt.r[j], q.r[j]);
  return diff != 0;  // :38,42-43
}

// guard bits (5,11,17,23,29) of a row -> 5 contiguous bits
Q_HD uint32_t movemask5(uint32_t g) { return (((g >> 5) * 0x01084210u) >> 24) & 31u; }
// 25-bit mask of empty cells, bit = row-major cell index (np.where order, :17)
Q_HD uint32_t empty_mask(const Board5& b) {
  uint32_t m = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) m |= movemask5(z5(b.r[i])) << (5 * i);
  return m;
}

// index of the k-th (0-based) set bit of a 32-bit mask, k < popcount(mask)
Q_HD uint32_t kth_set_bit32(uint32_t mask, uint32_t k) {
  uint32_t pos = 0, c;
  bool ge;
  c = popc(mask & 0xffffu); ge = k >= c; k -= ge ? c : 0u; pos += ge ? 16u : 0u; mask >>= ge ? 16u : 0u;
  c = popc(mask & 0xffu);   ge = k >= c; k -= ge ? c : 0u; pos += ge ? 8u : 0u;  mask >>= ge ? 8u : 0u;
  c = popc(mask & 0xfu);    ge = k >= c; k -= ge ? c : 0u; pos += ge ? 4u : 0u;  mask >>= ge ? 4u : 0u;
  c = popc(mask & 0x3u);    ge = k >= c; k -= ge ? c : 0u; pos += ge ? 2u : 0u;  mask >>= ge ? 2u : 0u;
  c = mask & 1u;            ge = k >= c; pos += ge ? 1u : 0u;
  return pos;
}

// Game2048.add_number (:16-20) without a dense cell mask: the k-th empty cell in row-major order is
// the (k - empties of the rows above)-th empty field of its row, found on that row's guard bits
// (clear the lowest set bit that many times, isolate the next).  Returns the number of empty
// cells BEFORE the spawn; spawns only if `place` (and a cell is free, :18).
Q_HD uint32_t spawn_counted(Board5& b, bool place, uint32_t draw_pos, uint32_t draw_val) {
  uint32_t g[5], before[6];
  before[0] = 0u;
#pragma unroll
  for (int i = 0; i < 5; ++i) { g[i] = z5(b.r[i]); before[i + 1] = before[i] + popc(g[i]); }
  const uint32_t n = before[5];
  if (!place || n == 0u) return n;
  const uint32_t k = draw_index(draw_pos, n);
  uint32_t x = g[0], base = 0u, row = 0u;
#pragma unroll
  for (int i = 1; i < 5; ++i) {
    const bool ge = k >= before[i];
    x = ge ? g[i] : x; base = ge ? before[i] : base; row += ge ? 1u : 0u;
  }
  const uint32_t skip = k - base;
#pragma unroll
  for (uint32_t j = 0; j < 4u; ++j) x = skip > j ? x & (x - 1u) : x;
  x &= 0u - x;                                              // guard bit (6c + 5) of the chosen cell
  const uint32_t w = x >> (draw_is_four(draw_val) ? 4u : 5u);  // log2 tile 2 or 1 in its field
#pragma unroll
  for (int i = 0; i < 5; ++i) b.r[i] |= row == (uint32_t)i ? w : 0u;
  return n;
}
Q_HD void spawn(Board5& b, uint32_t draw_pos, uint32_t draw_val) { spawn_counted(b, true, draw_pos, draw_val); }

// Game2048.is_game_over (:65-75), closed form
Q_HD bool game_over(const Board5& b) {
  uint32_t live = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    live |= z5(b.r[i]);                                          // an empty cell
    live |= z5(b.r[i] ^ (b.r[i] >> 6)) & (k5High >> 6);          // equal horizontal neighbours
    if (i < 4) live |= z5(b.r[i] ^ b.r[i + 1]);                  // equal vertical neighbours
  }
  return live == 0u;
}

// On a board WITHOUT empty cells: does some cell equal a neighbour?  (A field of x + 0x1f keeps its
// guard bit clear iff the field of x is 0.  Field 4 of r ^ (r >> 6) is cell 4 itself: never 0 here.)
Q_HD bool equal_neighbours_on_full(const Board5& b) {
  uint32_t acc = k5High;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    acc &= (b.r[i] ^ (b.r[i] >> 6)) + k5Low;
    if (i < 4) acc &= (b.r[i] ^ b.r[i + 1]) + k5Low;
  }
  return (acc & k5High) != k5High;
}

// the spawn of a valid move (Game2048_env.py:61-62) and is_game_over (:99) after it, sharing the
// count of empty cells
Q_HD bool spawn_then_over(Board5& b, bool valid, uint32_t draw_pos, uint32_t draw_val) {
  const uint32_t n = spawn_counted(b, valid, draw_pos, draw_val);
  const uint32_t left = n - ((valid && n != 0u) ? 1u : 0u);
  return left == 0u && !equal_neighbours_on_full(b);
}

Q_HD uint32_t fieldmax5(uint32_t a, uint32_t b) {
  const uint32_t m = fill5(((a | k5High) - b) & k5High);
  return bsel(m, a, b);
}
Q_HD uint32_t max_log2(const Board5& b) {
  uint32_t x = fieldmax5(fieldmax5(b.r[0], b.r[1]), fieldmax5(fieldmax5(b.r[2], b.r[3]), b.r[4]));
  uint32_t m = field5(x, 0);
#pragma unroll
  for (int c = 1; c < 5; ++c) { const uint32_t f = field5(x, c); m = f > m ? f : m; }
  return m;
}

// state key (Agent/main.py:82): 25 cells x 5 bits = 125 bits in two words, each with bit 63 set
// so that neither is ever 0 (0 marks an empty / not yet published key word)
struct Key5 { uint64_t k0, k1; };
Q_HD uint32_t pack_row5(uint32_t r) {  // five 6-bit fields -> 25 contiguous bits
  return (r & 31u) | ((r >> 1) & (31u << 5)) | ((r >> 2) & (31u << 10)) | ((r >> 3) & (31u << 15)) |
         ((r >> 4) & (31u << 20));
}
Q_HD Key5 pack_key(const Board5& b) {
  const uint64_t p0 = pack_row5(b.r[0]), p1 = pack_row5(b.r[1]), p2 = pack_row5(b.r[2]),
                 p3 = pack_row5(b.r[3]), p4 = pack_row5(b.r[4]);
  return Key5{(p0 | (p1 << 25) | ((p2 & 0x1fffull) << 50)) | (1ull << 63),
              ((p2 >> 13) | (p3 << 12) | (p4 << 37)) | (1ull << 63)};
}
Q_HD uint32_t unpack_row5(uint32_t p) {
  return (p & 31u) | ((p & (31u << 5)) << 1) | ((p & (31u << 10)) << 2) | ((p & (31u << 15)) << 3) |
         ((p & (31u << 20)) << 4);
}
Q_HD Board5 unpack_key(const Key5& k) {
  const uint64_t a = k.k0 & ~(1ull << 63), c = k.k1 & ~(1ull << 63);
  Board5 b;
  b.r[0] = unpack_row5((uint32_t)(a & 0x1ffffffull));
  b.r[1] = unpack_row5((uint32_t)((a >> 25) & 0x1ffffffull));
  b.r[2] = unpack_row5((uint32_t)(((a >> 50) & 0x1fffull) | ((c & 0xfffull) << 13)));
  b.r[3] = unpack_row5((uint32_t)((c >> 12) & 0x1ffffffull));
  b.r[4] = unpack_row5((uint32_t)((c >> 37) & 0x1ffffffull));
  return b;
}

// memory image: uint8[25] row-major log2 tiles <-> fields
Q_HD Board5 board5_from_bytes(const uint8_t* p) {
  Board5 b;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    uint32_t w = 0;
#pragma unroll
    for (int c = 0; c < 5; ++c) w |= ((uint32_t)p[5 * i + c] & 31u) << (6 * c);
    b.r[i] = w;
  }
  return b;
}
Q_HD void board5_to_bytes(const Board5& b, uint8_t* p) {
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int c = 0; c < 5; ++c) p[5 * i + c] = (uint8_t)(field5(b.r[i], c) & 31u);
}

}  // namespace q2048
