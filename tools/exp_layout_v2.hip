// Measurement tool (not product code) -- round 6, the gate of "table line layout v2" (VERDICT r5 item 2).
//
// Question: the rollout's table work is bound by the NUMBER of scattered requests a lane issues.  Today a
// probe looks at ONE slot per 16-byte request ({key, q0, q1} of a 32-byte slot); on a table at load 0.35-0.6 an
// absent key's probe meets 1-3 occupied slots first, each one more dependent request.  A line laid out
// [keys][rows] shows several keys per request.  Does that pay, and does it cost anything on a young table?
//
// Three layouts of the same 128-byte line (4 slots), the same key set, the same probe ORDER where it can be:
//   A  today         slot j = 32 B at 32 j: {key, q[4], hi}.  Probe: 16-B load of {key, q0, q1} per slot, home slot
//                    first, then the line's other slots cyclically, then the next line; a hit reads {q2, q3} (8 B).
//   B  key pairs     {key[4] | q[4][4] | hi[4]}: 16-B load of the key PAIR the slot lies in; order within a line
//                    home, home^1 (same request), home^2, home^3 (second request); a hit reads its row (16 B).
//   C  key vector    same line as B; BOTH pair loads of a line issued together, one wait (the "one 32-B key-vector
//                    request" of the review: two 16-B instructions, there is no 32-B global load on gfx950).
// Workload per lane-step, as the fused rollout's (bench statistics: 0.87 probes per step, ~0.17 of them hits,
// the rest absent; every absent one is claimed while the table learns; one 4-byte Q write per step):
//   learn   probe (hit with probability `hit`/1024, else an absent key) -> compare-and-swap claim of the empty slot
//           the probe ended at -> 4-byte store into the row touched one step earlier
//   frozen  the same without the claim (Q2048_FLAG_NO_NEW_ROWS: the key set is closed)
// The table is pre-filled (untimed) to each load with keys mix(0..R) through the layout's own insert, so a "hit" key
// can be drawn by index; absent keys come from a disjoint index range (mix is a bijection).
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/variants/exp_layout_v2 tools/exp_layout_v2.hip
//   tools/variants/exp_layout_v2 [cap_log2=30] [lanes_log2=20] [steps=16] [hit per 1024 = 200] [work=1]
// Prints one JSON line per (load, mode): us per 2^20 lane-steps for A, B, C, requests per step counted by the kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      std::exit(1);                                                                  \
    }                                                                                \
  } while (0)

typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ __forceinline__ u64 mix(u64 x) {   // a bijection of 64-bit words
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
  return x;
}
__device__ __forceinline__ u64 key_of(u64 index) { const u64 k = mix(index); return k ? k : 1ull; }
__device__ __forceinline__ uint32_t grind(uint32_t a, uint32_t b, int work) {   // one unit ~ one Philox4x32-10 call
  for (int w = 0; w < work; ++w) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const uint64_t p = (uint64_t)a * 0xD2511F53u, q = (uint64_t)b * 0xCD9E8D57u;
      a = (uint32_t)(q >> 32) ^ b ^ (uint32_t)(0x9E3779B9u * (uint32_t)r);
      b = (uint32_t)(p >> 32) ^ a ^ (uint32_t)q ^ (uint32_t)p;
    }
  }
  return a ^ b;
}
__device__ __forceinline__ u32x4 ld16(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void ld16x2(const void* p, const void* q, u32x4& a, u32x4& b) {
  asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(a), "=&v"(b) : "v"(p), "v"(q) : "memory");
}
__device__ __forceinline__ u64 ld8(const void* p) {
  return __hip_atomic_load(const_cast<u64*>(reinterpret_cast<const u64*>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int kMaxLines = 64;   // a probe gives up after this many lines (never reached below load 0.95)
// addresses of slot (line, j)
template <int L> __device__ __forceinline__ char* key_addr(char* t, u64 line, uint32_t j) {
  return L == 0 ? t + line * 128 + 32 * j : t + line * 128 + 8 * j;
}
template <int L> __device__ __forceinline__ char* row_addr(char* t, u64 line, uint32_t j) {
  return L == 0 ? t + line * 128 + 32 * j + 8 : t + line * 128 + 32 + 16 * j;
}

struct Found { char* row; char* empty_key; uint32_t requests; bool hit; };

// L = 0 (A), 1 (B), 2 (C)
template <int L>
__device__ __forceinline__ Found probe(char* t, u64 lmask, u64 key, uint32_t& acc) {
  const u64 h = mix(key ^ 0x9E3779B97F4A7C15ull);
  u64 line = (h >> 2) & lmask;
  const uint32_t j0 = (uint32_t)h & 3u;
  Found f{nullptr, nullptr, 0u, false};
  for (int n = 0; n < kMaxLines; ++n, line = (line + 1) & lmask) {
    if constexpr (L == 0) {
#pragma unroll 1
      for (uint32_t p = 0; p < 4; ++p) {
        const uint32_t j = (j0 + p) & 3u;
        const u32x4 v = ld16(key_addr<0>(t, line, j));
        ++f.requests;
        const u64 k = (u64)v.x | ((u64)v.y << 32);
        if (k == key) {
          const u64 hi = ld8(row_addr<0>(t, line, j) + 8);
          ++f.requests;
          acc ^= v.z ^ (uint32_t)hi;
          f.row = row_addr<0>(t, line, j); f.hit = true;
          return f;
        }
        if (k == 0ull) { f.empty_key = key_addr<0>(t, line, j); return f; }
      }
    } else {
      u32x4 pr[2];
      const uint32_t first = j0 >> 1;                 // the pair the home slot lies in
      if constexpr (L == 2) {
        ld16x2(t + line * 128 + 16 * first, t + line * 128 + 16 * (first ^ 1u), pr[0], pr[1]);
        f.requests += 2;
      }
#pragma unroll 1
      for (uint32_t half = 0; half < 2; ++half) {
        if constexpr (L == 1) { pr[half] = ld16(t + line * 128 + 16 * (first ^ half)); ++f.requests; }
        const u32x4 v = pr[half];
        const u64 k_lo = (u64)v.x | ((u64)v.y << 32), k_hi = (u64)v.z | ((u64)v.w << 32);
#pragma unroll
        for (uint32_t q = 0; q < 2; ++q) {
          const uint32_t j = j0 ^ (half << 1) ^ q;    // home, home^1 | home^2, home^3
          const u64 k = (j & 1u) ? k_hi : k_lo;
          if (k == key) {
            const u32x4 r = ld16(row_addr<1>(t, line, j));
            ++f.requests;
            acc ^= r.x ^ r.w;
            f.row = row_addr<1>(t, line, j); f.hit = true;
            return f;
          }
          if (k == 0ull) { f.empty_key = key_addr<1>(t, line, j); return f; }
        }
      }
    }
  }
  return f;
}

template <int L>
__global__ __launch_bounds__(256) void k_fill(char* t, u64 lmask, u64 first, u64 count, u64* failed) {
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const u64 key = key_of(first + i);
  uint32_t acc = 0;
  for (int attempt = 0; attempt < 64; ++attempt) {
    const Found f = probe<L>(t, lmask, key, acc);
    if (f.hit) return;
    if (f.empty_key == nullptr) break;
    if (atomicCAS(reinterpret_cast<u64*>(f.empty_key), 0ull, key) == 0ull) return;
  }
  atomicAdd(failed, 1ull);
}

// COUNT: the untimed pass that adds up requests / hits / claims (16 Ki waves x 3 same-address atomics cost a launch
// 0.5 ms -- 30 us per step at 16 steps -- so the timed launches carry none)
template <int L, bool COUNT>
__global__ __launch_bounds__(256, 6) void k_step(char* t, u64 lmask, u64 rows, int64_t lanes, int steps, int hit1024,
                                                  int learn, int work, uint32_t ctr0, u64* counters, uint32_t* sink) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= lanes) return;
  uint32_t acc = (uint32_t)i, requests = 0, hits = 0, claims = 0;
  char* prev = nullptr;
  for (int s = 0; s < steps; ++s) {
    acc = grind(acc, (uint32_t)s + ctr0, work);
    const u64 r = mix(((u64)i << 32) ^ (u64)(ctr0 + (uint32_t)s));
    const bool want_hit = rows > 0 && (int)(r & 1023ull) < hit1024;
    const u64 key = want_hit ? key_of((r >> 10) % rows) : key_of((1ull << 40) | (r >> 24));
    Found f = probe<L>(t, lmask, key, acc);
    requests += f.requests;
    hits += f.hit;
    char* row = f.row;
    if (!f.hit && learn && f.empty_key != nullptr) {
      const u64 old = atomicCAS(reinterpret_cast<u64*>(f.empty_key), 0ull, key);
      acc ^= (uint32_t)old;
      ++claims;
      if (old == 0ull)                                 // the claimed slot's row (address arithmetic only)
        row = L == 0 ? f.empty_key + 8 : t + ((u64)(f.empty_key - t) & ~127ull) + 32 + 2 * ((u64)(f.empty_key - t) & 127ull);
    }
    if (prev != nullptr) *reinterpret_cast<uint32_t*>(prev + 4 * (r >> 62)) = acc;   // the TD write of the step before
    prev = row;
  }
  sink[i] = acc ^ (COUNT ? 0u : requests ^ hits ^ claims);
  if constexpr (COUNT) {
    atomicAdd(&counters[0], (u64)requests);
    atomicAdd(&counters[1], (u64)hits);
    atomicAdd(&counters[2], (u64)claims);
  }
}

template <int L>
static double time_steps(char* t, u64 lmask, u64 rows, int64_t lanes, int steps, int hit, int learn, int work,
                         uint32_t& ctr, u64* counters, uint32_t* sink, double* req_per_step, double* hits_per_step,
                         double* claims_per_step) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ms;
  u64 host[3];
  const dim3 grid((unsigned)((lanes + 255) / 256));
  CK(hipMemset(counters, 0, 24));
  hipLaunchKernelGGL((k_step<L, true>), grid, dim3(256), 0, 0, t, lmask, rows, lanes, steps, hit, learn, work, ctr, counters, sink);
  ctr += (uint32_t)steps;
  CK(hipMemcpy(host, counters, 24, hipMemcpyDeviceToHost));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_step<L, false>), grid, dim3(256), 0, 0, t, lmask, rows, lanes, steps, hit, learn, work, ctr, counters, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    ctr += (uint32_t)steps;
    float m;
    CK(hipEventElapsedTime(&m, e0, e1));
    ms.push_back(m);
  }
  std::sort(ms.begin(), ms.end());
  const double n = (double)lanes * steps;
  *req_per_step = host[0] / n; *hits_per_step = host[1] / n; *claims_per_step = host[2] / n;
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return ms[1] * 1e3 / steps * (1048576.0 / (double)lanes);
}

int main(int argc, char** argv) {
  const int cap_log2 = argc > 1 ? atoi(argv[1]) : 30, lanes_log2 = argc > 2 ? atoi(argv[2]) : 20;
  const int steps = argc > 3 ? atoi(argv[3]) : 16, hit = argc > 4 ? atoi(argv[4]) : 200, work = argc > 5 ? atoi(argv[5]) : 1;
  const u64 slots = 1ull << cap_log2, lmask = (slots >> 2) - 1ull;
  const int64_t lanes = 1ll << lanes_log2;
  const size_t bytes = (size_t)slots * 32;
  char* tab[3];
  for (int l = 0; l < 3; ++l) { CK(hipMalloc(&tab[l], bytes)); CK(hipMemset(tab[l], 0, bytes)); }
  u64* counters; uint32_t* sink;
  CK(hipMalloc(&counters, 64)); CK(hipMalloc(&sink, (size_t)lanes * 4));
  const double loads[] = {0.0, 0.05, 0.10, 0.20, 0.35, 0.45, 0.60, 0.70, 0.80};
  u64 filled = 0, extra = 0;                              // rows pre-filled (keys mix(0..filled)); rows the learn passes claimed
  uint32_t ctr = 1;
  for (double load : loads) {
    const u64 want = (u64)(load * (double)slots);
    if (want > filled + extra) {                          // the same keys into the three layouts
      const u64 count = want - filled - extra;
      CK(hipMemset(counters, 0, 64));
      const unsigned blocks = (unsigned)((count + 255) / 256);
      hipLaunchKernelGGL(k_fill<0>, dim3(blocks), dim3(256), 0, 0, tab[0], lmask, filled, count, counters + 4);
      hipLaunchKernelGGL(k_fill<1>, dim3(blocks), dim3(256), 0, 0, tab[1], lmask, filled, count, counters + 4);
      hipLaunchKernelGGL(k_fill<2>, dim3(blocks), dim3(256), 0, 0, tab[2], lmask, filled, count, counters + 4);
      CK(hipDeviceSynchronize());
      u64 failed = 0;
      CK(hipMemcpy(&failed, counters + 4, 8, hipMemcpyDeviceToHost));
      if (failed) std::fprintf(stderr, "fill to load %.2f: %llu rows found no slot\n", load, failed);
      filled += count;
    }
    for (int learn = 0; learn <= 1; ++learn) {            // frozen first: it leaves the table as it is
      if (learn && load > 0.71) continue;                 // (a learning table never gets there)
      double us[3], req[3], hits[3], claims[3];
      const double load_before = (double)(filled + extra) / (double)slots;
      // a learn pass adds ~0.8 rows per lane-step to every layout alike (4 x `steps` x 2^20 of them: 0.05 of a
      // 2^30-slot table at 16 steps); the next load point fills up to its target from there
      us[0] = time_steps<0>(tab[0], lmask, filled, lanes, steps, hit, learn, work, ctr, counters, sink, &req[0], &hits[0], &claims[0]);
      us[1] = time_steps<1>(tab[1], lmask, filled, lanes, steps, hit, learn, work, ctr, counters, sink, &req[1], &hits[1], &claims[1]);
      us[2] = time_steps<2>(tab[2], lmask, filled, lanes, steps, hit, learn, work, ctr, counters, sink, &req[2], &hits[2], &claims[2]);
      if (learn) extra += (u64)(claims[0] * (double)lanes * steps * 4);
      std::printf("{\"cap_log2\": %d, \"load\": %.3f, \"mode\": \"%s\", \"hit_per_1024\": %d, \"work\": %d, \"steps\": %d, "
                  "\"us_per_step\": {\"A_slots\": %.2f, \"B_key_pairs\": %.2f, \"C_key_vector\": %.2f}, "
                  "\"probe_requests_per_step\": {\"A\": %.3f, \"B\": %.3f, \"C\": %.3f}, \"hits_per_step\": %.3f, "
                  "\"claims_per_step\": %.3f, \"B_vs_A\": %.3f, \"C_vs_A\": %.3f}\n",
                  cap_log2, load_before, learn ? "learn" : "frozen", hit, work, steps, us[0], us[1], us[2], req[0], req[1], req[2],
                  hits[0], claims[0], us[1] / us[0], us[2] / us[0]);
      std::fflush(stdout);
    }
  }
  return 0;
}
