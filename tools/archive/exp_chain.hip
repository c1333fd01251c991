// Measurement tool (not product code): does a one-step kernel whose lanes walk a dependent chain
//   coalesced 32-B load -> hash -> scattered 16-B load -> scattered compare-and-swap on that line
//   -> coalesced 20-B store
// (the shape of the deterministic mode's phase 1) get faster when a lane walks TWO chains at once
// (both loads in flight, one wait; both swaps in flight, one wait)?  2^20 chains either way.
//   hipcc -O3 --offload-arch=gfx950 -o tools/variants/exp_chain tools/archive/exp_chain.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)
using u64 = unsigned long long;
struct Slot { u64 key; float q[4]; u64 pad; };
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 32; x *= 0xd6e8feb86659fd93ull; x ^= x >> 32; x *= 0xd6e8feb86659fd93ull; x ^= x >> 32; return x; }
__device__ __forceinline__ uint32_t grind(uint32_t a, uint32_t b) {   // ~ one Philox call + a slide
#pragma unroll
  for (int r = 0; r < 30; ++r) { const u64 p = (u64)a * 0xD2511F53u; a = (uint32_t)(p >> 32) ^ b ^ (uint32_t)r; b = (uint32_t)p ^ a; }
  return a ^ b;
}

template <int K>
__global__ __launch_bounds__(256) void k_chain(Slot* table, u64 mask, const uint4* in0, const uint4* in1, int64_t lanes,
                                               uint32_t ctr, u64* out_g, double* out_t, uint32_t* out_c, int cas_pct) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= lanes) return;
  uint4 a[K], b[K];
#pragma unroll
  for (int k = 0; k < K; ++k) { a[k] = in0[t * K + k]; b[k] = in1[t * K + k]; }
  u64 h[K]; u32x4 v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const uint32_t g = grind(a[k].x ^ ctr, b[k].y + (uint32_t)(t * K + k));
    h[k] = mix(((u64)g << 32) | (uint32_t)(t * K + k) | ((u64)ctr << 20)) & mask;
  }
  if constexpr (K == 1) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v[0]) : "v"(&table[h[0]]) : "memory");
  } else {
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]) : "v"(&table[h[0]]), "v"(&table[h[1]]) : "memory");
  }
  u64 r[K];
  bool claim[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    claim[k] = (int)(h[k] % 100ull) < cas_pct;
    r[k] = 0;
  }
#pragma unroll
  for (int k = 0; k < K; ++k)                                 // issued back to back, consumed below
    if (claim[k]) r[k] = atomicCAS(&table[h[k]].key, 0ull, h[k] | 1ull);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    out_g[t * K + k] = (h[k] << 2) | (r[k] & 3ull) | v[k].x;
    out_t[t * K + k] = (double)v[k].z + (double)a[k].w;
    out_c[t * K + k] = (uint32_t)h[k];
  }
}

int main(int argc, char** argv) {
  const int cap_log2 = argc > 1 ? std::atoi(argv[1]) : 30;
  const int64_t n = 1 << 20;
  Slot* table; uint4 *in0, *in1; u64* og; double* ot; uint32_t* oc;
  CK(hipMalloc(&table, sizeof(Slot) << cap_log2)); CK(hipMemset(table, 0, sizeof(Slot) << cap_log2));
  CK(hipMalloc(&in0, n * 16)); CK(hipMalloc(&in1, n * 16)); CK(hipMalloc(&og, n * 8)); CK(hipMalloc(&ot, n * 8)); CK(hipMalloc(&oc, n * 4));
  CK(hipMemset(in0, 3, n * 16)); CK(hipMemset(in1, 5, n * 16));
  const u64 mask = (1ull << cap_log2) - 1;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int cas_pct : {0, 67, 100})
    for (int K : {1, 2}) {
      uint32_t ctr = 1000u * (uint32_t)cas_pct + 77u * (uint32_t)K;
      auto go = [&]() {
        ++ctr;
        if (K == 1) hipLaunchKernelGGL(k_chain<1>, dim3((unsigned)(n / 256)), dim3(256), 0, 0, table, mask, in0, in1, n, ctr, og, ot, oc, cas_pct);
        else hipLaunchKernelGGL(k_chain<2>, dim3((unsigned)(n / 512)), dim3(256), 0, 0, table, mask, in0, in1, n / 2, ctr, og, ot, oc, cas_pct);
      };
      for (int w = 0; w < 5; ++w) go();
      const int reps = 40;
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; ++r) go();
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      std::printf("{\"chains_per_lane\": %d, \"claims_pct\": %d, \"cap_log2\": %d, \"us_per_2e20_chains\": %.2f}\n", K, cas_pct, cap_log2, ms * 1e3 / reps);
      std::fflush(stdout);
    }
  return 0;
}
