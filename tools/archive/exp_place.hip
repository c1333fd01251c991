// Measurement tool (not product code): does the scattered-request rate depend on WHICH 32 GiB of
// the device a table occupies?  Times the same load+cas+store
// kernel (see exp_requests.hip) on buffers allocated and freed in the order the command line gives.
//   hipcc -O3 --offload-arch=gfx950 -o tools/variants/exp_place tools/archive/exp_place.hip
//   tools/variants/exp_place a0:32 m0 f0 a0:32 m0 ...
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

struct Slot { unsigned long long key; float q[4]; unsigned long long pad; };

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 32; x *= 0xd6e8feb86659fd93ull; x ^= x >> 32; x *= 0xd6e8feb86659fd93ull; x ^= x >> 32;
  return x;
}

__global__ __launch_bounds__(256) void k_requests(Slot* table, uint64_t mask, int64_t lanes, int steps,
                                                  int what, uint32_t ctr0, uint32_t* sink) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= lanes) return;
  uint32_t acc = (uint32_t)i;
  uint64_t prev = mix((uint64_t)i) & mask;
  for (int t = 0; t < steps; ++t) {
    const uint64_t key = mix(((uint64_t)i << 32) ^ (uint64_t)(ctr0 + (uint32_t)t)) | 1ull;
    const uint64_t at = (key >> 7) & mask;
    uint64_t seen = 0ull;
    if (what & 1) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      u32x4 v;
      asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(&table[at]) : "memory");
      seen = (uint64_t)v.x | ((uint64_t)v.y << 32);
      acc ^= v.z;
    }
    if ((what & 2) && seen == 0ull) acc ^= (uint32_t)atomicCAS(&table[at].key, 0ull, key);
    if (what & 4) *reinterpret_cast<uint32_t*>(&table[prev].q[key & 3ull]) = acc;
    if (what & 8) __hip_atomic_fetch_or(reinterpret_cast<uint32_t*>(&table[at].key), ctr0 >> 31, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // | 0: identity
    prev = at;
  }
  if (acc == 0x12345u) *sink = acc;
}

// Ops (argv, in order):  aK:G  allocate buffer K of G GiB;  mK  time the kernel on the first
// 32 GiB of buffer K (pK: scattered atomic-or-0 only, sK: scattered stores only);  fK  free buffer K.  Example: a0:32 m0 f0 a1:224 f1 a0:32 m0
int main(int argc, char** argv) {
  int cap_log2 = 30;                      // "-cN" as the first argument: measure 2^N slots
  const int steps = 64;
  int first = 1;
  if (argc > 1 && argv[1][0] == '-' && argv[1][1] == 'c') { cap_log2 = std::atoi(argv[1] + 2); first = 2; }
  if (cap_log2 < 20 || cap_log2 > 30) return 2;
  const uint64_t cap = 1ull << cap_log2;
  const int64_t lanes = 1 << 20;
  uint32_t* sink;
  CK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  Slot* buf[100] = {};
  for (int a = first; a < argc; ++a) {
    const char op = argv[a][0];
    const int k = std::atoi(argv[a] + 1);
    if (k < 0 || k > 99) return 2;
    if (op == 'a') {
      const char* colon = std::strchr(argv[a], ':');
      const unsigned long long gib = colon ? std::strtoull(colon + 1, nullptr, 10) : 0ull;
      if ((gib << 30) < cap * sizeof(Slot) || gib > 260 || buf[k]) return 2;
      CK(hipMalloc(&buf[k], gib << 30));
      std::printf("%-8s buffer %d = %llu GiB at %p\n", argv[a], k, gib, (void*)buf[k]);
    } else if (op == 'f') {
      if (!buf[k]) return 2;
      CK(hipFree(buf[k]));
      buf[k] = nullptr;
      std::printf("%-8s freed\n", argv[a]);
    } else if (op == 'm' || op == 'p' || op == 's' || op == 'w' || op == 'x') {
      // mK pK sK: load+cas+store / atomic-or / store on the first 2^cap_log2 slots of buffer K;
      // wK:OFF  stores on the 2^cap_log2-slot window that starts OFF GiB into buffer K;
      // xK:LOG2 stores spread over the first 2^LOG2 slots of buffer K.
      const int what = op == 'm' ? 7 : (op == 'p' ? 8 : 4);
      if (!buf[k]) return 2;
      const char* colon = std::strchr(argv[a], ':');
      const unsigned long long arg = colon ? std::strtoull(colon + 1, nullptr, 10) : 0ull;
      Slot* base = buf[k] + (op == 'w' ? (arg << 30) / sizeof(Slot) : 0ull);
      const uint64_t span = op == 'x' ? (1ull << arg) : cap;
      if (op != 'x') CK(hipMemsetAsync(base, 0, span * sizeof(Slot), 0));
      uint32_t ctr = 0;
      hipLaunchKernelGGL(k_requests, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, base, span - 1, lanes, steps, what, ctr, sink);
      ctr += steps;
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(k_requests, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, base, span - 1, lanes, steps, what, ctr, sink);
        ctr += steps;
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      std::printf("%-8s %.2f us/step\n", argv[a], (double)ms * 1e3 / (3 * steps));
    } else {
      return 2;
    }
    std::fflush(stdout);
  }
  return 0;
}
