// Measurement tool (not product code): does the scattered-request rate of a fresh MI355X change
// with how long the device has been busy?  Launches the same scattered-store / load+cas+store
// kernel back to back for `seconds` and prints the time per 1 Mi-lane step against elapsed time.
//   hipcc -O3 --offload-arch=gfx950 -o tools/variants/exp_ramp tools/archive/exp_ramp.hip
//   tools/variants/exp_ramp [seconds=20] [what=4 (store) | 7 (load+cas+store)] [idle_ms_between=0]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>

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
    prev = at;
  }
  if (acc == 0x12345u) *sink = acc;
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? std::atof(argv[1]) : 20.0;
  const int what = argc > 2 ? std::atoi(argv[2]) : 4;
  const int idle_ms = argc > 3 ? std::atoi(argv[3]) : 0;
  const int cap_log2 = 30, steps = 64;
  const int64_t lanes = 1 << 20;
  const uint64_t cap = 1ull << cap_log2;
  Slot* table; uint32_t* sink;
  CK(hipMalloc(&table, cap * sizeof(Slot)));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(table, 0, cap * sizeof(Slot)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const auto t0 = std::chrono::steady_clock::now();
  uint32_t ctr = 0;
  double next_print = 0.0;
  int launches = 0;
  std::printf("# what=%d idle_ms=%d; columns: elapsed_s us_per_step\n", what, idle_ms);
  for (;;) {
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (el > seconds) break;
    if ((what & 2) && launches % 8 == 0) CK(hipMemsetAsync(table, 0, cap * sizeof(Slot), 0));  // keep the fill low
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_requests, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, table, cap - 1, lanes, steps, what, ctr, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    ctr += steps; ++launches;
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (el >= next_print) {
      std::printf("%.2f %.2f\n", el, (double)ms * 1e3 / steps);
      std::fflush(stdout);
      next_print = el + 0.25;
    }
    if (idle_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(idle_ms));
  }
  return 0;
}
