// Measurement tool (not product code): is a table SPREAD over the device's physical memory faster
// for scattered writes than one physically contiguous table?  Creates `n` physical chunks with
// the HIP virtual-memory API, then maps 32 GiB worth of them into one virtual range three ways
// (the first chunks, the last chunks, every (n/32)-th chunk) and times the scattered requests of
// exp_requests.hip on each mapping.
//   hipcc -O3 --offload-arch=gfx950 -o tools/variants/exp_vmm tools/archive/exp_vmm.hip
//   tools/variants/exp_vmm [chunk_mib=1024] [total_gib=256] [aligned_va=0] [reserve_first=0] [table_gib=32]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
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
    if (what & 8) __hip_atomic_fetch_or(reinterpret_cast<uint32_t*>(&table[at].key), ctr0 >> 31, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    prev = at;
  }
  if (acc == 0x12345u) *sink = acc;
}

static double time_kernel(Slot* table, uint64_t cap, int what, uint32_t* sink) {
  const int64_t lanes = 1 << 20;
  const int steps = 64;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipMemsetAsync(table, 0, cap * sizeof(Slot), 0));
  uint32_t ctr = 0;
  hipLaunchKernelGGL(k_requests, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, table, cap - 1, lanes, steps, what, ctr, sink);
  ctr += steps;
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < 3; ++r) {
    hipLaunchKernelGGL(k_requests, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, table, cap - 1, lanes, steps, what, ctr, sink);
    ctr += steps;
  }
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return (double)ms * 1e3 / (3 * steps);
}

int main(int argc, char** argv) {
  const size_t chunk = (size_t)(argc > 1 ? std::atoi(argv[1]) : 1024) << 20;
  const size_t total = (size_t)(argc > 2 ? std::atoi(argv[2]) : 256) << 30;
  const bool aligned_va = argc > 3 && std::atoi(argv[3]) != 0;     // reserve the range chunk-aligned
  const bool reserve_first = argc > 4 && std::atoi(argv[4]) != 0;  // reserve before creating chunks
  const size_t table_bytes = (size_t)(argc > 5 ? std::atoi(argv[5]) : 32) << 30;   // table size in GiB (a power of two)
  const uint64_t cap = table_bytes / sizeof(Slot);
  CK(hipSetDevice(0));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran_min = 0, gran_rec = 0;
  CK(hipMemGetAllocationGranularity(&gran_min, &prop, hipMemAllocationGranularityMinimum));
  CK(hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended));
  std::printf("granularity: minimum %zu, recommended %zu; chunk %zu MiB\n", gran_min, gran_rec, chunk >> 20);
  if (chunk % gran_min != 0 || table_bytes % chunk != 0) return 2;
  const size_t n = total / chunk, per_table = table_bytes / chunk;
  std::vector<hipMemGenericAllocationHandle_t> h(n);
  void* va = nullptr;
  if (reserve_first) CK(hipMemAddressReserve(&va, table_bytes, aligned_va ? chunk : 0, nullptr, 0));
  for (size_t i = 0; i < n; ++i) CK(hipMemCreate(&h[i], chunk, &prop, 0));
  std::printf("created %zu chunks (aligned_va %d, reserve_first %d), table %zu GiB = %zu chunks\n", n, (int)aligned_va, (int)reserve_first, table_bytes >> 30, per_table);
  if (!reserve_first) CK(hipMemAddressReserve(&va, table_bytes, aligned_va ? chunk : 0, nullptr, 0));
  std::printf("va %p\n", va);
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  uint32_t* sink;
  CK(hipMalloc(&sink, 4));
  // permutation of the first per_table chunks (fixed LCG shuffle)
  std::vector<size_t> perm(per_table);
  for (size_t k = 0; k < per_table; ++k) perm[k] = k;
  uint64_t lcg = 0x9E3779B97F4A7C15ull;
  for (size_t k = per_table - 1; k > 0; --k) {
    lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
    std::swap(perm[k], perm[(size_t)((lcg >> 33) % (k + 1))]);
  }
  struct { const char* name; size_t first, stride; } maps[] = {
      {"first chunks (consecutive)", 0, 1},
      {"first chunks, shuffled order", 0, 0},
      {"last chunks (consecutive)", n - per_table, 1},
      {"every (n/per_table)-th chunk (spread)", 0, n / per_table},
      {"first chunks again", 0, 1}};
  for (auto& m : maps) {
    for (size_t k = 0; k < per_table; ++k)
      CK(hipMemMap((char*)va + k * chunk, chunk, 0, h[m.stride == 0 ? perm[k] : m.first + k * m.stride], 0));
    CK(hipMemSetAccess(va, table_bytes, &acc, 1));
    const double t_lcs = time_kernel((Slot*)va, cap, 7, sink), t_or = time_kernel((Slot*)va, cap, 8, sink),
                 t_st = time_kernel((Slot*)va, cap, 4, sink), t_ld = time_kernel((Slot*)va, cap, 1, sink);
    std::printf("%-32s load+cas+store %.2f  atomic-or %.2f  store %.2f  load %.2f us/step\n", m.name, t_lcs, t_or, t_st, t_ld);
    std::fflush(stdout);
    CK(hipDeviceSynchronize());
    CK(hipMemUnmap(va, table_bytes));
  }
  CK(hipMemAddressFree(va, table_bytes));
  for (auto& x : h) CK(hipMemRelease(x));
  // for comparison: plain hipMalloc tables in this process
  for (int k = 0; k < 3; ++k) {
    Slot* t;
    CK(hipMalloc(&t, table_bytes));
    std::printf("hipMalloc #%d                      load+cas+store %.2f us/step\n", k, time_kernel(t, cap, 7, sink));
  }
  return 0;
}
