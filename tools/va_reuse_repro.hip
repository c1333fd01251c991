// Library-free reproducer attempt (no torch, no libq2048) for the row loss seen on a RE-USED virtual address
// range (profiles/r04_reused_address_range.txt; include/q2048.h "q2048_table_free").
//   reserve -> create/map 2 MiB chunks -> memset 0 -> scattered load + CAS + store of a known pattern ->
//   count occupied slots (must equal the claims the kernel counted) -> unmap / release / hipMemAddressFree ->
//   reserve again (the runtime hands the same range out) -> map FRESH chunks -> ... eight iterations.
// Variants (argv[1]):  0  the range is never freed (every table gets a fresh range: the shipped rule)
//                      1  hipMemAddressFree after the unmap, no synchronize
//                      2  the same + hipDeviceSynchronize after the free
//                      3  freed, but the next reservation asks for a different address (hint = old + 1 TiB)
//   hipcc -O3 --offload-arch=gfx950 -o tools/variants/va_reuse_repro tools/va_reuse_repro.hip
//   tools/variants/va_reuse_repro MODE [cap_log2=27] [iterations=8] [tables per iteration=4]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

struct Slot { unsigned long long key; float q[4]; unsigned long long pad; };   // 32 B, as q2048_slot

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 32; x *= 0xd6e8feb86659fd93ull; x ^= x >> 32; x *= 0xd6e8feb86659fd93ull; x ^= x >> 32;
  return x;
}

// per lane and step: one key, linear probing from its home slot; a slot is claimed by CAS on the key word,
// the claimer stores a value derived from the key.  counters[0] = claims, [1] = keys met again, [2] = gave up
__global__ __launch_bounds__(256) void k_insert(Slot* table, uint64_t mask, int64_t lanes, int steps, uint32_t salt,
                                                unsigned long long* counters) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= lanes) return;
  unsigned claims = 0, again = 0, gave_up = 0;
  for (int t = 0; t < steps; ++t) {
    const uint64_t key = mix(((uint64_t)i << 32) ^ ((uint64_t)salt << 8) ^ (uint64_t)t) | 1ull;
    uint64_t at = (key >> 7) & mask;
    int tries = 0;
    for (; tries < 4096; ++tries, at = (at + 1) & mask) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      u32x4 v;
      asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(&table[at]) : "memory");
      uint64_t seen = (uint64_t)v.x | ((uint64_t)v.y << 32);
      if (seen == 0ull) seen = atomicCAS(&table[at].key, 0ull, key);
      if (seen == 0ull) { table[at].q[key & 3ull] = (float)(key & 0xFFFF); ++claims; break; }
      if (seen == key) { ++again; break; }
    }
    gave_up += tries == 4096;
  }
  if (claims) atomicAdd(&counters[0], (unsigned long long)claims);
  if (again) atomicAdd(&counters[1], (unsigned long long)again);
  if (gave_up) atomicAdd(&counters[2], (unsigned long long)gave_up);
}

// occupied slots, and slots whose stored value does not belong to their key
__global__ __launch_bounds__(256) void k_count(const Slot* table, uint64_t cap, unsigned long long* counters) {
  unsigned occ = 0, bad = 0;
  for (uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x; s < cap; s += (uint64_t)gridDim.x * 256) {
    const Slot v = table[s];
    if (v.key != 0ull) { ++occ; bad += v.q[v.key & 3ull] != (float)(v.key & 0xFFFF); }
  }
  if (occ) atomicAdd(&counters[4], (unsigned long long)occ);
  if (bad) atomicAdd(&counters[5], (unsigned long long)bad);
}

// scattered atomic OR of 0 (contents unchanged): what the library's placement probe does to every candidate
__global__ __launch_bounds__(256) void k_touch(Slot* table, uint64_t mask, int64_t lanes, int steps) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= lanes) return;
  for (int t = 0; t < steps; ++t)
    atomicOr(&table[mix(((uint64_t)i << 20) ^ (uint64_t)t) & mask].key, 0ull);
}

struct Table { void* va; std::vector<hipMemGenericAllocationHandle_t> h; };
static size_t g_chunk = (size_t)2 << 20, g_bytes, g_n;
static hipMemAllocationProp g_prop = {};

static Table map_table(void* hint) {
  Table t;
  t.va = nullptr;
  CK(hipMemAddressReserve(&t.va, g_bytes, g_chunk, hint, 0));
  t.h.resize(g_n);
  for (size_t k = 0; k < g_n; ++k) {
    CK(hipMemCreate(&t.h[k], g_chunk, &g_prop, 0));
    CK(hipMemMap((char*)t.va + k * g_chunk, g_chunk, 0, t.h[k], 0));
  }
  hipMemAccessDesc acc = {};
  acc.location = g_prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(t.va, g_bytes, &acc, 1));
  CK(hipMemset(t.va, 0, g_bytes));
  return t;
}

static void free_table(Table& t, int mode) {
  CK(hipDeviceSynchronize());
  for (size_t k = 0; k < g_n; ++k) CK(hipMemUnmap((char*)t.va + k * g_chunk, g_chunk));
  for (size_t k = 0; k < g_n; ++k) CK(hipMemRelease(t.h[k]));
  if (mode >= 1) CK(hipMemAddressFree(t.va, g_bytes));
  if (mode == 2) CK(hipDeviceSynchronize());
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? std::atoi(argv[1]) : 1;
  const int cap_log2 = argc > 2 ? std::atoi(argv[2]) : 27;
  const int iterations = argc > 3 ? std::atoi(argv[3]) : 8;
  const int candidates = argc > 4 ? std::atoi(argv[4]) : 4;   // tables mapped per iteration; all but the last freed at once
  g_bytes = sizeof(Slot) << cap_log2;
  g_n = g_bytes / g_chunk;
  const uint64_t cap = 1ull << cap_log2;
  CK(hipSetDevice(0));
  g_prop.type = hipMemAllocationTypePinned;
  g_prop.location.type = hipMemLocationTypeDevice;
  g_prop.location.id = 0;
  unsigned long long* counters;
  CK(hipMalloc(&counters, 64));
  void* hint = nullptr;
  std::printf("mode %d: 2^%d slots = %zu MiB from %zu chunks of 2 MiB, %d iterations, %d tables mapped per iteration\n", mode,
              cap_log2, g_bytes >> 20, g_n, iterations, candidates);
  int failures = 0;
  const int64_t lanes = 1 << 20;
  for (int it = 0; it < iterations; ++it) {
    // as the library's place_table did when the loss was seen: several tables mapped and touched, all but one freed
    std::vector<Table> tables;
    for (int c = 0; c < candidates; ++c) {
      tables.push_back(map_table(hint));
      hipLaunchKernelGGL(k_touch, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, (Slot*)tables.back().va, cap - 1, lanes, 16);
    }
    for (int c = 0; c + 1 < candidates; ++c) free_table(tables[c], mode);
    Table& t = tables.back();
    CK(hipMemset(counters, 0, 64));
    hipLaunchKernelGGL(k_count, dim3(2048), dim3(256), 0, 0, (const Slot*)t.va, cap, counters);
    unsigned long long c[8];
    CK(hipMemcpy(c, counters, 64, hipMemcpyDeviceToHost));
    const unsigned long long nonzero_on_arrival = c[4];
    CK(hipMemset(counters, 0, 64));
    for (int launch = 0; launch < 3; ++launch)      // load 0.35 at 2^27 slots
      hipLaunchKernelGGL(k_insert, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, (Slot*)t.va, cap - 1, lanes, 15,
                         (uint32_t)(it * 16 + launch), counters);
    hipLaunchKernelGGL(k_count, dim3(2048), dim3(256), 0, 0, (const Slot*)t.va, cap, counters);
    CK(hipMemcpy(c, counters, 64, hipMemcpyDeviceToHost));
    const bool ok = c[0] == c[4] && c[5] == 0 && c[2] == 0 && nonzero_on_arrival == 0;
    failures += !ok;
    std::printf("it %d va %p nonzero_on_arrival %llu claims %llu met_again %llu gave_up %llu occupied %llu wrong_value %llu %s\n",
                it, t.va, nonzero_on_arrival, c[0], c[1], c[2], c[4], c[5], ok ? "ok" : "MISMATCH");
    std::fflush(stdout);
    hint = mode == 3 ? (void*)((char*)t.va + ((size_t)1 << 40)) : nullptr;
    free_table(t, mode);
  }
  std::printf("mode %d: %d of %d iterations lost or corrupted rows\n", mode, failures, iterations);
  return 0;
}
