// Measurement tool (not product code): what do N scattered requests per lane cost on MI355X, and
// do they overlap with arithmetic?  One lane = one pseudo-board; per step a lane derives a
// random slot of a 32-B-row table and issues any subset of {16-B probe load, 8-B compare-and-swap
// on the key word, 4-B store into the row it touched one step earlier}, next to `work` rounds of
// Philox-like integer arithmetic.  Prints microseconds per 1 Mi-lane step for each combination.
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/variants/exp_requests tools/archive/exp_requests.hip
//   tools/variants/exp_requests [cap_log2=28] [lanes_log2=20] [steps=64] [alloc_log2=cap_log2]
//                               [only: run the combinations whose name contains this] [alloc mode:
//                               0 hipMalloc, 1 fine-grained, 3 uncached (hipExtMallocWithFlags),
//                               4 mapped from 2 MiB physical chunks (HIP virtual-memory API: what
//                               q2048_table_alloc does)] [claim rate in 1/1024: how many of the lane-steps
//                               that found their slot empty go on to claim it, default 1024] [MiB per chunk, mode 4]
// Round 4 added: the 5x5 rollout's pattern (pub8: the winner of the claim publishes the second key
// word with an 8-byte write-through store), a claim rate, and chunked tables.
// Round 2 added: whole 64- / 128-byte lines written by one lane, a 1-bit-per-slot occupancy
// bitmap (load + atomic OR as the claim), every cache-policy flavour of the probe load.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      std::exit(1);                                                                  \
    }                                                                                \
  } while (0)

struct Slot { unsigned long long key; float q[4]; unsigned long long pad; };
static_assert(sizeof(Slot) == 32, "slot");

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 32; x *= 0xd6e8feb86659fd93ull; x ^= x >> 32; x *= 0xd6e8feb86659fd93ull; x ^= x >> 32;
  return x;
}
// `work` units of dependent 32x32->64 multiplies and xors (one unit ~ one Philox4x32-10 call)
__device__ __forceinline__ uint32_t grind(uint32_t a, uint32_t b, int work) {
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

enum { kLoad = 1, kCas = 2, kStore = 4, kCasAlways = 8, kStore16 = 16, kStore32 = 32, kCas2 = 64, kKey16 = 128,
       kStore64 = 256, kStore128 = 512, kBmLoad = 1024, kBmOr = 2048, kBmOrAlways = 4096, kLoadIfSet = 8192,
       kPub8 = 16384, kLoad2 = 32768 };

__global__ __launch_bounds__(256) void k_requests(Slot* table, uint64_t mask, int64_t lanes, int steps,
                                                  int what, int work, uint32_t ctr0, uint32_t* sink,
                                                  uint32_t* bitmap, int flavor, uint32_t claim_rate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= lanes) return;
  uint32_t acc = (uint32_t)i;
  uint64_t prev = mix((uint64_t)i) & mask;
  for (int t = 0; t < steps; ++t) {
    acc = grind(acc, (uint32_t)t + ctr0, work);
    const uint64_t key = mix(((uint64_t)i << 32) ^ (uint64_t)(ctr0 + (uint32_t)t) ^ ((uint64_t)acc << 13)) | 1ull;
    const uint64_t at = (key >> 7) & mask;
    uint64_t seen = 0ull;
    // occupancy bitmap (1 bit per slot, cache-resident up to 2^31 slots): a 4-B sc1 load, then an
    // atomic OR that claims the slot when the bit was clear
    bool bit_set = false;
    if (what & kBmLoad) {
      uint32_t w;
      asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(&bitmap[at >> 5]) : "memory");
      bit_set = (w >> (at & 31ull)) & 1u;
    }
    if ((what & kBmOrAlways) || ((what & kBmOr) && !bit_set)) {
      const uint32_t old = atomicOr(&bitmap[at >> 5], 1u << (at & 31ull));
      acc ^= old;
      bit_set = bit_set || ((old >> (at & 31ull)) & 1u);
    }
    if ((what & kLoad) || ((what & kLoadIfSet) && bit_set)) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      u32x4 v, v2 = {0u, 0u, 0u, 0u};
      const Slot* p = &table[at];
      const char* p2 = reinterpret_cast<const char*>(p) + 16;
      // cache-policy flavours of the probe load (flavor & 7) and, with flavor & 8, the second half
      // of the row in the same round trip
#define LD(BITS)                                                                                   \
  if (flavor & 8)                                                                                  \
    asm volatile("global_load_dwordx4 %0, %2, off " BITS "\n\tglobal_load_dwordx4 %1, %3, off " BITS \
                 "\n\ts_waitcnt vmcnt(0)" : "=&v"(v), "=&v"(v2) : "v"(p), "v"(p2) : "memory");     \
  else                                                                                             \
    asm volatile("global_load_dwordx4 %0, %1, off " BITS "\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
      switch (flavor & 7) {
        case 0: LD("sc1") break;
        case 1: LD("") break;
        case 2: LD("nt") break;
        case 3: LD("sc0 sc1") break;
        case 4: LD("sc0 sc1 nt") break;
        case 5: LD("sc1 nt") break;
        default: LD("sc0") break;
      }
#undef LD
      seen = (uint64_t)v.x | ((uint64_t)v.y << 32);
      acc ^= v.z ^ v2.x;
    }
    if ((what & kLoad2) && ((key >> 40) & 1023ull) >= claim_rate) {   // a hit: the row's second half (5x5: two loads)
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      u32x4 v;
      asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(reinterpret_cast<const char*>(&table[at]) + 16) : "memory");
      acc ^= v.x;
    }
    if ((what & kCasAlways) || ((what & kCas) && seen == 0ull && ((key >> 40) & 1023ull) < claim_rate)) {
      const uint64_t r = atomicCAS(&table[at].key, 0ull, key);
      acc ^= (uint32_t)r;                       // the result is consumed (like the real claim)
      if ((what & kPub8) && r == 0ull)          // 5x5: the winner publishes the second key word, write-through
        __hip_atomic_store(&table[at].pad, key ^ 0x5555ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (what & kCas2) {                       // a second word of the same row (two-word keys)
        const uint64_t r2 = atomicCAS(&table[at].pad, 0ull, key ^ 0x5555ull);
        acc ^= (uint32_t)r2;
      }
      if (what & kKey16)                        // ... or the key body written by one 16-B store
        reinterpret_cast<uint4*>(&table[at])[1] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), acc | 1u, 7u);
    }
    if (what & kStore) *reinterpret_cast<uint32_t*>(&table[prev].q[key & 3ull]) = acc;
    if (what & (kStore16 | kStore32)) {        // whole 16-B half / whole 32-B row (no partial sector)
      uint4* row = reinterpret_cast<uint4*>(&table[prev]);
      row[0] = make_uint4((uint32_t)key | 1u, (uint32_t)(key >> 32), acc, acc);
      if (what & kStore32) row[1] = make_uint4(acc, acc, 0u, 0u);
    }
    if (what & (kStore64 | kStore128)) {       // a whole 64-B / 128-B line written by one lane
      const uint64_t lines = (what & kStore128) ? 3ull : 1ull;
      uint4* row = reinterpret_cast<uint4*>(&table[prev & ~lines]);
      const int n16 = (what & kStore128) ? 8 : 4;
      for (int k = 0; k < n16; ++k) row[k] = make_uint4((uint32_t)key | 1u, (uint32_t)(key >> 32), acc, (uint32_t)k);
    }
    prev = at;
  }
  if (acc == 0x12345u) *sink = acc;
}

int main(int argc, char** argv) {
  const int cap_log2 = argc > 1 ? std::atoi(argv[1]) : 28;
  const int lanes_log2 = argc > 2 ? std::atoi(argv[2]) : 20;
  const int steps = argc > 3 ? std::atoi(argv[3]) : 64;
  const int alloc_log2 = argc > 4 ? std::atoi(argv[4]) : cap_log2;   // allocate more than is used
  const char* only = argc > 5 ? argv[5] : nullptr;                   // run only combos whose name contains this
  const int alloc_mode = argc > 6 ? std::atoi(argv[6]) : 0;          // 0 hipMalloc, 1 fine-grained, 3 uncached, 4 chunks
  const uint32_t claim_rate = argc > 7 ? (uint32_t)std::atoi(argv[7]) : 1024u;
  const size_t chunk_mib = argc > 8 ? (size_t)std::atoi(argv[8]) : 2;   // alloc mode 4: MiB per physical chunk
  if (cap_log2 < 10 || cap_log2 > 32 || alloc_log2 < cap_log2 || alloc_log2 > 32 || lanes_log2 < 6 || lanes_log2 > 24 || steps < 1 || steps > 4096) {
    std::fprintf(stderr, "bad arguments\n");
    return 2;
  }
  const uint64_t cap = 1ull << cap_log2;
  const int64_t lanes = (int64_t)1 << lanes_log2;
  Slot* table;
  uint32_t* sink;
  uint32_t* bitmap;
  if (alloc_mode == 0) CK(hipMalloc(&table, (1ull << alloc_log2) * sizeof(Slot)));
  else if (alloc_mode == 4) {                  // 2 MiB physical chunks mapped into one reserved range
    int dev = 0;
    CK(hipGetDevice(&dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    const size_t chunk = chunk_mib << 20, bytes = (1ull << alloc_log2) * sizeof(Slot);
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, bytes + chunk, chunk, nullptr, 0));
    va = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(va) + chunk - 1) / chunk * chunk);   // (aligned by hand)
    for (size_t k = 0; k < bytes / chunk; ++k) {
      hipMemGenericAllocationHandle_t h;
      CK(hipMemCreate(&h, chunk, &prop, 0));
      CK(hipMemMap(static_cast<char*>(va) + k * chunk, chunk, 0, h, 0));
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, bytes, &acc, 1));
    table = static_cast<Slot*>(va);
  } else CK(hipExtMallocWithFlags(reinterpret_cast<void**>(&table), (1ull << alloc_log2) * sizeof(Slot), (unsigned)alloc_mode));
  CK(hipMalloc(&bitmap, cap / 8));
  CK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const unsigned grid = (unsigned)((lanes + 255) / 256);
  struct { const char* name; int what; int flavor = 0; } combos[] = {
      {"none", 0}, {"load", kLoad}, {"store", kStore}, {"cas(always)", kCasAlways},
      {"load+cas", kLoad | kCas}, {"load+store", kLoad | kStore},
      {"load+cas+store", kLoad | kCas | kStore}, {"cas(always)+store", kCasAlways | kStore},
      {"store16", kStore16}, {"store32", kStore32}, {"load+store32", kLoad | kStore32},
      {"load+cas+store32", kLoad | kCas | kStore32},
      {"load+cas+cas2+store", kLoad | kCas | kCas2 | kStore},
      {"load+cas+key16+store", kLoad | kCas | kKey16 | kStore},
      {"load+cas+pub8+store", kLoad | kCas | kPub8 | kStore},
      {"load+load2+cas+pub8+store", kLoad | kLoad2 | kCas | kPub8 | kStore},
      {"load+load2+cas+store", kLoad | kLoad2 | kCas | kStore},
      {"store64", kStore64}, {"store128", kStore128}, {"load+store64", kLoad | kStore64},
      {"bmload", kBmLoad}, {"bmor(always)", kBmOrAlways}, {"bmload+bmor", kBmLoad | kBmOr},
      {"bmload+bmor+store32", kBmLoad | kBmOr | kStore32},
      {"bmload+bmor+store64", kBmLoad | kBmOr | kStore64},
      {"bmload+bmor+store", kBmLoad | kBmOr | kStore},
      {"bmor(always)+store32", kBmOrAlways | kStore32},
      {"bmload+bmor+load(if set)+store32", kBmLoad | kBmOr | kLoadIfSet | kStore32},
      {"load[plain]", kLoad, 1}, {"load[nt]", kLoad, 2}, {"load[sc0 sc1]", kLoad, 3},
      {"load[sc0 sc1 nt]", kLoad, 4}, {"load[sc1 nt]", kLoad, 5}, {"load[sc0]", kLoad, 6},
      {"load32[sc1]", kLoad, 8}, {"load32[nt]", kLoad, 8 | 2}, {"load32[sc0 sc1]", kLoad, 8 | 3},
      {"load32[sc0 sc1 nt]", kLoad, 8 | 4},
      {"load[nt]+cas+store", kLoad | kCas | kStore, 2}, {"load[sc0 sc1]+cas+store", kLoad | kCas | kStore, 3},
      {"load[sc0 sc1 nt]+cas+store", kLoad | kCas | kStore, 4}, {"load[sc1 nt]+cas+store", kLoad | kCas | kStore, 5},
      {"load[nt]+store", kLoad | kStore, 2}, {"load[sc0 sc1]+store", kLoad | kStore, 3},
      {"load32[sc0 sc1]+cas+store", kLoad | kCas | kStore, 8 | 3}};
  std::printf("{\"chunk_mib\": %zu, \"claim_rate_1024\": %u, \"alloc_mode\": %d, \"alloc_log2\": %d, \"cap_log2\": %d, \"lanes\": %lld, \"steps\": %d, \"unit\": \"us per step per 2^20 lanes\", \"rows\": [\n",
              chunk_mib, claim_rate, alloc_mode, alloc_log2, cap_log2, (long long)lanes, steps);
  bool first = true;
  for (auto& c : combos) {
    if (only != nullptr && std::strstr(c.name, only) == nullptr) continue;
    for (int work : {0}) {
      CK(hipMemsetAsync(table, 0, cap * sizeof(Slot), 0));       // every run starts on an empty table
      CK(hipMemsetAsync(bitmap, 0, cap / 8, 0));
      uint32_t ctr = 0;
      hipLaunchKernelGGL(k_requests, dim3(grid), dim3(256), 0, 0, table, cap - 1, lanes, steps, c.what,
                         work, ctr, sink, bitmap, c.flavor, claim_rate);      // warm-up (fills steps*lanes keys)
      ctr += (uint32_t)steps;
      CK(hipEventRecord(e0, 0));
      const int reps = 3;
      for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(k_requests, dim3(grid), dim3(256), 0, 0, table, cap - 1, lanes, steps, c.what,
                           work, ctr, sink, bitmap, c.flavor, claim_rate);
        ctr += (uint32_t)steps;
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = (double)ms * 1e3 / (reps * steps) * (double)(1 << 20) / (double)lanes;
      std::printf("%s  {\"requests\": \"%s\", \"work\": %d, \"us\": %.2f}", first ? "" : ",\n", c.name, work, us);
      first = false;
      std::fflush(stdout);
    }
  }
  std::printf("\n]}\n");
  if (alloc_mode != 4) CK(hipFree(table));     // (chunked: the process ends here)
  CK(hipFree(bitmap));
  CK(hipFree(sink));
  return 0;
}
