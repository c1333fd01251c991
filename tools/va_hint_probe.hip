// Measurement tool (not product code), round 6.  Two questions about virtual addresses, asked WITHOUT touching any memory
// whose mapping is in doubt:
//   1. When a large hipMalloc is freed, does the next hipMemAddressReserve (no hint) land inside the range it occupied?
//      (a virtual range that is mapped a second time serves stale translations on ROCm 7.2.  q2048_table_free
//      keeps its own ranges reserved for ever -- but a range freed by ANOTHER allocator, e.g. torch's, could be handed to
//      the table allocator just the same.  tests: the GPU fault of gpurun_out/r06e, right after a 275 GiB tensor was freed.)
//   2. Is an address hint far away from where the runtime allocates (a private region, here 16 TiB upward) honoured, and
//      does a chunk mapped there work?  (One 2 MiB chunk: filled, read back, unmapped.)
//   hipcc -O2 --offload-arch=gfx950 -o tools/variants/va_hint_probe tools/va_hint_probe.hip && tools/variants/va_hint_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      std::printf("{\"error\": \"%s:%d %s\"}\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      return 1;                                                                                \
    }                                                                                          \
  } while (0)

int main(int argc, char** argv) {
  const size_t gib = argc > 1 ? (size_t)atoll(argv[1]) : 64;
  const size_t big = gib << 30, align = (size_t)32 << 20;
  // 1. where do hipMalloc and an un-hinted reservation land, before and after a free
  void* p = nullptr;
  CK(hipMalloc(&p, big));
  void* r0 = nullptr;
  CK(hipMemAddressReserve(&r0, big, align, nullptr, 0));          // while p is alive: cannot overlap it
  CK(hipMemAddressFree(r0, big));                                 // (never mapped: safe to give back)
  CK(hipFree(p));
  CK(hipDeviceSynchronize());
  void* r1 = nullptr;
  CK(hipMemAddressReserve(&r1, big, align, nullptr, 0));          // after the free
  const uintptr_t a = (uintptr_t)p, b = (uintptr_t)r1;
  const bool overlap = b < a + big && a < b + big;
  CK(hipMemAddressFree(r1, big));
  // 2. a hint in a private region
  const uintptr_t hint = (uintptr_t)0x100000000000ull;             // 16 TiB
  void* r2 = nullptr;
  const hipError_t he = hipMemAddressReserve(&r2, big, align, (void*)hint, 0);
  bool works = false;
  if (he == hipSuccess && (uintptr_t)r2 == hint) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    int dev = 0;
    CK(hipGetDevice(&dev));
    prop.location.id = dev;
    const size_t chunk = (size_t)2 << 20;
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, chunk, &prop, 0));
    CK(hipMemMap(r2, chunk, 0, h, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(r2, chunk, &acc, 1));
    CK(hipMemset(r2, 0x5a, chunk));
    std::vector<unsigned char> host(chunk);
    CK(hipMemcpy(host.data(), r2, chunk, hipMemcpyDeviceToHost));
    works = true;
    for (size_t k = 0; k < chunk; k += 4097) works = works && host[k] == 0x5a;
    CK(hipMemUnmap(r2, chunk));
    CK(hipMemRelease(h));
  }
  std::printf("{\"hipMalloc_GiB\": %zu, \"hipMalloc_at\": \"%p\", \"reserve_while_alive\": \"%p\", \"reserve_after_free\": \"%p\", "
              "\"reserve_after_free_overlaps_the_freed_range\": %s, \"hint\": \"%p\", \"hint_result\": \"%s\", \"reserved_at\": \"%p\", "
              "\"hint_honoured\": %s, \"chunk_mapped_at_the_hint_works\": %s}\n",
              gib, p, r0, r1, overlap ? "true" : "false", (void*)hint, hipGetErrorString(he), r2,
              (he == hipSuccess && (uintptr_t)r2 == hint) ? "true" : "false", works ? "true" : "false");
  return 0;
}
