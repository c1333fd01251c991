// Measurement tool (not product code): rocPRIM radix_sort_pairs of 2^20 (u64 key, u32 value) pairs
// over `bits` key bits, under several onesweep configurations; prints microseconds per sort.
//   hipcc -O3 --offload-arch=gfx950 -o tools/variants/exp_sort tools/archive/exp_sort.hip
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)
using u64 = unsigned long long;

template <class Config>
static void run(const char* name, u64* kin, u64* kout, uint32_t* vin, uint32_t* vout, size_t n, unsigned bits) {
  size_t bytes = 0;
  CK((rocprim::radix_sort_pairs<Config>(nullptr, bytes, kin, kout, vin, vout, n, 0u, bits, (hipStream_t)0)));
  void* tmp;
  CK(hipMalloc(&tmp, bytes + 256));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) CK((rocprim::radix_sort_pairs<Config>(tmp, bytes, kin, kout, vin, vout, n, 0u, bits, (hipStream_t)0)));
  CK(hipEventRecord(e0, 0));
  const int reps = 20;
  for (int r = 0; r < reps; ++r) CK((rocprim::radix_sort_pairs<Config>(tmp, bytes, kin, kout, vin, vout, n, 0u, bits, (hipStream_t)0)));
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  std::printf("{\"config\": \"%s\", \"n\": %zu, \"bits\": %u, \"us_per_sort\": %.1f, \"temp_bytes\": %zu}\n", name, n, bits, ms * 1e3 / reps, bytes);
  std::fflush(stdout);
  CK(hipFree(tmp));
}

template <class Config>
static void run16(const char* name, uint16_t* kin, uint16_t* kout, uint32_t* vin, uint32_t* vout, size_t n, unsigned bits) {
  size_t bytes = 0;
  CK((rocprim::radix_sort_pairs<Config>(nullptr, bytes, kin, kout, vin, vout, n, 0u, bits, (hipStream_t)0)));
  void* tmp;
  CK(hipMalloc(&tmp, bytes + 256));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) CK((rocprim::radix_sort_pairs<Config>(tmp, bytes, kin, kout, vin, vout, n, 0u, bits, (hipStream_t)0)));
  CK(hipEventRecord(e0, 0));
  const int reps = 20;
  for (int r = 0; r < reps; ++r) CK((rocprim::radix_sort_pairs<Config>(tmp, bytes, kin, kout, vin, vout, n, 0u, bits, (hipStream_t)0)));
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  std::printf("{\"config\": \"%s\", \"n\": %zu, \"bits\": %u, \"us_per_sort\": %.1f}\n", name, n, bits, ms * 1e3 / reps);
  std::fflush(stdout);
  CK(hipFree(tmp));
}

template <class Config>
static void run32(const char* name, uint32_t* kin, uint32_t* kout, uint32_t* vin, uint32_t* vout, size_t n, unsigned bits) {
  size_t bytes = 0;
  CK((rocprim::radix_sort_pairs<Config>(nullptr, bytes, kin, kout, vin, vout, n, 0u, bits, (hipStream_t)0)));
  void* tmp;
  CK(hipMalloc(&tmp, bytes + 256));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) CK((rocprim::radix_sort_pairs<Config>(tmp, bytes, kin, kout, vin, vout, n, 0u, bits, (hipStream_t)0)));
  CK(hipEventRecord(e0, 0));
  const int reps = 20;
  for (int r = 0; r < reps; ++r) CK((rocprim::radix_sort_pairs<Config>(tmp, bytes, kin, kout, vin, vout, n, 0u, bits, (hipStream_t)0)));
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  std::printf("{\"config\": \"%s\", \"n\": %zu, \"bits\": %u, \"us_per_sort\": %.1f}\n", name, n, bits, ms * 1e3 / reps);
  std::fflush(stdout);
  CK(hipFree(tmp));
}

template <unsigned BS, unsigned IPT, unsigned RB, rocprim::block_radix_rank_algorithm A>
using Cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                       rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<BS, IPT>, RB, A>, 8192>;

int main(int argc, char** argv) {
  const size_t n = (size_t)1 << (argc > 1 ? std::atoi(argv[1]) : 20);
  const unsigned bits = argc > 2 ? (unsigned)std::atoi(argv[2]) : 33;
  std::vector<u64> k(n); std::vector<uint32_t> v(n);
  u64 x = 88172645463325252ull;
  for (size_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; k[i] = x & ((1ull << bits) - 1); v[i] = (uint32_t)i; }
  u64 *kin, *kout; uint32_t *vin, *vout;
  CK(hipMalloc(&kin, n * 8)); CK(hipMalloc(&kout, n * 8)); CK(hipMalloc(&vin, n * 4)); CK(hipMalloc(&vout, n * 4));
  CK(hipMemcpy(kin, k.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(vin, v.data(), n * 4, hipMemcpyHostToDevice));
  using R = rocprim::block_radix_rank_algorithm;
  run<rocprim::default_config>("default (merge sort up to 2^20)", kin, kout, vin, vout, n, bits);
  run<rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 8192>>("default onesweep", kin, kout, vin, vout, n, bits);
  run<Cfg<256, 12, 8, R::match>>("256x12 8b match", kin, kout, vin, vout, n, bits);
  run<Cfg<256, 16, 8, R::match>>("256x16 8b match", kin, kout, vin, vout, n, bits);
  run<Cfg<512, 8, 8, R::match>>("512x8 8b match", kin, kout, vin, vout, n, bits);
  run<Cfg<512, 12, 8, R::match>>("512x12 8b match", kin, kout, vin, vout, n, bits);
  run<Cfg<1024, 4, 8, R::match>>("1024x4 8b match", kin, kout, vin, vout, n, bits);
  run<Cfg<256, 12, 8, R::basic_memoize>>("256x12 8b basic_memoize", kin, kout, vin, vout, n, bits);
  run<Cfg<256, 12, 7, R::match>>("256x12 7b match", kin, kout, vin, vout, n, bits);
  run<Cfg<256, 12, 6, R::match>>("256x12 6b match", kin, kout, vin, vout, n, bits);
  run<Cfg<256, 8, 8, R::match>>("256x8 8b match", kin, kout, vin, vout, n, bits);
  run<Cfg<1024, 6, 8, R::match>>("1024x6 8b match", kin, kout, vin, vout, n, bits);
  run<Cfg<1024, 8, 8, R::match>>("1024x8 8b match", kin, kout, vin, vout, n, bits);
  run<Cfg<1024, 2, 8, R::match>>("1024x2 8b match", kin, kout, vin, vout, n, bits);
  run<Cfg<1024, 4, 7, R::match>>("1024x4 7b match", kin, kout, vin, vout, n, bits);
  {  // 16-bit keys: the deterministic mode only needs the low 16 bits of the group word sorted
    uint16_t *k16, *k16o;
    CK(hipMalloc(&k16, n * 2)); CK(hipMalloc(&k16o, n * 2));
    std::vector<uint16_t> kk(n);
    for (size_t i = 0; i < n; ++i) kk[i] = (uint16_t)k[i];
    CK(hipMemcpy(k16, kk.data(), n * 2, hipMemcpyHostToDevice));
    run16<rocprim::default_config>("u16 keys, default", k16, k16o, vin, vout, n, 16);
    run16<rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 8192>>("u16 keys, default onesweep", k16, k16o, vin, vout, n, 16);
    run16<rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<1024, 4>, 8, R::match>, 8192>>("u16 keys, 1024x4 8b match", k16, k16o, vin, vout, n, 16);
    run16<rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<512, 8>, 8, R::match>, 8192>>("u16 keys, 512x8 8b match", k16, k16o, vin, vout, n, 16);
    run16<rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<1024, 8>, 8, R::match>, 8192>>("u16 keys, 1024x8 8b match", k16, k16o, vin, vout, n, 16);
  }
  // 32-bit keys (tables of up to 2^29 slots: slot << 2 | action, + the drop bit, fit)
  if (bits <= 32) {
    uint32_t* k32 = reinterpret_cast<uint32_t*>(kin);
    uint32_t* k32o = reinterpret_cast<uint32_t*>(kout);
    std::vector<uint32_t> kk(n);
    for (size_t i = 0; i < n; ++i) kk[i] = (uint32_t)k[i];
    CK(hipMemcpy(k32, kk.data(), n * 4, hipMemcpyHostToDevice));
    run32<rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 8192>>("u32 keys, default onesweep", k32, k32o, vin, vout, n, bits);
    run32<rocprim::default_config>("u32 keys, default (merge sort)", k32, k32o, vin, vout, n, bits);
    run32<rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<1024, 4>, 8, R::match>, 8192>>("u32 keys, 1024x4 8b match", k32, k32o, vin, vout, n, bits);
  }
  return 0;
}
