// Measurement tool (round 5, not product code): what the HIP virtual-memory calls behind q2048_table_grow cost,
// by table size, chunk size and what the process already holds -- and whether they can run in a host thread
// next to a stream of kernel launches without slowing them (the plan for q2048_table_grow_begin).
//   hipcc -O3 --offload-arch=gfx950 -pthread -o tools/variants/exp_vmm_cost tools/exp_vmm_cost.hip
//   exp_vmm_cost map   GIB CHUNK_MIB [HOLD_GIB=0]   map a table of GIB from chunks (holding HOLD_GIB mapped before), phases timed
//   exp_vmm_cost malloc GIB                          hipMalloc + hipMemset + hipFree of GIB
//   exp_vmm_cost bg    GIB CHUNK_MIB [HOLD_GIB=0]   the same mapping in a host thread while the main thread launches
//                                                    a ~50 us kernel back to back: launches per ms before / during / after
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Mapped { void* va = nullptr; size_t bytes = 0, chunk = 0; std::vector<hipMemGenericAllocationHandle_t> h; };

// phases: reserve, create (all chunks), map (all chunks), set access, memset (on `stream`, waited for)
static Mapped map_table(size_t bytes, size_t chunk, hipStream_t stream, const char* tag, bool interleave) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  Mapped m;
  m.bytes = bytes;
  m.chunk = chunk;
  const size_t n = bytes / chunk;
  m.h.resize(n);
  double t0 = now_ms();
  CK(hipMemAddressReserve(&m.va, bytes, chunk, nullptr, 0));
  double t1 = now_ms(), t_create = 0, t_map = 0;
  if (interleave) {
    for (size_t k = 0; k < n; ++k) {
      double a = now_ms();
      CK(hipMemCreate(&m.h[k], chunk, &prop, 0));
      double b = now_ms();
      CK(hipMemMap((char*)m.va + k * chunk, chunk, 0, m.h[k], 0));
      t_create += b - a;
      t_map += now_ms() - b;
    }
  } else {
    for (size_t k = 0; k < n; ++k) CK(hipMemCreate(&m.h[k], chunk, &prop, 0));
    t_create = now_ms() - t1;
    double a = now_ms();
    for (size_t k = 0; k < n; ++k) CK(hipMemMap((char*)m.va + k * chunk, chunk, 0, m.h[k], 0));
    t_map = now_ms() - a;
  }
  double t3 = now_ms();
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(m.va, bytes, &acc, 1));
  double t4 = now_ms();
  CK(hipMemsetAsync(m.va, 0, bytes, stream));
  CK(hipStreamSynchronize(stream));
  double t5 = now_ms();
  std::printf("%s: %zu GiB from %zu chunks of %zu MiB: reserve %.2f ms, create %.1f ms, map %.1f ms, set access %.1f ms, "
              "memset %.1f ms (%.2f TB/s), total %.1f ms\n", tag, bytes >> 30, n, chunk >> 20, t1 - t0, t_create, t_map,
              t4 - t3, t5 - t4, bytes / (t5 - t4) * 1e-9, t5 - t0);
  std::fflush(stdout);
  return m;
}

static void unmap_table(Mapped& m, const char* tag) {
  double t0 = now_ms();
  const size_t n = m.h.size();
  for (size_t k = 0; k < n; ++k) CK(hipMemUnmap((char*)m.va + k * m.chunk, m.chunk));
  double t1 = now_ms();
  for (size_t k = 0; k < n; ++k) CK(hipMemRelease(m.h[k]));
  double t2 = now_ms();
  std::printf("%s: unmap %.1f ms, release %.1f ms\n", tag, t1 - t0, t2 - t1);
  std::fflush(stdout);
}

__global__ __launch_bounds__(256) void k_spin(uint32_t* buf, int iters) {
  uint32_t x = threadIdx.x + blockIdx.x * 256;
  for (int i = 0; i < iters; ++i) x = x * 1664525u + 1013904223u + buf[(x >> 8) & 0xFFFFF];
  if (x == 0x12345u) buf[0] = x;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const char* mode = argv[1];
  const size_t gib = (size_t)std::atoi(argv[2]);
  const size_t chunk = (size_t)(argc > 3 ? std::atoi(argv[3]) : 64) << 20;
  const size_t hold = (size_t)(argc > 4 ? std::atoi(argv[4]) : 0);
  CK(hipSetDevice(0));
  CK(hipFree(nullptr));
  hipStream_t side;
  CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  size_t fr = 0, tot = 0;
  CK(hipMemGetInfo(&fr, &tot));
  std::printf("-- %s %zu GiB, chunk %zu MiB, holding %zu GiB; device free %.1f of %.1f GiB\n", mode, gib, chunk >> 20, hold,
              fr / 1073741824.0, tot / 1073741824.0);
  if (!std::strcmp(mode, "malloc")) {
    void* p = nullptr;
    double t0 = now_ms();
    CK(hipMalloc(&p, gib << 30));
    double t1 = now_ms();
    CK(hipMemset(p, 0, gib << 30));
    CK(hipDeviceSynchronize());
    double t2 = now_ms();
    CK(hipFree(p));
    double t3 = now_ms();
    std::printf("hipMalloc %.1f ms, memset %.1f ms, hipFree %.1f ms\n", t1 - t0, t2 - t1, t3 - t2);
    t0 = now_ms();
    CK(hipMalloc(&p, gib << 30));
    t1 = now_ms();
    CK(hipFree(p));
    std::printf("again: hipMalloc %.1f ms, hipFree %.1f ms\n", t1 - t0, now_ms() - t1);
    return 0;
  }
  Mapped held;
  if (hold) held = map_table(hold << 30, hold >= 8 ? ((size_t)32 << 20) : ((size_t)2 << 20), side, "held table", true);
  if (!std::strcmp(mode, "map")) {
    Mapped m = map_table(gib << 30, chunk, side, "table", true);
    unmap_table(m, "table");
    if (hold == 0) {
      Mapped m2 = map_table(gib << 30, chunk, side, "table, second time (fresh range)", false);
      unmap_table(m2, "second");
    }
    return 0;
  }
  // bg: launches per ms on the main thread before / during / after the mapping thread's work
  uint32_t* buf;
  CK(hipMalloc(&buf, 4 << 20));
  CK(hipMemset(buf, 0, 4 << 20));
  hipStream_t main_s;
  CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
  std::atomic<int> phase{0};          // 0 before, 1 mapping, 2 after
  std::thread worker;
  struct Bucket { double ms = 0, worst = 0; long launches = 0; } bucket[3];
  const int per_sync = 20;            // a "region": 20 launches, then a stream synchronize (as train.py reports)
  double t_start = now_ms();
  bool started = false;
  Mapped m;
  while (true) {
    const int ph = phase.load();
    double a = now_ms();
    for (int k = 0; k < per_sync; ++k) hipLaunchKernelGGL(k_spin, dim3(2048), dim3(256), 0, main_s, buf, 400);
    CK(hipStreamSynchronize(main_s));
    double d = now_ms() - a;
    bucket[ph].ms += d;
    bucket[ph].launches += per_sync;
    if (d > bucket[ph].worst) bucket[ph].worst = d;
    if (!started && now_ms() - t_start > 300) {
      started = true;
      phase = 1;
      worker = std::thread([&] {
        CK(hipSetDevice(0));
        m = map_table(gib << 30, chunk, side, "background table", true);
        phase = 2;
      });
    }
    if (phase.load() == 2 && bucket[2].ms > 300) break;
  }
  worker.join();
  const char* names[3] = {"before", "while mapping", "after"};
  for (int p = 0; p < 3; ++p)
    std::printf("%-14s %ld launches in %.1f ms: %.1f us per launch, worst 20-launch region %.2f ms\n", names[p], bucket[p].launches,
                bucket[p].ms, bucket[p].ms * 1e3 / (bucket[p].launches ? bucket[p].launches : 1), bucket[p].worst);
  // and the release of the held table in the background
  if (hold) {
    phase = 0;
    Bucket b2[2];
    std::atomic<int> done{0};
    std::thread w2([&] { CK(hipSetDevice(0)); unmap_table(held, "held table, in the background"); done = 1; });
    while (true) {
      const int ph = done.load();
      double a = now_ms();
      for (int k = 0; k < per_sync; ++k) hipLaunchKernelGGL(k_spin, dim3(2048), dim3(256), 0, main_s, buf, 400);
      CK(hipStreamSynchronize(main_s));
      double d = now_ms() - a;
      b2[ph].ms += d;
      b2[ph].launches += per_sync;
      if (d > b2[ph].worst) b2[ph].worst = d;
      if (ph == 1 && b2[1].ms > 200) break;
    }
    w2.join();
    std::printf("while unmapping %ld launches in %.1f ms: %.1f us per launch, worst region %.2f ms; after: %.1f us per launch\n",
                b2[0].launches, b2[0].ms, b2[0].ms * 1e3 / (b2[0].launches ? b2[0].launches : 1), b2[0].worst,
                b2[1].ms * 1e3 / (b2[1].launches ? b2[1].launches : 1));
  }
  return 0;
}
