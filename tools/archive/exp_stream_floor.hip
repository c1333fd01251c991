// Measurement tool (not product code): the time of a kernel that moves exactly the bytes of the
// unfused 4x4 env step -- per board 16 B board + 16 B aux + 1 B action read, 16 + 16 + 4 + 1 + 1 B
// written -- with no arithmetic, at the env step's launch geometry (one board per thread, blocks of
// 256).  It bounds what k_env_step could reach at a given batch: launch ramp and tail included.
//   hipcc -O3 --offload-arch=gfx950 -o tools/variants/exp_stream_floor tools/archive/exp_stream_floor.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_move(uint4* boards, uint4* aux, const uint8_t* act, float* reward,
                                              uint8_t* done, uint8_t* mx, int64_t B) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B) return;
  uint4 b = boards[i], a = aux[i];
  const uint32_t k = act[i];
  b.x ^= k; a.y += b.z;
  boards[i] = b; aux[i] = a;
  reward[i] = (float)(a.x ^ b.w);
  done[i] = (uint8_t)(b.y >> 7);
  mx[i] = (uint8_t)(b.x & 15u);
}

int main() {
  for (int lg : {20, 22, 23}) {
    const int64_t B = (int64_t)1 << lg;
    uint4 *boards, *aux; uint8_t *act, *done, *mx; float* reward;
    CK(hipMalloc(&boards, B * 16)); CK(hipMalloc(&aux, B * 16)); CK(hipMalloc(&act, B)); CK(hipMalloc(&done, B));
    CK(hipMalloc(&mx, B)); CK(hipMalloc(&reward, B * 4));
    CK(hipMemset(boards, 1, B * 16)); CK(hipMemset(aux, 0, B * 16)); CK(hipMemset(act, 2, B));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(k_move, dim3((unsigned)(B / 256)), dim3(256), 0, 0, boards, aux, act, reward, done, mx, B);
    const int reps = 200;
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_move, dim3((unsigned)(B / 256)), dim3(256), 0, 0, boards, aux, act, reward, done, mx, B);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    std::printf("{\"kernel\": \"stream floor (70 B per board, no arithmetic)\", \"B\": %lld, \"us_per_launch\": %.2f, \"GBps\": %.1f}\n",
                (long long)B, us, 70.0 * B / us * 1e-3);
    CK(hipFree(boards)); CK(hipFree(aux)); CK(hipFree(act)); CK(hipFree(done)); CK(hipFree(mx)); CK(hipFree(reward));
  }
  return 0;
}
