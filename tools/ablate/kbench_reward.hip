// kbench_reward.hip -- micro-benchmark / ablation harness for sdc_reward_kernel (measurement tooling, not product).
// Includes the product translation unit so the ablations run the SAME device code:
//   full      the product kernel
//   read      only the ring stream (same grid, same 10 x dwordx4 per lane), keys folded so nothing is elided
//   compute   reward_one_env on synthetic register-resident keys (no ring loads)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o kbench_reward kbench_reward.hip
#include "../../dc-rl_amd/csrc/sdc_reward.hip"

#include <algorithm>
#include <cstring>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

extern "C" __global__ __launch_bounds__(SDC_BLOCK) void k_read(SdcDev S, unsigned* sink) {
  const int env = blockIdx.x, tid = threadIdx.x;
  const uint4* hp = reinterpret_cast<const uint4*>(S.hist + (size_t)env * SDC_HIST_STRIDE);
  unsigned acc = 0;
#pragma unroll
  for (int q = 0; q < SDC_HIST_PER_THREAD / 4; q++) {
    const uint4 v = hp[q * SDC_BLOCK + tid];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[env] = acc;
}

// the product's per-env body on L2-resident data: every workgroup reads one of 8 rings (320 KB in total), so the
// HBM / Infinity-Cache stream is gone but the instruction stream is exactly the product's
extern "C" __global__ __launch_bounds__(SDC_BLOCK) void k_compute(SdcDev S, float* rew, float* info) {
  __shared__ RewardShared sh;
  const int env = blockIdx.x, src = blockIdx.x & 7, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned key[SDC_HIST_PER_THREAD];
  const uint4* hp = reinterpret_cast<const uint4*>(S.hist + (size_t)src * SDC_HIST_STRIDE);
#pragma unroll
  for (int q = 0; q < SDC_HIST_PER_THREAD / 4; q++) {
    const uint4 v = hp[q * SDC_BLOCK + tid];
    key[4 * q + 0] = v.x; key[4 * q + 1] = v.y; key[4 * q + 2] = v.z; key[4 * q + 3] = v.w;
  }
  const unsigned hd = S.hdr[(size_t)src * SDC_HDR_DWORDS + (lane & (SDC_HDR_DWORDS - 1))];
  reward_one_env(S, sh, env, key, hd, rew, info, tid, lane, wave);
}

static float time_kernel(const char* name, int iters, hipStream_t st, const std::function<void()>& launch) {
  for (int i = 0; i < 5; i++) launch();
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a, st));
  for (int i = 0; i < iters; i++) launch();
  CK(hipEventRecord(b, st));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  printf("%-10s %8.2f us/launch\n", name, ms * 1e3f / iters);
  return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 4096;
  const int iters = argc > 2 ? atoi(argv[2]) : 200;
  SdcDev S{};
  S.n_envs = N; S.hist_cap = 10000;
  std::vector<unsigned> hist((size_t)N * SDC_HIST_STRIDE, 0xFFFFFFFFu), hdr((size_t)N * SDC_HDR_DWORDS, 0u);
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 70.f);
  for (int e = 0; e < N; e++) {
    for (int i = 0; i < 10000; i++) {
      float f = nd(rng);
      unsigned b; std::memcpy(&b, &f, 4);
      hist[(size_t)e * SDC_HIST_STRIDE + i] = sdc_f32_key(b);
    }
    unsigned* h = &hdr[(size_t)e * SDC_HDR_DWORDS];
    h[H_N] = 10000;
    // steady-state emulation: the "new" key equals the "evicted" key (ring unchanged between launches), so the
    // order-statistic trackers stay valid after the bootstrap launch and the timed launches take the fast path
    h[H_XNEW] = hist[(size_t)e * SDC_HIST_STRIDE + 17];
    h[H_XOLD] = hist[(size_t)e * SDC_HIST_STRIDE + 17];
    double eo = 12.5, nci = 0.4, old = 0.1;
    std::memcpy(h + H_EOFF, &eo, 8); std::memcpy(h + H_NORM_CI, &nci, 8); std::memcpy(h + H_OLDEST, &old, 8);
  }
  float *rew, *info; unsigned* sink;
  CK(hipMalloc(&S.hist, hist.size() * 4)); CK(hipMalloc(&S.hdr, hdr.size() * 4));
  CK(hipMalloc(&rew, (size_t)N * 3 * 4)); CK(hipMalloc(&info, (size_t)N * SDC_INFO_DIM * 4)); CK(hipMalloc(&sink, (size_t)N * 4));
  CK(hipMemcpy(S.hist, hist.data(), hist.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(S.hdr, hdr.data(), hdr.size() * 4, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  // first launches take the bisection fallback and store the quartile keys; afterwards the fast path runs
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(sdc_reward_kernel, dim3(N), dim3(SDC_BLOCK), 0, st, S, rew, info);
  CK(hipStreamSynchronize(st));
  std::vector<float> inf((size_t)N * SDC_INFO_DIM);
  CK(hipMemcpy(inf.data(), info, inf.size() * 4, hipMemcpyDeviceToHost));
  int p1 = 0, p2 = 0;
  for (int e = 0; e < N; e++) {
    p1 += inf[(size_t)e * SDC_INFO_DIM + SDC_INFO_RESERVED] == 1.f;
    p2 += inf[(size_t)e * SDC_INFO_DIM + SDC_INFO_RESERVED] == 2.f;
  }
  printf("N=%d  envs on the rebuild / bisection path in the timed launches: %d / %d   bytes/launch=%.1f MB\n", N, p1, p2,
         N * 40124.0 / 1e6);
  float tf = time_kernel("full", iters, st, [&] { hipLaunchKernelGGL(sdc_reward_kernel, dim3(N), dim3(SDC_BLOCK), 0, st, S, rew, info); });
  float tr = time_kernel("read", iters, st, [&] { hipLaunchKernelGGL(k_read, dim3(N), dim3(SDC_BLOCK), 0, st, S, sink); });
  float tc = time_kernel("compute", iters, st, [&] { hipLaunchKernelGGL(k_compute, dim3(N), dim3(SDC_BLOCK), 0, st, S, rew, info); });
  printf("full %.2f TB/s   read %.2f TB/s   (compute+read)/full = %.2f\n", N * 40124.0 / tf / 1e6, N * 40960.0 / tr / 1e6,
         (tc + tr) / tf);
  return 0;
}
