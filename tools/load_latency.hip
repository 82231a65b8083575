// Micro-benchmark: what does a memory round trip cost at the START of a launch, when all 2176 wavefronts of a
// 544 x 256-thread grid ask at once (as the step kernel's wavefronts do for their state records)?  Every wavefront loads
// 512 contiguous bytes, then -- dependent on the value -- another 512 bytes elsewhere, and stamps the wall clock around both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define NB 544
__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}
__global__ __launch_bounds__(256) void probe(uint2* a, const uint2* __restrict__ b, unsigned long long* out, size_t stride_b, int wr) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const unsigned long long t0 = now();
  uint2 x = a[(size_t)wave * 64 + lane];
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(x.x), "+v"(x.y) :: "memory");
  const unsigned long long t1 = now();
  uint2 y = b[(size_t)wave * stride_b + (x.x & 63u)];
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(y.x), "+v"(y.y) :: "memory");
  const unsigned long long t2 = now();
  if (wr == 1) a[(size_t)wave * 64 + lane] = make_uint2(x.x, x.y + 1);                                   // the state goes back: plain store
  if (wr == 2) __builtin_nontemporal_store(((unsigned long long)(x.y + 1) << 32) | x.x, reinterpret_cast<unsigned long long*>(a) + (size_t)wave * 64 + lane);
  if (lane == 0) { out[wave * 4 + 0] = t0; out[wave * 4 + 1] = t1; out[wave * 4 + 2] = t2; out[wave * 4 + 3] = x.y + y.y; }
}
int main() {
  const int W = NB * 4;
  uint2 *a, *b; unsigned long long* d;
  const size_t stride_b = 86 * 1024 / 8;     // like one feature row per env, 86 KB apart
  hipMalloc(&a, (size_t)W * 512); hipMalloc(&b, (size_t)W * stride_b * 8); hipMalloc(&d, W * 32);
  std::vector<uint2> ha((size_t)W * 64);
  for (size_t i = 0; i < ha.size(); i++) ha[i] = make_uint2((unsigned)(i % 64), 1u);
  hipMemcpy(a, ha.data(), ha.size() * 8, hipMemcpyHostToDevice);
  hipMemset(b, 0, (size_t)W * stride_b * 8);
  std::vector<unsigned long long> h(W * 4);
  for (int variant = 0; variant < 4; variant++) {
    const size_t sb = variant == 0 ? stride_b : 64; const int wr = variant < 2 ? 0 : variant - 1;    // scattered (one row per 86 KB) vs adjacent rows
    double l1 = 0, l2 = 0, l1max = 0, l2max = 0, span = 0;
    for (int rep = 0; rep < 24; rep++) {
      hipLaunchKernelGGL(probe, dim3(NB), dim3(256), 0, 0, a, b, d, sb, wr);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), d, W * 32, hipMemcpyDeviceToHost);
      if (rep < 4) continue;
      double s1 = 0, s2 = 0, m1 = 0, m2 = 0; unsigned long long tmin = ~0ull, tmax = 0;
      for (int w = 0; w < W; w++) {
        const double d1 = (h[w * 4 + 1] - h[w * 4]) / 100.0, d2 = (h[w * 4 + 2] - h[w * 4 + 1]) / 100.0;
        s1 += d1; s2 += d2; m1 = std::max(m1, d1); m2 = std::max(m2, d2);
        tmin = std::min(tmin, h[w * 4]); tmax = std::max(tmax, h[w * 4 + 2]);
      }
      l1 += s1 / W; l2 += s2 / W; l1max += m1; l2max += m2; span += (tmax - tmin) / 100.0;
    }
    printf("%s: first load mean %.2f us (max %.2f), dependent load mean %.2f us (max %.2f), first entry -> last done %.2f us\n",
           variant == 0 ? "read only, second load scattered (86 KB apart)" : variant == 1 ? "read only, second load adjacent rows          " : variant == 2 ? "first line rewritten every launch (plain)     " : "first line rewritten every launch (nt)        ", l1 / 20, l1max / 20, l2 / 20, l2max / 20, span / 20);
  }
  return 0;
}
