// Micro-benchmark: does a launch start with a cold instruction cache?  Every wavefront of a 544 x 256 grid runs the same
// straight-line block of ~8 KB of code twice (a 2-trip loop): the first pass fetches it, the second finds it cached.
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
#define OP4 asm volatile("v_add_u32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1" : "+v"(x) : "v"(y));
#define OP16 OP4 OP4 OP4 OP4
#define OP64 OP16 OP16 OP16 OP16
#define OP256 OP64 OP64 OP64 OP64
__global__ __launch_bounds__(256) void probe(unsigned long long* out, unsigned* sink, unsigned y) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  unsigned x = threadIdx.x;
  unsigned long long t[3];
  t[0] = now();
#pragma unroll 1
  for (int k = 0; k < 2; k++) {
    OP256 OP256 OP256 OP256      // 1024 VALU instructions, 4 bytes each = 4 KB (VOP2)
    asm volatile("" : "+v"(x));
    t[k + 1] = now();
  }
  if (lane == 0) { out[wave * 4 + 0] = t[0]; out[wave * 4 + 1] = t[1]; out[wave * 4 + 2] = t[2]; }
  if (x == 0x12345678u) sink[0] = x;
}
int main() {
  const int W = NB * 4;
  unsigned long long* d; unsigned* s;
  hipMalloc(&d, W * 32); hipMalloc(&s, 4);
  std::vector<unsigned long long> h(W * 4);
  double p1 = 0, p2 = 0, p1max = 0;
  for (int rep = 0; rep < 24; rep++) {
    hipLaunchKernelGGL(probe, dim3(NB), dim3(256), 0, 0, d, s, 3u + rep);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, W * 32, hipMemcpyDeviceToHost);
    if (rep < 4) continue;
    double s1 = 0, s2 = 0, m1 = 0;
    for (int w = 0; w < W; w++) {
      const double d1 = (h[w * 4 + 1] - h[w * 4]) / 100.0, d2 = (h[w * 4 + 2] - h[w * 4 + 1]) / 100.0;
      s1 += d1; s2 += d2; m1 = std::max(m1, d1);
    }
    p1 += s1 / W; p2 += s2 / W; p1max += m1;
  }
  printf("1024 VALU instructions (4 KB of code): first pass %.2f us (max %.2f), second pass %.2f us\n", p1 / 20, p1max / 20, p2 / 20);
  return 0;
}
