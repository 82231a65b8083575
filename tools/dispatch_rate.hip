// Micro-benchmark: how fast does the dispatcher start the wavefronts of a launch -- 4096 workgroups of 64 threads vs
// 1024 of 256 (same 4096 wavefronts, same registers / LDS per wavefront)?  Every wavefront stamps the wall clock at entry.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int WPB>
__global__ __launch_bounds__(64 * WPB, 4 / WPB > 0 ? 4 / WPB : 1) void stamp(unsigned long long* out, float* sink, int spin) {
  __shared__ float pad[788 * WPB];   // ~3 KB of LDS per wavefront, like the step kernel
  const int wave = blockIdx.x * WPB + (threadIdx.x >> 6);
  const unsigned long long t0 = wall_clock64();
  float a = threadIdx.x;
  for (int i = 0; i < spin; i++) a = a * 1.0001f + 0.5f;   // ~20 us of work so that all wavefronts are co-resident
  pad[threadIdx.x] = a;
  if ((threadIdx.x & 63) == 0) out[wave] = t0;
  if (a == 12345.f) sink[0] = pad[(threadIdx.x + 1) % (64 * WPB)];
}
template <int WPB>
void run(const char* name) {
  unsigned long long* d; float* s;
  hipMalloc(&d, 4096 * 8); hipMalloc(&s, 4);
  std::vector<unsigned long long> h(4096);
  double p50 = 0, p90 = 0, p100 = 0;
  for (int rep = 0; rep < 20; rep++) {
    hipLaunchKernelGGL(stamp<WPB>, dim3(4096 / WPB), dim3(64 * WPB), 0, 0, d, s, 12000);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, 4096 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    if (rep >= 4) { p50 += (h[2048] - h[0]) / 100.0; p90 += (h[3686] - h[0]) / 100.0; p100 += (h[4095] - h[0]) / 100.0; }
  }
  printf("%s: wavefront start after the first: p50 %.2f us  p90 %.2f us  last %.2f us\n", name, p50 / 16, p90 / 16, p100 / 16);
}
int main() { run<1>("4096 x 64 threads "); run<2>("2048 x 128 threads"); run<4>("1024 x 256 threads"); return 0; }
