// Micro-benchmark: wavefront start times of a 544 x 256-thread launch (the step kernel's grid at 4096 envs) as a function
// of what the dispatcher has to allocate per wavefront: VGPRs, LDS, scratch.  Every wavefront stamps the wall clock at entry.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define NB 544
template <int VG, int LDS_FLOATS, int SCRATCH>
__global__ __launch_bounds__(256) void stamp(unsigned long long* out, float* sink, int spin, int idx) {
  __shared__ float pad[LDS_FLOATS > 256 ? LDS_FLOATS : 256];
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const unsigned long long t0 = wall_clock64();
  if (VG == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
  if (VG == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
  if (VG == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  if (VG == 168) asm volatile("v_mov_b32 v167, 0" ::: "v167");
  if (VG == 256) asm volatile("v_mov_b32 v255, 0" ::: "v255");
  float a = threadIdx.x;
  if (SCRATCH) {
    volatile float loc[8];
    for (int k = 0; k < 8; k++) loc[k] = a + k;
    a = loc[idx & 7];
  }
  for (int i = 0; i < spin; i++) a = a * 1.0001f + 0.5f;   // ~20 us of work so that all wavefronts are co-resident
  pad[threadIdx.x] = a;
  if ((threadIdx.x & 63) == 0) out[wave] = t0;
  if (a == 12345.f) sink[0] = pad[(threadIdx.x + 1) % 256];
}
template <int VG, int LDS_FLOATS, int SCRATCH>
void run(const char* name) {
  unsigned long long* d; float* s;
  hipMalloc(&d, NB * 4 * 8); hipMalloc(&s, 4);
  std::vector<unsigned long long> h(NB * 4);
  double p50 = 0, p90 = 0, p100 = 0;
  for (int rep = 0; rep < 20; rep++) {
    hipLaunchKernelGGL((stamp<VG, LDS_FLOATS, SCRATCH>), dim3(NB), dim3(256), 0, 0, d, s, 12000, rep);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, NB * 4 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    if (rep >= 4) { p50 += (h[NB * 2] - h[0]) / 100.0; p90 += (h[NB * 36 / 10] - h[0]) / 100.0; p100 += (h[NB * 4 - 1] - h[0]) / 100.0; }
  }
  printf("%s: wavefront start after the first: p50 %.2f us  p90 %.2f us  last %.2f us\n", name, p50 / 16, p90 / 16, p100 / 16);
}
int main() {
  run<0, 0, 0>("few VGPRs, 1 KB LDS, no scratch      ");
  run<64, 0, 0>("64 VGPRs                             ");
  run<96, 0, 0>("96 VGPRs                             ");
  run<128, 0, 0>("128 VGPRs                            ");
  run<168, 0, 0>("168 VGPRs                            ");
  run<256, 0, 0>("256 VGPRs                            ");
  run<0, 5496, 0>("few VGPRs, 22 KB LDS                 ");
  run<168, 5496, 0>("168 VGPRs, 22 KB LDS                 ");
  run<168, 5496, 1>("168 VGPRs, 22 KB LDS, 32 B scratch   ");
  run<0, 0, 1>("few VGPRs, 32 B scratch              ");
  run<128, 5496, 0>("128 VGPRs, 22 KB LDS                 ");
  return 0;
}
