// Micro-benchmark: back-to-back dependent launches on one stream -- the floor under a one-launch-per-step design.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ __launch_bounds__(256, 3) void empty_k(float* p, int spin) {
  float a = threadIdx.x;
  for (int i = 0; i < spin; i++) a = a * 1.0001f + 0.5f;
  if (a == 12345.f) p[0] = a;
}
__global__ __launch_bounds__(256, 3) void touch_k(float* p, int n) {   // every wave writes one line
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = (float)i;
}
template <class F> double timeit(F f, int n) {
  for (int i = 0; i < 50; i++) f();
  hipDeviceSynchronize();
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int i = 0; i < n; i++) f();
  hipDeviceSynchronize();
  return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / n;
}
int main() {
  float* d; hipMalloc(&d, 64 << 20);
  for (int blocks : {1, 64, 544, 1024}) {
    printf("empty kernel, %4d blocks x 256: %.2f us per launch\n", blocks, timeit([&] { hipLaunchKernelGGL(empty_k, dim3(blocks), dim3(256), 0, 0, d, 0); }, 2000));
  }
  printf("spin ~10us kernel 544 blocks: %.2f us per launch\n", timeit([&] { hipLaunchKernelGGL(empty_k, dim3(544), dim3(256), 0, 0, d, 6000); }, 1000));
  printf("spin ~10us kernel 1 block: %.2f us per launch\n", timeit([&] { hipLaunchKernelGGL(empty_k, dim3(1), dim3(256), 0, 0, d, 6000); }, 1000));
  printf("touch 22 MB, 21504 blocks: %.2f us per launch\n", timeit([&] { hipLaunchKernelGGL(touch_k, dim3(21504), dim3(256), 0, 0, d, 21504 * 256); }, 1000));
  printf("touch 2 MB, 544 blocks: %.2f us per launch\n", timeit([&] { hipLaunchKernelGGL(touch_k, dim3(544), dim3(256), 0, 0, d, 544 * 256); }, 1000));
  return 0;
}
