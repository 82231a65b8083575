// probe: issue rate of v_mfma_f32_4x4x1_16B_f32 / v_mfma_f32_16x16x4_f32, alone and with a second wavefront on the SIMD (development aid)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void rate(float* out, unsigned long long* cyc, int iters) {
  const int l = threadIdx.x;
  f4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0;
  float x = 1.0f + l, y = 0.5f * l;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {        // 6 independent accumulators, 4x4x1
      a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 4, 1, 0); a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a1, 4, 2, 0);
      a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a2, 4, 3, 0); a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a3, 4, 4, 0);
      a4 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a4, 4, 5, 0); a5 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a5, 4, 6, 0);
    } else if (MODE == 1) { // 3 accumulators round robin (as the kernel)
      a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 4, 1, 0); a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a1, 4, 2, 0);
      a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a2, 4, 3, 0); a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 4, 4, 0);
      a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a1, 4, 5, 0); a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a2, 4, 6, 0);
    } else if (MODE == 2) { // 1 accumulator: dependent chain
      a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 4, 1, 0); a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 4, 2, 0);
      a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 4, 3, 0); a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 4, 4, 0);
      a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 4, 5, 0); a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 4, 6, 0);
    } else if (MODE == 3) { // 16x16x4, 6 independent
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
      a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a4, 0, 0, 0); a5 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a5, 0, 0, 0);
    } else if (MODE == 4) { // 4x4x1 without the broadcast
      a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a3, 0, 0, 0);
      a4 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a4, 0, 0, 0); a5 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a5, 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[0] + a2[0] + a3[0] + a4[0] + a5[0];
  if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char* nm, int threads) {
  float* d; unsigned long long* c; (void)hipMalloc(&d, 4 * 1024 * 4); (void)hipMalloc(&c, 8);
  const int iters = 2000;
  rate<MODE><<<1, threads>>>(d, c, iters); (void)hipDeviceSynchronize();
  rate<MODE><<<1, threads>>>(d, c, iters); (void)hipDeviceSynchronize();
  unsigned long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("%-44s %d wavefronts/WG: %.2f cycles per MFMA per wavefront\n", nm, threads / 64, (double)h / (iters * 6.0));
}
int main() {
  for (int th : {64, 256, 512}) {
    run<0>("4x4x1 cbsz, 6 independent accumulators", th);
    run<1>("4x4x1 cbsz, 3 accumulators round robin", th);
    run<2>("4x4x1 cbsz, 1 accumulator (dependent)", th);
    run<4>("4x4x1 no broadcast, 6 independent", th);
    run<3>("16x16x4, 6 independent", th);
  }
  return 0;
}
