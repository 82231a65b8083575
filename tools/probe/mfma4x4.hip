// probe: operand layout of v_mfma_f32_4x4x1_16B_f32 with and without the A broadcast modifiers (development aid)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
  const int l = threadIdx.x;
  f4 z = {0.f, 0.f, 0.f, 0.f};
  // A = 1000 + lane, B = 1 -> D tells which lane's A lands in (lane, reg)
  f4 d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(1000.f + l, 1.0f, z, 0, 0, 0);
  // A = 1, B = 1000 + lane -> which lane's B
  f4 d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, 1000.f + l, z, 0, 0, 0);
  f4 d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(1000.f + l, 1.0f, z, 4, 5, 0);   // cbsz 4, abid 5
  f4 d3 = __builtin_amdgcn_mfma_f32_4x4x1f32(1000.f + l, 1.0f, z, 2, 1, 0);   // cbsz 2, abid 1
  for (int i = 0; i < 4; i++) { out[(0 * 64 + l) * 4 + i] = d0[i]; out[(1 * 64 + l) * 4 + i] = d1[i]; out[(2 * 64 + l) * 4 + i] = d2[i]; out[(3 * 64 + l) * 4 + i] = d3[i]; }
}
int main() {
  float* d; hipMalloc(&d, 4 * 64 * 4 * 4);
  probe<<<1, 64>>>(d);
  float h[4 * 64 * 4]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[4] = {"A=1000+lane (no bcast)", "B=1000+lane", "A, cbsz4 abid5", "A, cbsz2 abid1"};
  for (int t = 0; t < 4; t++) { printf("%s\n", nm[t]); for (int l = 0; l < 64; l += 1) { if (l % 8 == 0) printf("  "); printf("[%d:%g %g %g %g] ", l, h[(t*64+l)*4], h[(t*64+l)*4+1], h[(t*64+l)*4+2], h[(t*64+l)*4+3]); if (l % 8 == 7) printf("\n"); } }
  return 0;
}
