// xcd_entry.hip -- when do the workgroups of BACK-TO-BACK launches of the lane-per-env kernel's shape (NB workgroups x 128 threads,
// 40 KB of LDS, > 168 VGPRs: four workgroups per CU) enter, by XCD (workgroup b runs on XCD b % 8)?  Variants: no memory traffic / every
// workgroup reads RD KB at entry / writes WR KB before it ends (plain, non-temporal, or system-scope = written through the L2: sc0 sc1), ~10 us of dependent arithmetic in between.
// hipcc --offload-arch=gfx950 -O3 -o xcd_entry xcd_entry.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f4v __attribute__((ext_vector_type(4)));
template <int NT>
__global__ __launch_bounds__(128) void shape(unsigned long long* out, float* buf, const int rd_kb, const int wr_kb, const int spin, const int slot) {
  __shared__ float pad[10240];
  const unsigned long long t0 = wall_clock64();
  asm volatile("v_mov_b32 v232, 0" ::: "v232");
  float a = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * (128 * 1024 / 4);      // 128 KB of the buffer per workgroup
  for (int k = 0; k < rd_kb * 1024 / 16 / 128; k++) {
    const f4v v = *reinterpret_cast<const f4v*>(buf + base + (size_t)(k * 128 + threadIdx.x) * 4);
    a += v.x + v.w;
  }
  const unsigned long long t1 = wall_clock64();
  for (int i = 0; i < spin; i++) a = a * 1.0001f + 0.5f;
  pad[threadIdx.x] = a;
  const unsigned long long t2 = wall_clock64();
  const f4v w = {a, 1.0f, 2.0f, 3.0f};
  for (int k = 0; k < wr_kb * 1024 / 16 / 128; k++) {
    f4v* p = reinterpret_cast<f4v*>(buf + base + (size_t)(k * 128 + threadIdx.x) * 4);
    if (NT == 1) __builtin_nontemporal_store(w, p);
    else if (NT == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(w) : "memory");      // system scope: written through the L2
    else if (NT == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(w) : "memory");
    else if (NT == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(w) : "memory");
    else *p = w;
  }
  const unsigned long long t3 = wall_clock64();
  if (threadIdx.x == 0) {
    unsigned long long* o = out + ((size_t)slot * gridDim.x + blockIdx.x) * 4;
    o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3;
  }
  if (a == 12345.f) buf[0] = pad[(threadIdx.x + 1) % 128];
}
int main(int argc, char** argv) {
  const int NB = argc > 1 ? atoi(argv[1]) : 640, L = 24;
  unsigned long long* d; float* buf;
  hipMalloc(&d, (size_t)L * NB * 4 * 8);
  hipMalloc(&buf, (size_t)NB * 128 * 1024);
  hipMemset(buf, 0, (size_t)NB * 128 * 1024);
  std::vector<unsigned long long> h((size_t)L * NB * 4);
  struct V { const char* name; int nt, rd, wr; } vs[] = {
    {"no traffic                      ", 0, 0, 0}, {"read 40 KB                      ", 0, 40, 0}, {"write 76 KB plain               ", 0, 0, 76},
    {"write 76 KB non-temporal        ", 1, 0, 76}, {"read 40 + write 76 KB plain     ", 0, 40, 76}, {"read 40 + write 76 KB nt        ", 1, 40, 76},
    {"read 40 + write 16 KB plain     ", 0, 40, 16},
    {"write 76 KB sc0 sc1 (write-thru)", 2, 0, 76}, {"read 40 + write 76 KB sc0 sc1   ", 2, 40, 76}, {"read 40 + write 76 KB sc1       ", 3, 40, 76},
    {"read 40 + write 76 KB sc0 sc1 nt", 4, 40, 76}};
  for (const V& v : vs) {
    for (int s = 0; s < L; s++) {
      if (v.nt == 1) hipLaunchKernelGGL(shape<1>, dim3(NB), dim3(128), 0, 0, d, buf, v.rd, v.wr, 3000, s);
      else if (v.nt == 2) hipLaunchKernelGGL(shape<2>, dim3(NB), dim3(128), 0, 0, d, buf, v.rd, v.wr, 3000, s);
      else if (v.nt == 3) hipLaunchKernelGGL(shape<3>, dim3(NB), dim3(128), 0, 0, d, buf, v.rd, v.wr, 3000, s);
      else if (v.nt == 4) hipLaunchKernelGGL(shape<4>, dim3(NB), dim3(128), 0, 0, d, buf, v.rd, v.wr, 3000, s);
      else hipLaunchKernelGGL(shape<0>, dim3(NB), dim3(128), 0, 0, d, buf, v.rd, v.wr, 3000, s);
    }
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double ent[8] = {0}, rdone[8] = {0}, wdone[8] = {0}, span = 0, period = 0;
    int n = 0;
    for (int s = 8; s < L; s++) {
      const unsigned long long* r = h.data() + (size_t)s * NB * 4;
      unsigned long long first = ~0ull, last = 0;
      for (int b = 0; b < NB; b++) { first = std::min(first, r[b * 4]); last = std::max(last, r[b * 4 + 3]); }
      for (int b = 0; b < NB; b++) {
        ent[b % 8] += (r[b * 4] - first) / 100.0; rdone[b % 8] += (r[b * 4 + 1] - r[b * 4]) / 100.0; wdone[b % 8] += (r[b * 4 + 3] - r[b * 4 + 2]) / 100.0;
      }
      span += (last - first) / 100.0;
      if (s > 8) {
        const unsigned long long* q = r - (size_t)NB * 4;
        unsigned long long pf = ~0ull;
        for (int b = 0; b < NB; b++) pf = std::min(pf, q[b * 4]);
        period += (first - pf) / 100.0;
      }
      n++;
    }
    printf("%s NB=%d: first entry -> last exit %.2f us, launch period %.2f us; entry after the launch's first, by b %% 8:", v.name, NB, span / n, period / (n - 1));
    for (int x = 0; x < 8; x++) printf(" %.2f", ent[x] / n / (NB / 8));
    printf(" | read phase:");
    for (int x = 0; x < 8; x++) printf(" %.2f", rdone[x] / n / (NB / 8));
    printf(" | write phase:");
    for (int x = 0; x < 8; x++) printf(" %.2f", wdone[x] / n / (NB / 8));
    printf("\n");
  }
  return 0;
}
