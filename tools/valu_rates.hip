// Issue cost and dependent-issue latency of the VALU operations the reset / features / step kernels are made of, measured on
// gfx950 with 1, 2 and 4 wavefronts per SIMD (one-wavefront workgroups, grid = 1024 x {1, 2, 4}: the dispatcher spreads
// them evenly -- tools/reset_phases.py).  Each test: 2048 iterations x 16 instances of one instruction, either ONE dependent
// chain or FOUR independent ones; reports shader-clock cycles per instruction per wavefront (s_memtime).
//   build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/valu_rates tools/valu_rates.hip     run: tools/bin/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <string>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

template <int T>
__global__ __launch_bounds__(64) void k(unsigned long long* out, int iters) {
  __shared__ unsigned spread[10240 / 4];   // 16 workgroups per CU at most: 4096 one-wavefront workgroups land 4 per SIMD, evenly
  spread[threadIdx.x] = threadIdx.x;
  double a = threadIdx.x * 1e-3 + 1.0, b = 1.0000001, c = a + 1, d = a + 2, e = a + 3;
  float fa = threadIdx.x * 1e-3f + 1.5f, fb = 1.0000001f, fc = fa + 1, fd = fa + 2, fe = fa + 3;
  unsigned ua = threadIdx.x * 2654435761u + 1u, ub = 0x9E3779B9u, uc = ua + 7, ud = ua + 9, ue = ua + 11;
  unsigned long long w0 = ua, w1 = uc, w2 = ud, w3 = ue;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    if constexpr (T == 0) { REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(fa) : "v"(fb));) }
    if constexpr (T == 1) { REP4(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(fa), "+v"(fc), "+v"(fd), "+v"(fe) : "v"(fb));) }
    if constexpr (T == 2) { REP16(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b));) }
    if constexpr (T == 3) { REP4(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));) }
    if constexpr (T == 4) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(b));) }
    if constexpr (T == 5) { REP4(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));) }
    if constexpr (T == 6) { REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w0) : "v"((unsigned)w0), "v"(ub) : "vcc"); w0 ^= w0 >> 32;) }
    if constexpr (T == 7) { REP4(asm volatile("v_mad_u64_u32 %0, vcc, %4, %8, 0\n v_mad_u64_u32 %1, vcc, %5, %8, 0\n v_mad_u64_u32 %2, vcc, %6, %8, 0\n v_mad_u64_u32 %3, vcc, %7, %8, 0" : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3) : "v"((unsigned)w0), "v"((unsigned)w1), "v"((unsigned)w2), "v"((unsigned)w3), "v"(ub) : "vcc");) }
    if constexpr (T == 8) { REP16(asm volatile("v_min_f64 %0, %0, %1" : "+v"(a) : "v"(b));) }
    if constexpr (T == 9) { REP4(asm volatile("v_min_f64 %0, %0, %4\n v_max_f64 %1, %1, %4\n v_min_f64 %2, %2, %4\n v_max_f64 %3, %3, %4" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));) }
    if constexpr (T == 10) { REP16(asm volatile("v_xor_b32 %0, %0, %1" : "+v"(ua) : "v"(ub));) }
    if constexpr (T == 11) { REP16(asm volatile("v_log_f32 %0, %0" : "+v"(fa));) }
    if constexpr (T == 12) { REP4(asm volatile("v_log_f32 %0, %0\n v_sin_f32 %1, %1\n v_cos_f32 %2, %2\n v_sqrt_f32 %3, %3" : "+v"(fa), "+v"(fc), "+v"(fd), "+v"(fe));) }
    if constexpr (T == 13) { REP16(asm volatile("v_rcp_f64 %0, %0" : "+v"(a));) }
    if constexpr (T == 14) { REP4(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(a), "+v"(c), "+v"(d), "+v"(e));) }
    if constexpr (T == 15) { REP16(asm volatile("v_cvt_f64_f32 %0, %1\n v_cvt_f32_f64 %1, %0" : "+v"(a), "+v"(fa));) }   // 2 instructions per instance
    if constexpr (T == 16) { REP16(asm volatile("s_nop 1\n v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32 %0, %0, %1" : "+v"(fa), "+v"(fc));) }   // a scan stage (fp32): 2 VALU
    if constexpr (T == 17) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f32 %0, %0, %1\n s_or_b64 exec, exec, s[20:21]" : "+v"(fa) : "v"(fb) : "vcc", "s20", "s21");) }   // a divergent guard: 2 VALU + 2 SALU
    if constexpr (T == 18) { REP16(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));) }
    if constexpr (T == 19) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(ua) : "v"(ub));) }
    if constexpr (T == 20) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ua) : "v"(ub) : "vcc");) }
    if constexpr (T == 21) { REP16(asm volatile("v_readlane_b32 s10, %0, 63\n v_add_u32 %0, s10, %0" : "+v"(ua) : : "s10");) }   // VALU -> SGPR -> VALU
    if constexpr (T == 22) { REP4(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(ua), "+v"(uc), "+v"(ud), "+v"(ue) : "v"(ub));) }
    if constexpr (T == 24) { REP16(REP4(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(ua), "+v"(uc), "+v"(ud), "+v"(ue) : "v"(ub));)) }   // 256 x 4-byte encodings per iteration
    if constexpr (T == 25) { REP16(REP4(asm volatile("v_xor_b32_e64 %0, %0, %4\n v_xor_b32_e64 %1, %1, %4\n v_xor_b32_e64 %2, %2, %4\n v_xor_b32_e64 %3, %3, %4" : "+v"(ua), "+v"(uc), "+v"(ud), "+v"(ue) : "v"(ub));)) }   // 256 x 8-byte encodings
    if constexpr (T == 26) { REP16(REP16(asm volatile("v_xor_b32_e64 %0, %0, %4\n v_xor_b32_e64 %1, %1, %4\n v_xor_b32_e64 %2, %2, %4\n v_xor_b32_e64 %3, %3, %4" : "+v"(ua), "+v"(uc), "+v"(ud), "+v"(ue) : "v"(ub));)) }   // 1024 x 8 bytes = 8 KB of straight-line code
    if constexpr (T == 27) { REP16(REP16(asm volatile("v_add_f64 %0, %0, %4\n v_xor_b32_e64 %5, %5, %6\n v_add_f64 %2, %2, %4\n v_xor_b32_e64 %7, %7, %6" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(b), "+v"(ua), "+v"(ub), "+v"(uc));)) }   // 8 KB, half fp64
    if constexpr (T == 28) { REP16(asm volatile("s_add_u32 s20, s20, 1" : : : "s20", "scc");) }
    if constexpr (T == 29) { REP16(asm volatile("s_nop 0");) }
    if constexpr (T == 30) { REP16(asm volatile("v_xor_b32 %0, %0, %1\n s_add_u32 s20, s20, 1" : "+v"(ua) : "v"(ub) : "s20", "scc");) }   // 1 VALU + 1 SALU
    if constexpr (T == 31) { REP16(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n s_add_u32 s20, s20, 1" : "+v"(ua), "+v"(uc), "+v"(ud), "+v"(ue) : "v"(ub) : "s20", "scc");) }   // 4 VALU + 1 SALU
    if constexpr (T == 32) { REP16(asm volatile("s_mov_b32 s20, 0xcd9e8d57" : : : "s20");) }
    if constexpr (T == 33) { REP16(asm volatile("v_readfirstlane_b32 s20, %0\n v_xor_b32 %0, s20, %0" : "+v"(ua) : : "s20");) }   // 2 VALU
    if constexpr (T == 34) { REP16(asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");) }
    if constexpr (T == 35) { REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ua) : "v"(ub) : "vcc");) }   // 2 VALU through VCC
    if constexpr (T == 36) { REP16(asm volatile("v_cmp_lt_u32 s[20:21], %0, %1\n s_and_b64 s[20:21], s[20:21], exec\n v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(ua) : "v"(ub) : "s20", "s21", "scc");) }   // VALU -> SGPR -> SALU -> VALU
    if constexpr (T == 37) { REP16(asm volatile("v_add_f64 %0, %0, %1\n s_add_u32 s20, s20, 1" : "+v"(a) : "v"(b) : "s20", "scc");) }   // 1 fp64 VALU + 1 SALU
    if constexpr (T == 38) { REP16(asm volatile("s_cmp_lt_u32 s20, 7\n s_cselect_b32 s21, 1, 2\n s_add_u32 s20, s20, s21" : : : "s20", "s21", "scc");) }   // 3 SALU, dependent
    if constexpr (T == 39) { REP16(asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(ua) : "v"((threadIdx.x & 63) * 4));) }   // LDS round trip
    if constexpr (T == 23) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));) }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (spread[(threadIdx.x + 1) & 63] == 77u);
  if (a + c + d + e + fa + fc + fd + fe + ua + uc + ud + ue + (double)(w0 + w1 + w2 + w3) == 1.2345) out[0] = 0;
}

static const char* NAMES[] = {"v_add_f32 dependent", "v_add_f32 4 chains", "v_add_f64 dependent", "v_add_f64 4 chains", "v_fma_f64 dependent",
  "v_fma_f64 4 chains", "v_mad_u64_u32 + xor dependent (2 instr)", "v_mad_u64_u32 4 chains", "v_min_f64 dependent", "v_min/max_f64 4 chains",
  "v_xor_b32 dependent", "v_log_f32 dependent", "log/sin/cos/sqrt f32 4 chains", "v_rcp_f64 dependent", "v_rcp_f64 4 chains",
  "cvt f32->f64->f32 dependent (2 instr)", "dpp row_shr + add (2 VALU + s_nop)", "cmp + saveexec + add + restore (2 VALU + 2 SALU)",
  "v_mul_f64 dependent", "v_mul_lo_u32 dependent", "v_cndmask_b32 dependent", "readlane -> add (VALU->SGPR->VALU)", "v_xor_b32 4 chains",
  "v_pk_fma_f32 dependent", "v_xor_b32 4 chains, 256 per iteration (1 KB body)", "v_xor_b32_e64 4 chains, 256 per iteration (2 KB body)", "v_xor_b32_e64 4 chains, 1024 per iteration (8 KB body)", "add_f64 / xor_e64 mix, 1024 per iteration (8 KB body)", "s_add_u32 dependent", "s_nop 0", "v_xor + s_add (1 VALU + 1 SALU)", "4 v_xor + s_add (4 VALU + 1 SALU)", "s_mov_b32 literal", "readfirstlane -> xor (2 VALU via SGPR)", "s_waitcnt 0 (nothing outstanding)", "v_cmp vcc -> v_cndmask (2 VALU)", "v_cmp sgpr -> s_and -> v_cndmask (2 VALU + 1 SALU)", "v_add_f64 + s_add (1 VALU + 1 SALU)", "s_cmp + s_cselect + s_add (3 SALU)", "ds_read_b32 + wait (LDS round trip)"};

template <int T>
void run(unsigned long long* dev, int iters) {
  for (int wps : {1, 2, 4}) {
    const int grid = 1024 * wps;
    hipLaunchKernelGGL(k<T>, dim3(grid), dim3(64), 0, 0, dev, 64);
    hipLaunchKernelGGL(k<T>, dim3(grid), dim3(64), 0, 0, dev, (T >= 24 && T <= 27) ? iters / 16 : iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), dev, grid * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double inst = T == 24 || T == 25 ? 256 : (T == 26 || T == 27) ? 1024 : 16;
    const int it = (T >= 24 && T <= 27) ? iters / 16 : iters;
    const double per = (double)h[grid / 2] / ((double)it * inst);
    const double mx = (double)h[grid - 1] / ((double)it * inst);
    if (wps == 1) printf("%-52s", NAMES[T]);
    printf("  %dw: %6.1f (max %6.1f)", wps, per, mx);
  }
  printf("   cycles per instance per wavefront\n");
}

template <int T> void run_all(unsigned long long* dev, int iters) { run<T>(dev, iters); if constexpr (T + 1 < 40) run_all<T + 1>(dev, iters); }

int main() {
  unsigned long long* dev;
  hipMalloc(&dev, 8 * 4096);
  // s_memtime counts at a constant 100 MHz on this part; report both raw ticks and the wall time of one test to calibrate
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<0>, dim3(1024), dim3(64), 0, 0, dev, 2048);
  hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(k<0>, dim3(1024), dim3(64), 0, 0, dev, 20480); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(1024); hipMemcpy(h.data(), dev, 1024 * 8, hipMemcpyDeviceToHost);
  printf("calibration: %.1f us wall for %llu counter ticks -> %.1f ticks per us (the columns below are in these ticks)\n", ms * 1e3, h[512], h[512] / (ms * 1e3));
  run_all<0>(dev, 2048);
  return 0;
}
