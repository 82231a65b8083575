// counter_calibration.hip -- known byte counts in the access patterns of the lane-per-env step kernel (sdc_wide.hip), one kernel per
// pattern, so that rocprofv3's FETCH_SIZE / WRITE_SIZE / TCC_EA0_* can be read against the truth (the guide
// calibrates FETCH_SIZE x 2 for wide coalesced streaming reads only).  Every kernel touches a 256 MiB buffer once per launch
// (>> the 32 MiB of L2); the program prints the bytes each launch REQUESTS and, for the scattered patterns, the bytes of the 32 / 64 /
// 128-byte blocks those requests touch.  Run under rocprofv3 --pmc ... --kernel-trace: tools/calib/run_calibration.sh.
// Build: hipcc --offload-arch=gfx950 -O3 -o counter_calibration counter_calibration.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((address_space(1))) const void* gptr;
typedef __attribute__((address_space(3))) void* lptr;
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr size_t BUF = 256ull << 20;

// (A) block loads by LDS-DMA: a workgroup of 64 lanes brings one contiguous 16 KB block in with 16 global_load_lds_dwordx4
__global__ __launch_bounds__(64) void cal_ldsdma_x4(const char* __restrict__ src, unsigned* __restrict__ sink) {
  __shared__ unsigned lds[4096];
  const char* g = src + (size_t)blockIdx.x * 16384;
#pragma unroll
  for (int k = 0; k < 16; k++)
    __builtin_amdgcn_global_load_lds((gptr)(g + k * 1024 + threadIdx.x * 16), (lptr)(lds + k * 256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (lds[threadIdx.x * 61 & 4095] == 0x12345u) sink[0] = 1u;
}
// (A') the same with 12 of every 16 chunks (the state records: 192 of 256 bytes)
__global__ __launch_bounds__(64) void cal_ldsdma_x4_12of16(const char* __restrict__ src, unsigned* __restrict__ sink) {
  __shared__ unsigned lds[4096];
  const char* g = src + (size_t)blockIdx.x * 16384;
#pragma unroll
  for (int k = 0; k < 16; k++)
    if ((threadIdx.x & 15) < 12)
      __builtin_amdgcn_global_load_lds((gptr)(g + k * 1024 + threadIdx.x * 16), (lptr)(lds + k * 256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (lds[threadIdx.x * 61 & 4095] == 0x12345u) sink[0] = 1u;
}
// (B) plain coalesced 16 bytes per lane (the guide's reference point)
__global__ __launch_bounds__(64) void cal_load_x4(const uint4* __restrict__ src, unsigned* __restrict__ sink) {
  const uint4* g = src + (size_t)blockIdx.x * 1024;
  unsigned acc = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) { const uint4 v = g[k * 64 + threadIdx.x]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345u) sink[0] = acc;
}
// (C) gathers: every lane ELEM bytes at a stride of STRIDE bytes (ELEM 4: a ring key / a record field; 16: a record chunk)
template <int ELEM, int STRIDE>
__global__ __launch_bounds__(64) void cal_gather(const char* __restrict__ src, unsigned* __restrict__ sink, const size_t n) {
  const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  unsigned acc;
  if (ELEM == 4) acc = *reinterpret_cast<const unsigned*>(src + i * STRIDE);
  else { const uint4 v = *reinterpret_cast<const uint4*>(src + i * STRIDE); acc = v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345u) sink[0] = acc;
}
// (D) the rank windows' gather: per env (lane) 16 keys of its 1 KB window block [64 positions][4 windows]: window w, positions
// cb_w .. cb_w + 3 -> dwords (cb_w + j) * 4 + w: four spans of 52 bytes
__global__ __launch_bounds__(64) void cal_window_gather(const unsigned* __restrict__ src, unsigned* __restrict__ sink) {
  const size_t env = (size_t)blockIdx.x * 64 + threadIdx.x;
  const unsigned* qw = src + env * 256;
  unsigned acc = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const int cb = (int)((env * 2654435761ull + w * 40503ull) % 60ull);
#pragma unroll
    for (int j = 0; j < 4; j++) acc ^= qw[(cb + j) * 4 + w];
  }
  if (acc == 0x12345u) sink[0] = acc;
}
// (H) 64 consecutive dwords per wavefront, the wavefronts' rows far apart (the queue table's time-major mirror: five rows per step)
__global__ __launch_bounds__(64) void cal_row_probe(const unsigned* __restrict__ src, unsigned* __restrict__ sink, const int rows, const size_t row_stride) {
  unsigned acc = 0;
  for (int r = 0; r < rows; r++) acc ^= src[(size_t)r * row_stride + (size_t)blockIdx.x * 64 + threadIdx.x];
  if (acc == 0x12345u) sink[0] = acc;
}
// (E) whole-line non-temporal 16-byte stores
__global__ __launch_bounds__(64) void cal_store_nt_x4(float* __restrict__ dst) {
  float* g = dst + (size_t)blockIdx.x * 4096;
  const f4v v = {1.0f, 2.0f, 3.0f, (float)blockIdx.x};
#pragma unroll
  for (int k = 0; k < 16; k++) __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(g + 4 * (k * 64 + threadIdx.x)));
}
// (E') plain 16-byte stores of 12 of every 16 chunks (the state records going back: 192 of 256 bytes)
__global__ __launch_bounds__(64) void cal_store_x4_12of16(uint4* __restrict__ dst) {
  uint4* g = dst + (size_t)blockIdx.x * 1024;
  const uint4 v = make_uint4(1u, 2u, 3u, blockIdx.x);
#pragma unroll
  for (int k = 0; k < 16; k++)
    if ((threadIdx.x & 15) < 12) g[k * 64 + threadIdx.x] = v;
}
// (E'') plain 16-byte whole-line stores (the headers going back)
__global__ __launch_bounds__(64) void cal_store_x4(uint4* __restrict__ dst) {
  uint4* g = dst + (size_t)blockIdx.x * 1024;
  const uint4 v = make_uint4(1u, 2u, 3u, blockIdx.x);
#pragma unroll
  for (int k = 0; k < 16; k++) g[k * 64 + threadIdx.x] = v;
}
// (F) scattered stores: every lane ELEM bytes at a stride of STRIDE bytes (4: a ring key; 8: a queue-table entry)
template <int ELEM, int STRIDE>
__global__ __launch_bounds__(64) void cal_scatter(char* __restrict__ dst, const size_t n) {
  const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  if (ELEM == 4) *reinterpret_cast<unsigned*>(dst + i * STRIDE) = (unsigned)i;
  else *reinterpret_cast<uint2*>(dst + i * STRIDE) = make_uint2((unsigned)i, 7u);
}
// (G) coalesced 4-byte stores (64 consecutive dwords per wavefront: the mirror's append, `done`-like narrow rows)
__global__ __launch_bounds__(64) void cal_store4_coalesced(unsigned* __restrict__ dst) {
  dst[(size_t)blockIdx.x * 64 + threadIdx.x] = blockIdx.x;
}
// (I) 12-byte rows stored as three 4-byte stores per lane (the rewards [N][3])
__global__ __launch_bounds__(64) void cal_store_rows12(float* __restrict__ dst) {
  float* r = dst + ((size_t)blockIdx.x * 64 + threadIdx.x) * 3;
  r[0] = 1.0f; r[1] = 2.0f; r[2] = (float)blockIdx.x;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 3;
  char* buf;
  unsigned* sink;
  CHECK(hipMalloc(&buf, BUF));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(buf, 0, BUF));
  CHECK(hipMemset(sink, 0, 64));
  const unsigned wg16k = (unsigned)(BUF / 16384);
  printf("# pattern kernel requested_bytes_per_launch touched_32B touched_64B touched_128B\n");
  auto row = [&](const char* pat, const char* k, double req, double t32, double t64, double t128) {
    printf("%s %s %.0f %.0f %.0f %.0f\n", pat, k, req, t32, t64, t128);
  };
  for (int r = 0; r < reps; r++) {
    // the buffer leaves the caches between patterns: a different 256 MiB is not available, so flush by a pass of plain stores
    hipLaunchKernelGGL(cal_ldsdma_x4, dim3(wg16k), dim3(64), 0, 0, buf, sink);
    hipLaunchKernelGGL(cal_ldsdma_x4_12of16, dim3(wg16k), dim3(64), 0, 0, buf, sink);
    hipLaunchKernelGGL(cal_load_x4, dim3(wg16k), dim3(64), 0, 0, reinterpret_cast<const uint4*>(buf), sink);
    hipLaunchKernelGGL((cal_gather<4, 256>), dim3((unsigned)(BUF / 256 / 64)), dim3(64), 0, 0, buf, sink, BUF / 256);
    hipLaunchKernelGGL((cal_gather<16, 256>), dim3((unsigned)(BUF / 256 / 64)), dim3(64), 0, 0, buf, sink, BUF / 256);
    hipLaunchKernelGGL((cal_gather<4, 40960>), dim3((unsigned)((BUF / 40960 + 63) / 64)), dim3(64), 0, 0, buf, sink, BUF / 40960);
    hipLaunchKernelGGL(cal_window_gather, dim3((unsigned)(BUF / 1024 / 64)), dim3(64), 0, 0, reinterpret_cast<const unsigned*>(buf), sink);
    hipLaunchKernelGGL(cal_row_probe, dim3(512), dim3(64), 0, 0, reinterpret_cast<const unsigned*>(buf), sink, 512, (size_t)32768);
    hipLaunchKernelGGL(cal_store_nt_x4, dim3(wg16k), dim3(64), 0, 0, reinterpret_cast<float*>(buf));
    hipLaunchKernelGGL(cal_store_x4, dim3(wg16k), dim3(64), 0, 0, reinterpret_cast<uint4*>(buf));
    hipLaunchKernelGGL(cal_store_x4_12of16, dim3(wg16k), dim3(64), 0, 0, reinterpret_cast<uint4*>(buf));
    hipLaunchKernelGGL((cal_scatter<4, 40960>), dim3((unsigned)((BUF / 40960 + 63) / 64)), dim3(64), 0, 0, buf, BUF / 40960);
    hipLaunchKernelGGL((cal_scatter<4, 256>), dim3((unsigned)(BUF / 256 / 64)), dim3(64), 0, 0, buf, BUF / 256);
    hipLaunchKernelGGL((cal_scatter<8, 5376>), dim3((unsigned)((BUF / 5376 + 63) / 64)), dim3(64), 0, 0, buf, BUF / 5376);
    hipLaunchKernelGGL(cal_store4_coalesced, dim3((unsigned)(BUF / 256)), dim3(64), 0, 0, reinterpret_cast<unsigned*>(buf));
    hipLaunchKernelGGL(cal_store_rows12, dim3((unsigned)(BUF / 768 / 2)), dim3(64), 0, 0, reinterpret_cast<float*>(buf));
    CHECK(hipDeviceSynchronize());
  }
  const double B = (double)BUF;
  row("A_block_load_lds_dma_dwordx4", "cal_ldsdma_x4", B, B, B, B);
  row("A2_block_load_lds_dma_12_of_16_chunks", "cal_ldsdma_x4_12of16", B * 0.75, B * 0.75, B * 0.75, B);
  row("B_coalesced_load_dwordx4", "cal_load_x4", B, B, B, B);
  row("C1_gather_4B_stride_256B", "cal_gather<4, 256>", B / 256 * 4, B / 256 * 32, B / 256 * 64, B / 256 * 128);
  row("C2_gather_16B_stride_256B", "cal_gather<16, 256>", B / 256 * 16, B / 256 * 32, B / 256 * 64, B / 256 * 128);
  row("C3_gather_4B_stride_40KB", "cal_gather<4, 40960>", (double)(BUF / 40960) * 4, (double)(BUF / 40960) * 32, (double)(BUF / 40960) * 64, (double)(BUF / 40960) * 128);
  {
    // the blocks the window gather touches, counted exactly
    double t32 = 0, t64 = 0, t128 = 0;
    const size_t envs = BUF / 1024;
    for (size_t env = 0; env < envs; env++) {
      unsigned char m32[32] = {0};
      for (int w = 0; w < 4; w++) {
        const int cb = (int)((env * 2654435761ull + w * 40503ull) % 60ull);
        for (int j = 0; j < 4; j++) m32[(((cb + j) * 4 + w) * 4) / 32] = 1;
      }
      int c32 = 0, c64 = 0, c128 = 0;
      for (int i = 0; i < 32; i++) c32 += m32[i];
      for (int i = 0; i < 32; i += 2) c64 += (m32[i] | m32[i + 1]);
      for (int i = 0; i < 32; i += 4) c128 += (m32[i] | m32[i + 1] | m32[i + 2] | m32[i + 3]);
      t32 += c32 * 32; t64 += c64 * 64; t128 += c128 * 128;
    }
    row("D_window_gather_16_keys_per_env", "cal_window_gather", (double)envs * 64, t32, t64, t128);
  }
  row("H_row_probe_64_dwords_per_wave", "cal_row_probe", 512.0 * 512 * 256, 512.0 * 512 * 256, 512.0 * 512 * 256, 512.0 * 512 * 256);
  row("E_store_nontemporal_dwordx4", "cal_store_nt_x4", B, B, B, B);
  row("E2_store_dwordx4", "cal_store_x4(", B, B, B, B);
  row("E3_store_dwordx4_12_of_16_chunks", "cal_store_x4_12of16", B * 0.75, B * 0.75, B * 0.75, B);
  row("F1_scatter_4B_stride_40KB", "cal_scatter<4, 40960>", (double)(BUF / 40960) * 4, (double)(BUF / 40960) * 32, (double)(BUF / 40960) * 64, (double)(BUF / 40960) * 128);
  row("F2_scatter_4B_stride_256B", "cal_scatter<4, 256>", B / 256 * 4, B / 256 * 32, B / 256 * 64, B / 256 * 128);
  row("F3_scatter_8B_stride_5376B", "cal_scatter<8, 5376>", (double)(BUF / 5376) * 8, (double)(BUF / 5376) * 32, (double)(BUF / 5376) * 64, (double)(BUF / 5376) * 128);
  row("G_store_4B_coalesced", "cal_store4_coalesced", B, B, B, B);
  row("I_store_12B_rows_3x4B", "cal_store_rows12", (double)(BUF / 768 / 2) * 64 * 12, (double)(BUF / 768 / 2) * 64 * 12, (double)(BUF / 768 / 2) * 64 * 12, (double)(BUF / 768 / 2) * 64 * 12);
  return 0;
}
