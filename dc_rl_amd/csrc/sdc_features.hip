// sdc_features.hip -- the trace-only part of the observations, once per EPISODE instead of once per step.
//
// 44 of the 53 observation floats (sustaindc_env.py:302-433) depend only on the traces and on the episode's start:
// the hour's cos / sin, NC[i'], the 7 carbon-intensity features, W[i'], W[i'+1], NT[i'], NT[i'+1], the temperature
// slope and its 5 features.  The step kernel used to compute them every step with all 64 lanes of a wavefront
// cooperating on windows of 6..17 points (~350 VALU instructions per env-step, a fifth of the step).  Here, right
// after an env's reset, ONE LANE PER STEP computes the same rows for the whole episode: 64 steps per instruction,
// ~75x fewer instructions in total.  The step kernel then loads its row (32 floats, one coalesced 128-byte access).
//
// The arithmetic follows the reference operation by operation (and so NumPy's reduction orders): the 4-tap moving
// averages as written at sustaindc_env.py:313-317, np.polyfit's slope as the closed-form least squares, np.mean /
// np.std as NumPy's pairwise add.reduce (for n = 8 / 16: eight accumulators, then ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))),
// np.gradient as central differences with one-sided ends, the first sign change of the gradient as a scan.
#include "sdc_device.hpp"

namespace {

// np.polyfit(range(n), y, 1)[0]
// (both quotients have compile-time divisors -- N and the sum of the squared abscissae, 5 / 17.5 / 227.5 / 408 for N = 4 / 6 / 14 / 17:
// the correctly rounded 3-instruction form, sdc_device.hpp sdc_div_const, checked for these constants by tests/aux/div_const_check.c,
// instead of two ~30-instruction IEEE division sequences per fit -- six per row, a sixth of the kernel's instructions)
template <int N>
constexpr double slope_sxx() {
  double s = 0.0;
  for (int i = 0; i < N; i++) s += ((double)i - 0.5 * (double)(N - 1)) * ((double)i - 0.5 * (double)(N - 1));
  return s;
}
template <int N>
__device__ __forceinline__ double slope_of(const double (&y)[N]) {
  const double xm = 0.5 * (double)(N - 1);
  double ym = 0.0;
#pragma unroll
  for (int i = 0; i < N; i++) ym += y[i];
  ym = SDC_DIV_CONST(ym, N);
  double sxy = 0.0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    const double dx = (double)i - xm;
    sxy += dx * (y[i] - ym);
  }
  constexpr double sxx = slope_sxx<N>();      // (the same sum in the same order as the run-time loop it replaces: exact in fp64)
  static_assert(N != 4 || sxx == 5.0, ""); static_assert(N != 6 || sxx == 17.5, ""); static_assert(N != 14 || sxx == 227.5, "");
  static_assert(N != 17 || sxx == 408.0, "");
  return sdc_div_const(sxy, sxx, 1.0 / sxx);
}
// NumPy's pairwise sum for n = 8 or 16 contiguous doubles
template <int N>
__device__ __forceinline__ double np_sum(const double (&a)[N]) {
  static_assert(N == 8 || N == 16, "written out for the two sizes the observations use");
  double r[8];
#pragma unroll
  for (int j = 0; j < 8; j++) r[j] = N == 16 ? a[j] + a[(8 + j) % N] : a[j];
  return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
}
// extract_ci_features (sustaindc_env.py:266-300) on N values following `cur`: mean, std, (cur - mean) / (std + 1e-8),
// first index at which the gradient of [cur, values] turns from > 0 to <= 0 (else N) / N, likewise < 0 to >= 0
template <int N>
__device__ __forceinline__ void extract_features(const double cur, const double (&v)[N], float* out5) {
  const double inv_n = 1.0 / (double)N;   // N is a power of two: exact
  const double mean = np_sum(v) * inv_n;
  double sq[N];
#pragma unroll
  for (int i = 0; i < N; i++) {
    const double d = v[i] - mean;
    sq[i] = d * d;
  }
  const double sd = sqrt(np_sum(sq) * inv_n);
  double g[N + 1];
  g[0] = v[0] - cur;
  g[1] = (v[1] - cur) / 2.0;
#pragma unroll
  for (int i = 2; i < N; i++) g[i] = (v[i] - v[i - 2]) / 2.0;
  g[N] = v[N - 1] - v[N - 2];
  int peak = N, valley = N;
#pragma unroll
  for (int i = N - 1; i >= 0; i--) {
    if (g[i] > 0 && g[i + 1] <= 0) peak = i;
    if (g[i] < 0 && g[i + 1] >= 0) valley = i;
  }
  out5[0] = (float)mean;
  out5[1] = (float)sd;
  out5[2] = (float)((cur - mean) / (sd + 1e-8));
  out5[3] = (float)((double)peak * inv_n);
  out5[4] = (float)((double)valley * inv_n);
}

}  // namespace

// One workgroup of W wavefronts (W = blockDim.x / 64: 4, or 1 when the episode's windows leave no room for four tiles) per
// env that has just been reset (record: episode step 0, no feature rows yet); the wavefronts share the windows and deal the
// episode's 64-row passes round.  Dynamic LDS: (steps + 25) + lw doubles -- the episode's normalised carbon-intensity and
// temperature windows and the carbon-intensity window's four-tap moving averages -- + one 8.4 KB row tile per wavefront.  (One wavefront per env, rounds 2-4: 19.5 KB each = two per
// SIMD, and the kernel's compute half is the latency of a wavefront's dependent fp64 chains: it scales with residency.)
extern "C" __global__ __launch_bounds__(4 * SDC_WAVE) void sdc_features_kernel(SdcDev S, const int use_sma) {
  extern __shared__ double lds[];
  const int env = blockIdx.x;
  const int lane = threadIdx.x & (SDC_WAVE - 1);
  const int wv = threadIdx.x / SDC_WAVE, n_wv = blockDim.x / SDC_WAVE;
  unsigned* recp = S.rec + (size_t)env * SDC_REC_DWORDS;
  const unsigned r = recp[lane];
  if (rec_i32(r, R_TREL) != 0 || rec_i32(r, R_FEAT_OK) != 0) return;
  const int steps = S.episode_steps, TL = S.table_len;
  const int c0 = rec_i32(r, R_CURSOR), loc = rec_i32(r, R_LOC), hq0 = rec_i32(r, R_HOURQ);
  const double ci_min = rec_f64(r, R_CI_MIN), ci_den = rec_f64(r, R_CI_DEN);
  const double t_min = rec_f64(r, R_T_MIN), t_den = rec_f64(r, R_T_DEN);
  double* ncw = lds;                  // ncw[j] = NC[c0 - 16 + j], j in [0, steps + 25)
  double* ntw = lds + (steps + 25);   // ntw[k] = NT[c0 + k],      k in [0, lw)
  // sma[m] = (((ncw[m] + ncw[m+1]) + ncw[m+2]) + ncw[m+3]) / 4, m in [0, steps + 22): the four-tap moving average of
  // sustaindc_env.py:313-317, the same expression a row used to evaluate 20 times over (6 future + 14 past points; the window
  // of row s+1 is the window of row s moved by one) -- once per position, by the whole workgroup
  // (use_sma = 0: episodes whose windows leave no room for it -- 30-day ones -- evaluate the expression per row as before)
  double* sma = ntw + S.lw;
  const int n_sma = use_sma ? steps + 22 : 0;
  // a pass's 64 rows are assembled in LDS (row stride 33 floats: conflict-free for one lane per row) and go out as
  // whole 128-byte rows, two per store instruction
  constexpr int TS = SDC_FEAT_ROW + 1;
  float* tile = reinterpret_cast<float*>(sma + n_sma) + wv * (SDC_WAVE * TS);
  const double* tC = S.tabC + (size_t)loc * TL;
  const double* tW = S.tabW + (size_t)loc * TL;
  const double* tw = S.t_win + (size_t)env * S.lw;
  auto tix = [&](int idx) { return idx < 0 ? 0 : (idx > TL - 1 ? TL - 1 : idx); };
  for (int j = threadIdx.x; j < steps + 25; j += blockDim.x) ncw[j] = (tC[tix(c0 - 16 + j)] - ci_min) / ci_den;   // managers.py:437
  for (int k = threadIdx.x; k < S.lw; k += blockDim.x) ntw[k] = (tw[k] - t_min) / t_den;                           // managers.py:608
  __syncthreads();
  for (int m = threadIdx.x; m < n_sma; m += blockDim.x) sma[m] = (((ncw[m] + ncw[m + 1]) + ncw[m + 2]) + ncw[m + 3]) / 4;
  __syncthreads();
  for (int s0 = wv * SDC_WAVE; s0 <= steps; s0 += n_wv * SDC_WAVE) {   // (the tile is the wavefront's own: wave-level syncs)
   const int s = s0 + lane;                         // row s: the observation at i' = c0 + s
   if (s <= steps) {
    const int ip = c0 + s;
    const double* nc = ncw + s;   // nc[0..24] = NC[i'-16 .. i'+8]
    const double* nt = ntw + s;   // nt[0..16] = NT[i' .. i'+16]  (the last rows of an episode read up to ntw[lw - 1])
    float* o = tile + lane * TS;
    const int hq = (hq0 + s) % 96;
    o[SDC_P_COS] = (float)S.hour_lut[2 * hq];
    o[SDC_P_SIN] = (float)S.hour_lut[2 * hq + 1];
    o[SDC_P_NC] = (float)nc[16];
    o[SDC_P_W] = (float)tW[tix(ip)];
    o[SDC_P_WNEXT] = (float)tW[tix(ip + 1)];
    const bool short_nt = s + 16 >= S.lw;   // never: lw = steps + 18
    o[SDC_P_NT] = (float)nt[0];
    o[SDC_P_NTNEXT] = (float)nt[short_nt ? 0 : 1];
    {  // future: 4-tap moving average of [NC[i'], NC[i'+1..i'+8]]: 9 -> 6 points (sustaindc_env.py:313, 317)
      double sm[6];
#pragma unroll
      for (int j = 0; j < 6; j++) sm[j] = use_sma ? sma[s + 16 + j] : (((nc[16 + j] + nc[17 + j]) + nc[18 + j]) + nc[19 + j]) / 4;
      o[SDC_P_CI7 + 0] = (float)slope_of(sm);
    }
    if (ip >= 16) {  // past: [NC[i'-16..i'-1], NC[i']]: 17 -> 14 points
      double sm[14];
#pragma unroll
      for (int j = 0; j < 14; j++) sm[j] = use_sma ? sma[s + j] : (((nc[j] + nc[j + 1]) + nc[j + 2]) + nc[j + 3]) / 4;
      o[SDC_P_CI7 + 1] = (float)slope_of(sm);
    } else {         // EMPTY past slice (utils/managers.py:482-483): np.convolve then yields 4 copies of NC[i'] / 4
      const double sm[4] = {nc[16] / 4, nc[16] / 4, nc[16] / 4, nc[16] / 4};
      o[SDC_P_CI7 + 1] = (float)slope_of(sm);
    }
    {
      double v[8];
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = nc[17 + j];
      extract_features(nc[16], v, o + SDC_P_CI7 + 2);
    }
    {  // temperature: slope of [NT[i'], NT[i'+1..i'+16]] (17 points, no smoothing; sustaindc_env.py:331) + features
      double y[17], v[16];
#pragma unroll
      for (int j = 0; j < 17; j++) y[j] = nt[j];
#pragma unroll
      for (int j = 0; j < 16; j++) v[j] = nt[1 + j];
      o[SDC_P_TSLOPE] = (float)slope_of(y);
      extract_features(nt[0], v, o + SDC_P_T5);
    }
    // norm_CI = NC[i'+1] of the reward (sustaindc_env.py:681), fp64, in the row's last two floats
    auto put_f64 = [&](const int slot, const double v) {
      o[slot] = __int_as_float(__double2loint(v));
      o[slot + 1] = __int_as_float(__double2hiint(v));
    };
    put_f64(SDC_FEAT_NCNEXT, nc[17]);
    // the inputs of the step that leads here (from episode step s - 1, table cursor i = i' - 1)
    const int sp = s > 0 ? s - 1 : 0;
    put_f64(SDC_FEAT_W, tW[tix(c0 + sp)]);
    put_f64(SDC_FEAT_C, tC[tix(c0 + sp)]);
    put_f64(SDC_FEAT_T, tw[sp]);
    put_f64(SDC_FEAT_WB, S.wb_win[(size_t)env * S.lw + sp]);
    o[SDC_FEAT_T1] = (float)tw[sp + 1];
   }
   wave_sync();
   const int n_rows = min(SDC_WAVE, steps + 1 - s0);
#pragma unroll 4
   for (int j = 0; j < SDC_WAVE / 2; j++) {
     const int rr = 2 * j + (lane >> 5), k = lane & 31;
     // (non-temporal: 353 MB that nothing reads before the episode's steps do, one row per step -- 138 -> 131 us)
     if (rr < n_rows) __builtin_nontemporal_store(tile[rr * TS + k], &S.feat[feat_row_offset(S, env, s0 + rr) + k]);
   }
   wave_sync();
  }
  if (threadIdx.x == R_FEAT_OK) recp[lane] = 1u;
}
