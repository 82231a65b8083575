// sdc_device.hpp -- device-side state layout and shared device functions of the SustainDC step.
//
// Written for gfx950 (MI355X, CDNA4) only: 64-lane wavefronts.  Two kernels per timestep:
// sdc_dynamics_kernel -- one wavefront per environment instance integrates the coupled dynamics
// (lanes = racks for the IT model, wave shuffles for the rack reductions) and writes obs / info;
// sdc_reward_kernel -- one workgroup of 4 wavefronts per environment streams the env's 40 KB energy
// history ring from HBM (16 B per lane, coalesced) and holds it in VGPRs for the order statistics
// and the clipped mean / std of reward normalisation.
//
// Arithmetic: fp64 for the dynamics, observation features and reductions (the reference is Python
// float / NumPy float64, and its integer / decimal-rounding cliffs only reproduce in fp64);
// the history ring is stored fp32; obs / rewards / info are written fp32.
// Compiled with -ffp-contract=off so that a*b+c rounds twice, as in the reference.
//
// Reference citations are file:line under /root/reference.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sustaindc_hip.h"

#define SDC_BLOCK 256
#define SDC_WAVE 64
#define SDC_HIST_PER_THREAD 40  // 10 x float4 per thread -> 10240 ring slots per env
#define SDC_HIST_STRIDE (SDC_BLOCK * SDC_HIST_PER_THREAD)
#define SDC_NORM_WINDOW 2880    // 30 days x 96 (utils/managers.py:435, :606)
#define SDC_OBS_RAW 53
#define SDC_OBS_OUT (SDC_N_AGENTS * SDC_OBS_PAD)

// per-env carried load-shifting info (sustaindc_env.py:569-573): 8 doubles
enum { SDC_C_NORMQ = 0, SDC_C_OLDEST, SDC_C_AVG, SDC_C_H0, SDC_C_H1, SDC_C_H2, SDC_C_H3, SDC_C_H4, SDC_CARRY_DIM };

struct SdcDev {
  int n_envs, episode_steps, hist_cap, queue_max, table_len, lw, qstride, max_roll_days;
  unsigned long long seed;
  double noise_std, noise_weight;
  // shared, read-only
  const double* tabW;   // [n_loc][table_len]
  const double* tabC;
  const double* tabT;   // pre-noise dry bulb
  const double* tabWB;  // pre-noise wet bulb
  const sdc_dc_params* dc;  // [n_cfg]
  const double* hour_lut;   // [96][2] = cos, sin (utils/managers.py:66-88)
  // per-env assignment
  const int* loc_id;
  const int* cfg_id;
  const int* day_lo;
  const int* day_hi;
  // per-env state (struct of arrays)
  int* cursor;
  int* t_rel;
  int* day;
  int* hourq;
  int* q_popped;
  int* q_cum;
  unsigned* q_cumT;
  int* q_head;
  uint2* qtab;  // [N][qstride] {cum, cumT}
  int* last_delta;  // -2 = None
  int* consecutive;
  int* scale;
  int* hist_len;
  int* hist_pos;
  int* episode;
  unsigned* fault;
  double* stpt;
  double* bat_load;
  double* ci_min;
  double* ci_den;
  double* t_min;
  double* t_den;
  double* carry;  // [SDC_CARRY_DIM][N]
  double* t_win;  // [N][lw]
  double* wb_win;
  double* walk_tmp;  // [N][SDC_NORM_WINDOW] scratch of the device-side reset
  float* hist;       // [N][SDC_HIST_STRIDE]  energy - hist_ref, fp32
  double* hand;      // [4][N] dynamics -> reward kernel: energy, norm_CI next, oldest task age, overdue count
  unsigned* q_guess; // [2][N] fp32 keys of last step's order statistics at floor((n-1)/4), floor(3(n-1)/4)
  double* ep_return; // [3][N] running return of the current episode (cleared by reset)
  double* hist_ref;  // [N] first energy value of the env (fp64): the ring stores offsets from it
  unsigned char* reset_mask;  // [N] device copy of the caller's mask
};

// ------------------------------------------------------------------------------------------------
// wave helpers (64 lanes)

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}

// np.round(x, d) == rint(x * 10^d) / 10^d
__device__ __forceinline__ double np_round(double x, double p10) { return rint(x * p10) / p10; }

// ------------------------------------------------------------------------------------------------
// observation features (sustaindc_env.py:266-433).  Inputs are staged in LDS:
//   nc[0..24]  = NC[i'-16 .. i'+8]   (nc[16] = NC[i'];  entries for negative table indices unused)
//   nt[0..16]  = NT[i' .. i'+16]

__device__ __forceinline__ double lsq_slope(const double* y, int n) {
  // np.polyfit(range(n), y, 1)[0] as the closed-form least-squares slope
  const double xm = 0.5 * (double)(n - 1);
  double ym = 0.0;
  for (int i = 0; i < n; i++) ym += y[i];
  ym /= (double)n;
  double sxy = 0.0, sxx = 0.0;
  for (int i = 0; i < n; i++) {
    const double dx = (double)i - xm;
    sxy += dx * (y[i] - ym);
    sxx += dx * dx;
  }
  return sxy / sxx;
}

// NumPy's pairwise add.reduce for n in {8, 16} (8 accumulators, then a fixed tree) so that np.mean /
// np.std of the 8 CI futures and the 16 temperature futures round exactly as in the reference.
__device__ __forceinline__ double np_sum_8_16(const double* a, int n) {
  double r[8];
#pragma unroll
  for (int j = 0; j < 8; j++) r[j] = a[j];
  if (n == 16) {
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] += a[j + 8];
  }
  return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
}

// extract_ci_features(values[n], cur) -> mean, std, (cur-mean)/(std+1e-8), first peak / n, first valley / n
// n must be 8 or 16.
__device__ __forceinline__ void extract_features(const double* vals, int n, double cur, double* out5) {
  const double mean = np_sum_8_16(vals, n) / (double)n;
  double sq[16];
  for (int i = 0; i < n; i++) {
    const double d = vals[i] - mean;
    sq[i] = d * d;
  }
  const double sd = sqrt(np_sum_8_16(sq, n) / (double)n);
  // np.gradient of [cur, vals...]: one-sided ends, central interior
  int peak = n, valley = n;
  double xm1 = cur, x0 = cur, x1 = vals[0];
  double gprev = x1 - x0;  // g[0]
  for (int i = 1; i <= n; i++) {
    // g[i]
    xm1 = x0;
    x0 = x1;
    double g;
    if (i < n) {
      x1 = vals[i];
      g = (x1 - xm1) / 2.0;
    } else {
      g = x0 - xm1;
    }
    if (peak == n && gprev > 0 && g <= 0) peak = i - 1;
    if (valley == n && gprev < 0 && g >= 0) valley = i - 1;
    gprev = g;
  }
  out5[0] = mean;
  out5[1] = sd;
  out5[2] = (cur - mean) / (sd + 1e-8);
  out5[3] = (double)peak / (double)n;
  out5[4] = (double)valley / (double)n;
}

struct ObsScalars {
  double cos_h, sin_h;
  double w_cur, w_next;   // W[i'], W[i'+1]
  double soc;
  double normq, oldest, avg, hist[5];
  int have_past;          // i' >= 16
};

// Writes the 53 raw observation floats (ls 26 | dc 14 | bat 13) to s_obs (LDS).  Uniform over the wave;
// the caller guards the call with a single lane.
__device__ __forceinline__ void build_obs_raw(const double* nc, const double* nt, const ObsScalars& o, float* s_obs) {
  double sm[16], f7[7], tf[5];
  const double cur = nc[16];
  // future: 4-tap moving average of [cur, NC[i'+1..i'+8]] (9 -> 6 points), then slope (sustaindc_env.py:313,317)
  for (int j = 0; j < 6; j++) sm[j] = (((nc[16 + j] + nc[17 + j]) + nc[18 + j]) + nc[19 + j]) / 4;
  f7[0] = lsq_slope(sm, 6);
  // past: [NC[i'-16..i'-1], cur] (17 -> 14 points); EMPTY past slice when i' < 16 (managers.py:482-483)
  if (o.have_past) {
    for (int j = 0; j < 14; j++) sm[j] = (((nc[j] + nc[j + 1]) + nc[j + 2]) + nc[j + 3]) / 4;
    f7[1] = lsq_slope(sm, 14);
  } else {
    for (int j = 0; j < 4; j++) sm[j] = cur / 4;
    f7[1] = lsq_slope(sm, 4);
  }
  extract_features(nc + 17, 8, cur, f7 + 2);
  const double tslope = lsq_slope(nt, 17);
  extract_features(nt + 1, 16, nt[0], tf);
  int k = 0;
  // agent_ls (26): sustaindc_env.py:342-353
  s_obs[k++] = (float)o.cos_h; s_obs[k++] = (float)o.sin_h; s_obs[k++] = (float)cur;
  for (int j = 0; j < 7; j++) s_obs[k++] = (float)f7[j];
  s_obs[k++] = (float)o.oldest; s_obs[k++] = (float)o.avg; s_obs[k++] = (float)o.normq;
  s_obs[k++] = (float)o.w_cur; s_obs[k++] = (float)nt[0]; s_obs[k++] = (float)tslope;
  for (int j = 0; j < 5; j++) s_obs[k++] = (float)tf[j];
  for (int j = 0; j < 5; j++) s_obs[k++] = (float)o.hist[j];
  // agent_dc (14): sustaindc_env.py:386-393
  s_obs[k++] = (float)o.cos_h; s_obs[k++] = (float)o.sin_h; s_obs[k++] = (float)cur;
  for (int j = 0; j < 7; j++) s_obs[k++] = (float)f7[j];
  s_obs[k++] = (float)o.w_cur; s_obs[k++] = (float)o.w_next; s_obs[k++] = (float)nt[0]; s_obs[k++] = (float)nt[1];
  // agent_bat (13): sustaindc_env.py:426-432
  s_obs[k++] = (float)o.cos_h; s_obs[k++] = (float)o.sin_h; s_obs[k++] = (float)cur;
  for (int j = 0; j < 7; j++) s_obs[k++] = (float)f7[j];
  s_obs[k++] = (float)o.w_cur; s_obs[k++] = (float)nt[0]; s_obs[k++] = (float)o.soc;
}

// HARL layout (harlsustaindc_env.py:25-26, :78-80): obs [3][26] zero padded, share_obs [29].
// idx in [0, 78) -> value
__device__ __forceinline__ float obs_padded_at(const float* s_obs, int idx) {
  const int a = idx / SDC_OBS_PAD, k = idx - a * SDC_OBS_PAD;
  if (a == 0) return s_obs[k];
  if (a == 1) return k < 14 ? s_obs[26 + k] : 0.0f;
  return k < 13 ? s_obs[40 + k] : 0.0f;
}
__device__ __forceinline__ float share_obs_at(const float* s_obs, int idx) {
  if (idx < 26) return s_obs[idx];
  if (idx == 26) return s_obs[26 + 11];  // dc next workload
  if (idx == 27) return s_obs[26 + 13];  // dc next outside temperature
  return s_obs[40 + 12];                 // battery SoC
}

// stage the obs windows for table cursor ip (= i') into LDS.  tsrc points at T[i'] of the env's weather
// window (global memory, or LDS right after a device-side reset).  Called with tid = 0..63.
__device__ __forceinline__ void stage_windows(const SdcDev& S, int loc, int ip, const double* tsrc, double ci_min,
                                              double ci_den, double t_min, double t_den, int tid, double* s_nc,
                                              double* s_nt) {
  if (tid < 25) {
    int idx = ip - 16 + tid;
    idx = idx < 0 ? 0 : (idx >= S.table_len ? S.table_len - 1 : idx);
    const double c = S.tabC[(size_t)loc * S.table_len + idx];
    s_nc[tid] = (c - ci_min) / ci_den;  // managers.py:437
  } else if (tid >= 32 && tid < 49) {
    const int k = tid - 32;
    s_nt[k] = (tsrc[k] - t_min) / t_den;  // managers.py:608
  }
}

// ------------------------------------------------------------------------------------------------
// counter-based RNG for device-side resets: Philox4x32-10 (Salmon et al., SC'11)

struct Philox4 {
  unsigned x, y, z, w;
};
__device__ __forceinline__ Philox4 philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                                 unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
    const unsigned n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    const unsigned n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}
// uniform in (0, 1) from 2 x 32 bits (53-bit mantissa)
__device__ __forceinline__ double u01(unsigned hi, unsigned lo) {
  const unsigned long long b = (((unsigned long long)hi << 32) | lo) >> 11;
  return ((double)b + 0.5) * (1.0 / 9007199254740992.0);
}
