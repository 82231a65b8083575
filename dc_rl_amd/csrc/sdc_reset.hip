// sdc_reset.hip -- SustainDC.reset() for the masked environments, one wavefront per environment.
//
// Two modes:
//   * override: (day, hour, CI / temperature normalisation bounds, weather windows) were injected by the
//     caller (parity tests feed what the reference's managers produced);
//   * device:   the draws of sustaindc_env.py:454-455 and utils/managers.py:35-48, :596-613 are made on the
//     GPU with a counter-based RNG (Philox4x32-10): start day / hour, the 0..13-day roll, and the
//     35 040-step Gaussian random walk (fp32 Box-Muller, fp64 accumulation) scaled to std 0.75 that is added
//     to dry and wet bulb before
//     the roll, clip to [0, 45] and 30-day min/max normalisation.  Same distributions as the reference,
//     not the same stream (the reference uses MT19937 via `random` and `np.random`).
//
// What reset clears / keeps follows the reference: queue, battery SoC and the set-point scaling
// counters are cleared (carbon_ls.py:85, battery_model.py:90-91, dc_gym.py:114-116); the CRAC set-point
// and the energy history survive (dc_gym.py:91-140 never touches raw_curr_stpt; reward_creator.py:5).
#include "sdc_device.hpp"

namespace {

struct ResetShared {
  double nc[32];
  double nt[32];
  double tw[32];  // T[c0 .. c0+16] of a device-side reset
  float obs[64];
  unsigned rec[SDC_REC_DWORDS];
};

// inclusive prefix sum over the 64 lanes on the DPP data path (no LDS round trips: this scan runs 137 times per reset):
// row_shr 1, 2, 4, 8 scan each row of 16, row_bcast15 / row_bcast31 carry the row totals upwards
__device__ __forceinline__ double wave_incl_scan_f64(double v, int) {
  v += dpp_f64<0x111>(v);
  v += dpp_f64<0x112>(v);
  v += dpp_f64<0x114>(v);
  v += dpp_f64<0x118>(v);
  v += dpp_f64<SDC_DPP_BCAST15, 0xA>(v);
  v += dpp_f64<SDC_DPP_BCAST31, 0xC>(v);
  return v;
}

// four standard normals per Philox4x32-10 block: two Box-Muller pairs in fp32 (hardware log2 / sin / cos); the
// random walk itself is accumulated in fp64.  Block c of (env, episode) yields the normals of samples 4c .. 4c+3.
__device__ __forceinline__ void normals4(const SdcDev& S, int env, int episode, int c, float (&nz)[4]) {
  const Philox4 r = philox4x32_10((unsigned)c, (unsigned)(S.env_base + env), (unsigned)episode, 0x7E47u, (unsigned)S.seed,
                                  (unsigned)(S.seed >> 32));
  const float k24 = 1.0f / 16777216.0f;
  const float u1 = ((float)(r.x >> 8) + 0.5f) * k24, u2 = ((float)(r.y >> 8) + 0.5f) * k24;   // (0, 1)
  const float u3 = ((float)(r.z >> 8) + 0.5f) * k24, u4 = ((float)(r.w >> 8) + 0.5f) * k24;
  const float r1 = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // sqrt(-2 ln u) = sqrt(-2 ln2 log2 u)
  const float r2 = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u3));
  nz[0] = r1 * __builtin_amdgcn_cosf(u2);   // v_cos_f32 / v_sin_f32 take revolutions: cos(2 pi u)
  nz[1] = r1 * __builtin_amdgcn_sinf(u2);
  nz[2] = r2 * __builtin_amdgcn_cosf(u4);
  nz[3] = r2 * __builtin_amdgcn_sinf(u4);
}

}  // namespace

extern "C" __global__ __launch_bounds__(SDC_WAVE) void sdc_reset_kernel(SdcDev S, int use_override,
                                                                          const int* __restrict__ ovr_day,
                                                                          const int* __restrict__ ovr_hour,
                                                                          const double* __restrict__ ovr_ci_min,
                                                                          const double* __restrict__ ovr_ci_max,
                                                                          const double* __restrict__ ovr_t_min,
                                                                          const double* __restrict__ ovr_t_max,
                                                                          int only_done, float* __restrict__ obs,
                                                                          float* __restrict__ share_obs,
                                                                          const double* __restrict__ inj_noise,
                                                                          const int* __restrict__ inj_roll) {
  __shared__ ResetShared sh;
  const int env = blockIdx.x;
  const int lane = threadIdx.x;
  if (lane == 0) {   // measurement: workgroups that return early still bracket the launch
    prof_stamp(S, SDC_PROF_RESET, env, 0);
    prof_stamp(S, SDC_PROF_RESET, env, 1);
  }
  if (S.reset_mask && !S.reset_mask[env]) return;
  unsigned* recp = S.rec + (size_t)env * SDC_REC_DWORDS;
  const unsigned r = recp[lane];  // the env's state record, one dword per lane
  if (only_done && rec_i32(r, R_TREL) < S.episode_steps) return;  // auto-reset: finished envs only
  const int loc = rec_i32(r, R_LOC);
  const int TL = S.table_len;
  const int episode = rec_i32(r, R_EPISODE) + 1;
  unsigned fault = (unsigned)rec_i32(r, R_FAULT);
  int day, hour;
  double ci_min, ci_den, t_min, t_den;
  const double* tsrc;
  if (use_override == 1) {
    day = ovr_day[env];
    hour = ovr_hour[env];
    ci_min = ovr_ci_min[env];
    ci_den = ovr_ci_max[env] - ci_min;
    t_min = ovr_t_min[env];
    t_den = ovr_t_max[env] - t_min;
    tsrc = S.t_win + (size_t)env * S.lw;  // copied in by the host before this launch
    if (day * 96 + hour * 4 + S.episode_steps + 17 > TL - 1) fault |= SDC_FAULT_TABLE_RANGE;
  } else {
    // ---- draws: sustaindc_env.py:454-455 (day in [lo, hi], hour in [0, 23]); managers.py:601 (roll) ----
    // (use_override == 2: day, hour, roll and the year's noise array come from the caller -- what the reference's
    // reset drew -- and only the arithmetic below runs here: add, roll, clip, 30-day min / max)
    const bool injected = use_override == 2;
    const Philox4 px = philox4x32_10(0u, (unsigned)(S.env_base + env), (unsigned)episode, 0xD4A7u, (unsigned)S.seed,
                                     (unsigned)(S.seed >> 32));
    const int lo = rec_i32(r, R_DAY_LO), hi = rec_i32(r, R_DAY_HI);
    day = injected ? ovr_day[env] : lo + (int)(((unsigned long long)px.x * (unsigned)(hi - lo + 1)) >> 32);
    hour = injected ? ovr_hour[env] : (int)(((unsigned long long)px.y * 24u) >> 32);
    const int roll_days = injected ? inj_roll[env]
                                   : (S.max_roll_days > 0 ? (int)(((unsigned long long)px.z * (unsigned)S.max_roll_days) >> 32) : 0);
    const double* inj = injected ? inj_noise + (size_t)env * TL : nullptr;
    // year-end fence: the reference reads table[cursor + 1 .. + 17] and raises IndexError at the end of the
    // year; keep the whole episode inside the table instead
    {
      const int last_ok = TL - 1 - (S.episode_steps + 17);
      if (day * 96 + hour * 4 > last_ok) {
        const int c = last_ok < 0 ? 0 : last_ok;
        day = c / 96;
        hour = (c - day * 96) / 4;
      }
    }
    const int c0 = day * 96 + hour * 4;
    const int shift = roll_days * 96;
    const int wlen = min(max(SDC_NORM_WINDOW, S.lw), TL - c0 + 0);  // samples of the rolled table we may touch
    double* walk = S.walk_tmp + (size_t)env * max(SDC_NORM_WINDOW, S.lw);
    double walk_std = 0.0;
    if (S.noise_std > 0.0 && !injected) {
      // pass 1: CoherentNoise.generate (managers.py:35-48): random walk, its population std.
      // 256 samples per iteration: 4 per lane (one Philox block), lane-local prefix + wave scan of the lane totals.
      double carry = 0.0, sum = 0.0, sumsq = 0.0;
      for (int base = 0; base < TL; base += 4 * SDC_WAVE) {
        float nz[4];
        normals4(S, env, episode, (base >> 2) + lane, nz);
        const int j0 = base + 4 * lane;
        double p[4];
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          acc += (j0 + k < TL) ? S.noise_weight * (double)nz[k] : 0.0;
          p[k] = acc;
        }
        const double incl = wave_incl_scan_f64(acc, lane);
        const double off = carry + (incl - acc);
        carry += readlane_f64(incl, 63);
        // does this block of 256 samples reach the window the episode reads?  (wave-uniform: one env per wavefront; the
        // window is wlen of the year's 35 040 samples -- most of the 137 blocks skip the per-sample index arithmetic)
        int blk0 = base + shift;                         // rolled position of the block's first sample (np.roll, managers.py:602)
        if (blk0 >= TL) blk0 -= TL;
        const int blk1 = blk0 + 4 * SDC_WAVE;            // (may run past TL: the part beyond wraps to [0, blk1 - TL))
        const bool hit = (blk0 < c0 + wlen && blk1 > c0) || (blk1 > TL && c0 < blk1 - TL);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int j = j0 + k;
          if (j < TL) {
            const double w = off + p[k];
            sum += w;
            sumsq += w * w;
            if (hit) {
              // base index j lands at rolled position (j + shift) mod TL
              int pos = j + shift;
              if (pos >= TL) pos -= TL;
              const int rel = pos - c0;
              if (rel >= 0 && rel < wlen) walk[rel] = w;
            }
          }
        }
      }
      sum = wave_sum_f64(sum);
      sumsq = wave_sum_f64(sumsq);
      const double mean = sum / (double)TL;
      const double var = sumsq / (double)TL - mean * mean;
      walk_std = sqrt(var);
    }
    __threadfence();  // walk[] is re-read by other lanes of this wave below
    __syncthreads();
    // pass 2: add noise, roll, clip [0, 45]; 30-day min / max from the cursor (managers.py:598-613)
    const double* tT = S.tabT + (size_t)loc * TL;
    const double* tWB = S.tabWB + (size_t)loc * TL;
    const double* tC = S.tabC + (size_t)loc * TL;
    double tmin = 1e300, tmax = -1e300, cmin = 1e300, cmax = -1e300;
    double* tw = S.t_win + (size_t)env * S.lw;
    double* wbw = S.wb_win + (size_t)env * S.lw;
    for (int rel = lane; rel < wlen; rel += SDC_WAVE) {
      int j = c0 + rel - shift;
      if (j < 0) j += TL;
      const double nz = injected ? inj[j] : (walk_std > 0.0 ? (walk[rel] / walk_std) * S.noise_std : 0.0);  // managers.py:47
      const double t = fmin(fmax(tT[j] + nz, 0.0), 45.0);
      if (rel < SDC_NORM_WINDOW) {
        tmin = fmin(tmin, t);
        tmax = fmax(tmax, t);
        const double c = tC[c0 + rel];
        cmin = fmin(cmin, c);
        cmax = fmax(cmax, c);
      }
      if (rel < S.lw) {
        tw[rel] = t;
        wbw[rel] = fmin(fmax(tWB[j] + nz, 0.0), 45.0);
      }
      if (rel < 17) sh.tw[rel] = t;
    }
    tmin = wave_min_f64(tmin);
    tmax = wave_max_f64(tmax);
    cmin = wave_min_f64(cmin);
    cmax = wave_max_f64(cmax);
    ci_min = cmin;
    ci_den = cmax - cmin;
    t_min = tmin;
    t_den = tmax - tmin;
    tsrc = sh.tw;
  }
  const int c0 = day * 96 + hour * 4;  // utils/managers.py:122
  __syncthreads();
  stage_windows(S, loc, c0, tsrc, ci_min, ci_den, t_min, t_den, lane, sh.nc, sh.nt);
  __syncthreads();
  {
    const double* tW = S.tabW + (size_t)loc * TL;
    auto tix = [&](int idx) { return idx < 0 ? 0 : (idx > TL - 1 ? TL - 1 : idx); };
    ObsScalars o;
    const int hq = hour * 4;
    o.cos_h = S.hour_lut[2 * hq];
    o.sin_h = S.hour_lut[2 * hq + 1];
    o.w_cur = tW[tix(c0)];
    o.w_next = tW[tix(c0 + 1)];
    o.soc = 0.0;
    o.normq = 0.0;
    o.oldest = 0.0;
    o.avg = 0.0;
    for (int b = 0; b < 5; b++) o.hist[b] = 0.0;
    o.have_past = c0 >= 16;
    build_obs_pool(sh.nc, sh.nt, o, sh.obs, lane);
  }
  sh.rec[lane] = r;
  __syncthreads();
  if (lane == 0) {
    unsigned* o = sh.rec;
    auto put64 = [&](int idx, double v) {
      o[idx] = (unsigned)__double2loint(v);
      o[idx + 1] = (unsigned)__double2hiint(v);
    };
    put64(R_CI_MIN, ci_min);
    put64(R_CI_DEN, ci_den);
    put64(R_T_MIN, t_min);
    put64(R_T_DEN, t_den);
    put64(R_BAT, 0.0);                 // battery_model.py:90-91
    o[R_CURSOR] = (unsigned)c0;
    o[R_TREL] = 0u;
    o[R_FEAT_OK] = 0u;                 // the features kernel that follows fills the episode's rows
    o[R_DAY] = (unsigned)day;
    o[R_HOURQ] = (unsigned)(hour * 4);
    o[R_QPOPPED] = 0u;                 // carbon_ls.py:85
    o[R_QCUM] = 0u;
    o[R_QCUMT] = 0u;
    o[R_QHEAD] = 0u;
    o[R_QCUM_HM1] = 0u;
    o[R_QCUMT_HM1] = 0u;
    o[R_LAST_DELTA] = (unsigned)-2;    // dc_gym.py:114-116 (the set-point itself is kept)
    o[R_CONSEC] = 0u;
    o[R_SCALE] = 1u;
    o[R_EPISODE] = (unsigned)episode;
    o[R_FAULT] = fault;
    for (int b = 0; b < 3; b++) reinterpret_cast<double*>(S.hdr + (size_t)env * SDC_HDR_DWORDS + H_RET)[b] = 0.0;
  }
  __syncthreads();
  recp[lane] = sh.rec[lane];
  __syncthreads();
  if (obs) {
    obs[(size_t)env * SDC_OBS_OUT + lane] = obs_padded_at(sh.obs, lane);
    if (lane < SDC_OBS_OUT - 64) obs[(size_t)env * SDC_OBS_OUT + 64 + lane] = obs_padded_at(sh.obs, 64 + lane);
  }
  if (share_obs && lane < SDC_SHARE_OBS_DIM) share_obs[(size_t)env * SDC_SHARE_OBS_DIM + lane] = share_obs_at(sh.obs, lane);
  if (lane == 0) prof_stamp(S, SDC_PROF_RESET, env, 1);
}
