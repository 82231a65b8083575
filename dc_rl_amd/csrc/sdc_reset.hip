// sdc_reset.hip -- SustainDC.reset() for the masked environments, one wavefront per environment.
//
// Two modes:
//   * override: (day, hour, CI / temperature normalisation bounds, weather windows) were injected by the
//     caller (parity tests feed what the reference's managers produced);
//   * device:   the draws of sustaindc_env.py:454-455 and utils/managers.py:35-48, :596-613 are made on the
//     GPU with a counter-based RNG (Philox4x32-10 for the draws, -7 for the normals): start day / hour, the 0..13-day roll, and the
//     35 040-step Gaussian random walk (fp32 Box-Muller, fp64 accumulation) scaled to std 0.75 that is added
//     to dry and wet bulb before
//     the roll, clip to [0, 45] and 30-day min/max normalisation.  Same distributions as the reference,
//     not the same stream (the reference uses MT19937 via `random` and `np.random`).
//
// What reset clears / keeps follows the reference: queue, battery SoC and the set-point scaling
// counters are cleared (carbon_ls.py:85, battery_model.py:90-91, dc_gym.py:114-116); the CRAC set-point
// and the energy history survive (dc_gym.py:91-140 never touches raw_curr_stpt; reward_creator.py:5).
#include "sdc_device.hpp"
#include <type_traits>

namespace {

struct ResetShared {
  double nc[32];
  double nt[32];
  double tw[32];  // T[c0 .. c0+16] of a device-side reset
  float obs[64];
  unsigned rec[SDC_REC_DWORDS];
  enum { MAX_HIT = 14 };
  int hit_base[MAX_HIT];      // the blocks of the year's walk that reach the episode's window, and the walk's value before each
  double hit_carry[MAX_HIT];
};

// inclusive prefix sum over the 64 lanes on the DPP data path (no LDS round trips: this scan runs once per 512 samples of the year, ~76 times per reset):
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

// four standard normals per Philox4x32-7 block: two Box-Muller pairs in fp32 (hardware log2 / sqrt / sin / cos); the
// random walk itself is accumulated in fp64.  Block c of (env, episode) yields the normals of samples 4c .. 4c+3.
// uniform = (fp32(x) + 0.5) * 2^-32 as ONE fused multiply-add on the 32-bit word converted as it is (v_cvt_f32_u32 rounds to
// nearest even at 24 bits; fp32(x) * 2^-32 is exact, so there is one more rounding: in (0, 1], 1.0 for the top 128 words --
// a radius of 0 / an angle of one whole revolution, both fine).
__device__ __forceinline__ void normals4(const SdcDev& S, int env, int episode, int c, float* nz) {
  const Philox4 r = philox4x32<7>((unsigned)c, (unsigned)(S.env_base + env), (unsigned)episode, 0x7E47u, (unsigned)S.seed,
                                  (unsigned)(S.seed >> 32));
  const float k32 = 1.0f / 4294967296.0f, k33 = 1.0f / 8589934592.0f;
  const float u1 = __builtin_fmaf((float)r.x, k32, k33), u2 = __builtin_fmaf((float)r.y, k32, k33);
  const float u3 = __builtin_fmaf((float)r.z, k32, k33), u4 = __builtin_fmaf((float)r.w, k32, k33);
  // sqrt(-2 ln u) = sqrt(-2 ln2 log2 u): v_log_f32 and v_sqrt_f32 as they are (1 ulp; the argument is in [0, 35], no
  // denormal / special-case handling needed -- the IEEE expansion of sqrtf is 13 more instructions per root)
  const float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
  const float r2 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u3));
  nz[0] = r1 * __builtin_amdgcn_cosf(u2);   // v_cos_f32 / v_sin_f32 take revolutions: cos(2 pi u)
  nz[1] = r1 * __builtin_amdgcn_sinf(u2);
  nz[2] = r2 * __builtin_amdgcn_cosf(u4);
  nz[3] = r2 * __builtin_amdgcn_sinf(u4);
}

// v_min_f64 / v_max_f64 as they are: the compiler's fmin / fmax put a canonicalising v_max_f64 x, x in front of every operand
// that comes from memory (six more half-rate instructions per 64 samples of pass 2); nothing here is ever a NaN
__device__ __forceinline__ double min_f64_raw(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double max_f64_raw(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
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
#ifdef SDC_RT   // measurement build: phase stamps in the tail of the env's queue table (tools/reset_phases.py)
#define RT_STAMP(i) do { __builtin_amdgcn_s_waitcnt(0); if (lane == 0) reinterpret_cast<unsigned long long*>(S.qtab + (size_t)env * S.qstride + S.qstride - 8)[i] = wall_clock64(); } while (0)
#else
#define RT_STAMP(i) do { } while (0)
#endif
  RT_STAMP(0);
#ifdef SDC_RT
  if (lane == 0) S.qtab[(size_t)env * S.qstride + S.qstride - 9] = make_uint2(((__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu) << 16) | (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xFFFFu), 0u);
#endif
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
    // pass 2's outputs: add noise, roll, clip [0, 45]; 30-day min / max from the cursor (managers.py:598-613)
    const double* tT = S.tabT + (size_t)loc * TL;
    const double* tWB = S.tabWB + (size_t)loc * TL;
    const double* tC = S.tabC + (size_t)loc * TL;
    double tmin = 1e300, tmax = -1e300, cmin = 1e300, cmax = -1e300;
    double* tw = S.t_win + (size_t)env * S.lw;
    double* wbw = S.wb_win + (size_t)env * S.lw;
    RT_STAMP(1);
    if (S.noise_std > 0.0 && !injected) {
      // CoherentNoise.generate (managers.py:35-48): the year's random walk and its population std -- and then ONLY the blocks
      // of the walk that the episode's window reads, a second time.  The window needs walk / std, and std is known when the
      // whole year has been walked; round 1-3 parked the window's 2 880 raw walk values in global memory (23 KB per env
      // written, then read back: with 4096 resets in flight that and the three table streams made "pass 2" memory bound,
      // 130 us of the kernel).  A counter-based generator can simply be asked again: the window lies in 6-7 of the year's 69
      // blocks, the running total at each of their starts is kept (LDS), and the second visit produces bit-identical walk
      // values that go straight from registers into clip(T + noise), the bounds and the episode's windows.
      // 512 samples per block: 8 per lane (two Philox blocks: two independent chains in flight), lane-local prefix + ONE wave
      // scan of the lane totals.  TL is a multiple of 8, so a lane's 8 samples are all inside the year or all outside.
      constexpr int SPL = 8;                       // samples per lane and block
      constexpr int BLK = SPL * SDC_WAVE;
      const int shift_s = __builtin_amdgcn_readfirstlane(shift), c0_s = __builtin_amdgcn_readfirstlane(c0);
      const int wlen_s = __builtin_amdgcn_readfirstlane(wlen);
      // one block of the walk: p[k] = walk value of sample base + 8 lane + k (carry = the walk's value before the block)
      auto walk_block = [&](const int base, const double carry, double (&p)[SPL], double& total) {
        float nz[SPL];
        const int j0 = base + SPL * lane;
#pragma unroll
        for (int b = 0; b < SPL / 4; b++) normals4(S, env, episode, (j0 >> 2) + b, nz + 4 * b);
        const bool inside = j0 < TL;
        // p[k] = p[k - 1] + weight * z as ONE fused multiply-add
#pragma unroll
        for (int k = 0; k < SPL; k++) {
          const double z = inside ? (double)nz[k] : 0.0;
          p[k] = k == 0 ? S.noise_weight * z : __builtin_fma(S.noise_weight, z, p[k - 1]);
        }
        const double incl = wave_incl_scan_f64(p[SPL - 1], lane);
        const double off = carry + (incl - p[SPL - 1]);
        total = readlane_f64(incl, 63);
#pragma unroll
        for (int k = 0; k < SPL; k++) p[k] += off;
      };
      // does the block reach the window the episode reads?  rolled position of its first sample (np.roll, managers.py:602);
      // the block may run past TL: the part beyond wraps to [0, blk1 - TL)
      auto reaches_window = [&](const int base) {
        int blk0 = base + shift_s;
        if (blk0 >= TL) blk0 -= TL;
        const int blk1 = blk0 + BLK;
        return (blk0 < c0_s + wlen_s && blk1 > c0_s) || (blk1 > TL && c0_s < blk1 - TL);
      };
      double carry = 0.0, sum = 0.0, sumsq = 0.0;
      int n_hit = 0;
      // the SIMD's four resident wavefronts take turns at the issue priorities, block by block: the arbiter serves the OLDEST
      // wavefront first, so without this they finish the walk one after the other and the last one walks alone at a lone
      // wavefront's pace (157 -> 152 us per reset of 4096 envs)
      const int slot = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 4) & 3u);   // HW_ID.wave_id: the wavefront's slot on its SIMD
      for (int base = 0; base < TL; base += BLK) {
        switch ((slot + (base >> 9)) & 3) {
          case 0: __builtin_amdgcn_s_setprio(0); break;
          case 1: __builtin_amdgcn_s_setprio(1); break;
          case 2: __builtin_amdgcn_s_setprio(2); break;
          default: __builtin_amdgcn_s_setprio(3); break;
        }
        if (reaches_window(base)) {
          if (lane == 0 && n_hit < ResetShared::MAX_HIT) {
            sh.hit_base[n_hit] = base;
            sh.hit_carry[n_hit] = carry;
          }
          n_hit++;
        }
        double p[SPL], total;
        walk_block(base, carry, p, total);
        carry += total;
        if (base + SPL * lane < TL) {
#pragma unroll
          for (int k = 0; k < SPL; k++) {
            sum += p[k];
            sumsq = __builtin_fma(p[k], p[k], sumsq);
          }
        }
      }
      __builtin_amdgcn_s_setprio(0);
      RT_STAMP(2);
      sum = wave_sum_f64(sum);
      sumsq = wave_sum_f64(sumsq);
      const double mean = sum / (double)TL;
      const double var = sumsq / (double)TL - mean * mean;
      const double walk_std = sqrt(var);
      // (walk / std) * noise_std (managers.py:47) as walk * (noise_std / std): one division per reset instead of one per sample
      const double nscale = walk_std > 0.0 ? S.noise_std / walk_std : 0.0;
      if (n_hit > ResetShared::MAX_HIT) fault |= SDC_FAULT_TABLE_RANGE;   // (a window of more than ~6 600 samples: no configuration has one)
      RT_STAMP(3);
      __syncthreads();
      for (int h = 0; h < min(n_hit, (int)ResetShared::MAX_HIT); h++) {
        const int base = sh.hit_base[h];
        double p[SPL], total;
        walk_block(base, sh.hit_carry[h], p, total);
        const int j0 = base + SPL * lane;
        int pos0 = j0 + shift_s;                 // base index j lands at rolled position (j + shift) mod TL; shift, TL and j0 are
        if (pos0 >= TL) pos0 -= TL;              // multiples of 8: a lane's 8 samples never straddle the wrap
        const int rel0 = pos0 - c0_s;
        if (j0 < TL && rel0 > -SPL && rel0 < wlen_s) {
          double tv[SPL], cv[SPL], wv[SPL];
          const bool want_wb = rel0 < S.lw;
#pragma unroll
          for (int k = 0; k < SPL; k += 2) {
            const double2 t2 = *reinterpret_cast<const double2*>(tT + j0 + k);
            const double2 c2 = *reinterpret_cast<const double2*>(tC + pos0 + k);
            tv[k] = t2.x; tv[k + 1] = t2.y;
            cv[k] = c2.x; cv[k + 1] = c2.y;
          }
          if (want_wb) {
#pragma unroll
            for (int k = 0; k < SPL; k += 2) {
              const double2 w2 = *reinterpret_cast<const double2*>(tWB + j0 + k);
              wv[k] = w2.x; wv[k + 1] = w2.y;
            }
          }
#pragma unroll
          for (int k = 0; k < SPL; k++) {
            const int rel = rel0 + k;
            if (rel >= 0 && rel < wlen_s) {
              const double nzk = p[k] * nscale;
              const double t = min_f64_raw(max_f64_raw(tv[k] + nzk, 0.0), 45.0);
              if (rel < SDC_NORM_WINDOW) {
                tmin = min_f64_raw(tmin, t);
                tmax = max_f64_raw(tmax, t);
                cmin = min_f64_raw(cmin, cv[k]);
                cmax = max_f64_raw(cmax, cv[k]);
              }
              if (rel < S.lw) {
                tw[rel] = t;
                wbw[rel] = min_f64_raw(max_f64_raw(wv[k] + nzk, 0.0), 45.0);
              }
              if (rel < 17) sh.tw[rel] = t;
            }
          }
        }
      }
    } else {
      // the caller's noise (what the reference's reset drew, tests) or none: U slices of 64 samples at a time, every load of a
      // batch issued before the first use (one memory round trip per batch instead of one per slice)
      constexpr int U = 8;
      for (int r0 = 0; r0 < wlen; r0 += U * SDC_WAVE) {
        const bool in_lw = r0 < S.lw;   // (uniform) only the episode's own window keeps its wet bulb
        double tv[U], nv[U], cv[U], wv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int rel = min(r0 + u * SDC_WAVE + lane, wlen - 1);
          int j = c0 + rel - shift;
          if (j < 0) j += TL;
          tv[u] = tT[j];
          nv[u] = injected ? inj[j] : 0.0;
          cv[u] = tC[c0 + rel];
          wv[u] = in_lw ? tWB[j] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int rel = r0 + u * SDC_WAVE + lane;
          if (rel < wlen) {
            const double t = min_f64_raw(max_f64_raw(tv[u] + nv[u], 0.0), 45.0);
            if (rel < SDC_NORM_WINDOW) {
              tmin = min_f64_raw(tmin, t);
              tmax = max_f64_raw(tmax, t);
              cmin = min_f64_raw(cmin, cv[u]);
              cmax = max_f64_raw(cmax, cv[u]);
            }
            if (rel < S.lw) {
              tw[rel] = t;
              wbw[rel] = min_f64_raw(max_f64_raw(wv[u] + nv[u], 0.0), 45.0);
            }
            if (rel < 17) sh.tw[rel] = t;
          }
        }
      }
    }
    RT_STAMP(4);
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
  RT_STAMP(5);
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
  RT_STAMP(6);
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
  RT_STAMP(7);
}
