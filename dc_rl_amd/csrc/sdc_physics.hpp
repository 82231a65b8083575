// sdc_physics.hpp -- what every mapping of the step shares below the lane level: the order of a data-centre config's scalars,
// the fp64 constant table, the short log2 / exp2 the rack model's powers are made of, the chiller model.
// (Everything in an unnamed namespace: each translation unit that includes this gets its own copy.)
#pragma once
#include "sdc_device.hpp"
#include "sdc_tuning.hpp"

namespace {

constexpr int HL = 32;    // lanes of a half wavefront (= racks the rack model covers in one pass of a half)

// the scalars of a data-centre config, in the order they lie in SdcDcDev from sdc_dc_params::m_cpu on
enum {
  P_M_CPU = 0, P_C_CPU, P_RS_CPU, P_M_FAN, P_C_FAN, P_RS_FAN, P_ITFAN_REF_P, P_ITFAN_REF_V_RATIO, P_IT_FAN_FULL_LOAD_V,
  P_C_AIR, P_RHO_AIR, P_CRAC_SUPPLY_PU, P_CT_FAN_REF_P, P_CTAFR, P_MIN_TEMP, P_MAX_TEMP, P_INIT_SETPOINT, P_BAT_CAP,
  P_RC_N_RACKS, P_RC_ITFAN_REF_V_RATIO, P_RC_RHO_AIR, P_RC_CTAFR, P_RC_BAT_CAP, P_K_OUTLET, P_N_RACKS, P_RET_SUM, P_COUNT
};
static_assert(offsetof(SdcDcDev, k_outlet) - offsetof(SdcDcDev, p.m_cpu) == P_K_OUTLET * sizeof(double), "config scalars must be contiguous");
static_assert(offsetof(SdcDcDev, n_racks_f) - offsetof(SdcDcDev, p.m_cpu) == P_N_RACKS * sizeof(double), "config scalars must be contiguous");
static_assert(offsetof(SdcDcDev, ret_sum) - offsetof(SdcDcDev, p.m_cpu) == P_RET_SUM * sizeof(double), "config scalars must be contiguous");
static_assert(P_COUNT <= HL, "one config scalar per lane of a half");

// ---- fp64 CONSTANT TABLE in LDS -------------------------------------------------------------------------------------
// A double that is not one of the hardware's inline constants (or a value whose low 32 bits are zero) costs TWO
// instructions every time it is used: two s_mov_b32 (or v_mov_b32) building it.  The step's hot path used ~100 of them:
// 9 % of its instructions.  Each wavefront instead copies this table from memory to LDS once (two coalesced loads issued
// with its state record), and a use is a broadcast LDS read -- one instruction per constant, or per PAIR of constants
// that sit next to each other here (ds_read_b128; the polynomials' coefficients are listed in the order they are used).
// KC(v) is the table entry holding the literal v, looked up at COMPILE time (a value missing from the list does not
// compile), so the formulas keep their literals and both kernel variants read the very same bits.
#define SDC_KVALS_LIST                                                                                                      \
  /* log2_pos_normal */ 0.70710678118654752, 1.0 / 17.0, 1.0 / 15.0, 1.0 / 13.0, 1.0 / 11.0, 1.0 / 9.0, 1.0 / 7.0, 1.0 / 5.0, \
      1.0 / 3.0, 2.8853900817779268, /* the rack model's two exponents */ 1.096, 0.824,                                         \
      /* exp2_short / exp2_plain */ 0.6931471805599453, 1.0 / 40320.0, 1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0,     \
      1.0 / 6.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.4426950408889634,                     \
      /* chiller_power */ 1.0 / 2.778, 0.94483600, -0.05700880, 0.00185486, 2.333, -1.975, 0.6121, 0.03303, 0.6852, 0.2818,   \
      0.05, 1.0 / 0.05, 6.67 - 35.0,                                                                           \
      /* rack model, water, battery, load shifting */ 3.8, 5.3, -14.01, 0.3528, 0.101, 0.044, 0.01, 0.8, 0.2, 1.0 / 100.0,      \
      1.0 / 20.0, 1.0 / 1e3, 1.0 / 60.0, 1.0 / 1e4, 1.0 / 1e8, 0.04, 0.1, 1e-300, 1e300
constexpr double SDC_KVALS[] = {SDC_KVALS_LIST};
constexpr int SDC_K_COUNT = (int)(sizeof(SDC_KVALS) / sizeof(double));
constexpr int SDC_K_LDS = 128;     // table entries in LDS (two per lane)
static_assert(SDC_K_COUNT <= SDC_K_LDS, "grow the LDS constant table");
// ... and behind the constants, in the table's last ten entries, 78 BYTES: the pool index of every entry of the padded [3][26]
// observation block (obs_pool_index; 0xFF = a padding zero), read by the output code (obs_padded_lut)
constexpr int SDC_K_OBS_SRC = SDC_K_LDS - 10;
static_assert(SDC_K_COUNT <= SDC_K_OBS_SRC && SDC_OBS_OUT <= 80, "the observation-source bytes sit behind the constants");
struct SdcKTabInit {
  double v[SDC_K_LDS];
};
constexpr SdcKTabInit sdc_make_ktab() {
  SdcKTabInit t{};
  for (int i = 0; i < SDC_K_COUNT; i++) t.v[i] = SDC_KVALS[i];
  for (int w = 0; w < 10; w++) {
    unsigned long long bits = 0ull;
    for (int b = 0; b < 8; b++) {
      const int j = 8 * w + b;
      const int src = j < SDC_OBS_OUT ? obs_pool_index(j) : -1;
      bits |= (unsigned long long)(src < 0 ? 0xFF : src) << (8 * b);
    }
    t.v[SDC_K_OBS_SRC + w] = __builtin_bit_cast(double, bits);
  }
  return t;
}
__device__ const SdcKTabInit SDC_KTAB_S = sdc_make_ktab();
#define SDC_KTAB SDC_KTAB_S.v
constexpr int sdc_kfind(const double v) {
  for (int i = 0; i < SDC_K_COUNT; i++)
    if (SDC_KVALS[i] == v) return i;
  return -1;
}
template <int I>
struct SdcKIdx {
  static_assert(I >= 0, "this literal is not in SDC_KVALS_LIST");
  static constexpr int idx = I;
};
#define KC(LITERAL) (kt[SdcKIdx<sdc_kfind(LITERAL)>::idx])
// where the formulas get their constants from: the LDS table (the kernels specialised for the common case), or the
// literals themselves (the general kernels, whose rack loop would otherwise hold the table's values in registers)
struct KLds {
  const double* t;
  __device__ __forceinline__ double operator[](const int i) const { return t[i]; }
};
struct KLit {
  __device__ __forceinline__ constexpr double operator[](const int i) const { return SDC_KVALS[i]; }
};
template <bool FAST> struct KSel { using type = KLit; };
template <> struct KSel<true> { using type = KLds; };
// x / C and np.round(x, d) with C, 1 / C (10^d, 10^-d) from the table where they are not free literals
#define KDIV(x, C) sdc_div_const((x), (double)(C), KC(1.0 / (double)(C)))
#define k_round(x, P10) KDIV(rint((x) * (P10)), (P10))
// each wavefront copies the table to LDS (the workgroup's wavefronts write the same values: no barrier needed)
__device__ __forceinline__ void ktab_fetch(const int lane, double& k0, double& k1) {
  k0 = SDC_KTAB[lane];
  k1 = SDC_KTAB[lane + SDC_WAVE];
}
__device__ __forceinline__ void ktab_store(double* kt, const int lane, const double k0, const double k1) {
  kt[lane] = k0;
  kt[lane + SDC_WAVE] = k1;
}

// log2 of a positive, normal, finite double: |error| <= 3e-15 absolute (for the rack model's x^y = exp2(y log2 x), nine
// orders below what the fp32 outputs resolve) in 29 instructions -- the library's correctly rounded, every-special-case
// log2 is 82.  x = m 2^e with m in [sqrt(1/2), sqrt(2)); log m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 0.172: the odd
// series through s^17 leaves 9e-16 relative.
template <class KT>
__device__ __forceinline__ double log2_pos_normal(const double x, const KT kt) {
  double m = __builtin_amdgcn_frexp_mant(x);           // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  const int up = m < KC(0.70710678118654752) ? 1 : 0;
  m = __builtin_amdgcn_ldexp(m, up);
  e -= up;
  const double f = m - 1.0, d = m + 1.0;
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  double q = f * r;
  q = fma(fma(-d, q, f), r, q);                         // s = f / d to the last place or so
  const double s2 = q * q;
  double p = KC(1.0 / 17.0);
  p = fma(p, s2, KC(1.0 / 15.0));
  p = fma(p, s2, KC(1.0 / 13.0));
  p = fma(p, s2, KC(1.0 / 11.0));
  p = fma(p, s2, KC(1.0 / 9.0));
  p = fma(p, s2, KC(1.0 / 7.0));
  p = fma(p, s2, KC(1.0 / 5.0));
  p = fma(p, s2, KC(1.0 / 3.0));
  p = fma(p, s2, 1.0);
  return fma(q * p, KC(2.8853900817779268) /* 2 / ln 2 */, (double)e);
}

// exp(t) for |t| <= 700: 2^(t log2 e) with the fraction's power from a degree-12 Taylor polynomial (<= 2e-16 relative)
// + the scaling by ldexp; 17 instructions against the library's 45
template <class KT>
__device__ __forceinline__ double exp2_plain(const double y, const KT kt);
template <class KT>
__device__ __forceinline__ double exp_plain(const double t, const KT kt) { return exp2_plain(t * KC(1.4426950408889634), kt); }
// 2^y for |y| <= 1000, same way
template <class KT>
__device__ __forceinline__ double exp2_plain(const double y, const KT kt) {
  const double n = __builtin_rint(y);
  const double f = (y - n) * KC(0.6931471805599453);     // |f| <= 0.3466
  double p = KC(1.0 / 479001600.0);
  p = fma(p, f, KC(1.0 / 39916800.0));
  p = fma(p, f, KC(1.0 / 3628800.0));
  p = fma(p, f, KC(1.0 / 362880.0));
  p = fma(p, f, KC(1.0 / 40320.0));
  p = fma(p, f, KC(1.0 / 5040.0));
  p = fma(p, f, KC(1.0 / 720.0));
  p = fma(p, f, KC(1.0 / 120.0));
  p = fma(p, f, KC(1.0 / 24.0));
  p = fma(p, f, KC(1.0 / 6.0));
  p = fma(p, f, 0.5);
  p = fma(p, f, 1.0);
  p = fma(p, f, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)n);
}

// 2^y to <= 3e-10 relative (degree 8): for the rack outlet-temperature rise, whose consumers resolve 1e-7 at best
template <class KT>
__device__ __forceinline__ double exp2_short(const double y, const KT kt) {
  const double n = __builtin_rint(y);
  const double f = (y - n) * KC(0.6931471805599453);
  double p = KC(1.0 / 40320.0);
  p = fma(p, f, KC(1.0 / 5040.0));
  p = fma(p, f, KC(1.0 / 720.0));
  p = fma(p, f, KC(1.0 / 120.0));
  p = fma(p, f, KC(1.0 / 24.0));
  p = fma(p, f, KC(1.0 / 6.0));
  p = fma(p, f, 0.5);
  p = fma(p, f, 1.0);
  p = fma(p, f, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)n);
}

// envs/datacenter.py:356-429 calculate_chiller_power
template <class KT>
__device__ __forceinline__ double chiller_power(double max_cooling_cap, double load, double ambient_temp, const KT kt) {
  const double min_plr = KC(0.05), max_plr = 1.0, design_cond_temp = 35.0;
  // temp_rise_coef = 2.778, rated_cop = 3.0 (divisors below); design_evp_out_temp = 6.67
  const double delta_temp = (ambient_temp - design_cond_temp) * KC(1.0 / 2.778) - KC(6.67 - 35.0);
  const double cap_rat = KC(0.94483600) + KC(-0.05700880) * delta_temp + KC(0.00185486) * (delta_temp * delta_temp);
  const double avail = cap_rat != 0 ? max_cooling_cap * cap_rat : 0.0;
  const double fpr = KC(2.333) + KC(-1.975) * cap_rat + KC(0.6121) * (cap_rat * cap_rat);
  const double ratio = sdc_div_fast(load, avail);   // (one division: the reference evaluates load / avail three times; only used where avail > 0)
  const double plr = avail > 0 ? fmax(min_plr, fmin(ratio, max_plr)) : 0.0;
  const double fflp = KC(0.03303) + KC(0.6852) * plr + KC(0.2818) * (plr * plr);
  double oper;
  if (avail > 0)
    oper = (ratio < min_plr) ? ratio : plr;
  else
    oper = 0.0;
  const double frac = oper < min_plr ? fmin(1.0, oper * KC(1.0 / 0.05)) : 1.0;   // / min_plr
  const double power = ((fflp * fpr * avail) * KC(1.0 / 3.0)) * frac;
  return oper > 0 ? power : 0.0;
}

// =====================================================================================================================================
// THE STEP'S EXPRESSIONS, once: what sdc_pairstep.hpp (two / four envs per wavefront) and sdc_wide.hip (one lane per env) both
// evaluate per env.  Which lane holds an env's values, where the operands come from (LDS broadcast, v_readlane, registers) and how the
// racks are spread over lanes stays with the callers; the arithmetic -- and so the bits -- is here.  KT: the constant source (KLds / KLit).

// ---- load shifting: envs/carbon_ls.py:172-324 -------------------------------------------------------------------------------------------
// The reference keeps a deque of per-task enqueue timestamps and only ever removes a FIFO prefix (overdue `remove()` loop :225-226 and
// popleft :257-258).  Equivalent state: cum[t] = tasks ever enqueued up to step t of the episode, popped = tasks ever removed.  Tasks
// still queued that were enqueued at or before step t: max(0, cum[t] - popped).
struct LsStep {
  int ns, shf, overdue, od_proc, popped, add, dropped, processed, util_tasks, cum_now, total, a24, a48, a72, a96;
  unsigned cumT_now;
};
// cum97 .. cum96: cum[now - 97], cum[now - 24], [now - 48], [now - 72], [now - 96] (0 before the episode's start)
template <class KT>
__device__ __forceinline__ LsStep ls_algebra(const KT kt, const double wl, const int a_ls, const int popped0, const int cum_prev, const unsigned cumT_prev,
                                             const int now, const int queue_max, const int cum97, const int cum24, const int cum48, const int cum72,
                                             const int cum96) {
  LsStep o;
  static_assert(1 - 0.2 == 0.8, "nonflex");
  const double flex = KC(0.2);     // class default; make_ls_env never forwards flexible_load (make_envs_pyenv.py:37-41)
  const double nonflex = KC(0.8);  // 1 - flex
  o.ns = (int)ceil(wl * nonflex * 100);
  o.shf = (int)floor(wl * flex * 100);
  int popped = popped0;
  o.overdue = max(0, cum97 - popped);       // overdue: age > 24 h  <=>  enqueued at step <= now - 97  (carbon_ls.py:208)
  int avail = 90 - (o.ns + o.shf);
  o.od_proc = 0;
  if (avail > 0 && o.overdue > 0) o.od_proc = min(o.overdue, avail);
  popped += o.od_proc;
  avail = 90 - (o.ns + o.shf + o.od_proc);
  // (selects, not branches: the envs of a wavefront usually take different actions)
  const int qlen = cum_prev - popped;                                  // queued after the overdue tasks have run
  const bool defer = a_ls == 0, drain = a_ls == 2 && avail >= 1;
  o.add = defer ? min(o.shf, queue_max - qlen) : 0;                    // a = 0: enqueue what fits, the rest is dropped (:231-242)
  o.dropped = defer ? o.shf - o.add : 0;
  o.processed = drain ? min(min(o.shf, avail), qlen) : 0;              // a = 2: pop from the left (:244-264)
  popped += o.processed;
  o.popped = popped;
  o.util_tasks = o.od_proc + (defer ? o.shf - o.add : o.shf + o.processed);   // the flexible part of the utilisation, in tasks (a = 1: :266-268)
  o.cum_now = cum_prev + o.add;
  o.cumT_now = cumT_prev + (unsigned)o.add * (unsigned)now;
  o.total = o.cum_now - popped;
  // age histogram, bins [0,6,12,18,24,inf] hours = [0,24,48,72,96,inf) steps (carbon_ls.py:63-73)
  o.a24 = max(0, cum24 - popped); o.a48 = max(0, cum48 - popped);
  o.a72 = max(0, cum72 - popped); o.a96 = max(0, cum96 - popped);
  return o;
}
template <class KT>
__device__ __forceinline__ double ls_utilisation(const KT kt, const LsStep& ls) {
  double util = KDIV((double)ls.util_tasks, 100);
  util += KDIV((double)ls.ns, 100);
  return util;
}
// the five bins as fractions of the queue (four divisions by the same count: one reciprocal, then the exact 3-instruction form)
__device__ __forceinline__ void ls_age_hist(const LsStep& ls, const double den, const double rden, double (&hist)[5]) {
  hist[0] = sdc_div_const((double)(ls.total - ls.a24), den, rden);
  hist[1] = sdc_div_const((double)(ls.a24 - ls.a48), den, rden);
  hist[2] = sdc_div_const((double)(ls.a48 - ls.a72), den, rden);
  hist[3] = sdc_div_const((double)(ls.a72 - ls.a96), den, rden);
  hist[4] = ls.a96 > 0 ? 1.0 : 0.0;
}
// oldest / mean task age in hours once the oldest task's step `head` is known (the callers search for it their own way), and the
// cached prefix counts in front of it.  sum of enqueue steps over the queued tasks = cumT[now] - cumT[h-1] - (popped - cum[h-1]) * h
__device__ __forceinline__ void ls_ages(const LsStep& ls, const int now, const int cum_prev, const unsigned cumT_prev, const bool was_empty,
                                        const double den, const double rden, int& head, int& cum_hm1, unsigned& cumT_hm1, double& oldest, double& avg) {
  oldest = 0.0;
  avg = 0.0;
  if (ls.total > 0) {
    if (was_empty) {          // everything queued was enqueued now
      head = now;
      cum_hm1 = cum_prev;
      cumT_hm1 = cumT_prev;
    }
    const long long sum_t = (long long)ls.cumT_now - (long long)cumT_hm1 - (long long)(ls.popped - cum_hm1) * head;
    const long long sum_age_steps = (long long)ls.total * now - sum_t;
    oldest = (double)(now - head) * 0.25;                              // hours, exact
    avg = sdc_div_const((double)sum_age_steps * 0.25, den, rden);      // / total (> 0 here); sum(ages) is exact in the reference too
  } else {
    head = now;
    cum_hm1 = ls.cum_now;
    cumT_hm1 = ls.cumT_now;
  }
}

// ---- rule-based policies (sdc_config.policy) ---------------------------------------------------------------------------------------------
// utils/trim_and_respond.py:28-38 on the room temperature the previous step reported (dc_int_temperature); response_duration_limit = 4
__device__ __forceinline__ int trim_and_respond_action(const double tr_limit, const double room, int& tr_count) {
  int a_dc;
  if (tr_limit >= room) {
    if (tr_count > 4) {
      tr_count = 0;
      a_dc = 2;
    } else {
      tr_count += 1;
      a_dc = 1;
    }
  } else {
    a_dc = 0;
  }
  return a_dc;
}
// utils/rbc_agents.py:21-47 (look_ahead 3, smooth_window 1) on [ci, ci_future] of the step's info: charge when the carbon intensity three
// steps ahead is above the current one, else discharge (on the NORMALISED values the reference's agent is given: managers.py:437)
__device__ __forceinline__ int rbc_battery_action(const double ci_3, const double ci_now, const double cmin, const double cden) {
  return (ci_3 - cmin) / cden > (ci_now - cmin) / cden ? 0 : 1;
}

// ---- CRAC set-point integrator: envs/dc_gym.py:160-174 (delta = a_dc - 1: make_envs_pyenv.py:127-131) -----------------------------------
__device__ __forceinline__ double setpoint_step(const int a_dc, const int last_delta, int& consecutive, int& scale, const double stpt0,
                                                const double max_temp, const double min_temp) {
  const int delta = a_dc - 1;
  if (last_delta != -2 && delta == last_delta && a_dc != 0) {
    consecutive += 1;
  } else {
    consecutive = 1;
    scale = 1;
  }
  if (consecutive > 3) scale += 1;
  const double stpt = stpt0 + (double)(delta * scale);
  return fmax(fmin(stpt, max_temp), min_temp);
}

// ---- rack model: envs/datacenter.py:250-317, :157-181 -------------------------------------------------------------------------------------
// the part of a rack that depends on (number of CPUs, supply approach temperature) and the env's set-point / load only
struct RackAir {
  double inlet, ratio, pf, vtot;      // inlet temperature, CPU power ratio, the rack's fan power, its air volume flow
};
struct RackEnv {                       // per env and step: the config's curve parameters and the load's shifts
  double m_cpu, c_cpu, m_fan, c_fan, cpu_shift, fan_shift, itfan_ref_p, rc_itfan_ref_v_ratio, it_fan_full_load_v, k_outlet;
};
template <class KT>
__device__ __forceinline__ RackAir rack_air(const KT kt, const RackEnv& E, const double r_n, const double r_supply, const double stpt) {
  RackAir a;
  const double sa = fmax(KC(3.8), fmin(r_supply, KC(5.3)));  // datacenter.py:209-215
  a.inlet = sa + stpt;
  a.ratio = ((E.m_cpu + KC(0.05)) * a.inlet + E.c_cpu) + E.cpu_shift;
  const double v = (E.m_fan * 10 * a.inlet + E.c_fan * 5) + E.fan_shift;
  const double fan1 = E.itfan_ref_p * (v * E.rc_itfan_ref_v_ratio);
  const double vf1 = E.it_fan_full_load_v * v;
  a.pf = r_n * fan1;
  a.vtot = r_n * vf1;
  return a;
}
__device__ __forceinline__ double rack_cpu_power(const RackAir& a, const double r_n, const double r_full, const double r_idle) {
  const double cpu1 = fmax(r_idle, r_full * a.ratio);
  return r_n * cpu1;
}
// positive, normal, finite -- always, for a valid config.  Anything else has no outlet temperature in the reference either (a power of a
// negative number; datacenter.py:295-300 then raises): it is flagged like an outlet below the inlet and evaluated at 1.
template <class KT>
__device__ __forceinline__ bool rack_plain(const KT kt, const double x) { return x > KC(1e-300) && x < KC(1e300); }
// outlet temperature from the two logarithms: x^y as exp2(y log2 x) (<= 5e-14 relative against the correctly rounded power, eight orders
// below the fp32 outputs' resolution), power^1.096 / airflow^0.824 as ONE exp2 of the difference of the two scaled logarithms:
// 1.918 power^1.096 / (c_air rho_air airflow^0.824 0.526) - 14.01
template <class KT>
__device__ __forceinline__ double rack_outlet(const KT kt, const RackEnv& E, const double inlet, const double l2_power, const double l2_airflow) {
  const double rise = exp2_short(KC(1.096) * l2_power - KC(0.824) * l2_airflow, kt);
  return inlet + E.k_outlet * rise + KC(-14.01);
}
// one rack, everything: {cpu power, fan power, outlet temperature, plain}
struct RackOut {
  double pc, pf, out;
  bool plain;
};
template <class KT>
__device__ __forceinline__ RackOut rack_point(const KT kt, const RackEnv& E, const double r_n, const double r_supply, const double r_full,
                                              const double r_idle, const double stpt, double& inlet) {
  const RackAir a = rack_air(kt, E, r_n, r_supply, stpt);
  RackOut o;
  o.pc = rack_cpu_power(a, r_n, r_full, r_idle);
  o.pf = a.pf;
  const double pw = o.pc + o.pf;
  o.plain = rack_plain(kt, pw) && rack_plain(kt, a.vtot);
  o.out = rack_outlet(kt, E, a.inlet, log2_pos_normal(o.plain ? pw : 1.0, kt), log2_pos_normal(o.plain ? a.vtot : 1.0, kt));
  inlet = a.inlet;
  return o;
}

// ---- HVAC: envs/datacenter.py:432-474 ; water :325-353 -------------------------------------------------------------------------------------
struct HvacPrm {
  double c_air, rho_air, ct_fan_ref_p, crac_supply_pu, rc_rho_air, rc_ctafr;
};
struct HvacOut {
  double comp, ct, water, total_kw;
};
template <class KT>
__device__ __forceinline__ HvacOut hvac_water(const KT kt, const HvacPrm& P, const double p_it, const double avg_ret, const double stpt,
                                              const double amb, const double wet_bulb) {
  HvacOut o;
  const double m_sys = P.rho_air * P.crac_supply_pu * p_it;
  const double q_cool = m_sys * P.c_air * fmax(0.0, avg_ret - stpt);
  o.comp = chiller_power(P.ct_fan_ref_p, q_cool, amb, kt);
  {
    const double dlt = fmax(50 - (amb - stpt), 1);
    const double m_air = sdc_div_fast(q_cool, P.c_air * dlt);
    const double v_air = m_air * P.rc_rho_air;
    const double x = fmin(v_air * P.rc_ctafr, 1);
    o.ct = amb < 5 ? 0.0 : P.ct_fan_ref_p * (x * x * x);
  }
  {
    const double range_temp = avg_ret - stpt;
    const double y_int = KC(0.3528) * range_temp + KC(0.101);
    double w = KC(0.044) * wet_bulb + y_int;
    if (w < 0) w = 0;
    w += w * KC(0.01);
    o.water = k_round((w * 1000) / 4, 1e4);
  }
  o.total_kw = KDIV(p_it + o.ct + o.comp, 1e3);
  return o;
}

// ---- battery: envs/bat_env_fwd_view.py:84-245, envs/battery_model.py:94-132 --------------------------------------------------------------
// charge and discharge share one sigmoid and one division (selected operands, the reference's expressions)
struct BatOut {
  double e_nobat, energy, co2, soc_after;
};
template <class KT>
__device__ __forceinline__ BatOut battery_step(const KT kt, const int a_bat, double& bat_load, const double cap, const double rc_cap,
                                               const double total_kw, const double ci_i, unsigned& fault) {
  BatOut o;
  const double dcload = KDIV(total_kw, 1e3);  // MW (sustaindc_env.py:652)
  o.e_nobat = dcload * 1e3 * 0.25;
  o.energy = o.e_nobat;
  if (a_bat != 2) {
    const bool chg = a_bat == 0;
    const double soc = sdc_div_const(bat_load - 0, cap - 0, rc_cap);
    const double sg = 1 / (1 + exp_plain(-(10 * (soc - (chg ? 0.5 : 0.25))), kt));       // sigmoid (|argument| <= 10)
    const double rate = chg ? k_round(0.5 * (1 - sg), 1e4) : fmax(0.5, 4 * sg);
    const double tu = KDIV(rate * 15, 60);
    // charge:    (1 * cap - bat_load) / ((1 * tu) - (-0.04))        discharge: (bat_load - 0 * cap) / (0.01 + (1 * tu))
    const double quo = (chg ? cap - bat_load : bat_load) / (chg ? tu + KC(0.04) : KC(0.01) + tu);
    if (chg) {
      const double max_charge = fmin((cap / 1) * KC(0.1), quo);
      const double charging_load = fmin(max_charge, cap) * 1 * tu;
      bat_load = k_round(bat_load + charging_load, 1e8);
      o.energy = o.e_nobat + charging_load * 1e3;
    } else {
      const double max_d = fmin(fmin((cap / 1) * 1, quo), dcload * 0.25);   // dcload / 4
      bat_load = k_round(bat_load - (fmin(max_d, cap) * 1 * tu), 1e8);
      const double discharge = max_d < cap ? max_d * tu : cap * tu;
      if (!(o.e_nobat >= discharge * 1e3)) fault |= SDC_FAULT_BAT_DISCHARGE;
      o.energy = o.e_nobat - discharge * 1e3;
    }
  }
  o.co2 = (a_bat == 1 ? fmax(o.energy, 0.0) : o.energy) * ci_i;
  o.soc_after = sdc_div_const(bat_load, cap, rc_cap);
  return o;
}

// ---- history append (utils/reward_creator.py:7-14): the ring slot this step's value goes to ------------------------------------------------
__device__ __forceinline__ int hist_append_slot(int& hl, int& hpos, const int hist_cap) {
  int slot;
  if (hl < hist_cap) {
    slot = hl;
    hl += 1;
  } else {
    slot = hpos;
    hpos = hpos + 1 == hist_cap ? 0 : hpos + 1;
  }
  return slot;
}

// ---- the info row's entries (sustaindc_env.py:600-700 common info; SDC_INFO_*), into whatever `inf` indexes -------------------------------
struct InfoLs {
  double wl, util, normq, oldest_norm, avg_norm, hour;
  int total, dropped, processed, overdue;
};
template <class ROW>
__device__ __forceinline__ void info_put_ls(ROW&& inf, const InfoLs& v, const double (&hist)[5]) {
  inf[SDC_INFO_LS_ORIGINAL_WORKLOAD] = (float)v.wl;
  inf[SDC_INFO_LS_SHIFTED_WORKLOAD] = (float)v.util;
  inf[SDC_INFO_LS_TASKS_IN_QUEUE] = (float)v.total;
  inf[SDC_INFO_LS_NORM_TASKS_IN_QUEUE] = (float)v.normq;
  inf[SDC_INFO_LS_TASKS_DROPPED] = (float)v.dropped;
  inf[SDC_INFO_LS_TASKS_PROCESSED] = (float)v.processed;
  inf[SDC_INFO_LS_OLDEST_TASK_AGE] = (float)v.oldest_norm;
  inf[SDC_INFO_LS_AVERAGE_TASK_AGE] = (float)v.avg_norm;
  inf[SDC_INFO_LS_OVERDUE_PENALTY] = (float)v.overdue;
  inf[SDC_INFO_LS_COMPUTED_TASKS] = (float)(int)(v.util * 100);
  inf[SDC_INFO_LS_CURRENT_HOUR] = (float)v.hour;
#pragma unroll
  for (int b = 0; b < 5; b++) inf[SDC_INFO_LS_AGE_HIST0 + b] = (float)hist[b];
  inf[SDC_INFO_DC_CPU_WORKLOAD_FRACTION] = (float)v.util;
}
struct InfoDc {
  double p_it, ct, comp, total_kw, stpt, mean_outlet, amb, water, soc_after, co2, ci_i, e_nobat, energy, norm_ci, amb_next;
  int delta, a_bat, day_n, hourq_n, rel_next;
  unsigned fault;
};
template <class KT, class ROW>
__device__ __forceinline__ void info_put_dc(const KT kt, ROW&& inf, const InfoDc& v) {
  inf[SDC_INFO_DC_ITE_TOTAL_POWER_KW] = (float)(v.p_it * KC(1.0 / 1e3));
  inf[SDC_INFO_DC_CT_TOTAL_POWER_KW] = (float)(v.ct * KC(1.0 / 1e3));
  inf[SDC_INFO_DC_COMPRESSOR_TOTAL_POWER_KW] = (float)(v.comp * KC(1.0 / 1e3));
  inf[SDC_INFO_DC_HVAC_TOTAL_POWER_KW] = (float)((v.ct + v.comp) * KC(1.0 / 1e3));
  inf[SDC_INFO_DC_TOTAL_POWER_KW] = (float)v.total_kw;
  inf[SDC_INFO_DC_CRAC_SETPOINT_DELTA] = (float)v.delta;
  inf[SDC_INFO_DC_CRAC_SETPOINT] = (float)v.stpt;
  inf[SDC_INFO_DC_INT_TEMPERATURE] = (float)v.mean_outlet;
  inf[SDC_INFO_DC_EXTERIOR_AMBIENT_TEMP] = (float)v.amb;
  inf[SDC_INFO_DC_WATER_USAGE] = (float)v.water;
  inf[SDC_INFO_BAT_ACTION] = (float)v.a_bat;
  inf[SDC_INFO_BAT_SOC] = (float)v.soc_after;
  inf[SDC_INFO_BAT_CO2_FOOTPRINT] = (float)v.co2;
  inf[SDC_INFO_BAT_AVG_CI] = (float)v.ci_i;
  inf[SDC_INFO_BAT_TOTAL_ENERGY_WITHOUT_BATTERY_KWH] = (float)v.e_nobat;
  inf[SDC_INFO_BAT_TOTAL_ENERGY_WITH_BATTERY_KWH] = (float)v.energy;
  inf[SDC_INFO_NORM_CI] = (float)v.norm_ci;
  inf[SDC_INFO_OUTSIDE_TEMP] = (float)v.amb_next;
  inf[SDC_INFO_DAY] = (float)v.day_n;
  inf[SDC_INFO_HOUR] = (float)((double)v.hourq_n * 0.25);
  inf[SDC_INFO_FAULT] = (float)v.fault;
  inf[SDC_INFO_ENERGY_Z] = 0.0f;       // the five columns below are filled by the reward part of the step
  inf[SDC_INFO_RESERVED] = 0.0f;
  inf[SDC_INFO_EP_RETURN_LS] = 0.0f;
  inf[SDC_INFO_EP_RETURN_DC] = 0.0f;
  inf[SDC_INFO_EP_RETURN_BAT] = 0.0f;
  inf[SDC_INFO_EPISODE_STEP] = (float)v.rel_next;
}

}  // namespace
