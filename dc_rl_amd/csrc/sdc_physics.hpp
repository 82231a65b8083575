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

}  // namespace
