/*
 * sdc_oracle.c -- CPU restatement of the reference's coupled SustainDC step (see sdc_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY (checker + timed CPU baseline).  Plain C99, fp64 everywhere the
 * reference uses Python float / NumPy float64; observations are cast to float32 at the very end,
 * exactly where the reference does (sustaindc_env.py:342, :386, :426).
 *
 * Paths in comments are relative to /root/reference.
 */
#include "sdc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ NumPy reduction semantics */

/* NumPy's pairwise summation for contiguous float64 add.reduce (np.mean / np.std / np.sum);
 * published algorithm: blocks of 128, 8 accumulators, recursive halving above 128. */
static double np_pairwise_sum(const double *a, long n) {
  if (n < 8) {
    double res = 0.0;
    for (long i = 0; i < n; i++) res += a[i];
    return res;
  } else if (n <= 128) {
    double r[8];
    long i;
    for (int j = 0; j < 8; j++) r[j] = a[j];
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
  } else {
    long n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
  }
}

static double np_mean(const double *a, long n) { return np_pairwise_sum(a, n) / (double)n; }

/* np.std (population): sqrt(mean((x - mean)^2)), numpy/_core/_methods.py::_var */
static double np_std(const double *a, long n, double *scratch) {
  double m = np_mean(a, n);
  for (long i = 0; i < n; i++) {
    double d = a[i] - m;
    scratch[i] = d * d;
  }
  return sqrt(np_pairwise_sum(scratch, n) / (double)n);
}

/* np.round(x, d): rint(x * 10^d) / 10^d */
static double np_round(double x, double p10) { return rint(x * p10) / p10; }

/* ------------------------------------------------------------------ order statistics */

static void swap_d(double *a, double *b) {
  double t = *a;
  *a = *b;
  *b = t;
}

/* quickselect: after return a[k] is the k-th smallest and a[0..k) <= a[k] <= a(k..n) */
static void select_kth(double *a, long lo, long hi, long k) {
  while (hi > lo) {
    long mid = lo + (hi - lo) / 2;
    if (a[mid] < a[lo]) swap_d(&a[mid], &a[lo]);
    if (a[hi] < a[lo]) swap_d(&a[hi], &a[lo]);
    if (a[hi] < a[mid]) swap_d(&a[hi], &a[mid]);
    double pivot = a[mid];
    long i = lo, j = hi;
    while (i <= j) {
      while (a[i] < pivot) i++;
      while (a[j] > pivot) j--;
      if (i <= j) {
        swap_d(&a[i], &a[j]);
        i++;
        j--;
      }
    }
    if (k <= j)
      hi = j;
    else if (k >= i)
      lo = i;
    else
      return;
  }
}

/* np.percentile(a, q) with the default 'linear' method on a scratch copy (which is permuted).
 * virtual index (n-1)*q/100; numpy's _lerp: a + (b-a)*t, and b - (b-a)*(1-t) where t >= 0.5. */
double sdco_percentile_linear(double *s, int n, double q) {
  double vi = (double)(n - 1) * (q / 100.0);
  long lo = (long)floor(vi);
  double t = vi - (double)lo;
  long hi = lo + 1 < n ? lo + 1 : n - 1;
  select_kth(s, 0, n - 1, lo);
  double a = s[lo];
  double b = a;
  if (hi != lo) { /* the successor is the minimum of the right partition */
    b = s[lo + 1];
    for (long i = lo + 2; i < n; i++)
      if (s[i] < b) b = s[i];
  }
  double d = b - a;
  return (t >= 0.5) ? b - d * (1.0 - t) : a + d * t;
}

/* utils/reward_creator.py:16-45 normalize_energy (history already contains `value`) */
static double normalize_energy_scratch(const double *hist, int n, double value, double *s);

double sdco_normalize_energy(const double *hist, int n, double value) {
  if (n < 2) return 0.0;
  double *s = (double *)malloc(sizeof(double) * (size_t)n * 2);
  double z = normalize_energy_scratch(hist, n, value, s);
  free(s);
  return z;
}

/* same, on caller-provided scratch of 2n doubles (the per-step path: a 160 KB malloc per step would go
 * through mmap/munmap and serialise the host threads of the CPU baseline on the kernel's mm lock) */
static double normalize_energy_scratch(const double *hist, int n, double value, double *s) {
  if (n < 2) return 0.0;
  double *sq = s + n;
  memcpy(s, hist, sizeof(double) * (size_t)n);
  double q1 = sdco_percentile_linear(s, n, 25.0);
  double q3 = sdco_percentile_linear(s, n, 75.0);
  double iqr = q3 - q1;
  double lb = q1 - 1.5 * iqr;
  double ub = q3 + 1.5 * iqr;
  for (int i = 0; i < n; i++) { /* np.clip keeps the history order -> same pairwise tree as numpy */
    double v = hist[i];
    s[i] = v < lb ? lb : (v > ub ? ub : v);
  }
  double mean = np_mean(s, n);
  double sd = np_std(s, n, sq);
  return (value - mean) / (sd > 0 ? sd : 1.0);
}

/* ------------------------------------------------------------------ time / obs features */

/* utils/managers.py:66-88 sc_obs: round(hour/24, 3) * 2pi -> cos/sin * 0.5 + 0.5.
 * Python's round() is round-half-even on the decimal value; x/96 hits exact .5 ties only at values
 * exactly representable in binary, where rint(x*1000) (ties-to-even) agrees. */
void sdco_hour_sincos(double hour, double *cos_h, double *sin_h) {
  double two_pi = 3.141592653589793 * 2;
  double nh = (rint((hour / 24) * 1000.0) / 1000.0) * two_pi;
  *cos_h = cos(nh) * 0.5 + 0.5;
  *sin_h = sin(nh) * 0.5 + 0.5;
}

/* np.polyfit(range(n), y, 1)[0] restated as the closed-form least-squares slope */
double sdco_polyfit_slope(const double *y, int n) {
  double xm = 0.5 * (double)(n - 1);
  double ym = 0.0;
  for (int i = 0; i < n; i++) ym += y[i];
  ym /= (double)n;
  double sxy = 0.0, sxx = 0.0;
  for (int i = 0; i < n; i++) {
    double dx = (double)i - xm;
    sxy += dx * (y[i] - ym);
    sxx += dx * dx;
  }
  return sxy / sxx;
}

/* sustaindc_env.py:266-300 extract_ci_features(ci_values, current_ci) */
void sdco_extract_features(const double *vals, int n, double cur, double out5[5]) {
  double sq[32], g[33], x[33] = {0};
  double mean = np_mean(vals, n);
  double sd = np_std(vals, n, sq);
  /* np.gradient(np.hstack((current, values))): central differences, one-sided at the ends */
  x[0] = cur;
  for (int i = 0; i < n; i++) x[i + 1] = vals[i];
  int m = n + 1;
  g[0] = x[1] - x[0];
  for (int i = 1; i < m - 1; i++) g[i] = (x[i + 1] - x[i - 1]) / 2.0;
  g[m - 1] = x[m - 1] - x[m - 2];
  int peak = n, valley = n;
  for (int i = 0; i < m - 1; i++)
    if (g[i] > 0 && g[i + 1] <= 0) {
      peak = i;
      break;
    }
  for (int i = 0; i < m - 1; i++)
    if (g[i] < 0 && g[i + 1] >= 0) {
      valley = i;
      break;
    }
  out5[0] = mean;
  out5[1] = sd;
  out5[2] = (cur - mean) / (sd + 1e-8);
  out5[3] = (double)peak / (double)n;
  out5[4] = (double)valley / (double)n;
}

/* 4-tap 'valid' moving average (np.convolve(x, ones(4), 'valid') / 4) */
static int smooth4(const double *x, int n, double *out) {
  if (n < 4) { /* numpy swaps operands when the kernel is longer: len-1 input -> 4 copies of x/4 */
    int m = 4 - n + 1;
    for (int j = 0; j < m; j++) {
      double s = 0.0;
      for (int k = 0; k < n; k++) s += x[k];
      out[j] = s / 4;
    }
    return m;
  }
  for (int j = 0; j + 3 < n; j++) out[j] = (((x[j] + x[j + 1]) + x[j + 2]) + x[j + 3]) / 4;
  return n - 3;
}

/* the 7 shared CI features (sustaindc_env.py:311-328) */
static void ci_features(const sdco_env *e, int ip, double out7[7]) {
  const double *NC = e->NC - e->win_lo; /* absolute indexing */
  double cur = NC[ip];
  double buf[20], sm[20];
  /* future: ci_future holds 8 values (n_vars_ci = 8, sustaindc_env.py:147,202); [:16] is a no-op */
  buf[0] = cur;
  for (int k = 0; k < 8; k++) buf[1 + k] = NC[ip + 1 + k];
  int m = smooth4(buf, 9, sm);
  out7[0] = sdco_polyfit_slope(sm, m);
  /* past: norm_carbon[ts-16:ts] -- EMPTY when ts < 16 (negative start; utils/managers.py:482-483) */
  int np_ = 0;
  if (ip >= 16)
    for (int k = 0; k < 16; k++) buf[np_++] = NC[ip - 16 + k];
  buf[np_++] = cur;
  m = smooth4(buf, np_, sm);
  out7[1] = sdco_polyfit_slope(sm, m);
  sdco_extract_features(&NC[ip + 1], 8, cur, &out7[2]);
}

static void build_obs(const sdco_env *e, const sdco_params *p, float *obs) {
  const int ip = e->cursor;
  const double *W = e->W - e->win_lo, *NC = e->NC - e->win_lo, *NT = e->NT - e->win_lo;
  double ch, sh, cif[7], tf[5], tbuf[17];
  sdco_hour_sincos(e->hour, &ch, &sh);
  ci_features(e, ip, cif);
  /* temperature slope over [NT[i'], NT[i'+1..i'+16]] (sustaindc_env.py:331) */
  for (int k = 0; k < 17; k++) tbuf[k] = NT[ip + k];
  double tslope = sdco_polyfit_slope(tbuf, 17);
  sdco_extract_features(&NT[ip + 1], 16, NT[ip], tf);
  double soc = e->bat_load / p->bat_capacity; /* battery_model.py:137-139 */
  double o[SDCO_OBS_DIM];
  int k = 0;
  /* agent_ls (26) sustaindc_env.py:342-353 */
  o[k++] = ch; o[k++] = sh; o[k++] = NC[ip];
  for (int j = 0; j < 7; j++) o[k++] = cif[j];
  o[k++] = e->ls_oldest_age; o[k++] = e->ls_avg_age; o[k++] = e->ls_norm_tasks_in_queue;
  o[k++] = W[ip]; o[k++] = NT[ip]; o[k++] = tslope;
  for (int j = 0; j < 5; j++) o[k++] = tf[j];
  for (int j = 0; j < 5; j++) o[k++] = e->ls_hist[j];
  /* agent_dc (14) sustaindc_env.py:386-393 */
  o[k++] = ch; o[k++] = sh; o[k++] = NC[ip];
  for (int j = 0; j < 7; j++) o[k++] = cif[j];
  o[k++] = W[ip]; o[k++] = W[ip + 1]; o[k++] = NT[ip]; o[k++] = NT[ip + 1];
  /* agent_bat (13) sustaindc_env.py:426-432 */
  o[k++] = ch; o[k++] = sh; o[k++] = NC[ip];
  for (int j = 0; j < 7; j++) o[k++] = cif[j];
  o[k++] = W[ip]; o[k++] = NT[ip]; o[k++] = soc;
  for (int j = 0; j < SDCO_OBS_DIM; j++) obs[j] = (float)o[j];
}

/* ------------------------------------------------------------------ data-centre model */

/* envs/datacenter.py:356-429 */
double sdco_chiller_power(double max_cooling_cap, double load, double ambient_temp) {
  const double min_plr = 0.05, max_plr = 1.0, design_cond_temp = 35.0, design_evp_out_temp = 6.67;
  const double temp_rise_coef = 2.778, rated_cop = 3.0;
  double delta_temp = (ambient_temp - design_cond_temp) / temp_rise_coef - (design_evp_out_temp - design_cond_temp);
  double cap_rat = 0.94483600 + -0.05700880 * delta_temp + 0.00185486 * (delta_temp * delta_temp);
  double avail = cap_rat != 0 ? max_cooling_cap * cap_rat : 0.0;
  double fpr = 2.333 + -1.975 * cap_rat + 0.6121 * (cap_rat * cap_rat);
  double plr = avail > 0 ? fmax(min_plr, fmin(load / avail, max_plr)) : 0.0;
  double fflp = 0.03303 + 0.6852 * plr + 0.2818 * (plr * plr);
  double oper;
  if (avail > 0)
    oper = (load / avail < min_plr) ? load / avail : plr;
  else
    oper = 0.0;
  double frac = oper < min_plr ? fmin(1.0, oper / min_plr) : 1.0;
  double power = fflp * fpr * avail / rated_cop * frac;
  return oper > 0 ? power : 0.0;
}

/* IT load + outlet temps (envs/datacenter.py:250-317, :157-181), CRAC return (:531-541),
 * HVAC (:432-474), water (:325-353).
 * out = {P_it_W, CT_W, compressor_W, avg_return, mean_outlet, water_L, Q_cooling_W, sum_outlet} */
void sdco_dc_model(const sdco_params *p, double stpt, double load_pct, double ambient, double wet_bulb,
                   double out[8], unsigned *fault) {
  const double c = 1.918, d = 1.096, ee = 0.824, f = 0.526, g = -14.01;
  double sum_cpu = 0.0, sum_fan = 0.0, sum_ret = 0.0;
  double outlets[SDCO_MAX_RACKS];
  for (int r = 0; r < p->R; r++) {
    double sa = fmax(3.8, fmin(p->rack_supply[r], 5.3)); /* datacenter.py:209-215 */
    double inlet = sa + stpt;
    /* all n_r CPUs of a rack are identical (dc_config_reader.py:97): vector ops collapse to n * per-CPU */
    double ratio = ((p->m_cpu + 0.05) * inlet + p->c_cpu) + p->rs_cpu * (load_pct / 100);
    double cpu1 = fmax(p->rack_idle[r], p->rack_full[r] * ratio);
    double v = (p->m_fan * 10 * inlet + p->c_fan * 5) + p->rs_fan * (load_pct / 20);
    double fan1 = p->itfan_ref_p * (v / p->itfan_ref_v_ratio);
    double vf1 = p->it_fan_full_load_v * v;
    /* np.sum over n identical elements: exact multiples are not guaranteed bitwise, but n*x differs
     * from the pairwise sum by <= 1e-16 relative (documented in DESIGN.md) */
    double pcpu = p->rack_n[r] * cpu1, pfan = p->rack_n[r] * fan1, vtot = p->rack_n[r] * vf1;
    double power_term = pow(pcpu + pfan, d);
    double airflow_term = p->c_air * p->rho_air * pow(vtot, ee) * f;
    double outlet = inlet + c * power_term / airflow_term + g;
    if (outlet - inlet < 2) *fault |= SDCO_FAULT_OUTLET_DELTA;
    sum_cpu += pcpu;
    sum_fan += pfan;
    outlets[r] = outlet;
    sum_ret += p->rack_return[r] + outlet;
  }
  double avg_ret = sum_ret / (double)p->R;
  double p_it = sum_cpu + sum_fan;
  double m_sys = p->rho_air * p->crac_supply_pu * p_it;
  double q = m_sys * p->c_air * fmax(0.0, avg_ret - stpt);
  double comp = sdco_chiller_power(p->ct_fan_ref_p, q, ambient);
  double ct;
  if (ambient < 5) {
    ct = 0.0;
  } else {
    double delta = fmax(50 - (ambient - stpt), 1);
    double m_air = q / (p->c_air * delta);
    double v_air = m_air / p->rho_air;
    double x = fmin(v_air / p->ctafr, 1);
    ct = p->ct_fan_ref_p * (x * x * x);
  }
  double range_temp = avg_ret - stpt;
  double y_int = 0.3528 * range_temp + 0.101;
  double w = 0.044 * wet_bulb + y_int;
  if (w < 0) w = 0;
  w += w * 0.01;
  double water = np_round((w * 1000) / 4, 1e4);
  out[0] = p_it; out[1] = ct; out[2] = comp; out[3] = avg_ret;
  out[4] = np_mean(outlets, p->R); out[5] = water; out[6] = q; out[7] = np_pairwise_sum(outlets, p->R);
}

/* utils/make_envs_pyenv.py:139-197 + envs/datacenter.py:476-529 */
void sdco_size_datacenter(sdco_params *p, double max_amb, double rg[8]) {
  unsigned fault = 0;
  double o[8];
  /* chiller_sizing(min_CRAC=min_temp, max_CRAC=max_temp, max_ambient): IT at 100 % load, stpt = max_temp */
  sdco_dc_model(p, p->max_temp, 100.0, 20.0, 0.0, o, &fault);
  double m_sys = p->rho_air * p->crac_supply_pu * o[0];
  double q = m_sys * p->c_air * fmax(0.0, o[3] - p->min_temp);
  double delta = fmax(50 - (max_amb - p->min_temp), 1);
  double m_air = q / (p->c_air * delta);
  p->ctafr = m_air / p->rho_air;
  p->ct_fan_ref_p = q;
  double it_min = 1e300, it_max = -1e300, t_min = 1e300, t_max = -1e300;
  for (int s = 15; s < 23; s++)
    for (int l = 0; l < 110; l += 10) {
      sdco_dc_model(p, (double)s, (double)l, 20.0, 0.0, o, &fault);
      double mo = o[7] / (double)p->R; /* sum(outlets)/len (python sum) */
      if (o[0] < it_min) it_min = o[0];
      if (o[0] > it_max) it_max = o[0];
      if (mo < t_min) t_min = mo;
      if (mo > t_max) t_max = mo;
    }
  double chiller_max = sdco_chiller_power(p->ct_fan_ref_p, it_max, max_amb);
  double max_dc_power_w = 1.1 * it_max + 1.1 * p->ct_fan_ref_p + 1.1 * chiller_max;
  double max_dc_energy = (max_dc_power_w / 4) * (4 * 1);
  p->bat_capacity = max_dc_energy / 1e6;
  rg[0] = 0.9 * t_min; rg[1] = 1.1 * t_max;                                  /* zone air */
  rg[2] = 0.0; rg[3] = 1.1 * p->ct_fan_ref_p + 1.1 * chiller_max;           /* HVAC */
  rg[4] = 0.9 * it_min; rg[5] = 1.1 * it_max + 1.1 * p->ct_fan_ref_p + 1.1 * chiller_max; /* total */
  rg[6] = 0.9 * it_min; rg[7] = 1.1 * it_max;                               /* IT */
}

/* ------------------------------------------------------------------ env */

void sdco_env_init(sdco_env *e) {
  memset(e, 0, sizeof(*e));
  e->stpt = 18.0; /* utils/make_envs_pyenv.py:124 */
  e->scale = 1;
}

static double q_age(const sdco_env *e, int idx) {
  int j = (e->q_head + idx) % SDCO_QUEUE_CAP;
  return (double)((e->day - e->q_day[j]) * 24) + (e->hour - e->q_hour[j]);
}

void sdco_episode_begin(sdco_env *e, const sdco_params *p, const double *W, const double *C, const double *NC,
                        const double *T, const double *WB, const double *NT, int win_lo, int win_len,
                        int init_day, int init_hour, int episode_steps, float *obs53) {
  e->W = W; e->C = C; e->NC = NC; e->T = T; e->WB = WB; e->NT = NT;
  e->win_lo = win_lo; e->win_len = win_len;
  e->day = init_day;
  e->hour = (double)init_hour;
  e->cursor = init_day * 96 + init_hour * 4; /* managers.py:122 */
  e->t_end = e->cursor + episode_steps;      /* managers.py:123 */
  e->q_head = 0; e->q_len = 0;               /* carbon_ls.py:85 */
  e->ls_norm_tasks_in_queue = 0; e->ls_oldest_age = 0; e->ls_avg_age = 0;
  memset(e->ls_hist, 0, sizeof(e->ls_hist));
  e->has_last_delta = 0; e->consecutive = 0; e->scale = 1; /* dc_gym.py:114-116 (stpt untouched) */
  e->bat_load = 0.0;                                         /* battery_model.py:90-91 */
  if (obs53) build_obs(e, p, obs53);
}

static double sigmoid(double x) { return 1 / (1 + exp(-x)); }

int sdco_step(sdco_env *e, const sdco_params *p, const int32_t act[3], float *obs53, double rew[3],
              double info[SDCO_INFO_DIM]) {
  const int i = e->cursor;
  const double *W = e->W - e->win_lo, *C = e->C - e->win_lo, *NC = e->NC - e->win_lo;
  const double *T = e->T - e->win_lo, *WB = e->WB - e->win_lo;
  unsigned fault = 0;
  for (int k = 0; k < SDCO_INFO_DIM; k++) info[k] = 0.0;

  /* ---- load shifting: envs/carbon_ls.py:172-324 */
  const double wl = W[i];
  if (wl < 0 || wl > 1) fault |= SDCO_FAULT_WORKLOAD;
  const double flex = 0.2; /* class default; make_ls_env never forwards flexible_load (make_envs_pyenv.py:37-41) */
  const double nonflex = 1 - flex;
  int ns = (int)ceil(wl * nonflex * 100);
  int sh = (int)floor(wl * flex * 100);
  int dropped = 0, processed = 0;
  int overdue = 0;
  for (int k = 0; k < e->q_len; k++)
    if (q_age(e, k) > 24) overdue++;
  int avail = 90 - (ns + sh);
  int od_proc = 0;
  if (avail > 0 && overdue > 0) {
    od_proc = overdue < avail ? overdue : avail;
    e->q_head = (e->q_head + od_proc) % SDCO_QUEUE_CAP; /* overdue tasks are the FIFO prefix */
    e->q_len -= od_proc;
  }
  avail = 90 - (ns + sh + od_proc);
  double util;
  if (act[0] == 0) {
    int room = p->queue_max_len - e->q_len;
    int add = sh < room ? sh : room;
    dropped += sh - add;
    for (int k = 0; k < add; k++) {
      int j = (e->q_head + e->q_len) % SDCO_QUEUE_CAP;
      e->q_day[j] = e->day;
      e->q_hour[j] = e->hour;
      e->q_len++;
    }
    util = (double)(od_proc + (sh - add)) / 100;
  } else if (act[0] == 2) {
    if (avail >= 1) {
      int tp = sh;
      if (avail < tp) tp = avail;
      if (e->q_len < tp) tp = e->q_len;
      processed = tp;
      e->q_head = (e->q_head + tp) % SDCO_QUEUE_CAP;
      e->q_len -= tp;
      util = (double)(sh + tp + od_proc) / 100;
    } else {
      util = (double)(sh + od_proc) / 100;
    }
  } else {
    util = (double)(sh + od_proc) / 100;
  }
  util += (double)ns / 100;
  double oldest = 0.0, avg = 0.0, hist[5] = {0, 0, 0, 0, 0};
  if (e->q_len > 0) {
    double sum = 0.0;
    for (int k = 0; k < e->q_len; k++) {
      double a = q_age(e, k);
      if (a > oldest) oldest = a;
      sum += a;
      int b = a < 6 ? 0 : a < 12 ? 1 : a < 18 ? 2 : a < 24 ? 3 : 4; /* np.histogram bins [0,6,12,18,24,inf] */
      hist[b] += 1;
    }
    avg = sum / (double)e->q_len;
  }
  {
    double den = (double)(e->q_len > 1 ? e->q_len : 1);
    for (int b = 0; b < 5; b++) hist[b] = hist[b] / den;
    hist[4] = hist[4] > 0 ? 1 : 0; /* carbon_ls.py:72 */
  }
  e->ls_norm_tasks_in_queue = (double)e->q_len / (double)p->queue_max_len;
  e->ls_oldest_age = oldest / 24;
  e->ls_avg_age = avg / 24;
  memcpy(e->ls_hist, hist, sizeof(hist));
  info[SDCO_I_LS_ORIGINAL_WORKLOAD] = wl;
  info[SDCO_I_LS_SHIFTED_WORKLOAD] = util;
  info[SDCO_I_LS_TASKS_IN_QUEUE] = e->q_len;
  info[SDCO_I_LS_NORM_TASKS_IN_QUEUE] = e->ls_norm_tasks_in_queue;
  info[SDCO_I_LS_TASKS_DROPPED] = dropped;
  info[SDCO_I_LS_TASKS_PROCESSED] = processed;
  info[SDCO_I_LS_OLDEST_TASK_AGE] = e->ls_oldest_age;
  info[SDCO_I_LS_AVERAGE_TASK_AGE] = e->ls_avg_age;
  info[SDCO_I_LS_OVERDUE_PENALTY] = overdue;
  info[SDCO_I_LS_COMPUTED_TASKS] = (double)(int)(util * 100);
  info[SDCO_I_LS_CURRENT_HOUR] = e->hour;
  for (int b = 0; b < 5; b++) info[SDCO_I_LS_HIST0 + b] = hist[b];

  /* ---- data centre: envs/dc_gym.py:142-237 */
  if (util < 0.0 || util > 1.0) fault |= SDCO_FAULT_CPU_LOAD; /* dc_gym.py:288-290 */
  int delta = act[1] - 1;                                     /* make_envs_pyenv.py:127-131 */
  if (e->has_last_delta && delta == e->last_delta && act[1] != 0) {
    e->consecutive += 1;
  } else {
    e->consecutive = 1;
    e->scale = 1;
  }
  if (e->consecutive > 3) e->scale += 1;
  e->stpt += (double)(delta * e->scale);
  e->stpt = fmax(fmin(e->stpt, p->max_temp), p->min_temp);
  double dc[8];
  sdco_dc_model(p, e->stpt, util * 100, T[i], WB[i], dc, &fault);
  e->has_last_delta = 1;
  e->last_delta = delta;
  double total_kw = (dc[0] + dc[1] + dc[2]) / 1e3;
  info[SDCO_I_DC_ITE_KW] = dc[0] / 1e3;
  info[SDCO_I_DC_CT_KW] = dc[1] / 1e3;
  info[SDCO_I_DC_COMPRESSOR_KW] = dc[2] / 1e3;
  info[SDCO_I_DC_HVAC_KW] = (dc[1] + dc[2]) / 1e3;
  info[SDCO_I_DC_TOTAL_KW] = total_kw;
  info[SDCO_I_DC_SETPOINT_DELTA] = delta;
  info[SDCO_I_DC_SETPOINT] = e->stpt;
  info[SDCO_I_DC_CPU_FRACTION] = util;
  info[SDCO_I_DC_INT_TEMPERATURE] = dc[4];
  info[SDCO_I_DC_AMBIENT_TEMP] = T[i];
  info[SDCO_I_DC_WATER_USAGE] = dc[5];

  /* ---- battery: envs/bat_env_fwd_view.py:84-245, envs/battery_model.py:94-132 */
  const double cap = p->bat_capacity;
  const double dcload = total_kw / 1e3; /* MW (sustaindc_env.py:652) */
  const double ci = C[i];
  double energy, co2;
  if (act[2] == 0) { /* charge */
    double soc = (e->bat_load - 0) / (cap - 0);
    double rate = np_round(0.5 * (1 - sigmoid(10 * (soc - 0.5))), 1e4);
    double tu = rate * 15 / 60;
    double max_charge = fmin((cap / 1) * 0.1, (1 * cap - e->bat_load) / ((1 * tu) - (-0.04)));
    double charging_load = fmin(max_charge, cap) * 1 * tu;
    e->bat_load = np_round(e->bat_load + charging_load, 1e8);
    energy = dcload * 1e3 * 0.25 + charging_load * 1e3;
    co2 = energy * ci;
  } else if (act[2] == 1) { /* discharge */
    double soc = (e->bat_load - 0) / (cap - 0);
    double rate = fmax(0.5, 4 * sigmoid(10 * (soc - 0.25)));
    double tu = rate * 15 / 60;
    double max_d = fmin(fmin((cap / 1) * 1, (e->bat_load - 0 * cap) / (0.01 + (1 * tu))), dcload / 4);
    e->bat_load = np_round(e->bat_load - (fmin(max_d, cap) * 1 * tu), 1e8);
    double discharge = max_d < cap ? max_d * tu : cap * tu;
    if (!(dcload * 1e3 * 0.25 >= discharge * 1e3)) fault |= SDCO_FAULT_BAT_DISCHARGE;
    energy = dcload * 1e3 * 0.25 - discharge * 1e3;
    co2 = fmax(energy, 0) * ci;
  } else { /* idle */
    energy = dcload * 1e3 * 0.25;
    co2 = energy * ci;
  }
  info[SDCO_I_BAT_ACTION] = act[2];
  info[SDCO_I_BAT_SOC] = e->bat_load / cap;
  info[SDCO_I_BAT_CO2] = co2;
  info[SDCO_I_BAT_AVG_CI] = ci;
  info[SDCO_I_BAT_ENERGY_WITHOUT_KWH] = dcload * 1e3 * 0.25;
  info[SDCO_I_BAT_ENERGY_WITH_KWH] = energy;

  /* ---- managers step: utils/managers.py:127-147 (+ :285, :452, :633: cursor += 1) */
  e->cursor += 1;
  e->hour += 1.0 / 4;
  if (e->hour >= 24) {
    e->hour = 0;
    e->day += 1;
  }
  int terminal = e->cursor >= e->t_end;
  const int ip = e->cursor;

  /* ---- obs at i' (sustaindc_env.py:565-585) */
  if (obs53) build_obs(e, p, obs53);

  /* ---- rewards: sustaindc_env.py:676-737, utils/reward_creator.py:48-130 */
  if (p->reward_method[0] == 0) { /* only default_ls_reward appends (reward_creator.py:63) */
    if (e->hist_len < SDCO_HIST_CAP) {
      e->hist[e->hist_len++] = energy;
    } else {
      e->hist[e->hist_pos] = energy;
      e->hist_pos = (e->hist_pos + 1) % SDCO_HIST_CAP;
    }
  }
  /* numpy sees the deque in insertion order; order only affects the pairwise-sum tree (<=1e-16) */
  double z = normalize_energy_scratch(e->hist, e->hist_len, energy, e->scratch);
  double norm_ci = NC[ip + 1];
  double foot = -1.0 * (norm_ci * z / 0.50);
  double overdue_pen = -0.3 * sqrt((double)overdue) + 0.3;
  double age_pen = -0.1 * e->ls_oldest_age;
  double rls = foot + overdue_pen + age_pen;
  rls = rls < -10 ? -10 : (rls > 10 ? 10 : rls);
  for (int a = 0; a < 3; a++) {
    double r;
    switch (p->reward_method[a]) {
      case 0: r = a == 0 ? rls : foot; break;
      case 1: r = foot; break;
      case 2: r = 0.0; break; /* custom_agent_reward, reward_creator.py:133-146 */
      case 3: { /* tou_reward, reward_creator.py:154-198.  DEVIATION: the reference indexes its price table with the
                   float hour and raises KeyError off the full hour; here the hour is truncated. */
        static const double tou[24] = {0.25, 0.25, 0.25, 0.25, 0.25, 0.25, 0.41, 0.41, 0.41, 0.41, 0.41, 0.30,
                                       0.30, 0.30, 0.30, 0.30, 0.27, 0.27, 0.27, 0.27, 0.27, 0.27, 0.25, 0.25};
        r = -1.0 * energy * tou[(int)e->hour % 24];
        break;
      }
      case 4: r = info[SDCO_I_DC_ITE_KW] / info[SDCO_I_DC_TOTAL_KW]; break; /* energy_efficiency_reward :223-239 */
      case 5: { /* energy_PUE_reward :242-265 */
        const double it = info[SDCO_I_DC_ITE_KW];
        const double pue = it != 0 ? info[SDCO_I_DC_TOTAL_KW] / it : INFINITY;
        r = -fabs(pue - 1);
        break;
      }
      case 6: r = -0.01 * info[SDCO_I_DC_WATER_USAGE]; break; /* water_usage_efficiency_reward :297-318 */
      default: r = 0.0;
    }
    rew[a] = r;
  }
  info[SDCO_I_NORM_CI] = norm_ci;
  info[SDCO_I_OUTSIDE_TEMP] = T[ip];
  info[SDCO_I_DAY] = e->day;
  info[SDCO_I_HOUR] = e->hour;
  info[SDCO_I_FAULT] = (double)fault;
  return terminal;
}

long sdco_run_steps(sdco_env *e, const sdco_params *p, const int32_t *actions, long nsteps, int episode_steps,
                    int init_day, int init_hour, double *rew_sum3) {
  float obs[SDCO_OBS_DIM];
  double rew[3], info[SDCO_INFO_DIM];
  double acc[3] = {0, 0, 0};
  for (long s = 0; s < nsteps; s++) {
    int done = sdco_step(e, p, &actions[3 * s], obs, rew, info);
    acc[0] += rew[0]; acc[1] += rew[1]; acc[2] += rew[2];
    if (done)
      sdco_episode_begin(e, p, e->W, e->C, e->NC, e->T, e->WB, e->NT, e->win_lo, e->win_len, init_day,
                         init_hour, episode_steps, obs);
  }
  if (rew_sum3) { rew_sum3[0] = acc[0]; rew_sum3[1] = acc[1]; rew_sum3[2] = acc[2]; }
  return nsteps;
}
