// sdc_dynamics.hip -- coupled per-timestep dynamics, ONE WAVEFRONT PER ENVIRONMENT (block = 64 lanes).
//
// Memory plan (two dependent round trips per step instead of one per field):
//   level 0  the env's 256-byte state record, one dword per lane (coalesced), + the 3 actions;
//   level 1  ONE 8-byte gather per lane: the workload / carbon / weather samples of the step, the 25-sample CI
//            and 17-sample temperature observation windows, the 5 queue prefix-count probes and the hour LUT
//            entry -- staged through LDS and read back wave-uniformly; the per-rack constants (lane = rack);
//   (level 2 only on steps that pop tasks: the 64-ary search for the new oldest task in the queue.)
// Compute:
//   * the load-shifting queue is O(1) prefix-count algebra (see below);
//   * the rack model runs lane = rack with wave-shuffle (DPP) reductions for total IT power, CRAC return and
//     outlet temperature;
//   * chiller / cooling tower / water / battery / set-point integrator are wave-uniform scalar fp64;
//   * the observation features (3 least-squares slopes, 2 x mean/std/peak/valley) run lane-parallel with segmented
//     butterfly reductions; lane 0 assembles the info block and the new state record in LDS; all lanes store
//     obs / share_obs / info / record coalesced.
//   * rewards: the step's energy is appended to the env's history ring, and the history-normalised z-score comes
//     from the env's reward state (four 64-key rank windows, running sums: sdc_trackers.hpp) in O(1) without
//     reading the ring; the rare ring sweeps that keep that state ahead of need run in-wave (sdc_ringpath.hpp).
// One launch of this kernel is one env-step of all N environments.
//
// Reference: sustaindc_env.py:533-737 and the sub-environment steps it drives (see per-block citations).
#include "sdc_ringpath.hpp"

namespace {

// gather slots (one 8-byte load per lane)
enum {
  G_W0 = 0, G_W1, G_W2,   // W[i], W[i+1], W[i+2]
  G_C0,                   // C[i]
  G_T0, G_WB0, G_T1,      // T[i], WB[i], T[i+1] from the env's weather window
  G_LUT,                  // hour LUT {cos, sin} is 16 bytes: two slots
  G_LUT2,
  G_Q97, G_Q24, G_Q48, G_Q72, G_Q96,   // queue prefix counts cum[now - a]
  G_NC = 16,              // 25 slots: C[i'-16 .. i'+8]
  G_NT = 41,              // 17 slots: T[i' .. i'+16]
  G_END = 58
};

struct DynShared {
  double g[64];     // gathered raw values; g[G_NC..] / g[G_NT..] are normalised in place to NC / NT
  float pool[32];   // observation pool (see build_obs_pool)
  float info[SDC_INFO_DIM];
  unsigned rec[SDC_REC_DWORDS];
  unsigned long long dbg_t;
  sdc_rw::TailLds tl;   // scratch of the ring paths (window refill, rebuild)
};

// envs/datacenter.py:356-429 calculate_chiller_power
__device__ __forceinline__ double chiller_power(double max_cooling_cap, double load, double ambient_temp) {
  const double min_plr = 0.05, max_plr = 1.0, design_cond_temp = 35.0, design_evp_out_temp = 6.67;
  // temp_rise_coef = 2.778, rated_cop = 3.0 (divisors below)
  const double delta_temp = SDC_DIV_CONST(ambient_temp - design_cond_temp, 2.778) - (design_evp_out_temp - design_cond_temp);
  const double cap_rat = 0.94483600 + -0.05700880 * delta_temp + 0.00185486 * (delta_temp * delta_temp);
  const double avail = cap_rat != 0 ? max_cooling_cap * cap_rat : 0.0;
  const double fpr = 2.333 + -1.975 * cap_rat + 0.6121 * (cap_rat * cap_rat);
  const double plr = avail > 0 ? fmax(min_plr, fmin(load / avail, max_plr)) : 0.0;
  const double fflp = 0.03303 + 0.6852 * plr + 0.2818 * (plr * plr);
  double oper;
  if (avail > 0)
    oper = (load / avail < min_plr) ? load / avail : plr;
  else
    oper = 0.0;
  const double frac = oper < min_plr ? fmin(1.0, SDC_DIV_CONST(oper, 0.05)) : 1.0;   // / min_plr
  const double power = SDC_DIV_CONST(fflp * fpr * avail, 3.0) * frac;
  return oper > 0 ? power : 0.0;
}

__device__ __forceinline__ double sigmoid(double x) { return 1 / (1 + exp(-x)); }

// ------------------------------------------------------------------------------------------------
// the coupled dynamics at cursor i and the observation at i' = i + 1; one wavefront, lane in [0, 64)
__device__ __forceinline__ void step_dynamics(const SdcDev& S, const SdcDcDev& PD, const int env, const int lane,
                                              const unsigned r, const int a_ls, const int a_dc, const int a_bat,
                                              unsigned fault, const unsigned x_old_l, const unsigned hd0, const uint4 qw, const bool feat_ok, const float frow,
                                              float* __restrict__ rew,
                                              DynShared& sh) {
  const sdc_dc_params& P = PD.p;
  const int i = rec_i32(r, R_CURSOR);
  const int rel = rec_i32(r, R_TREL);
  const int day = rec_i32(r, R_DAY);
  const int hourq = rec_i32(r, R_HOURQ);
  const double hour = (double)hourq * 0.25;
  const double wl = sh.g[G_W0], w_ip = sh.g[G_W1], w_ip1 = sh.g[G_W2];
  const double ci_i = sh.g[G_C0];
  const double amb = sh.g[G_T0], wet_bulb = sh.g[G_WB0], amb_next = sh.g[G_T1];
  const double* nc = sh.g + G_NC;
  const double* nt = sh.g + G_NT;
  // norm_CI = NC[i'+1] (sustaindc_env.py:681): from the episode's feature row (a double in its last two floats), or
  // from the window gathered by this step
  const double norm_ci = feat_ok ? __hiloint2double(__builtin_amdgcn_readlane(__float_as_int(frow), SDC_FEAT_NCNEXT + 1),
                                                    __builtin_amdgcn_readlane(__float_as_int(frow), SDC_FEAT_NCNEXT))
                                 : nc[17];

  // ---- load shifting: envs/carbon_ls.py:172-324 ------------------------------------------------
  // The reference keeps a deque of per-task enqueue timestamps and only ever removes a FIFO prefix
  // (overdue `remove()` loop :225-226 and popleft :257-258).  Equivalent state: cum[t] = tasks ever
  // enqueued up to step t of the episode, popped = tasks ever removed.  Tasks still queued that were
  // enqueued at or before step t: max(0, cum[t] - popped).
  if (wl < 0 || wl > 1) fault |= SDC_FAULT_WORKLOAD;
  const double flex = 0.2;        // class default; make_ls_env never forwards flexible_load (make_envs_pyenv.py:37-41)
  const double nonflex = 1 - flex;
  const int ns = (int)ceil(wl * nonflex * 100);
  const int shf = (int)floor(wl * flex * 100);
  const uint2* qt = S.qtab + (size_t)env * S.qstride;
  const int now = rel;
  const int popped0 = rec_i32(r, R_QPOPPED);
  int popped = popped0;
  const int cum_prev = rec_i32(r, R_QCUM);
  const unsigned cumT_prev = (unsigned)rec_i32(r, R_QCUMT);
  auto cum_g = [&](int slot) -> int { return (int)(unsigned)__double2loint(sh.g[slot]); };  // .x of the gathered uint2 (0 if t < 0)
  // overdue: age > 24 h  <=>  enqueued at step <= now - 97  (carbon_ls.py:208)
  const int overdue = max(0, cum_g(G_Q97) - popped);
  int avail = 90 - (ns + shf);
  int od_proc = 0;
  if (avail > 0 && overdue > 0) od_proc = min(overdue, avail);
  popped += od_proc;
  avail = 90 - (ns + shf + od_proc);
  int add = 0, dropped = 0, processed = 0;
  double util;
  if (a_ls == 0) {
    const int room = S.queue_max - (cum_prev - popped);
    add = min(shf, room);
    dropped = shf - add;
    util = SDC_DIV_CONST((double)(od_proc + (shf - add)), 100);
  } else if (a_ls == 2) {
    if (avail >= 1) {
      processed = min(min(shf, avail), cum_prev - popped);
      popped += processed;
      util = SDC_DIV_CONST((double)(shf + processed + od_proc), 100);
    } else {
      util = SDC_DIV_CONST((double)(shf + od_proc), 100);
    }
  } else {
    util = SDC_DIV_CONST((double)(shf + od_proc), 100);
  }
  util += SDC_DIV_CONST((double)ns, 100);
  const int cum_now = cum_prev + add;
  const unsigned cumT_now = cumT_prev + (unsigned)add * (unsigned)now;
  const int total = cum_now - popped;
  // age histogram, bins [0,6,12,18,24,inf] hours = [0,24,48,72,96,inf) steps (carbon_ls.py:63-73)
  const int a24 = max(0, cum_g(G_Q24) - popped), a48 = max(0, cum_g(G_Q48) - popped);
  const int a72 = max(0, cum_g(G_Q72) - popped), a96 = max(0, cum_g(G_Q96) - popped);
  double hist[5];
  const double den = (double)max(total, 1), rden = 1.0 / den;   // (an integer <= 1000: significand never all ones)
  {
    // four divisions by the same count: one reciprocal, then the exact 3-instruction form (sdc_div_const)
    hist[0] = sdc_div_const((double)(total - a24), den, rden);
    hist[1] = sdc_div_const((double)(a24 - a48), den, rden);
    hist[2] = sdc_div_const((double)(a48 - a72), den, rden);
    hist[3] = sdc_div_const((double)(a72 - a96), den, rden);
    hist[4] = a96 > 0 ? 1.0 : 0.0;
  }
  // oldest task: smallest step h in [head, now] with cum[h] > popped.  It only moves when tasks were popped
  // (or the queue was empty): then a 64-ary wave search (<= 2 rounds) finds it and cum/cumT[h-1] are cached.
  int head = rec_i32(r, R_QHEAD);
  int cum_hm1 = rec_i32(r, R_QCUM_HM1);
  unsigned cumT_hm1 = (unsigned)rec_i32(r, R_QCUMT_HM1);
  double oldest = 0.0, avg = 0.0;
  if (total > 0) {
    const bool was_empty = (cum_prev - popped0) == 0;
    if (was_empty) {          // everything queued was enqueued now
      head = now;
      cum_hm1 = cum_prev;
      cumT_hm1 = cumT_prev;
    } else if (popped != popped0) {
      int lo = head, hi = now;
      while (hi - lo + 1 > SDC_WAVE) {
        const int len = hi - lo + 1;
        const int stride = (len + SDC_WAVE - 1) / SDC_WAVE;
        const int t = min(lo + (lane + 1) * stride - 1, hi);
        const int c = (t == now) ? cum_now : (int)qt[t].x;
        const unsigned long long m = __ballot(c > popped);
        const int f = __ffsll((long long)m) - 1;  // exists: cum[now] > popped
        const int nlo = lo + f * stride;
        hi = min(lo + (f + 1) * stride - 1, hi);
        lo = nlo;
      }
      {
        const int t = lo + lane;
        int c = 0;
        if (t <= hi) c = (t == now) ? cum_now : (int)qt[t].x;
        const unsigned long long m = __ballot(t <= hi && c > popped);
        head = lo + (__ffsll((long long)m) - 1);
      }
      if (head == 0) {
        cum_hm1 = 0;
        cumT_hm1 = 0;
      } else if (head == now) {
        cum_hm1 = cum_prev;
        cumT_hm1 = cumT_prev;
      } else {
        const uint2 e = qt[head - 1];
        cum_hm1 = (int)e.x;
        cumT_hm1 = e.y;
      }
    }
    // sum of enqueue steps over the queued tasks = cumT[now] - cumT[h-1] - (popped - cum[h-1]) * h
    const long long sum_t = (long long)cumT_now - (long long)cumT_hm1 - (long long)(popped - cum_hm1) * head;
    const long long sum_age_steps = (long long)total * now - sum_t;
    oldest = (double)(now - head) * 0.25;                  // hours, exact
    avg = sdc_div_const((double)sum_age_steps * 0.25, den, rden);  // / total (> 0 here); sum(ages) is exact in the reference too
  } else {
    head = now;
    cum_hm1 = cum_now;
    cumT_hm1 = cumT_now;
  }
  const double normq = sdc_div_const((double)total, (double)S.queue_max, S.rc_queue_max);
  const double oldest_norm = SDC_DIV_CONST(oldest, 24), avg_norm = SDC_DIV_CONST(avg, 24);

  // ---- CRAC set-point integrator: envs/dc_gym.py:160-174 ----------------------------------------
  if (util < 0.0 || util > 1.0) fault |= SDC_FAULT_CPU_LOAD;
  const int delta = a_dc - 1;  // make_envs_pyenv.py:127-131
  int last_delta = rec_i32(r, R_LAST_DELTA), consecutive = rec_i32(r, R_CONSEC), scale = rec_i32(r, R_SCALE);
  if (last_delta != -2 && delta == last_delta && a_dc != 0) {
    consecutive += 1;
  } else {
    consecutive = 1;
    scale = 1;
  }
  if (consecutive > 3) scale += 1;
  double stpt = rec_f64(r, R_STPT) + (double)(delta * scale);
  stpt = fmax(fmin(stpt, P.max_temp), P.min_temp);

  // ---- rack model, lane = rack: envs/datacenter.py:250-317, :157-181 ------------------------------
  const int R = P.n_racks;
  const double load_pct = util * 100;
  double pcpu = 0.0, pfan = 0.0, outlet = 0.0, ret_plus_out = 0.0;
  unsigned bad_delta = 0;
  if (lane < R) {
    const double sa = fmax(3.8, fmin(P.rack_supply[lane], 5.3));  // datacenter.py:209-215
    const double inlet = sa + stpt;
    const double ratio = ((P.m_cpu + 0.05) * inlet + P.c_cpu) + P.rs_cpu * SDC_DIV_CONST(load_pct, 100);
    const double cpu1 = fmax(P.rack_idle[lane], P.rack_full[lane] * ratio);
    const double v = (P.m_fan * 10 * inlet + P.c_fan * 5) + P.rs_fan * SDC_DIV_CONST(load_pct, 20);
    const double fan1 = P.itfan_ref_p * sdc_div_const(v, P.itfan_ref_v_ratio, PD.rc_itfan_ref_v_ratio);
    const double vf1 = P.it_fan_full_load_v * v;
    const double n = P.rack_n[lane];
    pcpu = n * cpu1;
    pfan = n * fan1;
    const double vtot = n * vf1;
    // x^y as exp2(y log2 x): <= 3e-15 relative against the correctly rounded power (the reference's libm pow is
    // <= 1.3e-16), nine orders below the fp32 outputs' resolution, at less than half the instructions of pow()
    // ... and power^1.096 / airflow^0.824 as ONE exp2 of the difference of the two scaled logarithms
    const double rise = exp2(1.096 * log2(pcpu + pfan) - 0.824 * log2(vtot));
    outlet = inlet + PD.k_outlet * rise + -14.01;   // 1.918 power^1.096 / (c_air rho_air airflow^0.824 0.526) - 14.01
    if (outlet - inlet < 2) bad_delta = 1;
    ret_plus_out = P.rack_return[lane] + outlet;
  }
  if (__ballot(bad_delta) != 0ull) fault |= SDC_FAULT_OUTLET_DELTA;
  const double sum_cpu = wave_sum_f64(pcpu), sum_fan = wave_sum_f64(pfan);
  const double avg_ret = sdc_div_const(wave_sum_f64(ret_plus_out), (double)R, PD.rc_n_racks);  // datacenter.py:531-541
  const double mean_outlet = sdc_div_const(wave_sum_f64(outlet), (double)R, PD.rc_n_racks);
  const double p_it = sum_cpu + sum_fan;

  // ---- HVAC: envs/datacenter.py:432-474 ; water :325-353 ------------------------------------------
  const double m_sys = P.rho_air * P.crac_supply_pu * p_it;
  const double q_cool = m_sys * P.c_air * fmax(0.0, avg_ret - stpt);
  const double comp = chiller_power(P.ct_fan_ref_p, q_cool, amb);
  double ct;
  if (amb < 5) {
    ct = 0.0;
  } else {
    const double dlt = fmax(50 - (amb - stpt), 1);
    const double m_air = q_cool / (P.c_air * dlt);
    const double v_air = sdc_div_const(m_air, P.rho_air, PD.rc_rho_air);
    const double x = fmin(sdc_div_const(v_air, P.ctafr, PD.rc_ctafr), 1);
    ct = P.ct_fan_ref_p * (x * x * x);
  }
  double water;
  {
    const double range_temp = avg_ret - stpt;
    const double y_int = 0.3528 * range_temp + 0.101;
    double w = 0.044 * wet_bulb + y_int;
    if (w < 0) w = 0;
    w += w * 0.01;
    water = np_round((w * 1000) / 4, 1e4);
  }
  const double total_kw = SDC_DIV_CONST(p_it + ct + comp, 1e3);

  // ---- battery: envs/bat_env_fwd_view.py:84-245, envs/battery_model.py:94-132 ----------------------
  const double cap = P.bat_capacity_mwh;
  const double dcload = SDC_DIV_CONST(total_kw, 1e3);  // MW (sustaindc_env.py:652)
  double bat_load = rec_f64(r, R_BAT);
  double energy, co2;
  if (a_bat == 0) {  // charge
    const double soc = sdc_div_const(bat_load - 0, cap - 0, PD.rc_bat_capacity);
    const double rate = np_round(0.5 * (1 - sigmoid(10 * (soc - 0.5))), 1e4);
    const double tu = SDC_DIV_CONST(rate * 15, 60);
    const double max_charge = fmin((cap / 1) * 0.1, (1 * cap - bat_load) / ((1 * tu) - (-0.04)));
    const double charging_load = fmin(max_charge, cap) * 1 * tu;
    bat_load = np_round(bat_load + charging_load, 1e8);
    energy = dcload * 1e3 * 0.25 + charging_load * 1e3;
    co2 = energy * ci_i;
  } else if (a_bat == 1) {  // discharge
    const double soc = sdc_div_const(bat_load - 0, cap - 0, PD.rc_bat_capacity);
    const double rate = fmax(0.5, 4 * sigmoid(10 * (soc - 0.25)));
    const double tu = SDC_DIV_CONST(rate * 15, 60);
    const double max_d = fmin(fmin((cap / 1) * 1, (bat_load - 0 * cap) / (0.01 + (1 * tu))), dcload / 4);
    bat_load = np_round(bat_load - (fmin(max_d, cap) * 1 * tu), 1e8);
    const double discharge = max_d < cap ? max_d * tu : cap * tu;
    if (!(dcload * 1e3 * 0.25 >= discharge * 1e3)) fault |= SDC_FAULT_BAT_DISCHARGE;
    energy = dcload * 1e3 * 0.25 - discharge * 1e3;
    co2 = fmax(energy, 0.0) * ci_i;
  } else {  // idle
    energy = dcload * 1e3 * 0.25;
    co2 = energy * ci_i;
  }
  const double soc_after = sdc_div_const(bat_load, cap, PD.rc_bat_capacity);

  // ---- time: utils/managers.py:127-147 -------------------------------------------------------------
  int hourq_n = hourq + 1, day_n = day;
  if (hourq_n >= 96) {
    hourq_n = 0;
    day_n += 1;
  }
  const int ip = i + 1;

  // ---- observations at i' (sustaindc_env.py:565-585): all lanes cooperate -------------------------------------
  {
    ObsScalars o;
    o.cos_h = sh.g[G_LUT];
    o.sin_h = sh.g[G_LUT2];
    o.w_cur = w_ip;
    o.w_next = w_ip1;
    o.soc = soc_after;
    o.normq = normq;
    o.oldest = oldest_norm;
    o.avg = avg_norm;
    for (int b = 0; b < 5; b++) o.hist[b] = hist[b];
    o.have_past = ip >= 16;
    if (__builtin_expect(feat_ok, 1)) {
      // the trace-only entries come from the episode's feature row (sdc_features.hip), one float per lane; lane 1
      // adds the nine entries that depend on the step
      constexpr unsigned TRACE_ONLY = 0x7u | (0x7Fu << SDC_P_CI7) | (1u << SDC_P_W) | (1u << SDC_P_NT) | (1u << SDC_P_TSLOPE) |
                                      (0x1Fu << SDC_P_T5) | (1u << SDC_P_WNEXT) | (1u << SDC_P_NTNEXT);
      if (lane < SDC_POOL_DIM && ((TRACE_ONLY >> lane) & 1u)) sh.pool[lane] = frow;
      if (lane == 1) {
        sh.pool[SDC_P_OLDEST] = (float)o.oldest;
        sh.pool[SDC_P_AVG] = (float)o.avg;
        sh.pool[SDC_P_NORMQ] = (float)o.normq;
        for (int b = 0; b < 5; b++) sh.pool[SDC_P_HIST + b] = (float)o.hist[b];
        sh.pool[SDC_P_SOC] = (float)o.soc;
      }
    } else {
      build_obs_pool(nc, nt, o, sh.pool, lane);
    }
  }

  // ---- history append (utils/reward_creator.py:7-14) --------------------------------------------------------
  // The ring holds fp32 OFFSETS from the env's first energy value (kept in fp64): normalize_energy is
  // shift-invariant, and offsets keep the fp32 rounding error proportional to the spread of the history
  // instead of to the ~300 kWh magnitude (two nearly equal energies would otherwise lose the z-score).
  // Stored as order-preserving keys for the order-statistic trackers.
  // As in the reference, only default_ls_reward appends (reward_creator.py:63): with another ls reward method the
  // history stays as it is and the other agents' footprint rewards are normalised against it.
  const bool append = S.reward_method[0] == SDC_REWARD_DEFAULT;
  int hl = rec_i32(r, R_HIST_LEN), hpos = rec_i32(r, R_HIST_POS);
  const double href = hl == 0 ? energy : rec_f64(r, R_HIST_REF);
  const double e_off = energy - href;
  int slot;
  if (!append) {
    slot = -1;
  } else if (hl < S.hist_cap) {
    slot = hl;
    hl += 1;
  } else {
    slot = hpos;
    hpos = hpos + 1 == S.hist_cap ? 0 : hpos + 1;
  }
  const unsigned x_new = sdc_rw::sfl(sdc_f32_key(__float_as_uint((float)e_off)));

  // every lane keeps its own dword of the record; lane 0 patches the fields that changed (below)
  sh.rec[lane] = r;
  wave_sync();
  if (lane == 0) {
    // ---- info block --------------------------------------------------------------------------------
    float* inf = sh.info;
    inf[SDC_INFO_LS_ORIGINAL_WORKLOAD] = (float)wl;
    inf[SDC_INFO_LS_SHIFTED_WORKLOAD] = (float)util;
    inf[SDC_INFO_LS_TASKS_IN_QUEUE] = (float)total;
    inf[SDC_INFO_LS_NORM_TASKS_IN_QUEUE] = (float)normq;
    inf[SDC_INFO_LS_TASKS_DROPPED] = (float)dropped;
    inf[SDC_INFO_LS_TASKS_PROCESSED] = (float)processed;
    inf[SDC_INFO_LS_OLDEST_TASK_AGE] = (float)oldest_norm;
    inf[SDC_INFO_LS_AVERAGE_TASK_AGE] = (float)avg_norm;
    inf[SDC_INFO_LS_OVERDUE_PENALTY] = (float)overdue;
    inf[SDC_INFO_LS_COMPUTED_TASKS] = (float)(int)(util * 100);
    inf[SDC_INFO_LS_CURRENT_HOUR] = (float)hour;
    for (int b = 0; b < 5; b++) inf[SDC_INFO_LS_AGE_HIST0 + b] = (float)hist[b];
    inf[SDC_INFO_DC_ITE_TOTAL_POWER_KW] = (float)SDC_DIV_CONST(p_it, 1e3);
    inf[SDC_INFO_DC_CT_TOTAL_POWER_KW] = (float)SDC_DIV_CONST(ct, 1e3);
    inf[SDC_INFO_DC_COMPRESSOR_TOTAL_POWER_KW] = (float)SDC_DIV_CONST(comp, 1e3);
    inf[SDC_INFO_DC_HVAC_TOTAL_POWER_KW] = (float)SDC_DIV_CONST(ct + comp, 1e3);
    inf[SDC_INFO_DC_TOTAL_POWER_KW] = (float)total_kw;
    inf[SDC_INFO_DC_CRAC_SETPOINT_DELTA] = (float)delta;
    inf[SDC_INFO_DC_CRAC_SETPOINT] = (float)stpt;
    inf[SDC_INFO_DC_CPU_WORKLOAD_FRACTION] = (float)util;
    inf[SDC_INFO_DC_INT_TEMPERATURE] = (float)mean_outlet;
    inf[SDC_INFO_DC_EXTERIOR_AMBIENT_TEMP] = (float)amb;
    inf[SDC_INFO_DC_WATER_USAGE] = (float)water;
    inf[SDC_INFO_BAT_ACTION] = (float)a_bat;
    inf[SDC_INFO_BAT_SOC] = (float)soc_after;
    inf[SDC_INFO_BAT_CO2_FOOTPRINT] = (float)co2;
    inf[SDC_INFO_BAT_AVG_CI] = (float)ci_i;
    inf[SDC_INFO_BAT_TOTAL_ENERGY_WITHOUT_BATTERY_KWH] = (float)(dcload * 1e3 * 0.25);
    inf[SDC_INFO_BAT_TOTAL_ENERGY_WITH_BATTERY_KWH] = (float)energy;
    inf[SDC_INFO_NORM_CI] = (float)norm_ci;
    inf[SDC_INFO_OUTSIDE_TEMP] = (float)amb_next;
    inf[SDC_INFO_DAY] = (float)day_n;
    inf[SDC_INFO_HOUR] = (float)((double)hourq_n * 0.25);
    const unsigned f_all = (unsigned)rec_i32(r, R_FAULT) | fault;
    inf[SDC_INFO_FAULT] = (float)f_all;
    inf[SDC_INFO_ENERGY_Z] = 0.0f;       // the five columns below are filled by the reward part of the step (below)
    inf[SDC_INFO_RESERVED] = 0.0f;
    inf[SDC_INFO_EP_RETURN_LS] = 0.0f;
    inf[SDC_INFO_EP_RETURN_DC] = 0.0f;
    inf[SDC_INFO_EP_RETURN_BAT] = 0.0f;
    inf[SDC_INFO_EPISODE_STEP] = (float)(rel + 1);

    // ---- history append: the ring slot gets this step's key --------------------------------------------------
    if (append) S.hist[(size_t)env * SDC_HIST_STRIDE + slot] = x_new;
    S.qtab[(size_t)env * S.qstride + now] = make_uint2((unsigned)cum_now, cumT_now);

    // ---- new state record ------------------------------------------------------------------------------------
    unsigned* o = sh.rec;
    o[R_CURSOR] = (unsigned)ip;
    o[R_TREL] = (unsigned)(rel + 1);
    o[R_DAY] = (unsigned)day_n;
    o[R_HOURQ] = (unsigned)hourq_n;
    o[R_QPOPPED] = (unsigned)popped;
    o[R_QCUM] = (unsigned)cum_now;
    o[R_QCUMT] = cumT_now;
    o[R_QHEAD] = (unsigned)head;
    o[R_QCUM_HM1] = (unsigned)cum_hm1;
    o[R_QCUMT_HM1] = cumT_hm1;
    o[R_LAST_DELTA] = (unsigned)delta;
    o[R_CONSEC] = (unsigned)consecutive;
    o[R_SCALE] = (unsigned)scale;
    o[R_HIST_LEN] = (unsigned)hl;
    o[R_HIST_POS] = (unsigned)hpos;
    o[R_FAULT] = f_all;
    o[R_STPT] = (unsigned)__double2loint(stpt);
    o[R_STPT + 1] = (unsigned)__double2hiint(stpt);
    o[R_BAT] = (unsigned)__double2loint(bat_load);
    o[R_BAT + 1] = (unsigned)__double2hiint(bat_load);
    o[R_HIST_REF] = (unsigned)__double2loint(href);
    o[R_HIST_REF + 1] = (unsigned)__double2hiint(href);
  }

  // ---- rewards (utils/reward_creator.py:16-130).  Four rank windows and running sums (sdc_trackers.hpp) normally
  // answer without reading the history ring; a miss rebuilds them from the ring right here.
  {
    using namespace sdc_rw;
    if (__builtin_expect((S.debug_flags & 8) != 0, 0) && lane == 0) sh.dbg_t = wall_clock64();
    const int n = (int)sfl((unsigned)hl);
    const unsigned x_old = (unsigned)__builtin_amdgcn_readlane((int)x_old_l, 63);   // (lane 63 loaded it at the start)
    const bool has_old = append && x_old != KEY_NONE;
    unsigned o0 = hd0;
    double mean = 0.0, sd = 0.0;
    int path = 0;   // diagnostics: 0 no ring read, 1 a window re-centred ahead of need, 3 rebuilt
    const RingView R = {reinterpret_cast<const uint4*>(S.hist + (size_t)env * SDC_HIST_STRIDE), slot, x_new};
    if (__builtin_expect(n >= 2, 1)) {
      int k1, k3;
      quartile_ranks(n, k1, k3);
      // quartile windows q1 / q3; clip-bound windows bu (upper bound, keys as they are) / bl (lower bound, keys
      // complemented, so that on both sides "beyond the bound" means "at or above it")
      QTrack q1 = qt_load(hd0, H_Q1, qw.x), q3 = qt_load(hd0, H_Q3, qw.y);
      QTrack bu = qt_load(hd0, H_BU, qw.z), bl = qt_load(hd0, H_BL, qw.w);
      bool wd1 = false, wd3 = false, wdu = false, wdl = false;   // a window goes back to memory only if its lanes changed
      double A1 = rec_f64(hd0, H_A1), A2 = rec_f64(hd0, H_A2);
      bool ok = n >= SMALL_N && qt_valid(q1) && qt_valid(q3) && qt_valid(bu) && qt_valid(bl) && rec_i32(hd0, H_VALID) == 1;
      int why = ok ? 0 : 1;                            // diagnostics (debug_flags bit 1): why a rebuild was needed
      if (__builtin_expect(ok && append, 1)) {
        // O(1) updates: running sums, the four windows
        const double vn = key_f64(x_new), vo = has_old ? key_f64(x_old) : 0.0;
        const int n_prev = has_old ? n : n - 1;
        A1 += vn - vo;
        A2 += vn * vn - vo * vo;
        wd1 = qt_update(q1, x_new, x_old, has_old, n_prev, lane);
        wd3 = qt_update(q3, x_new, x_old, has_old, n_prev, lane);
        wdu = qt_update(bu, x_new, x_old, has_old, n_prev, lane);
        wdl = qt_update(bl, ~x_new, ~x_old, has_old, n_prev, lane);
        if (!(qt_valid(q1) && qt_valid(q3) && qt_valid(bu) && qt_valid(bl))) { ok = false; why = 2; }
      }
      unsigned kb0 = 0u, kb1 = 0u;
      // running (count, sum v, sum v^2) over the keys at or beyond each clip bound
      int qc0 = 0, qc1 = 0;
      double qs1_0 = 0.0, qs1_1 = 0.0, qs2_0 = 0.0, qs2_1 = 0.0;
      bool done_eval = false;
      if (__builtin_expect(ok, 1)) {
        unsigned a1, b1, a3, b3;
        if (__builtin_expect(qt_resolve(q1, k1, n, a1, b1) && qt_resolve(q3, k3, n, a3, b3), 1)) {
          const Bounds b = clip_bounds(n, a1, b1, a3, b3);
          kb0 = b.kub;               // upper tail: keys >= kub
          kb1 = ~(b.klb - 1u);       // lower tail, flipped: ~x >= ~(klb-1)  <=>  x < klb
          // first the value that came and the one that went against last step's bounds kbl, then the keys the bounds
          // have moved across since -- which a bound's window lists, as long as both the old and the new bound lie
          // inside its span
          const unsigned kbl0 = (unsigned)rec_i32(hd0, H_KB), kbl1 = (unsigned)rec_i32(hd0, H_KB + 1);
          qc0 = rec_i32(hd0, H_QC);
          qc1 = rec_i32(hd0, H_QC + 1);
          qs1_0 = rec_f64(hd0, H_QS1);
          qs1_1 = rec_f64(hd0, H_QS1 + 2);
          qs2_0 = rec_f64(hd0, H_QS2_HI);
          qs2_1 = rec_f64(hd0, H_QS2_LO);
          if (append) {
            const double vn = key_f64(x_new), vo = key_f64(x_old);
            if (has_old && x_old >= kbl0) { qc0 -= 1; qs1_0 -= vo; qs2_0 -= vo * vo; }
            if (has_old && ~x_old >= kbl1) { qc1 -= 1; qs1_1 -= vo; qs2_1 -= vo * vo; }
            if (x_new >= kbl0) { qc0 += 1; qs1_0 += vn; qs2_0 += vn * vn; }
            if (~x_new >= kbl1) { qc1 += 1; qs1_1 += vn; qs2_1 += vn * vn; }
          }
          const unsigned lo0 = min(kb0, kbl0), hi0 = max(kb0, kbl0), lo1 = min(kb1, kbl1), hi1 = max(kb1, kbl1);
          const bool cov0 = kb0 == kbl0 || qt_spans(bu, lo0, hi0, n), cov1 = kb1 == kbl1 || qt_spans(bl, lo1, hi1, n);
          if (__builtin_expect(cov0 && cov1, 1)) {
            int dc0 = 0, dc1 = 0;
            double d1_0 = 0.0, d2_0 = 0.0, d1_1 = 0.0, d2_1 = 0.0;
            const bool x0 = kb0 != kbl0 && win_crossing(bu, lo0, hi0, 0u, dc0, d1_0, d2_0);
            const bool x1 = kb1 != kbl1 && win_crossing(bl, lo1, hi1, KEY_NONE, dc1, d1_1, d2_1);
            if (__builtin_expect(x0, 0)) {
              const double sg = kb0 > kbl0 ? -1.0 : 1.0;   // bound moved out: the keys in between leave the tail
              qc0 += (kb0 > kbl0 ? -1 : 1) * (int)wave_sum_u32((unsigned)dc0);
              qs1_0 += sg * wave_sum_f64(d1_0);
              qs2_0 += sg * wave_sum_f64(d2_0);
            }
            if (__builtin_expect(x1, 0)) {
              const double sg = kb1 > kbl1 ? -1.0 : 1.0;
              qc1 += (kb1 > kbl1 ? -1 : 1) * (int)wave_sum_u32((unsigned)dc1);
              qs1_1 += sg * wave_sum_f64(d1_1);
              qs2_1 += sg * wave_sum_f64(d2_1);
            }
            // sum (v - bound), sum (v^2 - bound^2) over the keys beyond the bounds
            const double t1 = (qs1_0 - (double)qc0 * b.ub) + (qs1_1 - (double)qc1 * b.lb);
            const double t2 = (qs2_0 - (double)qc0 * (b.ub * b.ub)) + (qs2_1 - (double)qc1 * (b.lb * b.lb));
            clipped_moments(n, b, A1, A2, t1, t2, mean, sd, S.hist_cap, S.rc_hist_cap);
            done_eval = true;
          } else {
            why = cov0 ? 7 : 6;
          }
        } else {
          why = 4;
        }
      }
      bool valid = true;
      if (__builtin_expect(!done_eval, 0)) {
        // miss (no state yet, a window that did not cover, an inconsistency): rebuild everything from the ring
        const Rebuilt rb = rebuild_state(R, lane, n, sh.tl);
        if (n < SMALL_N || !rb.ok) {   // tiny history (nothing to keep), or a ring no window can describe
          mean = rb.mean;
          sd = rb.sd;
          q1.hi = q3.hi = bu.hi = bl.hi = 0;
          valid = false;
        } else {
          q1 = rb.q1;
          q3 = rb.q3;
          bu = rb.bu;
          bl = rb.bl;
          wd1 = wd3 = wdu = wdl = true;
          A1 = rb.A1;
          A2 = rb.A2;
          kb0 = rb.b.kub;
          kb1 = ~(rb.b.klb - 1u);
          qc0 = rb.qc[0]; qc1 = rb.qc[1];
          qs1_0 = rb.qs1[0]; qs1_1 = rb.qs1[1];
          qs2_0 = rb.qs2[0]; qs2_1 = rb.qs2[1];
          const double t1 = (qs1_0 - (double)qc0 * rb.b.ub) + (qs1_1 - (double)qc1 * rb.b.lb);
          const double t2 = (qs2_0 - (double)qc0 * (rb.b.ub * rb.b.ub)) + (qs2_1 - (double)qc1 * (rb.b.lb * rb.b.lb));
          clipped_moments(n, rb.b, A1, A2, t1, t2, mean, sd);
        }
        path = 3 + ((S.debug_flags & 2) ? why : 0);
      }
      put_u32(o0, H_KB, kb0);
      put_u32(o0, H_KB + 1, kb1);
      put_u32(o0, H_VALID, valid ? 1u : 0u);
      put_f64(o0, H_A1, A1);
      put_f64(o0, H_A2, A2);
      put_running_tails(o0, qc0, qc1, qs1_0, qs1_1, qs2_0, qs2_1);
      // Windows AHEAD of need: if, in the worst case for the keys the next step removes and adds, a window would no
      // longer cover what is asked of it, re-centre it now -- at the end of this wavefront's life, when the memory
      // system is quiet and the other wavefronts of its SIMD are finishing -- instead of at the start of the next
      // launch, where the sweep's loads would queue behind every env's start-of-step traffic.
      if (__builtin_expect(valid && n >= SMALL_N, 1)) {
        int k1n, k3n;
        quartile_ranks((append && n < S.hist_cap) ? n + 1 : n, k1n, k3n);
        // a bound's window is centred on the rank of the first key beyond the bound
        const int req = qt_refill_ahead(q1, k1n, n, 3, 6) | (qt_refill_ahead(q3, k3n, n, 3, 6) << 2) |
                        (qt_refill_ahead(bu, n - qc0, n, 10, 10) << 4) | (qt_refill_ahead(bl, n - qc1, n, 10, 10) << 6);
        if (__builtin_expect(req != 0, 0)) {
          __builtin_amdgcn_s_setprio(3);   // the step ends when the slowest wavefront does: let this one issue first
          // one copy of the refill code: the windows take turns through it
#pragma unroll 1
          for (int t = 0; t < 4; t++) {
            const int d = (req >> (2 * t)) & 3;
            if (d == REFILL_NONE) continue;
            QTrack A = t == 0 ? q1 : (t == 1 ? q3 : (t == 2 ? bu : bl));
            const int kt = t == 0 ? k1n : (t == 1 ? k3n : (t == 2 ? n - qc0 : n - qc1));
            qt_refill(A, d, kt, n, R, lane, sh.tl, t == 3 ? KEY_NONE : 0u);
            if (t == 0) { q1 = A; wd1 = true; }
            if (t == 1) { q3 = A; wd3 = true; }
            if (t == 2) { bu = A; wdu = true; }
            if (t == 3) { bl = A; wdl = true; }
          }
          path = max(path, 1);
        }
      }
      qt_put(o0, H_Q1, q1);
      qt_put(o0, H_Q3, q3);
      qt_put(o0, H_BU, bu);
      qt_put(o0, H_BL, bl);
      if (wd1 || wd3 || wdu || wdl)
        reinterpret_cast<uint4*>(S.qwin)[(size_t)env * SDC_WIN + lane] = make_uint4(q1.w, q3.w, bu.w, bl.w);
    } else {
      put_u32(o0, H_VALID, 0u);
    }
    put_u32(o0, H_N, (unsigned)n);
    put_f64(o0, H_EOFF, e_off);                                 // bat_total_energy_with_battery_KWh - hist_ref
    const double z = n < 2 ? 0.0 : (e_off - mean) / (sd > 0 ? sd : 1.0);
    const RewardIn rin = {z, norm_ci, oldest_norm, (double)overdue, energy, (double)hourq_n * 0.25, SDC_DIV_CONST(p_it, 1e3), total_kw, water};
    const Rewards rr = step_rewards(rin, S.reward_method, hd0);
    put_f64(o0, H_RET, rr.ret[0]);
    put_f64(o0, H_RET + 2, rr.ret[1]);
    put_f64(o0, H_RET + 4, rr.ret[2]);
    if (lane == 0) store_rewards(rr, z, path, env, rew, sh.info);
    __builtin_nontemporal_store(o0, &S.hdr[(size_t)env * SDC_HDR_DWORDS + lane]);
  }
}


// One env-step of env `env` by its wavefront (lane in [0, 64), LDS block sh): loads the env's state, runs the dynamics,
// the rewards and the reward-state upkeep, stores the new state and the outputs.
__device__ __forceinline__ void env_step(const SdcDev& S, DynShared& sh, const int env, const int lane, const int rel_hint,
                                         const int32_t* __restrict__ actions, float* __restrict__ obs,
                                         float* __restrict__ share_obs, unsigned char* __restrict__ done,
                                         float* __restrict__ info, float* __restrict__ final_obs,
                                         float* __restrict__ rew) {
  const int TL = S.table_len;
  // When the host knows the episode step every env is at (envs in lock-step: rel_hint >= 0), the step's feature row
  // -- which also holds its trace inputs -- and its queue-history probes are requested together with the state
  // record: ONE memory round trip before the dynamics start instead of two (record, then what it points to).
  const bool pre = rel_hint >= 0 && S.feat != nullptr;
  float frow_pre = 0.0f;
  double q_pre = 0.0;
  if (pre) {
    if (lane < SDC_FEAT_ROW) frow_pre = S.feat[((size_t)env * (S.episode_steps + 1) + (rel_hint + 1)) * SDC_FEAT_ROW + lane];
    if (lane >= G_Q97 && lane <= G_Q96) {
      const int back = lane == G_Q97 ? 97 : 24 * (lane - G_Q97);   // 97, 24, 48, 72, 96
      const int t = rel_hint - back;
      if (t >= 0) q_pre = *reinterpret_cast<const double*>(S.qtab + (size_t)env * S.qstride + t);
    }
  }
  const unsigned long long dbg_entry = (S.debug_flags & 16) ? wall_clock64() : 0ull;

  // ---- level 0: state record (coalesced) + actions ---------------------------------------------------------------
  unsigned* recp = S.rec + (size_t)env * SDC_REC_DWORDS;
  const unsigned r = recp[lane];
  unsigned hd0 = S.hdr[(size_t)env * SDC_HDR_DWORDS + lane];            // reward-side state: returns, trackers, sums

  int a_ls = actions[env * 3 + 0], a_dc = actions[env * 3 + 1], a_bat = actions[env * 3 + 2];
  // actions outside {0,1,2}: the reference's dict lookups raise (envs/dc_gym.py:160, bat_env_fwd_view.py:99); here
  // the step flags SDC_FAULT_ACTION and treats the action as "do nothing" / "no change" / "idle"
  const bool bad_action = (unsigned)a_ls > 2u || (unsigned)a_dc > 2u || (unsigned)a_bat > 2u;
  if (__builtin_expect(bad_action, 0)) {
    if ((unsigned)a_ls > 2u) a_ls = 1;
    if ((unsigned)a_dc > 2u) a_dc = 1;
    if ((unsigned)a_bat > 2u) a_bat = 2;
  }
  const int i = rec_i32(r, R_CURSOR), rel = rec_i32(r, R_TREL);
  const int loc = rec_i32(r, R_LOC);
  const SdcDcDev& PD = S.dc[rec_i32(r, R_CFG)];
  const double ci_min = rec_f64(r, R_CI_MIN), ci_den = rec_f64(r, R_CI_DEN);
  const double t_min = rec_f64(r, R_T_MIN), t_den = rec_f64(r, R_T_DEN);
  const int hourq = rec_i32(r, R_HOURQ);
  const int hourq_n = hourq + 1 >= 96 ? 0 : hourq + 1;
  unsigned fault = 0;
  if (i + 9 > TL - 1) fault |= SDC_FAULT_TABLE_RANGE;
  if (__builtin_expect(bad_action, 0)) fault |= SDC_FAULT_ACTION;
  // the ring slot this step's energy will overwrite: its current key is the evicted value the reward state's
  // order-statistic trackers need (issued with the gather below; 0xFFFFFFFF while the ring is still filling)
  const int hl0 = rec_i32(r, R_HIST_LEN);
  const int slot0 = hl0 < S.hist_cap ? hl0 : rec_i32(r, R_HIST_POS);
  unsigned x_old_l = 0xFFFFFFFFu;
  const bool append = S.reward_method[0] == SDC_REWARD_DEFAULT;   // else the history does not change this step
  if (lane == 63 && hl0 >= S.hist_cap && append) x_old_l = S.hist[(size_t)env * SDC_HIST_STRIDE + slot0];
  // A quartile window one step from the point where it is re-centred (sdc_ringpath.hpp qt_refill: a sweep over the
  // env's 40 KB ring at the END of the step) makes this wavefront the likely straggler of the launch: give it issue
  // priority from the start and pull its ring into L2 now (one dword per 128-byte line, 5 loads per lane, results
  // unused), so that the sweep finds it there.  ~1.5 % of the envs per step.
  if (hl0 >= sdc_rw::SMALL_N && append) {
    int k1, k3;
    sdc_rw::quartile_ranks(hl0 < S.hist_cap ? hl0 + 1 : hl0, k1, k3);
    const int r1 = rec_i32(hd0, H_Q1 + T_R0), h1 = rec_i32(hd0, H_Q1 + T_HI);
    const int r3 = rec_i32(hd0, H_Q3 + T_R0), h3 = rec_i32(hd0, H_Q3 + T_HI);
    const bool near1 = h1 > 0 && ((k1 - r1 >= h1 - 7 && r1 + h1 < hl0) || (k1 - r1 <= 4 && r1 > 0));
    const bool near3 = h3 > 0 && ((k3 - r3 >= h3 - 7 && r3 + h3 < hl0) || (k3 - r3 <= 4 && r3 > 0));
    // (likewise a clip-bound window whose bound sits within a dozen ranks of its edge)
    const int ru = rec_i32(hd0, H_BU + T_R0), hu = rec_i32(hd0, H_BU + T_HI), tu = hl0 - rec_i32(hd0, H_QC) - ru;
    const int rl = rec_i32(hd0, H_BL + T_R0), hl_ = rec_i32(hd0, H_BL + T_HI), tl = hl0 - rec_i32(hd0, H_QC + 1) - rl;
    const bool nearu = hu > 0 && ((tu >= hu - 12 && ru + hu < hl0) || (tu <= 12 && ru > 0));
    const bool nearl = hl_ > 0 && ((tl >= hl_ - 12 && rl + hl_ < hl0) || (tl <= 12 && rl > 0));
    if (__builtin_expect(near1 || near3 || nearu || nearl, 0)) {
      __builtin_amdgcn_s_setprio(2);
      const volatile unsigned* ring = S.hist + (size_t)env * SDC_HIST_STRIDE;
#pragma unroll
      for (int j = 0; j < SDC_HIST_STRIDE / 32 / SDC_WAVE; j++) (void)ring[(j * SDC_WAVE + lane) * 32];
    }
  }

  // ---- level 1: one 8-byte gather per lane -------------------------------------------------------------------------
  // (Staging these values one step ahead -- the previous step gathers them and the record load brings them in -- was
  // measured and is 1 % SLOWER: the start of a launch is bound by how much every env loads, not by round trips.)
  // The trace-only observation entries of this step come precomputed (sdc_features.hip), unless the episode has no
  // feature rows (a host write since the reset, an episode too long for that kernel): then the CI / temperature
  // windows are gathered and the features computed here.
  const bool feat_ok = S.feat != nullptr && rec_i32(r, R_FEAT_OK) == 1;
  const bool fast = pre && feat_ok && rel == rel_hint;   // what was requested up front is what this step needs
  float frow = frow_pre;
  if (feat_ok && !fast && lane < SDC_FEAT_ROW)
    frow = S.feat[((size_t)env * (S.episode_steps + 1) + (rel + 1)) * SDC_FEAT_ROW + lane];
  auto gather = [&](const int gi, const int grel, const int ghq) -> double {
    auto tix = [&](int idx) { return idx < 0 ? 0 : (idx > TL - 1 ? TL - 1 : idx); };
    const double* tW = S.tabW + (size_t)loc * TL;
    const double* tC = S.tabC + (size_t)loc * TL;
    const double* tw = S.t_win + (size_t)env * S.lw + grel;
    const double* wbw = S.wb_win + (size_t)env * S.lw + grel;
    const uint2* qt = S.qtab + (size_t)env * S.qstride;
    const double* src = nullptr;
    if (lane <= (feat_ok ? G_W0 : G_W2)) src = tW + tix(gi + lane);   // (W[i+1], W[i+2], the hour LUT: observations only)
    else if (lane == G_C0) src = tC + tix(gi);
    else if (lane == G_T0) src = tw;
    else if (lane == G_WB0) src = wbw;
    else if (lane == G_T1) src = tw + 1;
    else if (!feat_ok && lane == G_LUT) src = S.hour_lut + 2 * ghq;
    else if (!feat_ok && lane == G_LUT2) src = S.hour_lut + 2 * ghq + 1;
    else if (lane >= G_Q97 && lane <= G_Q96) {
      const int back = lane == G_Q97 ? 97 : 24 * (lane - G_Q97);   // 97, 24, 48, 72, 96
      const int t = grel - back;
      if (t >= 0) src = reinterpret_cast<const double*>(qt + t);
    } else if (!feat_ok && lane >= G_NC && lane < G_NC + 25) src = tC + tix(gi + 1 - 16 + (lane - G_NC));
    else if (!feat_ok && lane >= G_NT && lane < G_NT + 17) src = tw + 1 + (lane - G_NT);
    double v = 0.0;
    if (src) v = *src;
    if (!feat_ok) {
      // NC = (C - min) / (max - min) (utils/managers.py:437), NT likewise (:608): ONE division sequence for both windows
      const bool is_nc = lane >= G_NC && lane < G_NC + 25, is_nt = lane >= G_NT && lane < G_NT + 17;
      if (is_nc || is_nt) v = (v - (is_nc ? ci_min : t_min)) / (is_nc ? ci_den : t_den);
    }
    return v;
  };
  if (__builtin_expect(fast, 1)) {
    // the row's input slots and the probes go to the places the gather would have put them
    unsigned* g32 = reinterpret_cast<unsigned*>(sh.g);
    const unsigned fb = (unsigned)__float_as_int(frow);
    if (lane == SDC_FEAT_W || lane == SDC_FEAT_W + 1) g32[2 * G_W0 + (lane - SDC_FEAT_W)] = fb;
    if (lane == SDC_FEAT_C || lane == SDC_FEAT_C + 1) g32[2 * G_C0 + (lane - SDC_FEAT_C)] = fb;
    if (lane == SDC_FEAT_T || lane == SDC_FEAT_T + 1) g32[2 * G_T0 + (lane - SDC_FEAT_T)] = fb;
    if (lane == SDC_FEAT_WB || lane == SDC_FEAT_WB + 1) g32[2 * G_WB0 + (lane - SDC_FEAT_WB)] = fb;
    if (lane == SDC_FEAT_T1) sh.g[G_T1] = (double)frow;
    if (lane >= G_Q97 && lane <= G_Q96) sh.g[lane] = q_pre;
  } else {
    sh.g[lane] = gather(i, rel, hourq_n);
  }
  wave_sync();

  const unsigned long long dbg_a0 = (S.debug_flags & 8) ? wall_clock64() : 0ull;
  // the four rank windows, one key each per lane: wanted at the end of the step, so the load is issued here -- after
  // the start-of-launch burst of every env's record / header / gather loads -- and rides along in 4 registers
  const uint4 qw0 = reinterpret_cast<const uint4*>(S.qwin)[(size_t)env * SDC_WIN + lane];
  const unsigned long long dbg_a1 = (S.debug_flags & 8) ? wall_clock64() : 0ull;
  step_dynamics(S, PD, env, lane, r, a_ls, a_dc, a_bat, fault, x_old_l, hd0, qw0, feat_ok, frow, rew, sh);
  if (__builtin_expect((S.debug_flags & 8) != 0, 0)) {
    wave_sync();
    if (lane == 0) {
      const unsigned long long dbg_a3 = wall_clock64();
      sh.info[40] = (S.debug_flags & 16) ? (float)(dbg_a0 & 0xFFFFFull) : (float)(dbg_a1 - dbg_a0);
      sh.info[41] = (S.debug_flags & 16) ? (float)(dbg_a0 - dbg_entry) : (float)(sh.dbg_t - dbg_a1);
      sh.info[42] = (float)(dbg_a3 - sh.dbg_t);
      sh.info[43] = (S.debug_flags & 16) ? (float)(dbg_a3 & 0xFFFFFull) : (float)(dbg_a3 - dbg_a0);
    }
  }
  wave_sync();

  // (non-temporal: nothing in this launch reads them again, and whole lines that have already left the L2 shorten the
  // write-back at the end of the launch; partial-line stores -- rew, done, the ring slot -- must NOT be: they turn
  // into read-modify-writes at the memory and add 10 us)
  // ---- coalesced stores: record, obs [3][26] (78 floats), share_obs [29], info --------------------------------------
  __builtin_nontemporal_store(sh.rec[lane], &recp[lane]);
  const int terminal = (rel + 1 >= S.episode_steps) ? 1 : 0;
  {
    const float v0 = obs_padded_at(sh.pool, lane);
    __builtin_nontemporal_store(v0, &obs[(size_t)env * SDC_OBS_OUT + lane]);
    if (terminal && final_obs) final_obs[(size_t)env * SDC_OBS_OUT + lane] = v0;
    if (lane < SDC_OBS_OUT - SDC_WAVE) {
      const float v1 = obs_padded_at(sh.pool, SDC_WAVE + lane);
      __builtin_nontemporal_store(v1, &obs[(size_t)env * SDC_OBS_OUT + SDC_WAVE + lane]);
      if (terminal && final_obs) final_obs[(size_t)env * SDC_OBS_OUT + SDC_WAVE + lane] = v1;
    }
  }
  if (share_obs && lane < SDC_SHARE_OBS_DIM) __builtin_nontemporal_store(share_obs_at(sh.pool, lane), &share_obs[(size_t)env * SDC_SHARE_OBS_DIM + lane]);
  if (info && lane < SDC_INFO_DIM) __builtin_nontemporal_store(sh.info[lane], &info[(size_t)env * SDC_INFO_DIM + lane]);
  if (lane == 0) done[env] = (unsigned char)terminal;
}

}  // namespace

extern "C" __global__ __launch_bounds__(SDC_WAVE * SDC_WPB, 4) void sdc_dynamics_kernel_v1(SdcDev S, const int rel_hint, const int32_t* __restrict__ actions,
                                                                                float* __restrict__ obs,
                                                                                float* __restrict__ share_obs,
                                                                                unsigned char* __restrict__ done,
                                                                                float* __restrict__ info,
                                                                                float* __restrict__ final_obs,
                                                                                float* __restrict__ rew) {
  __shared__ DynShared shs[SDC_WPB];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / SDC_WAVE));   // wave-uniform, in an SGPR
  // Workgroup b runs on XCD b % 8 (the dispatcher deals workgroups round-robin to the 8 XCDs, each with its own L2):
  // give every XCD a CONTIGUOUS range of envs, so that output lines shared by neighbouring envs (rew, done, the
  // unaligned obs rows) are assembled in one L2 instead of being written back in pieces from several.
  const int nb = (int)gridDim.x;
  const int vb = (nb % 8 == 0) ? (int)(blockIdx.x % 8) * (nb / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int env = vb * SDC_WPB + wave;
  const int lane = threadIdx.x % SDC_WAVE;
  if (env >= S.n_envs) return;
  if (lane == 0) prof_stamp(S, SDC_PROF_DYNAMICS, env, 0);
  env_step(S, shs[wave], env, lane, rel_hint, actions, obs, share_obs, done, info, final_obs, rew);
  if (lane == 0) prof_stamp(S, SDC_PROF_DYNAMICS, env, 1);
}

// K env-steps per launch for action sequences that are known up front (scripted / rule-based policies, open-loop
// evaluation): every wavefront advances its own env K times -- envs do not interact, so there is nothing to wait for
// between steps; the dispatch ramp, the launch gap and the tail of a launch are paid once per K steps.  actions
// [K][N][3]; obs [K][N][3][26], share_obs [K][N][29] (or null), rew [K][N][3], done [K][N], info [K][N][44] (or null)
// hold every step's outputs.  The host keeps K within the episode (sdc_rollout).
extern "C" __global__ __launch_bounds__(SDC_WAVE * SDC_WPB, 4) void sdc_rollout_kernel_v1(SdcDev S, const int K, const int rel_hint,
                                                                               const int32_t* __restrict__ actions,
                                                                               float* __restrict__ obs,
                                                                               float* __restrict__ share_obs,
                                                                               unsigned char* __restrict__ done,
                                                                               float* __restrict__ info,
                                                                               float* __restrict__ final_obs,
                                                                               float* __restrict__ rew) {
  __shared__ DynShared shs[SDC_WPB];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / SDC_WAVE));
  const int nb = (int)gridDim.x;   // (each XCD a contiguous range of envs: see sdc_dynamics_kernel)
  const int vb = (nb % 8 == 0) ? (int)(blockIdx.x % 8) * (nb / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int env = vb * SDC_WPB + wave;
  const int lane = threadIdx.x % SDC_WAVE;
  const size_t N = (size_t)S.n_envs;
  if (env >= S.n_envs) return;
  if (lane == 0) prof_stamp(S, SDC_PROF_DYNAMICS, env, 0);
#pragma unroll 1
  for (int k = 0; k < K; k++) {
    __builtin_amdgcn_s_setprio(0);
    // (opaque copies: otherwise every per-env / per-lane address of the step is hoisted out of the loop and held in
    // registers across it -- 50 VGPRs the step itself needs)
    int env_k = env, lane_k = lane;
    asm volatile("" : "+s"(env_k), "+v"(lane_k));
    env_step(S, shs[wave], env_k, lane_k, rel_hint >= 0 ? rel_hint + k : -1, actions + (size_t)k * N * 3, obs + (size_t)k * N * SDC_OBS_OUT,
             share_obs ? share_obs + (size_t)k * N * SDC_SHARE_OBS_DIM : nullptr, done + (size_t)k * N,
             info ? info + (size_t)k * N * SDC_INFO_DIM : nullptr, k == K - 1 ? final_obs : nullptr, rew + (size_t)k * N * 3);
    // this wavefront's stores of step k are the loads of its step k + 1: complete them and drop stale lines of the
    // CU's vector L1 (workgroup scope: the L2 behind it is the same for both)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    wave_sync();
  }
  if (lane == 0) prof_stamp(S, SDC_PROF_DYNAMICS, env, 1);
}
