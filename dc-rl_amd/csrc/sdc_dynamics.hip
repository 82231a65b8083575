// sdc_dynamics.hip -- coupled per-timestep dynamics, ONE WAVEFRONT PER ENVIRONMENT (block = 64 lanes).
//
//   * lanes 0..24 / 32..48 stage the carbon-intensity and temperature observation windows into LDS
//     (coalesced reads of the struct-of-arrays trace tables);
//   * the load-shifting queue is O(1) prefix-count algebra plus a 64-ary wave search for the oldest task;
//   * the rack model runs lane = rack with per-rack constants read coalesced from the config table and
//     wave-shuffle (DPP) reductions for total IT power, CRAC return and outlet temperature;
//   * chiller / cooling tower / water / battery / set-point integrator are wave-uniform scalar fp64;
//   * the observation features (3 least-squares slopes, 2 x mean/std/peak/valley) run lane-parallel with
//     segmented butterfly reductions; lane 0 assembles the info block in LDS; all lanes store coalesced.
// The energy value and the three reward terms that need the history normaliser are handed to
// sdc_reward_kernel (sdc_reward.hip) through a 32-byte per-env record.
//
// Reference: sustaindc_env.py:533-737 and the sub-environment steps it drives (see per-block citations).
#include "sdc_device.hpp"

namespace {

struct DynShared {
  double nc[32];
  double nt[32];
  float obs[64];
  float info[SDC_INFO_DIM];
  int terminal;
};

// envs/datacenter.py:356-429 calculate_chiller_power
__device__ __forceinline__ double chiller_power(double max_cooling_cap, double load, double ambient_temp) {
  const double min_plr = 0.05, max_plr = 1.0, design_cond_temp = 35.0, design_evp_out_temp = 6.67;
  const double temp_rise_coef = 2.778, rated_cop = 3.0;
  const double delta_temp = (ambient_temp - design_cond_temp) / temp_rise_coef - (design_evp_out_temp - design_cond_temp);
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
  const double frac = oper < min_plr ? fmin(1.0, oper / min_plr) : 1.0;
  const double power = fflp * fpr * avail / rated_cop * frac;
  return oper > 0 ? power : 0.0;
}

__device__ __forceinline__ double sigmoid(double x) { return 1 / (1 + exp(-x)); }

// ------------------------------------------------------------------------------------------------
// the coupled dynamics at cursor i and the observation at i' = i + 1; one wavefront, lane in [0, 64)
__device__ void step_dynamics(const SdcDev& S, const int env, const int lane, const int32_t* __restrict__ actions,
                              DynShared& sh) {
  const int loc = S.loc_id[env];
  const sdc_dc_params& P = S.dc[S.cfg_id[env]];
  const int TL = S.table_len;
  const int i = S.cursor[env];
  const int rel = S.t_rel[env];
  const int a_ls = actions[env * 3 + 0], a_dc = actions[env * 3 + 1], a_bat = actions[env * 3 + 2];
  unsigned fault = 0;
  if (i + 9 > TL - 1) fault |= SDC_FAULT_TABLE_RANGE;
  auto tix = [&](int idx) { return idx < 0 ? 0 : (idx > TL - 1 ? TL - 1 : idx); };
  const double* tW = S.tabW + (size_t)loc * TL;
  const double* tC = S.tabC + (size_t)loc * TL;
  const double wl = tW[tix(i)];
  const double w_ip = tW[tix(i + 1)], w_ip1 = tW[tix(i + 2)];
  const double ci_i = tC[tix(i)];
  const double* tw = S.t_win + (size_t)env * S.lw;
  const double* wbw = S.wb_win + (size_t)env * S.lw;
  const double amb = tw[rel], wet_bulb = wbw[rel], amb_next = tw[rel + 1];
  const int day = S.day[env];
  const int hourq = S.hourq[env];
  const double hour = (double)hourq * 0.25;

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
  int popped = S.q_popped[env];
  const int cum_prev = S.q_cum[env];
  const unsigned cumT_prev = S.q_cumT[env];
  auto cum_at = [&](int t) -> int { return t < 0 ? 0 : (int)qt[t].x; };  // t <= now - 1
  // overdue: age > 24 h  <=>  enqueued at step <= now - 97  (carbon_ls.py:208)
  const int overdue = max(0, cum_at(now - 97) - popped);
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
    util = (double)(od_proc + (shf - add)) / 100;
  } else if (a_ls == 2) {
    if (avail >= 1) {
      processed = min(min(shf, avail), cum_prev - popped);
      popped += processed;
      util = (double)(shf + processed + od_proc) / 100;
    } else {
      util = (double)(shf + od_proc) / 100;
    }
  } else {
    util = (double)(shf + od_proc) / 100;
  }
  util += (double)ns / 100;
  const int cum_now = cum_prev + add;
  const unsigned cumT_now = cumT_prev + (unsigned)add * (unsigned)now;
  const int total = cum_now - popped;
  // age histogram, bins [0,6,12,18,24,inf] hours = [0,24,48,72,96,inf) steps (carbon_ls.py:63-73)
  auto older_eq = [&](int a) -> int { return max(0, cum_at(now - a) - popped); };  // tasks with age >= a steps (a > 0)
  const int a24 = older_eq(24), a48 = older_eq(48), a72 = older_eq(72), a96 = older_eq(96);
  double hist[5];
  {
    const double den = (double)max(total, 1);
    hist[0] = (double)(total - a24) / den;
    hist[1] = (double)(a24 - a48) / den;
    hist[2] = (double)(a48 - a72) / den;
    hist[3] = (double)(a72 - a96) / den;
    hist[4] = a96 > 0 ? 1.0 : 0.0;
  }
  // oldest task: smallest step h in [head, now] with cum[h] > popped (64-ary search, <= 2 rounds)
  int head = S.q_head[env];
  double oldest = 0.0, avg = 0.0;
  if (total > 0) {
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
    // sum of enqueue steps over the queued tasks = cumT[now] - cumT[h-1] - (popped - cum[h-1]) * h
    int cum_hm1 = 0;
    unsigned cumT_hm1 = 0;
    if (head > 0) {
      if (head == now) {
        cum_hm1 = cum_prev;
        cumT_hm1 = cumT_prev;
      } else {
        const uint2 e = qt[head - 1];
        cum_hm1 = (int)e.x;
        cumT_hm1 = e.y;
      }
    }
    const long long sum_t = (long long)cumT_now - (long long)cumT_hm1 - (long long)(popped - cum_hm1) * head;
    const long long sum_age_steps = (long long)total * now - sum_t;
    oldest = (double)(now - head) * 0.25;             // hours, exact
    avg = ((double)sum_age_steps * 0.25) / (double)total;  // sum(ages) is exact in the reference too
  } else {
    head = now;
  }
  const double normq = (double)total / (double)S.queue_max;
  const double oldest_norm = oldest / 24, avg_norm = avg / 24;

  // ---- CRAC set-point integrator: envs/dc_gym.py:160-174 ----------------------------------------
  if (util < 0.0 || util > 1.0) fault |= SDC_FAULT_CPU_LOAD;
  const int delta = a_dc - 1;  // make_envs_pyenv.py:127-131
  int last_delta = S.last_delta[env], consecutive = S.consecutive[env], scale = S.scale[env];
  if (last_delta != -2 && delta == last_delta && a_dc != 0) {
    consecutive += 1;
  } else {
    consecutive = 1;
    scale = 1;
  }
  if (consecutive > 3) scale += 1;
  double stpt = S.stpt[env] + (double)(delta * scale);
  stpt = fmax(fmin(stpt, P.max_temp), P.min_temp);

  // ---- rack model, lane = rack: envs/datacenter.py:250-317, :157-181 ------------------------------
  const int R = P.n_racks;
  const double load_pct = util * 100;
  double pcpu = 0.0, pfan = 0.0, outlet = 0.0, ret_plus_out = 0.0;
  unsigned bad_delta = 0;
  if (lane < R) {
    const double sa = fmax(3.8, fmin(P.rack_supply[lane], 5.3));  // datacenter.py:209-215
    const double inlet = sa + stpt;
    const double ratio = ((P.m_cpu + 0.05) * inlet + P.c_cpu) + P.rs_cpu * (load_pct / 100);
    const double cpu1 = fmax(P.rack_idle[lane], P.rack_full[lane] * ratio);
    const double v = (P.m_fan * 10 * inlet + P.c_fan * 5) + P.rs_fan * (load_pct / 20);
    const double fan1 = P.itfan_ref_p * (v / P.itfan_ref_v_ratio);
    const double vf1 = P.it_fan_full_load_v * v;
    const double n = P.rack_n[lane];
    pcpu = n * cpu1;
    pfan = n * fan1;
    const double vtot = n * vf1;
    const double power_term = pow(pcpu + pfan, 1.096);
    const double airflow_term = P.c_air * P.rho_air * pow(vtot, 0.824) * 0.526;
    outlet = inlet + 1.918 * power_term / airflow_term + -14.01;
    if (outlet - inlet < 2) bad_delta = 1;
    ret_plus_out = P.rack_return[lane] + outlet;
  }
  if (__ballot(bad_delta) != 0ull) fault |= SDC_FAULT_OUTLET_DELTA;
  const double sum_cpu = wave_sum_f64(pcpu), sum_fan = wave_sum_f64(pfan);
  const double avg_ret = wave_sum_f64(ret_plus_out) / (double)R;  // datacenter.py:531-541
  const double mean_outlet = wave_sum_f64(outlet) / (double)R;
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
    const double v_air = m_air / P.rho_air;
    const double x = fmin(v_air / P.ctafr, 1);
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
  const double total_kw = (p_it + ct + comp) / 1e3;

  // ---- battery: envs/bat_env_fwd_view.py:84-245, envs/battery_model.py:94-132 ----------------------
  const double cap = P.bat_capacity_mwh;
  const double dcload = total_kw / 1e3;  // MW (sustaindc_env.py:652)
  double bat_load = S.bat_load[env];
  double energy, co2;
  if (a_bat == 0) {  // charge
    const double soc = (bat_load - 0) / (cap - 0);
    const double rate = np_round(0.5 * (1 - sigmoid(10 * (soc - 0.5))), 1e4);
    const double tu = rate * 15 / 60;
    const double max_charge = fmin((cap / 1) * 0.1, (1 * cap - bat_load) / ((1 * tu) - (-0.04)));
    const double charging_load = fmin(max_charge, cap) * 1 * tu;
    bat_load = np_round(bat_load + charging_load, 1e8);
    energy = dcload * 1e3 * 0.25 + charging_load * 1e3;
    co2 = energy * ci_i;
  } else if (a_bat == 1) {  // discharge
    const double soc = (bat_load - 0) / (cap - 0);
    const double rate = fmax(0.5, 4 * sigmoid(10 * (soc - 0.25)));
    const double tu = rate * 15 / 60;
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
  const double soc_after = bat_load / cap;

  // ---- time: utils/managers.py:127-147 -------------------------------------------------------------
  int hourq_n = hourq + 1, day_n = day;
  if (hourq_n >= 96) {
    hourq_n = 0;
    day_n += 1;
  }
  const int terminal = (rel + 1 >= S.episode_steps) ? 1 : 0;
  const int ip = i + 1;

  // ---- observations at i' (sustaindc_env.py:565-585): all lanes cooperate -------------------------------------
  {
    ObsScalars o;
    o.cos_h = S.hour_lut[2 * hourq_n];
    o.sin_h = S.hour_lut[2 * hourq_n + 1];
    o.w_cur = w_ip;
    o.w_next = w_ip1;
    o.soc = soc_after;
    o.normq = normq;
    o.oldest = oldest_norm;
    o.avg = avg_norm;
    for (int b = 0; b < 5; b++) o.hist[b] = hist[b];
    o.have_past = ip >= 16;
    build_obs_pool(sh.nc, sh.nt, o, sh.obs, lane);
  }

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
    inf[SDC_INFO_DC_ITE_TOTAL_POWER_KW] = (float)(p_it / 1e3);
    inf[SDC_INFO_DC_CT_TOTAL_POWER_KW] = (float)(ct / 1e3);
    inf[SDC_INFO_DC_COMPRESSOR_TOTAL_POWER_KW] = (float)(comp / 1e3);
    inf[SDC_INFO_DC_HVAC_TOTAL_POWER_KW] = (float)((ct + comp) / 1e3);
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
    inf[SDC_INFO_NORM_CI] = (float)sh.nc[17];
    inf[SDC_INFO_OUTSIDE_TEMP] = (float)amb_next;
    inf[SDC_INFO_DAY] = (float)day_n;
    inf[SDC_INFO_HOUR] = (float)((double)hourq_n * 0.25);
    const unsigned f_all = S.fault[env] | fault;
    inf[SDC_INFO_FAULT] = (float)f_all;
    inf[SDC_INFO_ENERGY_Z] = 0.0f;
    inf[SDC_INFO_RESERVED] = 0.0f;
    inf[SDC_INFO_EP_RETURN_LS] = 0.0f;   // the four columns below are filled by sdc_reward_kernel
    inf[SDC_INFO_EP_RETURN_DC] = 0.0f;
    inf[SDC_INFO_EP_RETURN_BAT] = 0.0f;
    inf[SDC_INFO_EPISODE_STEP] = (float)(rel + 1);

    // ---- hand-off to the reward kernel (4 doubles per env, struct of arrays) ----------------------------
    {
      const int N = S.n_envs;
      // history append (utils/reward_creator.py:7-14).  The ring holds fp32 OFFSETS from the env's first energy
      // value (kept in fp64): normalize_energy is shift-invariant, and offsets keep the fp32 rounding error
      // proportional to the spread of the history instead of to the ~300 kWh magnitude (two nearly equal
      // energies would otherwise lose the z-score).  Stored as order-preserving keys for the reward kernel.
      const int hl = S.hist_len[env];
      const double href = hl == 0 ? energy : S.hist_ref[env];
      const double e_off = energy - href;
      int slot;
      if (hl < S.hist_cap) {
        slot = hl;
        S.hist_len[env] = hl + 1;
        if (hl == 0) S.hist_ref[env] = href;
      } else {
        slot = S.hist_pos[env];
        S.hist_pos[env] = slot + 1 == S.hist_cap ? 0 : slot + 1;
      }
      S.hist[(size_t)env * SDC_HIST_STRIDE + slot] = sdc_f32_key(__float_as_uint((float)e_off));
      S.hand[env] = e_off;                     // bat_total_energy_with_battery_KWh - hist_ref
      S.hand[N + env] = sh.nc[17];             // norm_CI = NC[i'+1]  (sustaindc_env.py:681)
      S.hand[2 * N + env] = oldest_norm;       // ls_oldest_task_age
      S.hand[3 * N + env] = (double)overdue;   // ls_overdue_penalty
    }
    sh.terminal = terminal;

    // ---- state write-back ------------------------------------------------------------------------------
    S.cursor[env] = ip;
    S.t_rel[env] = rel + 1;
    S.day[env] = day_n;
    S.hourq[env] = hourq_n;
    S.q_popped[env] = popped;
    S.q_cum[env] = cum_now;
    S.q_cumT[env] = cumT_now;
    S.q_head[env] = head;
    S.qtab[(size_t)env * S.qstride + now] = make_uint2((unsigned)cum_now, cumT_now);
    S.last_delta[env] = delta;
    S.consecutive[env] = consecutive;
    S.scale[env] = scale;
    S.stpt[env] = stpt;
    S.bat_load[env] = bat_load;
    S.fault[env] = f_all;
    double* cr = S.carry;
    const int N = S.n_envs;
    cr[SDC_C_NORMQ * N + env] = normq;
    cr[SDC_C_OLDEST * N + env] = oldest_norm;
    cr[SDC_C_AVG * N + env] = avg_norm;
    for (int b = 0; b < 5; b++) cr[(SDC_C_H0 + b) * N + env] = hist[b];
  }
}

}  // namespace

extern "C" __global__ __launch_bounds__(SDC_WAVE, 4) void sdc_dynamics_kernel(SdcDev S, const int32_t* __restrict__ actions,
                                                                             float* __restrict__ obs,
                                                                             float* __restrict__ share_obs,
                                                                             unsigned char* __restrict__ done,
                                                                             float* __restrict__ info,
                                                                             float* __restrict__ final_obs) {
  __shared__ DynShared sh;
  const int env = blockIdx.x;
  const int lane = threadIdx.x;
  stage_windows(S, S.loc_id[env], S.cursor[env] + 1, S.t_win + (size_t)env * S.lw + S.t_rel[env] + 1, S.ci_min[env],
                S.ci_den[env], S.t_min[env], S.t_den[env], lane, sh.nc, sh.nt);
  __syncthreads();
  step_dynamics(S, env, lane, actions, sh);
  __syncthreads();
  const int terminal = sh.terminal;
  // coalesced stores: obs [3][26] (78 floats), share_obs [29], info [SDC_INFO_DIM]
  {
    const float v0 = obs_padded_at(sh.obs, lane);
    obs[(size_t)env * SDC_OBS_OUT + lane] = v0;
    if (terminal && final_obs) final_obs[(size_t)env * SDC_OBS_OUT + lane] = v0;
    if (lane < SDC_OBS_OUT - SDC_WAVE) {
      const float v1 = obs_padded_at(sh.obs, SDC_WAVE + lane);
      obs[(size_t)env * SDC_OBS_OUT + SDC_WAVE + lane] = v1;
      if (terminal && final_obs) final_obs[(size_t)env * SDC_OBS_OUT + SDC_WAVE + lane] = v1;
    }
  }
  if (share_obs && lane < SDC_SHARE_OBS_DIM) share_obs[(size_t)env * SDC_SHARE_OBS_DIM + lane] = share_obs_at(sh.obs, lane);
  if (info && lane < SDC_INFO_DIM) info[(size_t)env * SDC_INFO_DIM + lane] = sh.info[lane];
  if (lane == 0) done[env] = (unsigned char)terminal;
}
