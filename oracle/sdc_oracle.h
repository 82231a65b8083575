/*
 * sdc_oracle.h -- CPU restatement (plain C99, fp64, scalar, one env at a time) of the
 * reference's coupled SustainDC step.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (dc_rl_amd/, include/) may include,
 * link or call this.  Allowed users: tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg -- and there only as the checker / the timed CPU baseline.
 *
 * Pinned against golden vectors captured from the imported Python reference
 * (the .npz files under tests/golden/, generator tests/golden/gen_golden.py); see
 * tests/test_oracle_golden.py.  Every function cites the reference file:line it restates
 * (paths relative to /root/reference).
 */
#ifndef SDC_ORACLE_H
#define SDC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDCO_MAX_RACKS 64
#define SDCO_HIST_CAP 10000  /* utils/reward_creator.py:5  deque(maxlen=10000) */
#define SDCO_QUEUE_CAP 1000  /* sustaindc_env.py:149 queue_max_len=1000 */
#define SDCO_OBS_DIM 53      /* 26 (ls) + 14 (dc) + 13 (bat) */
#define SDCO_INFO_DIM 40

/* info columns (same order as include/sustaindc_hip.h SDC_INFO_*) */
enum {
  SDCO_I_LS_ORIGINAL_WORKLOAD = 0, SDCO_I_LS_SHIFTED_WORKLOAD, SDCO_I_LS_TASKS_IN_QUEUE,
  SDCO_I_LS_NORM_TASKS_IN_QUEUE, SDCO_I_LS_TASKS_DROPPED, SDCO_I_LS_TASKS_PROCESSED,
  SDCO_I_LS_OLDEST_TASK_AGE, SDCO_I_LS_AVERAGE_TASK_AGE, SDCO_I_LS_OVERDUE_PENALTY,
  SDCO_I_LS_COMPUTED_TASKS, SDCO_I_LS_CURRENT_HOUR,
  SDCO_I_LS_HIST0, SDCO_I_LS_HIST1, SDCO_I_LS_HIST2, SDCO_I_LS_HIST3, SDCO_I_LS_HIST4,
  SDCO_I_DC_ITE_KW, SDCO_I_DC_CT_KW, SDCO_I_DC_COMPRESSOR_KW, SDCO_I_DC_HVAC_KW, SDCO_I_DC_TOTAL_KW,
  SDCO_I_DC_SETPOINT_DELTA, SDCO_I_DC_SETPOINT, SDCO_I_DC_CPU_FRACTION, SDCO_I_DC_INT_TEMPERATURE,
  SDCO_I_DC_AMBIENT_TEMP, SDCO_I_DC_WATER_USAGE,
  SDCO_I_BAT_ACTION, SDCO_I_BAT_SOC, SDCO_I_BAT_CO2, SDCO_I_BAT_AVG_CI,
  SDCO_I_BAT_ENERGY_WITHOUT_KWH, SDCO_I_BAT_ENERGY_WITH_KWH,
  SDCO_I_NORM_CI, SDCO_I_OUTSIDE_TEMP, SDCO_I_DAY, SDCO_I_HOUR,
  SDCO_I_FAULT, SDCO_I_RESERVED0, SDCO_I_RESERVED1
};

/* fault bits (the reference raises / asserts; see sdco_step) */
#define SDCO_FAULT_OUTLET_DELTA 1u  /* envs/datacenter.py:295-300 */
#define SDCO_FAULT_CPU_LOAD 2u      /* envs/dc_gym.py:290 */
#define SDCO_FAULT_BAT_DISCHARGE 4u /* envs/bat_env_fwd_view.py:237 */
#define SDCO_FAULT_WORKLOAD 8u      /* envs/carbon_ls.py:333-336 */

typedef struct {
  int R;
  double rack_n[SDCO_MAX_RACKS];      /* CPUs per rack after the MAX_W_PER_RACK cap */
  double rack_full[SDCO_MAX_RACKS];   /* full-load W per CPU */
  double rack_idle[SDCO_MAX_RACKS];   /* idle W per CPU */
  double rack_supply[SDCO_MAX_RACKS]; /* supply approach temp (unclamped) */
  double rack_return[SDCO_MAX_RACKS]; /* return approach temp */
  double m_cpu, c_cpu, rs_cpu;        /* envs/datacenter.py:31-39 */
  double m_fan, c_fan, rs_fan;        /* envs/datacenter.py:41-49 */
  double itfan_ref_p, itfan_ref_v_ratio, it_fan_full_load_v;
  double c_air, rho_air, crac_supply_pu;
  double ct_fan_ref_p, ctafr;         /* SIZED values (utils/make_envs_pyenv.py:159-161) */
  double min_temp, max_temp;          /* 15.0 / 21.6 (utils/make_envs_pyenv.py:125-126) */
  double bat_capacity;                /* MWh (sustaindc_env.py:152) */
  int queue_max_len;
  /* reward function per agent slot (ls, dc, bat), utils/reward_creator.py:322-334 REWARD_METHOD_MAP:
   * 0 the slot's default (default_ls_reward / default_dc_reward / default_bat_reward), 1 footprint only
   * (default_dc_reward = default_bat_reward), 2 custom_agent_reward, 3 tou_reward, 4 energy_efficiency_reward,
   * 5 energy_PUE_reward, 6 water_usage_efficiency_reward.  Only default_ls_reward appends to the energy history
   * (reward_creator.py:63), i.e. the history grows iff reward_method[0] == 0. */
  int reward_method[3];
} sdco_params;

typedef struct {
  /* trace windows, element k <-> absolute table index win_lo + k */
  const double *W, *C, *NC, *T, *WB, *NT;
  int win_lo, win_len;
  /* time (utils/managers.py:91-147) */
  int cursor;
  int day;
  double hour;
  int t_end; /* terminal when cursor >= t_end */
  /* load-shifting queue: FIFO of enqueue timestamps (envs/carbon_ls.py:59) */
  int q_day[SDCO_QUEUE_CAP];
  double q_hour[SDCO_QUEUE_CAP];
  int q_head, q_len;
  /* ls info carried into the next obs (sustaindc_env.py:569-573) */
  double ls_norm_tasks_in_queue, ls_oldest_age, ls_avg_age, ls_hist[5];
  /* CRAC set-point integrator (envs/dc_gym.py:160-174); stpt survives reset */
  double stpt;
  int has_last_delta;
  int last_delta;
  int consecutive;
  int scale;
  /* battery (envs/battery_model.py) */
  double bat_load;
  /* energy history (utils/reward_creator.py:5), survives reset */
  double hist[SDCO_HIST_CAP];
  int hist_len, hist_pos;
  double scratch[2 * SDCO_HIST_CAP]; /* work area of normalize_energy */
} sdco_env;

/* zero-initialise; stpt = 18 (utils/make_envs_pyenv.py:124) */
void sdco_env_init(sdco_env *e);

/* SustainDC.reset() given the post-reset trace windows (sustaindc_env.py:436-531). */
void sdco_episode_begin(sdco_env *e, const sdco_params *p, const double *W, const double *C, const double *NC,
                        const double *T, const double *WB, const double *NT, int win_lo, int win_len,
                        int init_day, int init_hour, int episode_steps, float *obs53);

/* SustainDC.step() (sustaindc_env.py:533-621).  Returns 1 when the episode is over (truncated). */
int sdco_step(sdco_env *e, const sdco_params *p, const int32_t actions[3], float *obs53, double rew[3],
              double info[SDCO_INFO_DIM]);

/* pieces exposed for unit tests */
double sdco_normalize_energy(const double *hist, int n, double value); /* reward_creator.py:16-45 */
double sdco_percentile_linear(double *scratch, int n, double q);        /* np.percentile(..., q) on a scratch copy */
void sdco_hour_sincos(double hour, double *cos_h, double *sin_h);      /* managers.py:66-88 */
double sdco_polyfit_slope(const double *y, int n);                     /* np.polyfit(range(n), y, 1)[0] */
void sdco_extract_features(const double *vals, int n, double cur, double out5[5]); /* sustaindc_env.py:266-300 */
void sdco_dc_model(const sdco_params *p, double stpt, double load_pct, double ambient, double wet_bulb,
                   double out[8], unsigned *fault); /* IT + HVAC + water; see .c */
double sdco_chiller_power(double max_cooling_cap, double load, double ambient_temp); /* datacenter.py:356-429 */

/* init-time sizing (utils/make_envs_pyenv.py:139-197, envs/datacenter.py:476-529).  Fills
 * p->ctafr, p->ct_fan_ref_p, p->bat_capacity and the min/max of the 88-point sweep. */
void sdco_size_datacenter(sdco_params *p, double max_ambient_temp, double out_ranges[8]);

/* run `nsteps` steps of env with a pre-generated action stream (CPU baseline helper);
 * restarts episodes in place (same windows) when they end.  Returns steps done. */
long sdco_run_steps(sdco_env *e, const sdco_params *p, const int32_t *actions, long nsteps, int episode_steps,
                    int init_day, int init_hour, double *rew_sum3);

#ifdef __cplusplus
}
#endif
#endif
