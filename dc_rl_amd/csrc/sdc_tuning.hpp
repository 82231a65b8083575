// sdc_tuning.hpp -- every compile-time switch of the step kernels in ONE place.
//
// Fixed choices (measured alternatives and why they lost: docs/HISTORY.md) are plain constants here.  What is left as a
// preprocessor switch is measurement instrumentation only: a default build contains none of it.
//   -DSDC_STAMP_A=i -DSDC_STAMP_B=j   two wall-clock stamps at the marks SDC_AT(i) / SDC_AT(j) of the step (tools/phase_scan.sh)
//   -DSDC_FAST_DEBUG=1                the common-case kernels keep the in-kernel clock stamps of debug_flags 8 / 16 / 32
//                                     (tools/wave_timeline.py, tools/wave_phases.py see THEM and not the general kernels)
//   -DSDC_ACTOR_CLOCK / -DSDC_ACTOR_SKIP   closed-loop kernel: shader clocks per phase / the kernel without the networks
//   -DSDC_RT                          sdc_reset.hip: phase stamps of the reset kernel (tools/reset_phases.py)
#pragma once

#ifndef SDC_STAMP_A
#define SDC_STAMP_A 0
#endif
#ifndef SDC_STAMP_B
#define SDC_STAMP_B 0
#endif
#define SDC_AT(k, SH, LANE0)                                                        \
  do {                                                                              \
    if ((k) == SDC_STAMP_A && SDC_STAMP_A != 0 && (LANE0)) (SH).dbg_s[0] = wall_clock64(); \
    if ((k) == SDC_STAMP_B && SDC_STAMP_B != 0 && (LANE0)) (SH).dbg_s[1] = wall_clock64(); \
  } while (0)
#ifndef SDC_FAST_DEBUG
#define SDC_FAST_DEBUG 0
#endif
#define SDC_DBG_OK(FAST_) (!(FAST_) || SDC_FAST_DEBUG)
// which rare paths a wavefront's step took (reported in info[reserved] above bit 3 when debug_flags has bit 3):
// 1 oldest-task table search, 2 a key inside a rank window, 4 a clip bound crossed keys, 8 a deferred window arrived,
// 16 a re-centring request filed
#define SDC_DBG_BIT(FAST_, SH, B)                                                                   \
  do {                                                                                              \
    if (SDC_DBG_OK(FAST_) && (S.debug_flags & 8) && (threadIdx.x & 63) == 0) (SH).dbg_bits |= (B); \
  } while (0)

// outputs (obs, share_obs, info): non-temporal stores -- nothing in the launch reads them again, and whole lines that have
// already left the L2 shorten the write-back at the end of the launch
#define SDC_OUT_STORE(v, p) __builtin_nontemporal_store((v), (p))

#define SDC_STEP_WPB 4          // wavefronts per workgroup of the step kernels; they share nothing (no s_barrier)
#define SDC_CUS 256
#define SDC_BASE_PRIO 0         // issue priority of the env wavefronts (the late dispatch round: + 1 during the dynamics) ...
#define SDC_LATE_PRIO 1
#define SDC_SWEEP_PRIO 3        // ... and of the sweep wavefronts: the launch cannot end before its sweeps have
#define SDC_STEP_WAVES_PER_EU 3 // resident wavefronts per SIMD the single-step kernels are compiled for (<= 168 VGPRs)
#define SDC_QUAD_WAVES_PER_EU 3
#define SDC_STEP_BOUNDS __launch_bounds__(SDC_WAVE * SDC_STEP_WPB) __attribute__((amdgpu_waves_per_eu(SDC_STEP_WAVES_PER_EU, SDC_STEP_WAVES_PER_EU)))
#define SDC_QUAD_BOUNDS \
  __launch_bounds__(SDC_WAVE * SDC_STEP_WPB) __attribute__((amdgpu_waves_per_eu(SDC_QUAD_WAVES_PER_EU, SDC_QUAD_WAVES_PER_EU)))
