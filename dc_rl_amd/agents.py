"""Rule-based baseline agents of the reference (utils/base_agents.py, utils/rbc_agents.py) with the same class and
method names, plus batched device-side versions (`act_batch`) that produce actions for N environments from torch
tensors without leaving the GPU.

Action encodings (sustaindc_env.py, envs/*):  load shifting {0 defer, 1 do nothing, 2 process the queue};
HVAC set-point {0 down, 1 hold, 2 up};  battery {0 charge, 1 discharge, 2 idle}.
"""
from __future__ import annotations

import numpy as np


class BaseLoadShiftingAgent:
    """utils/base_agents.py:3-27 -- always 'do nothing' (1)."""

    def __init__(self, parameters=None):
        self.parameters = parameters
        self.do_nothing_action_value = 1

    def do_nothing_action(self):
        return self.do_nothing_action_value

    def act(self, *args, **kwargs):
        return self.do_nothing_action()

    def act_batch(self, n_envs, device=None):
        import torch
        return torch.full((n_envs,), self.do_nothing_action_value, dtype=torch.int32, device=device)


class BaseHVACAgent(BaseLoadShiftingAgent):
    """utils/base_agents.py:29-58 -- always 'hold the set-point' (1).  The reference class has no `act()` although
    sustaindc_env.py:640 calls it when agent_dc is not trained (AttributeError there); here `act()` returns the
    do-nothing action, which is what that call site means."""

    def __init__(self, parameters=None):
        self.parameters = parameters
        self.do_nothing_action_value = np.int64(1)


class BaseBatteryAgent(BaseLoadShiftingAgent):
    """utils/base_agents.py:60-98 -- always 'idle' (2)."""

    def __init__(self, parameters=None):
        self.parameters = parameters
        self.do_nothing_action_value = 2


class RBCBatteryAgent:
    """utils/rbc_agents.py:3-48 -- charge (0) when the smoothed carbon-intensity forecast `look_ahead` steps ahead is
    above the current value, else discharge (1)."""

    def __init__(self, look_ahead=3, smooth_window=1, max_soc=0.9, min_soc=0.2):
        self.look_ahead = look_ahead
        self.smooth_window = smooth_window
        self.max_soc = max_soc
        self.min_soc = min_soc

    def act(self, carbon_intensity_values, current_soc):
        window = self.smooth_window
        smoothed = np.convolve(carbon_intensity_values, np.ones(window), "valid") / window
        return 0 if smoothed[self.look_ahead] > carbon_intensity_values[0] else 1

    def act_batch(self, carbon_intensity_values, current_soc=None):
        """carbon_intensity_values: torch tensor [N, K] (current value first, then the forecast) -> int32 [N]."""
        import torch
        w = self.smooth_window
        x = carbon_intensity_values
        sm = x.unfold(1, w, 1).sum(-1) / w if w > 1 else x
        return torch.where(sm[:, self.look_ahead] > x[:, 0], 0, 1).to(torch.int32)


class trim_and_respond_ctrl:
    """utils/trim_and_respond.py:8-38 -- "trim and respond" supply-air reset for agent_dc: while the monitored
    temperature stays at or below the limit, hold the set-point (1) and every fifth consecutive call trim it up (2);
    above the limit respond by lowering it (0).  Same class name, constructor arguments and `action()` as the reference."""

    def __init__(self, TandR_monitor_idx=6, TandR_monitor: str = "avg_room_temp", TandR_monitor_limit: float = 27):
        assert (TandR_monitor == "avg_room_temp") | (TandR_monitor == "crac_return_temp"), \
            f"invalid TandR_monitor monitor string : {TandR_monitor}"
        self.TandR_monitor = TandR_monitor
        self.TandR_monitor_limit = TandR_monitor_limit
        self.TandR_monitor_idx = TandR_monitor_idx
        self.response_duration_counter = 0
        self.response_duration_limit = 4  # 1 hour at a 15-minute sampling interval

    def set_limit(self, x):
        self.TandR_monitor_limit = x

    def action(self, obs):
        curr_val = obs
        if self.TandR_monitor_limit >= curr_val:
            if self.response_duration_counter > self.response_duration_limit:
                self.response_duration_counter = 0
                return 2
            self.response_duration_counter += 1
            return 1
        return 0


# Device-side counterparts: `SdcEngine(policy=(ls, dc, bat))` / sdc_config.policy run these inside the step kernel
# (include/sustaindc_hip.h enum sdc_policy), so `SdcEngine.rollout_policy` plays whole episodes closed-loop without an
# action array:
#   BaseLoadShiftingAgent / BaseHVACAgent / BaseBatteryAgent -> POLICY_DO_NOTHING on that slot
#   RBCBatteryAgent(look_ahead=3, smooth_window=1)           -> POLICY_RBC on the battery slot: fed [ci, ci_future...] of the
#                                                               env's `infos["__common__"]` (sustaindc_env.py:601-603)
#   trim_and_respond_ctrl(limit)                              -> POLICY_TRIM_AND_RESPOND on the dc slot: fed the
#                                                               dc_int_temperature the previous step reported
POLICY_EXTERNAL, POLICY_DO_NOTHING, POLICY_RBC, POLICY_TRIM_AND_RESPOND = 0, 1, 2, 3
