"""One-off long-horizon parity run (not part of the test suite): the production rig of tests/production_rig.py -- 4096 envs,
debug_flags 0, rings full, device resets, deferred re-centring -- for many whole 672-step episodes, sampled envs against the
fp64 oracle on every step.  usage: python tools/long_parity.py [episodes] [n_envs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.production_rig import ProductionRig
episodes = int(sys.argv[1]) if len(sys.argv) > 1 else 30
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
t0 = time.time()
rig = ProductionRig(N, debug_flags=0, episode_steps=672, seed=2024, envs_per_wave=4 if N >= 5636 else 2, n_random=40)
obs, _ = rig.eng.reset()
rig.begin_all(obs)
rig.single_steps(672 * episodes + 5)
rig.assert_ok()
print(f"{N} envs, {episodes} episodes of 672 steps ({672 * episodes + 5} steps), {len(rig.sample)} sampled envs = "
      f"{len(rig.sample) * (672 * episodes + 5)} env-steps against the oracle: worst relative error {rig.worst}, "
      f"reward-state paths (none / inline / deferred / rebuilt) {rig.paths[:4].tolist()}, auto-resets {rig.resets}, "
      f"{time.time() - t0:.0f} s")
