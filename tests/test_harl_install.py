"""The zero-edit drop-in (dc_rl_amd.install_into_harl): the reference's OWN runner modules, imported unchanged from /root/reference
(build container only; third-party packages this image lacks are stood in for by tests/golden/_shims and tests/aux/harl_stubs),
end up calling this package's factories -- harl/runners/on_policy_base_runner.py:17-23, :103, :110, :127;
off_policy_base_runner.py:10-16, :77-86."""
import os
import subprocess
import sys
import textwrap

import pytest

from tests.conftest import ROOT

REF = os.environ.get("SDC_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "harl")), reason="needs the reference tree (build container)")

PRE = f"""
import sys
sys.path[:0] = [{os.path.join(ROOT, 'tests', 'aux', 'harl_stubs')!r}, {os.path.join(ROOT, 'tests', 'golden', '_shims')!r}, {REF!r}, {ROOT!r}]
NAMES = ("make_train_env", "make_eval_env", "make_render_env", "get_num_agents")
"""


def _run(body):
    r = subprocess.run([sys.executable, "-c", PRE + textwrap.dedent(body)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_install_before_the_runners_are_imported():
    out = _run("""
        import dc_rl_amd, dc_rl_amd.envs_tools as ours
        bound = dc_rl_amd.install_into_harl()
        assert sorted(bound["harl.utils.envs_tools"]) == sorted(NAMES), bound
        import harl.runners.on_policy_base_runner as on_policy
        import harl.runners.off_policy_base_runner as off_policy
        import harl.utils.envs_tools as theirs
        for mod in (theirs, on_policy, off_policy):
            for n in NAMES:
                assert getattr(mod, n) is getattr(ours, n), (mod.__name__, n)
        # what the runners did not ask to be replaced stays the reference's own
        assert on_policy.set_seed.__module__ == "harl.utils.envs_tools" and on_policy.set_seed is theirs.set_seed
        # the runner classes themselves are the reference's, untouched
        assert on_policy.OnPolicyBaseRunner.__module__ == "harl.runners.on_policy_base_runner"
        import inspect
        src = inspect.getsource(on_policy.OnPolicyBaseRunner.__init__)
        assert "make_train_env(" in src and "make_render_env(" in src
        print("OK")
    """)
    assert "OK" in out


def test_install_after_the_runners_were_imported_and_uninstall():
    out = _run("""
        import harl.runners.on_policy_base_runner as on_policy
        import harl.runners.off_policy_base_runner as off_policy
        import harl.utils.envs_tools as theirs
        orig = {n: getattr(theirs, n) for n in NAMES}
        import dc_rl_amd.envs_tools as ours
        bound = ours.install_into_harl()
        assert set(bound) == {"harl.utils.envs_tools", "harl.runners.on_policy_base_runner", "harl.runners.off_policy_base_runner"}
        for mod in (theirs, on_policy, off_policy):
            for n in NAMES:
                assert getattr(mod, n) is getattr(ours, n), (mod.__name__, n)
        ours.uninstall_from_harl()
        for mod in (theirs, on_policy, off_policy):
            for n in NAMES:
                assert getattr(mod, n) is orig[n], (mod.__name__, n)
        print("OK")
    """)
    assert "OK" in out


def test_install_with_options_binds_wrappers_that_keep_the_references_signature():
    out = _run("""
        import inspect
        import dc_rl_amd.envs_tools as ours
        ours.install_into_harl(device=0, return_torch=True, devices=[0, 0])
        import harl.runners.on_policy_base_runner as on_policy
        f = on_policy.make_train_env
        assert f is not ours.make_train_env and f.__name__ == "make_train_env"
        assert f.keywords == {"device": 0, "return_torch": True, "devices": [0, 0]}
        assert list(inspect.signature(f).parameters)[:4] == ["env_name", "seed", "n_threads", "env_args"]
        assert on_policy.make_eval_env.keywords == {"device": 0, "return_torch": True}
        assert on_policy.make_render_env.keywords == {"device": 0}
        assert on_policy.get_num_agents is ours.get_num_agents
        print("OK")
    """)
    assert "OK" in out
