"""Stand-in for PsychroLib used ONLY by tests/golden/gen_golden.py (see gymnasium stub).

Delegates to the product's own restatement (dc_rl_amd/psychro.py).  Wet-bulb
*generation* is therefore "parity unpinned" (SURVEY.md section 8c); the WB table the
reference ends up with is captured into the fixtures as an input.
"""
import importlib.util
import os

_p = os.path.join(os.path.dirname(__file__), "..", "..", "..", "dc_rl_amd", "psychro.py")
_spec = importlib.util.spec_from_file_location("_sdc_psychro", os.path.abspath(_p))
_m = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_m)

SI = 2


def SetUnitSystem(_units):
    return None


def GetTWetBulbFromRelHum(t, rh, p):
    return _m.t_wet_bulb_from_rel_hum(float(t), float(rh), float(p))
