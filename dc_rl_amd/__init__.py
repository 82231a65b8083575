"""Import shim: the package directory is named `dc-rl_amd` (not a valid Python identifier), so
`import dc_rl_amd` resolves here and forwards to it."""
import os as _os

_real = _os.path.abspath(_os.path.join(_os.path.dirname(__file__), "..", "dc-rl_amd"))
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
