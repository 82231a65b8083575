#!/bin/bash
# build locally (sources -> in-tree .so), then run the given command on an MI355X box
set -e
cd /root/repo
python - <<'PY'
import sys; sys.path.insert(0, '.')
from dc_rl_amd import _lib
import os, glob
so = _lib.LIB_PATH
src = glob.glob(os.path.join(_lib.CSRC, '*.h*')) + ['include/sustaindc_hip.h']
if not os.path.exists(so) or max(os.path.getmtime(f) for f in src) > os.path.getmtime(so):
    _lib.build(verbose=False); print('rebuilt', so)
PY
T=${GTIMEOUT:-900}
exec timeout $((T + 900)) gpurun --timeout $T -- "$@"
