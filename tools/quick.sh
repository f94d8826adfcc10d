#!/bin/bash
# quick GPU check: the bitwise / parity tests of the hot kernels + the two bench blocks (developer tool)
# usage on the box: bash tools/quick.sh [tag]
T=${1:-q}
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_wide.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/$T.tests.log 2>&1 || { tail -30 gpurun_out/$T.tests.log; exit 1; }
tail -1 gpurun_out/$T.tests.log
timeout -k 10 120 python bench.py --contacts --envs 65536 --steps 200 --warmup 50 --no-cpu-baseline > gpurun_out/$T.c.log 2>&1 || exit 1
timeout -k 10 120 python bench.py --envs 4096 --steps 1000 --warmup 100 --no-cpu-baseline > gpurun_out/$T.f.log 2>&1 || exit 1
python - <<PY
import json
for f in ("gpurun_out/$T.c.log", "gpurun_out/$T.f.log"):
    r = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f, round(r["value"] / 1e6, 2), "M env-steps/s", r["ms_per_step"], "ms/step", r["roofline"].get("kernel"))
PY
