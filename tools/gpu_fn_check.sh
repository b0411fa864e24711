#!/bin/bash
# GPU-box script: parity suite + findNeighbors timing + phase shares. Usage: tools/gpu_fn_check.sh TAG
set -o pipefail
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1 || { tail -40 $OUT/gpu_tests.log; exit 1; }
tail -2 $OUT/gpu_tests.log
timeout -k 10 120 python tools/time_find_neighbors.py 20 | tee $OUT/fn_time.txt
timeout -k 10 120 python tools/time_find_neighbors.py 20 0.3 | tee -a $OUT/fn_time.txt
if [ -f smoothed-particle-hydrodynamics_amd/libsphmi_stamps.so ]; then timeout -k 10 120 python tools/fn_phase_shares.py 5 | tee $OUT/fn_phases.txt; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-steps 0 > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python - <<P
import json
d=json.load(open("$OUT/bench.json"))
print("16M ms/step", d["ms_per_step"], d["stages_ms"])
print("1M", d["config2_1M_cube"]["ms_per_step"], d["config2_1M_cube"]["stages_ms"])
P
