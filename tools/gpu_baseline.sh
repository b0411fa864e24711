#!/bin/bash
# GPU-box script: parity suite, default bench line, and SQ/LDS counter passes on findNeighbors. Usage: tools/gpu_baseline.sh TAG
set -o pipefail
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1 || { tail -30 $OUT/gpu_tests.log; exit 1; }
tail -2 $OUT/gpu_tests.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -20 $OUT/bench_default.err; exit 1; }
cat $OUT/bench_default.json
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --output-format csv -d $OUT/pmcA -- python3 tools/time_find_neighbors.py 5 > $OUT/pmcA.log 2>&1 || { tail -20 $OUT/pmcA.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES \
  --output-format csv -d $OUT/pmcB -- python3 tools/time_find_neighbors.py 5 > $OUT/pmcB.log 2>&1 || { tail -20 $OUT/pmcB.log; exit 1; }
python tools/pmc_table.py $OUT/pmcA/*/*counter_collection.csv > $OUT/pmc_SQ_A.txt
python tools/pmc_table.py $OUT/pmcB/*/*counter_collection.csv > $OUT/pmc_SQ_B.txt
grep -E "kernel|find_neighbors" $OUT/pmc_SQ_A.txt $OUT/pmc_SQ_B.txt
rm -rf $OUT/pmcA $OUT/pmcB
