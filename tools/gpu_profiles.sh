#!/bin/bash
# GPU-box script: the evidence kept under profiles/<round>/ — parity suite log, the driver's bench line, rocprofv3 kernel statistics
# of the same bench command, HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) per kernel at 16.5 M particles, SQ counters of
# findNeighbors at 1 M. Usage: tools/gpu_profiles.sh TAG   (results under gpurun_out/TAG/)
set -o pipefail
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1 || { tail -40 $OUT/gpu_tests.log; exit 1; }
tail -2 $OUT/gpu_tests.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -20 $OUT/bench_default.err; exit 1; }
cat $OUT/bench_default.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-stage-pass --no-extra > $OUT/kt.log 2>&1 || { tail -20 $OUT/kt.log; exit 1; }
cp $OUT/kt/*/*kernel_stats.csv $OUT/config4_16M_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt1 -- python3 bench.py --workload config2_1M_cube --steps 50 --warmup 5 --cpu-steps 0 --no-stage-pass --no-extra > $OUT/kt1.log 2>&1 || { tail -20 $OUT/kt1.log; exit 1; }
cp $OUT/kt1/*/*kernel_stats.csv $OUT/config2_1M_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pf -- python3 tools/profile_step.py 16M 3 > $OUT/pf.log 2>&1 || { tail -20 $OUT/pf.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pw -- python3 tools/profile_step.py 16M 3 > $OUT/pw.log 2>&1 || { tail -20 $OUT/pw.log; exit 1; }
python tools/pmc_table.py $OUT/pf/*/*counter_collection.csv > $OUT/pmc_FETCH_SIZE_16M.txt
python tools/pmc_table.py $OUT/pw/*/*counter_collection.csv > $OUT/pmc_WRITE_SIZE_16M.txt
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --output-format csv -d $OUT/sa -- python3 tools/time_find_neighbors.py 5 > $OUT/sa.log 2>&1 || { tail -20 $OUT/sa.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD \
  --output-format csv -d $OUT/sb -- python3 tools/time_find_neighbors.py 5 > $OUT/sb.log 2>&1 || { tail -20 $OUT/sb.log; exit 1; }
python tools/pmc_table.py $OUT/sa/*/*counter_collection.csv > $OUT/pmc_SQ_A_1M.txt
python tools/pmc_table.py $OUT/sb/*/*counter_collection.csv > $OUT/pmc_SQ_B_1M.txt
rm -rf $OUT/kt $OUT/kt1 $OUT/pf $OUT/pw $OUT/sa $OUT/sb
grep -E "kernel|k_density|find_neighbors" $OUT/pmc_FETCH_SIZE_16M.txt $OUT/pmc_WRITE_SIZE_16M.txt $OUT/pmc_SQ_A_1M.txt | cut -c1-220
head -12 $OUT/config4_16M_kernel_stats.csv | cut -c1-200
