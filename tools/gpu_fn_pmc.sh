#!/bin/bash
# GPU-box script (a refused counter set leaves rocprofv3 hanging: every pass runs under timeout, `python3 script` directly after --): SQ counter passes + phase stamps of findNeighbors on the config #2 cube. Usage: tools/gpu_fn_pmc.sh TAG
set -o pipefail
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 120 python tools/time_find_neighbors.py 20 | tee $OUT/fn_time.txt
if [ -f smoothed-particle-hydrodynamics_amd/libsphmi_stamps.so ]; then timeout -k 10 120 python tools/fn_phase_shares.py 5 | tee $OUT/fn_phases.txt; fi
timeout -k 10 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --output-format csv -d $OUT/pmcA -- python3 tools/time_find_neighbors.py 5 > $OUT/pmcA.log 2>&1 || { tail -20 $OUT/pmcA.log; exit 1; }
timeout -k 10 150 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD \
  --output-format csv -d $OUT/pmcB -- python3 tools/time_find_neighbors.py 5 > $OUT/pmcB.log 2>&1 || { tail -20 $OUT/pmcB.log; exit 1; }
timeout -k 10 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_LEVEL_WAVES \
  --output-format csv -d $OUT/pmcC -- python3 tools/time_find_neighbors.py 5 > $OUT/pmcC.log 2>&1 || { tail -20 $OUT/pmcC.log; }
# (pass C may be refused on some boxes: its table only if its CSV exists)
if ls $OUT/pmcC/*/*counter_collection.csv > /dev/null 2>&1; then python tools/pmc_table.py $OUT/pmcC/*/*counter_collection.csv > $OUT/pmc_SQ_C.txt; else echo "pass C produced no counters" > $OUT/pmc_SQ_C.txt; fi
python tools/pmc_table.py $OUT/pmcA/*/*counter_collection.csv > $OUT/pmc_SQ_A.txt
python tools/pmc_table.py $OUT/pmcB/*/*counter_collection.csv > $OUT/pmc_SQ_B.txt
grep -E "kernel|find_neighbors" $OUT/pmc_SQ_A.txt $OUT/pmc_SQ_B.txt $OUT/pmc_SQ_C.txt
grep -h find_neighbors $OUT/pmcA/*/*kernel_trace.csv | head -3
rm -rf $OUT/pmcA $OUT/pmcB $OUT/pmcC
