#!/bin/bash
# GPU-box script: texture-addresser / L1 counters of the three gather kernels (tools/time_stage.py, config #2 cube).
# Usage: tools/gpu_gather_pmc.sh TAG [stage ...]
set -o pipefail
TAG=${1:-run}; shift
STAGES=${@:-predict_density pressure_force forces}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
PASS_A="GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum"
PASS_B="TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_GATE_EN2_sum"
PASS_D="GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES"
PASS_C="TA_BUFFER_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"
for st in $STAGES; do
  for pass in ${PASSES:-A B C D}; do
    eval "ctrs=\$PASS_$pass"
    # (each block takes only a few counters per pass: a refused set aborts the tool, which then does not exit by itself)
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/p${pass}_$st -- python3 tools/time_stage.py $st 5 > $OUT/p${pass}_$st.log 2>&1 \
      || { grep -v "^    @" $OUT/p${pass}_$st.log | tail -8; echo "pass $pass of $st failed"; continue; }
    python tools/pmc_table.py $OUT/p${pass}_$st/*/*counter_collection.csv > $OUT/pmc_TA_${pass}_$st.txt
    grep -E "^kernel|k_predict_density|k_pressure_force|k_forces" $OUT/pmc_TA_${pass}_$st.txt
    rm -rf $OUT/p${pass}_$st
  done
done
