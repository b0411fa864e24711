#!/bin/bash
# GPU-box script: SQ instruction counters of k_find_neighbors for the default library and diagnostic variants. Usage: tools/gpu_fn_variant_pmc.sh TAG lib...
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for lib in default "$@"; do
  if [ "$lib" != default ]; then export SPHMI_LIB=$PWD/smoothed-particle-hydrodynamics_amd/libsphmi_$lib.so; fi
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY \
    --output-format csv -d $OUT/p_$lib -- python3 tools/time_find_neighbors.py 5 > $OUT/p_$lib.log 2>&1 || { tail -5 $OUT/p_$lib.log; continue; }
  python tools/pmc_table.py $OUT/p_$lib/*/*counter_collection.csv | grep -E "^kernel|find_neighbors" | sed "s/^/$lib: /"
  rm -rf $OUT/p_$lib
done
