#!/bin/bash
# GPU-box script: FETCH_SIZE / WRITE_SIZE of k_density at 16.5 M on the library as built (separate passes), for profiles/density_traffic.json
set -o pipefail
export TMPDIR=/tmp SPHMI_NO_STEPS=1
OUT=gpurun_out/${1:-traffic}; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p_$C -- python3 tools/time_density.py 16M 5 > $OUT/p_$C.log 2>&1 || { tail -5 $OUT/p_$C.log; exit 1; }
  python tools/pmc_table.py $OUT/p_$C/*/*counter_collection.csv | grep -E "^kernel|k_density" | tee -a $OUT/density_pmc.txt
  rm -rf $OUT/p_$C
done
