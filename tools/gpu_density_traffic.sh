#!/bin/bash
# HBM bytes of the density pass from PMC counters (separate FETCH_SIZE / WRITE_SIZE passes), 16.5 M and 1 M particles.
set -o pipefail
OUT=gpurun_out/${1:-traffic}
mkdir -p $OUT
export TMPDIR=/tmp
for sz in 16M 1M; do
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pf$sz -- python3 tools/profile_step.py $sz 3 > $OUT/pf$sz.log 2>&1 || { tail -20 $OUT/pf$sz.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pw$sz -- python3 tools/profile_step.py $sz 3 > $OUT/pw$sz.log 2>&1 || { tail -20 $OUT/pw$sz.log; exit 1; }
python tools/pmc_table.py $OUT/pf$sz/*/*counter_collection.csv > $OUT/pmc_FETCH_SIZE_$sz.txt
python tools/pmc_table.py $OUT/pw$sz/*/*counter_collection.csv > $OUT/pmc_WRITE_SIZE_$sz.txt
rm -rf $OUT/pf$sz $OUT/pw$sz
done
grep -h "kernel\|k_density" $OUT/pmc_*.txt
