set -o pipefail
export TMPDIR=/tmp SPHMI_NO_STEPS=1
OUT=gpurun_out/r03q; mkdir -p $OUT
for L in libsphmi.so libsphmi_rhorec.so; do
  for i in 1 2; do SPHMI_LIB=$PWD/smoothed-particle-hydrodynamics_amd/$L timeout -k 10 100 python tools/time_density.py 16M 50 2>&1 | grep -v amdgpu.ids; done
  for C in FETCH_SIZE WRITE_SIZE; do
    SPHMI_LIB=$PWD/smoothed-particle-hydrodynamics_amd/$L timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p_${L}_$C -- python3 tools/time_density.py 16M 5 > $OUT/p_${L}_$C.log 2>&1 || { tail -5 $OUT/p_${L}_$C.log; }
    python tools/pmc_table.py $OUT/p_${L}_$C/*/*counter_collection.csv | grep -E "^kernel|k_density" | sed "s/^/$L /"
    rm -rf $OUT/p_${L}_$C
  done
done
