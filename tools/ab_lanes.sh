set -o pipefail
mkdir -p gpurun_out/r02g
L2=$PWD/smoothed-particle-hydrodynamics_amd/libsphmi_lanes2.so
python tools/time_find_neighbors.py 20
SPHMI_LIB=$L2 python tools/time_find_neighbors.py 20
python tools/time_find_neighbors.py 20 0.3
SPHMI_LIB=$L2 python tools/time_find_neighbors.py 20 0.3
SPHMI_LIB=$L2 timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02g/gpu_tests_lanes2.log 2>&1; tail -3 gpurun_out/r02g/gpu_tests_lanes2.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02g/gpu_tests_lanes4.log 2>&1; tail -3 gpurun_out/r02g/gpu_tests_lanes4.log
