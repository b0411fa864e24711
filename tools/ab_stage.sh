mkdir -p gpurun_out
for sz in 1M 16M; do for st in predict_density pressure_force forces; do
python tools/time_stage.py $st 20 $sz 2>&1 | grep -v amdgpu | tee -a gpurun_out/ab_stage.txt
done; done
