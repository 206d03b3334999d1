mkdir -p gpurun_out/r06f
nproc; free -g | head -2; ulimit -l
( timeout 400 taskset -c 0 python tools/calib_stability.py --procs 4 --reps 15 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06f/cal4_onecpu.txt
tail -3 gpurun_out/r06f/cal4_onecpu.txt | cut -c1-200
( timeout 400 taskset -c 0,1 python tools/repro_four_procs.py --iters 10 --procs 4 --extra --calibrate=true 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06f/repro4_twocpu.txt
tail -4 gpurun_out/r06f/repro4_twocpu.txt | cut -c1-200
# host hogs on every core
for i in $(seq 1 $(nproc)); do ( timeout 150 python -c "
while True: pass" & ) ; done
( timeout 400 python tools/repro_four_procs.py --iters 10 --procs 4 --extra --calibrate=true 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06f/repro4_hogs.txt
tail -4 gpurun_out/r06f/repro4_hogs.txt | cut -c1-200
( timeout 300 python tools/calib_stability.py --procs 4 --reps 10 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06f/cal4_hogs.txt
tail -3 gpurun_out/r06f/cal4_hogs.txt | cut -c1-200
wait
