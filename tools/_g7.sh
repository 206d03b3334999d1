mkdir -p gpurun_out/r06g
for mode in self shared fixed; do
  ( timeout 900 python tools/repro_four_procs.py --mode $mode --iters 50 --procs 4 --noise 900 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06g/repro4_${mode}_50.txt
  tail -2 gpurun_out/r06g/repro4_${mode}_50.txt | cut -c1-200
done
( NOISE_TOOL=noise_latency.py timeout 900 python tools/repro_four_procs.py --mode self --iters 30 --procs 8 --noise 900 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06g/repro8_self_lat_30.txt
tail -2 gpurun_out/r06g/repro8_self_lat_30.txt
