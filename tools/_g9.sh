rocm-smi --showuniqueid 2>/dev/null | grep -i "unique" | head -2
python -m pytest tests/test_gpu_cli.py tests/test_gpu_shards.py -m gpu -q -x 2>&1 | tail -5
bash tools/profile_gpu.sh v5 --topology v5_cvector > gpurun_out/prof_v5.log 2>&1; tail -3 gpurun_out/prof_v5.log
