mkdir -p gpurun_out/r06m
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id" | head -1
python -m pytest tests -m gpu -q -rs -x 2>&1 | tail -30 > gpurun_out/r06m/gpu_tests.txt
tail -8 gpurun_out/r06m/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r06m/bench.json 2> gpurun_out/r06m/bench.err; python - <<'PY'
import json
j=json.loads(open('gpurun_out/r06m/bench.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic'], j['roofline']['avg_launch_ms'], j['cpu_baseline']['value'])
PY
