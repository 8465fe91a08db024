mkdir -p gpurun_out/r2d
python -m pytest tests -m gpu -x -q > gpurun_out/r2d/pytest_full.log 2>&1; tail -3 gpurun_out/r2d/pytest_full.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2d/bench_base8.json 2> gpurun_out/r2d/bench_base8.err
python bench.py --workload large4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2d/bench_large4.json 2>&1
python bench.py --workload imu4 --steps 5 --warmup 2 > gpurun_out/r2d/bench_imu4.json 2>&1
for f in base8 large4 imu4; do python - gpurun_out/r2d/bench_$f.json <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith('{'):
        d=json.loads(line); print(sys.argv[1], 'value %.1f ms %.2f' % (d['value'], d['ms_per_step']), d.get('roofline',{}).get('frac'), (d.get('prompts256') or {}).get('value'), (d.get('secondary') or {}).get('value'))
PY
done
