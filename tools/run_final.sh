# Final measurement set of a round (GPU box): python tests, benches of every BASELINE config, latency table, microbenchmarks.
TAG=${1:-r2}
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -1 $OUT/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_base8.json 2> $OUT/bench_base8.err
python bench.py --steps 20 --warmup 5 --lanes 1 --no-cpu-baseline --no-prompts > $OUT/bench_base8_lanes1.json 2>/dev/null
python bench.py --workload large4 --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_large4.json 2>/dev/null
python bench.py --workload imu4 --steps 8 --warmup 2 > $OUT/bench_imu4.json 2>/dev/null
python bench.py --workload prompts256 --steps 8 --warmup 2 > $OUT/bench_prompts256.json 2>/dev/null
python tools/latency.py > $OUT/latency.log 2>&1
python tools/microbench.py gemm > $OUT/microbench_gemm_b8.log 2>&1
python tools/mb_attn.py > $OUT/microbench_attn_remap.log 2>&1
SHAPES=b1 VARIANTS=0:0,0:32,0:4 python tools/mb_variants.py > $OUT/microbench_gemm_b1.log 2>&1
VARIANTS=0:0,1:0,1:64,4:0 python tools/mb_variants.py > $OUT/microbench_gemm_variants.log 2>&1
python tools/mb_fold.py > $OUT/microbench_ln_fold.log 2>&1
python tools/ab_step.py ln_fuse 0 1 2 > $OUT/ab_ln_fuse.log 2>&1
python tools/ab_step.py attn_remap 1 0 2 > $OUT/ab_attn_remap.log 2>&1
python tools/power_probe.py > $OUT/power_probe.log 2>&1
for f in base8 base8_lanes1 large4 imu4 prompts256; do python - $OUT/bench_$f.json <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith('{'):
        d=json.loads(line); print(sys.argv[1].split('/')[-1], 'value %.1f ms %.2f' % (d['value'], d['ms_per_step']), 'frac', d.get('roofline',{}).get('frac'), 'prompts', (d.get('prompts256') or {}).get('value'), 'fast', (d.get('secondary') or {}).get('value'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
done
cat $OUT/latency.log
