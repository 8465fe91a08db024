# Final measurement set of a round (GPU box): python tests, benches of every BASELINE config, latency table, microbenchmarks.
#   gpurun --timeout 3000 -- 'bash tools/run_final.sh r6 > gpurun_out/final_r6.log 2>&1'
TAG=${1:-r6}
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; tail -1 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_base8.json 2> $OUT/bench_base8.err
timeout 600 python bench.py --steps 20 --warmup 5 --lanes 1 --no-cpu-baseline --no-prompts > $OUT/bench_base8_lanes1.json 2>/dev/null
timeout 600 python bench.py --workload large4 --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_large4.json 2> $OUT/bench_large4.err
timeout 600 python bench.py --workload large4 --mode fast --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_large4_fast.json 2>/dev/null
timeout 600 python bench.py --workload imu4 --steps 8 --warmup 2 > $OUT/bench_imu4.json 2> $OUT/bench_imu4.err
timeout 600 python bench.py --workload imu4 --mode fast --steps 8 --warmup 2 > $OUT/bench_imu4_fast.json 2>/dev/null
timeout 600 python bench.py --workload prompts256 --steps 8 --warmup 2 > $OUT/bench_prompts256.json 2>/dev/null
timeout 300 python bench.py --workload flowstats --steps 10 > $OUT/bench_flowstats.json 2> $OUT/bench_flowstats.err
timeout 300 python bench.py --workload prompt_build --steps 20 > $OUT/bench_prompt_build.json 2> $OUT/bench_prompt_build.err
timeout 600 python tools/latency.py > $OUT/latency.log 2>&1
timeout 600 python tools/microbench.py gemm > $OUT/microbench_gemm_b8.log 2>&1
timeout 600 python tools/microbench.py attn > $OUT/microbench_attn.log 2>&1
timeout 600 python tools/ab_direct.py --no-check > $OUT/ab_gemm_direct.log 2>&1
timeout 600 python tools/ab_step.py gemm_direct 0 1 2 > $OUT/ab_step_gemm_direct.log 2>&1
KEY=attn_ksplit timeout 600 python tools/mb_attn.py > $OUT/microbench_attn_ksplit.log 2>&1
timeout 600 python tools/ab_conj.py conj_ctx_stream 0 1 2 > $OUT/ab_conj_ctx_stream.log 2>&1
timeout 400 bash tools/run_bench_2ranks_1gpu.sh > $OUT/bench_2ranks_1gpu.log 2>&1
for f in base8 base8_lanes1 large4 large4_fast imu4 imu4_fast prompts256; do python - $OUT/bench_$f.json <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith('{'):
        d=json.loads(line); r=d.get('roofline',{})
        print(sys.argv[1].split('/')[-1], 'value %.1f ms %.2f' % (d['value'], d['ms_per_step']), 'dominant', r.get('kernel'), 'frac', r.get('frac'), 'traffic', r.get('traffic'), 'busy', r.get('mfma_busy'),
              'prompts', (d.get('prompts256') or {}).get('value'), 'fast', (d.get('secondary') or {}).get('value'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
done
cat $OUT/latency.log
# a Python traceback in any output of the set = the set is not evidence
if grep -l "Traceback (most recent call last)" $OUT/*.log $OUT/*.err 2>/dev/null; then echo "TRACEBACK in the files listed above"; exit 1; fi
