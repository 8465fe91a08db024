#!/bin/bash
# Build container, after `gpurun -- 'bash tools/run_final.sh TAG; bash tools/collect_profiles.sh TAG'`: copy what is judged from the scratch directory
# gpurun_out/ into profiles/ (tracked) -- the bench lines, the tool logs, and (tools/summarize_profiles.py) the rocprofv3 summaries.  Fails if any of it holds a traceback.
TAG=${1:-r6}
F=gpurun_out/final_$TAG
set -e
for w in base8 base8_lanes1 large4 large4_fast imu4 imu4_fast prompts256 flowstats prompt_build; do grep '^{' $F/bench_$w.json | tail -1 > profiles/bench_${w}_$TAG.json; done
grep -v amdgpu.ids $F/latency.log > profiles/${TAG}_latency_small_batch.log
grep -v amdgpu.ids $F/microbench_gemm_b8.log > profiles/${TAG}_microbench_gemm_b8.log
grep -v amdgpu.ids $F/microbench_attn.log > profiles/${TAG}_microbench_attn.log
grep -v amdgpu.ids $F/bench_2ranks_1gpu.log > profiles/${TAG}_bench_2ranks_1gpu.log
grep -v amdgpu.ids $F/ab_gemm_direct.log > profiles/${TAG}_ab_gemm_direct.log
python tools/summarize_profiles.py $TAG > /tmp/summarize_$TAG.txt 2>&1 || { cat /tmp/summarize_$TAG.txt; exit 1; }
grep "L/4" /tmp/summarize_$TAG.txt
if grep -l "Traceback (most recent call last)" profiles/${TAG}_* profiles/*_$TAG.json 2>/dev/null; then echo "TRACEBACK in the files listed above"; exit 1; fi
grep -h "passed\|failed" $F/pytest_gpu.log | tail -1
tail -1 $F/smoke.log
