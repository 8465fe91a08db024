#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel stats + PMC passes for every bench workload (base8 = BASELINE configs[1], large4 =
# configs[2], imu4 = configs[4]) on ONE lane (bench.py's kernel region: launches alone on the chip), the default two-lane command, fast
# mode, the ViT-L/4 decoder attention alone and the batch-1 latency run.  Outputs under gpurun_out/prof_$1; tools/summarize_profiles.py
# $1 (build container) turns them into profiles/$1_* and profiles/pmc_summary_latest.json.
# Counters are collected in their own passes with --kernel-trace only (FETCH_SIZE uses 3 of the 4 TCC slots, WRITE_SIZE 2).
set -u
TAG=${1:-r4}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# (the IMU model's context stream runs on the lane's own stream in these runs: with the side stream two kernels overlap and a kernel's
# duration in the trace is not its own)
export CWM_CONJ_CTX_STREAM=0
fail=0
for wl in base8 large4 imu4; do
  steps=5; [ $wl = base8 ] || steps=3
  BENCH="python3 bench.py --workload $wl --steps $steps --warmup 1 --no-cpu-baseline --no-secondary --no-prompts --lanes 1"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_${wl}_parity -- $BENCH > $OUT/stats_${wl}_parity.log 2>&1
  grep '^{"metric"' $OUT/stats_${wl}_parity.log | tail -1 > $OUT/bench_under_rocprof_${wl}_parity.json
  [ -s $OUT/bench_under_rocprof_${wl}_parity.json ] || { echo "NO BENCH LINE for $wl (see $OUT/stats_${wl}_parity.log)"; fail=1; }
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$wl -- $BENCH > $OUT/pmc_fetch_$wl.log 2>&1 || fail=1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$wl -- $BENCH > $OUT/pmc_write_$wl.log 2>&1 || fail=1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma_$wl -- $BENCH > $OUT/pmc_mfma_$wl.log 2>&1 || fail=1
done
unset CWM_CONJ_CTX_STREAM
BENCH="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --no-prompts"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_base8_fast -- $BENCH --lanes 1 --mode fast > $OUT/stats_base8_fast.log 2>&1
# the default command (two lanes in the timed region + the one-lane kernel region): launch durations of the first overlap
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_base8_parity_default -- $BENCH > $OUT/stats_base8_parity_default.log 2>&1
# ViT-L/4 decoder attention (B=8, H=8, N=6272): MFMA busy (north star: >= 40 %)
for mode in fast parity; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_attn_l4dec_$mode -- python3 tools/one_kernel.py attn 8 8 6272 $mode > $OUT/pmc_attn_l4dec_$mode.log 2>&1
done
# batch-1 latency (the small-launch kernels: 4-stage ring, split-K)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b1 -- python3 tools/latency_b1.py 1 > $OUT/stats_b1.log 2>&1
# the kernels either side of the predictor path: kernel stats + FETCH / WRITE passes of the two edge workloads
for wl in flowstats prompt_build; do
  BENCH="python3 bench.py --workload $wl --steps 5 --no-cpu-baseline"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_${wl}_f32 -- $BENCH > $OUT/stats_${wl}_f32.log 2>&1
  grep '^{"metric"' $OUT/stats_${wl}_f32.log | tail -1 > $OUT/bench_under_rocprof_${wl}_f32.json
  [ -s $OUT/bench_under_rocprof_${wl}_f32.json ] || { echo "NO BENCH LINE for $wl (see $OUT/stats_${wl}_f32.log)"; fail=1; }
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$wl -- $BENCH > $OUT/pmc_fetch_$wl.log 2>&1 || fail=1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$wl -- $BENCH > $OUT/pmc_write_$wl.log 2>&1 || fail=1
done
# the covariance kernel runs on the fp32 matrix pipe: its MFMA-busy share and clock
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma_flowstats -- python3 bench.py --workload flowstats --steps 5 --no-cpu-baseline > $OUT/pmc_mfma_flowstats.log 2>&1 || fail=1
find $OUT -name "*.csv" | wc -l
# a Python traceback in any log of the set = the set is not evidence (round 5 committed one as a "per-shape kernel rate" log)
if grep -l "Traceback (most recent call last)" $OUT/*.log 2>/dev/null; then echo "TRACEBACK in the logs listed above"; fail=1; fi
exit $fail
