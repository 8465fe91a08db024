#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel stats + PMC passes for the bench command and the L/4 decoder
# attention; outputs under gpurun_out/prof_$1 (copy the summaries you want judged into profiles/).
set -u
TAG=${1:-r1}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# --lanes 1: the configuration of bench.py's kernel region (launches alone on the chip), which is what `roofline` is taken from
BENCH="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --no-prompts --lanes 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_parity -- $BENCH > $OUT/stats_parity.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_fast -- $BENCH --mode fast > $OUT/stats_fast.log 2>&1
# the default command (two lanes in the timed region + the one-lane kernel region): launch durations of the first overlap
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_parity_default -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --no-prompts > $OUT/stats_parity_default.log 2>&1
# HBM traffic of the dominant kernel (separate passes: FETCH_SIZE uses 3 of the 4 TCC slots)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -- $BENCH > /dev/null 2>&1
# ViT-L/4 decoder attention (B=8, H=8, N=6272): MFMA busy
for mode in fast parity; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_attn_l4dec_$mode -- python3 tools/one_kernel.py attn 8 8 6272 $mode > /dev/null 2>&1
done
# batch-1 latency (the small-launch kernels: 4-stage ring, split-K)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b1 -- python3 tools/latency_b1.py 1 > $OUT/stats_b1.log 2>&1
find $OUT -name "*.csv" | head -40
