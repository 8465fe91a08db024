#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel stats + MFMA-busy PMC pass for BASELINE configs[2] (ViT-L/4 batch 8) and configs[4]
# (IMU-conditioned ViT-B/4 batch 16), one lane (launches alone on the chip).  Outputs under gpurun_out/prof_$1; the kernel-stats
# CSVs are copied to gpurun_out/prof_$1/<tag>_kernel_stats_bench_{large4,imu4}_parity.csv for profiles/.
set -u
TAG=${1:-r3}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for wl in large4 imu4; do
  BENCH="python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-prompts --lanes 1"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$wl -- $BENCH > $OUT/stats_$wl.log 2>&1
  cp $(find $OUT/stats_$wl -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats_bench_${wl}_parity.csv
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma_$wl -- $BENCH > /dev/null 2>&1
  python3 tools/pmc_mfma_summary.py $OUT/pmc_mfma_$wl > $OUT/${TAG}_pmc_mfma_${wl}.json
  tail -1 $OUT/stats_$wl.log > $OUT/${TAG}_bench_under_rocprof_${wl}.json
done
ls $OUT
