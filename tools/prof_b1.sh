set -u
OUT=gpurun_out/prof_b1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b1 -- python3 tools/latency_b1.py 1 > $OUT/b1.log 2>&1
f=$(find $OUT/b1 -name "*kernel_stats.csv" | head -1)
cut -c1-200 $f | head -14
