# kernel trace of the default (two-lane) bench step: gpurun_out/trace_lanes/
set -u
OUT=gpurun_out/trace_lanes
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d $OUT/l2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-prompts > $OUT/l2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/l1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-prompts --lanes 1 > $OUT/l1.log 2>&1
python3 tools/trace_lanes.py $OUT/l2 > $OUT/summary_l2.txt
python3 tools/trace_lanes.py $OUT/l1 > $OUT/summary_l1.txt
cat $OUT/summary_l2.txt $OUT/summary_l1.txt
