mkdir -p gpurun_out/lanes
for l in 2 3 4 2 4; do
python bench.py --steps 20 --warmup 5 --lanes $l --no-cpu-baseline --no-prompts --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes $l value %.1f ms %.2f' % (d['value'], d['ms_per_step']))"
done
