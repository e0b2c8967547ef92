timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for v in 0 1; do
if [ $v = 1 ]; then export TLOAM_NO_HOST_MIRROR=1; fi
timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('kitti pair ms/frame', d['ms_per_step'], 'GN it/s', d['value'], 'm1', d['m1_frame']['ms_per_frame'], 'seq', d.get('kitti_sequence', {}).get('ms_per_frame'), 'odo', d.get('odometry_loop', {}).get('ms_per_frame'))"
done
