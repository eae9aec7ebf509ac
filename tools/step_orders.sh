mkdir -p gpurun_out/r04
for o in given q5,q3,q1 q3,q5,q1 q5,q1,q3 q1,q5,q3 q3,q1,q5; do
  python bench.py --no-cpu-baseline --no-reference-width --extra-queries "" --launch-order $o --steady-steps 1000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$o', d['ms_per_step'], d['steady_state']['ms_per_step'], round(d['value']/1e9,1))"
done
for l in 2 4 6; do
  python bench.py --no-cpu-baseline --no-reference-width --extra-queries "" --lanes $l --steady-steps 1000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lanes $l', d['ms_per_step'], d['steady_state']['ms_per_step'], round(d['value']/1e9,1))"
done
