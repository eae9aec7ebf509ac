mkdir -p gpurun_out/r06
F="--no-cpu-baseline --no-reference-width --no-scan-form --no-hash-path --extra-queries \"\" --steady-steps 300"
for cfg in "base|" "always|SDQLPY_AMD_PLAN_GRAPHS_ALWAYS=1" ; do
  name=${cfg%%|*}; envs=${cfg#*|}
  for order in given q5,q3,q1 q3,q5,q1 q5,q1,q3; do
    env $envs python bench.py --no-cpu-baseline --no-reference-width --no-scan-form --no-hash-path --extra-queries "" --steady-steps 300 --launch-order $order > gpurun_out/r06/exp_${name}_${order}.json 2>/dev/null
    python - <<PY
import json
d=json.loads(open("gpurun_out/r06/exp_${name}_${order}.json").read().strip().splitlines()[-1])
print("$name", "$order", "ms_per_step", d["ms_per_step"], "steady", d["steady_state"]["ms_per_step"], "host", d["step"]["host_launch_ms"], "graphs", d["step"]["plan_graphs"])
PY
  done
done
