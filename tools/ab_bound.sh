#!/bin/bash
# A/B of the planes' histogram bound on the bench workload + the step's kernel timeline (run on the GPU box)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for m in 0 1; do
  M3D_PLANE_BOUND=$m python bench.py --no-cpu-baseline > gpurun_out/bench_pb$m.json 2> gpurun_out/bench_pb$m.err
  python - <<PY
import json
d = json.loads(open('gpurun_out/bench_pb$m.json').read().strip().splitlines()[-1])
r = d['roofline']
print('plane_bound=$m', 'ms_per_step', round(d['ms_per_step'], 4), 'value', round(d['value'] / 1e6, 2), 'launch_ms', round(r.get('launch_ms', 0), 4), 'pairs', r.get('tile_hypothesis_pairs_per_launch'), 'frac', round(r['frac'], 3))
PY
done
rm -rf gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o b -- python bench.py --steps 30 --no-cpu-baseline > /dev/null 2>&1
f=$(find gpurun_out/tl -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py $f minimal_fit_k 15 2>&1 | tail -14
