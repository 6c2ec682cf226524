#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for rep in 1 2; do for gpb in 8 12 16; do
  M3D_GPB=$gpb python bench.py --no-cpu-baseline > gpurun_out/sw.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open('gpurun_out/sw.json').read().strip().splitlines()[-1])
print('gpb=$gpb', 'ms_per_step', round(d['ms_per_step'], 4), 'launch', round(d['roofline']['launch_ms'], 4), 'frac', round(d['roofline']['frac'], 3), 'plain', round(d['without_kernel_timing_events']['ms_per_step'], 4))
PY
done; done
