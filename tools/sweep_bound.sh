#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for tpb in 8 16 32 64; do for gd in 6; do
  M3D_BOUND_TPB=$tpb M3D_BOUND_GDIV=$gd python bench.py --no-cpu-baseline > gpurun_out/sw.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open('gpurun_out/sw.json').read().strip().splitlines()[-1])
print('tpb=$tpb gdiv=$gd', 'ms_per_step', round(d['ms_per_step'], 4))
PY
done; done
