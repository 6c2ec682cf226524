#!/bin/bash
# One C2 kernel trace + the step timeline, on the GPU box: bash tools/quick_timeline.sh [out-name]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
out=${1:-qt}
rm -rf gpurun_out/$out
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$out -o t -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/$out.json 2> gpurun_out/$out.err
f=$(find gpurun_out/$out -name 't_kernel_trace.csv' | head -1)
python tools/step_timeline.py "$f" | tee gpurun_out/${out}_timeline.txt

tail -1 gpurun_out/$out.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step')}, d.get('roofline'))"
