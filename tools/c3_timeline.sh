#!/bin/bash
# Kernel timeline of one C3 fit_cylinder and one fit_sphere (1 M points, 50 000 hypotheses), on the GPU box:
#   bash tools/c3_timeline.sh > gpurun_out/c3_timeline.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
rm -rf gpurun_out/c3t
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/c3t -o t -- python tools/bench_configs.py C3 --no-cpu-baseline > /dev/null 2>&1
f=$(find gpurun_out/c3t -name 't_kernel_trace.csv' | head -1)
echo "== fit_cylinder"; python tools/fit_timeline.py "$f" 2
echo "== fit_sphere"; python tools/fit_timeline.py "$f" 1
rm -rf gpurun_out/c3t
