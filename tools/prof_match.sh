#!/bin/bash
# kernel stats of the C4 matcher (200 k x 200 k x 33): bash tools/prof_match.sh [out]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
out=${1:-pm}
rm -rf gpurun_out/$out; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$out -o t -- python tools/time_match.py > gpurun_out/$out.out 2> gpurun_out/$out.err
f=$(find gpurun_out/$out -name 't_kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:9.1f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
PY
cat gpurun_out/$out.out
