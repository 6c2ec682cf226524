#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_c4
M3D_C4_POINTS=${M3D_C4_POINTS:-200000} timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c4 -o c4 -- python tools/bench_configs.py C4 --no-cpu-baseline > gpurun_out/prof_c4/c4.out 2> gpurun_out/prof_c4/c4.err
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_c4/c4_kernel_stats.csv')))
for r in rows[:12]:
    print(f"{r['Name'].replace('void ','').split('(')[0]:34s} calls={r['Calls']:>5s} total_ms={float(r['TotalDurationNs'])/1e6:10.2f} avg_us={float(r['AverageNs'])/1e3:10.1f} pct={r['Percentage']}")
PY
