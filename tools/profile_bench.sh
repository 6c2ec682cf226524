#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
rm -rf gpurun_out/prof_b; mkdir -p gpurun_out/prof_b
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b -o b -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_b/b.out 2> gpurun_out/prof_b/b.err
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_b/b_kernel_stats.csv')))
for r in rows[:14]:
    print(f"{r['Name'].replace('void ','').split('(')[0]:34s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} min_us={float(r['MinNs'])/1e3:9.1f} max_us={float(r['MaxNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
