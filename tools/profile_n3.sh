#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
rm -rf gpurun_out/prof_n3; mkdir -p gpurun_out/prof_n3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_n3 -o n3 -- python tools/bench_configs.py N3 > gpurun_out/prof_n3/n3.out 2> gpurun_out/prof_n3/n3.err
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_n3/n3_kernel_stats.csv')))
for r in rows[:8]:
    print(f"{r['Name'].replace('void ','').split('(')[0]:40s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} min_us={float(r['MinNs'])/1e3:9.1f} max_us={float(r['MaxNs'])/1e3:9.1f}")
PY
cat gpurun_out/prof_n3/n3.out | cut -c1-200
