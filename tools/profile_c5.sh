#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
rm -rf gpurun_out/prof_c5; mkdir -p gpurun_out/prof_c5
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5 -o c5 -- python tools/bench_configs.py C5 > gpurun_out/prof_c5/c5.out 2> gpurun_out/prof_c5/c5.err
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_c5/c5_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:16]:
    print(f"{r['Name'].replace('void ','').split('(')[0]:34s} calls={r['Calls']:>5s} total_ms={float(r['TotalDurationNs'])/1e6:8.2f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
cut -c1-160 gpurun_out/prof_c5/c5.out
