#!/bin/bash
# cull_lead_k's duration against the number of leading hypotheses (M3D_LEAD): which half of the fused launch binds?
# On the GPU box: bash tools/sweep_lead.sh > gpurun_out/sweep_lead.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for lead in ${LEADS:-64 128 256 512}; do
    rm -rf gpurun_out/sl_$lead
    M3D_LEAD=$lead timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sl_$lead -o t -- python bench.py --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/sl_$lead.json 2> gpurun_out/sl_$lead.err
    f=$(find gpurun_out/sl_$lead -name 't_kernel_stats.csv' | head -1)
    echo "== M3D_LEAD=$lead  $(tail -1 gpurun_out/sl_$lead.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],4))")"
    python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("cull_lead_k", "plane_bound_k", "score_screen_k", "lead_fold", "sum_replicas", "minimal_fit", "compact_")):
        print(f"   {n.split('(')[0][:60]:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
    rm -rf gpurun_out/sl_$lead
done
