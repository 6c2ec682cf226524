cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
rm -rf gpurun_out/c3p; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c3p -o t -- python bench.py --workload c3cyl --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/c3p.json 2>gpurun_out/c3p.err
f=$(find gpurun_out/c3p -name 't_kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  {float(r["TotalDurationNs"])/tot*100:5.1f} %')
PY
tail -1 gpurun_out/c3p.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])"
