cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6n
bash tools/pmc_boundary.sh > gpurun_out/r6n/pmc_boundary.txt 2>&1
tail -12 gpurun_out/r6n/pmc_boundary.txt
timeout 1500 python tools/bench_configs.py > gpurun_out/r6n/bench_configs.jsonl 2> gpurun_out/r6n/bench_configs.err
tail -5 gpurun_out/r6n/bench_configs.err
python - <<'PY'
import json
for l in open('gpurun_out/r6n/bench_configs.jsonl'):
    d=json.loads(l)
    cb=d.get('cpu_baseline') or {}
    rf=d.get('roofline') or {}
    print(d['config'][:60], '| ms', round(d.get('ms', d.get('ms_total', 0)),3), '| cpu', (round(cb['value'],1), cb.get('unit','')[:24]) if 'value' in cb else cb, '| roof', round(rf.get('frac',0),3) if rf else None)
PY
