cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6s
show() { python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if 'LANES' in d['config']: print({m:{k:round(v['fits_per_s']) for k,v in d[m].items()} for m in ('resident','one_shot','batch')})
    elif 'ms' in d: print(d['config'][:50], round(d['ms'],3))
"; }
echo "--- default queues: C4 then LANES"; python tools/bench_configs.py C4 LANES --no-cpu-baseline 2>/dev/null | show
echo "--- GPU_MAX_HW_QUEUES=8: C4 then LANES"; GPU_MAX_HW_QUEUES=8 python tools/bench_configs.py C4 LANES --no-cpu-baseline 2>/dev/null | show
echo "--- GPU_MAX_HW_QUEUES=8: C2 C3"; GPU_MAX_HW_QUEUES=8 python tools/bench_configs.py C2 C3 --no-cpu-baseline 2>/dev/null | show
echo "--- default: C2 C3"; python tools/bench_configs.py C2 C3 --no-cpu-baseline 2>/dev/null | show
