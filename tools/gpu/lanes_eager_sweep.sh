cd $GRAFT_REPO_ROOT
show() { python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if 'LANES' in d['config']: print({m:{k:round(v['fits_per_s']) for k,v in d[m].items()} for m in ('resident','one_shot','batch')})
    elif 'N2' in d['config']: print('N2', {k:round(v['pairs_per_s']) for k,v in d['in_flight'].items()}, {k:round(v['pairs_per_s']) for k,v in d['fragments_resident'].items()})
    elif 'ms' in d: print(d['config'][:50], round(d['ms'],3))
"; }
for mode in "0 4" "4 8" "8 8" "4 4"; do set -- $mode; echo "--- eager lanes $1, hw queues $2"; M3D_DBG_LANES_EAGER=$1 GPU_MAX_HW_QUEUES=$2 python tools/bench_configs.py C2 C3 C4 C5 N2 LANES --no-cpu-baseline 2>/dev/null | show; done
