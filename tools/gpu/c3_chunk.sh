cd $GRAFT_REPO_ROOT
show() { python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if 'ms' in d: print(d['config'][:40], round(d['ms'],4), 'kernel_ms', round(d['roofline']['kernel_ms_total'],4), 'pairs', d['roofline']['tile_hypothesis_pairs'])
"; }
for i in 1 2 3; do python tools/bench_configs.py C3 --no-cpu-baseline 2>/dev/null | show; done
