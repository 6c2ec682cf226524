cd $GRAFT_REPO_ROOT
show() { python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if 'ms' in d: print(d['config'][:40], round(d['ms'],4), 'kernel_ms', round(d['roofline']['kernel_ms_total'],4), 'pairs', d['roofline']['tile_hypothesis_pairs'], 'launches', d['roofline']['launches'])
"; }
for ph in -1 0 2 3; do echo "--- score_phases $ph"; M3D_SCORE_PHASES=$ph python tools/bench_configs.py C3 --no-cpu-baseline 2>/dev/null | show; done
for pb in 0 1 2; do echo "--- plane_bound $pb (phases auto)"; M3D_PLANE_BOUND=$pb python tools/bench_configs.py C3 --no-cpu-baseline 2>/dev/null | show; done
