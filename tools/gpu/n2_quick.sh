cd $GRAFT_REPO_ROOT
python tools/time_c4_default.py | tail -3
python tools/bench_configs.py N2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k:round(v['pairs_per_s']) for k,v in d['in_flight'].items()}, {k:round(v,3) for k,v in d['per_pair_serial_ms'].items()}, {k:round(v['pairs_per_s']) for k,v in d['fragments_resident'].items()})"
