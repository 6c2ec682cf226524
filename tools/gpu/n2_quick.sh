cd $GRAFT_REPO_ROOT
python tools/bench_configs.py N2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('N2', {k:round(v['pairs_per_s']) for k,v in d['in_flight'].items()}, {k:round(v,3) for k,v in d['per_pair_serial_ms'].items()})"
python - <<'PY'
import sys, time
sys.path.insert(0, '.')
import numpy as np
from misc3d_amd import capi, synth
for n in (20000, 50000, 100000):
    d = synth.registration_pair_c4(n, seed=5)
    capi.match_mutual_nn(d["feat_src"], d["feat_dst"])
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); i0, i1 = capi.match_mutual_nn(d["feat_src"], d["feat_dst"]); ts.append((time.perf_counter() - t0) * 1e3)
    print(n, "match ms", [round(t, 3) for t in ts], len(i0), capi.match_last_fallbacks())
PY
