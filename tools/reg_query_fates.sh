#!/bin/bash
# Where do the queries of reg_validate_k end?  Builds the library with -DM3D_REG_TRIP_STATS into misc3d_amd/lib/stats (run the
# build where hipcc is; the .so travels with gpurun), then C4's forced run prints the counters on stderr:
#   tools/reg_query_fates.sh build          (here)
#   tools/reg_query_fates.sh run > out.txt  (on the GPU box)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "build" ]; then
    make -C misc3d_amd/csrc lib -j8 DEFS=-DM3D_REG_TRIP_STATS OBJDIR=../lib/obj_stats LIBDIR=../lib/stats > /dev/null
    ls -la misc3d_amd/lib/stats/libmisc3d_amd.so
else
    M3D_LIB_VARIANT=stats python - <<'PY'
import sys, time
import numpy as np
sys.path.insert(0, ".")
from misc3d_amd import capi, synth
n = 200000
d = synth.registration_pair_c4(n, seed=5)
i0, i1 = capi.match_mutual_nn(d["feat_src"], d["feat_dst"])
for conf, label in ((1.0, "forced 100 000 iterations"), (0.999, "the reference's confidence 0.999")):
    t0 = time.perf_counter()
    T, st = capi.registration_ransac(d["src"], d["dst"], i0, i1, threshold=0.03, max_iter=100_000, edge_length_threshold=0.9,
                                     confidence=conf, seed=17)
    print(f"C4 {label}: {(time.perf_counter() - t0) * 1e3:.1f} ms (diagnostic build), validations {st['validations']}", flush=True)
PY
fi
