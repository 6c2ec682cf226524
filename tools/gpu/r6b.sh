cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6b
timeout 300 python tools/gpu/dbg_cache.py > gpurun_out/r6b/dbg.txt 2>&1
cat gpurun_out/r6b/dbg.txt | head -60
cat > /tmp/c4on.py <<'P'
import os, sys, time
sys.path.insert(0, os.getcwd())
from misc3d_amd import capi, synth
d = synth.registration_pair_c4(200000, seed=5)
i0, i1 = capi.match_mutual_nn(d["feat_src"], d["feat_dst"])
for _ in range(2):
    t0=time.perf_counter()
    T, st = capi.registration_ransac(d["src"], d["dst"], i0, i1, threshold=0.03, max_iter=100000, edge_length_threshold=0.9, confidence=1.0, seed=17)
    print((time.perf_counter()-t0)*1e3, st)
P
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r6b/prof -o c4 -- python /tmp/c4on.py > $GRAFT_REPO_ROOT/gpurun_out/r6b/prof.out 2>&1
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/r6b/prof.out
find gpurun_out/r6b/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {} | cut -c1-200'
