#!/bin/bash
# A/B of the histogram bound on C3's fits (cylinder: on by default since round 5; sphere: only when forced), the two settings in turn
# on ONE box -- boxes of the pool differ by up to 15 % on the same build --, then the cylinder fit's kernel timeline.  Run on the GPU box:
#   bash tools/ab_bound_c3.sh > gpurun_out/r05_c3_bound_ab.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for rep in 1 2 3; do for m in 0 1 2; do
  M3D_PLANE_BOUND=$m python tools/bench_configs.py C3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    r = d['roofline']
    print('plane_bound=$m  %-30s ms %.4f  (without timing events %.4f)  (tile, hypothesis) pairs %8d  scoring launches %.4f ms  best %d  inliers %d' % (d['config'], d['ms'], d['ms_without_timing_events'], r['tile_hypothesis_pairs'], r['kernel_ms_total'], d['best_index'], d['n_inliers']))"
done; done
echo; echo "kernel timeline of one fit_cylinder (plane_bound = 1):"
bash tools/c3_timeline.sh 2>&1 | sed -n '/== fit_cylinder/,/== fit_sphere/p' | head -40
