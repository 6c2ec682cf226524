#!/bin/bash
# sweep the workgroup-count target of the scoring launch (tail effect vs re-read traffic)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for w in 2048 4096 8192 16384 32768; do
  echo "== M3D_SCORE_WGS=$w"; M3D_SCORE_WGS=$w python tools/pmc_target.py 2>&1 | grep "score_k ms"
done
