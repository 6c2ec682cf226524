#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for o in hilbert morton; do echo "== M3D_ORDER=$o"; M3D_ORDER=$o python tools/pmc_target.py 2>&1 | grep "score_list_k\|plane fit" | cut -c1-230; done
