cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6f
for ph in 4 8; do echo "--- phases $ph"; M3D_DBG_PHASES=$ph timeout 600 python tools/time_c4_forced.py 2>&1 | tee gpurun_out/r6f/c4_forced_ph$ph.txt; done
