cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6o
for g in 0 1 2 4 8 16; do echo "--- own-tests groups per block: $g (0 = product rule)"; M3D_DBG_OWN_GPB=$g M3D_C5_REPS=4 python tools/time_c5_plain.py 2>&1 | tee -a gpurun_out/r6o/sweep.txt; done
