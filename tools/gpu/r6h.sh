cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6h
timeout 900 python -m pytest tests/test_gpu_registration.py -x -q -m gpu -k "candidate_cache or prune or ransac_matches or sharded" > gpurun_out/r6h/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6h/pytest.log
tail -5 gpurun_out/r6h/pytest.log
for ph in 4 5 6; do echo "--- phases $ph"; M3D_DBG_PHASES=$ph timeout 600 python tools/time_c4_forced.py 2>&1 | tee gpurun_out/r6h/c4_forced_ph$ph.txt; done
