cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6k
timeout 900 python -m pytest tests/test_gpu_registration.py -x -q -m gpu -k "candidate_cache or prune or ransac_matches or sharded" > gpurun_out/r6k/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6k/pytest.log
tail -3 gpurun_out/r6k/pytest.log
echo "--- pair walk + sub-walks"; timeout 600 python tools/time_c4_forced.py 2>&1 | grep '"reg_cache": 1' | tee gpurun_out/r6k/c4_a.txt
echo "--- pair walk, no sub-walks"; M3D_DBG_NO_SUBWALK=1 timeout 600 python tools/time_c4_forced.py 2>&1 | grep '"reg_cache": 1' | tee gpurun_out/r6k/c4_b.txt
echo "--- block walk, no sub-walks"; M3D_DBG_NO_PAIRWALK=1 M3D_DBG_NO_SUBWALK=1 timeout 600 python tools/time_c4_forced.py 2>&1 | grep '"reg_cache": 1' | tee gpurun_out/r6k/c4_c.txt
