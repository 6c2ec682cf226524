cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6j
timeout 900 python -m pytest tests/test_gpu_registration.py -x -q -m gpu -k "candidate_cache or prune or ransac_matches or sharded" > gpurun_out/r6j/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6j/pytest.log
tail -3 gpurun_out/r6j/pytest.log
echo "--- sub-walks"; timeout 600 python tools/time_c4_forced.py 2>&1 | grep '"reg_cache": 1' | tee gpurun_out/r6j/c4_sub.txt
echo "--- no sub-walks"; M3D_DBG_NO_SUBWALK=1 timeout 600 python tools/time_c4_forced.py 2>&1 | grep '"reg_cache": 1' | tee gpurun_out/r6j/c4_nosub.txt
