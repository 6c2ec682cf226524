cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
timeout 900 python -m pytest tests/test_gpu_registration.py -x -q -m gpu -k "candidate_cache or nn_screen or prune or ransac_matches or sharded" > gpurun_out/r6a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6a/pytest.log
tail -15 gpurun_out/r6a/pytest.log
timeout 600 python tools/time_c4_forced.py > gpurun_out/r6a/c4_forced.txt 2>&1
cat gpurun_out/r6a/c4_forced.txt
