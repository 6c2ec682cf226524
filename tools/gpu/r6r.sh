cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6r
timeout 1200 python -m pytest tests/test_gpu_registration.py tests/test_gpu_full_size_vs_oracle.py -x -q -m gpu -k "candidate_cache or prune or ransac_matches or sharded or c4 or overhang or grid_edge" > gpurun_out/r6r/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6r/pytest.log
tail -3 gpurun_out/r6r/pytest.log
timeout 600 python tools/time_c4_forced.py 2>&1 | tee gpurun_out/r6r/c4_forced.txt
