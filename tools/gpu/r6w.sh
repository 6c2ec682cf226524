cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6w
timeout 1500 python -m pytest tests/test_gpu_match_sliced.py tests/test_gpu_registration.py tests/test_gpu_concurrency.py tests/test_gpu_full_size_vs_oracle.py -x -q -m gpu -k "mutual or match or c4 or global or concurrent or fragment" > gpurun_out/r6w/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6w/pytest.log
tail -3 gpurun_out/r6w/pytest.log
