cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6l
timeout 1500 python -m pytest tests/test_gpu_registration.py tests/test_gpu_full_size_vs_oracle.py tests/test_gpu_full_size.py tests/test_gpu_concurrency.py tests/test_gpu_sharded_driver.py tests/test_gpu_two_ranks.py -x -q -m gpu -k "not c5 and not c2 and not c3" > gpurun_out/r6l/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6l/pytest.log
tail -4 gpurun_out/r6l/pytest.log
timeout 600 python tools/time_c4_forced.py 2>&1 | tee gpurun_out/r6l/c4_forced.txt
