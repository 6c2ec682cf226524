cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6m
timeout 2400 python -m pytest tests/test_gpu_device_aliases.py tests/test_gpu_concurrency.py tests/test_capi_symbols.py tests/test_gpu_sharded_driver.py -x -q > gpurun_out/r6m/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6m/pytest.log
tail -30 gpurun_out/r6m/pytest.log
