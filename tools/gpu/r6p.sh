cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6p
timeout 900 python -m pytest tests/test_gpu_concurrency.py -x -q -k "fit_batch or set_config" > gpurun_out/r6p/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6p/pytest.log
tail -15 gpurun_out/r6p/pytest.log
