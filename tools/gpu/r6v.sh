cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6v
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size_vs_oracle.py tests/test_gpu_full_size.py tests/test_gpu_scratch_layout.py -x -q -m gpu -k "segment or c5 or tombstone or scratch" > gpurun_out/r6v/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6v/pytest.log
tail -3 gpurun_out/r6v/pytest.log
M3D_C5_REPS=5 python tools/time_c5_plain.py 2>&1 | tail -2 | cut -c1-260
