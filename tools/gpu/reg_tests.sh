cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/regt
timeout 1500 python -m pytest tests/test_gpu_registration.py tests/test_gpu_concurrency.py tests/test_gpu_full_size_vs_oracle.py tests/test_gpu_boundary.py -x -q -m gpu -k "not mutual and not c2 and not c3 and not c5" > gpurun_out/regt/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/regt/pytest.log
tail -3 gpurun_out/regt/pytest.log
python tools/time_c4_default.py | tail -3
python tools/bench_configs.py N2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('N2', {k:round(v['pairs_per_s']) for k,v in d['in_flight'].items()}, d['per_pair_serial_ms'])"
