cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6q
timeout 1200 python -m pytest tests/test_gpu_plane_bound.py tests/test_gpu_full_size_vs_oracle.py -x -q -m gpu -k "bound or c3" > gpurun_out/r6q/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6q/pytest.log
tail -4 gpurun_out/r6q/pytest.log
python tools/bench_configs.py C3 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print(d['config'], round(d['ms'],4), 'plain', round(d['ms_without_timing_events'],4), 'frac', round(d['roofline']['frac'],3), 'kernel_ms', round(d['roofline']['kernel_ms_total'],4), 'pairs', d['roofline']['tile_hypothesis_pairs'])
"
bash tools/c3_timeline.sh 2>/dev/null | grep -E "plane_bound|score_screen_k<2>|first kernel" | head -12
