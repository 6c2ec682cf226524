cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6g
timeout 900 python -m pytest tests/test_gpu_registration.py -x -q -m gpu -k "candidate_cache or nn_screen or prune or ransac_matches or sharded" > gpurun_out/r6g/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6g/pytest.log
tail -8 gpurun_out/r6g/pytest.log
timeout 600 python tools/time_c4_forced.py > gpurun_out/r6g/c4_forced.txt 2>&1
cat gpurun_out/r6g/c4_forced.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6g/prof -o c4 -- python tools/gpu/c4on.py > gpurun_out/r6g/prof.out 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r6g/prof/**/c4_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:12]:
    print(f"{r['Name'].replace('void ','').split('(')[0][:50]:50s} calls={r['Calls']:>5s} total_ms={float(r['TotalDurationNs'])/1e6:10.2f} avg_us={float(r['AverageNs'])/1e3:10.1f} pct={r['Percentage']}")
PY
