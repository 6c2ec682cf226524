cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6i
timeout 900 python -m pytest tests/test_gpu_registration.py -x -q -m gpu -k "candidate_cache or prune or ransac_matches or sharded" > gpurun_out/r6i/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6i/pytest.log
tail -5 gpurun_out/r6i/pytest.log
for ph in 4; do echo "--- phases $ph"; M3D_DBG_PHASES=$ph timeout 600 python tools/time_c4_forced.py 2>&1 | tee gpurun_out/r6i/c4_forced_ph$ph.txt; done
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6i/prof -o c4 -- python tools/gpu/c4on.py > gpurun_out/r6i/prof.out 2>&1
tail -2 gpurun_out/r6i/prof.out | cut -c1-600
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r6i/prof/**/c4_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:10]:
    print(f"{r['Name'].replace('void ','').split('(')[0][:50]:50s} calls={r['Calls']:>5s} total_ms={float(r['TotalDurationNs'])/1e6:10.2f} avg_us={float(r['AverageNs'])/1e3:10.1f} pct={r['Percentage']}")
PY
