cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6e
timeout 600 python tools/time_c4_forced.py > gpurun_out/r6e/c4_forced.txt 2>&1
cat gpurun_out/r6e/c4_forced.txt
echo "--- no phase 2 (wrong results, timing only)"
M3D_DBG_NO_P2=1 timeout 600 python tools/time_c4_forced.py > gpurun_out/r6e/c4_forced_nop2.txt 2>&1
cat gpurun_out/r6e/c4_forced_nop2.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6e/prof -o c4 -- python tools/gpu/c4on.py > gpurun_out/r6e/prof.out 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r6e/prof/**/c4_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:6]:
    print(f"{r['Name'].replace('void ','').split('(')[0][:50]:50s} calls={r['Calls']:>5s} total_ms={float(r['TotalDurationNs'])/1e6:10.2f} avg_us={float(r['AverageNs'])/1e3:10.1f} pct={r['Percentage']}")
PY
