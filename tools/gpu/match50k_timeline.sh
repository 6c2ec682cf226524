cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cat > /tmp/m50.py <<'PY'
import sys, time
sys.path.insert(0, '.')
from misc3d_amd import capi, synth
d = synth.registration_pair_c4(50000, seed=5)
for _ in range(4):
    t0 = time.perf_counter(); i0, i1 = capi.match_mutual_nn(d["feat_src"], d["feat_dst"]); print((time.perf_counter() - t0) * 1e3)
PY
rm -rf /tmp/m50; timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/m50 -o t -- python /tmp/m50.py 2>/dev/null | tail -4
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob('/tmp/m50/**/t_kernel_trace.csv', recursive=True):
    rows += [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:44]) for r in csv.DictReader(open(f))]
for f in glob.glob('/tmp/m50/**/t_memory_copy_trace.csv', recursive=True):
    rows += [(int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r['Direction'][-14:]) for r in csv.DictReader(open(f))]
rows.sort()
idx = [i for i, r in enumerate(rows) if 'mutual_write' in r[2]]
a, b = idx[-2] + 1, idx[-1]
t0 = rows[a][0]; prev = t0
for s, e, n in rows[a:b + 1]:
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f} gap {(s - prev) / 1e3:6.1f}  {n}")
    prev = max(prev, e)
PY
