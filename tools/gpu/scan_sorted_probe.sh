cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for srt in 1024 256; do
  echo "== SORTED=$srt"
  SORTED=$srt M3D_LIB_VARIANT=st M3D_MATCH_PIPELINE=0 python tools/gpu/scan_sorted_probe.py 2>&1 | tail -2
  rm -rf /tmp/sab; SORTED=$srt M3D_MATCH_PIPELINE=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sab -o m -- python tools/gpu/scan_sorted_probe.py > /dev/null 2>&1
  python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/sab/**/m_kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:60]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if 'nn16' in k or 'verify' in k or 'rev_bin' in k: print(f"   {k:60s} n={len(v):3d} avg {sum(v)/len(v):9.1f} us  min {min(v):9.1f}")
PY
done
