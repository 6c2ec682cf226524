cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
rm -rf gpurun_out/pmc_match_fetch; mkdir -p gpurun_out/pmc_match_fetch
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_match_fetch -o m -- python tools/time_match.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_match_fetch/**/m_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0][:50]].append((float(r['Counter_Value']), int(r['End_Timestamp'])-int(r['Start_Timestamp'])))
for k,v in sorted(acc.items()):
    print(f"{k:52s} n={len(v)} HBM read {sum(a for a,_ in v)/len(v)*2048/1e6:10.1f} MB  {sum(b for _,b in v)/len(v)/1e3:9.1f} us")
PY
rm -rf gpurun_out/pmc_match_fetch
