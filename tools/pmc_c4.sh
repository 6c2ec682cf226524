#!/bin/bash
# FETCH_SIZE pass over the C4 registration run (reg_validate_k memory traffic); run on the GPU box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_c4; mkdir -p gpurun_out/pmc_c4
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_c4 -o c4 -- python tools/bench_configs.py C4 > gpurun_out/pmc_c4/c4.out 2> gpurun_out/pmc_c4/c4.err
python - <<'PY'
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/pmc_c4/c4_counter_collection.csv')))
tr={r['Dispatch_Id']:r for r in csv.DictReader(open('gpurun_out/pmc_c4/c4_kernel_trace.csv'))} if False else {}
acc=collections.defaultdict(lambda:[0,0.0])
for r in rows:
    if r['Counter_Name']!='FETCH_SIZE': continue
    k=r['Kernel_Name'].split('(')[0].replace('void ','')
    acc[k][0]+=1; acc[k][1]+=float(r['Counter_Value'])
for k,(n,v) in sorted(acc.items(), key=lambda kv:-kv[1][1])[:6]:
    print(f"{k:40s} launches={n:5d} FETCH_SIZE(KB)={v:14.0f}  => HBM bytes (x2 gfx950 correction, KB->B) = {v*1024*2/1e9:10.2f} GB")
PY
