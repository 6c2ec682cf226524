#!/bin/bash
# SQ counters of plane_bound_k inside bench.py's steps (run on the GPU box)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
rm -rf gpurun_out/pmc_bound
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_bound/a -o a -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_GDS --kernel-trace --output-format csv -d gpurun_out/pmc_bound/b -o b -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_bound/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'plane_bound_k' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print(f"{k:28s} mean {sum(v)/len(v):14.0f}  (n={len(v)})")
PY
