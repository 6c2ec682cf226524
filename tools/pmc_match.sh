#!/bin/bash
# SQ counters of the matcher's scan kernel (run on the GPU box)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_match; mkdir -p gpurun_out/pmc_match
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_match/$tag -o m -- python tools/time_match.py > /dev/null 2> gpurun_out/pmc_match/$tag.err
done
python - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_match/*/m_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'nn16_scan_k' in r['Kernel_Name'] or 'nn32_scan_k' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(f"{k:36s} n={len(v)} avg={sum(v)/len(v):.4g}")
PY
