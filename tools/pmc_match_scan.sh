#!/bin/bash
# SQ counters of the matcher's MAIN scan launch alone (nn16_scan_k<false, true>; both matrices uploaded first, so that the scan
# is one launch): instruction mix per wave-tile-pair and where the waves wait.  Run on the GPU box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp M3D_MATCH_PIPELINE=0
rm -rf gpurun_out/pmc_scan; mkdir -p gpurun_out/pmc_scan
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_IFETCH"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_scan/$tag -o m -- python tools/time_match.py > /dev/null 2> gpurun_out/pmc_scan/$tag.err
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_scan/*/m_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'nn16_scan_kILb0ELb1E' in r['Kernel_Name'] or 'nn16_scan_k<false, true>' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
pairs = (200000 / 64) * 6250   # wave-tile-pairs of one scan
for k, v in sorted(acc.items()):
    a = sum(v) / len(v)
    print(f"{k:34s} n={len(v)} avg={a:.4g}   per wave-tile-pair {a / pairs:.2f}")
PY
