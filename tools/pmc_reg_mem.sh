#!/bin/bash
# Memory-pipeline counters (TA / TCP / TD) of reg_validate_k on a 20 000-iteration C4 run; one group per pass.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_reg_mem
rm -rf $OUT; mkdir -p $OUT
cat > $OUT/run.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from misc3d_amd import capi, synth
d = synth.registration_pair_c4(200000, seed=5)
i0, i1 = capi.match_mutual_nn(d["feat_src"], d["feat_dst"])
T, st = capi.registration_ransac(d["src"], d["dst"], i0, i1, threshold=0.03, max_iter=20000, edge_length_threshold=0.9, confidence=1.0, seed=17)
print(st)
PY
i=0
for grp in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TD_TD_BUSY_sum TCP_UTCL1_TRANSLATION_MISS_sum" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o r -- python $OUT/run.py > $OUT/g$i.out 2> $OUT/g$i.err
done
python - "$OUT" <<'PY'
import csv, collections, glob, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/g*/r_counter_collection.csv") + glob.glob(out + "/g*/*/r_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "reg_validate" not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]] += 1
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:44s} {v:18.0f}   ({n[k][c]} rows)")
PY
