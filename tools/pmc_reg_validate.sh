#!/bin/bash
# SQ counters of the registration validation kernels on C4 (one counter group per pass; --kernel-trace only, as the
# pool requires).  Usage on the GPU box: bash tools/pmc_reg_validate.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_reg_0
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
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o r -- python $OUT/run.py > $OUT/g$i.out 2> $OUT/g$i.err
done
python - "$OUT" <<'PY'
import csv, collections, glob, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(int)
for f in glob.glob(out + "/g*/r_counter_collection.csv") + glob.glob(out + "/g*/*/r_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "reg_validate" not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {v:16.0f}")
PY
