#!/bin/bash
# SQ counters of the scoring launches INSIDE the timed fit of bench.py (lead launch and pruned main launch of
# score_screen_k<0>, told apart by their grid size); one counter group per pass, --kernel-trace only.
# M3D_PMC_CMD: another command to profile instead (e.g. "python tools/bench_configs.py C3": the sphere's and cylinder's).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_score_bench
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_INST_LEVEL_LDS SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o r -- ${M3D_PMC_CMD:-python bench.py --steps 10 --warmup 2 --no-cpu-baseline} > $OUT/g$i.out 2> $OUT/g$i.err
done
python - "$OUT" <<'PY'
import csv, collections, glob, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/g*/r_counter_collection.csv") + glob.glob(out + "/g*/*/r_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "score_screen_k" not in k and "cull_lead_k" not in k and "score_mfma_k" not in k:
            continue
        key = (k, r["Grid_Size"])
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        n[key][r["Counter_Name"]] += 1
for key, d in sorted(acc.items()):
    print(key)
    for c, v in sorted(d.items()):
        print(f"   {c:24s} {v / n[key][c]:16.0f} per launch   ({n[key][c]} launches)")
PY
