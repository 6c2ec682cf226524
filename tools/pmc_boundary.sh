#!/bin/bash
# SQ counters of boundary_k (N4) on bench_configs' own workload: instructions per point for its issue roofline.
# Usage on the GPU box: bash tools/pmc_boundary.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_boundary
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o r -- python tools/bench_configs.py N4 --no-cpu-baseline > $OUT/g$i.out 2> $OUT/g$i.err
done
python - "$OUT" <<'PY'
import csv, collections, glob, sys
out = sys.argv[1]
acc = collections.defaultdict(float)
calls = collections.defaultdict(int)
for f in glob.glob(out + "/g*/r_counter_collection.csv") + glob.glob(out + "/g*/*/r_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "boundary_k" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        calls[r["Counter_Name"]] += 1
print("boundary_k, summed over its launches in one `bench_configs.py N4` run (a 1000-point warm-up + 3 x 500 000 points)")
for c, v in sorted(acc.items()):
    print(f"   {c:28s} {v:16.0f}   launches {calls[c]}")
pts = 1000 + 3 * 500000
if acc.get("SQ_INSTS_VALU"):
    print(f"per wave of 64 points (= per point: one point per lane): VALU {acc['SQ_INSTS_VALU'] / (pts / 64.0):.0f}  LDS {acc.get('SQ_INSTS_LDS', 0) / (pts / 64.0):.0f}  "
          f"SALU {acc.get('SQ_INSTS_SALU', 0) / (pts / 64.0):.0f}  VMEM_RD {acc.get('SQ_INSTS_VMEM_RD', 0) / (pts / 64.0):.0f}   (bench_configs.py N4_COUNTS = VALU, LDS)")
PY
