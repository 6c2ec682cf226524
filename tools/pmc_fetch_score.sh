#!/bin/bash
# HBM bytes of the scoring launches of bench.py (FETCH_SIZE / WRITE_SIZE in separate passes, --kernel-trace only; KiB; FETCH_SIZE x 2
# on gfx950: MI355X_MICROARCH.md, calibrated in profiles/r0*_pmc_summary.json).  On the GPU box: bash tools/pmc_fetch_score.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_fetch_score
rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o r -- ${M3D_PMC_CMD:-python bench.py --steps 10 --warmup 2 --no-cpu-baseline} > $OUT/$c.out 2> $OUT/$c.err
done
python - "$OUT" <<'PY'
import csv, collections, glob, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/r_counter_collection.csv") + glob.glob(out + "/*/*/r_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "score_screen_k" in k or "cull_lead_k" in k or "plane_bound_k" in k:
            acc[(k, r["Grid_Size"])][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for key, d in sorted(acc.items()):
    rd = [v for v, _ in d.get("FETCH_SIZE", [])]
    wr = [v for v, _ in d.get("WRITE_SIZE", [])]
    ns = [t for _, t in d.get("FETCH_SIZE", [])]
    if rd:
        print(f"{key[0]} grid {key[1]}: launches {len(rd)}  HBM read {sum(rd) / len(rd) * 2048 / 1e6:8.2f} MB  written {(sum(wr) / max(len(wr), 1)) * 1024 / 1e6:7.2f} MB  per launch, {sum(ns) / len(ns) / 1e3:7.1f} us under the counter pass")
PY
