#!/bin/bash
# Kernel timeline of ONE compute_transformation_ransac call with the reference's confidence (C4 sizes: 200 k <-> 200 k): the set-up is
# what the call consists of.  On the GPU box: bash tools/c4_default_timeline.sh > gpurun_out/r05_c4_default_timeline.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
rm -rf gpurun_out/c4d
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/c4d -o t -- python tools/time_c4_default.py > gpurun_out/c4d.out 2>&1
tail -5 gpurun_out/c4d.out
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/c4d/**/t_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
for f in glob.glob("gpurun_out/c4d/**/t_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
rows.sort()
# the last call: everything after the last gap of more than 2 ms
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - rows[i - 1][1] > 2_000_000:
        cut = i
sel = rows[cut:]
t0 = sel[0][0]
prev = t0
for s, e, n in sel:
    print(f"{(s - t0) / 1e3:9.1f} us  + {(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:7.1f}  {n}")
    prev = e
print(f"first to last: {(sel[-1][1] - t0) / 1e3:.1f} us over {len(sel)} operations")
PY
