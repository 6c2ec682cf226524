#!/bin/bash
# Kernel + copy timeline of ONE m3d_global_registration call (the last of six), on the GPU box:
#   [M3D_N2_POINTS=200000] bash tools/global_registration_timeline.sh > gpurun_out/r05_global_registration_timeline.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
rm -rf gpurun_out/grt
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/grt -o t -- python tools/time_global_registration.py > gpurun_out/grt.out 2>&1
tail -3 gpurun_out/grt.out
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/grt/**/t_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
for f in glob.glob("gpurun_out/grt/**/t_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
rows.sort()
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - rows[i - 1][1] > 700_000:
        cut = i
sel = rows[cut:]
t0 = sel[0][0]
prev = t0
for s, e, n in sel:
    if "fillBuffer" in n: 
        prev = max(prev, e); continue
    print(f"{(s - t0) / 1e3:9.1f} us  + {(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:7.1f}  {n}")
    prev = max(prev, e)
print(f"first to last: {(max(r[1] for r in sel) - t0) / 1e3:.1f} us over {len(sel)} operations")
PY
