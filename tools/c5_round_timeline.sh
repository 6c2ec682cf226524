#!/bin/bash
# kernel trace of one C5 call (10 M-point segmentation) + the launches of a few tail rounds: bash tools/c5_round_timeline.sh [out] [first minimal_fit_k #]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
out=${1:-c5t}; first=${2:-200}
rm -rf gpurun_out/$out
M3D_C5_REPS=2 timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$out -o t -- python tools/time_c5_plain.py > gpurun_out/$out.out 2> gpurun_out/$out.err
f=$(find gpurun_out/$out -name 't_kernel_trace.csv' | head -1)
python - "$f" "$first" <<'PY' | tee gpurun_out/${out}_timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mf = [i for i, r in enumerate(rows) if "minimal_fit_k" in r["Kernel_Name"]]
a = mf[int(sys.argv[2])]
b = mf[int(sys.argv[2]) + 6]
t0 = int(rows[a]["Start_Timestamp"]); prev = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("m3d::", "").split("(")[0]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:6.1f}  {name}  grid {r['Grid_Size_X']} wg {r['Workgroup_Size_X']}")
    prev = e
print(f"{(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us for 3 rounds")
# totals per kernel over the whole trace
from collections import defaultdict
tot = defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["Kernel_Name"].replace("void ", "").replace("m3d::", "").split("(")[0]
    tot[n][0] += 1; tot[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print("kernel ms total %.2f, launches %d" % (sum(v[1] for v in tot.values()), sum(v[0] for v in tot.values())))
for n, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{n:44s} calls {v[0]:5d} total {v[1]:7.2f} ms avg {v[1] / v[0] * 1e3:7.1f} us")
PY
cat gpurun_out/$out.out
