"""Kernel timeline of the LAST fit of a kind in a rocprofv3 kernel trace: python tools/fit_timeline.py <trace.csv> <kind 0|1|2>
(start offset, duration, gap to the previous kernel's end, stream) -- fits of several chunks (C3) with their pre-stream kernels."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
kind = sys.argv[2] if len(sys.argv) > 2 else "2"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fits = [i for i, r in enumerate(rows) if f"compact_write_k<{kind}, 0" in r["Kernel_Name"]]
# the last stretch between two compactions of that kind that holds a whole fit (a RefineModel retry or a consensus-only call leaves
# stretches of two or three kernels)
a = b = None
for j in range(len(fits) - 1, 0, -1):
    if fits[j] - fits[j - 1] > 12:
        a, b = fits[j - 1] + 1, fits[j] + 1
        break
if a is None:
    raise SystemExit("no complete fit of that kind in the trace")
t0 = int(rows[a]["Start_Timestamp"])
prev = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").split("(")[0]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {(s - prev) / 1e3:7.1f}  q{r.get('Queue_Id', '?')}  {name[:44]}  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}")
    prev = max(prev, e)
print(f"first kernel to the end of the list's compaction: {(prev - t0) / 1e3:.1f} us")
