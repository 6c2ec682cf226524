"""Per-kernel timeline of ONE bench step from a rocprofv3 kernel trace (csv): start offset, duration and the gap to the
previous kernel, for one step of the trace.  Usage: python tools/step_timeline.py <trace.csv> [first-kernel-substring] [step index]
(default: the last complete step; bench.py ends with its one-shot fits, so the resident-cloud step wants an index,
e.g. 15 = inside the timed region)"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "minimal_fit_k"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
if len(starts) < 3:
    raise SystemExit("not enough steps in the trace")
k = int(sys.argv[3]) if len(sys.argv) > 3 else len(starts) - 3
a, b = starts[k], starts[k + 1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").split("(")[0]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f}  {name}  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}")
    prev_end = e
print(f"step period (first kernel to first kernel): {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us; last kernel of the step ends at {(prev_end - t0) / 1e3:.1f} us")
