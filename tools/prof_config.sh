#!/bin/bash
# kernel stats of ONE configuration of tools/bench_configs.py on the GPU box: bash tools/prof_config.sh C5 [top-N]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
cfg=${1:-C5}; top=${2:-16}
rm -rf gpurun_out/pc_$cfg; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pc_$cfg -o t -- python tools/bench_configs.py $cfg --no-cpu-baseline > gpurun_out/pc_$cfg.out 2> gpurun_out/pc_$cfg.err
f=$(find gpurun_out/pc_$cfg -name 't_kernel_stats.csv' | head -1)
python - "$f" "$top" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time total %.2f ms, %d launches" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
for r in rows[:int(sys.argv[2])]:
    print(f'{r["Name"][:64]:64s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  total {float(r["TotalDurationNs"])/1e6:7.2f} ms {float(r["TotalDurationNs"])/tot*100:5.1f} %')
PY
cut -c1-300 gpurun_out/pc_$cfg.out
