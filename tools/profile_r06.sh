#!/bin/bash
# Round-6 evidence, run ON THE GPU BOX (gpurun): everything lands under gpurun_out/prof/ (+ prof_c4, prof_cfg), and
# tools/summarize_profiles.py r06 condenses it into profiles/.  PMC passes are separate runs with --kernel-trace only.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
rm -rf gpurun_out/prof gpurun_out/prof_c4 gpurun_out/prof_cfg
bash tools/profile_gpu.sh > gpurun_out/profile_gpu.log 2>&1
mkdir -p gpurun_out/prof_c4 gpurun_out/prof_cfg
# plain runs (the numbers DESIGN.md quotes)
python bench.py > gpurun_out/bench_latest.json 2> gpurun_out/bench_latest.err
M3D_BENCH_FORCE_SHARDED=1 python bench.py --no-cpu-baseline > gpurun_out/bench_sharded_world1.json 2> gpurun_out/bench_sharded_world1.err
python tools/bench_configs.py > gpurun_out/bench_configs.jsonl 2> gpurun_out/bench_configs.err
# kernel stats of all configurations, and of C4 alone
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg -o cfg -- python tools/bench_configs.py C2 C3 C5 --no-cpu-baseline > gpurun_out/prof_cfg/cfg.out 2> gpurun_out/prof_cfg/cfg.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c4 -o c4 -- python tools/bench_configs.py C4 --no-cpu-baseline > gpurun_out/prof_c4/c4.out 2> gpurun_out/prof_c4/c4.err
# counters of the registration validation kernel
bash tools/pmc_reg_validate.sh > gpurun_out/pmc_reg_validate.txt 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_reg_fetch -o f -- python gpurun_out/pmc_reg_0/run.py > /dev/null 2>&1
python - <<'PY' >> gpurun_out/pmc_reg_validate.txt
import csv, glob
tot = n = 0
for f in glob.glob("gpurun_out/pmc_reg_fetch/**/f_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "reg_validate_k" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            tot += float(r["Counter_Value"]); n += 1
print(f"reg_validate_k FETCH_SIZE over {n} launches (20 000 iterations of C4): {tot:.0f} KiB -> x1024 x2 (gfx950) = {tot*2048/1e9:.2f} GB")
PY
M3D_C4_ENV=1 python gpurun_out/pmc_reg_0/run.py >> gpurun_out/pmc_reg_validate.txt 2>&1
echo "---- memory pipeline (tools/pmc_reg_mem.sh)" >> gpurun_out/pmc_reg_validate.txt
bash tools/pmc_reg_mem.sh >> gpurun_out/pmc_reg_validate.txt 2>&1
python tools/step_timeline.py gpurun_out/prof/trace_bench/bench_kernel_trace.csv minimal_fit_k 15 > gpurun_out/step_timeline.txt 2>&1; python tools/step_timeline.py gpurun_out/prof/trace_bench/bench_kernel_trace.csv >> gpurun_out/step_timeline.txt 2>&1
ls gpurun_out/prof gpurun_out/prof_c4 gpurun_out/prof_cfg
# matcher: kernel stats + the MFMA-busy share of the one-pass scan
bash tools/prof_match.sh prof_match > gpurun_out/prof_match.txt 2>&1
bash tools/pmc_match.sh > gpurun_out/pmc_match.txt 2>&1
# C5: per-round timeline + the library's own breakdown
bash tools/c5_round_timeline.sh c5t 200 > /dev/null 2>&1
python tools/time_c5_plain.py > gpurun_out/c5_plain.txt 2>&1
python tools/time_oneshot.py > gpurun_out/oneshot.txt 2>&1
# SQ counters of the sphere's / cylinder's scoring launches (the C3 fractions)
M3D_PMC_CMD="python tools/bench_configs.py C3 --no-cpu-baseline" bash tools/pmc_score_bench.sh > gpurun_out/pmc_score_c3.txt 2>&1
python tools/model_strong_scaling.py > gpurun_out/strong_scaling_model.jsonl 2> gpurun_out/strong_scaling_model.err
# round 5: pairs in flight (lanes), C3's kernel timeline, the driver-style run (what BENCH_r05 will ask for)
python tools/bench_configs.py N2 --no-cpu-baseline > gpurun_out/n2_pairs_in_flight.jsonl 2> /dev/null
bash tools/c3_timeline.sh > gpurun_out/c3_timeline.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver_style.json 2> /dev/null
# round 6: the validation's candidate cache on / off, the LANES batch mode is part of bench_configs; boundary_k's counters
python tools/time_c4_forced.py > gpurun_out/c4_forced_cache_ab.txt 2>&1
bash tools/pmc_boundary.sh > gpurun_out/pmc_boundary.txt 2>&1
# the matcher's one-launch scan: counters per wave-tile-pair; the default-confidence C4 call
bash tools/pmc_match_scan.sh > gpurun_out/pmc_match_scan.txt 2>&1
python tools/time_c4_default.py > gpurun_out/c4_default.txt 2>&1
