"""Turn the rocprofv3 outputs under gpurun_out/prof/ into the small tracked summaries under
profiles/: kernel-trace stats (CSV, as written by rocprofv3 --stats) and a PMC summary (JSON).

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE
are collected in SEPARATE passes, are in KiB, and on gfx950 FETCH_SIZE reports 1/2 of the bytes of a
coalesced streaming read -> multiplied by 2.  Both corrections are CALIBRATED here on aos_to_soa_k,
whose traffic is known exactly (reads N x 24 B, writes n_pad x 24 B)."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01"


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


def counters(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(
                (float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size"])))
    return agg


def main():
    os.makedirs(DST, exist_ok=True)
    for sub, name in (("trace_bench", "bench"), ("trace_target", "target")):
        p = os.path.join(SRC, sub, f"{name}_kernel_stats.csv")
        if os.path.exists(p):
            shutil.copy(p, os.path.join(DST, f"{TAG}_{name}_kernel_stats.csv"))
    p = os.path.join(SRC, "bench_under_rocprof.json")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(DST, f"{TAG}_bench_under_rocprof.json"))
    for src_rel, dst_name in (("../prof_c4/c4_kernel_stats.csv", f"{TAG}_c4_kernel_stats.csv"),
                              ("../prof_cfg/cfg_kernel_stats.csv", f"{TAG}_configs_kernel_stats.csv"),
                              ("../bench_configs.jsonl", f"{TAG}_bench_configs.jsonl"),
                              ("../bench_latest.json", f"{TAG}_bench.json"),
                              ("../bench_sharded_world1.json", f"{TAG}_bench_sharded_world1.json"),
                              ("../pmc_reg_validate.txt", f"{TAG}_pmc_reg_validate.txt"),
                              ("../step_timeline.txt", f"{TAG}_c2_step_timeline.txt"),
                              ("../pmc_score_c3.txt", f"{TAG}_pmc_score_c3.txt"),
                              ("../prof_match.txt", f"{TAG}_match_kernel_stats.txt"),
                              ("../pmc_match.txt", f"{TAG}_pmc_match.txt"),
                              ("../c5t_timeline.txt", f"{TAG}_c5_round_timeline.txt"),
                              ("../c5_plain.txt", f"{TAG}_c5_plain.txt"),
                              ("../oneshot.txt", f"{TAG}_oneshot.txt"),
                              ("../c4_full_size_vs_oracle.txt", f"{TAG}_c4_full_size_vs_oracle.txt"),
                              ("../strong_scaling_model.jsonl", f"{TAG}_strong_scaling_model.jsonl"),
                              ("../n2_pairs_in_flight.jsonl", f"{TAG}_n2_pairs_in_flight.jsonl"),
                              ("../c3_timeline.txt", f"{TAG}_c3_timeline.txt"),
                              ("../bench_driver_style.json", f"{TAG}_bench_driver_style.json"),
                              ("../c4_forced_cache_ab.txt", f"{TAG}_c4_forced_cache_ab.txt"),
                              ("../pmc_boundary.txt", f"{TAG}_pmc_boundary.txt"),
                              ("../pmc_match_scan.txt", f"{TAG}_pmc_match_scan.txt"),
                              ("../c4_default.txt", f"{TAG}_c4_default_times.txt")):
        q = os.path.join(SRC, src_rel)
        if os.path.exists(q):
            shutil.copy(q, os.path.join(DST, dst_name))
    p = os.path.join(SRC, "target.out")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(DST, f"{TAG}_pmc_target_stdout.txt"))
    fetch = counters(os.path.join(SRC, "pmc_fetch", "fetch_counter_collection.csv"))
    write = counters(os.path.join(SRC, "pmc_write", "write_counter_collection.csv"))
    summary = {"units": "FETCH_SIZE/WRITE_SIZE in KiB as reported; hbm_read_bytes = FETCH_SIZE*1024*2 (gfx950 "
                        "half-count correction), hbm_write_bytes = WRITE_SIZE*1024", "kernels": {}}
    n_points = int(os.environ.get("M3D_PMC_POINTS", "1000000"))
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("m3d::"):
            continue
        # the largest dispatch of each kernel (the target script launches the full-size one last/most)
        fv = fetch.get(k, {}).get("FETCH_SIZE", [])
        wv = write.get(k, {}).get("WRITE_SIZE", [])
        fmax = max(fv, key=lambda t: t[0]) if fv else (0.0, 0, 0)
        wmax = max(wv, key=lambda t: t[0]) if wv else (0.0, 0, 0)
        rd = fmax[0] * 1024 * 2
        wr = wmax[0] * 1024
        summary["kernels"][k] = {"launches_seen": len(fv), "FETCH_SIZE_KiB_max": fmax[0],
                                 "WRITE_SIZE_KiB_max": wmax[0], "hbm_read_bytes": rd, "hbm_write_bytes": wr,
                                 "hbm_bytes": rd + wr, "duration_ns_of_that_launch": fmax[1], "grid": fmax[2]}
    cal = summary["kernels"].get("m3d::aos_to_soa_k")
    if cal:
        n_pad = -(-n_points // 2048) * 2048
        summary["calibration"] = {"kernel": "m3d::aos_to_soa_k", "known_read_bytes": n_points * 24,
                                  "known_write_bytes": n_pad * 24,
                                  "measured_read_over_known": cal["hbm_read_bytes"] / (n_points * 24),
                                  "measured_write_over_known": cal["hbm_write_bytes"] / (n_pad * 24)}
    for extra in ("pmc_sq/sq_counter_collection.csv", "pmc_grbm/grbm_counter_collection.csv"):
        p = os.path.join(SRC, extra)
        if not os.path.exists(p):
            continue
        for k, cs in counters(p).items():
            if not (k.startswith("m3d::score_k") or k.startswith("m3d::score_mask_k") or k.startswith("m3d::score_screen_k") or k.startswith("m3d::cull_")):
                continue
            d = summary["kernels"].setdefault(k, {})
            for cname, vals in cs.items():
                big = max(vals, key=lambda t: t[0])
                d[cname] = big[0]
                d.setdefault("pmc_launch_ns", {})[cname] = big[1]
    with open(os.path.join(DST, f"{TAG}_pmc_summary.json"), "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    # traffic of the dominant kernel PER LAUNCH, averaged over the launches of bench.py itself (the same
    # launches bench.py times with HIP events and rocprofv3 --stats averages)
    bf = os.path.join(SRC, "pmc_bench_fetch", "fetch_counter_collection.csv")
    bw = os.path.join(SRC, "pmc_bench_write", "write_counter_collection.csv")
    if os.path.exists(bf) and os.path.exists(bw):
        fv = counters(bf).get("m3d::score_screen_k<0>", {}).get("FETCH_SIZE", [])
        wv = counters(bw).get("m3d::score_screen_k<0>", {}).get("WRITE_SIZE", [])
        if fv and wv:
            rd = sum(t[0] for t in fv) / len(fv) * 1024 * 2
            wr = sum(t[0] for t in wv) / len(wv) * 1024
            latest = {"kernel": "m3d::score_screen_k<0>", "hbm_bytes_per_launch": rd + wr, "hbm_read_bytes": rd,
                      "hbm_write_bytes": wr, "launches_averaged": len(fv),
                      "avg_launch_ns_under_pmc": sum(t[1] for t in fv) / len(fv),
                      "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python bench.py --steps 10 "
                                 "--warmup 2 --no-cpu-baseline",
                      "corrections": "KiB -> B; FETCH_SIZE x2 on gfx950 (calibrated on aos_to_soa_k, see "
                                     f"{TAG}_pmc_summary.json)",
                      "measured_at_commit": os.environ.get("M3D_PROFILE_COMMIT", f"round {TAG} (set M3D_PROFILE_COMMIT to the hash)")}
            summary["bench_score_mask_k"] = latest
            with open(os.path.join(DST, "pmc_score_latest.json"), "w") as f:
                json.dump(latest, f, indent=1)
            with open(os.path.join(DST, f"{TAG}_pmc_summary.json"), "w") as f:
                json.dump(summary, f, indent=1, sort_keys=True)
    q = os.path.join(SRC, "bench_kernel_detail.json")
    if os.path.exists(q):
        shutil.copy(q, os.path.join(DST, f"{TAG}_bench_kernel_detail.json"))
    print(json.dumps(summary.get("calibration"), indent=1))
    print({k: (v.get("hbm_bytes"), v.get("launches_seen")) for k, v in summary["kernels"].items()})


if __name__ == "__main__":
    main()
