"""The randomised soak (tools/stress_round3.py's generators against the oracle) for a few minutes under several configurations of the
library, one line per configuration: seed, m3d_config overrides, cases compared.  Run on the GPU box:
    python tools/soak_long.py [seconds per configuration, default 240] > gpurun_out/soak_long.txt"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from misc3d_amd import capi  # noqa: E402

spec = importlib.util.spec_from_file_location("stress_round3", os.path.join(ROOT, "tools", "stress_round3.py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
# (round 6: + the validation's candidate cache forced on from the first incumbent, with most of the time on registrations)
for seed, cfg in ((201, {}), (202, {"plane_bound": 2}), (203, {"plane_bound": 2, "lanes": 1}),
                  (204, {"plane_bound": 2, "score_fp32_screen": 0, "cull_fp32": 0}), (205, {"reg_cache": 2}), (206, {"reg_cache": 2, "reg_prune": 0}),
                  (207, {"cull_fp32": 2, "match_pipeline": 2})):   # (round 6, later: box tests with a lane per tile at every size, every match sliced)
    reg_share = 0.7 if "reg_cache" in cfg else 0.2
    old = capi.set_config(**cfg)
    try:
        n = mod.run(budget=budget * (1.0 - reg_share), reg_budget=budget * reg_share, seed=seed, log=lambda s: None)
    finally:
        capi.restore_config(old)
    print(seed, cfg, n, flush=True)
