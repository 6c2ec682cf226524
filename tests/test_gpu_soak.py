"""The randomised soak against the oracle, INSIDE the suite (round 3 ran it from the command line and
left no artefact).  tools/stress_round3.py's generators with a fixed master seed for a fixed budget: random sizes (tiny and
ragged included), seeds, iteration limits, scene scales of 10^-2 .. 10^2 and offsets of up to 10^3 scene sizes through the
mutual matcher, the iterative segmentation (tombstones, deferred RefineModel, lead-less rounds, ties), one-shot fits of all
three models on the dense and the sorted path, and the correspondence RANSAC (T bit for bit, iterations, validations, est_k).
The log of the run lands in gpurun_out/soak_suite.log (copied to profiles/ per round)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_soak_fixed_seed_fixed_budget(capi, orc):
    spec = importlib.util.spec_from_file_location("stress_round3", os.path.join(ROOT, "tools", "stress_round3.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lines = []
    n = mod.run(budget=45.0, reg_budget=15.0, seed=20260929, log=lines.append)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "soak_suite.log"), "w") as f:
        f.write("\n".join(lines) + "\n")
    assert n["match"] >= 5 and n["segment"] >= 5 and n["fit"] >= 10 and n["registration"] >= 3, n
