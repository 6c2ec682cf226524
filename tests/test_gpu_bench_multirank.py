"""`python bench.py --gpus 2` end to end on a one-GPU box: bench.py starts its own ranks (torch.distributed.run), the
control plane comes up, every rank runs the sharded C++ driver, the strong-scaling block runs, rank 0 prints ONE JSON
line.  M3D_BENCH_REHEARSAL=1 puts every rank on device 0 with the records over gloo (RCCL refuses two ranks on one
device); everything else is the code path the 8-GPU run takes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_and_prints_one_json_line():
    env = dict(os.environ, M3D_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2",
                        "--points", "200000"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    # the headline is ONE workload at every N: BASELINE's C2 with the per-GPU work fixed (the driver derives the efficiency);
    # the fixed-total question is the strong_scaling block, with the one-GPU model's prediction beside every measurement
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["hypotheses_total"] == 20000
    assert "rehearsal" in d["config"]["parallelism"] and d["collectives_per_step"] >= 1
    assert d["value"] > 0 and 0 < d["roofline"]["frac"] <= 1
    st = d["strong_scaling"]
    assert set(st) == {"c2", "c3cyl", "c3sph"} and all(v["identical_to_1gpu"] for v in st.values())
    assert all("model_predicted_speedup" in v for v in st.values()) and st["c3cyl"]["model_predicted_speedup"]["exchange_30us"] > 1
    tr = d["transport"]
    assert len(tr["per_rank"]) == 2 and tr["valid_headline"] and "c5_note" in d
    # N independent fits in flight (no collective): what the GPUs deliver when the jobs do not share a hypothesis stream
    assert d["replicas"]["collectives"] == 0 and d["replicas"]["value"] > 0


@pytest.mark.gpu
def test_bench_falls_back_to_the_host_transport_when_rccl_does_not_come_up():
    """If the RCCL communicator cannot be created on some rank (simulated), every rank switches to the host transport (same
    shard loop and kernels, records over gloo) and the JSON line says so -- a slower number, not a crash."""
    env = dict(os.environ, M3D_BENCH_REHEARSAL="1", M3D_BENCH_FAKE_RCCL_FAILURE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1",
                        "--points", "100000"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert "RCCL communicator unavailable" in d["config"]["parallelism"] or "rehearsal" in d["config"]["parallelism"]
    assert d["transport"]["valid_headline"] is False and "RCCL communicator unavailable" in d["transport"]["per_rank"][0]["transport"]
    # ... and outside a rehearsal that is a FAILED run: the line is printed for diagnosis, the exit code is 3
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--points", "100000", "--no-strong-extra"], env=dict(env, M3D_BENCH_STRICT="1"), capture_output=True,
                       text=True, timeout=900, cwd=ROOT)
    assert p.returncode != 0 and any(ln.startswith("{") for ln in p.stdout.splitlines())
