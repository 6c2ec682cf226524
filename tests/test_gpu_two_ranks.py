"""The N > 1 drivers with the PRODUCT scorer: world size 2 and 3, gloo records, all ranks sharing GPU 0
(tests/two_rank_gpu_worker.py).  Complements tests/test_distributed_gloo.py (CPU, oracle scorer)."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
@pytest.mark.parametrize("world,phases", [(2, None), (3, None), (2, "3")])
def test_sharded_drivers_on_one_gpu(world, phases):
    """phases = "3": M3D_SCORE_PHASES=3 in the workers -- every window of every rank scored in three phases with re-pruning in
    between (the explicit setting engages them on the workers' small clouds too); the results must still equal the one-GPU ones."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    if phases:
        env["M3D_SCORE_PHASES"] = phases
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(HERE, "two_rank_gpu_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    out = p.stdout + p.stderr
    assert p.returncode == 0 and f"TWO_RANK_OK world {world}" in out, out[-3000:]
