"""N ranks, real kernels, one GPU: the N > 1 path with every rank rendering its row tile through the HIP pass (tests/dist_worker_gpu.py).
The ranks share the box's single device and exchange their tiles over gloo; the RCCL transport itself is rehearsed with one rank in
tests/test_abi_hardening.py and by `torch.distributed.run --nproc-per-node 1 bench.py`."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT
from test_distributed_cpu import free_port

pytestmark = pytest.mark.gpu


# at most 3 ranks: with the test process itself that is 4 processes on the GPU (the boxes allow 6)
@pytest.mark.parametrize("world,case", [(2, "cornell_32x24_r4"), (3, "cornell_teapot3_32x24_r4"), (3, "own_gems_48x36_r4")])
def test_ranks_render_their_tiles_on_the_device(world, case):
    """the teapot scene has the grid kernels and two lights; uneven tile heights are covered on one process by
    test_gpu_parity.py::test_row_tiles_compose_to_the_full_frame"""
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker_gpu.py"), case]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["ok"] and out["world"] == world
