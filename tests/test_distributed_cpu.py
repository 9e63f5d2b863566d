"""CPU, gloo, world_size 2 and 3: the N>1 path -- row tiling, padded all_gather of the tiles, assembly,
max-over-ranks reduction -- exactly the code bench.py runs over RCCL on the GPUs."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_row_tiles_partition(pkg):
    from raytracing_amd.pyhost import tiling
    for h in (1, 7, 24, 1080, 2160):
        for w in (1, 2, 3, 4, 8):
            t = tiling.row_tiles(h, w)
            assert len(t) == w and t[0][0] == 0 and sum(n for _, n in t) == h
            assert all(t[i][0] + t[i][1] == t[i + 1][0] for i in range(w - 1))
            assert max(n for _, n in t) - min(n for _, n in t) <= 1
    assert tiling.row_tiles(1080, 8) == [(135 * i, 135) for i in range(8)]


@pytest.mark.parametrize("world,case", [(2, "cornell_32x24_r4"), (3, "own_gems_64x48_r1"), (2, "threeLights_32x24_r1")])
def test_gloo_tiles_assemble_to_the_golden_frame(world, case):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), case]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["ok"] and out["world"] == world
