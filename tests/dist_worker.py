"""Worker of tests/test_distributed_cpu.py: one rank of a world_size-N gloo job on CPU.  Each rank produces
its row tile of a small frame (the CPU oracle stands in for the GPU renderer: same tile contract, rows
[row0, row0+nrows), global ray ids), the tiles are exchanged with the product's tiling.gather_tiles, and
rank 0 checks the assembled frame against the golden fixture."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402
import a10_pass as A  # noqa: E402
from conftest import load_fixture  # noqa: E402


def main():
    case = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    graft.load_package()
    from raytracing_amd.pyhost import tiling
    fx, sc = load_fixture(case)
    tiles = tiling.row_tiles(sc.height, world)
    row0, nrows = tiles[rank]
    # the stand-in renderer: full-frame oracle pass, then this rank's rows (rays are independent; ids global)
    st = A.PassState(sc, A.make_seeds(sc.total_rays))
    A.run_pass(A.load_oracle(), sc, st)
    pr = tiling.padded_rows(sc.height, world)
    pix = torch.zeros(pr * sc.width * 4, dtype=torch.uint8)
    rad = torch.zeros(pr * sc.width * 4, dtype=torch.float32)
    sl = slice(row0 * sc.width, (row0 + nrows) * sc.width)
    pix[: nrows * sc.width * 4] = torch.from_numpy(st.pixel[sl].reshape(-1).copy())
    rad[: nrows * sc.width * 4] = torch.from_numpy(A.radiance_sums(st.acu, sc.rpp)[sl].reshape(-1).copy())
    frame = tiling.assemble(tiling.gather_tiles(pix), sc.height, sc.width, world)
    radiance = tiling.assemble(tiling.gather_tiles(rad), sc.height, sc.width, world)
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = (np.array_equal(frame.numpy().reshape(-1, 4), fx["pixel"])
          and np.array_equal(radiance.numpy().reshape(-1, 4).view(np.uint32), fx["radiance"].view(np.uint32))
          and t.item() == world and sum(n for _, n in tiles) == sc.height)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"ok": bool(ok), "world": world, "tiles": tiles}))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
