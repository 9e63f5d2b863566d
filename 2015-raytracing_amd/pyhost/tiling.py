"""Row-tile sharding of a frame over ranks and the one exchange step of the path.

Every ray is independent, so a frame shards by contiguous row tiles with no data-path collective while
rendering; ray ids stay GLOBAL (id = (row*W + col)*rpp + i), so seeds -- and therefore every pixel -- do
not depend on the number of ranks.  The only exchange is the final assembly of the RGBA8 (or fp32
radiance) tiles: one all_gather over RCCL ("nccl" backend) on GPUs, gloo in the CPU tests.  Tiles are
padded to the tallest tile so the collective is regular; `assemble` drops the padding.
"""
import torch
import torch.distributed as dist


def row_tiles(height, world):
    """[(row0, nrows)] per rank: contiguous, sizes differ by at most one row, taller tiles first."""
    base, extra = divmod(height, world)
    out, r0 = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((r0, n))
        r0 += n
    return out


def padded_rows(height, world):
    return max(n for _, n in row_tiles(height, world))


def gather_tiles(tile, out=None):
    """all_gather of equally sized 1-D tiles (one per rank) into `out` (world * tile.numel())."""
    if not dist.is_initialized():
        return tile
    world = dist.get_world_size()
    if out is None:
        out = torch.empty(world * tile.numel(), dtype=tile.dtype, device=tile.device)
    dist.all_gather_into_tensor(out, tile)
    return out


def assemble(gathered, height, width, world, channels=4):
    """Drop the padding rows: [world * padded_rows * width * channels] -> [height, width, channels]."""
    pr = padded_rows(height, world)
    g = gathered.view(world, pr, width, channels)
    return torch.cat([g[r, :n] for r, (_, n) in enumerate(row_tiles(height, world))], dim=0)
