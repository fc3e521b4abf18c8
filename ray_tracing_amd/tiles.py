"""Row-tile partition of one frame over the ranks of a node, and the gather back to rank 0.

The frame is cut into tiles of TILE_ROWS rows; rank r of `world` renders tiles r, r+world, ... into a
compact buffer (tile k of the rank at local rows k*TILE_ROWS...).  Interleaving instead of contiguous bands
keeps the per-rank work even: rows of sky cost one ray segment per sample, rows of spheres several.
Pixels are independent and the RNG is keyed on the global pixel index, so the assembled frame is
bit-identical for every world size.  The only exchange is one gather of the finished tiles (RCCL over
xGMI on GPUs, gloo in the CPU tests); the reference itself is single-device (ray_tracing.rs:107).
"""
import torch
import torch.distributed as dist

TILE_ROWS = 8


def n_tiles(height, tile_rows=TILE_ROWS):
    return (height + tile_rows - 1) // tile_rows


def tiles_of_rank(height, rank, world, tile_rows=TILE_ROWS):
    """Global tile indices rendered by `rank`."""
    return list(range(rank, n_tiles(height, tile_rows), world))


def max_tiles_per_rank(height, world, tile_rows=TILE_ROWS):
    return (n_tiles(height, tile_rows) + world - 1) // world


def local_buffer(height, width, world, device, channels=4, dtype=torch.uint8, tile_rows=TILE_ROWS):
    """Per-rank compact buffer, padded to the largest per-rank tile count so the gather is uniform."""
    return torch.zeros((max_tiles_per_rank(height, world, tile_rows) * tile_rows, width, channels), dtype=dtype, device=device)


def assemble(gathered, height, tile_rows=TILE_ROWS):
    """De-interleave the per-rank buffers (list indexed by rank) into the frame [height, W, C].
    Tile t = k*world + r sits at gathered[r][k]: stack as [k][r] and flatten."""
    world = len(gathered)
    rows, W, C = gathered[0].shape
    k = rows // tile_rows
    g = torch.stack(list(gathered), 0).view(world, k, tile_rows, W, C)
    return g.permute(1, 0, 2, 3, 4).reshape(k * world * tile_rows, W, C)[:height]


def gathered_buffer(local, world):
    """Rank 0's destination of the gather: every rank's compact buffer back to back, [world, rows, W, C] --
    the layout rt_assemble_tiles (the library's copy kernel) reads."""
    return torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)


def gather_contiguous(local, gathered, rank, world):
    """One gather of every rank's compact buffer into rank 0's contiguous `gathered` (from gathered_buffer;
    None on the other ranks).  The collective's receive list is the buffer's own slices, so no copy follows."""
    if world == 1:
        if gathered is not None:
            gathered[0].copy_(local)
        return
    dist.gather(local, [gathered[r] for r in range(world)] if rank == 0 else None, dst=0)


def gather_frame(local, height, rank, world, scratch=None, tile_rows=TILE_ROWS):
    """Gathers every rank's compact buffer to rank 0 and returns the assembled frame there (None elsewhere).
    `scratch` may hold a reusable list of `world` receive buffers on rank 0."""
    if world == 1:
        return local[:height]
    if rank == 0 and scratch is None:
        scratch = [torch.empty_like(local) for _ in range(world)]
    dist.gather(local, scratch if rank == 0 else None, dst=0)
    if local.is_cuda:
        # the collective is only enqueued; `local` is rewritten by the next render on another stream
        torch.cuda.current_stream(local.device).synchronize()
    if rank != 0:
        return None
    return assemble(scratch, height, tile_rows)
