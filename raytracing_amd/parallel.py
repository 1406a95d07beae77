"""Multi-GPU partitioning of one frame: one process per GPU, torch.distributed over RCCL/xGMI.

The path shards by SAMPLES (spp split): rank r renders global sample indices [r*S/N, (r+1)*S/N) of every pixel into its
own accumulation buffer; RNG streams are keyed by (seed, pixel, global sample index), so the union over ranks is the
same set of samples a single GPU would draw.  The only exchange is one SUM reduce of the width*height*3 accumulation
buffer onto rank 0 (99.5 MB fp32 at 3840x2160), after which rank 0 resolves the frame.  A row-tile split
(`row_range`) is the alternative for low-spp frames: disjoint rows, no reduce, a gather of slabs instead.

The reference has no counterpart (single process, single device).
"""


def sample_range(rank, world, spp):
    """Global sample indices [k0, k1) of `spp` samples owned by `rank` of `world`; ranges tile [0, spp) exactly."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return rank * spp // world, (rank + 1) * spp // world


def row_range(rank, world, height):
    """Rows [y0, y1) owned by `rank` for the tile split."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return rank * height // world, (rank + 1) * height // world


def reduce_frame(accum, world, dst=0, force=False):
    """SUM-reduce the per-rank accumulation buffers onto `dst` (RCCL on GPUs, gloo in the CPU tests). In place.
    `force`: issue the collective even in a group of one (rehearses the RCCL call on a one-GPU box)."""
    if world > 1 or force:
        import torch.distributed as dist
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum
