"""Multi-GPU sharding of the traversal path.

Every ray is independent given a read-only scene (the reference relies on the same fact: worker
threads share `Scene` and write disjoint pixel ranges, /root/reference/main.cpp:94-141), so the
path shards with NO data-path collective: one process per GPU, a full scene replica per GPU, rank r
traces its own contiguous slice of the ray batch.  The only exchange steps are (a) the timing
reduction of a benchmark and (b), for a render, the sum of the per-GPU radiance accumulation buffers
(the operand of the reference's saveImageState, imagestate.h:5-19) -- an all-reduce over RCCL/xGMI
on GPUs, gloo on CPU in the tests.
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous slice [begin, end) of n items for `rank`; sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(int(n), world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def max_over_ranks(value, device="cpu"):
    """max of a python float over all ranks (1 rank: identity)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def combine_radiance(buf, spp):
    """Sum per-GPU float32 radiance sums [H, W, 3] and sample counts in place over all ranks
    (sample-pass sharding: each rank rendered its own passes over the full frame)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        c = torch.tensor([int(spp)], dtype=torch.int64, device=buf.device)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        spp = int(c.item())
    return buf, spp


def render_sharded(renderer, spp_per_rank, rank, world, device=None):
    """Sample-range sharding of a render (the reference's unit of work is the sample pass, main.cpp:295): rank k of
    `world` renders samples [k * S, (k + 1) * S) of the whole frame, then the colour buffers are summed over all ranks
    (RCCL when `device` is a GPU and the process group is nccl; gloo sums the host copy).  Returns the combined
    colour buffer as a torch tensor [h * w * 3] and the total sample count."""
    renderer.reset(rank * spp_per_rank)
    renderer.render(spp_per_rank)
    n = renderer.width * renderer.height * 3
    if device is not None and torch.device(device).type == "cuda" and (not dist.is_initialized() or dist.get_backend() == "nccl"):
        buf = torch.zeros(n, dtype=torch.float32, device=device)
        renderer.read_dev(buf.data_ptr(), torch.cuda.current_stream(device).cuda_stream)
        torch.cuda.synchronize(device)
    else:
        host, _ = renderer.read()
        buf = torch.from_numpy(host.reshape(-1).copy())
    buf, total = combine_radiance(buf, spp_per_rank)
    return buf, total


def tile_rows(height, rank, world):
    """Tile sharding: the image rows [begin, end) rank `rank` renders -- a contiguous band; bands of
    all ranks cover the frame once.  Pass as RenderDesc(rows=...)."""
    return shard_range(height, rank, world)


def gather_tiles(band, height, width, rank, world):
    """All-gather of the per-rank bands into the whole frame (every rank gets it): `band` is this rank's colour buffer,
    a float32 tensor of tile_rows(height, rank, world) rows x width x 3 (on the GPU with the nccl backend = RCCL all-gather over
    xGMI, 12.4 MB per GPU at 4K on 8 GPUs; on the host with gloo).  Bands differ in size by at most one row, so they are padded
    to the largest for the collective and cut back after it.  Returns [height, width, 3]."""
    b, e = tile_rows(height, rank, world)
    band = band.reshape(e - b, width, 3)
    if not (dist.is_available() and dist.is_initialized()) or world == 1:
        return band.clone()
    rows_max = -(-height // world)
    mine = torch.zeros((rows_max, width, 3), dtype=band.dtype, device=band.device)
    mine[: e - b] = band
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    out = []
    for r in range(world):
        rb, re = tile_rows(height, r, world)
        out.append(parts[r][: re - rb])
    return torch.cat(out, 0)


def render_tiled(renderer, spp, height, width, rank, world, device=None):
    """Tile sharding of a render: `renderer` was created with rows=tile_rows(height, rank, world); every rank renders `spp`
    samples of its band, then the bands are all-gathered.  Because pixel indices (camera rays, random numbers) are those of the
    whole frame, the result equals a single renderer's whole frame bit for bit."""
    renderer.reset(0)
    renderer.render(spp)
    if device is not None and torch.device(device).type == "cuda" and (not dist.is_initialized() or dist.get_backend() == "nccl"):
        band = torch.zeros(renderer.width * renderer.height * 3, dtype=torch.float32, device=device)
        renderer.read_dev(band.data_ptr(), torch.cuda.current_stream(device).cuda_stream)
        torch.cuda.synchronize(device)
    else:
        host, _ = renderer.read()
        band = torch.from_numpy(host.reshape(-1).copy())
    return gather_tiles(band, height, width, rank, world)
