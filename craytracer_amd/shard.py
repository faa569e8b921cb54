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
