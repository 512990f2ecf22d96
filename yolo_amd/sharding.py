"""Batch data-parallel sharding across the GPUs of one node.  The reference is
single-GPU (config.py:39, inference.py:342-343); images are independent, so rank r simply
owns the contiguous slice [r*B, (r+1)*B) of the global batch and no collective sits on the
data path.  The only exchange is one all_gather of per-image detection counts for the node
report (RCCL over xGMI when the backend is nccl; gloo in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_range(global_batch, rank, world_size):
    """Contiguous image range of `rank`; the remainder goes to the lowest ranks."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, rem = divmod(global_batch, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gather_counts(local_counts, max_per_rank=None):
    """all_gather of int32 per-image detection counts -> 1-D tensor over the global batch in
    rank order.  Ranks may hold different numbers of images (padded to max_per_rank)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_counts.clone()
    world = dist.get_world_size()
    n = torch.tensor([local_counts.numel()], dtype=torch.int64, device=local_counts.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    cap = int(max(int(s.item()) for s in sizes)) if max_per_rank is None else max_per_rank
    padded = torch.zeros(cap, dtype=torch.int32, device=local_counts.device)
    padded[:local_counts.numel()] = local_counts.to(torch.int32)
    out = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(out, padded)
    return torch.cat([o[:int(s.item())] for o, s in zip(out, sizes)])
