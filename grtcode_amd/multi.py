"""Column sharding across the GPUs of one node: one process per GPU, contiguous column blocks,
replicated read-only spectroscopy, and a single gather of the per-column integrated fluxes
(12 doubles per column) to rank 0 -- RCCL over xGMI on GPUs (`nccl` backend), gloo on CPU in tests.
The reference has no communication layer at all (it fans out processes with -x/-X column ranges and
merges netCDF files afterwards: GRTworkflow/run-rfmip-irf.sh:103-148); this is its in-node equivalent."""
import torch
import torch.distributed as dist


def shard(num_columns, rank, world_size):
    """Contiguous block [first, first+count) of `num_columns` for `rank` (ceil split, last ranks may be short)."""
    per = -(-num_columns // world_size)
    first = min(rank * per, num_columns)
    return first, max(0, min(per, num_columns - first))


def gather_fluxes(local, rank, world_size, out=None):
    """Gather equally sized per-rank [cols, 12] blocks to rank 0 (returns the list there, else None)."""
    if world_size == 1:
        return [local]
    if rank == 0 and out is None:
        out = [torch.empty_like(local) for _ in range(world_size)]
    dist.gather(local, out if rank == 0 else None, dst=0)
    return out if rank == 0 else None


def max_over_ranks(seconds, device):
    if not (dist.is_available() and dist.is_initialized()):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
