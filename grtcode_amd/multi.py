"""Column sharding across the GPUs of one node: one process per GPU, contiguous column blocks,
replicated read-only spectroscopy, and a single gather of the per-column integrated fluxes
(12 doubles per column) to rank 0.

Two carriers of that gather:
  * torch.distributed (`nccl` backend = RCCL over xGMI on GPUs, gloo on CPU in tests) -- what bench.py uses, because
    the driver launches it with torch.distributed.run;
  * the library's own C entry points grt_multi_* (include/grt_ext.h: ncclGather through librccl, or per-rank files in
    a rendezvous directory), wrapped by `Multi` below -- what a C caller such as examples/rfmip_batch_driver.c uses.
The reference has no communication layer at all (it fans out processes with -x/-X column ranges and merges netCDF
files afterwards: GRTworkflow/run-rfmip-irf.sh:103-148); this is its in-node equivalent."""
import ctypes as C

import torch
import torch.distributed as dist


def shard(num_columns, rank, world_size):
    """Contiguous block [first, first+count) of `num_columns` for `rank`: ceil-sized blocks, the last ranks may be
    short or empty (100 columns over 8 ranks: 13 x 7 + 9).  Same rule as grt_multi_shard."""
    per = -(-num_columns // world_size)
    first = min(rank * per, num_columns)
    return first, max(0, min(per, num_columns - first))


def gather_fluxes(local, rank, world_size, out=None, num_columns=None):
    """Gather the ranks' [count, 12] blocks of a `num_columns`-column set sharded by `shard` to rank 0.

    dist.gather needs equally sized tensors on every rank, so a short (or empty) block is padded to
    per = ceil(num_columns/world) rows; on rank 0 the padded blocks laid end to end ARE the global array (row
    r*per + i is column r*per + i), trimmed to num_columns rows.  Returns that [num_columns, 12] tensor on rank 0,
    None elsewhere.  `out`: optional list of world_size [per, 12] receive tensors (rank 0) reused across steps."""
    if num_columns is None:
        num_columns = world_size * local.shape[0]          # equal blocks (weak scaling)
    per = -(-num_columns // world_size)
    if world_size == 1:
        return local[:num_columns]
    send = local
    if local.shape[0] != per:
        send = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    if rank == 0 and out is None:
        out = [torch.empty_like(send) for _ in range(world_size)]
    dist.gather(send, out if rank == 0 else None, dst=0)
    return torch.cat(out)[:num_columns] if rank == 0 else None


def max_over_ranks(seconds, device):
    if not (dist.is_available() and dist.is_initialized()):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


RCCL, FILES = 0, 1


class Multi:
    """ctypes mirror of grt_multi_* (C, no torch): what a C driver calls.  `transport` RCCL (device pointers,
    asynchronous on the library stream) or FILES (per-rank files in `rendezvous_dir`, host or device pointers)."""

    def __init__(self, transport, device, rank, world, rendezvous_dir):
        from . import api
        self.api, self.lib = api, api.load_library()
        self.rank, self.world, self.device = rank, world, device
        self.m = C.c_void_p()
        api.check(self.lib.grt_multi_create(C.byref(self.m), transport, device, rank, world, str(rendezvous_dir).encode()))

    def shard(self, num_columns):
        first, count = C.c_int(), C.c_int()
        self.api.check(self.lib.grt_multi_shard(num_columns, self.rank, self.world, C.byref(first), C.byref(count)))
        return first.value, count.value

    def gather_fluxes(self, local_ptr, num_columns, all_ptr, on_device):
        self.api.check(self.lib.grt_multi_gather_fluxes(self.m, C.c_void_p(local_ptr), num_columns, C.c_void_p(all_ptr),
                                                        int(on_device)))

    def max(self, value):
        v = C.c_double(value)
        self.api.check(self.lib.grt_multi_max(self.m, C.byref(v)))
        return v.value

    def destroy(self):
        self.api.check(self.lib.grt_multi_destroy(C.byref(self.m)))
