"""Rehearsal on the CPU of the multi-GPU path: contiguous column shards, ONE gather of the [columns, 12]
flux blocks to rank 0, max-over-ranks timing -- with column counts that do NOT divide by the number of ranks (the
north-star's 100 columns over 8 GPUs: 13 x 7 + 9), which dist.gather only accepts because short blocks are padded.

  * torch.distributed over gloo, world size 2 (what bench.py does over RCCL): every rank computes REAL flux blocks for its
    own columns -- the CPU checker stands in for Pipeline.fluxes, which needs a GPU -- and rank 0 must end up with exactly
    what a single process computes for all columns;
  * the library's own C entry points grt_multi_* (what examples/rfmip_batch_driver.c calls) with the file transport, 8
    processes x 100 columns and 3 x 10, host buffers.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from grtcode_amd import multi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_is_a_partition(lib):
    import ctypes as C
    for ncol, world in [(1800, 8), (100, 8), (7, 2), (3, 8), (16, 1), (9, 8)]:
        blocks = [multi.shard(ncol, r, world) for r in range(world)]
        cols = [c for first, n in blocks for c in range(first, first + n)]
        assert cols == list(range(ncol))
        assert max(n for _, n in blocks) == -(-ncol // world)
        for r in range(world):                                   # the C rule is the same rule
            f, n = C.c_int(), C.c_int()
            assert lib.grt_multi_shard(ncol, r, world, C.byref(f), C.byref(n)) == 0
            assert (f.value, n.value) == blocks[r]
    assert multi.shard(1800, 3, 8) == (675, 225)        # 1 800 columns -> 225 per GPU
    assert [multi.shard(100, r, 8)[1] for r in range(8)] == [13] * 7 + [9]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def cpu_flux_block(columns, root):
    """[len(columns), 12] integrated fluxes of synthetic columns on a small band, from the CPU checker: the stand-in for
    the block a rank's Pipeline.fluxes() holds on its GPU."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from grtcode_amd import api, synthetic as syn
    from oracle.bindings import Oracle
    from scenario import Band
    from test_gpu_pipeline import oracle_column
    orc, lib = Oracle(), api.load_library()
    lw = Band(os.path.join(root, "lw"), 600.0, 680.0, 1.0, 300)
    sw = Band(os.path.join(root, "sw"), 2000.0, 2400.0, 10.0, 300, sw=True)
    emis, alb = np.full(lw.nw, 0.98), np.full(sw.nw, 0.2)
    w, y = sw._csv_values(*sw.tab["solar"])
    solar = orc.normalize_solar(sw.w0, sw.dw, orc.interp_to_grid(sw.w0, sw.dw, sw.nw, w, y))
    out = np.zeros((len(columns), 12))
    for k, c in enumerate(columns):
        col = syn.profile(c, 7)
        out[k, :6] = oracle_column(orc, lib, lw, col, True, emis, alb, solar)["integ"]
        out[k, 6:] = oracle_column(orc, lib, sw, col, False, emis, alb, solar)["integ"]
    return out


def _worker(rank, world, port, ncol, root):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GRT_TIPS_QUIET"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = multi.shard(ncol, rank, world)
    local = torch.from_numpy(cpu_flux_block(range(first, first + count), os.path.join(root, f"rank{rank}")))
    got = multi.gather_fluxes(local, rank, world, num_columns=ncol)
    slowest = multi.max_over_ranks(0.5 + rank, torch.device("cpu"))
    dist.barrier()
    if rank == 0:
        np.save(os.path.join(root, "gathered.npy"), got.numpy())
        assert slowest == 0.5 + (world - 1)
    else:
        assert got is None
    dist.destroy_process_group()


def test_two_rank_gather_of_real_flux_blocks_uneven_shards(tmp_path):
    world, ncol = 2, 5                                    # blocks of 3 and 2 columns
    mp.spawn(_worker, args=(world, _free_port(), ncol, str(tmp_path)), nprocs=world, join=True)
    got = np.load(str(tmp_path / "gathered.npy"))
    os.environ["GRT_TIPS_QUIET"] = "1"
    want = cpu_flux_block(range(ncol), str(tmp_path / "serial"))
    assert got.shape == (ncol, 12)
    assert np.array_equal(got, want)                      # same arithmetic on every rank: bit-identical
    assert np.all(got[:, 0] > 0) and len({round(x, 9) for x in got[:, 0]}) == ncol


C_RANK = r"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, %(root)r)
from grtcode_amd import multi
rank, world, ncol, rdv = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
m = multi.Multi(multi.FILES, 0, rank, world, rdv)
first, count = m.shard(ncol)
per = -(-ncol // world)
for step in range(3):                                            # three gathers in a row: the files of one call never meet the next
    local = np.array([[1000.0 * c + k + 0.25 * step for k in range(12)] for c in range(first, first + count)], dtype=np.float64).reshape(count, 12)
    out = np.full((per * world, 12), -1.0) if rank == 0 else None
    m.gather_fluxes(local.ctypes.data if count else 0, ncol, out.ctypes.data if rank == 0 else 0, False)
    if rank == 0:
        want = np.array([[1000.0 * c + k + 0.25 * step for k in range(12)] for c in range(ncol)])
        assert np.array_equal(out[:ncol], want), (step, out[:ncol], want)
        assert np.all(out[ncol:] == 0.0)
    assert m.max(10.0 + rank + step) == 10.0 + (world - 1) + step
m.destroy()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("world,ncol", [(8, 100), (3, 10), (4, 3)])
def test_c_entry_points_gather_through_files(tmp_path, world, ncol):
    """grt_multi_create / _shard / _gather_fluxes / _max / _destroy across `world` processes, host buffers, file transport
    (the RCCL transport needs one GPU per rank: exercised on the GPU box at world size 1 and by the driver's 8-GPU runs)."""
    script = tmp_path / "rank.py"
    script.write_text(C_RANK % {"root": ROOT})
    rdv = tmp_path / "rdv"
    rdv.mkdir()
    env = dict(os.environ, GRT_MULTI_TIMEOUT="120")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(ncol), str(rdv)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, (r, out, err[-2000:])
        assert f"rank {r} ok" in out
    assert os.listdir(rdv) == []                           # every exchange file, marker and the job's nonce are gone


def test_rendezvous_directory_is_reusable_and_stale_files_are_ignored(tmp_path):
    """a job used to leave its last seen_<epoch>_rank*.bin markers behind; the next job in the same directory
    restarted its epochs at 0, met them, passed a barrier early and hung its peer.  Now rank 0 clears what its job wrote
    when it is destroyed, and file names carry a job tag (GRT_MULTI_JOB) so that files of other jobs are never read.
    Here: the same directory three times in a row under one tag, a fourth time under the default tag, with leftovers of
    a crashed job under a third tag -- markers, blocks of the right size with wrong numbers -- lying in it throughout."""
    script = tmp_path / "rank.py"
    script.write_text(C_RANK % {"root": ROOT})
    rdv = tmp_path / "rdv"
    rdv.mkdir()
    stale = ["seen_crashed_0_rank0.bin", "seen_crashed_0_rank1.bin", "seen_crashed_1_rank1.bin", "max_crashed_1_rank1.bin",
             "fluxes_crashed_0_rank1.bin", "done_crashed_0_rank1.bin", "notes.txt"]
    for name in stale:
        (rdv / name).write_bytes(np.full(60, -7.0).tobytes()[: 384 if name.startswith("fluxes") else 8])
    world, ncol = 2, 9
    for job, tag in enumerate(["nightly-7", "nightly-7", "nightly-7", None]):
        env = dict(os.environ, GRT_MULTI_TIMEOUT="60")
        env.pop("GRT_MULTI_JOB", None)
        if tag:
            env["GRT_MULTI_JOB"] = tag
        procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(ncol), str(rdv)], stdout=subprocess.PIPE,
                                  stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
        for r, p in enumerate(procs):
            out, err = p.communicate(timeout=200)
            assert p.returncode == 0, (job, r, out, err[-2000:])
        assert sorted(os.listdir(rdv)) == sorted(stale), (job, os.listdir(rdv))


def test_ranks_need_not_be_alive_together_for_a_gather(tmp_path):
    """Rank 1 writes its block, says it is done and exits before rank 0 even starts (how 8 ranks take turns on one GPU in
    tests/test_gpu_baseline_configs.py): no step of the gather may wait for rank 0."""
    script = tmp_path / "rank.py"
    script.write_text(C_RANK.replace("assert m.max(10.0 + rank + step) == 10.0 + (world - 1) + step", "pass") % {"root": ROOT})
    rdv = tmp_path / "rdv"
    rdv.mkdir()
    env = dict(os.environ, GRT_MULTI_TIMEOUT="30")
    for r in (1, 0):
        p = subprocess.run([sys.executable, str(script), str(r), "2", "5", str(rdv)], capture_output=True, text=True, timeout=120, env=env)
        assert p.returncode == 0, (r, p.stdout, p.stderr[-2000:])
    assert os.listdir(rdv) == []


def test_c_gather_reports_a_missing_rank(tmp_path):
    """A rank that never shows up is an error with a message, not a hang."""
    script = tmp_path / "rank.py"
    script.write_text(C_RANK % {"root": ROOT})
    rdv = tmp_path / "rdv"
    rdv.mkdir()
    p = subprocess.run([sys.executable, str(script), "0", "2", "4", str(rdv)], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, GRT_MULTI_TIMEOUT="1"))
    assert p.returncode != 0
    assert "timed out" in p.stderr and "rank1" in p.stderr
