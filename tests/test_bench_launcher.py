"""`python bench.py --gpus N` must produce N ranks by itself: with no launcher around it the parent --
before torch or HIP exist in it -- starts N fresh child processes with RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, relays
rank 0's one JSON line and exits non-zero if any rank fails.  The reference fans out plain processes the same way
(GRTworkflow/run-rfmip-irf.sh:103-132).  Rehearsed here WITHOUT a GPU (GRT_BENCH_REHEARSAL=1 on a box with no device:
placeholder flux blocks whose values encode the column index, gloo instead of RCCL): the launcher, the sharding (100
columns over 8 ranks = 13 x 7 + 9), the ONE padded gather of the job's output and rank 0's checks of what it delivered."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, timeout=300, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(GRT_BENCH_REHEARSAL="1", **env)
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)


def only_line(r):
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, (r.stdout, r.stderr[-2000:])
    return json.loads(lines[0])


def test_two_ranks_weak_scaling_one_gather():
    r = run(["--gpus", "2", "--cols", "8", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-3000:]
    line = only_line(r)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["columns_per_step"] == 16
    assert line["config"]["shards"] == [[0, 8], [8, 8]]
    c = line["collective"]
    assert c["gathers_in_timed_region"] == 1                    # ONE gather, after the last step
    assert c["bytes_per_rank_per_gather"] == 3 * 8 * 12 * 8     # the job's [steps][columns per rank][12] doubles
    assert "NO DEVICE" in line["rehearsal"] and line["rccl_ranks"] == 0 and c["backend"] == "gloo"


def test_eight_ranks_rfmip_100_columns_strong_scaling():
    r = run(["--gpus", "8", "--columns", "100", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-3000:]
    line = only_line(r)
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["config"]["columns_per_step"] == 100
    assert [s[1] for s in line["config"]["shards"]] == [13] * 7 + [9]
    assert [s[0] for s in line["config"]["shards"]] == [13 * k for k in range(8)]
    assert abs(line["value"] - 100 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-9


def test_gather_every_k_steps():
    r = run(["--gpus", "2", "--cols", "4", "--steps", "5", "--warmup", "0", "--gather-every", "2"])
    assert r.returncode == 0, r.stderr[-3000:]
    line = only_line(r)
    assert line["collective"]["gathers_in_timed_region"] == 3   # after steps 2, 4 and the last


def test_more_ranks_than_columns_leaves_empty_shards():
    r = run(["--gpus", "4", "--columns", "3", "--steps", "1", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert only_line(r)["config"]["shards"] == [[0, 1], [1, 1], [2, 1], [3, 0]]


def test_a_failing_rank_fails_the_job_and_prints_no_line():
    r = run(["--gpus", "3", "--cols", "2", "--steps", "1", "--warmup", "0"], GRT_BENCH_TEST_FAIL_RANK="1", timeout=120)
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])
    assert r.stdout.strip() == ""
    assert "rank 1 exited with code 7" in r.stderr


def test_under_a_launcher_the_ranks_are_not_spawned_again():
    """torch.distributed.run (the driver's N > 1 command) sets WORLD_SIZE: bench.py is then ONE rank of N."""
    r = run(["--gpus", "1", "--cols", "2", "--steps", "1", "--warmup", "0"], RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    assert r.returncode == 0, r.stderr[-2000:]
    assert only_line(r)["n_gpus"] == 1
    r = run(["--gpus", "2"], RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_under_torch_distributed_run_as_the_driver_launches_it():
    """The driver's N > 1 command line: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e["GRT_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--cols", "4"],
                       capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = only_line(r)
    assert line["n_gpus"] == 2 and line["config"]["shards"] == [[0, 4], [4, 4]] and line["collective"]["gathers_in_timed_region"] == 1


def test_the_parent_never_imports_torch():
    """The launcher must not initialise anything GPU-related before it starts the ranks: spawn_ranks and everything
    main() runs before it import neither torch nor the library."""
    src = open(BENCH).read()
    head = src[: src.index("    rank = int(os.environ.get(\"RANK\", 0))")]
    launcher = head[head.index("def spawn_ranks"): head.index("class PlaceholderEngine")]
    assert "import torch" not in launcher and "grtcode_amd" not in launcher
    main_head = head[head.index("def main():"):]
    assert "import torch" not in main_head and "grtcode_amd" not in main_head
    top = src[: src.index("def cpu_baseline")]
    assert "import torch" not in top and "from grtcode_amd" not in top


def test_both_launch_paths_give_a_rank_the_same_environment(tmp_path):
    """`python bench.py --gpus 2` (spawn_ranks) and `python -m torch.distributed.run ... bench.py --gpus 2` (the driver) must not
    differ in what a rank runs under: the rank sets RANK_ENVIRONMENT itself, before torch exists in it (task 6).
    Both paths start from an environment WITHOUT the variable; each rank dumps what it ended up with."""
    import json
    import socket
    base = {k: v for k, v in os.environ.items()
            if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")}
    base["GRT_BENCH_REHEARSAL"] = "1"
    dumps = {}
    for how in ("spawned", "torchrun"):
        e = dict(base, GRT_BENCH_DUMP_ENV=str(tmp_path / how))
        if how == "spawned":
            cmd = [sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--cols", "2"]
        else:
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                port = so.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                   "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--cols", "2"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        line = only_line(r)
        assert len(line["ms_per_step_by_rank"]) == 2 and all(t > 0. for t in line["ms_per_step_by_rank"])
        dumps[how] = [json.load(open(str(tmp_path / how) + f".rank{k}")) for k in range(2)]
    assert dumps["spawned"] == dumps["torchrun"]
    assert all(d == {"HSA_ENABLE_IPC_MODE_LEGACY": "0"} for d in dumps["spawned"])
