#!/usr/bin/env python3
"""bench.py -- columns/s of the line-by-line hot path on N MI355X (one process per GPU).

Metric (BASELINE.json): columns/sec, 60 layers, 1 cm-1 LBL, LW+SW.  Workload = SURVEY.md §8(d)
grid G1 (LW 1-3250 + SW 1-50000 cm-1 @ 1 cm-1) with 1.0 M / 1.5 M synthetic lines, seeded
synthetic columns.  A step = one batch of COLS columns per GPU through the whole path
(host prologue -> line-by-line tau -> Rayleigh+combine -> LW/SW solver -> spectral integration
-> [N>1] RCCL gather of the 12 integrated fluxes per column to rank 0).  Line lists and tables
are resident in HBM before the timed region; per-column inputs are 61-level host profiles.
Weak scaling (default): every rank processes its own COLS columns per step (columns shard with no
data-path collective other than that final gather).  Strong scaling (--columns N, e.g. 100 for
BASELINE.json's RFMIP-IRF config, 1800 for the replicated set): a step is the whole fixed set of N
columns, sharded over the ranks in ceil-sized contiguous blocks (100 over 8 GPUs: 13 x 7 + 9), each
rank running its block in chunks of COLS.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector


def cpu_baseline(lw_grid, sw_grid, lw_lines, sw_lines, thin_lw=1, thin_sw=1):
    """Reference OpenMP path (oracle/_ref, the reference's own C) -- or our restatement when the
    prebuilt reference library is absent -- timed on a bounded sample: column 0 of the bench's own
    workload, full LW+SW grids and solvers (line lists thinned by 1/thin only if asked; gas-optics time
    is then scaled back by `thin`).  Returns (the cpu_baseline object of the bench line, the checker's
    tau and fluxes of that column per band -- what the GPU's column 0 is compared with)."""
    from oracle import reference_column as RC
    from grtcode_amd import api, synthetic as syn, workload as W
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from scenario import MOLTAB, mol_mass
    cores = RC.set_omp_threads(min(os.cpu_count() or 1, 16))
    kind, chk, orc = RC.checker(omp=True)
    if kind != "reference":
        cores = 1                       # the restatement is a scalar port: ~100 s for the full-size column, so the sample is
        thin_lw = max(thin_lw, 10)      # bounded by thinning the line lists (gas-optics time scaled back by the factor);
        thin_sw = max(thin_sw, 10)      # a thinned column is no parity reference for the full one (parity: null)
    lib = api.load_library()
    col = syn.profile(0, W.NUM_LEVELS)
    total = 0.0
    detail, bands = {}, {}
    for band, grid, nlines, thin, seed, sw in (("lw", lw_grid, lw_lines, thin_lw, 20261003, False),
                                               ("sw", sw_grid, sw_lines, thin_sw, 20261004, True)):
        r = RC.band_column(kind, chk, orc, lib.Q, col, grid, W.band_lines(nlines, grid, seed), syn.tables(sw=sw),
                           W.MOL_ORDER, MOLTAB, mol_mass, W.CIA_PAIRS, sw, thin=thin)
        detail[band] = dict(gas_optics_s_sample=round(r["t_gas"], 3), rest_s=round(r["t_rest"], 3), thin=thin)
        total += r["t_gas"] * thin + r["t_rest"]
        bands[band] = r
    base = {"value": 1.0 / total, "unit": "columns/s", "cores": cores, "kind": kind,
            "sample": ("1 column of the bench workload at full size (LW+SW at 1 cm-1, 60 layers, all lines)"
                       if thin_lw == 1 and thin_sw == 1 else
                       f"1 column, LW+SW at 1 cm-1, 60 layers, line lists thinned 1/{thin_lw} (LW) and 1/{thin_sw} (SW); "
                       f"gas-optics time scaled back by the thinning factor, all other stages at full size"),
            "seconds_per_column_est": round(total, 2), "detail": detail}
    return base, bands


def fine_grid_column(wl, dw, reps, compare_layers=0):
    """One longwave column (1-3250 cm-1, the bench's 1.0 M lines, 60 layers) on a fine grid through the production form
    -- SURVEY §8(d) grids G2 (0.1 cm-1) and G3 (0.001 cm-1: BASELINE.json's "~3M-wavenumber" grid, n = 3 249 001, windows
    of 50 001 points) -- timed with HIP events on the library stream.  compare_layers > 0: the cell hierarchy against the
    ring kernel (every window point evaluated) on a column of that many layers."""
    from grtcode_amd import api, synthetic as syn, workload as W
    grid_spec = (W.LW_GRID[0], W.LW_GRID[1], dw)
    go, grid = W.build_band(wl.device, grid_spec, wl.lw_lines, wl.lw_files, W.NUM_LEVELS)
    col = syn.profile(0, W.NUM_LEVELS)

    def set_column(g, c):
        for m in W.MOL_ORDER:
            g.set_molecule_ppmv(m, c["ppmv"][m])
        g.set_cfc_ppmv(0, c["cfc_ppmv"][0])
        g.set_cfc_ppmv(1, c["cfc_ppmv"][1])
        g.set_cia_ppmv(0, c["ppmv"][syn.N2])
        g.set_cia_ppmv(1, c["ppmv"][syn.O2])
    set_column(go, col)
    go.tune(fast=3)
    opt = api.OpticsObject(W.NUM_LEVELS - 1, grid, wl.device)
    go.calculate_optical_depth(col["p"], col["t"], opt)          # warm-up (allocations)
    api.profile_read(1, reset=True)                               # (a reset clears the records of every tag)
    t0 = time.perf_counter()
    for _ in range(reps):
        go.calculate_optical_depth(col["p"], col["t"], opt)
    wall = (time.perf_counter() - t0) / reps
    tags = {t: api.profile_read(t) for t in (1, 2, 6, 7)}
    api.profile_read(1, reset=True)
    kern_ms = sum(v[0] for v in tags.values()) / reps
    info = go.last_launch()
    L, n = W.NUM_LEVELS - 1, int(grid.n)
    fsteps = int(np.ceil(25.0 / dw))
    out = {"dw": dw, "n": n, "lines": wl.total_lines["lw"], "ms_per_column": kern_ms, "columns_per_s": 1e3 / kern_ms,
           "wall_ms_per_column": 1e3 * wall, "ran": {k: info[k] for k in ("fast", "tile", "tree_levels", "halo", "moments")},
           "voigt_points_per_column": float(L) * wl.total_lines["lw"] * (2 * fsteps + 1),
           # B_band = 60 S + n (8 C_tab + 48 L + 16 V), C_tab = 11 in the longwave
           "hbm_algorithmic_gb_per_s": (60.0 * wl.total_lines["lw"] + n * (88.0 + 48.0 * L + 16.0 * (L + 1))) / (kern_ms * 1e-3) / 1e9}
    opt.destroy()
    go.destroy()
    if compare_layers:
        V = compare_layers + 1
        go, grid = W.build_band(wl.device, grid_spec, wl.lw_lines, wl.lw_files, V)
        c = syn.profile(0, V)
        set_column(go, c)
        opt = api.OpticsObject(V - 1, grid, wl.device)
        taus = {}
        for fast in (3, 2):
            go.tune(fast=fast)
            go.calculate_optical_depth(c["p"], c["t"], opt)
            taus[fast] = opt.read()[0]
        scale = np.abs(taus[2]).max(axis=1, keepdims=True)
        out["tree_vs_ring"] = {"layers": compare_layers, "max_diff_of_layer_max": float(np.max(np.abs(taus[3] - taus[2]) / scale))}
        opt.destroy()
        go.destroy()
    return out


def reference_abi_rate(wl, ncol=3):
    """What an UNCHANGED caller of the reference's one-column interface gets (framework/src/driver.c:360-424): per band
    set_*_ppmv -> calculate_optical_depth -> rayleigh_scattering -> add_optics -> calculate_{lw,sw}_fluxes with HOST flux
    arrays (2 V n doubles over PCIe per call) -> the caller's trapezoid -> destroy_optics.  Reference-order arithmetic
    (fast = 0, the default of a new gas-optics object) unless GRT_GAS_OPTICS_FAST is set; here both are timed."""
    from grtcode_amd import api, synthetic as syn, workload as W
    V = W.NUM_LEVELS
    lw = api.LongwaveObject(V, wl.grid_lw, wl.device)
    sw = api.ShortwaveObject(V, wl.grid_sw, wl.device)
    objs = {}
    for name, go, grid in (("lw", wl.go_lw, wl.grid_lw), ("sw", wl.go_sw, wl.grid_sw)):
        objs[name] = (go, api.OpticsObject(V - 1, grid, wl.device), api.OpticsObject(V - 1, grid, wl.device), grid,
                      (np.zeros((V, grid.n)), np.zeros((V, grid.n))))       # flux arrays allocated once, as driver.c:682-688

    def column(c):
        col = syn.profile(c, V)
        total = []
        for name in ("lw", "sw"):
            go, gas, ray, grid, bufs = objs[name]
            for m in W.MOL_ORDER:
                go.set_molecule_ppmv(m, col["ppmv"][m])
            go.set_cfc_ppmv(0, col["cfc_ppmv"][0])
            go.set_cfc_ppmv(1, col["cfc_ppmv"][1])
            go.set_cia_ppmv(0, col["ppmv"][syn.N2])
            go.set_cia_ppmv(1, col["ppmv"][syn.O2])
            go.calculate_optical_depth(col["p"], col["t"], gas)
            ray.rayleigh(col["p"])
            tot = api.add_optics([gas, ray])
            if name == "lw":
                up, dn = lw.fluxes(tot, col["t_surf"], col["t_layer"], col["t"], wl.emis, bufs)
            else:
                up, dn = sw.fluxes(tot, col["mu0"], 0.5, wl.albedo, wl.albedo, col["tsi"], wl.solar, bufs)
            dw = grid.dw
            total += [float(np.sum(0.5 * (r[:-1] + r[1:]) * dw)) for r in (up[0], up[-1], dn[0], dn[-1])]   # driver.c:302-326
            tot.destroy()
        return total
    out = {}
    for fast in (0, 3):
        wl.go_lw.tune(fast=fast)
        wl.go_sw.tune(fast=fast)
        column(0)
        t0 = time.perf_counter()
        for c in range(ncol):
            fl = column(c)
        out["fast%d_columns_per_s" % fast] = ncol / (time.perf_counter() - t0)
    out["note"] = ("one column per call, synchronous, 2*V*n doubles of spectral flux copied to the host per solver call "
                   "(49 MB per shortwave column); fast0 = what an unchanged driver gets, fast3 = the same driver with "
                   "GRT_GAS_OPTICS_FAST=3 in its environment")
    for o in (lw, sw):
        o.destroy()
    for name in objs:
        objs[name][1].destroy()
        objs[name][2].destroy()
    return out


def parity_of_column0(wl, fluxes, bands, kind):
    """The GPU's column 0 of the last timed step against the CPU checker's column 0 (same inputs): the twelve
    integrated fluxes [W m-2] and the spectral gas optical depths of both bands."""
    from oracle.reference_column import tau_metrics
    from grtcode_amd import api, workload as W
    L = W.NUM_LEVELS - 1
    out = {"kind": kind, "column": 0, "tolerance_w_m2": 1e-3}
    want = np.concatenate([bands["lw"]["integ"], bands["sw"]["integ"]])
    out["max_abs_flux_diff_w_m2"] = float(np.abs(fluxes[0] - want).max())
    out["flux_diff_w_m2"] = {k: float(fluxes[0, i] - want[i]) for k, i in
                             (("rlut", 0), ("rlus", 1), ("rlds", 4), ("rsut", 6), ("rsus", 7), ("rsdt", 9), ("rsds", 10))}
    worst = {"of_layer_max": 0.0, "pointwise_rel": 0.0, "transmission": 0.0}
    for bi, band in enumerate(("lw", "sw")):
        nw = bands[band]["nw"]
        got = api.device_to_host(wl.device, wl.pipe.views(bi)["tau_gas"], (L, nw))
        m = tau_metrics(got, bands[band]["tau_gas"])
        out["tau_" + band] = m
        worst = {k: max(worst[k], m[k]) for k in worst}
    out["max_tau_err_of_layer_max"] = worst["of_layer_max"]
    out["max_tau_err_pointwise_rel"] = worst["pointwise_rel"]
    out["max_transmission_err"] = worst["transmission"]
    out["ok"] = bool(out["max_abs_flux_diff_w_m2"] <= out["tolerance_w_m2"])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cols", type=int, default=int(os.environ.get("GRT_BENCH_COLS", 8)), help="columns per GPU per step")
    ap.add_argument("--fast", type=int, default=int(os.environ.get("GRT_BENCH_FAST", 3)),
                    help="3: fused form, far wings by cell moments, two passes (production); 1: the same in one pass; "
                         "2: fused form, ring kernel; 0: reference operation order")
    ap.add_argument("--columns", type=int, default=0, help="strong scaling: a step is this fixed number of columns, sharded "
                    "over the ranks (0: weak scaling, --cols columns per GPU per step)")
    ap.add_argument("--tile", type=int, default=0, help="exploration only: wavenumbers (cells) per workgroup of the line kernel")
    ap.add_argument("--lw-nslice", type=int, default=0, help="exploration only: line slices per tile of the longwave launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the reference-order and fine-grid measurements after the timed region")
    ap.add_argument("--lw-lines", type=int, default=None)
    ap.add_argument("--sw-lines", type=int, default=None)
    ap.add_argument("--lw-dw", type=float, default=None, help="exploration only: longwave grid spacing (default 1 cm-1)")
    ap.add_argument("--sw-dw", type=float, default=None, help="exploration only: shortwave grid spacing (default 1 cm-1)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    # the C library reports loaded species on stdout (like the reference's log_mesg); keep stdout
    # for the single JSON line by pointing fd 1 at stderr until the result is printed
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from grtcode_amd import api, multi, workload as W

    # GRT_BENCH_REHEARSAL=1: a logic rehearsal of the N > 1 path on a box with fewer GPUs than ranks -- ranks share the
    # devices there are and the collectives go over gloo on host copies.  Timings of such a run mean nothing (the line
    # says so); what it exercises is the sharding, the line-list cache hand-over, the padded gather and the barriers.
    rehearsal = os.environ.get("GRT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    # GRT_BENCH_FORCE_DIST=1 exercises the RCCL path (init, stream-ordered gather, max-reduce) at world size 1
    force_dist = os.environ.get("GRT_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    device = api.create_device(local_rank)
    if world > 1:
        # the line lists are drawn once (rank 0) and shared through a cache directory instead of once per rank
        os.environ.setdefault("GRT_LINES_CACHE", os.path.join(os.environ.get("TMPDIR", "/tmp"), f"grt_lines_cache_{os.getuid()}"))
        if rank != 0:
            dist.barrier()
    lw_grid = (W.LW_GRID[0], W.LW_GRID[1], args.lw_dw or W.LW_GRID[2])
    sw_grid = (W.SW_GRID[0], W.SW_GRID[1], args.sw_dw or W.SW_GRID[2])
    wl = W.G1Workload(device, args.cols, lw_lines=args.lw_lines or W.LW_LINES,
                      sw_lines=args.sw_lines or W.SW_LINES, fast=args.fast, lw_grid=lw_grid, sw_grid=sw_grid, tile=args.tile, lw_nslice=args.lw_nslice)
    if world > 1 and rank == 0:
        dist.barrier()                                              # the lists are in the cache: the other ranks may build
    strong = args.columns > 0
    total_per_step = args.columns if strong else world * args.cols
    first, count = multi.shard(total_per_step, rank, world)         # weak scaling: args.cols columns per rank
    per = -(-total_per_step // world)
    # this rank's block, in chunks of at most args.cols columns (one chunk in weak mode)
    chunks = []
    for lo in range(0, count, args.cols):
        n = min(args.cols, count - lo)
        (gc, keep), _ = wl.columns(first + lo, n)
        chunks.append((lo, n, gc, keep))
    out = torch.zeros(max(per, 1), api.GRT_FLUXES_PER_COLUMN, dtype=torch.float64, device="cuda")
    use_dist = world > 1 or force_dist
    gathered = [torch.zeros_like(out) for _ in range(world)] if (use_dist and rank == 0) else None
    stream = torch.cuda.ExternalStream(wl.pipe.stream(), device=torch.device("cuda", local_rank))
    row_bytes = 8 * api.GRT_FLUXES_PER_COLUMN
    result = {}

    def step():
        for lo, n, gc, _ in chunks:
            wl.pipe.run(gc, out.data_ptr() + lo * row_bytes)
        if use_dist and rehearsal:
            wl.pipe.sync()
            result["all"] = multi.gather_fluxes(out[:count].cpu(), rank, world, None, num_columns=total_per_step)
        elif use_dist:
            with torch.cuda.stream(stream):      # RCCL gather ordered after the kernels, no host sync
                result["all"] = multi.gather_fluxes(out[:count], rank, world, gathered, num_columns=total_per_step) if world > 1 \
                    else (dist.gather(out, gathered, dst=0), gathered[0])[1]

    def barrier():
        wl.pipe.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    api.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = multi.max_over_ranks(elapsed, torch.device("cpu") if rehearsal else torch.device("cuda", local_rank))

    if rank == 0:
        fluxes = out.cpu().numpy()
        assert np.all(np.isfinite(fluxes)), "non-finite integrated fluxes"
        if world > 1:
            # what the gather delivered: every column of the step, in order, this rank's own block first
            allf = result["all"].cpu().numpy()
            assert allf.shape == (total_per_step, api.GRT_FLUXES_PER_COLUMN), allf.shape
            assert np.all(np.isfinite(allf)) and np.all(allf[:, 0] > 0.), "gathered fluxes incomplete"
            assert np.array_equal(allf[:count], fluxes[:count])
        ms = {tag: api.profile_read(tag) for tag in (1, 2, 3, 4, 5, 6, 7)}
        L, V = W.NUM_LEVELS - 1, W.NUM_LEVELS
        n_lw, n_sw = wl.grid_lw.n, wl.grid_sw.n
        S = wl.total_lines
        # algorithmic bytes of the dominant kernel (line-by-line tau, SW-band launch), per launch:
        # SURVEY.md §8(d) terms it owns: 60 B/line once per column + per wavenumber 8*C_tab table
        # reads + 8*L tau written once (C_tab = 4 H2O + 1 O3 + 2 CFC + 3 CIA = 10 tables)
        cols_launch = count / max(len(chunks), 1)          # columns per line-kernel launch on this rank (= --cols in weak mode)
        bytes_gas = lambda nlines, n: cols_launch * (60.0 * nlines + n * (8.0 * 10 + 8.0 * L))
        dom_ms = ms[2][0] / max(ms[2][1], 1)
        achieved = bytes_gas(S["sw"], n_sw) / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        fsteps = int(np.ceil(25.0 / sw_grid[2]))
        points = lambda nlines: float(L) * nlines * (2 * fsteps + 1) * cols_launch    # Voigt evaluations per launch
        valu_flop = 12.0 * points(S["sw"])                                            # ~12 flop far-wing point
        # HBM-side bytes per launch are PMC counters (FETCH_SIZE, WRITE_SIZE: separate rocprofv3 --pmc passes of this same
        # command, scripts/profile_round.sh) -- they cannot be read inside an ordinary run, so the figure of the last
        # profiled run is carried here, labelled as such, and dropped when the configuration differs
        traffic, traffic_src, solver_traffic = None, None, {}
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("cols") == args.cols and tj.get("fast") == args.fast and not strong:
                    traffic = tj["gas_optics_sw"]["hbm_bytes_per_launch"]
                    traffic_src = f"profiles/traffic_latest.json: rocprofv3 --pmc passes of this command, round tag {tj.get('tag')} (not measured in this run)"
                    solver_traffic = {k: v.get("hbm_bytes_per_launch") for k, v in tj.get("solvers", {}).items()}
            except Exception:
                traffic = None
        total_cols = total_per_step * args.steps
        line_kernel = "gas_optics_mp_kernel" if args.fast in (1, 3) else "gas_optics_kernel"
        line = {
            "metric": "columns/sec (60-layer, 1 cm\u207b\u00b9 LBL, LW+SW)" if (lw_grid[2], sw_grid[2]) == (1.0, 1.0) else
                      f"columns/sec (60-layer LBL, LW @{lw_grid[2]:g} + SW @{sw_grid[2]:g} cm\u207b\u00b9)",
            "value": total_cols / elapsed, "unit": "columns/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            **({"rehearsal": "ranks share devices, gloo on host copies: timings are meaningless"} if rehearsal else {}), "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("SURVEY §8d grid G1: LW 1-3250 + SW 1-50000 cm-1 @1 cm-1" if (lw_grid[2], sw_grid[2]) == (1.0, 1.0) else
                                    f"LW {lw_grid[0]:g}-{lw_grid[1]:g} cm-1 @{lw_grid[2]:g} cm-1 (n = {n_lw}) + SW {sw_grid[0]:g}-{sw_grid[1]:g} cm-1 "
                                    f"@{sw_grid[2]:g} cm-1 (n = {n_sw}) -- NOT the headline grid") +
                                   f", 60 layers, 7 absorbers, {S['lw']} LW + {S['sw']} SW synthetic lines, H2O/O3 continua, 2 CFC, 3 CIA, "
                                   "clear sky, integrated fluxes",
                       "columns_per_gpu_per_step": args.cols if not strong else None,
                       "columns_per_step": total_per_step, "chunk_columns": args.cols, "fast": args.fast,
                       "arithmetic": {0: "reference operation order", 1: "fused form, far wings by cell moments",
                                      2: "fused form, every window point in the ring",
                                      3: "fused form, far wings by cell moments, two passes"}.get(args.fast, str(args.fast)),
                       "parallelism": f"columns sharded over {world} GPU(s), RCCL gather of 12 fluxes/column"},
            "roofline": {"kernel": f"{line_kernel} (line-by-line tau), SW-band launch", "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": dom_ms, "launches": ms[2][1],
                         "note": "this kernel is FP32/FP64-VALU bound by construction, see roofline_valu"},
            "roofline_valu": {"kernel": f"{line_kernel}, SW-band launch", "bound": "valu_fp32",
                              "achieved": valu_flop / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0,
                              "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s (algorithmic: 12 flop x L*S*F line-shape points, SURVEY §8d; the moment kernel "
                                      "delivers the far-wing points without evaluating them one by one)",
                              "frac": (valu_flop / (dom_ms * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS) if dom_ms > 0 else 0.0,
                              "voigt_points_per_launch": points(S["sw"]),
                              "gpoints_per_s": points(S["sw"]) / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0},
            "kernel_ms_per_step": {"gas_optics_lw": ms[1][0] / args.steps, "gas_optics_sw": ms[2][0] / args.steps,
                                   "far_field_lw": ms[6][0] / args.steps, "far_field_sw": ms[7][0] / args.steps,
                                   "lw_solver": ms[3][0] / args.steps, "sw_solver": ms[4][0] / args.steps,
                                   "clear_sky_optics": ms[5][0] / args.steps},
            "sample_fluxes_col0": {"rlut": fluxes[0, 0], "rlus": fluxes[0, 1], "rlds": fluxes[0, 4],
                                   "rsut": fluxes[0, 6], "rsdt": fluxes[0, 9], "rsds": fluxes[0, 10]},
        }
        # the solvers (fused form: Rayleigh + optics combination + solver + spectral trapezoid in one kernel).  HBM view
        # with SURVEY a19/a20's algorithmic bytes of the materialised interface (1 944 / 2 440 B per wavenumber) and with
        # what the fused kernels must move (tau_gas once + the per-wavenumber tables); the shortwave kernel's arithmetic
        # (two delta-Eddington solutions per layer: ~13 divisions, 6 exp in fp64) is done once, in the first sweep, which
        # parks the five properties of every layer for the second: `traffic_with_park`
        sol = {}
        for name, tag, n, surv, fused_b in (("lw", 3, n_lw, 16.0 * L + 8 + 16.0 * V, 8.0 * L + 8), ("sw", 4, n_sw, 24.0 * L + 24 + 16.0 * V, 8.0 * L + 24)):
            if ms[tag][1]:
                t = ms[tag][0] / ms[tag][1] * 1e-3
                sol[name] = {"avg_launch_ms": t * 1e3, "columns_per_launch": cols_launch,
                             "algorithmic_bytes_survey": surv * n * cols_launch, "achieved_gb_per_s_survey": surv * n * cols_launch / t / 1e9,
                             "frac_hbm_survey": surv * n * cols_launch / t / 1e9 / HBM_PEAK_GBS,
                             "compulsory_bytes_fused": fused_b * n * cols_launch, "achieved_gb_per_s_fused": fused_b * n * cols_launch / t / 1e9,
                             "traffic": solver_traffic.get(name + "_kernel")}
                if name == "sw":
                    # tau_gas once, albedo + solar, reflectances (2 V rows) and layer properties (5 L rows) written and read back
                    park_b = (8.0 * L + 24 + 2 * 8.0 * (2 * V + 5 * L)) * n * cols_launch
                    sol[name].update({"traffic_with_park": park_b, "achieved_gb_per_s_with_park": park_b / t / 1e9,
                                      "frac_hbm_with_park": park_b / t / 1e9 / HBM_PEAK_GBS})
        line["roofline_solvers"] = dict(sol, bound="hbm (shortwave: layer properties parked by the first sweep, read back by the second); "
                                                   "latency (longwave: 26 000 threads)", peak=HBM_PEAK_GBS, unit="GB/s")
        if world == 1 and not args.no_extras:
            # what an unchanged caller of calculate_optical_depth gets: the reference-order form (fast = 0)
            wl.go_lw.tune(fast=0, tile=args.tile, nslice=args.lw_nslice)
            wl.go_sw.tune(fast=0, tile=args.tile)
            step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(2):
                step()
            barrier()
            line["reference_order_columns_per_s"] = 2 * total_per_step / (time.perf_counter() - t0)
            line["reference_abi"] = reference_abi_rate(wl)
            wl.go_lw.tune(fast=args.fast, tile=args.tile, nslice=args.lw_nslice)
            wl.go_sw.tune(fast=args.fast, tile=args.tile)
            step()                              # column 0 of the production form back in the buffers (parity below)
            barrier()
            fluxes = out.cpu().numpy()
            if args.lw_dw is None and args.lw_lines is None:
                line["fine_grid"] = {"G2_lw_0.1cm-1": fine_grid_column(wl, 0.1, 4),
                                     "G3_lw_0.001cm-1": fine_grid_column(wl, 0.001, 2, compare_layers=12)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], ref_bands = cpu_baseline(lw_grid, sw_grid, args.lw_lines or W.LW_LINES,
                                                           args.sw_lines or W.SW_LINES)
            full = all(v["thin"] == 1 for v in line["cpu_baseline"]["detail"].values())
            line["parity"] = parity_of_column0(wl, fluxes, ref_bands, line["cpu_baseline"]["kind"]) if full else None
            if line["parity"] is not None and not line["parity"]["ok"]:
                sys.stderr.write("bench.py: PARITY FAILURE, no result line: " + json.dumps(line["parity"]) + "\n")
                wl.destroy()
                raise SystemExit(3)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    wl.destroy()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
