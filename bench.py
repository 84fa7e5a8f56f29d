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
SIMDS = 256 * 4                # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9               # MI355X_MICROARCH.md: peak engine clock


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
    # one thread, same workload (SURVEY §8d asks for 1 thread and all cores): the line lists thinned 1/8 so that the sample
    # stays near ten seconds -- gas-optics time scaled back by the factor, every other stage at full size
    one = None
    if kind == "reference":
        RC.set_omp_threads(1)
        thin1, t1, d1 = 8, 0.0, {}
        for band, grid, nlines, seed, sw in (("lw", lw_grid, lw_lines, 20261003, False), ("sw", sw_grid, sw_lines, 20261004, True)):
            r = RC.band_column(kind, chk, orc, lib.Q, col, grid, W.band_lines(nlines, grid, seed), syn.tables(sw=sw),
                               W.MOL_ORDER, MOLTAB, mol_mass, W.CIA_PAIRS, sw, thin=thin1)
            d1[band] = dict(gas_optics_s_sample=round(r["t_gas"], 3), rest_s=round(r["t_rest"], 3), thin=thin1)
            t1 += r["t_gas"] * thin1 + r["t_rest"]
        RC.set_omp_threads(cores)
        one = {"value": 1.0 / t1, "unit": "columns/s", "cores": 1, "seconds_per_column_est": round(t1, 2), "detail": d1,
               "sample": f"1 column, line lists thinned 1/{thin1} (gas-optics time scaled back), solvers at full size"}
    base = {"value": 1.0 / total, "unit": "columns/s", "cores": cores, "kind": kind,
            "host_cores": os.cpu_count(), "cores_note": "the GPU box's share of host cores for one GPU is 16; threads = min(host cores, 16)",
            "one_thread": one,
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
    if dw == 0.001:
        # the vector pipe's share in the two kernels of this column, from the PMC run of scripts/profile_g3.sh (carried,
        # like roofline_valu.issue_utilisation: counters cannot be read inside an ordinary run)
        try:
            g3 = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json"))).get("g3")
            if g3:
                out["issue_utilisation"] = {k: {"avg_launch_ms": v["avg_launch_ms"], "issue_utilisation": v.get("issue_utilisation"),
                                                "valu_wave_instructions": v.get("sq", {}).get("SQ_INSTS_VALU")}
                                            for k, v in g3["kernels"].items()}
                out["issue_utilisation"]["source"] = f"profiles/traffic_latest.json: rocprofv3 --pmc pass of scripts/fine_grid.py --dw 0.001, round tag {g3['tag']} (not measured in this run)"
        except Exception:
            pass
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
        out["tree_vs_ring"] = {"layers": compare_layers, "max_diff_of_layer_max": float(np.max(np.abs(taus[3] - taus[2]) / scale)),
                               "what": "SELF-COMPARISON of two forms of this library (cell hierarchy vs every window point in the ring), not "
                                       "parity.  Parity of a column of THIS size -- the same grid (n = 3 249 001), atmosphere and 60 layers, 10^6 lines of the physically scaled list -- against the reference's own C "
                                       "(18 minutes on 16 host threads, a one-off run): profiles/r3_g3_vs_reference.json, tau within 2.6e-7 of "
                                       "each layer's maximum, 2.9e-6 in transmission; in every suite run: "
                                       "tests/test_gpu_moment_tree.py::test_full_3m_point_grid_against_the_reference_c (16 000 lines x 4 layers)"}
        opt.destroy()
        go.destroy()
    return out


def reference_abi_rate(wl, ncol=3):
    """What an UNCHANGED caller of the reference's one-column interface gets (framework/src/driver.c:360-424): per band
    set_*_ppmv -> calculate_optical_depth -> rayleigh_scattering -> add_optics -> calculate_{lw,sw}_fluxes with HOST flux
    arrays (2 V n doubles over PCIe per call) -> the caller's trapezoid -> destroy_optics.  Reference-order arithmetic
    The production arithmetic (fast = 3) is the default of a new gas-optics object, GRT_GAS_OPTICS_FAST=0 selects the
    reference's operation order; here both are timed."""
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
                   "(49 MB per shortwave column); fast3 = what an unchanged driver gets (the default of a new object), fast0 = the same "
                   "driver with GRT_GAS_OPTICS_FAST=0 in its environment (the reference's operation order); the real binary: "
                   "profiles/r4_reference_driver_timing.json")
    for o in (lw, sw):
        o.destroy()
    for name in objs:
        objs[name][1].destroy()
        objs[name][2].destroy()
    return out


def parity_physical_list(device, fast, lw_grid, sw_grid):
    """A flux check that discriminates (task 2): SURVEY §8(d)'s line list makes a nearly black atmosphere whose
    fluxes barely depend on tau, so one more column is compared -- the same grids, atmosphere and line counts with the
    physically scaled strengths (synthetic.PHYSICAL_BANDS: outgoing longwave ~270 W m-2, 0.68 of the incoming shortwave
    reaches the surface) -- production form on the GPU against the reference's own C, twelve integrated fluxes [W m-2]."""
    from oracle import reference_column as RC
    from oracle.reference_column import tau_metrics
    from grtcode_amd import api, synthetic as syn, workload as W
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from scenario import MOLTAB, mol_mass
    kind, chk, orc = RC.checker(omp=True)
    if kind != "reference":
        return None                     # (the scalar restatement would need minutes for a full-size column)
    RC.set_omp_threads(min(os.cpu_count() or 1, 16))
    wl = W.G1Workload(device, 1, physical=True, fast=fast, lw_grid=lw_grid, sw_grid=sw_grid, spectral=True)
    col = syn.profile(0, W.NUM_LEVELS)
    lib = api.load_library()
    ref = {}
    for band, grid, lines, sw in (("lw", lw_grid, wl.lw_lines, False), ("sw", sw_grid, wl.sw_lines, True)):
        ref[band] = RC.band_column(kind, chk, orc, lib.Q, col, grid, lines, syn.tables(sw=sw), W.MOL_ORDER, MOLTAB,
                                   mol_mass, W.CIA_PAIRS, sw)
    want = np.concatenate([ref["lw"]["integ"], ref["sw"]["integ"]])
    # (the reference's one defect is not reproduced: a range check tripping inside sw_flux makes sw_fluxes_kernel store the
    # PREVIOUS wavenumber's fluxes, shortwave.c:318-320,:443 -- such points take the value under test in its integrals;
    # tests/test_gpu_parity_production.py does the same)
    (gcols, keep), _ = wl.columns(0, 1)
    wl.pipe.run(gcols)
    got = wl.pipe.fluxes(1)[0]
    L, V = W.NUM_LEVELS - 1, W.NUM_LEVELS
    r = ref["sw"]
    stale = np.zeros(r["nw"], dtype=bool)
    stale[1:] = np.all(r["up"][:, 1:] == r["up"][:, :-1], axis=0) & np.all(r["dn"][:, 1:] == r["dn"][:, :-1], axis=0)
    if stale.any():
        v = wl.pipe.views(1)
        up = api.device_to_host(device, v["flux_up"], (V, r["nw"]))
        dn = api.device_to_host(device, v["flux_down"], (V, r["nw"]))
        rows = [np.where(stale, mine, theirs) for mine, theirs in ((up[0], r["up"][0]), (up[-1], r["up"][-1]), (dn[0], r["dn"][0]), (dn[-1], r["dn"][-1]))]
        want = want.copy()
        want[[6, 7, 9, 10]] = [orc.integrate_row(np.ascontiguousarray(x), sw_grid[2]) for x in rows]
    out = {"what": "column 0, physically scaled line list (same grids, atmosphere and line counts), production form vs the reference's own C",
           "reference_fluxes_w_m2": {"rlut": float(want[0]), "rlds": float(want[4]), "rsdt": float(want[9]), "rsds": float(want[10])},
           "rsds_over_rsdt": float(want[10]/want[9]),
           "max_abs_flux_diff_w_m2": float(np.abs(got - want).max()), "tolerance_w_m2": 1e-3,
           "reference_stale_sw_points": int(stale.sum())}
    for bi, band in enumerate(("lw", "sw")):
        tau = api.device_to_host(device, wl.pipe.views(bi)["tau_gas"], (L, ref[band]["nw"]))
        out["tau_" + band] = tau_metrics(tau, ref[band]["tau_gas"])
    out["ok"] = bool(out["max_abs_flux_diff_w_m2"] <= out["tolerance_w_m2"])
    wl.destroy()
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


# What a rank needs in its environment before torch / HIP / RCCL exist in it, whoever started it -- spawn_ranks below or
# `python -m torch.distributed.run` (the driver): set by the rank ITSELF, first thing, so that the two launch paths cannot
# differ (tests/test_bench_launcher.py diffs them).  HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver only supports dmabuf
# IPC; without it RCCL's cross-process buffers fail with hipIpcGetMemHandle: invalid argument.
RANK_ENVIRONMENT = {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}


def rank_environment():
    for k, v in RANK_ENVIRONMENT.items():
        os.environ.setdefault(k, v)
    if os.environ.get("GRT_BENCH_DUMP_ENV"):            # test hook: what this rank will run under
        with open(os.environ["GRT_BENCH_DUMP_ENV"] + f".rank{os.environ.get('RANK', '0')}", "w") as f:
            json.dump({k: os.environ.get(k) for k in sorted(RANK_ENVIRONMENT)}, f)


def spawn_ranks(ngpus):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves, as plain child processes --
    the way the reference fans out its column shards (GRTworkflow/run-rfmip-irf.sh:103-132: one process per -x/-X block).
    This parent never imports torch and never touches HIP; the children are FRESH interpreters (no fork of a GPU
    process, no exec from one).  Rank 0's stdout (the one JSON line) is relayed; everything else goes to stderr.
    Any rank failing, or the job outliving GRT_BENCH_TIMEOUT seconds, ends the others (by their PIDs) and the exit
    code is non-zero."""
    import socket
    import subprocess
    import threading
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs, relay = [], []
    for r in range(ngpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(ngpus), LOCAL_WORLD_SIZE=str(ngpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GRT_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))

    def pump():
        for raw in procs[0].stdout:
            relay.append(raw.decode(errors="replace"))
    t = threading.Thread(target=pump, daemon=True)
    t.start()
    deadline = time.time() + float(os.environ.get("GRT_BENCH_TIMEOUT", 1500))
    rc = 0
    while rc == 0:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            rc = bad[0][1] if bad[0][1] > 0 else 1
            sys.stderr.write(f"bench.py: rank {bad[0][0]} exited with code {bad[0][1]}; stopping the other ranks\n")
        elif all(c == 0 for c in codes):
            break
        elif time.time() > deadline:
            rc = 124
            sys.stderr.write("bench.py: GRT_BENCH_TIMEOUT reached; stopping the ranks\n")
        else:
            time.sleep(0.05)
    if rc != 0:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    t.join(timeout=10)
    if rc == 0:
        sys.stdout.write("".join(relay))
        sys.stdout.flush()
    return rc


class PlaceholderEngine:
    """GRT_BENCH_REHEARSAL=1 on a box WITHOUT a GPU: no kernel runs and nothing is measured.  A column's "fluxes" are a
    function of its global index, so that the launcher, the sharding, the padded gather and rank 0's checks of what the
    gather delivered can be rehearsed (tests/test_bench_launcher.py).  The product path has no CPU form; this is not one."""

    def __init__(self, torch):
        self.torch = torch

    @staticmethod
    def expected(col, k):
        return 1000.0 * (col + 1) + k

    def run(self, first, n, dst):
        for i in range(n):
            for k in range(dst.shape[1]):
                dst[i, k] = self.expected(first + i, k)

    def sync(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)       # (a timed region of >= 3 s at ~103 ms per step)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cols", type=int, default=int(os.environ.get("GRT_BENCH_COLS", 64)), help="columns per GPU per step (weak scaling)")
    ap.add_argument("--chunk", type=int, default=int(os.environ.get("GRT_BENCH_CHUNK", 64)), help="columns per launch of the pipeline "
                    "(round 4, G1, one stream: 32 -> 457.5, 64 -> 455.1 columns/s; the roofline's counters were taken at 64)")
    ap.add_argument("--fast", type=int, default=int(os.environ.get("GRT_BENCH_FAST", 3)),
                    help="3: fused form, far wings by cell moments, two passes (production); 1: the same in one pass; "
                         "2: fused form, ring kernel; 0: reference operation order")
    ap.add_argument("--columns", type=int, default=0, help="strong scaling: a step is this fixed number of columns, sharded "
                    "over the ranks (0: weak scaling, --cols columns per GPU per step)")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("GRT_BENCH_LANES", 1)), help="batches in flight: launches alternate between this many "
                    "pipelines, each on a HIP stream of its own, so that the end of one launch -- its last workgroups, the far-field "
                    "gather, the solvers -- overlaps the next launch's line kernel (round 4, G1: one stream 455-457 columns/s, two "
                    "streams x 32 columns 463-467, three x 22: 467.8 -- profiles/r4_two_streams_bench_line.json).  Default 1: with more, "
                    "the kernels' durations overlap (kernel_ms_per_step sums to more than ms_per_step) and the roofline's "
                    "per-launch duration no longer is the kernel's own")
    ap.add_argument("--gather-every", type=int, default=0, help="steps between gathers of the output fluxes to rank 0 "
                    "(0: ONE gather, after the last step -- the job's output, as the north star words it)")
    ap.add_argument("--tile", type=int, default=0, help="exploration only: wavenumbers (cells) per workgroup of the line kernel")
    ap.add_argument("--lw-nslice", type=int, default=0, help="exploration only: line slices per tile of the longwave launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the reference-order and fine-grid measurements after the timed region")
    ap.add_argument("--lw-lines", type=int, default=None)
    ap.add_argument("--sw-lines", type=int, default=None)
    ap.add_argument("--lw-dw", type=float, default=None, help="exploration only: longwave grid spacing (default 1 cm-1)")
    ap.add_argument("--sw-dw", type=float, default=None, help="exploration only: shortwave grid spacing (default 1 cm-1)")
    args = ap.parse_args()
    if args.gpus < 1 or args.steps < 1 or args.cols < 1 or args.chunk < 1:
        raise SystemExit("bench.py: --gpus, --steps, --cols and --chunk must be positive")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us (the driver's N > 1 runs come through torch.distributed.run, which sets WORLD_SIZE):
        # start the ranks ourselves -- before torch or HIP exist in this process
        raise SystemExit(spawn_ranks(args.gpus))

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("GRT_BENCH_TEST_FAIL_RANK") == str(rank):
        raise SystemExit(7)             # test hook (tests/test_bench_launcher.py): a rank that dies before the rendezvous

    # the C library reports loaded species on stdout (like the reference's log_mesg); keep stdout
    # for the single JSON line by pointing fd 1 at stderr until the result is printed
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    rank_environment()
    import torch
    import torch.distributed as dist
    from grtcode_amd import multi

    # GRT_BENCH_REHEARSAL=1: a logic rehearsal of the N > 1 path on a box with fewer GPUs than ranks -- ranks share the
    # devices there are (none: PlaceholderEngine) and the collectives go over gloo on host copies.  Timings of such a run
    # mean nothing (the line says so); what it exercises is the launcher, the sharding, the line-list cache hand-over,
    # the padded gather and the barriers.
    rehearsal = os.environ.get("GRT_BENCH_REHEARSAL") == "1"
    ndev = torch.cuda.device_count()
    placeholder = rehearsal and ndev == 0
    if rehearsal and ndev > 0:
        local_rank = local_rank % ndev
    if not placeholder:
        torch.cuda.set_device(local_rank)
    # GRT_BENCH_FORCE_DIST=1 exercises the RCCL path (init, stream-ordered gather, max-reduce) at world size 1
    force_dist = os.environ.get("GRT_BENCH_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
    backend = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        backend = "gloo" if rehearsal else "nccl"
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()
    strong = args.columns > 0
    total_per_step = args.columns if strong else world * args.cols
    first, count = multi.shard(total_per_step, rank, world)         # weak scaling: args.cols columns per rank
    per = -(-total_per_step // world)
    chunk = min(args.chunk, max(per, 1))
    lw_grid = sw_grid = None
    lanes, wls = 1, []
    if placeholder:
        engine = PlaceholderEngine(torch)
        dev_t = torch.device("cpu")
        chunks = [(lo, min(chunk, count - lo)) for lo in range(0, count, chunk)]
    else:
        from grtcode_amd import api, workload as W
        device = api.create_device(local_rank)
        dev_t = torch.device("cuda", local_rank)
        if world > 1:
            # the line lists are drawn once (rank 0) and shared through a cache directory instead of once per rank
            os.environ.setdefault("GRT_LINES_CACHE", os.path.join(os.environ.get("TMPDIR", "/tmp"), f"grt_lines_cache_{os.getuid()}"))
            if rank != 0:
                dist.barrier()
        lw_grid = (W.LW_GRID[0], W.LW_GRID[1], args.lw_dw or W.LW_GRID[2])
        sw_grid = (W.SW_GRID[0], W.SW_GRID[1], args.sw_dw or W.SW_GRID[2])
        # --lanes K > 1: K pipelines, each with gas-optics objects and buffers of its own on a stream of its own; launches
        # alternate between them, so that the end of one launch's work (far-field gather, solvers) can overlap the next
        # launch's line kernel.  Same work per step; nothing is shared between the lanes.  Default 1 (two lanes: +2 %, but overlapping kernel durations).
        lanes = max(1, min(args.lanes, 4, -(-max(count, 1) // chunk)))
        if lanes < args.lanes and rank == 0:
            # (a step of `count` columns is ceil(count/chunk) launches: that many can be in flight.  Round 5's first
            # "two streams" line had asked for two with one launch per step and timed one: say so)
            print(f"bench.py: --lanes {args.lanes} asked for, {lanes} used: a step of {count} columns is {-(-max(count, 1) // chunk)} "
                  f"launch(es) of {chunk}; give --cols {args.lanes * chunk} --chunk {chunk} for {args.lanes} launches in flight", file=sys.stderr)
        os.environ.setdefault("GRT_LINES_CACHE", os.path.join(os.environ.get("TMPDIR", "/tmp"), f"grt_lines_cache_{os.getuid()}"))
        wls = []
        for k in range(lanes):
            api.use_lane(device, k)
            wls.append(W.G1Workload(device, chunk, lw_lines=args.lw_lines or W.LW_LINES, sw_lines=args.sw_lines or W.SW_LINES,
                                    fast=args.fast, lw_grid=lw_grid, sw_grid=sw_grid, tile=args.tile, lw_nslice=args.lw_nslice))
        api.use_lane(device, 0)
        wl = wls[0]
        if world > 1 and rank == 0:
            dist.barrier()                                              # the lists are in the cache: the other ranks may build
        # this rank's block, in chunks of at most `chunk` columns
        chunks = []
        for lo in range(0, count, chunk):
            n = min(chunk, count - lo)
            (gc, keep), _ = wl.columns(first + lo, n)
            chunks.append((lo, n, gc, keep))
    # The job's output: every step's [per][12] block of integrated fluxes stays on the device, in the step's slot of a
    # job buffer, and the buffer is gathered to rank 0 ONCE, after the last step (--gather-every K: after every K steps).
    nfl = 12
    gather_every = args.gather_every if args.gather_every > 0 else args.steps
    slots = min(gather_every, args.steps)
    job = torch.zeros(slots, max(per, 1), nfl, dtype=torch.float64, device=dev_t)
    gathered = [torch.zeros_like(job) for _ in range(world)] if (use_dist and rank == 0) else None
    stream = None if placeholder else torch.cuda.ExternalStream(wl.pipe.stream(), device=dev_t)
    if not placeholder:
        torch.cuda.synchronize()        # (the zero fills ran on torch's stream; the library's streams are non-blocking: order them once)
    row_bytes = 8 * nfl
    gathers = {"count": 0}
    gathered_host = [torch.zeros(slots, max(per, 1), nfl, dtype=torch.float64) for _ in range(world)] if (use_dist and rank == 0 and rehearsal) else None

    def step(s):
        block = job[s % slots]
        if placeholder:
            for lo, n in chunks:
                engine.run(first + lo, n, block[lo: lo + n])
        else:
            for i, (lo, n, gc, _) in enumerate(chunks):
                k = (s * len(chunks) + i) % lanes
                if lanes > 1:
                    api.use_lane(device, k)
                wls[k].pipe.run(gc, block.data_ptr() + lo * row_bytes)
            if lanes > 1:
                api.use_lane(device, 0)

    def gather():
        """the job buffer -> rank 0: one collective (blocks are padded to `per` rows, so no sizes travel)"""
        if not use_dist:
            return
        gathers["count"] += 1
        if rehearsal:
            if not placeholder:
                api.device_synchronize(device)
            dist.gather(job.cpu(), gathered_host if rank == 0 else None, dst=0)
        else:
            if lanes > 1:
                api.device_synchronize(device)   # (one wait per gather, i.e. per job: every lane has delivered)
            with torch.cuda.stream(stream):      # RCCL gather ordered after the kernels on the library stream, no host sync
                dist.gather(job, gathered if rank == 0 else None, dst=0)
            if lanes > 1:
                # the gather reads `job` on lane 0's stream; the next step's launches on the other lanes write its slots and
                # are not ordered after it: wait here (several lanes AND several gathers per job is an exploration mode)
                stream.synchronize()

    def barrier():
        if not placeholder:
            api.device_synchronize(device)
            torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    for s in range(args.warmup):
        step(s)
    if args.warmup:
        gather()                # (the communicator's first collective is not part of the job)
    barrier()
    gathers["count"] = 0
    if not placeholder:
        api.profile_enable(True)
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(s)
        if (s + 1) % gather_every == 0 or s + 1 == args.steps:
            gather()
    barrier()
    elapsed = time.perf_counter() - t0
    own_elapsed = elapsed
    elapsed = multi.max_over_ranks(elapsed, torch.device("cpu") if rehearsal else dev_t)
    # every rank's own time, for the line (load balance of the shards): one more tiny gather, after the timed region
    rank_ms = [1e3 * own_elapsed / args.steps]
    if use_dist:
        mine = torch.tensor([rank_ms[0]], dtype=torch.float64, device=torch.device("cpu") if rehearsal else dev_t)
        every = [torch.zeros_like(mine) for _ in range(world)] if rank == 0 else None
        dist.gather(mine, every, dst=0)
        if rank == 0:
            rank_ms = [float(t.item()) for t in every]

    if rank == 0:
        fluxes = job[(args.steps - 1) % slots].cpu().numpy()           # this rank's block of the last step
        assert np.all(np.isfinite(fluxes)), "non-finite integrated fluxes"
        if use_dist:
            # what the gathers delivered: [rank][slot][per][12]; a slot's blocks laid end to end are that step's columns in order
            got = torch.stack([g.cpu() for g in (gathered_host if rehearsal else gathered)]).numpy()
            allf = got.transpose(1, 0, 2, 3).reshape(slots, world * max(per, 1), nfl)[:, :total_per_step]
            assert np.all(np.isfinite(allf)) and np.all(allf[:, :, 0] > 0.), "gathered fluxes incomplete"
            assert np.array_equal(allf[:, :count], job.cpu().numpy()[:, :count]), "rank 0's own block changed in the gather"
            if placeholder:
                want = np.array([[PlaceholderEngine.expected(c, k) for k in range(nfl)] for c in range(total_per_step)])
                assert all(np.array_equal(allf[sl], want) for sl in range(slots)), "gathered blocks are not the columns in order"
        total_cols = total_per_step * args.steps
        dist_info = {"n_gpus": world, "rccl_ranks": world if backend == "nccl" else 0, "ms_per_step_by_rank": rank_ms,
                     "collective": {"backend": backend, "gathers_in_timed_region": gathers["count"],
                                    "bytes_per_rank_per_gather": int(job.numel()) * 8 if use_dist else 0,
                                    "what": "one gather of the job's [steps][columns][12] integrated fluxes to rank 0" if use_dist else None}}
        if placeholder:
            line = {"metric": "columns/sec (60-layer, 1 cm\u207b\u00b9 LBL, LW+SW)", "value": total_cols / elapsed, "unit": "columns/s",
                    **dist_info, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
                    "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64",
                    "data": "synthetic",
                    "rehearsal": "NO DEVICE: placeholder flux blocks, no kernel ran, nothing was measured -- launcher, sharding and gather only",
                    "config": {"workload": "none (rehearsal)", "columns_per_step": total_per_step, "chunk_columns": chunk,
                               "shards": [list(multi.shard(total_per_step, r, world)) for r in range(world)]}}
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            print(json.dumps(line), flush=True)
            os.dup2(2, 1)
    if placeholder:
        if use_dist:
            dist.destroy_process_group()
        return
    if rank == 0:
        ms = {tag: api.profile_read(tag) for tag in (1, 2, 3, 4, 5, 6, 7)}
        L, V = W.NUM_LEVELS - 1, W.NUM_LEVELS
        n_lw, n_sw = wl.grid_lw.n, wl.grid_sw.n
        S = wl.total_lines
        # algorithmic bytes of the dominant kernel (line-by-line tau, SW-band launch), per launch:
        # SURVEY.md §8(d) terms it owns: 60 B/line once per column + per wavenumber 8*C_tab table
        # reads + 8*L tau written once (C_tab = 4 H2O + 1 O3 + 2 CFC + 3 CIA = 10 tables)
        cols_launch = count / max(len(chunks), 1)          # columns per line-kernel launch on this rank (= --chunk)
        bytes_gas = lambda nlines, n: cols_launch * (60.0 * nlines + n * (8.0 * 10 + 8.0 * L))
        dom_ms = ms[2][0] / max(ms[2][1], 1)
        achieved = bytes_gas(S["sw"], n_sw) / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        fsteps = int(np.ceil(25.0 / sw_grid[2]))
        points = lambda nlines: float(L) * nlines * (2 * fsteps + 1) * cols_launch    # Voigt evaluations per launch
        valu_flop = 12.0 * points(S["sw"])                                            # ~12 flop far-wing point
        # HBM-side bytes per launch are PMC counters (FETCH_SIZE, WRITE_SIZE: separate rocprofv3 --pmc passes of this same
        # command, scripts/profile_round.sh) -- they cannot be read inside an ordinary run, so the figure of the last
        # profiled run is carried here, labelled as such, and dropped when the configuration differs
        traffic, traffic_src, solver_traffic, sq = None, None, {}, {}
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                import hashlib
                live_sha = hashlib.sha256(b"".join(open(os.path.join(ROOT, "grtcode_amd", "csrc", "hip", f), "rb").read()
                                                   for f in ("k_gas_optics_mp.hip", "mp_general_block.inc", "mp_lean_block.inc", "k_gas_optics_far.hip", "gas_optics_mp_dev.h", "gas_optics_dev.h"))).hexdigest()
                if tj.get("kernel_source_sha256") != live_sha:
                    # (counters of another version of the kernel: not carried -- ADVICE r4)
                    traffic_src = (f"profiles/traffic_latest.json (round tag {tj.get('tag')}) was taken on another version of the "
                                   "kernel source: its counters are not carried into this line")
                elif tj.get("cols") == round(cols_launch) and tj.get("fast") == args.fast and (lw_grid[2], sw_grid[2]) == (1.0, 1.0) \
                        and args.lw_lines is None and args.sw_lines is None:
                    traffic = tj["gas_optics_sw"]["hbm_bytes_per_launch"]
                    traffic_src = f"profiles/traffic_latest.json: rocprofv3 --pmc passes of this command, round tag {tj.get('tag')} (not measured in this run)"
                    solver_traffic = {k: v.get("hbm_bytes_per_launch") for k, v in tj.get("solvers", {}).items()}
                    sq = tj["gas_optics_sw"].get("sq", {})
            except Exception:
                traffic = None
        lean_on = args.fast == 3 and os.environ.get("GRT_LEAN", "1") != "0" and (lw_grid[2], sw_grid[2]) == (1.0, 1.0)
        line_kernel = ("gas_optics_lean_kernel" if lean_on else "gas_optics_mp_kernel") if args.fast in (1, 3) else "gas_optics_kernel"
        line = {
            "metric": "columns/sec (60-layer, 1 cm\u207b\u00b9 LBL, LW+SW)" if (lw_grid[2], sw_grid[2]) == (1.0, 1.0) else
                      f"columns/sec (60-layer LBL, LW @{lw_grid[2]:g} + SW @{sw_grid[2]:g} cm\u207b\u00b9)",
            "value": total_cols / elapsed, "unit": "columns/s",
            **dist_info, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            **({"rehearsal": "ranks share devices, gloo on host copies: timings are meaningless"} if rehearsal else {}), "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("SURVEY §8d grid G1: LW 1-3250 + SW 1-50000 cm-1 @1 cm-1" if (lw_grid[2], sw_grid[2]) == (1.0, 1.0) else
                                    f"LW {lw_grid[0]:g}-{lw_grid[1]:g} cm-1 @{lw_grid[2]:g} cm-1 (n = {n_lw}) + SW {sw_grid[0]:g}-{sw_grid[1]:g} cm-1 "
                                    f"@{sw_grid[2]:g} cm-1 (n = {n_sw}) -- NOT the headline grid") +
                                   f", 60 layers, 7 absorbers, {S['lw']} LW + {S['sw']} SW synthetic lines, H2O/O3 continua, 2 CFC, 3 CIA, "
                                   "clear sky, integrated fluxes",
                       "columns_per_gpu_per_step": args.cols if not strong else None,
                       "columns_per_step": total_per_step, "chunk_columns": chunk, "launches_in_flight": lanes, "fast": args.fast,
                       "arithmetic": {0: "reference operation order", 1: "fused form, far wings by cell moments",
                                      2: "fused form, every window point in the ring",
                                      3: "fused form, far wings by cell moments, two passes"}.get(args.fast, str(args.fast)),
                       "parallelism": f"columns sharded over {world} GPU(s), one process per GPU; one RCCL gather of the job's 12 fluxes/column to rank 0"},
            # The dominant kernel is bound by the vector pipe's ISSUE rate, not by HBM or MFMA (no dense contraction on this
            # path): `roofline` prices it against what binds it -- VALU wave64 instructions per second over the chip's issue
            # slots (1 024 SIMDs, one such instruction per two cycles at best; instruction count from the PMC pass of this
            # same command, launch duration measured live in this run) -- and `roofline_hbm` keeps the HBM view with SURVEY
            # §8(d)'s algorithmic bytes (task 2).
            "roofline": {"kernel": f"{line_kernel} (line-by-line tau), SW-band launch", "bound": "valu_issue",
                         "achieved": (sq["SQ_INSTS_VALU"] / (dom_ms * 1e-3) / 1e9) if sq.get("SQ_INSTS_VALU") and dom_ms > 0 else None,
                         "peak": SIMDS * CLOCK_HZ * 0.5 / 1e9, "unit": "G wave64 VALU instructions/s",
                         "frac": (sq["SQ_INSTS_VALU"] / (SIMDS * dom_ms * 1e-3 * CLOCK_HZ * 0.5)) if sq.get("SQ_INSTS_VALU") and dom_ms > 0 else None,
                         "algorithmic_fp32_frac": (valu_flop / (dom_ms * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS) if dom_ms > 0 else None,
                         "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": dom_ms, "launches": ms[2][1],
                         "instructions_source": traffic_src,
                         "note": "measured on gfx950 (scripts/valu_mix*.hip, profiles/r4_valu_mix*.txt): at this kernel's occupancy a wave64 "
                                 "fp32 fma/mul/add occupies the pipe ~2.5-4 cycles, every fp64 op, conversion, compare, select and DPP "
                                 "move ~4.6, rcp/exp/sqrt ~10 -- so the two-cycle issue peak is not reachable with this instruction mix; "
                                 "SQ_ACTIVE_INST_VALU of the PMC pass puts the pipe at >90 % busy.  The lean loop's fp32 arithmetic "
                                 "runs as packed instructions (v_pk_fma/mul/add_f32: two lines per instruction, 4.6 cycles), which "
                                 "this count takes as one: fewer, longer instructions lower `frac` while the launch gets shorter -- "
                                 "instructions_per_64_lines and avg_launch_ms are the figures to follow across rounds"},
            "roofline_hbm": {"kernel": f"{line_kernel} (line-by-line tau), SW-band launch", "bound": "hbm",
                             "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                             "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": dom_ms, "launches": ms[2][1],
                             "note": "this kernel is VALU-issue bound by construction, see roofline"},
            "roofline_valu": {"kernel": f"{line_kernel}, SW-band launch", "bound": "valu_fp32",
                              "achieved": valu_flop / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0,
                              "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s (algorithmic: 12 flop x L*S*F line-shape points, SURVEY §8d; the moment kernel "
                                      "delivers the far-wing points without evaluating them one by one)",
                              "frac": (valu_flop / (dom_ms * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS) if dom_ms > 0 else 0.0,
                              "voigt_points_per_launch": points(S["sw"]),
                              "gpoints_per_s": points(S["sw"]) / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0,
                              # What the vector pipe actually does (the algorithmic rate above is speed-up-shaped: far-wing
                              # points are delivered by the moment series, not evaluated): SQ_INSTS_VALU of the profiled run
                              # of this command over the issue slots of this run's launch -- 1 024 SIMDs, one wave64
                              # instruction per two cycles at best
                              "issue_utilisation": (sq["SQ_INSTS_VALU"] / (SIMDS * dom_ms * 1e-3 * CLOCK_HZ * 0.5)) if sq.get("SQ_INSTS_VALU") and dom_ms > 0 else None,
                              "valu_wave_instructions_per_launch": sq.get("SQ_INSTS_VALU"),
                              "instructions_per_64_lines": (sq["SQ_INSTS_VALU"] / (cols_launch * L * S["sw"] / 64.0)) if sq.get("SQ_INSTS_VALU") else None,
                              "utilisation_source": traffic_src},
            "kernel_ms_per_step": {"gas_optics_lw": ms[1][0] / args.steps, "gas_optics_sw": ms[2][0] / args.steps,
                                   "far_field_lw": ms[6][0] / args.steps, "far_field_sw": ms[7][0] / args.steps,
                                   "lw_solver": ms[3][0] / args.steps, "sw_solver": ms[4][0] / args.steps,
                                   "clear_sky_optics": ms[5][0] / args.steps},
            # the LONGWAVE column-band against HBM with SURVEY §8(d)'s algorithmic bytes B_band (line store once, 11 tables,
            # tau/omega/g written once and read once, fluxes out) -- on the ~3M-point grid (--lw-dw 0.001) this is the
            # north-star's "fraction of HBM roofline": the far-field gather there is gas_optics_tree_kernel (+ the coarse levels)
            "roofline_hbm_lw_band": (lambda b_lw, ms1, ms6, ms3: {
                "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes_per_column": b_lw,
                "far_field_kernel": "gas_optics_tree_kernel (cell hierarchy)" if lw_grid[2] < 0.12 else "gas_optics_far_kernel",
                "columns_per_step_on_this_rank": count,
                "far_field_ms_per_step": ms6, "first_pass_ms_per_step": ms1, "solver_ms_per_step": ms3,
                "achieved_far_field": b_lw * count / (ms6 * 1e-3) / 1e9 if ms6 > 0 else None,
                "frac_far_field": b_lw * count / (ms6 * 1e-3) / 1e9 / HBM_PEAK_GBS if ms6 > 0 else None,
                "achieved_band": b_lw * count / ((ms1 + ms6 + ms3) * 1e-3) / 1e9 if ms1 + ms6 + ms3 > 0 else None,
                "frac_band": b_lw * count / ((ms1 + ms6 + ms3) * 1e-3) / 1e9 / HBM_PEAK_GBS if ms1 + ms6 + ms3 > 0 else None,
                "note": "frac_far_field prices the gather ALONE against the whole band's algorithmic bytes (VERDICT r4, task 3); "
                        "frac_band the band's three kernels together; times are the kernels' HIP-event durations per step"})(
                60.0 * S["lw"] + n_lw * (8.0 * 11 + 24.0 * L + 24.0 * L + 16.0 * V),
                ms[1][0] / args.steps, ms[6][0] / args.steps, ms[3][0] / args.steps),
            "sample_fluxes_col0": {"rlut": fluxes[0, 0], "rlus": fluxes[0, 1], "rlds": fluxes[0, 4],
                                   "rsut": fluxes[0, 6], "rsdt": fluxes[0, 9], "rsds": fluxes[0, 10]},
        }
        # the solvers (fused form: Rayleigh + optics combination + solver + spectral trapezoid in one kernel).  HBM view
        # with SURVEY a19/a20's algorithmic bytes of the materialised interface (1 944 / 2 440 B per wavenumber) and with
        # what the fused kernels must move (tau_gas once + the per-wavenumber tables); the shortwave kernel's arithmetic
        # (two delta-Eddington solutions per layer: ~13 divisions, 6 exp in fp64) is done once, in ONE sweep from the top
        # (round 4: the slab's direct-beam reflectance and upward transmission ride along, nothing is parked; with
        # GRT_SW_TWO_SWEEPS=1, or a user level between top and surface, the first of two sweeps parks the five properties
        # of every layer for the second: `traffic_with_park`)
        sol = {}
        for name, tag, n, surv, fused_b in (("lw", 3, n_lw, 16.0 * L + 8 + 16.0 * V, 8.0 * L + 8), ("sw", 4, n_sw, 24.0 * L + 24 + 16.0 * V, 8.0 * L + 24)):
            if ms[tag][1]:
                t = ms[tag][0] / ms[tag][1] * 1e-3
                sol[name] = {"avg_launch_ms": t * 1e3, "columns_per_launch": cols_launch,
                             "algorithmic_bytes_survey": surv * n * cols_launch, "achieved_gb_per_s_survey": surv * n * cols_launch / t / 1e9,
                             "frac_hbm_survey": surv * n * cols_launch / t / 1e9 / HBM_PEAK_GBS,
                             "compulsory_bytes_fused": fused_b * n * cols_launch, "achieved_gb_per_s_fused": fused_b * n * cols_launch / t / 1e9,
                             "traffic": solver_traffic.get(name + "_kernel")}
                if name == "sw" and os.environ.get("GRT_SW_TWO_SWEEPS", "0") == "1":
                    # tau_gas once, albedo + solar, layer properties (5 L rows) written and read back
                    park_b = (8.0 * L + 24 + 2 * 8.0 * (5 * L)) * n * cols_launch
                    sol[name].update({"traffic_with_park": park_b, "achieved_gb_per_s_with_park": park_b / t / 1e9,
                                      "frac_hbm_with_park": park_b / t / 1e9 / HBM_PEAK_GBS})
        line["roofline_solvers"] = dict(sol, bound="fp64 arithmetic (shortwave: two delta-Eddington solutions per layer and wavenumber in one "
                                                   "sweep, tau_gas read once); latency (longwave: 26 000 threads)", peak=HBM_PEAK_GBS, unit="GB/s")
        if world == 1 and not args.no_extras:
            # what an unchanged caller of calculate_optical_depth gets: the reference-order form (fast = 0)
            wl.go_lw.tune(fast=0, tile=args.tile, nslice=args.lw_nslice)
            wl.go_sw.tune(fast=0, tile=args.tile)
            lo, n, gc, _ = chunks[0]
            wl.pipe.run(gc, job[0].data_ptr())
            barrier()
            t0 = time.perf_counter()
            for _ in range(2):
                wl.pipe.run(gc, job[0].data_ptr())
            barrier()
            line["reference_order_columns_per_s"] = 2 * n / (time.perf_counter() - t0)
            line["reference_abi"] = reference_abi_rate(wl)
            wl.go_lw.tune(fast=args.fast, tile=args.tile, nslice=args.lw_nslice)
            wl.go_sw.tune(fast=args.fast, tile=args.tile)
            if args.lw_dw is None and args.lw_lines is None:
                line["fine_grid"] = {"G2_lw_0.1cm-1": fine_grid_column(wl, 0.1, 4),
                                     "G3_lw_0.001cm-1": fine_grid_column(wl, 0.001, 2, compare_layers=12)}
        if world == 1 and not args.no_cpu_baseline:
            # the production form's column 0 back in the pipeline's buffers and in job[0] (parity below)
            wl.pipe.run(chunks[0][2], job[0].data_ptr())
            barrier()
            fluxes = job[0].cpu().numpy()
            line["cpu_baseline"], ref_bands = cpu_baseline(lw_grid, sw_grid, args.lw_lines or W.LW_LINES,
                                                           args.sw_lines or W.SW_LINES)
            full = all(v["thin"] == 1 for v in line["cpu_baseline"]["detail"].values())
            line["parity"] = parity_of_column0(wl, fluxes, ref_bands, line["cpu_baseline"]["kind"]) if full else None
            if line["parity"] is not None and (lw_grid[2], sw_grid[2]) == (1.0, 1.0) and args.lw_lines is None and args.sw_lines is None:
                line["parity"]["physical_list"] = parity_physical_list(device, args.fast, lw_grid, sw_grid)
                if line["parity"]["physical_list"] is not None:
                    line["parity"]["ok"] = bool(line["parity"]["ok"] and line["parity"]["physical_list"]["ok"])
            if line["parity"] is not None and not line["parity"]["ok"]:
                sys.stderr.write("bench.py: PARITY FAILURE, no result line: " + json.dumps(line["parity"]) + "\n")
                for w in wls:
                    w.destroy()
                raise SystemExit(3)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    for w in wls:
        w.destroy()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
