"""§8(f)-1, data-adapter side: examples/rfmip_batch_driver.c -- RFMIP-style columns (Pa, layer mole fractions for
H2O/O3, global means for the rest, zenith angle in degrees; rfmip-irf/src/rfmip-irf.c:175-325) from a flat dump,
through the batched device pipeline from plain C linked against the static archives -- against the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from grtcode_amd import api, synthetic as syn
from scenario import Band
from test_gpu_pipeline import oracle_column

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "grtcode_amd", "lib")
ARCHIVES = ["-lgrtcode_hip_ext", "-lshortwave", "-llongwave", "-lgas_optics", "-lgrtcode_utilities"]
GM = {syn.CO2: 4.0e-4, syn.CH4: 1.8e-6, syn.N2O: 3.3e-7, syn.CO: 1.0e-7, syn.O2: 0.209}     # mole fractions


def rfmip_like_columns(n, V):
    cols, raw = [], []
    for c in range(n):
        base = syn.profile(c, V)
        p_pa = base["p"] * 100.0
        play_pa = 0.5 * (p_pa[:-1] + p_pa[1:])
        h2o_lay = 0.5e-6 * (base["ppmv"][syn.H2O][:-1] + base["ppmv"][syn.H2O][1:])
        o3_lay = 0.5e-6 * (base["ppmv"][syn.O3][:-1] + base["ppmv"][syn.O3][1:])
        sza = [20.0, 55.0, 100.0, 70.0, 0.0][c % 5]                  # column 2 of every five is a night column
        tsi, emis, alb = 1360.0, 0.97, 0.12
        raw.append(np.concatenate([p_pa, play_pa, base["t"], base["t_layer"],
                                   [base["t_surf"], emis, alb, sza, tsi], h2o_lay, o3_lay]))
        # what the driver must make of it (rfmip-irf.c:186,295-308,318-325)
        p, pl = p_pa * 0.01, play_pa * 0.01
        L = V - 1

        def to_levels(ab):
            out = np.zeros(V)
            out[0], out[L] = ab[0] * 1e6, ab[L - 1] * 1e6
            for k in range(1, L):
                out[k] = 1e6 * (ab[k - 1] + (ab[k] - ab[k - 1]) * (p[k] - pl[k - 1]) / (pl[k] - pl[k - 1]))
            return out
        ppmv = {m: np.full(V, x * 1e6) for m, x in GM.items()}
        ppmv[syn.H2O], ppmv[syn.O3] = to_levels(h2o_lay), to_levels(o3_lay)
        ppmv[syn.N2] = np.full(V, 0.781e6)
        cols.append(dict(p=p, t=base["t"], t_layer=base["t_layer"], t_surf=base["t_surf"], ppmv=ppmv,
                         mu0=float(np.cos(2.0 * np.pi * sza / 360.0)), tsi=tsi,
                         cfc_ppmv={0: np.full(V, 2.3e-4), 1: np.full(V, 5.2e-4)}, emis=emis, alb=alb))
    return cols, np.concatenate(raw)


def run_driver(args, env=None, timeout=600):
    """Run the C driver; if it dies on a signal, run it once more under rocgdb (where there is one) so that the failure
    report carries the backtrace."""
    r = subprocess.run(args, capture_output=True, text=True, timeout=timeout, env=env)
    if r.returncode < 0:
        import shutil
        gdb = shutil.which("rocgdb") or shutil.which("gdb")
        bt = ""
        if gdb:
            g = subprocess.run([gdb, "-batch", "-ex", "run", "-ex", "bt", "-ex", "info registers rip", "--args", *args],
                               capture_output=True, text=True, timeout=timeout, env=env)
            bt = "\n--- under " + gdb + " ---\n" + g.stdout[-6000:] + g.stderr[-2000:]
        raise AssertionError(f"driver died on signal {-r.returncode}\n{r.stderr[-1500:]}{bt}")
    return r


def test_batched_c_driver_matches_oracle(tmp_path, oracle, lib):
    V, ncol = 13, 7
    cols, raw = rfmip_like_columns(ncol, V)
    swb = Band(str(tmp_path / "data"), 1.0, 6000.0, 2.0, 8000, sw=True)
    lwb = Band(str(tmp_path / "lw_view"), 1.0, 2000.0, 1.0, 0, sw=True)
    lwb.par, lwb.h2o_dir, lwb.files, lwb.tab = swb.par, swb.h2o_dir, swb.files, swb.tab
    lwb.lines = {m: {k: a[(ln["v0"] >= lwb.w0) & (ln["v0"] <= lwb.wn)] for k, a in ln.items()}
                 for m, ln in swb.lines.items()}
    dump = str(tmp_path / "columns.bin")
    with open(dump, "wb") as f:
        f.write(struct.pack("<iii", 0x47525443, ncol, V))
        f.write(np.array([GM[syn.CO2], GM[syn.CH4], GM[syn.N2O], GM[syn.CO], GM[syn.O2]]).tobytes())
        f.write(raw.astype("<f8").tobytes())
    exe = str(tmp_path / "rfmip_batch_driver")
    r = subprocess.run(["gcc", "-std=gnu99", "-O2", "-g", "-Wall", "-DGRT_BACKTRACE", "-rdynamic", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "rfmip_batch_driver.c"), "-L" + LIBDIR, *ARCHIVES,
                        "-L/opt/rocm/lib", "-lamdhip64", "-lstdc++", "-lm", "-Wl,-rpath,/opt/rocm/lib", "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    args = [exe, swb.par, swb.files["solar"], dump, "-h2o-ctm", swb.h2o_dir, "-o3-ctm", swb.files["o3_ctm"],
            "-CFC-11", swb.files["cfc11"], "2.3e-4", "-CFC-12", swb.files["cfc12"], "5.2e-4",
            "-N2-N2", swb.files["cia_n2n2"], "-O2-N2", swb.files["cia_o2n2"], "-O2-O2", swb.files["cia_o2o2"],
            "-w-lw", "1", "-W-lw", "2000", "-r-lw", "1", "-w-sw", "1", "-W-sw", "6000", "-r-sw", "2",
            "-chunk", "3", "-fast", "3"]
    r = run_driver(args)
    assert r.returncode == 0, r.stderr[-3000:]
    # soak (GRT_TEST_REPEAT_DRIVER=N): a crash on the way in or out shows its backtrace; in the deterministic mode the
    # whole C driver's output repeats to the last digit
    first = None
    for _ in range(int(os.environ.get("GRT_TEST_REPEAT_DRIVER", 2))):
        again = run_driver(args, env=dict(os.environ, GRT_DETERMINISTIC="1"))
        assert again.returncode == 0, again.stderr[-3000:]
        first = first or again.stdout
        assert again.stdout == first
    got = {}
    for line in r.stdout.splitlines():
        if line.startswith("col "):
            head, vals = line.split(":")
            got[int(head.split()[1])] = np.array([float(x) for x in vals.split()])
    assert sorted(got) == list(range(ncol))
    grid_sw = api.create_spectral_grid(swb.w0, swb.wn, swb.dw)
    solar = api.create_solar_flux(grid_sw, swb.files["solar"])
    worst = 0.0
    for c, col in enumerate(cols):
        emis, alb = np.full(lwb.nw, col["emis"]), np.full(swb.nw, col["alb"])
        w = oracle_column(oracle, lib, lwb, col, True, emis)
        worst = max(worst, np.max(np.abs(got[c][:4] - w["integ"][[0, 1, 3, 4]])))
        if col["mu0"] > 0:
            w = oracle_column(oracle, lib, swb, col, False, emis, alb, solar)
            worst = max(worst, np.max(np.abs(got[c][4:] - w["integ"][[0, 1, 3, 4]])))
        else:
            assert np.all(got[c][4:] == 0.0)                          # night: no shortwave (driver.c:706)
    print(f"batched C driver, {ncol} columns: worst flux difference {worst:.2e} W m-2")
    assert worst < 1e-4

    # ---- the same run as THREE ranks (one process each, sharing this GPU), 7 columns -> blocks of 3, 3, 1, flux blocks
    # gathered to rank 0 through the library's C entry points grt_multi_* (SURVEY §8e).  File transport here: RCCL wants
    # one GPU per rank (below: RCCL at world size 1).
    rdv = tmp_path / "rdv"
    rdv.mkdir()
    procs = [subprocess.Popen(args + ["-ranks", "3", "-rank", str(k), "-rendezvous", str(rdv), "-transport", "files"],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(os.environ, GRT_MULTI_TIMEOUT="300")) for k in range(3)]
    outs = [p.communicate(timeout=900) for p in procs]
    for k, p in enumerate(procs):
        assert p.returncode == 0, (k, outs[k][1][-2000:])
    sharded = {}
    for line in outs[0][0].splitlines():
        if line.startswith("col "):
            head, vals = line.split(":")
            sharded[int(head.split()[1])] = np.array([float(x) for x in vals.split()])
    assert sorted(sharded) == list(range(ncol))
    assert not any(l.startswith("col ") for k in (1, 2) for l in outs[k][0].splitlines())     # only rank 0 reports
    for c in range(ncol):
        assert np.max(np.abs(sharded[c] - got[c])) < 1e-7          # a column's fluxes do not depend on its shard (atomics: ~1e-9)

    # ---- RCCL transport, world size 1: communicator through the rendezvous directory, ncclGather on the library stream
    rdv1 = tmp_path / "rdv1"
    rdv1.mkdir()
    r = subprocess.run(args + ["-ranks", "1", "-rank", "0", "-rendezvous", str(rdv1)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]                      # (-ranks 1: no communicator is made)
    from grtcode_amd import multi
    import ctypes as C
    m = multi.Multi(multi.RCCL, 0, 0, 1, str(rdv1))
    assert (rdv1 / "rccl_unique_id.bin").stat().st_size == 128
    block = np.arange(5 * 12, dtype=np.float64).reshape(5, 12)
    src, dst = api.DeviceBuffer(0, block.nbytes), api.DeviceBuffer(0, block.nbytes)
    api.check(lib.grt_host_to_device(0, src.ptr, block.ctypes.data_as(C.c_void_p), block.nbytes))
    m.gather_fluxes(src.ptr.value, 5, dst.ptr.value, True)
    assert m.max(3.25) == 3.25                                       # (synchronises the stream)
    assert np.array_equal(dst.to_host((5, 12)), block)
    m.destroy()
    assert not (rdv1 / "rccl_unique_id.bin").exists()               # rank 0 clears the id: the directory can be reused
    src.free()
    dst.free()
