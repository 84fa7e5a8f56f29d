"""The benchmark workload of SURVEY.md §8(d): grids G1 = longwave 1-3250 cm-1 @ 1 cm-1
(n = 3 250) + shortwave 1-50 000 cm-1 @ 1 cm-1 (n = 50 000), 60 layers, 7 absorbers with
1.0 M (LW) / 1.5 M (SW) synthetic lines, H2O + O3 continua, 2 CFCs, 3 CIA pairs.
Line lists go in through grt_add_molecule_lines (same arithmetic as the .par reader);
tables go in through the reference's CSV loaders."""
import os
import tempfile

import numpy as np

from . import api, synthetic as syn

MOL_ORDER = [syn.H2O, syn.CO2, syn.O3, syn.N2O, syn.CO, syn.CH4, syn.O2]
CIA_PAIRS = [(0, 0, "cia_n2n2"), (1, 0, "cia_o2n2"), (1, 1, "cia_o2o2")]
LW_GRID = (1.0, 3250.0, 1.0)
SW_GRID = (1.0, 50000.0, 1.0)
LW_LINES = 1_000_000
SW_LINES = 1_500_000
NUM_LEVELS = 61


def write_tables(root, sw):
    t = syn.tables(sw=sw)
    d = os.path.join(root, "h2o_ctm_sw" if sw else "h2o_ctm_lw")
    os.makedirs(d, exist_ok=True)
    syn.write_csv(os.path.join(d, "296MTCKD25_F.csv"), *t["h2o_foreign_296"])
    syn.write_csv(os.path.join(d, "296MTCKD25_S.csv"), *t["h2o_self_296"])
    syn.write_csv(os.path.join(d, "CKDF.csv"), *t["h2o_foreign_t"], extra_cols=2)
    syn.write_csv(os.path.join(d, "CKDS.csv"), *t["h2o_self_t"], extra_cols=2)
    files = {"h2o_dir": d}
    for name in ("o3_ctm", "cfc11", "cfc12", "cia_n2n2", "cia_o2n2", "cia_o2o2", "solar"):
        p = os.path.join(root, f"{name}_{'sw' if sw else 'lw'}.csv")
        syn.write_csv(p, *t[name])
        files[name] = p
    return files, t


def band_lines(total, grid, seed, physical=False):
    """The band's synthetic line lists.  With GRT_LINES_CACHE=<directory> in the environment they are generated once
    and kept there as .npz (atomic write), so that the ranks of a multi-GPU run do not each draw 2.5 M lines again."""
    cache = os.environ.get("GRT_LINES_CACHE")
    path = None
    if cache:
        os.makedirs(cache, exist_ok=True)
        path = os.path.join(cache, f"lines_{total}_{grid[0]:g}_{grid[1]:g}_{seed}_{int(physical)}.npz")
        if os.path.exists(path):
            with np.load(path) as z:
                return {m: {k: z[f"{m}_{k}"] for k in ("v0", "s0", "yair", "yself", "en", "nexp", "delta", "iso")} for m in MOL_ORDER}
    lists = syn.band_line_lists(total, grid[0], grid[1], seed, physical)
    out = {m: lists[m] for m in MOL_ORDER}
    if path is not None:
        tmp = f"{path}.{os.getpid()}.tmp.npz"
        np.savez(tmp, **{f"{m}_{k}": v for m, ln in out.items() for k, v in ln.items()})
        os.replace(tmp, path)
    return out


def build_band(device, grid_spec, lines, files, num_levels=NUM_LEVELS, method=api.LINE_SAMPLE):
    grid = api.create_spectral_grid(*grid_spec)
    go = api.GasOpticsObject(num_levels, grid, device, "", files["h2o_dir"], files["o3_ctm"], method=method)
    for m in MOL_ORDER:
        go.add_molecule_lines(m, lines[m])
    go.add_cfc(0, files["cfc11"])
    go.add_cfc(1, files["cfc12"])
    for a, b, name in CIA_PAIRS:
        go.add_cia(a, b, files[name])
    return go, grid


class G1Workload:
    """Both bands of the headline configuration, resident on one device."""

    def __init__(self, device, max_columns, lw_lines=LW_LINES, sw_lines=SW_LINES, num_levels=NUM_LEVELS,
                 lw_grid=LW_GRID, sw_grid=SW_GRID, root=None, fast=0, tile=0, lw_nslice=0, physical=False, spectral=False):
        self.root = root or tempfile.mkdtemp(prefix="grt_g1_")
        self.device, self.num_levels = device, num_levels
        self.lw_files, _ = write_tables(self.root, sw=False)
        self.sw_files, _ = write_tables(self.root, sw=True)
        self.physical = physical      # strengths with band structure (synthetic.PHYSICAL_BANDS) instead of SURVEY §8(d)'s uniform draw
        self.lw_lines = band_lines(lw_lines, lw_grid, 20261003, physical)
        self.sw_lines = band_lines(sw_lines, sw_grid, 20261004, physical)
        self.go_lw, self.grid_lw = build_band(device, lw_grid, self.lw_lines, self.lw_files, num_levels)
        self.go_sw, self.grid_sw = build_band(device, sw_grid, self.sw_lines, self.sw_files, num_levels)
        # (GRT_BENCH_LW_TILE / GRT_BENCH_SW_TILE / GRT_BENCH_SW_NSLICE: exploration only -- per-band tilings of the line kernel)
        self.go_lw.tune(fast=fast, tile=int(os.environ.get("GRT_BENCH_LW_TILE", tile)), nslice=lw_nslice)
        self.go_sw.tune(fast=fast, tile=int(os.environ.get("GRT_BENCH_SW_TILE", tile)), nslice=int(os.environ.get("GRT_BENCH_SW_NSLICE", 0)))
        self.emis = np.full(self.grid_lw.n, 0.98)
        self.albedo = np.full(self.grid_sw.n, 0.2)
        self.solar = api.create_solar_flux(self.grid_sw, self.sw_files["solar"])
        self.pipe = api.Pipeline(self.go_lw, self.go_sw, max_columns, -1, self.emis, self.albedo, self.solar, spectral=spectral)
        self.total_lines = {"lw": sum(v["v0"].size for v in self.lw_lines.values()),
                            "sw": sum(v["v0"].size for v in self.sw_lines.values())}

    def columns(self, first, count):
        cols = [syn.profile(first + c, self.num_levels) for c in range(count)]
        return api.make_columns(cols, MOL_ORDER, cfc_order=(0, 1)), cols

    def destroy(self):
        self.pipe.destroy()
        self.go_lw.destroy()
        self.go_sw.destroy()
