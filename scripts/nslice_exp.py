import time, json, sys
sys.path.insert(0, '/root/repo')
import numpy as np
from grtcode_amd import api, synthetic as syn, workload as W
device = api.create_device(0)
wl = W.G1Workload(device, 1)
V = W.NUM_LEVELS
gas = {"lw": api.OpticsObject(V - 1, wl.grid_lw, device), "sw": api.OpticsObject(V - 1, wl.grid_sw, device)}
ray = api.OpticsObject(V - 1, wl.grid_sw, device)
sw = api.ShortwaveObject(V, wl.grid_sw, device)
def setc(go, col):
    for m in W.MOL_ORDER: go.set_molecule_ppmv(m, col["ppmv"][m])
    go.set_cfc_ppmv(0, col["cfc_ppmv"][0]); go.set_cfc_ppmv(1, col["cfc_ppmv"][1])
    go.set_cia_ppmv(0, col["ppmv"][syn.N2]); go.set_cia_ppmv(1, col["ppmv"][syn.O2])
col = syn.profile(0, V)
setc(wl.go_lw, col); setc(wl.go_sw, col)
wl.go_lw.tune(fast=3); wl.go_sw.tune(fast=3)
wl.go_lw.calculate_optical_depth(col["p"], col["t"], gas["lw"])
wl.go_sw.calculate_optical_depth(col["p"], col["t"], gas["sw"])
ray.rayleigh(col["p"])
def lw_time(label):
    t0 = time.perf_counter(); wl.go_lw.calculate_optical_depth(col["p"], col["t"], gas["lw"]); print(label, "-> lw %.3f ms" % ((time.perf_counter() - t0) * 1e3))
lw_time("baseline")
time.sleep(0.05); lw_time("after 50 ms idle")
b = api.DeviceBuffer(device, 72_000_000); b.free(); lw_time("after hipMalloc+hipFree 72 MB")
tot = api.add_optics([gas["sw"], ray]); lw_time("after add_optics (sw)")
up, dn = sw.fluxes(tot, col["mu0"], 0.5, wl.albedo, wl.albedo, col["tsi"], wl.solar); lw_time("after calculate_sw_fluxes")
tot.destroy(); lw_time("after destroy_optics")
host = np.empty(6_000_000); t0 = time.perf_counter(); api.check(api.load_library().grt_device_to_host(device, host.ctypes.data, gas["sw"].c.tau, host.nbytes)); print("D2H 48 MB pageable %.2f ms" % ((time.perf_counter()-t0)*1e3)); lw_time("after pageable D2H")
lw_time("again")
