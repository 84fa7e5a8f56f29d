"""Does a live RCCL communicator slow this library's kernels?  (-0.8 to -1.4 % at world size 1.)
Times the same 8-column G1 launch sequence (HIP events of the shortwave first pass) before a communicator exists, while
one is alive (made through the library's own C entry points, no torch process group), and after it is destroyed.

    PYTHONPATH=. python scripts/rccl_cost_probe.py [--torch]    (--torch: import torch first, i.e. its bundled ROCm runtime)"""
import ctypes as C
import json
import sys
import tempfile

if "--torch" in sys.argv:
    import torch  # noqa: F401
from grtcode_amd import api, workload as W

lib = api.load_library()
device = api.create_device(0)
wl = W.G1Workload(device, 8, fast=3)
(gcols, keep), _ = wl.columns(0, 8)
api.profile_enable(True)


def timed(label, reps=12):
    wl.pipe.run(gcols)
    wl.pipe.sync()
    api.profile_read(2, reset=True)
    for _ in range(reps):
        wl.pipe.run(gcols)
    wl.pipe.sync()
    sw, lw = api.profile_read(2)[0] / reps, api.profile_read(1)[0] / reps
    print(label, "sw first pass ms", round(sw, 3), "lw", round(lw, 3), flush=True)
    return sw


out = {"before": timed("before    ")}
m = C.c_void_p()
api.check(lib.grt_multi_create(C.byref(m), 0, device, 0, 1, tempfile.mkdtemp(prefix="grt_rdv_").encode()))
out["communicator_alive"] = timed("comm alive")
v = C.c_double(1.0)
api.check(lib.grt_multi_max(m, C.byref(v)))
out["after_a_collective"] = timed("after coll")
api.check(lib.grt_multi_destroy(C.byref(m)))
out["after_destroy"] = timed("destroyed ")
print(json.dumps(out))
