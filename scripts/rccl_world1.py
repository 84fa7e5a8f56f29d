"""RCCL at world size 1 through the library's own C entry points (grt_multi_*): communicator via a rendezvous directory,
ncclGather on the library stream, max-reduce.  Diagnostic: run with NCCL_DEBUG=INFO to see RCCL's own account."""
import ctypes as C
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if "--with-torch" in sys.argv:
    import torch  # noqa: F401  (its bundled librccl is then already mapped when the library dlopens librccl.so.1)
from grtcode_amd import api

lib = api.load_library()
device = api.create_device(0)
rdv = tempfile.mkdtemp(prefix="grt_rdv_")
m = C.c_void_p()
api.check(lib.grt_multi_create(C.byref(m), 0, device, 0, 1, rdv.encode()))
local = np.arange(36, dtype=np.float64).reshape(3, 12)
buf = api.DeviceBuffer(device, local.nbytes)
api.check(lib.grt_host_to_device(device, buf.ptr, local.ctypes.data_as(C.c_void_p), local.nbytes))
allb = api.DeviceBuffer(device, local.nbytes)
api.check(lib.grt_multi_gather_fluxes(m, buf.ptr, 3, allb.ptr, 1))
v = C.c_double(1.5)
api.check(lib.grt_multi_max(m, C.byref(v)))
got = allb.to_host((3, 12))
assert np.array_equal(got, local) and v.value == 1.5
api.check(lib.grt_multi_destroy(C.byref(m)))
print("rccl world-1 through grt_multi: ok")
