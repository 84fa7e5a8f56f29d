"""GRTDUMP1: the flat binary container examples/driver_app_dump.c reads in place of the reference applications' netCDF
input (rfmip-irf/src/rfmip-irf.c, era5/src/era5.c): named float64 arrays with their dimensions and a units string.

    char magic[8] = "GRTDUMP1"; int32 nvars; per variable: char name[64]; char units[32]; int32 ndims (<= 4);
    int64 dims[4]; prod(dims) float64 values, row-major.  Little-endian throughout."""
import struct

import numpy as np


def write_dump(path, variables):
    """variables: {name: array} or {name: (array, units)}; arrays of at most four dimensions (scalars become [1])."""
    with open(path, "wb") as f:
        f.write(b"GRTDUMP1")
        f.write(struct.pack("<i", len(variables)))
        for name, value in variables.items():
            arr, units = value if isinstance(value, tuple) else (value, "")
            arr = np.atleast_1d(np.asarray(arr, dtype="<f8"))
            if arr.ndim > 4 or len(name.encode()) > 63 or len(units.encode()) > 31:
                raise ValueError(f"{name}: at most four dimensions, names of 63 and units of 31 bytes")
            f.write(name.encode().ljust(64, b"\0"))
            f.write(units.encode().ljust(32, b"\0"))
            f.write(struct.pack("<i", arr.ndim))
            f.write(struct.pack("<4q", *(list(arr.shape) + [0] * (4 - arr.ndim))))
            f.write(np.ascontiguousarray(arr).tobytes())


def read_dump(path):
    out = {}
    with open(path, "rb") as f:
        if f.read(8) != b"GRTDUMP1":
            raise ValueError(f"{path}: not a GRTDUMP1 file")
        (n,) = struct.unpack("<i", f.read(4))
        for _ in range(n):
            name = f.read(64).split(b"\0")[0].decode()
            units = f.read(32).split(b"\0")[0].decode()
            (nd,) = struct.unpack("<i", f.read(4))
            dims = struct.unpack("<4q", f.read(32))[:nd]
            count = int(np.prod(dims)) if nd else 1
            out[name] = (np.frombuffer(f.read(8 * count), dtype="<f8").reshape(dims), units)
    return out
