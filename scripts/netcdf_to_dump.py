#!/usr/bin/env python3
"""netCDF -> GRTDUMP1, for boxes that have netCDF4 (this image does not): the input of examples/driver_app_dump.c.

    python scripts/netcdf_to_dump.py multiple_input4MIPs_radiation_RFMIP_UColorado-RFMIP-1-2_none.nc rfmip.dump
    python scripts/netcdf_to_dump.py era5.nc era5.dump ; python scripts/netcdf_to_dump.py ghg.nc ghg.dump

Every numeric variable goes in under its own name with its dimensions (at most four) as float64, row-major, and its
"units" attribute (the RFMIP global means keep their scale there: "1e-6").  Layout: grtcode_amd.dumpfile.write_dump."""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from grtcode_amd.dumpfile import write_dump


def main(src, dst):
    import netCDF4
    variables = {}
    with netCDF4.Dataset(src) as nc:
        nc.set_auto_mask(False)
        for name, v in nc.variables.items():
            if v.dtype.kind in "fiu" and v.ndim <= 4:
                variables[name] = (np.asarray(v[...], dtype=np.float64), str(getattr(v, "units", "")))
    write_dump(dst, variables)


if __name__ == "__main__":
    main(*sys.argv[1:3])
