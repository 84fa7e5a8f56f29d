"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares."""
import ctypes as C
import os
import re

import pytest

from grtcode_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in ("grtcode_hip_api.h", "grt_ext.h", "debug.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"\bEXTERN\s+[\w\s\*]+?\b(\w+)\s*\(", src):
            names.add(m.group(1))
    return names


def test_library_loads_and_exports_every_declared_symbol(lib):
    decl = declared_symbols()
    assert len(decl) > 70
    missing = sorted(s for s in decl if not hasattr(lib, s))
    assert not missing, f"declared in include/*.h but not exported: {missing}"
    assert set(api.EXPORTS) == decl, (set(api.EXPORTS) ^ decl)


def test_forwarding_headers_cover_the_reference_include_names():
    want = ["gas_optics.h", "longwave.h", "shortwave.h", "rayleigh.h", "solar_flux.h", "optics.h",
            "spectral_grid.h", "device.h", "utilities.h", "verbosity.h", "return_codes.h",
            "floating_point_type.h", "grtcode_utilities.h", "molecules.h", "cfcs.h",
            "collision_induced_absorption.h", "tips2017.h", "parse_csv.h", "extern.h"]
    for h in want:
        assert os.path.exists(os.path.join(ROOT, "include", h)), h


def test_struct_layouts_match_between_c_and_ctypes(lib):
    kinds = [api.SpectralGrid, api.Optics, api.GasOptics, api.SolarFlux, api.Longwave, api.Shortwave]
    for k, t in enumerate(kinds):
        assert lib.grt_sizeof(k) == C.sizeof(t), t.__name__


def test_static_archives_with_the_reference_names_exist():
    for a in ("libgrtcode_utilities.a", "libgas_optics.a", "liblongwave.a", "libshortwave.a"):
        assert os.path.exists(os.path.join(ROOT, "grtcode_amd", "lib", a)), a


def test_driver_style_c_program_compiles_against_the_headers(tmp_path):
    """A caller written against the reference's include names compiles unchanged (syntax + types)."""
    src = tmp_path / "caller.c"
    src.write_text(r'''
#include "grtcode_utilities.h"
#include "gas_optics.h"
#include "longwave.h"
#include "shortwave.h"
#include "rayleigh.h"
#include "solar_flux.h"
int column(GasOptics_t lbl, Optics_t gas, Optics_t ray, Longwave_t lw, fp_t *p, fp_t *t, fp_t *tl, fp_t *e, fp_t *up, fp_t *dn)
{
    Optics_t total;
    int rc = set_molecule_ppmv(&lbl, H2O, p);
    if (rc != GRTCODE_SUCCESS) return rc;
    rc = calculate_optical_depth(&lbl, p, t, &gas);
    rc = rayleigh_scattering(&ray, p);
    Optics_t const *arr[2] = {&gas, &ray};
    rc = add_optics(arr, 2, &total);
    rc = calculate_lw_fluxes(&lw, &total, 290., tl, t, e, up, dn);
    fp_t s = 0.;
    for (uint64_t i = 0; i + 1 < lw.grid.n; ++i) s += 0.5*(up[i] + up[i + 1])*lw.grid.dw;
    (void)s;
    return destroy_optics(&total);
}
''')
    import subprocess
    r = subprocess.run(["gcc", "-std=gnu99", "-Wall", "-Werror", "-fsyntax-only", f"-I{ROOT}/include", str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_no_gpu_means_loud_failure_not_fallback(lib):
    n = C.c_int(-1)
    assert lib.get_num_gpus(C.byref(n), 0) == 0
    if n.value == 0:
        with pytest.raises(api.GrtError) as e:
            api.create_device()
        assert e.value.code == api.GPU_ERR
    with pytest.raises(api.GrtError) as e:
        api.create_device(api.HOST_ONLY)
    assert e.value.code == api.VALUE_ERR
