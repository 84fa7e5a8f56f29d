"""Build the C-ABI library: C99 host layer (gcc) + hand-written gfx950 kernels (hipcc).

    python -m grtcode_amd.build            # -> grtcode_amd/lib/libgrtcode_hip.so + the four .a
    python -m grtcode_amd.build --force

Cross-compiles without a GPU (hipcc --offload-arch=gfx950).  Outputs stay in-tree
(git-ignored) so that they travel with gpurun snapshots.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(HERE, "lib")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
SO = os.path.join(LIB, "libgrtcode_hip.so")

HOST_SRC = ["grt_error.c", "grt_util.c", "grt_grid.c", "grt_device.c", "grt_optics.c", "grt_tips.c",
            "grt_gas_optics.c", "grt_solvers.c", "grt_pipeline.c", "grt_multi.c", "grt_clouds.c"]
NOT_IN_SO = {"grt_clouds"}      # libclouds.a only: a maintainer links the reference's own libclouds.a in its place
HIP_SRC = ["k_gas_optics.hip", "k_gas_optics_mp.hip", "k_gas_optics_far.hip", "k_gas_optics_sweep.hip", "k_optics.hip", "k_longwave.hip", "k_shortwave.hip"]

# the reference's archive names (*/src/Makefile.am): which objects go where
ARCHIVES = {
    "libgrtcode_utilities.a": ["grt_error", "grt_util", "grt_grid", "grt_device", "grt_optics", "k_optics"],
    "libgas_optics.a": ["grt_tips", "grt_gas_optics", "k_gas_optics", "k_gas_optics_mp", "k_gas_optics_far", "k_gas_optics_sweep"],
    "liblongwave.a": ["k_longwave"],
    "libshortwave.a": ["k_shortwave"],
    # solvers' host entry points and the batched pipeline reference both bands
    "libgrtcode_hip_ext.a": ["grt_solvers", "grt_pipeline", "grt_multi"],
    # the reference's cloud-optics archive name: entry points only (SURVEY §8 f-4 is not built), so that
    # framework/src/driver.c links unchanged
    "libclouds.a": ["grt_clouds"],
}

CFLAGS = ["-std=gnu99", "-O2", "-ffp-contract=off", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-parameter",
          "-D__HIP_PLATFORM_AMD__", f"-I{ROOT}/include", f"-I{ROCM}/include", f"-I{CSRC}/host"]
HIPFLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-munsafe-fp-atomics", "-fno-slp-vectorize",
            f"-I{ROOT}/include"]
# -fno-slp-vectorize: the SLP vectoriser pairs neighbouring f32 operations into v_pk_* instructions and pays for the pairs
# with register moves; measured on G1: longwave launch 45.7 -> 44.7 ms, far-field gather 6.0 -> 5.4, 367.7 -> 371.6 columns/s
HIPFLAGS += os.environ.get("GRT_HIPFLAGS_EXTRA", "").split()       # exploration only


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in (src,) + tuple(extra))


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout)
        raise RuntimeError(f"build step failed: {cmd[0]} {cmd[-1]}")
    return r.stdout


def _flags_stamp():
    import hashlib
    return os.path.join(OBJ, "flags.sha256"), hashlib.sha256("\0".join(CFLAGS + ["|"] + HIPFLAGS).encode()).hexdigest()


def _flags_changed():
    """The flags the objects in lib/obj were built with are kept next to them: a build under other flags (GRT_HIPFLAGS_EXTRA
    set or dropped, a new default) rebuilds everything instead of silently reusing objects of the other configuration.
    The stamp is removed here and written again only after the link succeeded (build()), so a build that fails half way
    cannot leave objects of two configurations under a stamp that matches."""
    stamp, want = _flags_stamp()
    have = open(stamp).read().strip() if os.path.exists(stamp) else None
    if have != want:
        if os.path.exists(stamp):
            os.remove(stamp)
        return have is not None or any(n.endswith(".o") for n in os.listdir(OBJ))
    return False


def build(force=False, verbose=False):
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ, exist_ok=True)
    force = force or _flags_changed()
    headers = [os.path.join(ROOT, "include", h) for h in ("grtcode_hip_api.h", "grt_ext.h")]
    headers += [os.path.join(CSRC, "grt_kernels.h"), os.path.join(CSRC, "hip", "gas_optics_dev.h"), os.path.join(CSRC, "hip", "gas_optics_mp_dev.h"), os.path.join(CSRC, "hip", "mp_general_block.inc"), os.path.join(CSRC, "hip", "mp_lean_block.inc"),
                os.path.join(CSRC, "hip", "optics_dev.h"), os.path.join(CSRC, "hip", "exp_pair.h"), os.path.join(CSRC, "host", "grt_internal.h"), os.path.join(CSRC, "host", "grt_molecule_table.h")]
    objs, jobs = [], []
    for f in HOST_SRC:
        src, obj = os.path.join(CSRC, "host", f), os.path.join(OBJ, f[:-2] + ".o")
        if force or _newer(src, obj, headers):
            jobs.append(["gcc"] + CFLAGS + ["-c", src, "-o", obj])
        objs.append(obj)
    for f in HIP_SRC:
        src, obj = os.path.join(CSRC, "hip", f), os.path.join(OBJ, f[:-4] + ".o")
        if force or _newer(src, obj, headers):
            jobs.append(["hipcc"] + HIPFLAGS + ["-c", src, "-o", obj])
        objs.append(obj)
    # (the translation units are independent: compiled side by side, four at a time -- the container has 8 CPUs)
    with ThreadPoolExecutor(max_workers=int(os.environ.get("GRT_BUILD_JOBS", "4"))) as pool:
        for out in pool.map(_run, jobs):
            if verbose and out:
                print(out)
    if force or jobs or any(_newer(o, SO) for o in objs):
        _run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] +
             [o for o in objs if os.path.basename(o)[:-2] not in NOT_IN_SO] +
             [f"-L{ROCM}/lib", "-lamdhip64", "-lm", "-ldl", f"-Wl,-rpath,{ROCM}/lib"])
        for name, members in ARCHIVES.items():
            path = os.path.join(LIB, name)
            if os.path.exists(path):
                os.remove(path)
            _run(["ar", "rcs", path] + [os.path.join(OBJ, m + ".o") for m in members])
    stamp, want = _flags_stamp()
    with open(stamp, "w") as f:
        f.write(want + "\n")
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
