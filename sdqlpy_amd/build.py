"""In-tree native builds: the HIP library (hipcc, gfx950), the TPCH generator and the text-table loader (g++).

Everything is built next to its sources so the .so files travel with the repository snapshot to
the GPU box; nothing is installed into site-packages and nothing is JIT-cached under ~/.cache.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")

HIP_LIB = os.path.join(CSRC, "libsdqlhip.so")
GEN_LIB = os.path.join(CSRC, "libtpchgen.so")
TBL_LIB = os.path.join(CSRC, "libsdqltbl.so")

# Two translation units: the ahead-of-time kernels + their C ABI (minutes to compile: hundreds of template
# instances), and the row-program path (code generation + hiprtc; seconds).  Objects are kept next to the
# sources so a change to one does not recompile the other.
HIP_UNITS = [("sdqh_hip.hip", "sdqh_hip.o"), ("sdqh_x.hip", "sdqh_x.o"), ("sdqh_codes.hip", "sdqh_codes.o"), ("sdqh_aux.hip", "sdqh_aux.o")]
HIP_SOURCES = [os.path.join(CSRC, src) for src, _ in HIP_UNITS]
HIP_HEADERS = [os.path.join(INCLUDE, "sdqh.h"), os.path.join(CSRC, "sdqh_kernels.hpp"), os.path.join(CSRC, "sdqh_host.hpp"),
               os.path.join(CSRC, "sdqh_xkernels.hpp")]      # sdqh_x.hip packs XArgs / sink arguments from its structs
# what each unit includes: the run-time skeletons (sdqh_xkernels.hpp) are compiled into sdqh_x.hip only — editing them must not
# recompile the ahead-of-time unit (a quarter of an hour)
UNIT_HEADERS = {"sdqh_x.hip": HIP_HEADERS}
HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-ffp-contract=off",          # keep the reference's a*(1.0-b) association: no FMA contraction
    "-munsafe-fp-atomics",        # native global_atomic_add_f64, no CAS loop
    "-fno-gpu-rdc", "-pthread",
    "-Wall", "-Wno-unused-function",
]
HIP_LINK = ["-shared", "-fPIC", "-pthread", "-lhiprtc"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), proc.stdout))
    return proc.stdout


def _build_to(target, cmd_for, sources, force):
    """Compile into a temporary file and rename it over `target`, under a lock on the csrc directory:
    with one process per GPU several ranks can find the library stale at once, and a peer must never
    dlopen a half-written file.  The staleness test is repeated under the lock (the winner built it)."""
    import fcntl
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not (force or _stale(target, sources)):
                return target
            tmp = "%s.tmp.%d" % (target, os.getpid())
            try:
                out = _run(cmd_for(tmp))
                os.replace(tmp, target)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
            return out
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def build_tpchgen(force=False):
    src = os.path.join(CSRC, "tpchgen.cpp")
    if force or _stale(GEN_LIB, [src]):
        _build_to(GEN_LIB, lambda out: ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", out, src], [src], force)
    return GEN_LIB


def build_tblload(force=False):
    src = os.path.join(CSRC, "tblload.cpp")
    if force or _stale(TBL_LIB, [src]):
        _build_to(TBL_LIB, lambda out: ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", out, src], [src], force)
    return TBL_LIB


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def build_hip(force=False, save_temps=False):
    """Compile the HIP kernels + C ABI for gfx950.  hipcc cross-compiles without a GPU."""
    objs = [os.path.join(CSRC, obj) for _, obj in HIP_UNITS]
    if not (force or _stale(HIP_LIB, HIP_SOURCES + HIP_HEADERS)):
        return HIP_LIB
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("hipcc not found: the HIP backend cannot be built")
    for (src, _), obj in zip(HIP_UNITS, objs):
        path = os.path.join(CSRC, src)

        def cmd_for(out, path=path):
            cmd = [hipcc] + HIP_FLAGS + ["-c", "-I", INCLUDE, "-I", CSRC, "-o", out, path]
            if save_temps:
                cmd.insert(1, "-save-temps=obj")
            return cmd
        _build_to(obj, cmd_for, [path] + UNIT_HEADERS.get(src, HIP_HEADERS[:3]), force)
    _build_to(HIP_LIB, lambda out: [hipcc] + HIP_LINK + ["-o", out] + objs, objs, True)
    return HIP_LIB


def build_all(force=False):
    return {"tpchgen": build_tpchgen(force), "tblload": build_tblload(force), "hip": build_hip(force)}


if __name__ == "__main__":
    import sys
    print(build_all(force="--force" in sys.argv))
