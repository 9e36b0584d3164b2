"""Build libfs3d_hip.so (the C-ABI library of include/fs3d.h) for gfx950 with hipcc.

In-tree build: the .so lands next to this file so that it travels to the GPU box
with the repo snapshot.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libfs3d_hip.so")
SOURCES = ["fs3d_hip.hip", "fs3d_comm.hip", "kernels_line.hip", "kernels_pipe.hip", "kernels_part.hip"]
HEADERS = ["fs3d_common.h", "fs3d_rows.h", "fs3d_comm.h", os.path.join("..", "..", "include", "fs3d.h")]

# -ffp-contract=off: no FMA contraction, the reference's CPU path rounds after every operation.
# -fhip-fp32-correctly-rounded-divide-sqrt: IEEE fp32 division (the hipcc default, stated explicitly).
# no -ffast-math, denormals preserved (hipcc default).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value"]
# kernels_part.hip (partition solve, tolerance-based parity): FMA contraction allowed; everything else as above.
FLAGS_BY_SOURCE = {"kernels_part.hip": [f for f in FLAGS if f != "-ffp-contract=off"] + ["-ffp-contract=fast"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + FLAGS_BY_SOURCE.get(s, FLAGS) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + \
              ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


def build_variant(name, extra_flags, source="kernels_part.hip", experiments=True):
    """Kernel experiments: libfs3d_hip_<name>.so with `source` compiled with extra flags (load it with FS3D_LIB_PATH).
    experiments: -DFS3D_EXPERIMENTS (the timing-experiment environment knobs exist only in such builds)."""
    build()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    obj = os.path.join(CSRC, source.replace(".hip", "_var_%s.o" % name))
    subprocess.check_call([hipcc] + FLAGS_BY_SOURCE.get(source, FLAGS) + (["-DFS3D_EXPERIMENTS"] if experiments else []) + list(extra_flags) + ["-c", os.path.join(CSRC, source), "-o", obj])
    objs = [obj if s == source else os.path.join(CSRC, s.replace(".hip", ".o")) for s in SOURCES]
    lib = os.path.join(HERE, "libfs3d_hip_%s.so" % name)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"])
    return lib


DRIVER = os.path.join(HERE, "fs3d_run")
DRIVER2D = os.path.join(HERE, "fs2d_run")


def build_driver2d(force=False):
    """The CPU-only 2D command line (host/fs2d_run.cpp: Stable solver over a Grid2D); g++, no GPU library."""
    host = os.path.join(HERE, "host")
    deps = [os.path.join(host, f) for f in ("fs2d_run.cpp", "Stable2D.h", "Shape2D.h", "Config.h", "AdiSolver3D_hip.h")]
    if force or _stale(DRIVER2D, deps):
        subprocess.check_call([os.environ.get("CXX", "g++"), "-std=c++17", "-O2", "-Wall", "-ffp-contract=off", "-I" + os.path.join(HERE, "..", "include"),
                               os.path.join(host, "fs2d_run.cpp"), "-o", DRIVER2D])
    return DRIVER2D


def build_driver(force=False, verbose=False):
    """The C++ command-line driver (host/fs3d_run.cpp) above the C ABI; g++, links libfs3d_hip.so."""
    build(force=False, verbose=verbose)
    host = os.path.join(HERE, "host")
    deps = [os.path.join(host, f) for f in ("fs3d_run.cpp", "AdiSolver3D_hip.h", "Config.h", "Shape2D.h", "Shape3D.h", "SeaNetCDF.h", "Hdf5Min.h", "NetCDF3.h", "GridImage.h")] + [LIB]
    if force or _stale(DRIVER, deps):
        cmd = [os.environ.get("CXX", "g++"), "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(HERE, "..", "include"),
               os.path.join(host, "fs3d_run.cpp"), "-o", DRIVER, "-L" + HERE, "-lfs3d_hip", "-lz", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return DRIVER


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_driver(force="--force" in sys.argv, verbose=True))
