"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950,
loads, and exports every symbol include/fs3d.h declares (no compute, no GPU)."""
import ctypes
import os
import re

from cmc_fluid_solver_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "fs3d.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fs3d_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_what_the_binding_binds():
    assert set(declared_symbols()) == set(capi.SYMBOLS)


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), "libfs3d_hip.so does not export %s" % name
    assert b"gfx950" in capi.load().fs3d_version()


def test_no_oracle_in_the_product():
    """The shipped package must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "cmc_fluid_solver_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                text = open(os.path.join(dp, f)).read()
                assert "liboracle" not in text and "from oracle" not in text and "import oracle" not in text, f
    import subprocess
    out = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_create_fails_loudly_without_gpu(built):
    """No silent fallback: without a usable device fs3d_create returns an error status."""
    import torch
    if torch.cuda.is_available():
        return
    lib = capi.load()
    h = ctypes.c_void_p()
    st = lib.fs3d_create(ctypes.byref(h), 0, capi.F32, 8, 8, 8, 0.1, 0.1, 0.1, 0, 8)
    assert st != capi.OK and not h.value
    assert lib.fs3d_last_error(None)
