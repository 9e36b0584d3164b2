"""Failures must be loud (GPUplan.cpp:173-193: the reference throws on every device error): a relay hand-over that never
arrives, a slab thread that dies in the middle of the protocol, a kernel the dims do not allow."""
import threading
import time

import numpy as np
import pytest

from cmc_fluid_solver_amd import capi, grids

pytestmark = pytest.mark.gpu
PARAMS = (200.0, 0.72, 1.4)


def test_lost_relay_handover_is_an_error(built, monkeypatch):
    """Test hook FS3D_TEST_DROP_HANDOFF: one wave of the exact pipe kernel never signals its forward pass.  The waits are
    bounded (the GPU never hangs), the error word is set, and the call returns FS3D_ERR_HIP instead of wrong fields."""
    g = grids.box(24, 20, 64, h=0.03)
    monkeypatch.setenv("FS3D_TEST_DROP_HANDOFF", "1")
    s = capi.Solver(g, capi.fluid_params(np.float32, *PARAMS), np.float32)
    s.set_option(capi.OPT_SWEEP_KERNEL, capi.SWEEP_PIPE)
    with pytest.raises(capi.Fs3dError) as ei:
        s.sweep(1, 0.1, capi.LAYER_CUR, capi.LAYER_TEMP, capi.LAYER_NEXT, merge=True)
    assert ei.value.status == capi.ERR_HIP and "hand-over" in str(ei.value)
    monkeypatch.delenv("FS3D_TEST_DROP_HANDOFF")
    with pytest.raises(capi.Fs3dError):                  # the hook is read ONCE, at fs3d_create: this context keeps dropping
        s.sweep(1, 0.1, capi.LAYER_CUR, capi.LAYER_TEMP, capi.LAYER_NEXT, merge=True)
    s.close()
    s = capi.Solver(g, capi.fluid_params(np.float32, *PARAMS), np.float32)              # a context created without it is clean
    s.set_option(capi.OPT_SWEEP_KERNEL, capi.SWEEP_PIPE)
    s.sweep(1, 0.1, capi.LAYER_CUR, capi.LAYER_TEMP, capi.LAYER_NEXT, merge=True)
    s.close()


def test_a_failing_slab_releases_its_peers(built):
    """Rank 1 of a 3-slab in-process group aborts instead of taking its time step: ranks 0 and 2, blocked in the halo
    exchange / the cross-slab X sweep, return FS3D_ERR_COMM (no hang)."""
    g = grids.box(48, 20, 64, h=0.03)
    grp = capi.LocalGroup(g, capi.fluid_params(np.float32, *PARAMS), 3, np.float32)
    res = {}

    def work(r, sv):
        if r == 1:
            time.sleep(0.3)
            raise RuntimeError("slab 1 gives up")
        try:
            sv.UpdateBoundaries()
            sv.TimeStep(0.1, 4, 2, True)
            res[r] = "ok"
        except capi.Fs3dError as e:
            res[r] = e.status
        return None
    t0 = time.time()
    with pytest.raises(RuntimeError, match="slab 1 gives up"):
        grp.run(work)
    assert time.time() - t0 < 60
    assert res == {0: capi.ERR_COMM, 2: capi.ERR_COMM}
    grp.close()


def test_auto_reports_the_kernel_it_ran(built):
    g = grids.box(24, 20, 70, h=0.03)          # Z lines of 70 cells: no partition / pipe kernel takes them
    s = capi.Solver(g, capi.fluid_params(np.float32, *PARAMS), np.float32)
    assert s.last_sweep_kernels() == {"X": "none", "Y": "none", "Z": "none"}
    s.UpdateBoundaries(); s.TimeStep(0.1, 4, 2, True)
    assert s.last_sweep_kernels() == {"X": "part", "Y": "part", "Z": "line"}
    s.close()
