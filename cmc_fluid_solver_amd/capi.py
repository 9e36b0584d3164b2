"""ctypes binding of libfs3d_hip.so (include/fs3d.h) and a thin `Solver` class that
mirrors the reference's Solver3D / AdiSolver3D host interface
(Solver3D.h:24-49, AdiSolver3D.h:61-70): Init / UpdateBoundaries / TimeStep / GetLayer.

There is NO fallback: if the HIP library is missing or no GPU is present the calls
fail loudly (RuntimeError carrying fs3d_last_error()).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FS3D_LIB_PATH") or os.path.join(_HERE, "libfs3d_hip.so")   # override: kernel experiments

F32, F64 = 0, 1
OK, ERR_INVALID, ERR_HIP, ERR_DIVERGED, ERR_UNSUPPORTED, ERR_COMM = range(6)
DIR_X, DIR_Y, DIR_Z = 0, 1, 2
LAYER_CUR, LAYER_TEMP, LAYER_HALF, LAYER_NEXT = 0, 1, 2, 3
SWEEP_AUTO, SWEEP_LINE, SWEEP_PIPE, SWEEP_PART, SWEEP_EXACT = 0, 1, 2, 3, 4
KERNEL_NAMES = {0: "none", 1: "line", 2: "pipe", 3: "part"}
OPT_SWEEP_KERNEL, OPT_FUSE_MERGE, OPT_DIV_CORE, OPT_XSOLVE, OPT_OVERLAP, OPT_KEEP_TEMP = 0, 1, 2, 3, 4, 5
XSOLVE_AUTO, XSOLVE_PIPELINED, XSOLVE_REDUCED, XSOLVE_REDUCED_A2A = 0, 1, 2, 3

# every symbol include/fs3d.h declares: name -> (restype, argtypes)
_vp, _i, _d = C.c_void_p, C.c_int, C.c_double
SYMBOLS = {
    "fs3d_create": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i, _d, _d, _d, _i, _i]),
    "fs3d_destroy": (None, [_vp]),
    "fs3d_last_error": (C.c_char_p, [_vp]),
    "fs3d_set_params": (_i, [_vp, _d, _d, _d, _d]),
    "fs3d_set_option": (_i, [_vp, _i, _i]),
    "fs3d_upload_nodes": (_i, [_vp] + [_vp] * 7 + [C.POINTER(_i)]),
    "fs3d_init_layers_from_nodes": (_i, [_vp]),
    "fs3d_upload_layer": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "fs3d_download_layer": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "fs3d_field_dev_ptr": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "fs3d_update_boundaries": (_i, [_vp]),
    "fs3d_time_step": (_i, [_vp, _d, _i, _i, _i, C.POINTER(_d)]),
    "fs3d_time_step_async": (_i, [_vp, _d, _i, _i]),
    "fs3d_synchronize": (_i, [_vp]),
    "fs3d_sweep": (_i, [_vp, _i, _d, _i, _i, _i, _i]),
    "fs3d_merge": (_i, [_vp, _i, _i]),
    "fs3d_eval_div_error": (_i, [_vp, _i, C.POINTER(_d), C.POINTER(C.c_longlong)]),
    "fs3d_get_layer": (_i, [_vp, _vp, _vp, _i, _i, _i]),
    "fs3d_comm_unique_id": (_i, [_vp]),
    "fs3d_comm_init": (_i, [_vp, _vp, _i, _i]),
    "fs3d_local_group_create": (_i, [_i, C.POINTER(_vp)]),
    "fs3d_local_group_destroy": (None, [_vp]),
    "fs3d_local_group_abort": (None, [_vp]),
    "fs3d_comm_init_local": (_i, [_vp, _vp, _i]),
    "fs3d_comm_abort": (_i, [_vp]),
    "fs3d_comm_selftest": (_i, [_vp, C.c_size_t]),
    "fs3d_last_step_timing": (_i, [_vp, C.POINTER(C.c_float), C.POINTER(_i)]),
    "fs3d_enable_timing": (_i, [_vp, _i]),
    "fs3d_profiler_events": (_i, [_vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(_i)]),
    "fs3d_profile_sweep": (_i, [_vp, _i, _d, _i, _i, _i, C.POINTER(C.c_ulonglong), _i, C.POINTER(_i)]),
    "fs3d_last_sweep_kernel": (_i, [_vp, _i, C.POINTER(_i), C.POINTER(_i)]),
    "fs3d_version": (C.c_char_p, []),
}

_lib = None


class Fs3dError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("fs3d status %d: %s" % (status, msg))
        self.status = status


def load():
    """dlopen libfs3d_hip.so and bind every declared symbol.  No GPU is touched."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libfs3d_hip.so is not built (%s); run `python -m cmc_fluid_solver_amd.build` "
                               "or __graft_entry__.build()" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Solver:
    """Host-side mirror of the reference's AdiSolver3D for ONE GPU / one x-slab.

    Init(grid, params) -> __init__(nodes, params) ; UpdateBoundaries ; TimeStep ; GetLayer.
    `nodes` is the GLOBAL grids.Nodes object; x_range selects the owned planes.
    """

    def __init__(self, nodes, params, dtype=np.float32, device=0, x_range=None):
        self.lib = load()
        self.dtype = np.dtype(dtype)
        self.prec = F32 if self.dtype == np.float32 else F64
        x0, x1 = x_range if x_range is not None else (0, nodes.dimx)
        self.x0, self.x1 = x0, x1
        self.dims = (x1 - x0, nodes.dimy, nodes.dimz)
        self.gdims = nodes.shape
        self.h = C.c_void_p()
        st = self.lib.fs3d_create(C.byref(self.h), device, self.prec, x1 - x0, nodes.dimy, nodes.dimz,
                                  nodes.dx, nodes.dy, nodes.dz, x0, nodes.dimx)
        if st != OK:
            raise Fs3dError(st, (self.lib.fs3d_last_error(None) or b"").decode())
        self._chk(self.lib.fs3d_set_params(self.h, *[float(p) for p in params]))
        arrs = [np.ascontiguousarray(nodes.type, np.uint8), np.ascontiguousarray(nodes.bc_vel, np.uint8),
                np.ascontiguousarray(nodes.bc_temp, np.uint8)] + [
            np.ascontiguousarray(v, self.dtype) for v in (nodes.vx, nodes.vy, nodes.vz, nodes.T)]
        nseg = (C.c_int * 3)()
        self._chk(self.lib.fs3d_upload_nodes(self.h, *[_p(a) for a in arrs], nseg))
        self.num_segments = list(nseg)
        self._chk(self.lib.fs3d_init_layers_from_nodes(self.h))

    # -- plumbing -----------------------------------------------------------------
    def _chk(self, st):
        if st != OK:
            raise Fs3dError(st, (self.lib.fs3d_last_error(self.h) or b"").decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.fs3d_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, opt, val):
        self._chk(self.lib.fs3d_set_option(self.h, opt, val))

    def comm_init(self, unique_id, rank, nranks):
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        self._chk(self.lib.fs3d_comm_init(self.h, buf, rank, nranks))

    def comm_selftest(self, elems=1 << 18):
        """One-rank RCCL communicator on this context's device: grouped send/recv, all-gather, all-reduce, verified."""
        self._chk(self.lib.fs3d_comm_selftest(self.h, elems))

    def comm_abort(self):
        self._chk(self.lib.fs3d_comm_abort(self.h))

    def comm_init_local(self, group, rank):
        self._chk(self.lib.fs3d_comm_init_local(self.h, group.h, rank))
        self._group = group        # keep the group alive as long as the context

    # -- reference-shaped interface -----------------------------------------------
    def UpdateBoundaries(self):
        self._chk(self.lib.fs3d_update_boundaries(self.h))

    def TimeStep(self, dt, num_global, num_local, computeError=True):
        """Returns diffError.  Raises Fs3dError(status ERR_DIVERGED) where the reference throws."""
        err = C.c_double(0.0)
        self._chk(self.lib.fs3d_time_step(self.h, dt, num_global, num_local, int(computeError), C.byref(err)))
        return err.value

    def time_step_async(self, dt, num_global, num_local):
        self._chk(self.lib.fs3d_time_step_async(self.h, dt, num_global, num_local))

    def synchronize(self):
        self._chk(self.lib.fs3d_synchronize(self.h))

    def GetLayer(self, outdims=(0, 0, 0)):
        od = [o or d for o, d in zip(outdims, self.dims)]
        outV = np.empty(od + [3], dtype=self.dtype)
        outT = np.empty(od, dtype=np.float64)
        self._chk(self.lib.fs3d_get_layer(self.h, _p(outV), _p(outT), *outdims))
        return outV, outT

    # -- kernel-level access ------------------------------------------------------
    def sweep(self, d, dt, l_cur, l_temp, l_next, merge=False):
        self._chk(self.lib.fs3d_sweep(self.h, d, dt, l_cur, l_temp, l_next, int(merge)))

    def merge(self, l_src, l_dest):
        self._chk(self.lib.fs3d_merge(self.h, l_src, l_dest))

    def eval_div_error(self, layer=LAYER_NEXT):
        err, cnt = C.c_double(0.0), C.c_longlong(0)
        self._chk(self.lib.fs3d_eval_div_error(self.h, layer, C.byref(err), C.byref(cnt)))
        return err.value, cnt.value

    def download_layer(self, layer):
        out = [np.empty(self.dims, dtype=self.dtype) for _ in range(4)]
        self._chk(self.lib.fs3d_download_layer(self.h, layer, *[_p(a) for a in out]))
        return out

    def upload_layer(self, layer, fields):
        arrs = [None if f is None else np.ascontiguousarray(f, self.dtype) for f in fields]
        for a in arrs:
            assert a is None or a.shape == tuple(self.dims)
        self._chk(self.lib.fs3d_upload_layer(self.h, layer, *[_p(a) for a in arrs]))

    def profile_sweep(self, d, dt, l_cur=LAYER_CUR, l_temp=LAYER_TEMP, l_next=LAYER_NEXT, max_blocks=4096):
        """[blocks, 8 waves, 8 stamps] shader-clock stamps of one pipelined sweep (measurement aid)."""
        buf = np.zeros((max_blocks, 8, 8), dtype=np.uint64)   # up to 8 waves per workgroup (unused waves stay 0)
        nb = C.c_int(0)
        self._chk(self.lib.fs3d_profile_sweep(self.h, d, dt, l_cur, l_temp, l_next,
                                              buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), max_blocks, C.byref(nb)))
        return buf[:nb.value]

    def last_sweep_kernels(self):
        """{"X": "part", "Y": "part", "Z": "pipe"}: what the last sweep of each direction really ran."""
        out = {}
        for d, nm in enumerate("XYZ"):
            k, sg = C.c_int(0), C.c_int(0)
            self._chk(self.lib.fs3d_last_sweep_kernel(self.h, d, C.byref(k), C.byref(sg)))
            out[nm] = KERNEL_NAMES.get(k.value, str(k.value)) + ("-segmented" if sg.value & 1 else "") + \
                {0: "", 1: "+pipelined-ranks", 2: "+reduced-interface", 3: "+reduced-interface(on-chip)"}[(sg.value >> 1) & 3] + \
                ("+all-to-all" if sg.value & 8 else "")
        return out

    def profiler_events(self):
        """{event name of the reference's Profiler: (total ms, count)} since enable_timing(True)"""
        names, ms, n = (C.c_char_p * 9)(), (C.c_float * 9)(), (C.c_int * 9)()
        self._chk(self.lib.fs3d_profiler_events(self.h, names, ms, n))
        return {names[k].decode(): (ms[k], n[k]) for k in range(9)}

    def enable_timing(self, on=True):
        self._chk(self.lib.fs3d_enable_timing(self.h, int(on)))

    def last_step_timing(self):
        ms, n = (C.c_float * 4)(), (C.c_int * 4)()
        self._chk(self.lib.fs3d_last_step_timing(self.h, ms, n))
        return list(ms), list(n)


class LocalGroup:
    """In-process slab group (fs3d_local_group_create): one Solver per slab, each driven by its own
    thread.  `run(fn)` calls fn(rank, solver) on every slab concurrently and returns the results."""

    def __init__(self, nodes, params, nranks, dtype=np.float32, devices=None):
        from .slab import slab_range
        self.lib = load()
        self.h = C.c_void_p()
        st = self.lib.fs3d_local_group_create(nranks, C.byref(self.h))
        if st != OK:
            raise Fs3dError(st, "fs3d_local_group_create")
        self.solvers = []
        for r in range(nranks):
            s = Solver(nodes, params, dtype=dtype, device=(devices[r] if devices else 0),
                       x_range=slab_range(nodes.dimx, r, nranks))
            s.comm_init_local(self, r)
            self.solvers.append(s)

    def run(self, fn):
        import threading
        out, exc = [None] * len(self.solvers), [None] * len(self.solvers)

        def work(r):
            try:
                out[r] = fn(r, self.solvers[r])
            except BaseException as e:      # noqa: BLE001 - re-raised below
                exc[r] = e
                try:
                    self.solvers[r].comm_abort()        # the other slab threads return ERR_COMM instead of waiting for ever
                except Exception:
                    pass
        th = [threading.Thread(target=work, args=(r,)) for r in range(len(self.solvers))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for e in exc:
            if e is not None:
                raise e
        return out

    def close(self):
        for s in self.solvers:
            s.close()
        self.solvers = []
        if self.h:
            self.lib.fs3d_local_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fluid_params(dtype, Re, Pr, lam):
    """FluidParams(Re, Pr, lambda), Geometry.h:545-552, rounded to FTYPE."""
    dt = np.dtype(dtype).type
    return (dt(1.0), dt(1.0 / Re), dt(1.0 / (Re * Pr)), dt((lam - 1) / (lam * Re)))


def fluid_params_physical(dtype, vis, rho, R, k, cv):
    """FluidParams(vis, rho, R, k, cv), Geometry.h:554-561."""
    dt = np.dtype(dtype).type
    return (dt(R), dt(vis / rho), dt(k / (rho * cv)), dt(vis / (rho * cv)))
