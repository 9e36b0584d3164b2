"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of the CPU oracle (liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
The shipped HIP path (cmc_fluid_solver_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

X, Y, Z = 0, 1, 2
L_CUR, L_TEMP, L_HALF, L_NEXT = 0, 1, 2, 3
NODE_IN, NODE_OUT, NODE_BOUND, NODE_VALVE = 0, 1, 2, 3
BC_NOSLIP, BC_FREE = 0, 1


class Seg(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("posx", "posy", "posz", "endx", "endy", "endz", "size", "dir")]


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("fs3d_oracle.c", "fs3d_oracle_body.inc")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    ref_srcs = [os.path.join(_HERE, f) for f in ("ref_harness_f32.cpp", "ref_harness_f64.cpp", "ref_harness_cfg.cpp", "ref_harness_adi.cpp", "ref_harness_2d.cpp")]
    ref_sos = [os.path.join(_HERE, "_ref", f) for f in ("libref_pieces.so", "libref_config.so", "ref_adi_f32", "ref_adi_f64", "ref_stable2d")]
    ref_missing = os.path.isdir("/root/reference/src/Common") and (
        not all(os.path.exists(p) for p in ref_sos) or any(os.path.getmtime(s) > min(os.path.getmtime(p) for p in ref_sos) for s in ref_srcs))
    if force or stale or ref_missing:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


def _cap_threads():
    """The GPU box exposes every host core but grants a 16-core share: an uncapped OpenMP team
    oversubscribes it badly.  Cap the team before libgomp initialises."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(16, n))))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def num_threads():
    _cap_threads()
    return int(os.environ["OMP_NUM_THREADS"])


def lib():
    global _LIB
    if _LIB is None:
        _cap_threads()
        _LIB = C.CDLL(build())
        for sfx, ct in (("f32", C.c_float), ("f64", C.c_double)):
            P = C.POINTER(ct)
            f = lambda n: getattr(_LIB, "%s_%s" % (n, sfx))
            f("fs3d_oracle_create").restype = C.c_void_p
            f("fs3d_oracle_create").argtypes = [C.c_int] * 3 + [C.c_double] * 3
            f("fs3d_oracle_destroy").argtypes = [C.c_void_p]
            f("fs3d_oracle_set_params").argtypes = [C.c_void_p] + [C.c_double] * 4
            f("fs3d_oracle_set_nodes").argtypes = [C.c_void_p] + [C.c_void_p] * 7
            f("fs3d_oracle_create_segments").argtypes = [C.c_void_p]
            f("fs3d_oracle_num_segments").argtypes = [C.c_void_p, C.c_int]
            f("fs3d_oracle_num_segments").restype = C.c_int
            f("fs3d_oracle_segments").argtypes = [C.c_void_p, C.c_int]
            f("fs3d_oracle_segments").restype = C.POINTER(Seg)
            f("fs3d_oracle_warned_segs_per_row").argtypes = [C.c_void_p]
            f("fs3d_oracle_init_layers").argtypes = [C.c_void_p]
            f("fs3d_oracle_update_boundaries").argtypes = [C.c_void_p]
            f("fs3d_oracle_sweep").argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int]
            f("fs3d_oracle_merge").argtypes = [C.c_void_p, C.c_int, C.c_int]
            f("fs3d_oracle_eval_div_error").argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]
            f("fs3d_oracle_eval_div_error").restype = C.c_double
            f("fs3d_oracle_time_step").argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int,
                                                   C.POINTER(C.c_double)]
            f("fs3d_oracle_time_step").restype = C.c_int
            f("fs3d_oracle_get_layer").argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 3
            f("fs3d_oracle_get_field").argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
            f("fs3d_oracle_set_field").argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
            f("fs3d_oracle_diff_error").argtypes = [C.c_void_p]
            f("fs3d_oracle_diff_error").restype = C.c_double
            f("fs3d_oracle_tridiag").argtypes = [P] * 5 + [C.c_int]
            f("fs3d_oracle_fluid_params_normalized").argtypes = [C.c_double] * 3 + [P]
            f("fs3d_oracle_fluid_params_physical").argtypes = [C.c_double] * 5 + [P]
        _LIB.fs3d_oracle_align_by_32.argtypes = [C.c_int]
        _LIB.fs3d_oracle_align_by_32.restype = C.c_int
    return _LIB


def ref_lib():
    """oracle/_ref/libref_pieces.so: the reference's own Algorithms.h/Geometry.h, or None."""
    global _REF
    if _REF is None:
        build()
        p = os.path.join(_HERE, "_ref", "libref_pieces.so")
        if not os.path.exists(p):
            return None
        _REF = C.CDLL(p)
        pf, pd = C.POINTER(C.c_float), C.POINTER(C.c_double)
        _REF.ref_tridiag_f32.argtypes = [pf] * 5 + [C.c_int]
        _REF.ref_tridiag_f64.argtypes = [pd] * 5 + [C.c_int]
        _REF.ref_fluid_params_normalized_f32.argtypes = [C.c_double] * 3 + [pf]
        _REF.ref_fluid_params_physical_f32.argtypes = [C.c_double] * 5 + [pf]
        _REF.ref_align_by_32.argtypes = [C.c_int]
        _REF.ref_align_by_32.restype = C.c_int
        pi = C.POINTER(C.c_int)
        if hasattr(_REF, "ref_bbox2d_build"):
            _REF.ref_bbox2d_build.argtypes = [C.c_int, pi, pf, pf]
            _REF.ref_bbox3d_build.argtypes = [C.c_int, pi, pf, pf]
            _REF.ref_depth_resample.argtypes = [C.c_int] * 4 + [pf, pf]
    return _REF


def ref_config(path):
    """The reference's own Config::LoadFromFile (oracle/_ref/libref_config.so: Common/Config.h + LinuxIO.cpp compiled as they lie)
    on `path`, in a child process (the reference calls exit(0) on a config it rejects) -> dict, or None when it rejected the file /
    the library is not there."""
    build()
    so = os.path.join(_HERE, "_ref", "libref_config.so")
    if not os.path.exists(so):
        return None
    code = ("import ctypes as C, json, sys\n"
            "l = C.CDLL(%r)\n"
            "d = (C.c_double * 21)(); i = (C.c_int * 15)(); v = C.create_string_buffer(1024)\n"
            "l.ref_config_load(%r.encode(), d, i, v, 1024)\n"
            "print('CFG' + json.dumps({'d': list(d), 'i': list(i), 'vars': v.value.decode().split()}))\n") % (so, path)
    import json
    import sys
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("CFG")]
    if not line:
        return {"rejected": out.stdout.strip()}
    r = json.loads(line[0][3:])
    dn = ("R_specific k cv baseT bc_strength bc_inVx bc_inVy bc_inVz bc_inT viscosity density Re Pr lam depth_var frame_time dx dy dz depth _").split()
    inn = ("bc_noslip useNormalizedParams cycles time_steps out_time_steps outdimx outdimy outdimz num_global num_local problem_dim in_fmt "
           "out_fmt solver n_out_vars").split()
    res = dict(zip(dn, r["d"])); res.update(zip(inn, r["i"])); res["out_vars"] = r["vars"]
    return res


def _sfx(dtype):
    return "f32" if np.dtype(dtype) == np.float32 else "f64"


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def fluid_params(dtype, Re, Pr, lam):
    out = np.zeros(4, dtype=dtype)
    getattr(lib(), "fs3d_oracle_fluid_params_normalized_" + _sfx(dtype))(
        Re, Pr, lam, out.ctypes.data_as(C.POINTER(C.c_float if out.dtype == np.float32 else C.c_double)))
    return out


def tridiag(a, b, c, d):
    """Thomas solve with the oracle (copies inputs: the solver works in place)."""
    dt = a.dtype
    a, b, c, d = [np.array(v, dtype=dt, copy=True) for v in (a, b, c, d)]
    x = np.zeros_like(a)
    ct = C.c_float if dt == np.float32 else C.c_double
    getattr(lib(), "fs3d_oracle_tridiag_" + _sfx(dt))(*[v.ctypes.data_as(C.POINTER(ct)) for v in (a, b, c, d, x)],
                                                       len(a))
    return x


class Oracle:
    """One CPU solver instance (the reference's AdiSolver3D with backend == CPU)."""

    def __init__(self, nodes, params, dtype=np.float32):
        """nodes: cmc_fluid_solver_amd.grids.Nodes (SoA of the reference's Node array);
        params: (v_T, v_vis, t_vis, t_phi) already rounded to dtype."""
        self.dtype = np.dtype(dtype)
        self.sfx = _sfx(dtype)
        self.dims = (nodes.dimx, nodes.dimy, nodes.dimz)
        self.n = nodes.dimx * nodes.dimy * nodes.dimz
        self._f = lambda n: getattr(lib(), "%s_%s" % (n, self.sfx))
        self.h = C.c_void_p(self._f("fs3d_oracle_create")(nodes.dimx, nodes.dimy, nodes.dimz,
                                                           nodes.dx, nodes.dy, nodes.dz))
        self._f("fs3d_oracle_set_params")(self.h, *[float(p) for p in params])
        arrs = [np.ascontiguousarray(nodes.type, np.uint8), np.ascontiguousarray(nodes.bc_vel, np.uint8),
                np.ascontiguousarray(nodes.bc_temp, np.uint8)] + [
            np.ascontiguousarray(v, self.dtype) for v in (nodes.vx, nodes.vy, nodes.vz, nodes.T)]
        self._f("fs3d_oracle_set_nodes")(self.h, *[_ptr(a) for a in arrs])
        self._f("fs3d_oracle_create_segments")(self.h)
        self._f("fs3d_oracle_init_layers")(self.h)

    def close(self):
        if self.h:
            self._f("fs3d_oracle_destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def num_segments(self, d):
        return self._f("fs3d_oracle_num_segments")(self.h, d)

    def segments(self, d):
        n = self.num_segments(d)
        p = self._f("fs3d_oracle_segments")(self.h, d)
        return [(p[i].posx, p[i].posy, p[i].posz, p[i].endx, p[i].endy, p[i].endz, p[i].size) for i in range(n)]

    def update_boundaries(self):
        self._f("fs3d_oracle_update_boundaries")(self.h)

    def time_step(self, dt, G, L, compute_error=True):
        err = C.c_double(0.0)
        rc = self._f("fs3d_oracle_time_step")(self.h, dt, G, L, int(compute_error), C.byref(err))
        return rc, err.value

    def sweep(self, d, dt, l_cur, l_temp, l_next):
        self._f("fs3d_oracle_sweep")(self.h, d, dt, l_cur, l_temp, l_next)

    def merge(self, l_src, l_dest):
        self._f("fs3d_oracle_merge")(self.h, l_src, l_dest)

    def eval_div_error(self, layer=L_NEXT):
        cnt = C.c_longlong(0)
        e = self._f("fs3d_oracle_eval_div_error")(self.h, layer, C.byref(cnt))
        return e, cnt.value

    def get_field(self, layer, var):
        out = np.empty(self.dims, dtype=self.dtype)
        self._f("fs3d_oracle_get_field")(self.h, layer, var, _ptr(out))
        return out

    def set_field(self, layer, var, a):
        a = np.ascontiguousarray(a, self.dtype)
        assert a.size == self.n
        self._f("fs3d_oracle_set_field")(self.h, layer, var, _ptr(a))

    def get_layer_fields(self, layer):
        return [self.get_field(layer, v) for v in range(4)]

    def get_layer(self, outdims=(0, 0, 0)):
        od = [o or d for o, d in zip(outdims, self.dims)]
        outV = np.empty(od + [3], dtype=self.dtype)
        outT = np.empty(od, dtype=np.float64)
        self._f("fs3d_oracle_get_layer")(self.h, _ptr(outV), _ptr(outT), *outdims)
        return outV, outT
