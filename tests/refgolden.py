"""Access to tests/golden/ref_*.npz -- outputs of the reference's own CPU path (see tests/golden/make_ref_golden.py) -- and the
replay loop that drives an engine (the CPU oracle or the HIP library) through the reference driver's sequence of calls
(FluidSolver3D/FluidSolver3D.cpp:226-262) so that its fields can be held to the fixture step by step.

Nothing here touches /root/reference: the fixtures and tests/golden/inputs are all a GPU box has.
"""
import hashlib
import json
import os
import tempfile

import numpy as np

from cmc_fluid_solver_amd import grids, shape2d

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
INP = os.path.join(GOLD, "inputs")

ALL = [("u_bend", "f32"), ("u_bend", "f64"), ("box_pipe", "f32"), ("box_pipe", "f64"), ("non_uniform_pipe", "f32"),
       ("non_uniform_pipe", "f64"), ("box128", "f32"), ("box128", "f64"), ("box256", "f32"), ("non_uniform256", "f32"),
       ("box_pipe_g1l3", "f32"), ("box_pipe_g1l3", "f64"), ("box_pipe_g3l1", "f32"), ("box_pipe_g3l1", "f64"),
       ("heart_us", "f32"), ("box_pipe_3D", "f32"), ("tetra", "f32"), ("sphere_3D", "f32"), ("sphere_3D", "f64")]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


class Fixture:
    def __init__(self, name, prec):
        self.name, self.prec = name, prec
        self.z = np.load(os.path.join(GOLD, "ref_%s_%s.npz" % (name, prec)))
        self.meta = json.loads(str(self.z["meta"]))
        self.dtype = np.dtype(np.float32 if prec == "f32" else np.float64)
        self.dims = tuple(self.meta["dims"])
        self._cfg = None

    # ---- inputs ----------------------------------------------------------------------------------------------------------
    def config_path(self):
        f = tempfile.NamedTemporaryFile("w", suffix="_config.txt", delete=False)
        f.write(self.meta["config_text"])
        f.close()
        return f.name

    @property
    def data_path(self):
        return os.path.join(INP, self.meta["data"])

    def cfg(self):
        if self._cfg is None:
            p = self.config_path()
            try:
                self._cfg = shape2d.Config(p)
            finally:
                os.unlink(p)
        return self._cfg

    def nodes(self):
        """The reference's Node array as the fixture holds it (for f64: of the FTYPE-switched build, whose rasterisation runs in
        double and differs from the shipped float build in a few cells)."""
        m, z = self.meta, self.z
        ty = z["node_type"]
        c = m["node_consts"]
        fields = {}
        for k, comp in (("vx", 0), ("vy", 1), ("vz", 2), ("T", None)):
            a = np.empty(self.dims, np.float64)
            a[ty == 0] = c.get("in_" + k, 0.0)
            a[ty == 1] = c.get("out_" + k, 0.0)
            a[ty >= 2] = z["bnd_T"] if comp is None else z["bnd_vel"][:, comp]
            fields[k] = a
        return grids.Nodes(self.dims[0], self.dims[1], self.dims[2], m["dx"], m["dy"], m["dz"], ty, z["node_bc_vel"], z["node_bc_temp"],
                           fields["vx"], fields["vy"], fields["vz"], fields["T"])

    def loader(self, time=0.0):
        """(Nodes, Config, dt) through THIS repo's Shape2D / Shape3D loader from the same input files."""
        cfg = self.cfg()
        if cfg.in_fmt == "Shape3D":
            from cmc_fluid_solver_amd import shape3d
            text = open(self.data_path).read()
            sh = shape3d.Shape3D(shape3d.parse_shape3d(text), cfg.dx, cfg.dy, cfg.dz, self.meta["align"], time)
            nodes = shape3d.nodes_of(sh, cfg.dx, cfg.dy, cfg.dz, cfg.baseT)
            return nodes, cfg, cfg.frame_time / (len(sh.frames) * cfg.time_steps)       # GetCycleLength() = frame_time (Grid3D.cpp:303-336)
        p = self.config_path()
        try:
            return shape2d.load_case(self.data_path, p, align=self.meta["align"])
        finally:
            os.unlink(p)

    def params(self):
        return tuple(self.dtype.type(p) for p in self.meta["params"])

    def schedule(self, max_steps=None):
        """[(step, compute_error, output_layer)] -- FluidSolver3D.cpp:226-262 (substep counter restarts at every frame)."""
        cfg = self.cfg()
        n = self.meta["steps_run"] if max_steps is None else max_steps
        if cfg.in_fmt == "Shape3D":
            class Frames:                       # Grid3D.cpp:303-336: a Shape3D run has cycle length frame_time and frame 0 throughout
                num_frames = self.meta["frames"]
                cycle_length = staticmethod(lambda: cfg.frame_time)
                get_frame = staticmethod(lambda t: 0)
            g2 = Frames
        else:
            g2 = shape2d.Grid2D(shape2d.parse_shape2d(open(self.data_path).read()), cfg.dx, cfg.dy, 1.0, self.meta["align"])
        return [(s, ce, ol) for s, (t, i, fr, ce, ol) in enumerate(shape2d.time_loop(g2, cfg, n), 1)]

    def step_dt(self):
        dt = self.meta["dt"]
        return float(np.float32(dt)) if self.prec == "f32" else dt        # TimeStep((FTYPE)dt, ...)

    # ---- expected outputs ---------------------------------------------------------------------------------------------------
    def field(self, v, step):
        k = "%s_step%d" % (v, step)
        return self.z[k] if k in self.z else None

    def sample(self, v, step):
        k = "%s_sample%d" % (v, step)
        return self.z[k] if k in self.z else None


class OracleEngine:
    def __init__(self, fx, nodes=None):
        from oracle import oracle as O
        self.O = O
        self.o = O.Oracle(nodes if nodes is not None else fx.nodes(), fx.params(), fx.dtype)

    def update_boundaries(self):
        self.o.update_boundaries()

    def time_step(self, dt, G, L, ce):
        rc, err = self.o.time_step(dt, G, L, ce)
        assert rc == 0
        return err

    def fields(self):
        return self.o.get_layer_fields(self.O.L_CUR)

    def div_error(self):
        return self.o.eval_div_error(self.O.L_CUR)[0]

    def get_layer(self, od):
        return self.o.get_layer(od)

    def close(self):
        self.o.close()


class HipEngine:
    def __init__(self, fx, kernel=None, nodes=None):
        from cmc_fluid_solver_amd import capi
        self.capi = capi
        self.s = capi.Solver(nodes if nodes is not None else fx.nodes(), fx.params(), fx.dtype)
        if kernel is not None:
            self.s.set_option(capi.OPT_SWEEP_KERNEL, kernel)

    def update_boundaries(self):
        self.s.UpdateBoundaries()

    def time_step(self, dt, G, L, ce):
        return self.s.TimeStep(dt, G, L, ce)

    def fields(self):
        return self.s.download_layer(self.capi.LAYER_CUR)

    def div_error(self):
        return self.s.eval_div_error(self.capi.LAYER_CUR)[0]

    def get_layer(self, od):
        return self.s.GetLayer(od)

    def close(self):
        self.s.close()


def replay(fx, eng, on_step=None, on_layer=None, max_steps=None):
    """Drive `eng` as the reference driver does; returns the list of diffError values TimeStep reported (what it prints)."""
    cfg = fx.cfg()
    od = (cfg.outdimx, cfg.outdimy, cfg.outdimz)
    errs = []
    dumped = set(fx.meta["hashed_steps"]) | set(fx.meta["full_steps"])
    for step, ce, ol in fx.schedule(max_steps):
        eng.update_boundaries()
        errs.append(eng.time_step(fx.step_dt(), cfg.num_global, cfg.num_local, ce))
        if on_step is not None and step in dumped:
            on_step(step, eng)
        if ol:                                   # GetLayer is not read-only (OUT cells of the older layer := 99999): same call sites
            V, T = eng.get_layer(od)
            if on_layer is not None:
                on_layer(step, V, T)
    return errs


def rel_l2(a, b, mask=None):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if mask is not None:
        a, b = a[mask], b[mask]
    d = np.sqrt(((a - b) ** 2).sum())
    n = np.sqrt((b ** 2).sum())
    return d / n if n > 0 else d
