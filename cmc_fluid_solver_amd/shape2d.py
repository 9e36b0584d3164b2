"""Shape2D input surface: the reference's `in_fmt Shape2D` text geometry extruded in depth.

Restates, operation for operation (FTYPE = float, i.e. np.float32 arithmetic where the reference
computes in FTYPE), the part of the reference that turns `data/3D/**/*_2D_data.txt` + config values into
the Node array the solver consumes:
  Grid2D::LoadFromFile / ComputeBorderVelocities / Init / Prepare / ComputeSubframe / Build /
  RasterLine / FloodFill            (FluidSolver2D/Grid2D.cpp:109-372, 268-372, 376-480)
  BBox2D::Build                     (Common/Geometry.h:455-486)
  Grid3D::LoadFromFile / Prepare2D  (FluidSolver3D/Grid3D.cpp:488-513, 608-668)
Multi-frame inputs (moving walls, e.g. data/3D/large_tests/heart_us): all frames are read, the border velocities of every
frame come from the frame before it (ComputeBorderVelocities), the bounding box covers all frames, and the grid is built from
Prepare(time) = ComputeSubframe(frame, substep).  The reference's 3D time loop prepares the grid once, at time 0 (the
per-step `grid->Prepare(t)` is commented out, FluidSolver3D.cpp:237): the geometry is frame 0 with frame 0's border velocities.
Pins: tests/test_grid_loader.py checks grid dims and NODE_IN counts against the reference outputs
recorded in SURVEY.md sections 8c/8d.
"""
import math

import numpy as np

from .grids import BC_FREE, BC_NOSLIP, NODE_BOUND, NODE_IN, NODE_OUT, NODE_VALVE, Nodes

F = np.float32
GRID_SCALE_FACTOR = F(0.001)      # Grid2D.h:31
BBOX_PADDING = F(0.02)            # Geometry.h:24
INF = F(1e10)                     # Geometry.h:22


def align_by_32(n):               # Geometry.h:564-568
    return n if (n & 31) == 0 else (((n >> 5) + 1) << 5)


def _read_point(tok):
    """ReadPoint2D (IO.h:511-540): 'x y' with ',' accepted as decimal mark; (FTYPE)atof."""
    return F(float(tok[0].replace(",", "."))), F(float(tok[1].replace(",", ".")))


def parse_shape2d(text):
    """Grid2D::LoadFromFile (Grid2D.cpp:268-319).  CRLF input is accepted (the run scripts strip \\r)."""
    lines = [ln.strip() for ln in text.replace("\r", "").split("\n") if ln.strip()]
    it = iter(lines)
    num_frames = int(next(it))
    frames = []
    for _ in range(num_frames):
        dur = F(float(next(it)))
        nshapes = int(next(it))
        shapes = []
        for _ in range(nshapes):
            npts = int(next(it))
            pts = []
            for _ in range(npts):
                x, y = _read_point(next(it).split())
                pts.append((F(x * GRID_SCALE_FACTOR), F(y * GRID_SCALE_FACTOR)))
            word = next(it).split()                                # "Passive" | "Motion" + "vx vy" on the same or on the next line
            active = word[0][0] == "M"
            vx, vy = F(0), F(0)
            if active:
                vx, vy = _read_point(word[1:3] if len(word) >= 3 else next(it).split())
            vel = [(F(vx * GRID_SCALE_FACTOR), F(vy * GRID_SCALE_FACTOR)) for _ in range(npts)]
            shapes.append({"points": pts, "vel": vel, "active": active})
        frames.append({"duration": float(dur), "shapes": shapes})
    return frames


class Grid2D:
    """FluidSolver2D::Grid2D for one frame (Grid2D.h:42-104)."""

    def __init__(self, frames, dx, dy, startT, align, time=0.0):
        self.dx, self.dy, self.startT = dx, dy, startT
        self.frames = frames
        self.num_frames = nf = len(frames)
        self.duration = frames[0]["duration"]
        # ComputeBorderVelocities(j) for every frame j, in order (Grid2D.cpp:375-396, called at :365-366): the velocities of
        # frame j+1 -- passive shapes: (P[j+1] - P[j]) / duration[j]; active ones add (P[j] - P[j+1]) / duration[j].
        # One frame: next == frame, every difference is 0: passive shapes at rest, active ones keep theirs.
        for j in range(nf):
            nxt = frames[(j + 1) % nf]
            m = 1.0 / frames[j]["duration"]
            for sa, sb in zip(frames[j]["shapes"], nxt["shapes"]):
                if not sa["active"]:
                    sb["vel"] = [(F(F(pb[0] - pa[0]) * F(m)), F(F(pb[1] - pa[1]) * F(m))) for pa, pb in zip(sa["points"], sb["points"])]
                else:
                    sb["vel"] = [(F(v[0] + F(F(pa[0] - pb[0]) * F(m))), F(v[1] + F(F(pa[1] - pb[1]) * F(m))))
                                 for v, pa, pb in zip(sb["vel"], sa["points"], sb["points"])]
        # BBox2D::Build over all frames (Geometry.h:455-486)
        xs = [p[0] for fr in frames for sh in fr["shapes"] for p in sh["points"]]
        ys = [p[1] for fr in frames for sh in fr["shapes"] for p in sh["points"]]
        pminx, pminy, pmaxx, pmaxy = min(xs + [INF]), min(ys + [INF]), max(xs + [F(-INF)]), max(ys + [F(-INF)])
        wx, wy = F(pmaxx - pminx), F(pmaxy - pminy)
        pminx, pminy = F(pminx - F(wx * BBOX_PADDING)), F(pminy - F(wy * BBOX_PADDING))
        pmaxx, pmaxy = F(pmaxx + F(wx * BBOX_PADDING)), F(pmaxy + F(wy * BBOX_PADDING))
        self.bbox = (pminx, pminy, pmaxx, pmaxy)
        # Grid2D::Init (Grid2D.cpp:212-246): physical -> grid coordinates for the points of every frame
        self.dimx = int(math.ceil(float(F(pmaxx - pminx)) / dx)) + 1
        self.dimy = int(math.ceil(float(F(pmaxy - pminy)) / dy)) + 1
        if align:
            self.dimx, self.dimy = align_by_32(self.dimx), align_by_32(self.dimy)
        fdx, fdy = F(dx), F(dy)
        for fr in frames:
            for sh in fr["shapes"]:
                sh["gpoints"] = [(F(F(p[0] - pminx) / fdx), F(F(p[1] - pminy) / fdy)) for p in sh["points"]]
        self.prepare(time)

    # ---- frames in time (Grid2D.cpp:447-519) ---------------------------------------------------------------------
    def cycle_length(self):
        return float(sum(fr["duration"] for fr in self.frames))            # GetCycleLenght

    def _locate(self, time):
        a = [0.0]
        for fr in self.frames:
            a.append(a[-1] + fr["duration"])
        r = math.fmod(time, a[-1])
        frame = 0
        for i in range(1, self.num_frames):
            if a[i] < r:
                frame = i
        return frame, r, a

    def get_frame(self, time):
        return self._locate(time)[0]                                         # GetFrame

    def layer_time(self, time):
        frame, r, a = self._locate(time)
        return float(F(a[frame + 1] - r))                                    # GetLayerTime

    def prepare(self, time):
        """Grid2D::Prepare(time) -> ComputeSubframe(frame, substep) -> Build (Grid2D.cpp:398-461)."""
        frame, r, a = self._locate(time)
        sub = (r - a[frame]) / (a[frame + 1] - a[frame])
        f0, f1 = self.frames[frame], self.frames[(frame + 1) % self.num_frames]
        s, i_s = F(sub), F(1 - sub)
        self.shapes = []
        for sa, sb in zip(f0["shapes"], f1["shapes"]):
            g = [(F(F(pa[0] * i_s) + F(pb[0] * s)), F(F(pa[1] * i_s) + F(pb[1] * s))) for pa, pb in zip(sa["gpoints"], sb["gpoints"])]
            v = [(F(F(va[0] * i_s) + F(vb[0] * s)), F(F(va[1] * i_s) + F(vb[1] * s))) for va, vb in zip(sa["vel"], sb["vel"])]
            self.shapes.append({"gpoints": g, "vel": v, "active": sa["active"]})
        self.build()

    def _raster_line(self, p1, p2, v1, v2, color):
        """Grid2D::RasterLine (Grid2D.cpp:117-153), bc_noslip == true (Grid3D.cpp:28)."""
        ox, oy = F(p2[0] - p1[0]), F(p2[1] - p1[1])
        steps = int(max(abs(ox), abs(oy))) + 1
        dpx, dpy = F(ox / F(steps)), F(oy / F(steps))
        dvx, dvy = F(F(v2[0] - v1[0]) / F(steps)), F(F(v2[1] - v1[1]) / F(steps))
        px, py, vx, vy = p1[0], p1[1], v1[0], v1[1]
        for _ in range(steps + 1):
            x, y = int(px), int(py)
            self.cell[x, y] = color
            self.velx[x, y], self.vely[x, y] = vx, vy
            self.T[x, y] = F(self.startT)
            px, py, vx, vy = F(px + dpx), F(py + dpy), F(vx + dvx), F(vy + dvy)

    def build(self):
        """Grid2D::Build (Grid2D.cpp:248-285) + FloodFill (:167-210)."""
        nx, ny = self.dimx, self.dimy
        self.cell = np.full((nx, ny), NODE_IN, np.uint8)
        self.velx = np.zeros((nx, ny), np.float32)
        self.vely = np.zeros((nx, ny), np.float32)
        self.T = np.zeros((nx, ny), np.float32)
        for active, color in ((True, NODE_VALVE), (False, NODE_BOUND)):
            for sh in self.shapes:
                if sh["active"] != active:
                    continue
                g, v = sh["gpoints"], sh["vel"]
                for i in range(len(g) - 1):
                    self._raster_line(g[i], g[i + 1], v[i], v[i + 1], color)
        # flood fill NODE_OUT from (0,0) through NODE_IN cells, 4-neighbourhood
        stack = [(0, 0)]
        self.cell[0, 0] = NODE_OUT
        while stack:
            i, j = stack.pop()
            for di, dj in ((-1, 0), (1, 0), (0, -1), (0, 1)):
                a, b = i + di, j + dj
                if 0 <= a < nx and 0 <= b < ny and self.cell[a, b] == NODE_IN:
                    self.cell[a, b] = NODE_OUT
                    stack.append((a, b))
        inout = (self.cell == NODE_IN) | (self.cell == NODE_OUT)
        self.velx[inout], self.vely[inout], self.T[inout] = 0, 0, F(self.startT)


def load_shape2d(path_or_text, dx, dy, dz, depth, depth_var=0.0, baseT=1.0, align=True, is_text=False, time=0.0):
    """Grid3D(dx,dy,dz,depth,depth_var,baseT) + LoadFromFile + Prepare2D(time) -> (Nodes, Grid2D)."""
    text = path_or_text if is_text else open(path_or_text, "r").read()
    g2 = Grid2D(parse_shape2d(text), dx, dy, baseT, align, time)
    dimx, dimy = g2.dimx, g2.dimy
    active_dimz = int(math.ceil(depth / dz)) + 1                     # Grid3D.cpp:503-505
    dimz = align_by_32(active_dimz) if align else active_dimz
    sh = (dimx, dimy, dimz)
    # memset(nodes, 0): type NODE_IN(0), bc NOSLIP(0), v = 0, T = 0   (Grid3D.cpp:612)
    typ = np.zeros(sh, np.uint8); bcv = np.zeros(sh, np.uint8); bct = np.zeros(sh, np.uint8)
    vx = np.zeros(sh, np.float32); vy = np.zeros(sh, np.float32); vz = np.zeros(sh, np.float32); T = np.zeros(sh, np.float32)

    def set_bound(sel, bv, bt, ux, uy, t, ntype=NODE_BOUND):         # Node::SetBound, Grid3D.h:80-87
        typ[sel] = ntype; bcv[sel] = bv; bct[sel] = bt; vx[sel] = ux; vy[sel] = uy; vz[sel] = 0; T[sel] = t

    out2 = g2.cell == NODE_OUT
    typ[out2, :] = NODE_OUT
    height = max(active_dimz - 2 - 2, 0)
    for i in range(dimx):
        for j in range(dimy):
            c = g2.cell[i, j]
            if c == NODE_OUT:
                continue
            typ[i, j, active_dimz - 1:] = NODE_OUT                                       # :626-628
            set_bound((i, j, active_dimz - 2), BC_NOSLIP, BC_FREE, 0, 0, F(baseT))      # :629
            x = -1 + 2 * float(i) / dimx
            y = -1 + 2 * float(j) / dimy
            z = 1.0 - (x * x + y * y) * 0.5
            bottom = 1 + int(depth_var * z * height)                                    # :632-636
            typ[i, j, 0] = NODE_OUT
            set_bound((i, j, slice(1, bottom + 1)), BC_NOSLIP, BC_FREE, 0, 0, F(baseT))  # :638-639
            ks = slice(bottom + 1, active_dimz - 2)
            if c == NODE_BOUND:                                                         # :646-648
                set_bound((i, j, ks), BC_NOSLIP, BC_FREE, g2.velx[i, j], g2.vely[i, j], g2.T[i, j])
            elif c == NODE_VALVE:                                                       # :649-655
                if g2.velx[i, j] == 0 and g2.vely[i, j] == 0:
                    set_bound((i, j, ks), BC_FREE, BC_FREE, g2.velx[i, j], g2.vely[i, j], g2.T[i, j], NODE_VALVE)
                else:
                    set_bound((i, j, ks), BC_NOSLIP, BC_NOSLIP, g2.velx[i, j], g2.vely[i, j], g2.T[i, j], NODE_VALVE)
            else:                                                                        # NODE_IN, :656-659
                typ[i, j, ks] = NODE_IN
                T[i, j, ks] = F(baseT)
    nodes = Nodes(dimx, dimy, dimz, dx, dy, dz, typ, bcv, bct,
                  vx.astype(np.float64), vy.astype(np.float64), vz.astype(np.float64), T.astype(np.float64))
    return nodes, g2


def read_config(path):
    """The whitespace `key value` pairs of a reference config file (Config.h:195-245), as a dict of strings."""
    toks = open(path).read().replace("\r", "").split()
    cfg, i = {}, 0
    while i + 1 < len(toks):
        if toks[i] == "out_vars":
            n = int(toks[i + 1])
            cfg["out_vars"] = toks[i + 2:i + 2 + n]
            i += 2 + n
        else:
            cfg[toks[i]] = toks[i + 1]
            i += 2
    return cfg


class Config:
    """The reference's static Config (Common/Config.h:76-271): defaults, `key value` parsing with every
    real number read through "%f" into a float and widened (ReadDouble, :116-135), and the same
    validation (ValueError where the reference prints a message and calls exit(0))."""

    def __init__(self, path=None):
        self.R_specific, self.k, self.cv, self.baseT = 461.495, 0.6, 4200.0, 1.0
        self.bc_noslip, self.bc_strength, self.bc_inV, self.bc_inT = True, 0.5, (0.0, 0.0, 0.0), 1.0
        self.useNormalizedParams, self.viscosity, self.density = False, 0.05, 1000.0
        self.Re = self.Pr = self.lam = -1.0
        self.depth_var = 0.0
        self.cycles, self.time_steps, self.out_time_steps = 1, 50, 10
        self.outdimx = self.outdimy = self.outdimz = 50
        self.out_vars = []
        self.num_global, self.num_local = 2, 1
        self.problem_dim = self.in_fmt = self.out_fmt = self.solver = None
        self.frame_time = self.dx = self.dy = self.dz = self.depth = -1.0
        if path is not None:
            self.load(path)

    @staticmethod
    def _f(tok):
        return float(np.float32(float(tok)))

    def load(self, path):
        toks = open(path).read().replace("\r", "").split()
        i = 0
        real = {"viscosity": "viscosity", "density": "density", "bc_strenght": "bc_strength", "bc_initT": "bc_inT",
                "grid_dx": "dx", "grid_dy": "dy", "grid_dz": "dz", "frame_time": "frame_time", "depth": "depth",
                "depth_var": "depth_var"}
        ints = {"cycles": "cycles", "time_steps": "time_steps", "out_time_steps": "out_time_steps",
                "out_gridx": "outdimx", "out_gridy": "outdimy", "out_gridz": "outdimz",
                "num_global": "num_global", "num_local": "num_local"}
        while i < len(toks):
            k = toks[i]; i += 1
            if k in real:
                setattr(self, real[k], self._f(toks[i])); i += 1
            elif k in ints:
                setattr(self, ints[k], int(toks[i])); i += 1
            elif k in ("Re", "Pr", "lambda"):
                self.useNormalizedParams = True
                setattr(self, "lam" if k == "lambda" else k, self._f(toks[i])); i += 1
            elif k == "dimension":
                self.problem_dim = toks[i]; i += 1
            elif k == "in_fmt":
                self.in_fmt = toks[i]; i += 1
            elif k == "out_fmt":
                self.out_fmt = toks[i]; i += 1
            elif k == "solver":
                self.solver = toks[i]; i += 1
            elif k == "bc_type":
                self.bc_noslip = toks[i][0] in "Nn"; i += 1
            elif k == "bc_initv":
                self.bc_inV = tuple(self._f(t) for t in toks[i:i + 3]); i += 3
            elif k == "out_vars":
                n = int(toks[i]); self.out_vars = toks[i + 1:i + 1 + n]; i += 1 + n
        # Config.h:249-270
        if self.problem_dim is None: raise ValueError("must specify problem dimension!")
        if self.solver is None: raise ValueError("must specify solver!")
        if self.out_fmt is None: raise ValueError("must specify output format!")
        if self.dx < 0: raise ValueError("cannot find dx!")
        if self.dy < 0: raise ValueError("cannot find dy!")
        if self.problem_dim == "3D":
            if not self.out_vars: raise ValueError("must output at least 1 var!")
            if self.in_fmt is None: raise ValueError("must specify input format!")
            if self.dz < 0: raise ValueError("cannot find dz!")
            if self.in_fmt == "Shape2D" and self.depth < 0: raise ValueError("cannot find depth!")
        if self.useNormalizedParams and (self.Re < 0 or self.Pr < 0 or self.lam < 0):
            raise ValueError("must specify Re, Pr and lambda!")
        return self


def load_case(data_path, config_path, align=True):
    """What FluidSolver3D.cpp:107-200 sets up for a Shape2D run: (Nodes, Config, dt)."""
    cfg = Config(config_path)
    if cfg.in_fmt != "Shape2D":
        raise NotImplementedError("in_fmt %s: only Shape2D inputs are supported" % cfg.in_fmt)
    nodes, g2 = load_shape2d(data_path, cfg.dx, cfg.dy, cfg.dz, cfg.depth, cfg.depth_var, cfg.baseT, align)
    cfg.grid2d = g2
    dt = g2.cycle_length() / (g2.num_frames * cfg.time_steps)          # length / (frames * time_steps), FluidSolver3D.cpp:194-196
    return nodes, cfg, dt


def time_loop(g2, cfg, max_steps=-1):
    """The reference's 3D time loop as data (FluidSolver3D.cpp:193-266): yields (t, i, frame, compute_error, output_layer) per
    step.  The substep counter i restarts at every frame change; the error is evaluated when i % 10 == 0 or on the last step,
    a result layer is written when i % out_time_steps == 0."""
    length = g2.cycle_length()
    dt = length / (g2.num_frames * cfg.time_steps)
    finaltime = length * cfg.cycles
    t, i, last, n = dt, 0, -1, 0
    while t < finaltime and (max_steps < 0 or n < max_steps):
        frame = g2.get_frame(t)
        if frame != last:
            last, i = frame, 0
        yield t, i, frame, (i % 10 == 0) or (t + dt >= finaltime), i % cfg.out_time_steps == 0
        t += dt; i += 1; n += 1


def node_in_count(data_path, dx, dy, dz, depth, align=True):
    """NODE_IN count of the extruded grid without building it (depth_var = 0): the count the reference prints
    as 'NODE_IN points' (FluidSolver3D.cpp:163-170)."""
    g2 = Grid2D(parse_shape2d(open(data_path).read()), dx, dy, 1.0, align)
    active_dimz = int(math.ceil(depth / dz)) + 1
    dimz = align_by_32(active_dimz) if align else active_dimz
    return (g2.dimx, g2.dimy, dimz), int((g2.cell == NODE_IN).sum()) * max(active_dimz - 4, 0)
