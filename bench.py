#!/usr/bin/env python3
"""Benchmark of the FluidSolver3D step (BASELINE.json metric): Mcells/s of one
AdiSolver3D::TimeStep (+ UpdateBoundaries, + EvalDivError every 10th step, exactly the
reference's time loop, FluidSolver3D.cpp:226-262) on the synthetic 256^3 fp32 box.

  python bench.py --gpus N --steps K --warmup W

N = 1: one context on cuda:0.  N > 1 (launched by torch.distributed.run, one rank per
GPU): the 256^3 box is cut into N x-slabs (strong scaling); halo planes, the cross-slab
X solve and the 2-scalar error reduction go over RCCL inside libfs3d_hip.so; torch.distributed
only carries the ncclUniqueId and the barrier / max-over-ranks timing.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events on the
library's stream over the timed region; `cpu_baseline` times the CPU oracle (a port of
the reference's CPU path) on a bounded sample of the same workload on this host.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
RE, PR, LAMBDA = 200.0, 0.72, 1.4
NUM_GLOBAL, NUM_LOCAL = 4, 2   # data/3D/**/*config.txt: num_global 4, num_local 2
MGPU_TOL = 1e-6                # slab fields vs single-GPU fields (rel-L2): the bound of every one-card slab test


def cpu_baseline(n, dt, steps, dtype=np.float32):
    """The CPU oracle (port of the reference's CPU path, bit-equal to the reference's own binary on the fixtures of
    tests/golden/ref_*.npz) on the same workload, bounded sample.  Returns (json object, the oracle's fields after its steps)."""
    from oracle import oracle as O
    from cmc_fluid_solver_amd import capi, grids
    g = grids.box(n, h=1.0 / (n - 1))
    params = capi.fluid_params(dtype, RE, PR, LAMBDA)
    o = O.Oracle(g, params, dtype)
    o.update_boundaries(); o.time_step(dt, NUM_GLOBAL, NUM_LOCAL, False)   # warm-up (page faults, OpenMP team)
    t0 = time.perf_counter()
    for i in range(steps):
        o.update_boundaries()
        o.time_step(dt, NUM_GLOBAL, NUM_LOCAL, i % 10 == 0)
    sec = time.perf_counter() - t0
    fields = o.get_layer_fields(O.L_CUR)
    o.close()
    return {"value": round(n ** 3 * steps / sec / 1e6, 3), "unit": "Mcells/s", "cores": O.num_threads(),
            "kind": "port", "sample": "%d steps of the same %d^3 %s box (G=%d, L=%d), CPU oracle with OpenMP, "
            "%.1f s" % (steps, n, np.dtype(dtype).name, NUM_GLOBAL, NUM_LOCAL, sec)}, fields


def parity_check(n, dt, steps, dtype, cpu_fields, kernel, device):
    """Untimed: a second GPU context walks the SAME steps as the cpu_baseline leg from the same state (1 warm-up + `steps`);
    its fields against the CPU oracle's -- the tolerance the throughput number is quoted with (north_star: 1e-6 rel-L2)."""
    from cmc_fluid_solver_amd import capi, grids
    g = grids.box(n, h=1.0 / (n - 1))
    params = capi.fluid_params(dtype, RE, PR, LAMBDA)
    cpu = [np.asarray(f, np.float64) for f in cpu_fields]
    mask = g.type != grids.NODE_OUT

    def run(k):
        sv = capi.Solver(g, params, dtype, device=device)
        sv.set_option(capi.OPT_SWEEP_KERNEL, k)
        sv.UpdateBoundaries(); sv.TimeStep(dt, NUM_GLOBAL, NUM_LOCAL, False)
        for i in range(steps):
            sv.UpdateBoundaries()
            sv.TimeStep(dt, NUM_GLOBAL, NUM_LOCAL, i % 10 == 0)
        f = sv.download_layer(capi.LAYER_CUR)
        ran = sv.last_sweep_kernels()
        sv.close()
        return f, ran

    def rl2(a, b):
        d = np.sqrt(sum((((x.astype(np.float64) - y) * mask) ** 2).sum() for x, y in zip(a, b)))
        return float(d / np.sqrt(sum(((y * mask) ** 2).sum() for y in b)))

    got, ran = run(kernel)
    out = {"vs": "cpu oracle %s, %d steps, %d^3 (the oracle equals the reference binary bit for bit: tests/test_ref_golden.py)"
                 % (np.dtype(dtype).name, steps + 1, n),
           "rel_l2_velocity": rl2(got[:3], cpu[:3]), "rel_l2_T": rl2(got[3:], cpu[3:]),
           "bit_identical": bool(all(np.array_equal(a, b) for a, b in zip(got, cpu_fields))), "sweep_kernels": ran}
    if kernel != capi.SWEEP_EXACT:
        ex, ran_ex = run(capi.SWEEP_EXACT)
        out["exact_kernels_bit_identical"] = bool(all(np.array_equal(a, b) for a, b in zip(ex, cpu_fields)))
    return out


def self_launch(n):
    """`python bench.py --gpus N` started plainly: run the N ranks under torch.distributed.run as a CHILD process (never exec: this
    pool forbids replacing a process image once anything touched the GPU, and nothing has yet), relay rank 0's JSON line, return
    the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            print(ln)
        elif ln.strip():
            print(ln, file=sys.stderr)
    return p.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=256, help="box edge (default 256: BASELINE configs[2])")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--kernel", type=int, default=0, help="0 auto (partition kernels in fp32), 1 line, 2 pipe, 3 partition, 4 fastest bit-exact")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))         # before torch or the GPU is touched: the ranks are child processes

    import torch
    import torch.distributed as dist
    from cmc_fluid_solver_amd import capi, grids

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the HIP path)")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py --gpus %d: rank %d has no device (%d visible): one rank per GPU of ONE node" % (args.gpus, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # multi-rank runs have never met real wires: a rank that waits for ever in an exchange must not hold the
    # node until the driver's clock runs out -- after 15 minutes the rank reports and leaves
    if world > 1:
        import threading

        def _stuck():
            sys.stderr.write("bench.py rank %d: no result after 900 s -- an exchange between the ranks is stuck; giving up\n" % rank)
            sys.stderr.flush()
            os._exit(5)
        wd = threading.Timer(900.0, _stuck)
        wd.daemon = True
        wd.start()

    n = args.size
    dtype = np.float32 if args.dtype == "f32" else np.float64
    h = 1.0 / (n - 1)
    dt = 0.1                    # every shipped 3D config: duration 10 / time_steps 100 (FluidSolver3D.cpp:196)
    g = grids.box(n, h=h)
    params = capi.fluid_params(dtype, RE, PR, LAMBDA)
    from cmc_fluid_solver_amd.slab import slab_range

    def make_solver(grid, opts=None):
        """One context per rank on its x-slab; ranks joined by RCCL inside libfs3d_hip.so."""
        xa, xb = slab_range(grid.dimx, rank, world)
        sv = capi.Solver(grid, params, dtype, device=local_rank, x_range=(xa, xb))
        sv.set_option(capi.OPT_SWEEP_KERNEL, args.kernel)
        for k_, v_ in (opts or {}).items():
            sv.set_option(k_, v_)
        if world > 1:
            uid = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                buf = (capi.C.c_char * 128)()
                assert capi.load().fs3d_comm_unique_id(buf) == 0
                uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
            uid = uid.cuda()
            dist.broadcast(uid, 0)
            sv.comm_init(bytes(uid.cpu().numpy().tobytes()), rank, world)
        return sv, xa, xb

    # multi-GPU self-check (untimed): a 128^3 box stepped on N slabs against the single-GPU fields.  The default slab protocol (the
    # reduced-interface X solve + halo planes beside the interior sweep) has never run between real ranks:
    # if its fields are off, the check is repeated with the bit-exact rank pipeline and the exchange-first order, and the timed
    # run uses whichever passed; if neither does, the bench exits non-zero instead of printing a throughput of wrong fields.
    mgpu_check = None
    slab_opts = {}
    if world > 1:
        NC = 128
        gs = grids.box(NC, h=1.0 / (NC - 1))
        ref1 = errs1 = None
        if rank == 0:
            s1 = capi.Solver(gs, params, dtype, device=local_rank)
            s1.set_option(capi.OPT_SWEEP_KERNEL, args.kernel)
            errs1 = []
            for i in range(2):
                s1.UpdateBoundaries()
                errs1.append(s1.TimeStep(dt, NUM_GLOBAL, NUM_LOCAL, True))
            ref1 = np.stack(s1.download_layer(capi.LAYER_CUR))
            s1.close()

        def slab_check(opts):
            sv, xa, xb = make_solver(gs, opts)
            errs = []
            for i in range(2):
                sv.UpdateBoundaries()
                errs.append(sv.TimeStep(dt, NUM_GLOBAL, NUM_LOCAL, True))
            mine = np.stack(sv.download_layer(capi.LAYER_CUR))                    # [4, nx, NC, NC]
            ran = sv.last_sweep_kernels()
            pad = np.zeros((4, (NC + world - 1) // world + 1, NC, NC), dtype=dtype)
            pad[:, :xb - xa] = mine
            tl = [torch.empty(pad.shape, dtype=torch.float32 if dtype == np.float32 else torch.float64, device="cuda")
                  for _ in range(world)]
            dist.all_gather(tl, torch.from_numpy(pad).cuda())
            sv.close()
            ok = torch.zeros(1, dtype=torch.int32, device="cuda")
            res = None
            if rank == 0:
                full = np.concatenate([tl[r].cpu().numpy()[:, :slab_range(NC, r, world)[1] - slab_range(NC, r, world)[0]]
                                       for r in range(world)], axis=1)
                rl2 = float(np.linalg.norm(full.astype(np.float64) - ref1) / np.linalg.norm(ref1.astype(np.float64)))
                dre = float(abs(errs[-1] - errs1[-1]) / abs(errs1[-1]))
                res = {"grid": [NC, NC, NC], "steps": 2, "fields_bit_identical_to_single_gpu": bool(np.array_equal(full, ref1)),
                       "rel_l2_vs_single_gpu": rl2, "fields_match_single_gpu": bool(rl2 <= MGPU_TOL and dre <= 1e-4),
                       "tolerance": MGPU_TOL, "max_abs_diff": float(np.abs(full - ref1).max()), "div_error_rel_diff": dre,
                       "slab_options": opts or "default (reduced-interface X solve, halo planes beside the interior)",
                       "sweep_kernels_ran": ran}
                ok[0] = int(res["fields_match_single_gpu"])
            dist.broadcast(ok, 0)
            return res, bool(ok.item())

        mgpu_check, good = slab_check({})
        if not good:
            first = mgpu_check
            slab_opts = {capi.OPT_XSOLVE: capi.XSOLVE_PIPELINED, capi.OPT_OVERLAP: 0}
            mgpu_check, good = slab_check(slab_opts)
            if rank == 0:
                mgpu_check["default_protocol_failed"] = first
        if not good:
            if rank == 0:
                print(json.dumps({"error": "multi-GPU fields do not match the single-GPU fields", "multi_gpu_check": mgpu_check}))
            dist.barrier()
            dist.destroy_process_group()
            sys.exit(3)
        dist.barrier()

    s, x0, x1 = make_solver(g, slab_opts)

    def step(i):
        if i % 10 == 0:      # FluidSolver3D.cpp:242: computeError every 10th step
            s.UpdateBoundaries()
            s.TimeStep(dt, NUM_GLOBAL, NUM_LOCAL, True)
        else:
            s.time_step_async(dt, NUM_GLOBAL, NUM_LOCAL)   # UpdateBoundaries + TimeStep, enqueued

    def fence():
        s.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for i in range(args.warmup):
        step(i)
    fence()
    # HIP events around the kernel launches of the timed region -- of every 7th step (steps 0, 7, 14 of the default 20: 24 launches per sweep class): two events per launch cost a few microseconds
    # each (52 per step: measured 4-5 % of the step), the per-class averages need a sample, not every launch
    s.enable_timing(int(os.environ.get("FS3D_BENCH_EVENT_PERIOD", "7")))
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    sec = time.perf_counter() - t0
    ms_cls, n_cls = s.last_step_timing()
    s.enable_timing(False)
    err, _ = s.eval_div_error(capi.LAYER_CUR)
    kernels_ran = s.last_sweep_kernels()
    if world > 1:
        t = torch.tensor([sec], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sec = float(t.item())

    # what a plain device-to-device copy reaches on this card (read + write bytes per second): context for `peak`
    copy_gbps = None
    if world == 1:
        a = torch.empty(1 << 28, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)      # 1 GiB each
        b.copy_(a); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b.copy_(a)
        e1.record(); torch.cuda.synchronize()
        copy_gbps = round(10 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        del a, b

    if rank == 0:
        cells = n ** 3
        esize = 4 if dtype == np.float32 else 8
        # dominant kernel = the sweep class with the largest total device time
        names = ["sweep_Z", "sweep_Y", "sweep_X"]
        k = int(np.argmax(ms_cls[:3]))
        per_launch_ms = ms_cls[k] / max(1, n_cls[k])
        local_cells = (x1 - x0) * n * n
        ref_alg_bytes = 16 * esize * local_cells      # 16 words/cell/sweep (cur4+temp4 in, next4+temp4 out)
        # The fused time step does not store `next` in the local iterations whose `next` the following one overwrites
        # unread ((L-1) of every L launches of a class): those launches move 12 words/cell.  `achieved` is priced on
        # the launch mix actually issued, not on the 16-word figure.
        # (r3) The X sweep that closes a global iteration but the last does not store it either (the next iteration's Z sweep
        # overwrites it unread), and the step's very last X sweep stores `next` but no merged temp (the next step starts from
        # temp := cur): every X launch moves 12 words/cell.
        skip = (NUM_LOCAL - 1) / NUM_LOCAL
        skip_x = 1.0
        skip_k = skip_x if names[k] == "sweep_X" else skip
        words = 16.0 - 4.0 * skip_k
        alg_bytes = int(words * esize * local_cells)
        achieved = alg_bytes / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
        # HBM bytes per launch of the dominant kernel from the PMC passes (collected separately, profiles/)
        traffic = None
        try:
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            # measured once, in separate --pmc passes (tools/pmc_traffic.py): valid only for the kernel source it was measured with
            import hashlib
            src = b"".join(open(os.path.join(ROOT, "cmc_fluid_solver_amd", "csrc", f), "rb").read()
                           for f in ("kernels_part.hip", "fs3d_common.h"))
            if (pt.get("grid") == [n, n, n] and pt.get("dtype") == args.dtype and world == 1 and args.kernel == 0
                    and pt.get("kernel_source_sha16") == hashlib.sha256(src).hexdigest()[:16]):
                traffic = pt.get(names[k])
        except Exception:
            traffic = None
        out = {
            "metric": "Mcells/sec (FluidSolver3D step)", "value": round(cells * args.steps / sec / 1e6, 2),
            "unit": "Mcells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(sec / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "FluidSolver3D %d^3 %s empty box (shell NODE_BOUND, x=0 inflow valve U=1, x=max free "
                                   "outflow valve), Re 200 Pr 0.72 lambda 1.4, num_global 4, num_local 2, dt %.5g, "
                                   "UpdateBoundaries+TimeStep per step, EvalDivError every 10th step" % (n, args.dtype, dt),
                       "grid": [n, n, n], "parallelism": "x-slab x%d" % world, "sweep_kernel_requested": args.kernel,
                       "sweep_kernels_ran": kernels_ran,
                       "node_in_fraction": round(float((g.type == grids.NODE_IN).mean()), 4),
                       "final_div_error": err},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": names[k], "avg_launch_ms": round(per_launch_ms, 4),
                         "device_copy_GBps": copy_gbps,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "algorithmic_bytes_per_launch_all_stores": ref_alg_bytes,
                         "launches_moving_12_words_per_cell": {"sweep_Z": round(skip, 4), "sweep_Y": round(skip, 4), "sweep_X": round(skip_x, 4)},
                         "per_class_ms_per_launch": {nm: round(ms_cls[j] / max(1, n_cls[j]), 4)
                                                     for j, nm in enumerate(names + ["other"])},
                         "launches": dict(zip(names + ["other"], n_cls)),
                         "launches_timed": "HIP events around the launches of every %s-th step of the timed region" % os.environ.get("FS3D_BENCH_EVENT_PERIOD", "7"),
                         "step_frac_of_hbm_roofline_1760B": round(
                             (cells * 1760.0 * (esize / 4) * args.steps / sec / 1e9) / (HBM_PEAK_GBS * world), 4),
                         "step_frac_of_hbm_roofline_moved": round(
                             (cells * (1760.0 - 32.0 - 48.0 * NUM_GLOBAL - 16.0 * NUM_GLOBAL * (3 * NUM_LOCAL * skip + (skip_x - skip) * NUM_LOCAL)) * (esize / 4) * args.steps / sec / 1e9)
                             / (HBM_PEAK_GBS * world), 4)},
        }
        if mgpu_check is not None:
            out["multi_gpu_check"] = mgpu_check
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], cpu_fields = cpu_baseline(n, dt, args.cpu_steps, dtype)
            out["parity_check"] = parity_check(n, dt, args.cpu_steps, dtype, cpu_fields, args.kernel, local_rank)
        print(json.dumps(out))
    s.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
