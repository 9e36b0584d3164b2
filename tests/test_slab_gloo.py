"""Multi-rank protocol of the cross-slab X sweep, on CPU with torch.distributed/gloo, world_size 2
and 3: each rank owns a slab of the lines, forward carries travel rank r -> r+1, back-substitution
carries r+1 -> r (the order fs3d_hip.hip:xsweep_multi issues its RCCL send/recv).  The slabbed result
must equal the unsplit Thomas solve (the oracle's, pinned to the reference's Algorithms.h) bit for bit.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cmc_fluid_solver_amd import slab  # noqa: E402


def _make_system(n, nlines, dtype, seed):
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1, 1, (n, nlines)).astype(dtype)
    c = rng.uniform(-1, 1, (n, nlines)).astype(dtype)
    b = rng.uniform(2.1, 4, (n, nlines)).astype(dtype)
    d = rng.uniform(-5, 5, (n, nlines)).astype(dtype)
    a[0] = 0
    c[n - 1] = 0                      # Algorithms.h:23
    return a, b, c, d


def _worker(rank, world, port, n, nlines, dtype_name, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dtype = np.dtype(dtype_name)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    a, b, c, d = _make_system(n, nlines, dtype, seed=7)          # every rank holds the whole grid (as the reference does)
    x0, x1 = slab.slab_range(n, rank, world)
    sl = slice(x0, x1)
    carry = None
    if rank > 0:
        buf = torch.empty((2, nlines), dtype=tdt)
        dist.recv(buf, src=rank - 1)
        carry = (buf[0].numpy().copy(), buf[1].numpy().copy())
    cp, dp, out = slab.thomas_forward_slab(a[sl], b[sl], c[sl], d[sl], carry)
    if rank < world - 1:
        dist.send(torch.from_numpy(np.stack(out)), dst=rank + 1)
    xc = None
    if rank < world - 1:
        buf = torch.empty((nlines,), dtype=tdt)
        dist.recv(buf, src=rank + 1)
        xc = buf.numpy().copy()
    x, first = slab.thomas_backward_slab(cp, dp, xc)
    if rank > 0:
        dist.send(torch.from_numpy(first), dst=rank - 1)
    np.save(os.path.join(out_dir, "x_%d.npy" % rank), x)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("dtype_name", ["float32", "float64"])
def test_slab_pipelined_thomas_equals_unsplit(tmp_path, world, dtype_name):
    from oracle import oracle as O
    n, nlines = 37, 24
    port = 29600 + world * 10 + (0 if dtype_name == "float32" else 1)
    mp.spawn(_worker, args=(world, port, n, nlines, dtype_name, str(tmp_path)), nprocs=world, join=True)
    x = np.concatenate([np.load(tmp_path / ("x_%d.npy" % r)) for r in range(world)], axis=0)
    a, b, c, d = _make_system(n, nlines, np.dtype(dtype_name), seed=7)
    for line in range(nlines):
        ref = O.tridiag(a[:, line], b[:, line], c[:, line], d[:, line])
        assert np.array_equal(x[:, line], ref), "line %d differs from the unsplit solve" % line


def test_slab_ranges_cover_the_grid():
    for dimx in (7, 64, 255, 256):
        for nr in (1, 2, 3, 8):
            rs = [slab.slab_range(dimx, r, nr) for r in range(nr)]
            assert rs[0][0] == 0 and rs[-1][1] == dimx
            assert all(rs[i][1] == rs[i + 1][0] for i in range(nr - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1


# ---- reduced-interface form of the cross-slab X sweep (fs3d_hip.hip: xsweep_reduced) over gloo ----------------------
def _worker_reduced(rank, world, port, n, nlines, dtype_name, out_dir):
    """What every rank does: eliminate its own slab (k_xiface), ONE all-gather of the interface coefficients, the small
    R x R solve per line (k_xreduce), back-substitution on its own slab with the two boundary values given."""
    from cmc_fluid_solver_amd import partition as pt
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dtype = np.dtype(dtype_name)
    a, b, c, d = _make_system(n, nlines, dtype, seed=11)
    x0, x1 = slab.slab_range(n, rank, world)
    sl = slice(x0, x1)
    co = pt.chunk_eliminate(a[sl], b[sl], c[sl], d[sl])
    keys = ["A", "Bp", "cl", "Dp", "Vf", "Wf", "Gf"]
    mine = torch.from_numpy(np.stack([co[k] for k in keys]))                 # [7, nlines]: the slab's interface words
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)                                          # the sweep's only exchange
    cos = [{k: g[i].numpy() for i, k in enumerate(keys)} for g in gathered]
    X = pt.thomas(*pt.reduced_rows(cos))                                     # every rank solves the R x R systems redundantly
    x = pt.chunk_backsub(a[sl], b[sl], c[sl], d[sl], X[rank - 1] if rank else None, X[rank])
    np.save(os.path.join(out_dir, "xr_%d.npy" % rank), x)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("dtype_name", ["float32", "float64"])
def test_slab_reduced_interface_solve_equals_thomas(tmp_path, world, dtype_name):
    from oracle import oracle as O
    n, nlines = 41, 16
    port = 29700 + world * 10 + (0 if dtype_name == "float32" else 1)
    mp.spawn(_worker_reduced, args=(world, port, n, nlines, dtype_name, str(tmp_path)), nprocs=world, join=True)
    x = np.concatenate([np.load(tmp_path / ("xr_%d.npy" % r)) for r in range(world)], axis=0)
    a, b, c, d = _make_system(n, nlines, np.dtype(dtype_name), seed=11)
    ref = np.stack([O.tridiag(a[:, l].astype(np.float64), b[:, l].astype(np.float64), c[:, l].astype(np.float64), d[:, l].astype(np.float64))
                    for l in range(nlines)], axis=1)
    err = np.linalg.norm(x - ref) / np.linalg.norm(ref)
    assert err <= (1e-14 if dtype_name == "float64" else 5e-7), err


# ---- (r3) the interface solve distributed over the ranks: two all-to-alls instead of one all-gather ---------------------------------
def _alltoall(send, rank, world):
    """one grouped exchange of point-to-point transfers -- one send and one receive per peer, the own block copied (fs3d_comm_alltoall;
    gloo has no all_to_all of its own)"""
    recv = [torch.empty_like(t) for t in send]
    recv[rank].copy_(send[rank])
    reqs = [dist.isend(send[r], r) for r in range(world) if r != rank] + [dist.irecv(recv[r], r) for r in range(world) if r != rank]
    for q in reqs:
        q.wait()
    return recv


def _worker_a2a(rank, world, port, n, nlines, dtype_name, out_dir):
    """fs3d_hip.hip: xsweep_reduced with FS3D_XSOLVE 3 -- rank r owns the lines [r Lp, (r+1) Lp): all-to-all #1 brings it every rank's
    interface words of ITS lines, it solves their R x R systems once, all-to-all #2 hands every rank the value below and above its slab."""
    from cmc_fluid_solver_amd import partition as pt
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dtype = np.dtype(dtype_name)
    a, b, c, d = _make_system(n, nlines, dtype, seed=11)
    x0, x1 = slab.slab_range(n, rank, world)
    sl = slice(x0, x1)
    co = pt.chunk_eliminate(a[sl], b[sl], c[sl], d[sl])
    keys = ["A", "Bp", "cl", "Dp", "Vf", "Wf", "Gf"]
    lp = -(-nlines // world)                                                  # lines per owner (the last block is padded)
    mine = np.zeros((7, world * lp), dtype)
    mine[:, :nlines] = np.stack([co[k] for k in keys])
    send = [torch.from_numpy(np.ascontiguousarray(mine[:, r * lp:(r + 1) * lp])) for r in range(world)]      # k_xpack
    recv = _alltoall(send, rank, world)                                       # #1: the words of MY lines from every rank
    cos = [{k: g[i].numpy() for i, k in enumerate(keys)} for g in recv]
    nown = max(0, min(lp, nlines - rank * lp))
    X = np.zeros((world, lp), dtype)
    if nown:
        own = [{k: v[:nown] for k, v in cr.items()} for cr in cos]
        X[:, :nown] = pt.thomas(*pt.reduced_rows(own))                        # k_xreduce_a2a: every system once
    # the two boundary values of rank r on my lines: X_{r-1} and X_r (chunk_backsub derives x_first of the slab above from X_r itself)
    out = [torch.from_numpy(np.stack([X[r - 1] if r else np.zeros(lp, dtype), X[r]])) for r in range(world)]
    back = _alltoall(out, rank, world)                                        # #2
    bv = np.concatenate([t.numpy() for t in back], axis=1)[:, :nlines]        # k_xunpack: [2, nlines]
    x = pt.chunk_backsub(a[sl], b[sl], c[sl], d[sl], bv[0] if rank else None, bv[1])
    np.save(os.path.join(out_dir, "xa_%d.npy" % rank), x)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [3, 8])
def test_distributed_interface_solve_over_gloo(tmp_path, world):
    """The all-to-all protocol between real processes (gloo here, RCCL on the GPUs): bit-identical to the all-gather form, for a line
    count that is no multiple of the ranks (13 lines: the last owners hold padded or empty blocks)."""
    n, nlines, dtype_name = 41, 13, "float64"
    mp.spawn(_worker_reduced, args=(world, 29800 + world, n, nlines, dtype_name, str(tmp_path)), nprocs=world, join=True)
    mp.spawn(_worker_a2a, args=(world, 29850 + world, n, nlines, dtype_name, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("xr_%d.npy" % r)), np.load(tmp_path / ("xa_%d.npy" % r))), r
