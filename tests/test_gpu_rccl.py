"""The RCCL wire of the slab protocol on ONE card.  A one-GPU box cannot host two ranks (RCCL refuses
two ranks on one device), so the slab protocol itself is tested through the in-process transport (test_gpu_slabs.py,
test_gpu_part.py) and the algebra over gloo (test_slab_gloo.py); what is checked here is that the RCCL calls of
cmc_fluid_solver_amd/csrc/fs3d_comm.hip work inside libfs3d_hip.so, in a process that also holds torch's own RCCL."""
import numpy as np
import pytest

from cmc_fluid_solver_amd import capi, grids

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_rccl_shapes_of_the_slab_protocol_on_one_rank(built, dtype):
    """fs3d_comm_selftest: a 1-rank communicator; grouped send/recv (halo planes, carries) on a second stream, all-gather
    (interface words of the cross-slab X solve), 2-double all-reduce (EvalDivError) -- every result verified in the library."""
    s = capi.Solver(grids.box(16, h=1.0 / 15), capi.fluid_params(dtype, 200.0, 0.72, 1.4), dtype)
    s.comm_selftest(256 * 256)              # one 256^2 halo plane
    s.comm_selftest(18 * 4096)
    # the context still steps afterwards
    s.UpdateBoundaries()
    assert np.isfinite(s.TimeStep(0.1, 1, 1, True))
    s.close()


def test_rccl_unique_id_and_single_rank_init(built):
    """fs3d_comm_unique_id hands out 128 bytes from RCCL; a world of one needs no communicator."""
    buf = (capi.C.c_char * 128)()
    assert capi.load().fs3d_comm_unique_id(buf) == 0
    assert any(b != 0 for b in buf.raw)
    s = capi.Solver(grids.box(16, h=1.0 / 15), capi.fluid_params(np.float32, 200.0, 0.72, 1.4), np.float32)
    s.comm_init(bytes(buf.raw), 0, 1)
    s.UpdateBoundaries()
    assert np.isfinite(s.TimeStep(0.1, 1, 1, True))
    s.close()


def test_selftest_beside_torch_rccl(built):
    """torch.distributed's nccl backend (= RCCL) initialised in the same process, as in bench.py --gpus N: both users of
    librccl work side by side."""
    import os
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        t = torch.ones(8, device="cuda")
        dist.all_reduce(t)
        s = capi.Solver(grids.box(16, h=1.0 / 15), capi.fluid_params(np.float32, 200.0, 0.72, 1.4), np.float32)
        s.comm_selftest(1 << 16)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        assert float(t.sum()) == 8.0
        s.close()
    finally:
        dist.destroy_process_group()
