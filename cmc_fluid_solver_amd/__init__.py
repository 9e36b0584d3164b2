"""MI355X-native FluidSolver3D hot path (ADI line sweeps + merge + divergence check).

The product is the C-ABI library libfs3d_hip.so (include/fs3d.h, sources in csrc/);
`capi` binds it with ctypes and mirrors the reference's Solver3D interface, `grids`
holds node-array builders.  Nothing here imports the CPU oracle.
"""
from . import grids  # noqa: F401
from . import capi  # noqa: F401
