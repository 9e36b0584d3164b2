// Multi-GPU plumbing (RCCL) behind fs3d_comm_init: halo planes of the x-slab
// decomposition and the two-scalar all-reduce of EvalDivError.
#pragma once
#include "fs3d_common.h"

void fs3d_comm_destroy(fs3d_ctx *c);
// exchange the boundary x-planes of the first nfields fields of layer buffer `buf`
// with the slab neighbours (ScalarField3D::syncHalos, TimeLayer3D.h:272-335). No-op for one rank.
fs3d_status fs3d_comm_halo_exchange(fs3d_ctx *c, int buf, int nfields);
// in-place sum of two doubles on the device over all ranks (TimeLayer3D.h:630-637). No-op for one rank.
fs3d_status fs3d_comm_allreduce_sum2(fs3d_ctx *c, double *dev2);
// stream-ordered point-to-point transfer of `count` reals to/from a neighbouring rank
fs3d_status fs3d_comm_send(fs3d_ctx *c, const void *dev, size_t count, int peer);
fs3d_status fs3d_comm_recv(fs3d_ctx *c, void *dev, size_t count, int peer);
