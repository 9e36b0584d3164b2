// Multi-GPU plumbing (RCCL, or the in-process transport) behind fs3d_comm_init / fs3d_comm_init_local: halo planes of the x-slab
// decomposition and the two-scalar all-reduce of EvalDivError.
#pragma once
#include "fs3d_common.h"

void fs3d_comm_destroy(fs3d_ctx *c);
// exchange the boundary x-planes of the first nfields fields of layer buffer `buf`
// with the slab neighbours (ScalarField3D::syncHalos, TimeLayer3D.h:272-335). No-op for one rank.
fs3d_status fs3d_comm_halo_exchange(fs3d_ctx *c, int buf, int nfields);
// in-place sum of two doubles on the device over all ranks (TimeLayer3D.h:630-637). No-op for one rank.
fs3d_status fs3d_comm_allreduce_sum2(fs3d_ctx *c, double *dev2);
// one grouped transfer of elements [l0,l1) of each of `nrows` rows (row pitch `pitch` elements) to/from `peer`
fs3d_status fs3d_comm_xfer_rows(fs3d_ctx *c, void *dev, int nrows, size_t pitch, long long l0, long long l1, int peer, bool send);
// all-gather of `count` elements per rank: recv = [rank][count] (send may not alias recv); one grouped exchange
fs3d_status fs3d_comm_allgather(fs3d_ctx *c, const void *send, void *recv, size_t count);
// all-to-all of `count` elements per pair of ranks: block r of `send` goes to rank r, block r of `recv` comes from rank r (the own block is a
// device copy); one grouped exchange of point-to-point transfers -- one send and one receive per peer
fs3d_status fs3d_comm_alltoall(fs3d_ctx *c, const void *send, void *recv, size_t count);
extern "C" fs3d_status fs3d_comm_abort(fs3d_ctx *c);
