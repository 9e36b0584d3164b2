// RCCL halo exchange / all-reduce for the x-slab decomposition (one process per GPU).
#include "fs3d_comm.h"
#include <rccl/rccl.h>
#include <cstdio>
#include <cstring>

static fs3d_status cfail(fs3d_ctx *c, const char *what, ncclResult_t r)
{
    char b[256];
    snprintf(b, sizeof b, "GPU %d: %s failed: %s", c ? c->device : -1, what, ncclGetErrorString(r));
    if (c) c->err = b;
    return FS3D_ERR_COMM;
}
#define NCCLCHK(c, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return cfail((c), #call, r_); } while (0)

extern "C" fs3d_status fs3d_comm_unique_id(void *unique_id_128)
{
    if (!unique_id_128) return FS3D_ERR_INVALID;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return FS3D_ERR_COMM;
    memcpy(unique_id_128, &id, sizeof id);
    return FS3D_OK;
}

extern "C" fs3d_status fs3d_comm_init(fs3d_ctx *c, const void *unique_id_128, int rank, int nranks)
{
    if (!c || !unique_id_128 || nranks < 1 || rank < 0 || rank >= nranks) return FS3D_ERR_INVALID;
    if (nranks == 1) { c->rank = 0; c->nranks = 1; return FS3D_OK; }
    if (hipSetDevice(c->device) != hipSuccess) { c->err = "hipSetDevice failed"; return FS3D_ERR_HIP; }
    ncclUniqueId id;
    memcpy(&id, unique_id_128, sizeof id);
    ncclComm_t comm;
    NCCLCHK(c, ncclCommInitRank(&comm, nranks, id, rank));
    c->comm = comm; c->rank = rank; c->nranks = nranks;
    return FS3D_OK;
}

void fs3d_comm_destroy(fs3d_ctx *c)
{
    if (c && c->comm) { ncclCommDestroy((ncclComm_t)c->comm); c->comm = nullptr; }
    if (c) for (int i = 0; i < 4; i++) if (c->carry[i]) { hipFree(c->carry[i]); c->carry[i] = nullptr; }
}

fs3d_status fs3d_comm_halo_exchange(fs3d_ctx *c, int buf, int nfields)
{
    if (c->nranks == 1) return FS3D_OK;
    ncclComm_t comm = (ncclComm_t)c->comm;
    const ncclDataType_t dt = c->prec == FS3D_F32 ? ncclFloat : ncclDouble;
    const size_t pl = (size_t)c->plane;
    // layout per field: [ghost lo][dimx owned planes][ghost hi]; one grouped send/recv per neighbour
    NCCLCHK(c, ncclGroupStart());
    for (int v = 0; v < nfields; v++) {
        char *base = (char *)c->lay[buf] + (size_t)v * c->fstride * c->esize;
        char *first = base + pl * c->esize;                       // first owned plane
        char *last = base + (size_t)c->dimx * pl * c->esize;      // last owned plane
        char *glo = base;                                         // ghost below
        char *ghi = base + (size_t)(c->dimx + 1) * pl * c->esize; // ghost above
        if (c->rank > 0) {
            NCCLCHK(c, ncclSend(first, pl, dt, c->rank - 1, comm, c->stream));
            NCCLCHK(c, ncclRecv(glo, pl, dt, c->rank - 1, comm, c->stream));
        }
        if (c->rank < c->nranks - 1) {
            NCCLCHK(c, ncclSend(last, pl, dt, c->rank + 1, comm, c->stream));
            NCCLCHK(c, ncclRecv(ghi, pl, dt, c->rank + 1, comm, c->stream));
        }
    }
    NCCLCHK(c, ncclGroupEnd());
    return FS3D_OK;
}

fs3d_status fs3d_comm_allreduce_sum2(fs3d_ctx *c, double *dev2)
{
    if (c->nranks == 1) return FS3D_OK;
    NCCLCHK(c, ncclAllReduce(dev2, dev2, 2, ncclDouble, ncclSum, (ncclComm_t)c->comm, c->stream));
    return FS3D_OK;
}

fs3d_status fs3d_comm_send(fs3d_ctx *c, const void *dev, size_t count, int peer)
{
    NCCLCHK(c, ncclSend(dev, count, c->prec == FS3D_F32 ? ncclFloat : ncclDouble, peer, (ncclComm_t)c->comm, c->stream));
    return FS3D_OK;
}

fs3d_status fs3d_comm_recv(fs3d_ctx *c, void *dev, size_t count, int peer)
{
    NCCLCHK(c, ncclRecv(dev, count, c->prec == FS3D_F32 ? ncclFloat : ncclDouble, peer, (ncclComm_t)c->comm, c->stream));
    return FS3D_OK;
}
