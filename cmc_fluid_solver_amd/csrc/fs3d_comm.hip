// Halo / carry exchange and the error all-reduce of the x-slab decomposition.
// Two transports behind the same four calls:
//   * RCCL (one process per GPU, fs3d_comm_init)            -- the production path, stream-ordered p2p groups;
//   * in-process (one host thread per slab context, fs3d_comm_init_local) -- device-to-device copies with a
//     host rendezvous; the reference's own single-process multi-GPU mode (Common/GPUplan.h:29-108) and the
//     way the slab protocol is exercised on a single card.
#include "fs3d_comm.h"
#include <rccl/rccl.h>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <vector>

struct XOp { void *ptr; size_t count; int peer; bool send; };

// ---- in-process transport -------------------------------------------------------------------------------
struct LocalMsg { const void *ptr; size_t bytes; };
struct fs3d_local_group {
    int n;
    std::mutex m;
    std::condition_variable cv;
    std::vector<std::deque<LocalMsg>> box;      // [src * n + dst]
    std::vector<long long> posted, done;        // [src * n + dst]
    std::vector<double> red;                    // [rank][2]
    double sum[2][2];
    int arrived = 0;
    long long gen = 0;
    bool broken = false;
    explicit fs3d_local_group(int n_) : n(n_), box((size_t)n_ * n_), posted((size_t)n_ * n_, 0), done((size_t)n_ * n_, 0), red((size_t)n_ * 2, 0.0) {}
};

extern "C" fs3d_status fs3d_local_group_create(int nranks, void **group_out)
{
    if (!group_out || nranks < 1 || nranks > 64) return FS3D_ERR_INVALID;
    *group_out = new fs3d_local_group(nranks);
    return FS3D_OK;
}

extern "C" void fs3d_local_group_destroy(void *group) { delete (fs3d_local_group *)group; }

extern "C" void fs3d_local_group_abort(void *group)
{
    fs3d_local_group *g = (fs3d_local_group *)group;
    if (!g) return;
    { std::lock_guard<std::mutex> lk(g->m); g->broken = true; }
    g->cv.notify_all();
}

extern "C" fs3d_status fs3d_comm_init_local(fs3d_ctx *c, void *group, int rank)
{
    fs3d_local_group *g = (fs3d_local_group *)group;
    if (!c || !g || rank < 0 || rank >= g->n || c->comm) return FS3D_ERR_INVALID;
    c->local = g; c->rank = rank; c->nranks = g->n;
    if (const char *e = getenv("FS3D_XBLOCKS")) { int v = atoi(e); if (v >= 1 && v <= 64) c->xblocks = v; }
    return FS3D_OK;
}

static fs3d_status local_fail(fs3d_ctx *c, const char *what)
{
    fs3d_local_group *g = (fs3d_local_group *)c->local;
    { std::lock_guard<std::mutex> lk(g->m); g->broken = true; }
    g->cv.notify_all();
    c->err = std::string("in-process transport: ") + what;
    return FS3D_ERR_COMM;
}

static hipStream_t xs(fs3d_ctx *c) { return c->xstream ? c->xstream : c->stream; }

static fs3d_status local_exec(fs3d_ctx *c, const std::vector<XOp> &ops)
{
    fs3d_local_group *g = (fs3d_local_group *)c->local;
    const int n = g->n, me = c->rank;
    if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(xs(c)) != hipSuccess) return local_fail(c, "stream sync before send");
    {
        std::lock_guard<std::mutex> lk(g->m);
        for (const XOp &o : ops) if (o.send) { g->box[(size_t)me * n + o.peer].push_back({o.ptr, o.count * c->esize}); g->posted[(size_t)me * n + o.peer]++; }
    }
    g->cv.notify_all();
    std::vector<int> took(n, 0);
    for (const XOp &o : ops) {
        if (o.send) continue;
        LocalMsg msg;
        {
            std::unique_lock<std::mutex> lk(g->m);
            auto &q = g->box[(size_t)o.peer * n + me];
            g->cv.wait(lk, [&] { return !q.empty() || g->broken; });
            if (g->broken) { c->err = "in-process transport: a peer failed"; return FS3D_ERR_COMM; }
            msg = q.front(); q.pop_front();
        }
        if (msg.bytes != o.count * c->esize) return local_fail(c, "send/recv size mismatch");
        if (hipMemcpyAsync(o.ptr, msg.ptr, msg.bytes, hipMemcpyDefault, xs(c)) != hipSuccess) return local_fail(c, "device copy");
        took[o.peer]++;
    }
    if (hipStreamSynchronize(xs(c)) != hipSuccess) return local_fail(c, "stream sync after recv");
    {
        std::unique_lock<std::mutex> lk(g->m);
        for (int p = 0; p < n; p++) g->done[(size_t)p * n + me] += took[p];
        g->cv.notify_all();
        // the source buffers stay untouched until every peer has copied them
        g->cv.wait(lk, [&] {
            if (g->broken) return true;
            for (int p = 0; p < n; p++) if (g->done[(size_t)me * n + p] < g->posted[(size_t)me * n + p]) return false;
            return true;
        });
        if (g->broken) { c->err = "in-process transport: a peer failed"; return FS3D_ERR_COMM; }
    }
    return FS3D_OK;
}

static fs3d_status local_allreduce_sum2(fs3d_ctx *c, double *dev2)
{
    fs3d_local_group *g = (fs3d_local_group *)c->local;
    double v[2];
    if (hipSetDevice(c->device) != hipSuccess || hipMemcpyAsync(v, dev2, sizeof v, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) return local_fail(c, "all-reduce read");
    {
        std::unique_lock<std::mutex> lk(g->m);
        g->red[2 * c->rank] = v[0]; g->red[2 * c->rank + 1] = v[1];
        const long long my_gen = g->gen;
        if (++g->arrived == g->n) {
            double s0 = 0, s1 = 0;
            for (int r = 0; r < g->n; r++) { s0 += g->red[2 * r]; s1 += g->red[2 * r + 1]; }   // rank order
            g->sum[my_gen & 1][0] = s0; g->sum[my_gen & 1][1] = s1;
            g->arrived = 0; g->gen++;
            g->cv.notify_all();
        } else {
            g->cv.wait(lk, [&] { return g->gen != my_gen || g->broken; });
            if (g->broken) { c->err = "in-process transport: a peer failed"; return FS3D_ERR_COMM; }
        }
        v[0] = g->sum[my_gen & 1][0]; v[1] = g->sum[my_gen & 1][1];
    }
    if (hipMemcpyAsync(dev2, v, sizeof v, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) return local_fail(c, "all-reduce write");
    return FS3D_OK;
}

// ---- RCCL transport -------------------------------------------------------------------------------------

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
    if (const char *e = getenv("FS3D_XBLOCKS")) { int v = atoi(e); if (v >= 1 && v <= 64) c->xblocks = v; }
    return FS3D_OK;
}

// A rank that cannot go on (an error in the middle of a multi-step exchange, a driver thread that threw) takes the group
// down so that its peers return FS3D_ERR_COMM from their pending / next exchange instead of waiting for ever.
extern "C" fs3d_status fs3d_comm_abort(fs3d_ctx *c)
{
    if (!c) return FS3D_ERR_INVALID;
    if (c->local) {
        fs3d_local_group *g = (fs3d_local_group *)c->local;
        { std::lock_guard<std::mutex> lk(g->m); g->broken = true; }
        g->cv.notify_all();
    } else if (c->comm) {
        ncclCommAbort((ncclComm_t)c->comm);
        c->comm = nullptr;
    }
    return FS3D_OK;
}

extern "C" fs3d_status fs3d_comm_selftest(fs3d_ctx *c, size_t elems)
{
    if (!c || elems == 0 || elems > ((size_t)1 << 28)) return FS3D_ERR_INVALID;
    if (hipSetDevice(c->device) != hipSuccess) { c->err = "hipSetDevice failed"; return FS3D_ERR_HIP; }
    ncclUniqueId id;
    NCCLCHK(c, ncclGetUniqueId(&id));
    ncclComm_t comm;
    NCCLCHK(c, ncclCommInitRank(&comm, 1, id, 0));
    const size_t bytes = elems * c->esize;
    char *src = nullptr, *dst = nullptr; double *red = nullptr;
    std::vector<unsigned char> pat(bytes), back(bytes);
    for (size_t i = 0; i < bytes; i++) pat[i] = (unsigned char)((i * 2654435761u) >> 13);
    for (size_t i = c->esize - 1; i < bytes; i += c->esize) pat[i] &= 0x3f;          // finite values of either precision
    fs3d_status st = FS3D_OK;
    auto hipfail = [&](const char *what) { c->err = std::string("comm selftest: ") + what; st = FS3D_ERR_HIP; };
    auto ncclfail = [&](const char *what, ncclResult_t r) { st = cfail(c, what, r); };
    const ncclDataType_t dt = c->prec == FS3D_F32 ? ncclFloat : ncclDouble;
    hipStream_t s2 = nullptr;                      // a second, non-blocking stream like the one the halo planes travel on
    do {
        if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) { s2 = nullptr; hipfail("stream"); break; }
        if (hipMalloc((void **)&src, bytes) != hipSuccess || hipMalloc((void **)&dst, bytes) != hipSuccess || hipMalloc((void **)&red, 2 * sizeof(double)) != hipSuccess) { hipfail("hipMalloc"); break; }
        if (hipMemcpy(src, pat.data(), bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemset(dst, 0, bytes) != hipSuccess) { hipfail("upload"); break; }
        // 1. grouped send/recv (the halo-plane shape), own rank as the peer, on the exchange stream
        ncclResult_t r;
        if ((r = ncclGroupStart()) != ncclSuccess) { ncclfail("ncclGroupStart", r); break; }
        ncclResult_t r1 = ncclSend(src, elems, dt, 0, comm, s2), r2 = ncclRecv(dst, elems, dt, 0, comm, s2);
        r = ncclGroupEnd();
        if (r1 != ncclSuccess || r2 != ncclSuccess || r != ncclSuccess) { ncclfail("grouped ncclSend/ncclRecv", r1 != ncclSuccess ? r1 : r2 != ncclSuccess ? r2 : r); break; }
        if (hipStreamSynchronize(s2) != hipSuccess || hipMemcpy(back.data(), dst, bytes, hipMemcpyDeviceToHost) != hipSuccess) { hipfail("send/recv readback"); break; }
        if (back != pat) { c->err = "comm selftest: grouped send/recv delivered other bytes than were sent"; st = FS3D_ERR_COMM; break; }
        // 2. all-gather (the interface words of the cross-slab X solve), on the compute stream
        if (hipMemset(dst, 0, bytes) != hipSuccess) { hipfail("memset"); break; }
        if ((r = ncclAllGather(src, dst, elems, dt, comm, c->stream)) != ncclSuccess) { ncclfail("ncclAllGather", r); break; }
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(back.data(), dst, bytes, hipMemcpyDeviceToHost) != hipSuccess) { hipfail("all-gather readback"); break; }
        if (back != pat) { c->err = "comm selftest: all-gather delivered other bytes than were sent"; st = FS3D_ERR_COMM; break; }
        // 3. the 2-double all-reduce of EvalDivError, in place
        const double v[2] = {1.25, 42.0};
        double w[2] = {0, 0};
        if (hipMemcpy(red, v, sizeof v, hipMemcpyHostToDevice) != hipSuccess) { hipfail("upload"); break; }
        if ((r = ncclAllReduce(red, red, 2, ncclDouble, ncclSum, comm, c->stream)) != ncclSuccess) { ncclfail("ncclAllReduce", r); break; }
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(w, red, sizeof w, hipMemcpyDeviceToHost) != hipSuccess) { hipfail("all-reduce readback"); break; }
        if (w[0] != v[0] || w[1] != v[1]) { c->err = "comm selftest: all-reduce over one rank changed the values"; st = FS3D_ERR_COMM; break; }
    } while (0);
    if (src) hipFree(src);
    if (dst) hipFree(dst);
    if (red) hipFree(red);
    if (s2) hipStreamDestroy(s2);
    if (st == FS3D_OK) ncclCommDestroy(comm); else ncclCommAbort(comm);
    return st;
}

void fs3d_comm_destroy(fs3d_ctx *c)
{
    if (c && c->comm) { ncclCommDestroy((ncclComm_t)c->comm); c->comm = nullptr; }
    if (c) c->local = nullptr;
    if (c) for (int i = 0; i < 4; i++) if (c->carry[i]) { hipFree(c->carry[i]); c->carry[i] = nullptr; }
    if (c) for (int i = 0; i < 4; i++) if (c->xa2a[i]) { hipFree(c->xa2a[i]); c->xa2a[i] = nullptr; }
}

// one grouped exchange: every send and receive of `ops` is in flight together (no ordering deadlock)
static fs3d_status exec_group(fs3d_ctx *c, const std::vector<XOp> &ops)
{
    if (c->local) return local_exec(c, ops);
    if (!c->comm) { c->err = "slab context has no communicator (fs3d_comm_init / fs3d_comm_init_local)"; return FS3D_ERR_COMM; }
    const ncclDataType_t dt = c->prec == FS3D_F32 ? ncclFloat : ncclDouble;
    NCCLCHK(c, ncclGroupStart());
    for (const XOp &o : ops) {
        const ncclResult_t r = o.send ? ncclSend(o.ptr, o.count, dt, o.peer, (ncclComm_t)c->comm, xs(c))
                                      : ncclRecv(o.ptr, o.count, dt, o.peer, (ncclComm_t)c->comm, xs(c));
        if (r != ncclSuccess) { ncclGroupEnd(); return cfail(c, o.send ? "ncclSend" : "ncclRecv", r); }     // never leave the group open
    }
    NCCLCHK(c, ncclGroupEnd());
    return FS3D_OK;
}

fs3d_status fs3d_comm_halo_exchange(fs3d_ctx *c, int buf, int nfields)
{
    if (c->nranks == 1) return FS3D_OK;
    const size_t pl = (size_t)c->plane;
    // layout per field: [ghost lo][dimx owned planes][ghost hi]; one grouped send/recv per neighbour
    std::vector<XOp> ops;
    for (int v = 0; v < nfields; v++) {
        char *base = (char *)c->lay[buf] + (size_t)v * c->fstride * c->esize;
        char *first = base + pl * c->esize;                       // first owned plane
        char *last = base + (size_t)c->dimx * pl * c->esize;      // last owned plane
        char *glo = base;                                         // ghost below
        char *ghi = base + (size_t)(c->dimx + 1) * pl * c->esize; // ghost above
        if (c->rank > 0) { ops.push_back({first, pl, c->rank - 1, true}); ops.push_back({glo, pl, c->rank - 1, false}); }
        if (c->rank < c->nranks - 1) { ops.push_back({last, pl, c->rank + 1, true}); ops.push_back({ghi, pl, c->rank + 1, false}); }
    }
    return exec_group(c, ops);
}

fs3d_status fs3d_comm_allreduce_sum2(fs3d_ctx *c, double *dev2)
{
    if (c->nranks == 1) return FS3D_OK;
    if (c->local) return local_allreduce_sum2(c, dev2);
    if (!c->comm) { c->err = "slab context has no communicator"; return FS3D_ERR_COMM; }
    NCCLCHK(c, ncclAllReduce(dev2, dev2, 2, ncclDouble, ncclSum, (ncclComm_t)c->comm, c->stream));
    return FS3D_OK;
}

fs3d_status fs3d_comm_xfer_rows(fs3d_ctx *c, void *dev, int nrows, size_t pitch, long long l0, long long l1, int peer, bool send)
{
    std::vector<XOp> ops;
    for (int r = 0; r < nrows; r++)
        ops.push_back({(char *)dev + ((size_t)r * pitch + (size_t)l0) * c->esize, (size_t)(l1 - l0), peer, send});
    return exec_group(c, ops);
}

fs3d_status fs3d_comm_allgather(fs3d_ctx *c, const void *send, void *recv, size_t count)
{
    if (hipMemcpyAsync((char *)recv + (size_t)c->rank * count * c->esize, send, count * c->esize, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) {
        c->err = "all-gather: device copy of the own block failed"; return FS3D_ERR_HIP;
    }
    if (c->nranks == 1) return FS3D_OK;
    if (!c->local && c->comm) {
        const ncclDataType_t dt = c->prec == FS3D_F32 ? ncclFloat : ncclDouble;
        NCCLCHK(c, ncclAllGather(send, recv, count, dt, (ncclComm_t)c->comm, c->stream));
        return FS3D_OK;
    }
    std::vector<XOp> ops;
    for (int r = 0; r < c->nranks; r++) {
        if (r == c->rank) continue;
        ops.push_back({(void *)send, count, r, true});
        ops.push_back({(char *)recv + (size_t)r * count * c->esize, count, r, false});
    }
    return exec_group(c, ops);
}

fs3d_status fs3d_comm_alltoall(fs3d_ctx *c, const void *send, void *recv, size_t count)
{
    const size_t bb = count * c->esize;
    if (hipMemcpyAsync((char *)recv + (size_t)c->rank * bb, (const char *)send + (size_t)c->rank * bb, bb, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) {
        c->err = "all-to-all: device copy of the own block failed"; return FS3D_ERR_HIP;
    }
    if (c->nranks == 1) return FS3D_OK;
    std::vector<XOp> ops;
    for (int r = 0; r < c->nranks; r++) {
        if (r == c->rank) continue;
        ops.push_back({(char *)send + (size_t)r * bb, count, r, true});
        ops.push_back({(char *)recv + (size_t)r * bb, count, r, false});
    }
    return exec_group(c, ops);
}
