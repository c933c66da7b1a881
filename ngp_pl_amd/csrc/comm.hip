// RCCL communicator of the data-parallel step (reference: Lightning's DDPPlugin, /root/reference/train.py:268-272 -- one process per
// GPU, gradients averaged by an all-reduce every step).
//
// Why this is native: round 3 issued the collectives from Python through torch.distributed, on torch's NCCL stream, with two
// cross-stream hand-overs per collective -- even a ONE-rank group cost 10 % of the step (0.481 -> 0.530 ms), all of it host time and
// hand-over bubbles.  Here the communicator belongs to the library: the stepper's tail (csrc/stepper.hip, ngp_stepper_tail) enqueues
// the MLP all-reduce, the reduce-scatter of each finished chunk of the table gradient, the non-finite checks, the rank's share of
// Adam and the all-gather of the updated f16 table on the communicator's OWN stream, ordered by events the main stream only
// records; the main stream waits once, at the end.
//
// RCCL is resolved at run time (dlopen of librccl.so.1: inside a torch process that is the copy torch already loaded), so the
// library has no link-time dependency on it and single-GPU consumers of the C ABI never touch it.
#include "ngp_common.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>

#include "comm.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclReduceScatter) ReduceScatter = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    bool ok = false;
};

char g_last_error[256] = "";
std::mutex g_mutex;

void set_error(const char* what, const char* detail) {
    std::lock_guard<std::mutex> lock(g_mutex);
    snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, detail ? detail : "?");
}

Rccl& rccl() {
    static Rccl r = [] {
        Rccl t;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            t.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (t.handle) break;
        }
        if (!t.handle) { set_error("dlopen(librccl)", dlerror()); return t; }
#define NGP_SYM(field, name) t.field = reinterpret_cast<decltype(t.field)>(dlsym(t.handle, name)); if (!t.field) { set_error("dlsym", name); return t; }
        NGP_SYM(GetUniqueId, "ncclGetUniqueId") NGP_SYM(CommInitRank, "ncclCommInitRank") NGP_SYM(CommDestroy, "ncclCommDestroy")
        NGP_SYM(CommAbort, "ncclCommAbort") NGP_SYM(AllReduce, "ncclAllReduce") NGP_SYM(ReduceScatter, "ncclReduceScatter")
        NGP_SYM(AllGather, "ncclAllGather") NGP_SYM(Broadcast, "ncclBroadcast") NGP_SYM(GroupStart, "ncclGroupStart")
        NGP_SYM(GroupEnd, "ncclGroupEnd") NGP_SYM(GetErrorString, "ncclGetErrorString") NGP_SYM(GetVersion, "ncclGetVersion")
        NGP_SYM(Send, "ncclSend") NGP_SYM(Recv, "ncclRecv")
#undef NGP_SYM
        t.ok = true;
        return t;
    }();
    return r;
}

}  // namespace

// (shared with csrc/stepper.hip through comm.h)
int ngp_comm_check(ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return 0;
    set_error(what, rccl().ok ? rccl().GetErrorString(r) : "RCCL not loaded");
    return NGP_ECOMM;
}

ncclDataType_t ngp_comm_dtype(int dtype) { return dtype == NGP_COMM_F16 ? ncclFloat16 : (dtype == NGP_COMM_F32 ? ncclFloat32 : ncclUint8); }

int ngp_comm_group_begin() { return rccl().ok ? ngp_comm_check(rccl().GroupStart(), "ncclGroupStart") : NGP_ECOMM; }
int ngp_comm_group_end() { return rccl().ok ? ngp_comm_check(rccl().GroupEnd(), "ncclGroupEnd") : NGP_ECOMM; }

extern "C" {
#pragma GCC visibility push(default)

const char* ngp_comm_last_error(void) { return g_last_error; }

int ngp_comm_unique_id(void* id_bytes) {
    NGP_CHECK_PTR(id_bytes);
    static_assert(sizeof(ncclUniqueId) == NGP_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (!rccl().ok) return NGP_ECOMM;
    ncclUniqueId id;
    const int rc = ngp_comm_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    if (rc) return rc;
    memcpy(id_bytes, &id, sizeof(id));
    return 0;
}

int ngp_comm_create(const void* id_bytes, int world, int rank, ngp_comm** out) {
    if (!id_bytes || !out || world < 1 || rank < 0 || rank >= world) return NGP_EINVAL;
    *out = nullptr;
    if (!rccl().ok) return NGP_ECOMM;
    ngp_comm* c = new (std::nothrow) ngp_comm();
    if (!c) return NGP_EINVAL;
    c->world = world; c->rank = rank;
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; return NGP_EINVAL; }
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    int rc = ngp_comm_check(rccl().CommInitRank(&c->comm, world, id, rank), "ncclCommInitRank");
    if (rc) { delete c; return rc; }
    // the communicator's own stream, high priority: its kernels are short and everything behind them waits (the marching stream is
    // a high-priority stream too -- see trainer.py: a default-priority stream can land on the main stream's hardware queue)
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipError_t e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi);
    if (e != hipSuccess) { (void)rccl().CommDestroy(c->comm); delete c; return (int)e; }
    int v = 0;
    if (rccl().GetVersion(&v) == ncclSuccess) c->version = v;
    *out = c;
    return 0;
}

int ngp_comm_destroy(ngp_comm* c) {
    if (!c) return 0;
    if (c->stream) { (void)hipStreamSynchronize(c->stream); }
    if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int ngp_comm_info(const ngp_comm* c, int32_t* world, int32_t* rank, int32_t* rccl_version, ngp_stream_t* stream) {
    if (!c) return NGP_EINVAL;
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    if (rccl_version) *rccl_version = c->version;
    if (stream) *stream = (ngp_stream_t)c->stream;
    return 0;
}

int ngp_comm_all_reduce(ngp_comm* c, void* buf, int64_t count, int dtype, ngp_stream_t stream) {
    if (!c || count < 0 || (dtype != NGP_COMM_F16 && dtype != NGP_COMM_F32)) return NGP_EINVAL;
    if (count == 0) return 0;
    NGP_CHECK_PTR(buf);
    return ngp_comm_check(rccl().AllReduce(buf, buf, (size_t)count, ngp_comm_dtype(dtype), ncclSum, c->comm, stream ? ngp_stream(stream) : c->stream), "ncclAllReduce");
}

int ngp_comm_reduce_scatter(ngp_comm* c, const void* send, void* recv, int64_t recv_count, int dtype, ngp_stream_t stream) {
    if (!c || recv_count < 0 || (dtype != NGP_COMM_F16 && dtype != NGP_COMM_F32)) return NGP_EINVAL;
    if (recv_count == 0) return 0;
    NGP_CHECK_PTR(send); NGP_CHECK_PTR(recv);
    return ngp_comm_check(rccl().ReduceScatter(send, recv, (size_t)recv_count, ngp_comm_dtype(dtype), ncclSum, c->comm, stream ? ngp_stream(stream) : c->stream),
                          "ncclReduceScatter");
}

int ngp_comm_all_gather(ngp_comm* c, const void* send, void* recv, int64_t send_count, int dtype, ngp_stream_t stream) {
    if (!c || send_count < 0 || (dtype != NGP_COMM_F16 && dtype != NGP_COMM_F32)) return NGP_EINVAL;
    if (send_count == 0) return 0;
    NGP_CHECK_PTR(send); NGP_CHECK_PTR(recv);
    return ngp_comm_check(rccl().AllGather(send, recv, (size_t)send_count, ngp_comm_dtype(dtype), c->comm, stream ? ngp_stream(stream) : c->stream), "ncclAllGather");
}

int ngp_comm_broadcast(ngp_comm* c, void* buf, int64_t n_bytes, int root, ngp_stream_t stream) {
    if (!c || n_bytes < 0 || root < 0 || root >= c->world) return NGP_EINVAL;
    if (n_bytes == 0) return 0;
    NGP_CHECK_PTR(buf);
    return ngp_comm_check(rccl().Broadcast(buf, buf, (size_t)n_bytes, ncclUint8, root, c->comm, stream ? ngp_stream(stream) : c->stream), "ncclBroadcast");
}

// ---- the point-to-point forms (round 5): xGMI is a full mesh of point-to-point links, a ring collective is bound by ONE of them ----
// Slice q of `send` (world x count values) goes straight to rank q; what rank q sends to this rank lands in recv + q x count.  The
// rank's own slice is not copied (recv + rank x count is left alone).  One RCCL group: world - 1 sends and receives over world - 1
// different links at once -- the reduce-scatter's data movement without its additions (the caller sums the slices itself, in rank
// order: deterministic, in f32).
int ngp_comm_exchange_slices(ngp_comm* c, const void* send, void* recv, int64_t count, int dtype, ngp_stream_t stream) {
    if (!c || count < 0 || (dtype != NGP_COMM_F16 && dtype != NGP_COMM_F32)) return NGP_EINVAL;
    if (count == 0 || c->world == 1) return 0;
    NGP_CHECK_PTR(send); NGP_CHECK_PTR(recv);
    if (!rccl().ok) return NGP_ECOMM;
    const size_t bytes = (size_t)count * (dtype == NGP_COMM_F16 ? 2 : 4);
    hipStream_t st = stream ? ngp_stream(stream) : c->stream;
    int rc = ngp_comm_check(rccl().GroupStart(), "ncclGroupStart");
    if (rc) return rc;
    for (int q = 0; q < c->world && rc == 0; ++q) {
        if (q == c->rank) continue;
        rc = ngp_comm_check(rccl().Send(static_cast<const char*>(send) + (size_t)q * bytes, (size_t)count, ngp_comm_dtype(dtype), q, c->comm, st), "ncclSend");
        if (rc == 0) rc = ngp_comm_check(rccl().Recv(static_cast<char*>(recv) + (size_t)q * bytes, (size_t)count, ngp_comm_dtype(dtype), q, c->comm, st), "ncclRecv");
    }
    const int rc2 = ngp_comm_check(rccl().GroupEnd(), "ncclGroupEnd");
    return rc ? rc : rc2;
}

// All-gather by direct sends, in place: this rank's slice (buf + rank x count) goes to every peer, rank q's slice arrives at
// buf + q x count.  Same result as ngp_comm_all_gather(in place); world - 1 links at once instead of a ring.
int ngp_comm_all_gather_direct(ngp_comm* c, void* buf, int64_t count, int dtype, ngp_stream_t stream) {
    if (!c || count < 0 || (dtype != NGP_COMM_F16 && dtype != NGP_COMM_F32)) return NGP_EINVAL;
    if (count == 0 || c->world == 1) return 0;
    NGP_CHECK_PTR(buf);
    if (!rccl().ok) return NGP_ECOMM;
    const size_t bytes = (size_t)count * (dtype == NGP_COMM_F16 ? 2 : 4);
    hipStream_t st = stream ? ngp_stream(stream) : c->stream;
    char* b = static_cast<char*>(buf);
    int rc = ngp_comm_check(rccl().GroupStart(), "ncclGroupStart");
    if (rc) return rc;
    for (int q = 0; q < c->world && rc == 0; ++q) {
        if (q == c->rank) continue;
        rc = ngp_comm_check(rccl().Send(b + (size_t)c->rank * bytes, (size_t)count, ngp_comm_dtype(dtype), q, c->comm, st), "ncclSend");
        if (rc == 0) rc = ngp_comm_check(rccl().Recv(b + (size_t)q * bytes, (size_t)count, ngp_comm_dtype(dtype), q, c->comm, st), "ncclRecv");
    }
    const int rc2 = ngp_comm_check(rccl().GroupEnd(), "ncclGroupEnd");
    return rc ? rc : rc2;
}

#pragma GCC visibility pop
}  // extern "C"
