// Multi-GPU exchange steps of the hot path over RCCL, for hosts that do not go through torch.distributed
// (tf_comm_init / tf_allgather_kv / tf_sendrecv_pivot / tf_comm_destroy; tokenflow_amd/sharded.py
// describes the two exchange steps and is what the Python host uses).  The reference is single-process: new work.
//
// RCCL is bound at run time (dlopen of librccl.so.1 on the first tf_comm_* call): libtokenflow_hip.so itself keeps no
// link-time dependency on it, a single-GPU process never loads it, and inside a PyTorch process the loader hands back
// the copy torch already mapped (same soname), so there is one RCCL per process.
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include <mutex>

#include "tf_common.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

// resolved once per process; read-only afterwards
const Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) return;
        auto sym = [&](const char* n) { return dlsym(r.handle, n); };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.Send && r.Recv && r.GroupStart &&
               r.GroupEnd && r.GetErrorString;
    });
    return r;
}

int elem_bytes(int dtype) { return dtype == TF_F32 ? 4 : 2; }

}  // namespace

// Transport behind a communicator: RCCL (the product), a host-provided function table (tf_comm_init_hooks: an MPI
// host, the gloo-carried transport of the multi-process tests), or the wire-less stand-in (tf_comm_init_loopback:
// every exchange becomes same-size device-to-device copies on the stream -- the launch sequence, buffers and stream
// hand-overs of a rank of a W-GPU run on ONE GPU, for timing only: the data a world > 1 loopback "receives" is its own).
enum { COMM_RCCL = 0, COMM_HOOKS = 1, COMM_LOOPBACK = 2 };

struct tf_comm {
    ncclComm_t comm;
    int rank, world;
    int kind = COMM_RCCL;
    tf_comm_hooks hooks = {};
    bool copies = true;   // loopback only: false = the exchanges move nothing at all (tf_comm_loopback_copies)
    double wire_lat_us = 0.0, wire_gbps = 0.0;   // loopback only: the wire model (tf_comm_loopback_wire); 0 = off
};

namespace {

int copy_async(void* dst, const void* src, size_t bytes, hipStream_t st, const char* what, bool enabled = true) {
    if (bytes == 0 || dst == src || !enabled) return 0;
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) {
        tf_set_error("%s: hipMemcpyAsync: %s", what, hipGetErrorString(e));
        return (int)e;   // > 0: a HIP error code, as for launch failures
    }
    return 0;
}

// Wire model of the loopback transport: one wave that holds the stream for `ticks` of the constant 100 MHz clock
// (s_memrealtime) -- what an exchange of that duration does to the schedule, without its data.  One wave on one CU: the
// compute streams beside it lose nothing measurable.
__global__ void wire_hold_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
int wire_hold(const tf_comm* c, size_t busiest_link_bytes, hipStream_t st, const char* what) {
    if (c->wire_lat_us <= 0.0 && c->wire_gbps <= 0.0) return 0;
    double us = c->wire_lat_us > 0.0 ? c->wire_lat_us : 0.0;
    if (c->wire_gbps > 0.0) us += (double)busiest_link_bytes / (c->wire_gbps * 1e3);   // GB/s = 1e3 bytes per us
    const unsigned long long ticks = (unsigned long long)(us * 100.0 + 0.5);           // wall_clock64: 100 MHz
    if (ticks == 0) return 0;
    hipLaunchKernelGGL(wire_hold_kernel, dim3(1), dim3(64), 0, st, ticks);
    TF_LAUNCH_CHECK(what);
    return 0;
}

// loopback forms: one copy per message a real rank would receive, of that message's size, from this rank's own data
int loop_allgather_rows(const tf_comm* c, const void* local, void* bank, const int64_t* rows, size_t rb, hipStream_t st) {
    char* r = static_cast<char*>(bank);
    const size_t mine = (size_t)rows[c->rank] * rb;
    for (int p = 0; p < c->world; ++p) {
        const size_t n = (size_t)rows[p] * rb;
        if (const int rc = copy_async(r, local, n < mine ? n : mine, st, "tf_allgather_rows(loopback)", c->copies)) return rc;
        r += n;
    }
    size_t link = 0;   // the link to peer p carries this rank's rows out and p's rows in (full duplex)
    for (int p = 0; p < c->world; ++p)
        if (p != c->rank) {
            const size_t in = (size_t)rows[p] * rb, io = in > mine ? in : mine;
            link = io > link ? io : link;
        }
    return wire_hold(c, link, st, "tf_allgather_rows(loopback wire)");
}

int loop_all_to_all(const tf_comm* c, const void* send, void* recv, const int64_t* send_rows, const int64_t* recv_rows,
                    size_t rb, hipStream_t st) {
    size_t ns = 0, nr = 0;
    for (int p = 0; p < c->world; ++p) ns += (size_t)send_rows[p] * rb, nr += (size_t)recv_rows[p] * rb;
    if (const int rc = copy_async(recv, send, ns < nr ? ns : nr, st, "tf_all_to_all_rows(loopback)", c->copies)) return rc;
    size_t link = 0;   // the link to peer p carries the rows for p out and the rows from p in (full duplex)
    for (int p = 0; p < c->world; ++p)
        if (p != c->rank) {
            const size_t out = (size_t)send_rows[p] * rb, in = (size_t)recv_rows[p] * rb, io = out > in ? out : in;
            link = io > link ? io : link;
        }
    return wire_hold(c, link, st, "tf_all_to_all_rows(loopback wire)");
}

}  // namespace

#define TF_NEED_RCCL(what)                                                                          \
    const Rccl& R = rccl();                                                                         \
    if (!R.ok) {                                                                                    \
        tf_set_error("%s: RCCL (librccl.so.1) could not be loaded or lacks a symbol", what);        \
        return TF_ERR_COMM;                                                                         \
    }
#define TF_NCCL(call, what)                                                       \
    do {                                                                          \
        const ncclResult_t r_ = (call);                                           \
        if (r_ != ncclSuccess) {                                                  \
            tf_set_error("%s: %s", what, R.GetErrorString(r_));                   \
            return TF_ERR_COMM;                                                   \
        }                                                                         \
    } while (0)

// Inside ncclGroupStart/End: remember the FIRST failure and keep going to the GroupEnd -- returning early would leave
// the thread's RCCL group open, and every later collective of the process (torch.distributed's included: one RCCL per
// process) would be queued into a group that never closes.
#define TF_NCCL_IN_GROUP(call, first_err)                        \
    do {                                                         \
        const ncclResult_t r_ = (call);                          \
        if (r_ != ncclSuccess && (first_err) == ncclSuccess) (first_err) = r_; \
    } while (0)
#define TF_NCCL_GROUP_END(first_err, what)                                        \
    do {                                                                          \
        const ncclResult_t e_ = R.GroupEnd();                                     \
        const ncclResult_t r_ = (first_err) != ncclSuccess ? (first_err) : e_;    \
        if (r_ != ncclSuccess) {                                                  \
            tf_set_error("%s: %s", what, R.GetErrorString(r_));                   \
            return TF_ERR_COMM;                                                   \
        }                                                                         \
    } while (0)

extern "C" int tf_comm_available(void) {
    TF_NEED_RCCL("tf_comm_available");   // dlopen of librccl + every symbol this file calls; starts nothing
    return 0;
}

extern "C" int tf_comm_unique_id(void* id_out) {
    TF_ARG(id_out, TF_ERR_NULL, "tf_comm_unique_id: null pointer");
    TF_NEED_RCCL("tf_comm_unique_id");
    ncclUniqueId id;
    TF_NCCL(R.GetUniqueId(&id), "tf_comm_unique_id");
    memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

extern "C" int tf_comm_init(const void* unique_id, int rank, int world, tf_comm** comm_out) {
    TF_ARG(unique_id && comm_out, TF_ERR_NULL, "tf_comm_init: null pointer");
    TF_ARG(world > 0 && rank >= 0 && rank < world, TF_ERR_SHAPE, "tf_comm_init: rank %d of %d", rank, world);
    TF_NEED_RCCL("tf_comm_init");
    ncclUniqueId id;
    memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c;
    TF_NCCL(R.CommInitRank(&c, world, id, rank), "tf_comm_init");   // binds the CURRENT device (hipSetDevice first)
    *comm_out = new tf_comm{c, rank, world};
    return 0;
}

extern "C" int tf_comm_init_hooks(const tf_comm_hooks* hooks, int rank, int world, tf_comm** comm_out) {
    TF_ARG(hooks && comm_out, TF_ERR_NULL, "tf_comm_init_hooks: null pointer");
    TF_ARG(hooks->all_to_all_rows && hooks->allgather_rows && hooks->sendrecv, TF_ERR_NULL,
           "tf_comm_init_hooks: a transport needs all three entry points");
    TF_ARG(world > 0 && rank >= 0 && rank < world, TF_ERR_SHAPE, "tf_comm_init_hooks: rank %d of %d", rank, world);
    tf_comm* c = new tf_comm{nullptr, rank, world};
    c->kind = COMM_HOOKS;
    c->hooks = *hooks;
    *comm_out = c;
    return 0;
}

extern "C" int tf_comm_init_loopback(int rank, int world, tf_comm** comm_out) {
    TF_ARG(comm_out, TF_ERR_NULL, "tf_comm_init_loopback: null pointer");
    TF_ARG(world > 0 && rank >= 0 && rank < world, TF_ERR_SHAPE, "tf_comm_init_loopback: rank %d of %d", rank, world);
    tf_comm* c = new tf_comm{nullptr, rank, world};
    c->kind = COMM_LOOPBACK;
    *comm_out = c;
    return 0;
}

extern "C" int tf_comm_loopback_copies(tf_comm* comm, int enabled) {
    TF_ARG(comm && comm->kind == COMM_LOOPBACK, TF_ERR_COMM, "tf_comm_loopback_copies: not a loopback communicator");
    comm->copies = enabled != 0;
    return 0;
}

extern "C" int tf_comm_loopback_wire(tf_comm* comm, double latency_us, double gbps_per_link) {
    TF_ARG(comm && comm->kind == COMM_LOOPBACK, TF_ERR_COMM, "tf_comm_loopback_wire: not a loopback communicator");
    comm->wire_lat_us = latency_us > 0.0 ? latency_us : 0.0;
    comm->wire_gbps = gbps_per_link > 0.0 ? gbps_per_link : 0.0;
    return 0;
}

extern "C" int tf_comm_destroy(tf_comm* comm) {
    if (!comm) return 0;
    if (comm->kind != COMM_RCCL) {
        delete comm;
        return 0;
    }
    TF_NEED_RCCL("tf_comm_destroy");
    const ncclResult_t r = R.CommDestroy(comm->comm);
    delete comm;
    if (r != ncclSuccess) {
        tf_set_error("tf_comm_destroy: %s", R.GetErrorString(r));
        return TF_ERR_COMM;
    }
    return 0;
}

extern "C" int tf_comm_rank(const tf_comm* comm) { return comm ? comm->rank : -1; }
extern "C" int tf_comm_world(const tf_comm* comm) { return comm ? comm->world : 0; }

extern "C" int tf_allgather_kv(tf_comm* comm, const void* local, void* bank, int64_t elems_per_rank, int dtype,
                               void* stream) {
    TF_ARG(comm && local && bank, TF_ERR_NULL, "tf_allgather_kv: null pointer");
    TF_ARG(dtype == TF_BF16 || dtype == TF_F16 || dtype == TF_F32, TF_ERR_DTYPE, "tf_allgather_kv: dtype %d", dtype);
    TF_ARG(elems_per_rank > 0, TF_ERR_SHAPE, "tf_allgather_kv: elems_per_rank=%lld", (long long)elems_per_rank);
    if (comm->kind != COMM_RCCL) {   // the other transports know the row form only: one row of equal size per rank
        int64_t rows[TF_MAX_WORLD];
        TF_ARG(comm->world <= TF_MAX_WORLD, TF_ERR_SHAPE, "tf_allgather_kv: world %d > %d", comm->world, TF_MAX_WORLD);
        for (int p = 0; p < comm->world; ++p) rows[p] = 1;
        return tf_allgather_rows(comm, local, bank, rows, elems_per_rank, dtype, stream);
    }
    TF_NEED_RCCL("tf_allgather_kv");
    // bytes, not typed elements: a gather moves data, and ncclBfloat16 needs no special casing this way
    TF_NCCL(R.AllGather(local, bank, (size_t)elems_per_rank * elem_bytes(dtype), ncclUint8, comm->comm,
                        reinterpret_cast<hipStream_t>(stream)),
            "tf_allgather_kv");
    return 0;
}

extern "C" int tf_allgather_rows(tf_comm* comm, const void* local, void* bank, const int64_t* rows, int64_t row_elems,
                                 int dtype, void* stream) {
    TF_ARG(comm && local && bank && rows, TF_ERR_NULL, "tf_allgather_rows: null pointer");
    TF_ARG(dtype == TF_BF16 || dtype == TF_F16 || dtype == TF_F32, TF_ERR_DTYPE, "tf_allgather_rows: dtype %d", dtype);
    TF_ARG(row_elems > 0, TF_ERR_SHAPE, "tf_allgather_rows: row_elems=%lld", (long long)row_elems);
    for (int p = 0; p < comm->world; ++p)
        TF_ARG(rows[p] >= 0, TF_ERR_SHAPE, "tf_allgather_rows: negative row count, peer %d", p);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t rb = (size_t)row_elems * elem_bytes(dtype);
    if (comm->kind == COMM_HOOKS) {
        const int rc = comm->hooks.allgather_rows(comm->hooks.user, local, bank, rows, (int64_t)rb, stream);
        if (rc) tf_set_error("tf_allgather_rows: the host transport returned %d", rc);
        return rc ? TF_ERR_COMM : 0;
    }
    if (comm->kind == COMM_LOOPBACK) return loop_allgather_rows(comm, local, bank, rows, rb, st);
    TF_NEED_RCCL("tf_allgather_rows");
    const size_t mine = (size_t)rows[comm->rank] * rb;
    char* r = static_cast<char*>(bank);
    TF_NCCL(R.GroupStart(), "tf_allgather_rows");
    ncclResult_t first = ncclSuccess;
    for (int p = 0; p < comm->world; ++p) {   // every peer (self included) gets this rank's rows; theirs land in rank order
        if (mine) TF_NCCL_IN_GROUP(R.Send(local, mine, ncclUint8, p, comm->comm, st), first);
        if (rows[p]) TF_NCCL_IN_GROUP(R.Recv(r, (size_t)rows[p] * rb, ncclUint8, p, comm->comm, st), first);
        r += (size_t)rows[p] * rb;
    }
    TF_NCCL_GROUP_END(first, "tf_allgather_rows");
    return 0;
}

extern "C" int tf_all_to_all_rows(tf_comm* comm, const void* send, void* recv, const int64_t* send_rows,
                                  const int64_t* recv_rows, int64_t row_elems, int dtype, void* stream) {
    TF_ARG(comm && send && recv && send_rows && recv_rows, TF_ERR_NULL, "tf_all_to_all_rows: null pointer");
    TF_ARG(dtype == TF_BF16 || dtype == TF_F16 || dtype == TF_F32, TF_ERR_DTYPE, "tf_all_to_all_rows: dtype %d", dtype);
    TF_ARG(row_elems > 0, TF_ERR_SHAPE, "tf_all_to_all_rows: row_elems=%lld", (long long)row_elems);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t rb = (size_t)row_elems * elem_bytes(dtype);
    const char* s = static_cast<const char*>(send);
    char* r = static_cast<char*>(recv);
    for (int p = 0; p < comm->world; ++p)
        TF_ARG(send_rows[p] >= 0 && recv_rows[p] >= 0, TF_ERR_SHAPE, "tf_all_to_all_rows: negative row count, peer %d", p);
    if (comm->kind == COMM_HOOKS) {
        const int rc = comm->hooks.all_to_all_rows(comm->hooks.user, send, recv, send_rows, recv_rows, (int64_t)rb, stream);
        if (rc) tf_set_error("tf_all_to_all_rows: the host transport returned %d", rc);
        return rc ? TF_ERR_COMM : 0;
    }
    if (comm->kind == COMM_LOOPBACK) return loop_all_to_all(comm, send, recv, send_rows, recv_rows, rb, st);
    TF_NEED_RCCL("tf_all_to_all_rows");
    TF_NCCL(R.GroupStart(), "tf_all_to_all_rows");
    ncclResult_t first = ncclSuccess;
    for (int p = 0; p < comm->world; ++p) {
        if (send_rows[p]) TF_NCCL_IN_GROUP(R.Send(s, (size_t)send_rows[p] * rb, ncclUint8, p, comm->comm, st), first);
        if (recv_rows[p]) TF_NCCL_IN_GROUP(R.Recv(r, (size_t)recv_rows[p] * rb, ncclUint8, p, comm->comm, st), first);
        s += (size_t)send_rows[p] * rb;
        r += (size_t)recv_rows[p] * rb;
    }
    TF_NCCL_GROUP_END(first, "tf_all_to_all_rows");
    return 0;
}

extern "C" int tf_sendrecv_pivot(tf_comm* comm, const void* const* send, const int64_t* send_elems, int n_send,
                                 int send_peer, void* const* recv, const int64_t* recv_elems, int n_recv,
                                 int recv_peer, int dtype, void* stream) {
    TF_ARG(comm && (n_send == 0 || (send && send_elems)) && (n_recv == 0 || (recv && recv_elems)), TF_ERR_NULL,
           "tf_sendrecv_pivot: null pointer");
    TF_ARG(dtype == TF_BF16 || dtype == TF_F16 || dtype == TF_F32, TF_ERR_DTYPE, "tf_sendrecv_pivot: dtype %d", dtype);
    TF_ARG(n_send >= 0 && n_recv >= 0 && send_peer < comm->world && recv_peer < comm->world, TF_ERR_SHAPE,
           "tf_sendrecv_pivot: n_send=%d n_recv=%d peers=(%d,%d) world=%d", n_send, n_recv, send_peer, recv_peer,
           comm->world);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t eb = elem_bytes(dtype);
    if (comm->kind == COMM_HOOKS) {
        int64_t sb[16], rbs[16];
        TF_ARG(n_send <= 16 && n_recv <= 16, TF_ERR_SHAPE, "tf_sendrecv_pivot: more than 16 messages");
        for (int i = 0; i < n_send; ++i) sb[i] = send_elems[i] * (int64_t)eb;
        for (int i = 0; i < n_recv; ++i) rbs[i] = recv_elems[i] * (int64_t)eb;
        const int rc = comm->hooks.sendrecv(comm->hooks.user, send, sb, n_send, send_peer, recv, rbs, n_recv, recv_peer,
                                            stream);
        if (rc) tf_set_error("tf_sendrecv_pivot: the host transport returned %d", rc);
        return rc ? TF_ERR_COMM : 0;
    }
    if (comm->kind == COMM_LOOPBACK) {   // what arrives has the size of what the left neighbour sends: this rank's own
        if (recv_peer >= 0 && send_peer >= 0)
            for (int i = 0; i < n_recv && i < n_send; ++i) {
                const size_t n = (size_t)(recv_elems[i] < send_elems[i] ? recv_elems[i] : send_elems[i]) * eb;
                if (const int rc = copy_async(recv[i], send[i], n, st, "tf_sendrecv_pivot(loopback)", comm->copies)) return rc;
            }
        size_t out = 0, in = 0;   // one link to the right neighbour, another from the left one
        if (send_peer >= 0)
            for (int i = 0; i < n_send; ++i) out += (size_t)send_elems[i] * eb;
        if (recv_peer >= 0)
            for (int i = 0; i < n_recv; ++i) in += (size_t)recv_elems[i] * eb;
        if (out == 0 && in == 0) return 0;
        return wire_hold(comm, out > in ? out : in, st, "tf_sendrecv_pivot(loopback wire)");
    }
    TF_NEED_RCCL("tf_sendrecv_pivot");
    TF_NCCL(R.GroupStart(), "tf_sendrecv_pivot");
    ncclResult_t first = ncclSuccess;
    if (send_peer >= 0)
        for (int i = 0; i < n_send; ++i)
            TF_NCCL_IN_GROUP(R.Send(send[i], (size_t)send_elems[i] * eb, ncclUint8, send_peer, comm->comm, st), first);
    if (recv_peer >= 0)
        for (int i = 0; i < n_recv; ++i)
            TF_NCCL_IN_GROUP(R.Recv(recv[i], (size_t)recv_elems[i] * eb, ncclUint8, recv_peer, comm->comm, st), first);
    TF_NCCL_GROUP_END(first, "tf_sendrecv_pivot");
    return 0;
}
