// Panel-edge halo exchange over RCCL point-to-point, behind the C ABI.
//
// Replaces reference wx_factory/process_topology.py:259-261 (the dist-graph communicator), :269-386
// (start_exchange_scalars / start_exchange_vectors: rotate, flip, pack, device.synchronize(), Ineighbor_alltoall) and
// :564-606 (ExchangeRequest.wait).  The rotation, the flip and the packing are done by the extrapolation kernels
// (wx_euler3d_extrap_pack / wx_sw_extrap_pack write the neighbour's q_itf_{s,n,w,e}); what is left for this file is the
// movement: ONE message per tile edge instead of the reference's three, all of a rank's messages in one
// ncclGroupStart ... ncclSend / ncclRecv ... ncclGroupEnd on a communication stream that is forked from the compute stream
// with an event (no device.synchronize()), joined again by wx_exchange_wait - the interior elements are evaluated in
// between.  Messages between two tiles of the same rank never move: the receiver's halo pointer IS the sender's slot.
// The fork / join is two event records and two stream waits, which is also the shape a HIP-graph capture records: an
// evaluation with the exchange in flight beside the interior launch captures and replays as one graph.
#include "wx_common.h"
#include "wx_panels.h"

#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <array>
#include <map>
#include <new>
#include <vector>

using namespace wx;

#define WX_NCCL_TRY(expr)                                                                                      \
    do {                                                                                                       \
        ncclResult_t _r = (expr);                                                                              \
        if (_r != ncclSuccess)                                                                                 \
            return ::wx::fail(WX_ERR_COMM, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

struct wx_comm {
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0;
    bool owned = false;
    int users = 0;   // exchanges alive on this communicator: wx_comm_destroy refuses while there are any
};

struct wx_exchange {
    wx_comm* comm = nullptr;
    int rank = 0, world = 1, k = 1;
    size_t ec = 0;          // doubles per edge message
    bool loopback = false;
    std::vector<int> local;                               // tiles of this rank, ascending
    std::vector<size_t> send_count, recv_count;           // per peer rank, in doubles
    std::vector<size_t> send_off, recv_off;
    std::map<std::array<int, 2>, size_t> send_slot;       // (tile, edge) -> slot of the send buffer
    std::map<std::array<int, 2>, std::pair<int, size_t>> halo_src;   // (tile, edge) -> (0 send buffer | 1 recv buffer, slot)
    size_t n_remote_out = 0, n_remote_in = 0, n_send_slots = 0;
    double *send_buf = nullptr, *recv_buf = nullptr;
    bool own_bufs = false, bound = false;
    hipEvent_t fork = nullptr, join = nullptr;
    bool pending = false;   // a forked exchange is in flight: wx_exchange_wait has a join to make
    wx_phase_timer* timer = nullptr;   // wx_exchange_set_timer: the *_rhs_overlapped calls stamp the reference's timing row
};

namespace {

wx_status build_layout(wx_exchange* ex) {
    const CubeTiles T{ex->k};
    const int nt = T.ntiles(), rank = ex->rank, world = ex->world;
    auto owner = [&](int t) { return tile_owner(t, world, nt); };
    for (int t = 0; t < nt; ++t)
        if (owner(t) == rank) ex->local.push_back(t);
    // messages this rank sends: to another rank (or, in loopback mode, through the communicator to itself), keyed for the
    // canonical order inside a rank pair - by (destination tile, destination edge) - or to a tile of its own (no movement)
    std::vector<std::vector<std::array<int, 4>>> out(world), in(world);   // out[r]: (q, e2, p, e);  in[r]: (q, e2)
    std::vector<std::array<int, 4>> local_msgs;                           // (p, e, q, e2)
    for (int p : ex->local)
        for (int e = 0; e < 4; ++e) {
            const int q = T.neighbor(p, e), e2 = T.landing(p, e);
            if (e2 < 0) return fail(WX_ERR_INVALID, "wx_exchange_create: no landing edge between tiles %d and %d", p, q);
            if (owner(q) == rank && !ex->loopback) local_msgs.push_back({p, e, q, e2});
            else out[owner(q)].push_back({q, e2, p, e});
        }
    for (int q : ex->local)
        for (int e2 = 0; e2 < 4; ++e2) {
            const int p = T.neighbor(q, e2);
            if (owner(p) != rank || ex->loopback) in[owner(p)].push_back({q, e2, 0, 0});
        }
    ex->send_count.assign(world, 0); ex->recv_count.assign(world, 0);
    ex->send_off.assign(world, 0); ex->recv_off.assign(world, 0);
    size_t slot = 0;
    for (int r = 0; r < world; ++r) {
        std::sort(out[r].begin(), out[r].end());
        ex->send_off[r] = slot * ex->ec;
        ex->send_count[r] = out[r].size() * ex->ec;
        for (const auto& m : out[r]) ex->send_slot[{m[2], m[3]}] = slot++;
    }
    ex->n_remote_out = slot;
    for (const auto& m : local_msgs) {
        ex->send_slot[{m[0], m[1]}] = slot;
        ex->halo_src[{m[2], m[3]}] = {0, slot};
        ++slot;
    }
    ex->n_send_slots = slot;
    size_t rslot = 0;
    for (int r = 0; r < world; ++r) {
        std::sort(in[r].begin(), in[r].end());
        ex->recv_off[r] = rslot * ex->ec;
        ex->recv_count[r] = in[r].size() * ex->ec;
        for (const auto& m : in[r]) ex->halo_src[{m[0], m[1]}] = {1, rslot++};
    }
    ex->n_remote_in = rslot;
    return WX_OK;
}

bool needs_comm(const wx_exchange* ex) { return ex->world > 1 || ex->loopback; }

// hipRuntimeGetVersion of the first runtime whose Stream::EndCapture survives RCCL on a forked stream of a capture
// (HIP_VERSION = major * 10^7 + minor * 10^5 + patch: 7.0.2 in torch 2.10 wheels reports 70051831, ROCm 7.2 70226015)
constexpr int kForkedCaptureMinRuntime = 70200000;

hipError_t make_events(wx_exchange* ex) {
    hipError_t e = hipSuccess;
    if (!ex->fork) e = hipEventCreateWithFlags(&ex->fork, hipEventDisableTiming);
    if (e == hipSuccess && !ex->join) e = hipEventCreateWithFlags(&ex->join, hipEventDisableTiming);
    return e;
}

}  // namespace

extern "C" {

int wx_comm_rccl_version(void) {
    int v = 0;
    if (ncclGetVersion(&v) != ncclSuccess) return -1;
    return v;
}

wx_status wx_comm_unique_id(unsigned char id[WX_COMM_ID_BYTES]) {
    static_assert(WX_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "wxhip.h and rccl.h disagree on the size of a unique id");
    if (!id) return fail(WX_ERR_INVALID, "wx_comm_unique_id: null argument");
    ncclUniqueId u;
    WX_NCCL_TRY(ncclGetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return WX_OK;
}

wx_status wx_comm_init_rank(wx_comm** out, int nranks, const unsigned char id[WX_COMM_ID_BYTES], int rank) {
    if (!out || !id) return fail(WX_ERR_INVALID, "wx_comm_init_rank: null argument");
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(WX_ERR_INVALID, "wx_comm_init_rank: rank %d of %d", rank, nranks);
    wx_comm* c = new (std::nothrow) wx_comm();
    if (!c) return fail(WX_ERR_NOMEM, "out of host memory");
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(WX_ERR_COMM, "ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, ncclGetErrorString(r));
    }
    c->nranks = nranks; c->rank = rank; c->owned = true;
    *out = c;
    return WX_OK;
}

wx_status wx_comm_adopt(wx_comm** out, void* nccl_comm, int nranks, int rank) {
    if (!out || !nccl_comm) return fail(WX_ERR_INVALID, "wx_comm_adopt: null argument");
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(WX_ERR_INVALID, "wx_comm_adopt: rank %d of %d", rank, nranks);
    wx_comm* c = new (std::nothrow) wx_comm();
    if (!c) return fail(WX_ERR_NOMEM, "out of host memory");
    c->comm = static_cast<ncclComm_t>(nccl_comm); c->nranks = nranks; c->rank = rank; c->owned = false;
    *out = c;
    return WX_OK;
}

wx_status wx_comm_destroy(wx_comm* c) {
    if (!c) return WX_OK;
    if (c->users > 0)
        return fail(WX_ERR_INVALID, "wx_comm_destroy: %d exchange(s) made on this communicator are alive (wx_exchange_destroy them "
                    "first; HIP graphs that hold its RCCL nodes too)", c->users);
    ncclResult_t r = ncclSuccess;
    if (c->owned && c->comm) r = ncclCommDestroy(c->comm);
    delete c;
    if (r != ncclSuccess) return fail(WX_ERR_COMM, "ncclCommDestroy failed: %s", ncclGetErrorString(r));
    return WX_OK;
}

int wx_comm_users(const wx_comm* c) { return c ? c->users : -1; }

// solvers/global_operations.py:14-36, kiops.py:165-200, pmex.py:150-173, fgmres.py:41, simulation.py:399-408: the small
// reductions of the Krylov callers, in place, on the caller's stream (a graph node under capture: origin stream)
wx_status wx_comm_allreduce(wx_comm* c, double* buf, size_t count, wx_reduce_op op, wx_stream stream) {
    if (!c || !c->comm) return fail(WX_ERR_INVALID, "wx_comm_allreduce: null communicator");
    if (count == 0) return WX_OK;
    if (!buf) return fail(WX_ERR_INVALID, "wx_comm_allreduce: null buffer");
    ncclRedOp_t o;
    switch (op) {
        case WX_REDUCE_SUM: o = ncclSum; break;
        case WX_REDUCE_MAX: o = ncclMax; break;
        case WX_REDUCE_MIN: o = ncclMin; break;
        default: return fail(WX_ERR_INVALID, "wx_comm_allreduce: unknown reduction %d", (int)op);
    }
    WX_STREAM(st, stream);
    WX_NCCL_TRY(ncclAllReduce(buf, buf, count, ncclDouble, o, c->comm, st));
    return WX_OK;
}

int wx_hip_runtime_version(void) {
    int v = 0;
    if (hipRuntimeGetVersion(&v) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return v;
}

int wx_hip_driver_version(void) {
    int v = 0;
    if (hipDriverGetVersion(&v) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return v;
}

wx_status wx_exchange_create(wx_exchange** out, wx_comm* comm, int rank, int world, int tiles_per_side, size_t edge_doubles,
                             int loopback) {
    if (!out) return fail(WX_ERR_INVALID, "wx_exchange_create: null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(WX_ERR_INVALID, "wx_exchange_create: rank %d of %d", rank, world);
    if (tiles_per_side < 1 || tiles_per_side > 64) return fail(WX_ERR_INVALID, "wx_exchange_create: %d tiles per panel side", tiles_per_side);
    if (edge_doubles == 0) return fail(WX_ERR_INVALID, "wx_exchange_create: empty edge messages");
    if (comm && (comm->nranks != world || comm->rank != rank))
        return fail(WX_ERR_INVALID, "wx_exchange_create: the communicator is rank %d of %d, the exchange rank %d of %d",
                    comm->rank, comm->nranks, rank, world);
    wx_exchange* ex = new (std::nothrow) wx_exchange();
    if (!ex) return fail(WX_ERR_NOMEM, "out of host memory");
    ex->comm = comm; ex->rank = rank; ex->world = world; ex->k = tiles_per_side; ex->ec = edge_doubles;
    ex->loopback = loopback != 0;
    wx_status st;
    try {   // (the layout lives in standard containers: no exception crosses the C boundary)
        st = build_layout(ex);
    } catch (...) {
        st = fail(WX_ERR_NOMEM, "wx_exchange_create: out of host memory");
    }
    if (st != WX_OK) { delete ex; return st; }
    if (comm) ++comm->users;
    *out = ex;
    return WX_OK;
}

wx_status wx_exchange_destroy(wx_exchange* ex) {
    if (!ex) return WX_OK;
    hipError_t e = hipSuccess;
    if (ex->fork) e = hipEventDestroy(ex->fork);
    if (ex->join) (void)hipEventDestroy(ex->join);
    if (ex->own_bufs) {
        if (ex->send_buf) (void)hipFree(ex->send_buf);
        if (ex->recv_buf) (void)hipFree(ex->recv_buf);
    }
    if (ex->comm && ex->comm->users > 0) --ex->comm->users;
    delete ex;
    if (e != hipSuccess) return fail(WX_ERR_HIP, "hipEventDestroy failed: %s", hipGetErrorString(e));
    return WX_OK;
}

int wx_exchange_local_tiles(const wx_exchange* ex, int* tiles, int capacity) {
    if (!ex) return -1;
    if (tiles)
        for (int i = 0; i < (int)ex->local.size() && i < capacity; ++i) tiles[i] = ex->local[i];
    return (int)ex->local.size();
}

wx_status wx_exchange_neighbor(const wx_exchange* ex, int tile, int edge, int* neighbor_tile, int* landing_edge,
                               int* neighbor_rank) {
    if (!ex) return fail(WX_ERR_INVALID, "wx_exchange_neighbor: null exchange");
    const CubeTiles T{ex->k};
    if (tile < 0 || tile >= T.ntiles() || edge < 0 || edge > 3)
        return fail(WX_ERR_INVALID, "wx_exchange_neighbor: tile %d, edge %d", tile, edge);
    const int q = T.neighbor(tile, edge);
    if (neighbor_tile) *neighbor_tile = q;
    if (landing_edge) *landing_edge = T.landing(tile, edge);
    if (neighbor_rank) *neighbor_rank = tile_owner(q, ex->world, T.ntiles());
    return WX_OK;
}

int wx_exchange_needs_comm(const wx_exchange* ex) { return ex && needs_comm(ex) ? 1 : 0; }

size_t wx_exchange_send_doubles(const wx_exchange* ex) { return ex ? ex->n_send_slots * ex->ec : 0; }
size_t wx_exchange_recv_doubles(const wx_exchange* ex) { return ex ? (ex->n_remote_in ? ex->n_remote_in : 1) * ex->ec : 0; }

wx_status wx_exchange_peer_counts(const wx_exchange* ex, size_t* send_doubles, size_t* recv_doubles) {
    if (!ex) return fail(WX_ERR_INVALID, "wx_exchange_peer_counts: null exchange");
    for (int r = 0; r < ex->world; ++r) {
        if (send_doubles) send_doubles[r] = ex->send_count[r];
        if (recv_doubles) recv_doubles[r] = ex->recv_count[r];
    }
    return WX_OK;
}

wx_status wx_exchange_bind(wx_exchange* ex, double* send_buf, double* recv_buf) {
    if (!ex) return fail(WX_ERR_INVALID, "wx_exchange_bind: null exchange");
    if (ex->bound) return fail(WX_ERR_INVALID, "wx_exchange_bind: the exchange has its buffers already");
    if ((send_buf || recv_buf) && (!recv_buf || (!send_buf && wx_exchange_send_doubles(ex) > 0)))
        return fail(WX_ERR_INVALID, "wx_exchange_bind: give both buffers or neither");
    if (!send_buf && !recv_buf) {
        const size_t sb = wx_exchange_send_doubles(ex) * sizeof(double), rb = wx_exchange_recv_doubles(ex) * sizeof(double);
        hipError_t e = hipMalloc((void**)&ex->send_buf, sb ? sb : sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void**)&ex->recv_buf, rb);
        if (e == hipSuccess) e = hipMemset(ex->send_buf, 0, sb ? sb : sizeof(double));
        if (e == hipSuccess) e = hipMemset(ex->recv_buf, 0, rb);
        if (e != hipSuccess) {
            if (ex->send_buf) (void)hipFree(ex->send_buf);
            if (ex->recv_buf) (void)hipFree(ex->recv_buf);
            ex->send_buf = ex->recv_buf = nullptr;
            return fail(WX_ERR_NOMEM, "wx_exchange_bind: edge buffers of %zu + %zu bytes: %s", sb, rb, hipGetErrorString(e));
        }
        ex->own_bufs = true;
    } else {
        ex->send_buf = send_buf; ex->recv_buf = recv_buf;
    }
    ex->bound = true;
    // the fork / join events (no timing: cheaper).  A host without a GPU can still bind - the layout queries and their
    // tests need no device - and wx_exchange_start reports the missing events
    if (needs_comm(ex) && make_events(ex) != hipSuccess) (void)hipGetLastError();
    return WX_OK;
}

void* wx_exchange_send_ptr(const wx_exchange* ex, int tile, int edge) {
    if (!ex || !ex->bound) return nullptr;
    auto it = ex->send_slot.find({tile, edge});
    return it == ex->send_slot.end() ? nullptr : ex->send_buf + it->second * ex->ec;
}

const void* wx_exchange_halo_ptr(const wx_exchange* ex, int tile, int edge) {
    if (!ex || !ex->bound) return nullptr;
    auto it = ex->halo_src.find({tile, edge});
    if (it == ex->halo_src.end()) return nullptr;
    return (it->second.first == 0 ? ex->send_buf : ex->recv_buf) + it->second.second * ex->ec;
}

wx_status wx_exchange_start(wx_exchange* ex, wx_stream compute, wx_stream comm_stream) {
    if (!ex) return fail(WX_ERR_INVALID, "wx_exchange_start: null exchange");
    if (!needs_comm(ex)) return WX_OK;   // every neighbour is a tile of this rank: the halos alias the send slots
    if (!ex->bound) return fail(WX_ERR_INVALID, "wx_exchange_start: call wx_exchange_bind first");
    if (!ex->comm) return fail(WX_ERR_INVALID, "wx_exchange_start: messages travel, but the exchange was created without a communicator");
    if (ex->pending) return fail(WX_ERR_INVALID, "wx_exchange_start: the previous exchange was not waited for");
    WX_STREAM(cs, compute);
    hipStream_t ms = static_cast<hipStream_t>(comm_stream);
    const bool forked = ms != nullptr && ms != cs;
    if (forked) {
        // RCCL on a NON-origin stream of a capture: hip::Stream::EndCapture of HIP runtimes before 7.2 (the 7.0.2 inside
        // torch 2.10 wheels) then recurses over a cycle of parallel-capture lists until the stack is gone - a SIGSEGV at
        // hipStreamEndCapture, no error code (profiles/r04_capture_crash.md).  Refused here, where it can still be said.
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        WX_HIP_TRY(hipStreamIsCapturing(cs, &cap));
        if (cap != hipStreamCaptureStatusNone) {
            int ver = 0;
            WX_HIP_TRY(hipRuntimeGetVersion(&ver));
            if (ver < kForkedCaptureMinRuntime)
                return fail(WX_ERR_INVALID, "wx_exchange_start: `compute` is being captured and the exchange is forked to another "
                            "stream; the HIP runtime this process bound (%d) ends such a capture with a stack overflow in "
                            "hip::Stream::EndCapture (fixed in 7.2 = %d).  Capture the exchange on the capture's origin stream: "
                            "wx_exchange_start(ex, compute, compute) with the INTERIOR launches forked instead "
                            "(wx_exchange_fork / _join, wx_*_rhs_overlapped)", ver, kForkedCaptureMinRuntime);
        }
    }
    if (forked) {   // the messages were packed on the compute stream: the communication stream starts behind them
        if (!ex->fork || !ex->join) WX_HIP_TRY(make_events(ex));
        WX_HIP_TRY(hipEventRecord(ex->fork, cs));
        WX_HIP_TRY(hipStreamWaitEvent(ms, ex->fork, 0));
    } else {
        ms = cs;
    }
    ncclComm_t c = ex->comm->comm;
    WX_NCCL_TRY(ncclGroupStart());
    ncclResult_t r = ncclSuccess;
    for (int p = 0; p < ex->world && r == ncclSuccess; ++p)
        if (ex->recv_count[p]) r = ncclRecv(ex->recv_buf + ex->recv_off[p], ex->recv_count[p], ncclDouble, p, c, ms);
    for (int p = 0; p < ex->world && r == ncclSuccess; ++p)
        if (ex->send_count[p]) r = ncclSend(ex->send_buf + ex->send_off[p], ex->send_count[p], ncclDouble, p, c, ms);
    ncclResult_t rg = ncclGroupEnd();
    if (r != ncclSuccess) return fail(WX_ERR_COMM, "ncclSend / ncclRecv failed: %s", ncclGetErrorString(r));
    if (rg != ncclSuccess) return fail(WX_ERR_COMM, "ncclGroupEnd failed: %s", ncclGetErrorString(rg));
    if (forked) {
        WX_HIP_TRY(hipEventRecord(ex->join, ms));
        ex->pending = true;
    }
    return WX_OK;
}

wx_status wx_exchange_wait(wx_exchange* ex, wx_stream compute) {
    if (!ex) return fail(WX_ERR_INVALID, "wx_exchange_wait: null exchange");
    if (!ex->pending) return WX_OK;   // nothing travelled, or the exchange was enqueued on the compute stream itself
    WX_STREAM(cs, compute);
    WX_HIP_TRY(hipStreamWaitEvent(cs, ex->join, 0));
    ex->pending = false;
    return WX_OK;
}

// Event fork / join between the compute stream and a second stream, for the arrangement in which the SECOND stream carries
// the interior launches and the exchange stays on the compute stream (see wx_euler3d_rhs_overlapped)
wx_status wx_exchange_fork(wx_exchange* ex, wx_stream compute, wx_stream side) {
    if (!ex || !side) return fail(WX_ERR_INVALID, "wx_exchange_fork: null argument");
    WX_STREAM(cs, compute);
    if (!ex->fork || !ex->join) WX_HIP_TRY(make_events(ex));
    WX_HIP_TRY(hipEventRecord(ex->fork, cs));
    WX_HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(side), ex->fork, 0));
    return WX_OK;
}

wx_status wx_exchange_join(wx_exchange* ex, wx_stream compute, wx_stream side) {
    if (!ex || !side) return fail(WX_ERR_INVALID, "wx_exchange_join: null argument");
    WX_STREAM(cs, compute);
    if (!ex->fork || !ex->join) WX_HIP_TRY(make_events(ex));
    WX_HIP_TRY(hipEventRecord(ex->join, static_cast<hipStream_t>(side)));
    WX_HIP_TRY(hipStreamWaitEvent(cs, ex->join, 0));
    return WX_OK;
}

wx_status wx_exchange_set_timer(wx_exchange* ex, wx_phase_timer* timer) {
    if (!ex) return fail(WX_ERR_INVALID, "wx_exchange_set_timer: null exchange");
    ex->timer = timer;
    return WX_OK;
}

}  // extern "C" (the shared body of the two *_rhs_overlapped entry points is a template)

namespace {

// The evaluation of a rank's tiles with the exchange in flight beside the interior elements (rhs/rhs.py:88-118 with
// process_topology.py:269-386, 564-606 in between): pack on `compute`; fork `side` off it for the INTERIOR launches; the
// grouped sends / receives and then the BOUNDARY launches on `compute`; join.  One launch for ALL elements of each tile when
// nothing travels; side == NULL or == compute: everything in stream order on `compute`.
// pack(plan, q, send[4], stream), eval(plan, q, halo[4] | nullptr, rhs, region, stream): the Euler or shallow-water calls.
template <typename Plan, typename Pack, typename Eval>
wx_status overlapped_body(Plan* const plans[], int count, wx_exchange* ex, const void* const q[], void* const rhs[],
                          wx_stream compute, wx_stream side, bool& forked_open, Pack pack, Eval eval) {
    wx_status st;
    auto stamp = [&](int slot, wx_stream s) -> wx_status {
        return ex->timer ? wx_phase_timer_stamp(ex->timer, slot, s) : WX_OK;
    };
    if ((st = stamp(0, compute)) != WX_OK) return st;
    for (int i = 0; i < count; ++i) {
        void* send[4];
        for (int e = 0; e < 4; ++e) send[e] = wx_exchange_send_ptr(ex, ex->local[i], e);
        if ((st = pack(plans[i], q[i], send, compute)) != WX_OK) return st;
    }
    if ((st = stamp(1, compute)) != WX_OK) return st;
    const bool split = needs_comm(ex);
    // the second stream takes the INTERIOR launches; the exchange stays on `compute` (see the header: on the HIP runtime
    // that ships inside torch 2.10 a capture survives RCCL launches only on its origin stream)
    const bool forked = split && side != nullptr && side != compute;
    if (forked) {   // INTERIOR first: enqueued before the exchange occupies the host thread; stamps 2, 3 bracket it on ITS stream
        if ((st = wx_exchange_fork(ex, compute, side)) != WX_OK) return st;
        forked_open = true;   // from here on an error return must still join `side` (the caller of this body does)
        if ((st = stamp(2, side)) != WX_OK) return st;
        for (int i = 0; i < count; ++i)
            if ((st = eval(plans[i], q[i], nullptr, rhs[i], WX_REGION_INTERIOR, side)) != WX_OK) return st;
        if ((st = stamp(3, side)) != WX_OK) return st;
    }
    if ((st = wx_exchange_start(ex, compute, compute)) != WX_OK) return st;
    if (!forked) {
        if ((st = stamp(2, compute)) != WX_OK) return st;
        if (split)
            for (int i = 0; i < count; ++i)
                if ((st = eval(plans[i], q[i], nullptr, rhs[i], WX_REGION_INTERIOR, compute)) != WX_OK) return st;
        if ((st = stamp(3, compute)) != WX_OK) return st;
    }
    if ((st = stamp(5, compute)) != WX_OK) return st;   // the halos are there (in stream order on `compute`)
    for (int i = 0; i < count; ++i) {
        const void* halo[4];
        for (int e = 0; e < 4; ++e) halo[e] = wx_exchange_halo_ptr(ex, ex->local[i], e);
        if ((st = eval(plans[i], q[i], halo, rhs[i], split ? WX_REGION_BOUNDARY : WX_REGION_ALL, compute)) != WX_OK) return st;
    }
    if (forked) {
        forked_open = false;
        if ((st = wx_exchange_join(ex, compute, side)) != WX_OK) return st;
    }
    return stamp(8, compute);
}

template <typename Plan, typename Pack, typename Eval, typename Words>
wx_status rhs_overlapped(const char* who, Plan* const plans[], int count, wx_exchange* ex, const void* const q[],
                         void* const rhs[], wx_stream compute, wx_stream side, Pack pack, Eval eval, Words words_of) {
    if (!plans || !ex || !q || !rhs) return fail(WX_ERR_INVALID, "%s: null argument", who);
    if (count != (int)ex->local.size())
        return fail(WX_ERR_INVALID, "%s: %d plans for the %d tiles of this rank", who, count, (int)ex->local.size());
    if (!ex->bound) return fail(WX_ERR_INVALID, "%s: call wx_exchange_bind first", who);
    for (int i = 0; i < count; ++i) {
        if (!plans[i] || !q[i] || !rhs[i]) return fail(WX_ERR_INVALID, "%s: null entry %d", who, i);
        const size_t words = words_of(plans[i]);   // a mismatch would let the pack kernels write past their send slots
        if (words != ex->ec)
            return fail(WX_ERR_INVALID, "%s: plan %d packs %zu doubles per edge, the exchange moves %zu", who, i, words, ex->ec);
    }
    bool forked_open = false;
    const wx_status st = overlapped_body(plans, count, ex, q, rhs, compute, side, forked_open, pack, eval);
    if (st != WX_OK) {
        // leave the object usable and the streams joined: `side` rejoins `compute` (a capture of `compute` could not end
        // with a forked stream still open), no exchange is left marked in flight; the FIRST error is the one reported
        char first[512];
        snprintf(first, sizeof first, "%s", wx_last_error());
        if (forked_open) (void)wx_exchange_join(ex, compute, side);
        ex->pending = false;
        snprintf(last_error_buf(), 512, "%s", first);
    }
    return st;
}

}  // namespace

extern "C" {

wx_status wx_euler3d_rhs_overlapped(wx_euler3d_plan* const plans[], int count, wx_exchange* ex, const void* const q[],
                                    void* const rhs[], wx_stream compute, wx_stream side) {
    return rhs_overlapped(
        "wx_euler3d_rhs_overlapped", plans, count, ex, q, rhs, compute, side,
        [](wx_euler3d_plan* pl, const void* qq, void* const send[4], wx_stream s) { return wx_euler3d_extrap_pack(pl, qq, send, s); },
        [](wx_euler3d_plan* pl, const void* qq, const void* const halo[4], void* r, wx_region reg, wx_stream s) {
            return wx_euler3d_rhs(pl, qq, halo, r, reg, s);
        },
        [](const wx_euler3d_plan* pl) { return wx_euler3d_edge_count(pl) * (wx_euler3d_plan_dtype(pl) == WX_F64 ? 1 : 2); });
}

// the shallow-water twin (rhs/rhs_sw.py:76-150)
wx_status wx_sw_rhs_overlapped(wx_sw_plan* const plans[], int count, wx_exchange* ex, const void* const q[], void* const rhs[],
                               wx_stream compute, wx_stream side) {
    return rhs_overlapped(
        "wx_sw_rhs_overlapped", plans, count, ex, q, rhs, compute, side,
        [](wx_sw_plan* pl, const void* qq, void* const send[4], wx_stream s) { return wx_sw_extrap_pack(pl, qq, send, s); },
        [](wx_sw_plan* pl, const void* qq, const void* const halo[4], void* r, wx_region reg, wx_stream s) {
            return wx_sw_rhs(pl, qq, halo, r, reg, s);
        },
        [](const wx_sw_plan* pl) { return wx_sw_edge_count(pl) * (wx_sw_plan_dtype(pl) == WX_F64 ? 1 : 2); });
}

}  // extern "C"
