// fake_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl that lets several ranks of rt_render_frame_sharded
// run as host threads of ONE process on ONE GPU.  It implements exactly the entry points librt_hip.so binds at run
// time (RT_RCCL_LIB names the library to load): ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclCommAbort,
// ncclSend, ncclRecv, ncclGroupStart, ncclGroupEnd, ncclAllReduce (int32 max/sum of a few words), ncclGetErrorString.
// Ranks meet through a mutex and a condition variable; data moves with hipMemcpy on the shared device.  Every
// operation is synchronous (stronger than RCCL's stream ordering, so the caller's stream logic is still exercised
// in the right order).  Signatures come from <rccl/rccl.h>, so a mismatch with the real library fails to compile.
// Nothing in the product links or loads this file; tests/test_sharded_threads.py does.
//
// Extra exports for tests:
//   fake_rccl_fail_allreduce(rank, times)  the next `times` ncclAllReduce calls of `rank` fail locally
//   fake_rccl_calls(which)                 how many sends (0), receives (1), all-reduces (2), aborts (3) ran so far
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace {

constexpr int kMaxRanks = 16;
constexpr auto kTimeout = std::chrono::seconds(60);   // a rank that never arrives is a test failure, not a hang

struct Message { const void* src = nullptr; size_t bytes = 0; bool posted = false; };

struct Group {
    int n = 0, joined = 0, left = 0;
    std::mutex m;
    std::condition_variable cv;
    // barrier
    int count = 0, generation = 0;
    bool aborted = false;
    // all-reduce contributions
    int32_t words[kMaxRanks][16];
    // point-to-point: box[from][to]
    Message box[kMaxRanks][kMaxRanks];
};

std::mutex g_registry_lock;
std::map<uint64_t, Group*> g_registry;
std::atomic<uint64_t> g_next_id{1};
std::atomic<int> g_fail_allreduce[kMaxRanks];
std::atomic<long> g_calls[4];

// all ranks of the group arrive, or the group is aborted, or the wait times out (both: false)
bool barrier(Group* g, std::unique_lock<std::mutex>& lk) {
    if (g->aborted) return false;
    const int gen = g->generation;
    if (++g->count == g->n) {
        g->count = 0;
        g->generation++;
        g->cv.notify_all();
        return true;
    }
    const bool ok = g->cv.wait_for(lk, kTimeout, [&] { return g->generation != gen || g->aborted; });
    return ok && g->generation != gen;
}

struct Op { bool send; void* buf; size_t bytes; int peer; ncclComm_t comm; hipStream_t stream; };
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

size_t size_of(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

}  // namespace

struct ncclComm { Group* g; int rank; };

namespace {

ncclResult_t run_ops(std::vector<Op>& ops) {
    ncclResult_t rc = ncclSuccess;
    // 1. post the sends (the data must be complete: wait for the sender's stream)
    for (Op& o : ops) {
        if (!o.send) continue;
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
        Group* g = o.comm->g;
        std::unique_lock<std::mutex> lk(g->m);
        Message& b = g->box[o.comm->rank][o.peer];
        b.src = o.buf; b.bytes = o.bytes; b.posted = true;
        g_calls[0]++;
        g->cv.notify_all();
    }
    // 2. receive: wait for the peer's message, copy device to device, mark it taken
    for (Op& o : ops) {
        if (o.send) continue;
        Group* g = o.comm->g;
        const void* src = nullptr;
        {
            std::unique_lock<std::mutex> lk(g->m);
            Message& b = g->box[o.peer][o.comm->rank];
            if (!g->cv.wait_for(lk, kTimeout, [&] { return b.posted || g->aborted; }) || g->aborted) { rc = ncclRemoteError; continue; }
            if (b.bytes != o.bytes) { rc = ncclInvalidArgument; b.posted = false; g->cv.notify_all(); continue; }
            src = b.src;
        }
        if (hipStreamSynchronize(o.stream) != hipSuccess || hipMemcpy(o.buf, src, o.bytes, hipMemcpyDeviceToDevice) != hipSuccess)
            rc = ncclUnhandledCudaError;
        std::unique_lock<std::mutex> lk(g->m);
        g->box[o.peer][o.comm->rank].posted = false;
        g_calls[1]++;
        g->cv.notify_all();
    }
    // 3. a send is complete when the receiver has copied (the sender may then overwrite its buffer)
    for (Op& o : ops) {
        if (!o.send) continue;
        Group* g = o.comm->g;
        std::unique_lock<std::mutex> lk(g->m);
        Message& b = g->box[o.comm->rank][o.peer];
        if (!g->cv.wait_for(lk, kTimeout, [&] { return !b.posted || g->aborted; }) || g->aborted) rc = ncclRemoteError;
    }
    return rc;
}

ncclResult_t leave(ncclComm_t comm, bool abort) {
    if (!comm) return ncclInvalidArgument;
    Group* g = comm->g;
    bool last;
    {
        std::unique_lock<std::mutex> lk(g->m);
        if (abort) { g->aborted = true; g_calls[3]++; g->cv.notify_all(); }
        last = ++g->left == g->n;
    }
    if (last) {
        std::lock_guard<std::mutex> lk(g_registry_lock);
        for (auto it = g_registry.begin(); it != g_registry.end(); ++it)
            if (it->second == g) { g_registry.erase(it); break; }
        delete g;
    }
    delete comm;
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    std::memset(id, 0, sizeof *id);
    const uint64_t v = g_next_id++;
    std::memcpy(id->internal, &v, sizeof v);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    uint64_t key;
    std::memcpy(&key, id.internal, sizeof key);
    Group* g;
    {
        std::lock_guard<std::mutex> lk(g_registry_lock);
        Group*& slot = g_registry[key];
        if (!slot) { slot = new Group(); slot->n = nranks; }
        g = slot;
    }
    if (g->n != nranks) return ncclInvalidArgument;
    std::unique_lock<std::mutex> lk(g->m);
    g->joined++;
    if (!barrier(g, lk)) return ncclRemoteError;      // collective, as in RCCL: returns when every rank has joined
    *comm = new ncclComm{g, rank};
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) { return leave(comm, false); }
ncclResult_t ncclCommAbort(ncclComm_t comm) { return leave(comm, true); }

ncclResult_t ncclGroupStart() { t_depth++; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run_ops(ops);
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!comm || !sendbuff || peer < 0 || peer >= comm->g->n || peer == comm->rank || !size_of(datatype)) return ncclInvalidArgument;
    t_ops.push_back(Op{true, const_cast<void*>(sendbuff), count * size_of(datatype), peer, comm, stream});
    if (t_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run_ops(ops);
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!comm || !recvbuff || peer < 0 || peer >= comm->g->n || peer == comm->rank || !size_of(datatype)) return ncclInvalidArgument;
    t_ops.push_back(Op{false, recvbuff, count * size_of(datatype), peer, comm, stream});
    if (t_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run_ops(ops);
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream) {
    if (!comm || !sendbuff || !recvbuff || count == 0 || count > 16 || datatype != ncclInt32 || (op != ncclMax && op != ncclSum))
        return ncclInvalidArgument;
    Group* g = comm->g;
    if (g_fail_allreduce[comm->rank].load() > 0) { g_fail_allreduce[comm->rank]--; return ncclInternalError; }
    int32_t mine[16];
    if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(mine, sendbuff, count * 4, hipMemcpyDeviceToHost) != hipSuccess)
        return ncclUnhandledCudaError;
    int32_t out[16];
    {
        std::unique_lock<std::mutex> lk(g->m);
        std::memcpy(g->words[comm->rank], mine, count * 4);
        if (!barrier(g, lk)) return ncclRemoteError;
        for (size_t k = 0; k < count; k++) {
            int32_t v = g->words[0][k];
            for (int r = 1; r < g->n; r++) v = op == ncclMax ? (g->words[r][k] > v ? g->words[r][k] : v) : v + g->words[r][k];
            out[k] = v;
        }
        if (!barrier(g, lk)) return ncclRemoteError;  // everybody has read before the next call overwrites
        g_calls[2]++;
    }
    if (hipMemcpy(recvbuff, out, count * 4, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error (fake RCCL)";
        case ncclUnhandledCudaError: return "unhandled HIP error (fake RCCL)";
        case ncclInternalError: return "internal error (fake RCCL)";
        case ncclInvalidArgument: return "invalid argument (fake RCCL)";
        case ncclInvalidUsage: return "invalid usage (fake RCCL)";
        case ncclRemoteError: return "remote rank failed or never arrived (fake RCCL)";
        default: return "error (fake RCCL)";
    }
}

void fake_rccl_fail_allreduce(int rank, int times) {
    if (rank >= 0 && rank < kMaxRanks) g_fail_allreduce[rank] = times;
}

long fake_rccl_calls(int which) { return which >= 0 && which < 4 ? g_calls[which].load() : -1; }

}  // extern "C"
