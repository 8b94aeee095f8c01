// tests/c/mock_rccl.cpp -> <dir>/librccl.so.1 -- TEST ONLY.  An in-process stand-in for the eleven RCCL entry points
// libsdfgrid binds at run time (csrc/slab_comm.hip: dlopen("librccl.so.1")), so that the library's MULTI-RANK code -- ranks with
// one neighbour or two, the ghost offsets of non-periodic ends, the gathers, sdfv_slab_march's rounds -- can run on ONE GPU:
// RCCL itself refuses two ranks on one device, and this pool has never offered two devices.  Every rank is a host THREAD of one
// process, all on the same HIP device; a "communicator" is an entry in a process-wide table.
//
// Semantics kept from NCCL: sends and receives are stream-ordered; the k-th ncclSend from a to b matches the k-th ncclRecv on b
// from a; the operations of a group are posted together at the outermost ncclGroupEnd; counts must match.  Implementation: at
// ncclGroupEnd a thread first REGISTERS all of its group's operations (recording, on each operation's stream, the event that
// marks "my buffer is ready / free from here on"), then completes them in order -- the sender's thread waits for the matching
// receive to be registered, makes its stream wait for the receiver's event, enqueues the device-to-device copy and records a
// completion event; the receiver's thread waits for that and makes its stream wait for the completion event.  Every thread
// registers everything before it waits for anything, so matched groups cannot deadlock; an operation that is never matched (a
// bug in the caller's pairing) times out after 30 s with a message naming it.  ncclAllReduce (int32 sum, the sharded march's
// merge) is done on the host: synchronise, sum, copy back.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

namespace {

struct Op {
    bool is_send = false;
    void* buf = nullptr;
    size_t bytes = 0;
    int src = 0, dst = 0;
    uint64_t seq = 0;
    hipStream_t stream = nullptr;
    hipEvent_t own = nullptr;   // send: data produced; recv: buffer may be overwritten
    hipEvent_t done = nullptr;  // recorded by the sender's thread behind the copy
    bool copy_enqueued = false;
};

struct World {
    std::mutex m;
    std::condition_variable cv;
    int nranks = 0, joined = 0, left = 0;
    std::map<std::tuple<int, int, uint64_t>, Op*> sends, recvs;  // (src, dst, seq)
    std::map<std::pair<int, int>, uint64_t> send_seq, recv_seq;
    // all-reduce: one generation at a time
    int ar_arrived = 0, ar_departed = 0;
    uint64_t ar_generation = 0;
    std::vector<int32_t> ar_sum;
    bool failed = false;
};

struct Comm {
    std::shared_ptr<World> world;
    int rank = 0;
};

std::mutex g_worlds_m;
std::map<std::string, std::shared_ptr<World>> g_worlds;
uint64_t g_next_id = 1;

thread_local int t_group_depth = 0;
thread_local std::vector<std::pair<Comm*, Op*>> t_group;

constexpr int kSuccess = 0, kInternalError = 3, kInvalidArgument = 4;
const std::chrono::seconds kTimeout(30);

size_t type_bytes(int datatype) {
    switch (datatype) {
        case 0: case 1: return 1;  // int8 / uint8
        case 2: case 3: case 7: return 4;  // int32 / uint32 / float32
        case 4: case 5: case 8: return 8;
        case 6: return 2;
        default: return 0;
    }
}

bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    fprintf(stderr, "[mock rccl] %s: %s\n", what, hipGetErrorString(e));
    return false;
}

int complete_group(std::vector<std::pair<Comm*, Op*>>& ops) {
    int rc = kSuccess;
    // 1. register everything
    for (auto& [comm, op] : ops) {
        World& w = *comm->world;
        std::unique_lock<std::mutex> lock(w.m);
        if (!hip_ok(hipEventCreateWithFlags(&op->own, hipEventDisableTiming), "hipEventCreate") ||
            !hip_ok(hipEventCreateWithFlags(&op->done, hipEventDisableTiming), "hipEventCreate") ||
            !hip_ok(hipEventRecord(op->own, op->stream), "hipEventRecord"))
            rc = kInternalError;
        auto& seq = op->is_send ? w.send_seq[{op->src, op->dst}] : w.recv_seq[{op->src, op->dst}];
        op->seq = seq++;
        (op->is_send ? w.sends : w.recvs)[{op->src, op->dst, op->seq}] = op;
        w.cv.notify_all();
    }
    // 2. complete in order
    for (auto& [comm, op] : ops) {
        World& w = *comm->world;
        std::unique_lock<std::mutex> lock(w.m);
        const auto key = std::make_tuple(op->src, op->dst, op->seq);
        if (op->is_send) {
            if (!w.cv.wait_for(lock, kTimeout, [&] { return w.recvs.count(key) != 0 || w.failed; }) || w.failed) {
                fprintf(stderr, "[mock rccl] rank %d: send #%llu to rank %d (%zu bytes) was never matched by a receive\n", op->src,
                        (unsigned long long)op->seq, op->dst, op->bytes);
                w.failed = true;
                w.cv.notify_all();
                rc = kInternalError;
                continue;
            }
            Op* r = w.recvs[key];
            if (r->bytes != op->bytes) {
                fprintf(stderr, "[mock rccl] rank %d -> %d #%llu: send of %zu bytes meets a receive of %zu\n", op->src, op->dst,
                        (unsigned long long)op->seq, op->bytes, r->bytes);
                rc = kInvalidArgument;
            }
            const size_t n = op->bytes < r->bytes ? op->bytes : r->bytes;
            if (!hip_ok(hipStreamWaitEvent(op->stream, r->own, 0), "hipStreamWaitEvent") ||
                (n && !hip_ok(hipMemcpyAsync(r->buf, op->buf, n, hipMemcpyDeviceToDevice, op->stream), "hipMemcpyAsync")) ||
                !hip_ok(hipEventRecord(op->done, op->stream), "hipEventRecord"))
                rc = kInternalError;
            op->copy_enqueued = true;
            w.cv.notify_all();
        } else {
            if (!w.cv.wait_for(lock, kTimeout, [&] { return (w.sends.count(key) != 0 && w.sends[key]->copy_enqueued) || w.failed; }) || w.failed) {
                fprintf(stderr, "[mock rccl] rank %d: receive #%llu from rank %d (%zu bytes) was never matched by a send\n", op->dst,
                        (unsigned long long)op->seq, op->src, op->bytes);
                w.failed = true;
                w.cv.notify_all();
                rc = kInternalError;
                continue;
            }
            if (!hip_ok(hipStreamWaitEvent(op->stream, w.sends[key]->done, 0), "hipStreamWaitEvent")) rc = kInternalError;
        }
    }
    ops.clear();  // (the Op records and their events live as long as the process: a test's worth of them)
    return rc;
}

int post(Comm* comm, Op* op) {
    t_group.emplace_back(comm, op);
    if (t_group_depth > 0) return kSuccess;
    return complete_group(t_group);
}

}  // namespace

extern "C" {

typedef struct {
    char internal[128];
} ncclUniqueId;

__attribute__((visibility("default"))) int ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return kInvalidArgument;
    std::lock_guard<std::mutex> lock(g_worlds_m);
    memset(id->internal, 0, sizeof(id->internal));
    snprintf(id->internal, sizeof(id->internal), "mock-rccl-world-%llu", (unsigned long long)g_next_id++);
    return kSuccess;
}

__attribute__((visibility("default"))) int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return kInvalidArgument;
    std::shared_ptr<World> w;
    {
        std::lock_guard<std::mutex> lock(g_worlds_m);
        auto& slot = g_worlds[std::string(id.internal, strnlen(id.internal, sizeof(id.internal)))];
        if (!slot) slot = std::make_shared<World>();
        w = slot;
    }
    std::unique_lock<std::mutex> lock(w->m);
    if (w->nranks == 0) w->nranks = nranks;
    if (w->nranks != nranks) return kInvalidArgument;
    w->joined += 1;
    w->cv.notify_all();
    if (!w->cv.wait_for(lock, kTimeout, [&] { return w->joined >= w->nranks; })) {
        fprintf(stderr, "[mock rccl] ncclCommInitRank: %d of %d ranks arrived\n", w->joined, w->nranks);
        return kInternalError;
    }
    auto* c = new Comm;
    c->world = w;
    c->rank = rank;
    *comm = c;
    return kSuccess;
}

__attribute__((visibility("default"))) int ncclCommDestroy(void* comm) {
    delete static_cast<Comm*>(comm);
    return kSuccess;
}

__attribute__((visibility("default"))) int ncclCommCount(const void* comm, int* count) {
    if (!comm || !count) return kInvalidArgument;
    *count = static_cast<const Comm*>(comm)->world->nranks;
    return kSuccess;
}

__attribute__((visibility("default"))) int ncclCommUserRank(const void* comm, int* rank) {
    if (!comm || !rank) return kInvalidArgument;
    *rank = static_cast<const Comm*>(comm)->rank;
    return kSuccess;
}

__attribute__((visibility("default"))) int ncclGroupStart() {
    t_group_depth += 1;
    return kSuccess;
}

__attribute__((visibility("default"))) int ncclGroupEnd() {
    if (t_group_depth <= 0) return kInvalidArgument;
    t_group_depth -= 1;
    if (t_group_depth > 0) return kSuccess;
    return complete_group(t_group);
}

__attribute__((visibility("default"))) int ncclSend(const void* buf, size_t count, int datatype, int peer, void* comm, hipStream_t stream) {
    auto* c = static_cast<Comm*>(comm);
    if (!c || peer < 0 || peer >= c->world->nranks || type_bytes(datatype) == 0) return kInvalidArgument;
    auto* op = new Op;
    op->is_send = true;
    op->buf = const_cast<void*>(buf);
    op->bytes = count * type_bytes(datatype);
    op->src = c->rank;
    op->dst = peer;
    op->stream = stream;
    return post(c, op);
}

__attribute__((visibility("default"))) int ncclRecv(void* buf, size_t count, int datatype, int peer, void* comm, hipStream_t stream) {
    auto* c = static_cast<Comm*>(comm);
    if (!c || peer < 0 || peer >= c->world->nranks || type_bytes(datatype) == 0) return kInvalidArgument;
    auto* op = new Op;
    op->buf = buf;
    op->bytes = count * type_bytes(datatype);
    op->src = peer;
    op->dst = c->rank;
    op->stream = stream;
    return post(c, op);
}

// int32 sum only (what sdfv_slab_march's merge uses), on the host: every rank synchronises its stream, adds its words, waits for
// the others, copies the sum back.  In place or not.
__attribute__((visibility("default"))) int ncclAllReduce(const void* sendbuf, void* recvbuf, size_t count, int datatype, int op, void* comm,
                                                          hipStream_t stream) {
    auto* c = static_cast<Comm*>(comm);
    if (!c || op != 0 || (datatype != 2 && datatype != 3)) return kInvalidArgument;
    std::vector<int32_t> mine(count);
    if (!hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize") ||
        !hip_ok(hipMemcpy(mine.data(), sendbuf, count * 4, hipMemcpyDeviceToHost), "hipMemcpy"))
        return kInternalError;
    World& w = *c->world;
    std::unique_lock<std::mutex> lock(w.m);
    // wait for the previous generation to have been picked up by everybody
    if (!w.cv.wait_for(lock, kTimeout, [&] { return w.ar_arrived < w.nranks; })) return kInternalError;  // (the slot is filling, not emptying)
    if (w.ar_arrived == 0) w.ar_sum.assign(count, 0);
    if (w.ar_sum.size() != count) return kInvalidArgument;
    for (size_t i = 0; i < count; ++i) w.ar_sum[i] = (int32_t)((uint32_t)w.ar_sum[i] + (uint32_t)mine[i]);
    w.ar_arrived += 1;
    const uint64_t generation = w.ar_generation;
    w.cv.notify_all();
    if (!w.cv.wait_for(lock, kTimeout, [&] { return w.ar_arrived >= w.nranks || w.ar_generation != generation; })) {
        fprintf(stderr, "[mock rccl] ncclAllReduce: %d of %d ranks arrived\n", w.ar_arrived, w.nranks);
        return kInternalError;
    }
    std::vector<int32_t> sum = w.ar_sum;
    w.ar_departed += 1;
    if (w.ar_departed == w.nranks) {  // the last one out resets the slot for the next all-reduce
        w.ar_arrived = 0;
        w.ar_departed = 0;
        w.ar_generation += 1;
    }
    w.cv.notify_all();
    lock.unlock();
    return hip_ok(hipMemcpy(recvbuf, sum.data(), count * 4, hipMemcpyHostToDevice), "hipMemcpy") ? kSuccess : kInternalError;
}

__attribute__((visibility("default"))) const char* ncclGetErrorString(int code) {
    switch (code) {
        case kSuccess: return "no error (mock RCCL)";
        case kInternalError: return "internal error (mock RCCL: an operation was never matched, or a HIP call failed -- see stderr)";
        case kInvalidArgument: return "invalid argument (mock RCCL)";
        default: return "unknown error (mock RCCL)";
    }
}

}  // extern "C"
