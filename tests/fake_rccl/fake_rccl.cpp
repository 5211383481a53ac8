// fake_rccl.cpp -- TEST INFRASTRUCTURE: an in-process stand-in for the nine RCCL entry points librtmi.so uses, so that
// the multi-rank side of rtmi_film_gather can be executed on a machine with one GPU (RCCL itself refuses two ranks on
// one device).  Ranks are threads of one process, each with its own rtmi_device on the same GPU; the library is
// pointed here with RTMI_RCCL_LIB.  Nothing under raytracing_amd/ refers to this file.
//
// Semantics kept: communicator creation returns when all ranks have joined; sends and receives issued between
// ncclGroupStart and ncclGroupEnd complete together at ncclGroupEnd and match in order per (sender, receiver) pair;
// ncclAllGather is collective.  Everything is synchronous on the host (the streams are drained first), which is
// stricter than RCCL and therefore hides no ordering bug of the caller's.  Every wait has a deadline: a mismatch
// between the ranks shows up as an error return after FAKE_RCCL_TIMEOUT_S (default 20) seconds, never as a hang.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Msg {
    const void *src;
    size_t bytes;
    bool done = false;
    bool failed = false;
};

struct Group {
    int world = 0;
    std::mutex m;
    std::condition_variable cv;
    int joined = 0;
    std::map<std::pair<int, int>, std::deque<std::shared_ptr<Msg>>> box; // (sender, receiver) -> messages in order
    // barrier + all-gather slots
    int arrived = 0;
    unsigned long long generation = 0;
    std::vector<const void *> ag_src;
};

struct Comm {
    Group *g;
    int rank;
};

std::mutex g_groups_lock;
std::map<std::string, Group *> g_groups;
unsigned long long g_next_id = 1;

std::chrono::seconds deadline() {
    const char *v = getenv("FAKE_RCCL_TIMEOUT_S");
    return std::chrono::seconds(v ? atoi(v) : 20);
}

size_t size_of(ncclDataType_t t) {
    switch (t) {
    case ncclInt8:
    case ncclUint8:
        return 1;
    case ncclFloat16:
    case ncclBfloat16:
        return 2;
    case ncclInt32:
    case ncclUint32:
    case ncclFloat32:
        return 4;
    default:
        return 8;
    }
}

// All ranks of the group meet here; false on timeout.
bool barrier(Group *g, std::unique_lock<std::mutex> &lk) {
    const unsigned long long gen = g->generation;
    if (++g->arrived == g->world) {
        g->arrived = 0;
        ++g->generation;
        g->cv.notify_all();
        return true;
    }
    if (!g->cv.wait_for(lk, deadline(), [&] { return g->generation != gen; })) {
        --g->arrived;
        return false;
    }
    return true;
}

struct Op {
    bool send;
    void *ptr;
    size_t bytes;
    int peer;
    Comm *comm;
    hipStream_t stream;
};
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

ncclResult_t run_ops(std::vector<Op> &ops) {
    ncclResult_t result = ncclSuccess;
    for (const Op &op : ops)
        if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<std::pair<Group *, std::shared_ptr<Msg>>> mine;
    for (const Op &op : ops) {
        if (!op.send) continue;
        Group *g = op.comm->g;
        auto msg = std::make_shared<Msg>();
        msg->src = op.ptr;
        msg->bytes = op.bytes;
        {
            std::lock_guard<std::mutex> lk(g->m);
            g->box[{op.comm->rank, op.peer}].push_back(msg);
        }
        g->cv.notify_all();
        mine.push_back({g, msg});
    }
    for (const Op &op : ops) {
        if (op.send) continue;
        Group *g = op.comm->g;
        std::shared_ptr<Msg> msg;
        {
            std::unique_lock<std::mutex> lk(g->m);
            auto &q = g->box[{op.peer, op.comm->rank}];
            if (!g->cv.wait_for(lk, deadline(), [&] { return !q.empty(); })) {
                result = ncclInternalError;
                continue;
            }
            msg = q.front();
            q.pop_front();
        }
        bool ok = msg->bytes == op.bytes;
        if (ok && op.bytes) ok = hipMemcpy(op.ptr, msg->src, op.bytes, hipMemcpyDeviceToDevice) == hipSuccess;
        {
            std::lock_guard<std::mutex> lk(g->m);
            msg->done = true;
            msg->failed = !ok;
        }
        g->cv.notify_all();
        if (!ok) result = ncclInvalidArgument;
    }
    for (auto &s : mine) {
        std::unique_lock<std::mutex> lk(s.first->m);
        if (!s.first->cv.wait_for(lk, deadline(), [&] { return s.second->done; })) {
            // nobody received it: take it back so that a later receive does not pick up a dead buffer
            for (auto &kv : s.first->box)
                for (auto it = kv.second.begin(); it != kv.second.end(); ++it)
                    if (*it == s.second) {
                        kv.second.erase(it);
                        break;
                    }
            result = ncclInternalError;
        } else if (s.second->failed) {
            result = ncclInvalidArgument;
        }
    }
    ops.clear();
    return result;
}

ncclResult_t enqueue(bool send, void *ptr, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || peer < 0 || peer >= c->g->world) return ncclInvalidArgument;
    t_ops.push_back(Op{send, ptr, count * size_of(type), peer, c, stream});
    if (t_depth == 0) return run_ops(t_ops);
    return ncclSuccess;
}

} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    std::lock_guard<std::mutex> lk(g_groups_lock);
    const unsigned long long n = g_next_id++;
    memcpy(id->internal, "fake-rccl", 9);
    memcpy(id->internal + 16, &n, sizeof n);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Group *g;
    {
        std::lock_guard<std::mutex> lk(g_groups_lock);
        Group *&slot = g_groups[std::string(id.internal, sizeof id.internal)];
        if (!slot) {
            slot = new Group();
            slot->world = nranks;
            slot->ag_src.assign((size_t)nranks, nullptr);
        }
        g = slot;
    }
    if (g->world != nranks) return ncclInvalidArgument;
    std::unique_lock<std::mutex> lk(g->m);
    ++g->joined;
    g->cv.notify_all();
    if (!g->cv.wait_for(lk, deadline(), [&] { return g->joined >= g->world; })) return ncclInternalError;
    *comm = reinterpret_cast<ncclComm_t>(new Comm{g, rank});
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete reinterpret_cast<Comm *>(comm); // the group stays: other ranks may still be inside a call
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
    case ncclSuccess:
        return "no error";
    case ncclInvalidArgument:
        return "fake rccl: invalid argument or mismatched message size";
    case ncclInternalError:
        return "fake rccl: a peer did not show up before the deadline";
    case ncclUnhandledCudaError:
        return "fake rccl: HIP error";
    default:
        return "fake rccl: error";
    }
}

ncclResult_t ncclGroupStart(void) {
    ++t_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd(void) {
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    return run_ops(t_ops);
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm,
                      hipStream_t stream) {
    return enqueue(true, const_cast<void *>(sendbuff), count, datatype, peer, comm, stream);
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm,
                      hipStream_t stream) {
    return enqueue(false, recvbuff, count, datatype, peer, comm, stream);
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype,
                           ncclComm_t comm, hipStream_t stream) {
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return ncclInvalidArgument;
    Group *g = c->g;
    const size_t bytes = sendcount * size_of(datatype);
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    std::unique_lock<std::mutex> lk(g->m);
    g->ag_src[(size_t)c->rank] = sendbuff;
    if (!barrier(g, lk)) return ncclInternalError;
    std::vector<const void *> src = g->ag_src;
    lk.unlock();
    ncclResult_t result = ncclSuccess;
    for (int r = 0; r < g->world; ++r)
        if (bytes && hipMemcpy(static_cast<char *>(recvbuff) + (size_t)r * bytes, src[(size_t)r], bytes,
                               hipMemcpyDeviceToDevice) != hipSuccess)
            result = ncclUnhandledCudaError;
    lk.lock();
    if (!barrier(g, lk)) return ncclInternalError; // nobody overwrites its contribution before everyone has copied it
    return result;
}

} // extern "C"
