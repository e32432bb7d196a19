// TEST INFRASTRUCTURE ONLY — a stand-in for the few RCCL entry points libvrt_hip.so binds with dlopen
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclGroupStart/End, ncclSend, ncclRecv, ncclBroadcast,
// ncclGetErrorString), for running the multi-rank frame pipeline of vrt_dist_* with all "ranks" as contexts of ONE
// process on ONE GPU (tests/test_dist_fake_rccl.py).  Real RCCL refuses two ranks on one device, and the boxes
// have one GPU; without this the send/recv branch of vrt_dist_frame would first run in the driver's 8-GPU bench.
//
// Semantics kept: point-to-point operations between a pair of ranks match in FIFO order; a send is complete in the
// sender's stream order once its data is safe to overwrite; a receive is complete in the receiver's stream order
// once the data has landed.  Mechanism: a send copies its buffer to a staging buffer on the SENDER's stream and
// records an event; a receive blocks the calling HOST thread until the matching send has been posted (ranks are
// driven from different threads), then makes the RECEIVER's stream wait for that event and copies staging -> dst.
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct ncclComm;
typedef struct ncclComm *ncclComm_t;
}

namespace {
struct Posted {
    void *staging;
    size_t bytes;
    hipEvent_t ready;
    bool owned; // staging is this stand-in's buffer (false: zero-copy mode, the sender's own)
};
struct Staging {
    void *ptr;
    size_t bytes;
    hipEvent_t idle; // recorded after the receiver's copy out of it
};
struct World {
    int nranks = 0;
    std::map<std::pair<int, int>, std::deque<Posted>> sends; // (src, dst) -> FIFO
    std::vector<Staging> pool;                               // staging buffers, reused once idle; freed with the last communicator
    int live = 0;
};
std::mutex g_mu;
std::condition_variable g_cv;
std::map<std::string, World *> g_worlds;
uint64_t g_next_id = 1;
// measurement aid (tools/dist_emulate.py): no staging copy, the receiver copies straight out of the send buffer — only sound when the
// sender never rewrites that buffer, as the feeder thread of that tool.  FAKE_RCCL_ZERO_COPY sets the initial state.
bool g_zero_copy = std::getenv("FAKE_RCCL_ZERO_COPY") != nullptr;
}

struct ncclComm {
    World *world;
    int rank;
    std::string key;
};

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    std::lock_guard<std::mutex> lk(g_mu);
    std::memset(id, 0, sizeof *id);
    const uint64_t v = g_next_id++;
    std::memcpy(id->internal, "fake-rccl", 9);
    std::memcpy(id->internal + 16, &v, sizeof v);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_mu);
    const std::string key(id.internal, sizeof id.internal);
    World *&w = g_worlds[key];
    if (!w) {
        w = new World;
        w->nranks = nranks;
    }
    if (w->nranks != nranks) return ncclInvalidArgument;
    w->live++;
    *comm = new ncclComm{w, rank, key};
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (!comm) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_mu);
    World *w = comm->world;
    if (--w->live == 0) {
        (void)hipDeviceSynchronize();
        for (Staging &g : w->pool) {
            (void)hipFree(g.ptr);
            (void)hipEventDestroy(g.idle);
        }
        for (auto &q : w->sends)
            for (Posted &s : q.second) {
                if (s.owned) (void)hipFree(s.staging); // (a zero-copy send that was never received: the buffer is the sender's)
                (void)hipEventDestroy(s.ready);
            }
        g_worlds.erase(comm->key);
        delete w;
    }
    delete comm;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) {
    if (!comm || !count) return ncclInvalidArgument;
    *count = comm->world->nranks;
    return ncclSuccess;
}
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank) {
    if (!comm || !rank) return ncclInvalidArgument;
    *rank = comm->rank;
    return ncclSuccess;
}

void fake_rccl_set_zero_copy(int on) { // (between communicators only)
    std::lock_guard<std::mutex> lk(g_mu);
    g_zero_copy = on != 0;
}

ncclResult_t ncclGroupStart() { return ncclSuccess; } // operations are carried out as they are posted
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

// (implementations are file-local: a call from ncclBroadcast below must not go through the PLT, where it would bind to the
// real RCCL's ncclSend / ncclRecv if PyTorch's librccl is already in the process)
static ncclResult_t send_impl(const void *sendbuff, size_t count, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!comm || peer < 0 || peer >= comm->world->nranks) return ncclInvalidArgument;
    Posted s{nullptr, count, nullptr, true};
    if (g_zero_copy) {
        s.staging = const_cast<void *>(sendbuff);
        s.owned = false;
        if (hipEventCreateWithFlags(&s.ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(s.ready, stream) != hipSuccess)
            return ncclUnhandledCudaError;
        {
            std::lock_guard<std::mutex> lk(g_mu);
            comm->world->sends[{comm->rank, peer}].push_back(s);
        }
        g_cv.notify_all();
        return ncclSuccess;
    }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto &pool = comm->world->pool;
        for (size_t i = 0; i < pool.size(); i++)
            if (pool[i].bytes >= count && hipEventQuery(pool[i].idle) == hipSuccess) {
                s.staging = pool[i].ptr;
                (void)hipEventDestroy(pool[i].idle);
                pool.erase(pool.begin() + (long)i);
                break;
            }
    }
    if (!s.staging && hipMalloc(&s.staging, count ? count : 1) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventCreateWithFlags(&s.ready, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpyAsync(s.staging, sendbuff, count, hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventRecord(s.ready, stream) != hipSuccess) return ncclUnhandledCudaError;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        comm->world->sends[{comm->rank, peer}].push_back(s);
    }
    g_cv.notify_all();
    return ncclSuccess;
}

static ncclResult_t recv_impl(void *recvbuff, size_t count, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!comm || peer < 0 || peer >= comm->world->nranks) return ncclInvalidArgument;
    Posted s;
    {
        std::unique_lock<std::mutex> lk(g_mu);
        auto &q = comm->world->sends[{peer, comm->rank}];
        g_cv.wait(lk, [&] { return !q.empty(); }); // the matching send is posted by another host thread
        s = q.front();
        q.pop_front();
    }
    if (s.bytes != count) return ncclInvalidArgument;
    if (hipStreamWaitEvent(stream, s.ready, 0) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpyAsync(recvbuff, s.staging, count, hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    (void)hipEventDestroy(s.ready); // (destruction is deferred by the runtime until the wait has been carried out)
    if (!s.owned) return ncclSuccess; // (the buffer is the sender's)
    Staging g{s.staging, s.bytes, nullptr};
    if (hipEventCreateWithFlags(&g.idle, hipEventDisableTiming) != hipSuccess || hipEventRecord(g.idle, stream) != hipSuccess) return ncclUnhandledCudaError;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        comm->world->pool.push_back(g);
    }
    return ncclSuccess;
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t, int peer, ncclComm_t comm, hipStream_t stream) {
    return send_impl(sendbuff, count, peer, comm, stream);
}
ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t, int peer, ncclComm_t comm, hipStream_t stream) {
    return recv_impl(recvbuff, count, peer, comm, stream);
}
// in place or out of place: the root's sendbuff reaches every other rank's recvbuff (as point-to-point operations of this stand-in)
ncclResult_t ncclBroadcast(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t, int root, ncclComm_t comm, hipStream_t stream) {
    if (!comm || root < 0 || root >= comm->world->nranks) return ncclInvalidArgument;
    if (comm->rank != root) return recv_impl(recvbuff, count, root, comm, stream);
    for (int peer = 0; peer < comm->world->nranks; peer++) {
        if (peer == root) continue;
        const ncclResult_t r = send_impl(sendbuff, count, peer, comm, stream);
        if (r != ncclSuccess) return r;
    }
    if (recvbuff != sendbuff && hipMemcpyAsync(recvbuff, sendbuff, count, hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "unhandled HIP error (fake rccl)";
        case ncclInvalidArgument: return "invalid argument (fake rccl)";
        default: return "error (fake rccl)";
    }
}
}
