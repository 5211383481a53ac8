// comm.cpp -- the multi-GPU side of include/rtmi.h: a communicator over RCCL (one rank per GPU, xGMI between them)
// and the gather of per-rank film accumulators into rank `root`'s frame.
//
// The reference is a single process on shared memory (crates/rt-cli/src/executor.rs:307-330): its tiles write into
// one Vec of RaySeries.  Here each rank renders its own tiles into its own film, and the only exchange step of a
// frame is this gather: tiles are disjoint, so nothing is reduced, and peers send straight to the root (xGMI is point
// to point: 7 peers on 7 links, no ring).
//
// RCCL is opened at run time, by soname, when the first communicator is made: a single-GPU host needs no RCCL, and
// a process that already carries a copy (PyTorch does) shares it instead of loading a second one.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <algorithm>

#include "rtmi_internal.h"

using rti::fail;

namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    std::string error;
};

Rccl &rccl_state() {
    static Rccl r;
    return r;
}

Rccl *rccl() {
    Rccl &r = rccl_state();
    static std::once_flag once;
    std::call_once(once, [&r] {
        // The RCCL that belongs to the HIP runtime this process runs on: the copy in the directory libamdhip64 was
        // loaded from (a ROCm installation and a PyTorch wheel each ship a matched pair), opened by path so that a
        // copy from another installation that happens to be loaded already is not picked up by its soname.
        std::vector<std::string> names;
        Dl_info where;
        if (dladdr((void *)&hipGetDeviceCount, &where) && where.dli_fname) {
            std::string dir(where.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash + 1);
                names.push_back(dir + "librccl.so.1");
                names.push_back(dir + "librccl.so");
            }
        }
        names.push_back("librccl.so.1");
        names.push_back("librccl.so");
        for (const std::string &n : names) {
            r.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            r.error = std::string("RCCL not found (dlopen librccl.so.1): ") + (dlerror() ? dlerror() : "");
            return;
        }
        auto sym = [&](const char *name) -> void * {
            void *p = dlsym(r.lib, name);
            if (!p && r.error.empty()) r.error = std::string("RCCL lacks ") + name;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    });
    return r.error.empty() ? &r : nullptr;
}

std::string rccl_error() {
    (void)rccl();
    return rccl_state().error.empty() ? std::string("RCCL is not usable in this process") : rccl_state().error;
}

#define NCCL_TRY(expr)                                                                                  \
    do {                                                                                                \
        ncclResult_t r_ = (expr);                                                                       \
        if (r_ != ncclSuccess)                                                                          \
            return fail(RTMI_ERR_DEVICE, std::string(#expr) + ": " + rccl()->GetErrorString(r_));       \
    } while (0)

} // namespace

struct rtmi_comm {
    rtmi_device *dev = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

// What a (source film, communicator, root, frame) combination keeps between gathers: on the root, every rank's pixel
// count, frame positions and a staging buffer for its accumulator planes.
struct rtmi_gather_plan {
    const rtmi_comm *comm = nullptr;
    const rtmi_film *frame = nullptr; // root only
    int root = 0;
    uint32_t fx = 0, fy = 0, fw = 0, fh = 0;
    std::vector<size_t> counts;       // root only: pixels of every rank
    std::vector<uint32_t *> d_lin;    // root only: frame position of every pixel of every rank
    std::vector<float *> d_stage;     // root only: RTMI_ACCUM_PLANES planes of counts[r] floats
    void release() {
        for (uint32_t *p : d_lin)
            if (p) (void)hipFree(p);
        for (float *p : d_stage)
            if (p) (void)hipFree(p);
        d_lin.clear();
        d_stage.clear();
        counts.clear();
    }
};

void rtmi_gather_plan_free(rtmi_film *film) {
    if (!film->plan) return;
    film->plan->release();
    delete film->plan;
    film->plan = nullptr;
}

namespace {

// Builds film->plan for (comm, root, frame).  Collective when comm is set: every rank calls it at the same point.
int build_plan(rtmi_film *film, rtmi_comm *comm, int root, rtmi_film *frame) {
    rtmi_device *dev = film->dev;
    hipStream_t stream = dev->stream;
    rtmi_gather_plan_free(film);
    rtmi_gather_plan *plan = new rtmi_gather_plan();
    film->plan = plan;
    plan->comm = comm;
    plan->root = root;
    const int world = comm ? comm->world : 1, rank = comm ? comm->rank : 0;
    const bool is_root = rank == root;
    if (is_root) {
        plan->frame = frame;
        plan->fx = frame->frame_x;
        plan->fy = frame->frame_y;
        plan->fw = frame->frame_w;
        plan->fh = frame->frame_h;
    }
    // 1. every rank's pixel count, to every rank
    std::vector<unsigned long long> counts((size_t)world, 0ull);
    counts[(size_t)rank] = film->n_pixels;
    unsigned long long *d_counts = nullptr;
    if (comm) {
        HIP_TRY(hipMalloc((void **)&d_counts, sizeof(unsigned long long) * (size_t)(world + 1)));
        const unsigned long long mine = film->n_pixels;
        hipError_t e = hipMemcpyAsync(d_counts + world, &mine, sizeof mine, hipMemcpyHostToDevice, stream);
        ncclResult_t r = ncclSuccess;
        if (e == hipSuccess) r = rccl()->AllGather(d_counts + world, d_counts, 1, ncclUint64, comm->comm, stream);
        if (e == hipSuccess && r == ncclSuccess)
            e = hipMemcpyAsync(counts.data(), d_counts, sizeof(unsigned long long) * (size_t)world,
                               hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(stream);
        (void)hipFree(d_counts);
        if (r != ncclSuccess) return fail(RTMI_ERR_DEVICE, std::string("ncclAllGather: ") + rccl()->GetErrorString(r));
        if (e != hipSuccess) return fail(RTMI_ERR_DEVICE, std::string("gather plan: ") + hipGetErrorString(e));
    }
    // 2. every rank's pixel list to the root, which turns it into frame positions
    std::vector<uint32_t *> d_xy((size_t)world, nullptr);
    uint32_t *d_bad = nullptr;
    int rc = RTMI_OK;
    auto cleanup = [&] {
        for (int r = 0; r < world; ++r)
            if (d_xy[(size_t)r] && !(r == rank && !comm)) (void)hipFree(d_xy[(size_t)r]);
        if (d_bad) (void)hipFree(d_bad);
    };
    if (is_root) {
        plan->counts.assign(counts.begin(), counts.end());
        plan->d_lin.assign((size_t)world, nullptr);
        plan->d_stage.assign((size_t)world, nullptr);
        HIP_TRY(hipMalloc((void **)&d_bad, sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(d_bad, 0, sizeof(uint32_t), stream));
        for (int r = 0; r < world && rc == RTMI_OK; ++r) {
            const size_t n = plan->counts[(size_t)r];
            if (n == 0) continue;
            hipError_t e = hipMalloc((void **)&plan->d_lin[(size_t)r], n * sizeof(uint32_t));
            if (e == hipSuccess && comm) e = hipMalloc((void **)&d_xy[(size_t)r], n * 2 * sizeof(uint32_t));
            if (e == hipSuccess && comm)
                e = hipMalloc((void **)&plan->d_stage[(size_t)r], n * RTMI_ACCUM_PLANES * sizeof(float));
            if (e != hipSuccess) rc = fail(RTMI_ERR_DEVICE, std::string("gather plan: ") + hipGetErrorString(e));
        }
        if (!comm) d_xy[0] = film->d_xy;
    }
    if (comm) {
        // the exchange runs even when the root failed to allocate, so that no peer is left waiting in a send; the
        // root then receives nothing useful and reports its error afterwards
        ncclResult_t r = rccl()->GroupStart();
        if (r == ncclSuccess && film->n_pixels)
            r = rccl()->Send(film->d_xy, film->n_pixels * 2, ncclUint32, root, comm->comm, stream);
        if (is_root && rc == RTMI_OK)
            for (int p = 0; p < world && r == ncclSuccess; ++p)
                if (plan->counts[(size_t)p])
                    r = rccl()->Recv(d_xy[(size_t)p], plan->counts[(size_t)p] * 2, ncclUint32, p, comm->comm, stream);
        ncclResult_t r2 = rccl()->GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess && rc == RTMI_OK)
            rc = fail(RTMI_ERR_DEVICE, std::string("gather plan exchange: ") + rccl()->GetErrorString(r));
    }
    if (is_root && rc == RTMI_OK) {
        for (int r = 0; r < world && rc == RTMI_OK; ++r)
            if (rtk::launch_frame_index(d_xy[(size_t)r], (uint32_t)plan->counts[(size_t)r], plan->fx, plan->fy, plan->fw,
                                        plan->fh, plan->d_lin[(size_t)r], d_bad, stream))
                rc = fail(RTMI_ERR_DEVICE, "frame index kernel launch failed");
        uint32_t bad = 0;
        if (rc == RTMI_OK) {
            hipError_t e = hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) rc = fail(RTMI_ERR_DEVICE, std::string("gather plan: ") + hipGetErrorString(e));
            else if (bad) rc = fail(RTMI_ERR_ARG, "rtmi_film_gather: a film holds pixels outside the frame");
        }
    }
    hipError_t e = hipStreamSynchronize(stream);
    cleanup();
    if (rc == RTMI_OK && e != hipSuccess) rc = fail(RTMI_ERR_DEVICE, std::string("gather plan: ") + hipGetErrorString(e));
    if (rc != RTMI_OK) rtmi_gather_plan_free(film);
    return rc;
}

} // namespace

extern "C" {

int rtmi_comm_unique_id(unsigned char id[RTMI_COMM_ID_BYTES]) {
    if (!id) return fail(RTMI_ERR_ARG, "rtmi_comm_unique_id: NULL argument");
    if (!rccl()) return fail(RTMI_ERR_DEVICE, rccl_error());
    static_assert(sizeof(ncclUniqueId) == RTMI_COMM_ID_BYTES, "id size");
    ncclUniqueId u;
    NCCL_TRY(rccl()->GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return RTMI_OK;
}

int rtmi_comm_create(rtmi_device *dev, const unsigned char id[RTMI_COMM_ID_BYTES], int rank, int world,
                     rtmi_comm **out) {
    if (!dev || !id || !out) return fail(RTMI_ERR_ARG, "rtmi_comm_create: NULL argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(RTMI_ERR_ARG, "rtmi_comm_create: rank must be in [0, world)");
    if (!rccl()) return fail(RTMI_ERR_DEVICE, rccl_error());
    USE_DEVICE(dev);
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t c = nullptr;
    NCCL_TRY(rccl()->CommInitRank(&c, world, u, rank));
    rtmi_comm *comm = new rtmi_comm();
    comm->dev = dev;
    comm->comm = c;
    comm->rank = rank;
    comm->world = world;
    dev->refs.fetch_add(1);
    *out = comm;
    return RTMI_OK;
}

void rtmi_comm_destroy(rtmi_comm *comm) {
    if (!comm) return;
    rtmi_device *dev = comm->dev;
    {
        rti::DeviceScope scope(dev->ordinal);
        (void)hipStreamSynchronize(dev->stream);
        if (rccl() && comm->comm) (void)rccl()->CommDestroy(comm->comm);
    }
    delete comm;
    rtmi_device_release(dev);
}

int rtmi_film_gather(rtmi_film *film, rtmi_comm *comm, int root, rtmi_film *frame) {
    if (!film) return fail(RTMI_ERR_ARG, "rtmi_film_gather: film is NULL");
    const int world = comm ? comm->world : 1, rank = comm ? comm->rank : 0;
    if (root < 0 || root >= world) return fail(RTMI_ERR_ARG, "rtmi_film_gather: root is not a rank");
    if (comm && comm->dev != film->dev) return fail(RTMI_ERR_ARG, "rtmi_film_gather: film and communicator on different devices");
    const bool is_root = rank == root;
    // argument errors on the root must not leave the peers waiting: they are reported before anything is sent only
    // when they cannot differ between ranks; a bad frame on the root is the caller's to avoid
    if (is_root) {
        if (!frame) return fail(RTMI_ERR_ARG, "rtmi_film_gather: the root needs a frame");
        if (!frame->is_frame) return fail(RTMI_ERR_ARG, "rtmi_film_gather: frame was not made by rtmi_film_create_frame");
        if (frame->dev != film->dev) return fail(RTMI_ERR_ARG, "rtmi_film_gather: film and frame on different devices");
        if (frame == film) return fail(RTMI_ERR_ARG, "rtmi_film_gather: film and frame are the same object");
    }
    rtmi_device *dev = film->dev;
    std::lock_guard<std::mutex> g(dev->lock);
    USE_DEVICE(dev);
    hipStream_t stream = dev->stream;
    rtmi_gather_plan *plan = film->plan;
    const bool stale = !plan || plan->comm != comm || plan->root != root ||
                       (is_root && (plan->frame != frame || plan->fx != frame->frame_x || plan->fy != frame->frame_y ||
                                    plan->fw != frame->frame_w || plan->fh != frame->frame_h));
    if (stale) {
        if (int rc = build_plan(film, comm, root, frame)) return rc;
        plan = film->plan;
    }
    if (!comm) {
        if (rtk::launch_scatter_planes(film->d_accum, (uint32_t)film->n_pixels, plan->d_lin[0], frame->d_accum,
                                       (uint32_t)frame->n_pixels, stream))
            return fail(RTMI_ERR_DEVICE, "scatter kernel launch failed");
        HIP_TRY(hipStreamSynchronize(stream));
        return RTMI_OK;
    }
    // every rank, the root included, sends its planes to the root; the root receives one message per rank
    ncclResult_t r = rccl()->GroupStart();
    if (r == ncclSuccess && film->n_pixels)
        r = rccl()->Send(film->d_accum, film->n_pixels * RTMI_ACCUM_PLANES, ncclFloat32, root, comm->comm, stream);
    if (is_root)
        for (int p = 0; p < world && r == ncclSuccess; ++p)
            if (plan->counts[(size_t)p])
                r = rccl()->Recv(plan->d_stage[(size_t)p], plan->counts[(size_t)p] * RTMI_ACCUM_PLANES, ncclFloat32, p,
                                 comm->comm, stream);
    ncclResult_t r2 = rccl()->GroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) return fail(RTMI_ERR_DEVICE, std::string("rtmi_film_gather: ") + rccl()->GetErrorString(r));
    if (is_root)
        for (int p = 0; p < world; ++p)
            if (rtk::launch_scatter_planes(plan->d_stage[(size_t)p], (uint32_t)plan->counts[(size_t)p],
                                           plan->d_lin[(size_t)p], frame->d_accum, (uint32_t)frame->n_pixels, stream))
                return fail(RTMI_ERR_DEVICE, "scatter kernel launch failed");
    HIP_TRY(hipStreamSynchronize(stream));
    return RTMI_OK;
}

} // extern "C"
