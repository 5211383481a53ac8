// comm.cpp -- the multi-GPU side of include/rtmi.h: a communicator over RCCL (one rank per GPU, xGMI between them)
// and the gather of per-rank film accumulators into rank `root`'s frame.
//
// The reference is a single process on shared memory (crates/rt-cli/src/executor.rs:307-330): its tiles write into
// one Vec of RaySeries.  Here each rank renders its own tiles into its own film, and the only exchange step of a
// frame is this gather: tiles are disjoint, so nothing is reduced, and peers send straight to the root (xGMI is point
// to point: 7 peers on 7 links, no ring).  A gather starts with a 32-byte all-gather of every rank's status, so an
// argument error or a failed allocation on one rank makes every rank return an error instead of leaving the others
// waiting in a send, and so that all ranks agree on when the pixel lists have to be exchanged again: the status
// carries the identity and size of the film every rank brings, and the lists travel again whenever any rank's plan was
// made for a different set of films than the one present.
//
// RCCL is opened at run time, by soname, when the first communicator is made: a single-GPU host needs no RCCL, and
// a process that already carries a copy (PyTorch does) shares it instead of loading a second one.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdlib.h>
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
        // RTMI_RCCL_LIB names the library to use instead (an RCCL in an unusual place, or a stand-in with the same
        // nine entry points)
        if (const char *forced = getenv("RTMI_RCCL_LIB")) {
            r.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            if (!r.lib) {
                r.error = std::string("RTMI_RCCL_LIB=") + forced + " could not be opened: " + (dlerror() ? dlerror() : "");
                return;
            }
        }
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
            if (r.lib) break;
            r.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
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
    unsigned long long *d_status = nullptr; // STATUS_WORDS * (world + 1) words for the status rounds of rtmi_film_gather
};

// Words every rank contributes to a status round: pixels in its film, flags, the film's identity, and the epoch of the
// plan it holds for this (film, communicator, root) combination (0 = none).
constexpr size_t STATUS_WORDS = 4;

// What a (film, communicator, root) combination keeps between gathers.  On the root: every rank's pixel count and
// pixel list (received once), a staging buffer for its accumulator planes, and for the frame last gathered into the
// position of every pixel in it.
struct rtmi_gather_plan {
    const rtmi_comm *comm = nullptr;
    int root = 0;
    unsigned long long epoch = 0;    // hash of (root, every rank's film identity and pixel count) the plan was made for
    std::vector<size_t> counts;      // root only: pixels of every rank
    std::vector<uint32_t *> d_xy;    // root only: pixel list of every rank; the root's own entry is its film's list
    bool owns_xy = false;            // the entries of d_xy other than `root` were allocated by the plan
    std::vector<float *> d_stage;    // root only, comm only: RTMI_ACCUM_PLANES planes of counts[r] floats for every peer
    // frame-specific, root only
    const rtmi_film *frame = nullptr;
    uint32_t fx = 0, fy = 0, fw = 0, fh = 0;
    std::vector<uint32_t *> d_lin;   // frame position of every pixel of every rank
    void release() {
        for (uint32_t *p : d_lin)
            if (p) (void)hipFree(p);
        for (float *p : d_stage)
            if (p) (void)hipFree(p);
        if (owns_xy)
            for (size_t r = 0; r < d_xy.size(); ++r)
                if (d_xy[r] && (int)r != root) (void)hipFree(d_xy[r]);
        d_lin.clear();
        d_stage.clear();
        d_xy.clear();
        counts.clear();
        frame = nullptr;
    }
};

void rtmi_gather_plan_free(rtmi_film *film) {
    if (!film->plan) return;
    film->plan->release();
    delete film->plan;
    film->plan = nullptr;
}

namespace {

constexpr unsigned long long FLAG_ERROR = 1, FLAG_REINDEX = 2;

// One status round: every rank contributes STATUS_WORDS words and learns everybody's.  This is what keeps the ranks of
// a gather in step: an argument error or a failed allocation on one rank is seen by all of them before any of them
// commits to a send or a receive that the failing rank would not match.
int status_round(rtmi_comm *comm, hipStream_t stream, unsigned long long n_pixels, unsigned long long flags,
                 unsigned long long film_id, unsigned long long epoch, std::vector<unsigned long long> &all) {
    const size_t world = (size_t)comm->world;
    unsigned long long mine[STATUS_WORDS] = {n_pixels, flags, film_id, epoch};
    all.assign(STATUS_WORDS * world, 0ull);
    HIP_TRY(hipMemcpyAsync(comm->d_status + STATUS_WORDS * world, mine, sizeof mine, hipMemcpyHostToDevice, stream));
    NCCL_TRY(rccl()->AllGather(comm->d_status + STATUS_WORDS * world, comm->d_status, STATUS_WORDS, ncclUint64, comm->comm,
                               stream));
    HIP_TRY(hipMemcpyAsync(all.data(), comm->d_status, sizeof(unsigned long long) * STATUS_WORDS * world,
                           hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return RTMI_OK;
}

// First rank whose flags word has `bit` set, or -1.
int first_with(const std::vector<unsigned long long> &all, unsigned long long bit) {
    for (size_t r = 0; STATUS_WORDS * r + 1 < all.size(); ++r)
        if (all[STATUS_WORDS * r + 1] & bit) return (int)r;
    return -1;
}

// What a plan is valid for: the root and, rank by rank, which film (identity and pixel count) takes part.  Every rank
// computes it from the same status words, so all of them reach the same verdict on whether their plans still hold.
unsigned long long epoch_of(const std::vector<unsigned long long> &all, int root) {
    auto mix = [](unsigned long long h, unsigned long long v) {
        h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        h *= 0xFF51AFD7ED558CCDull;
        return h ^ (h >> 33);
    };
    unsigned long long h = mix(0x6a09e667f3bcc908ull, (unsigned long long)root);
    for (size_t r = 0; STATUS_WORDS * r + 2 < all.size(); ++r) {
        h = mix(h, all[STATUS_WORDS * r]);
        h = mix(h, all[STATUS_WORDS * r + 2]);
    }
    return h | 1ull; // never 0, which stands for "no plan"
}

// Root only: positions of every rank's pixels in `frame`; fails when a pixel lies outside it.  Local (no
// communication): the pixel lists are already here.
int index_frame(rtmi_gather_plan *plan, const rtmi_film *frame, hipStream_t stream) {
    const size_t world = plan->counts.size();
    for (uint32_t *p : plan->d_lin)
        if (p) (void)hipFree(p);
    plan->d_lin.assign(world, nullptr);
    plan->frame = nullptr;
    uint32_t *d_bad = nullptr;
    HIP_TRY(hipMalloc((void **)&d_bad, sizeof(uint32_t)));
    int rc = RTMI_OK;
    hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(uint32_t), stream);
    for (size_t r = 0; r < world && e == hipSuccess && rc == RTMI_OK; ++r) {
        const size_t n = plan->counts[r];
        if (n == 0) continue;
        e = hipMalloc((void **)&plan->d_lin[r], n * sizeof(uint32_t));
        if (e == hipSuccess && rtk::launch_frame_index(plan->d_xy[r], (uint32_t)n, frame->frame_x, frame->frame_y,
                                                       frame->frame_w, frame->frame_h, plan->d_lin[r], d_bad, stream))
            rc = fail(RTMI_ERR_DEVICE, "frame index kernel launch failed");
    }
    uint32_t bad = 0;
    if (e == hipSuccess && rc == RTMI_OK) e = hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, stream);
    hipError_t e2 = hipStreamSynchronize(stream);
    (void)hipFree(d_bad);
    if (rc == RTMI_OK && (e != hipSuccess || e2 != hipSuccess))
        rc = fail(RTMI_ERR_DEVICE, std::string("frame index: ") + hipGetErrorString(e != hipSuccess ? e : e2));
    if (rc == RTMI_OK && bad) rc = fail(RTMI_ERR_ARG, "rtmi_film_gather: a film holds pixels outside the frame");
    if (rc != RTMI_OK) return rc;
    plan->frame = frame;
    plan->fx = frame->frame_x;
    plan->fy = frame->frame_y;
    plan->fw = frame->frame_w;
    plan->fh = frame->frame_h;
    return RTMI_OK;
}

bool frame_indexed(const rtmi_gather_plan *plan, const rtmi_film *frame) {
    return plan->frame == frame && plan->fx == frame->frame_x && plan->fy == frame->frame_y &&
           plan->fw == frame->frame_w && plan->fh == frame->frame_h;
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
    unsigned long long *d_status = nullptr;
    HIP_TRY(hipMalloc((void **)&d_status, sizeof(unsigned long long) * STATUS_WORDS * (size_t)(world + 1)));
    ncclResult_t nr = rccl()->CommInitRank(&c, world, u, rank);
    if (nr != ncclSuccess) {
        (void)hipFree(d_status);
        return fail(RTMI_ERR_DEVICE, std::string("ncclCommInitRank: ") + rccl()->GetErrorString(nr));
    }
    rtmi_comm *comm = new rtmi_comm();
    comm->dev = dev;
    comm->comm = c;
    comm->rank = rank;
    comm->world = world;
    comm->d_status = d_status;
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
        if (comm->d_status) (void)hipFree(comm->d_status);
    }
    delete comm;
    rtmi_device_release(dev);
}

int rtmi_film_gather(rtmi_film *film, rtmi_comm *comm, int root, rtmi_film *frame) {
    if (!film) return fail(RTMI_ERR_ARG, "rtmi_film_gather: film is NULL");
    const int world = comm ? comm->world : 1, rank = comm ? comm->rank : 0;
    if (root < 0 || root >= world) return fail(RTMI_ERR_ARG, "rtmi_film_gather: root is not a rank");
    if (comm && comm->dev != film->dev)
        return fail(RTMI_ERR_ARG, "rtmi_film_gather: film and communicator on different devices");
    const bool is_root = rank == root;
    // What only this rank can see wrong with its arguments.  With a communicator it is not returned at once but told to
    // the other ranks in the status round below, so that all of them leave the call together.
    std::string local_error;
    if (is_root) {
        if (!frame) local_error = "the root needs a frame";
        else if (!frame->is_frame) local_error = "frame was not made by rtmi_film_create_frame";
        else if (frame->dev != film->dev) local_error = "film and frame on different devices";
        else if (frame == film) local_error = "film and frame are the same object";
    }
    if (!comm && !local_error.empty()) return fail(RTMI_ERR_ARG, "rtmi_film_gather: " + local_error);
    rtmi_device *dev = film->dev;
    std::lock_guard<std::mutex> g(dev->lock);
    USE_DEVICE(dev);
    hipStream_t stream = dev->stream;
    rtmi_gather_plan *plan = film->plan;
    const bool have_plan = plan && plan->comm == comm && plan->root == root;

    if (!comm) {
        // one rank, film and frame on one device: the film's own pixel list, no staging
        if (!have_plan) {
            rtmi_gather_plan_free(film);
            plan = film->plan = new rtmi_gather_plan();
            plan->root = root;
            plan->counts.assign(1, film->n_pixels);
            plan->d_xy.assign(1, film->d_xy);
        }
        if (!frame_indexed(plan, frame))
            if (int rc = index_frame(plan, frame, stream)) return rc;
        if (rtk::launch_scatter_planes(film->d_accum, (uint32_t)film->n_pixels, plan->d_lin[0], frame->d_accum,
                                       (uint32_t)frame->n_pixels, stream))
            return fail(RTMI_ERR_DEVICE, "scatter kernel launch failed");
        HIP_TRY(hipStreamSynchronize(stream));
        return RTMI_OK;
    }

    // ---- round 1: who is here with which film, who cannot take part, and whether the root has to index its frame
    int local_code = local_error.empty() ? RTMI_OK : RTMI_ERR_ARG;
    const bool reindex = is_root && local_code == RTMI_OK && !(have_plan && frame_indexed(plan, frame));
    std::vector<unsigned long long> all;
    if (int rc = status_round(comm, stream, film->n_pixels,
                              (local_code != RTMI_OK ? FLAG_ERROR : 0) | (reindex ? FLAG_REINDEX : 0), film->id,
                              have_plan ? plan->epoch : 0ull, all))
        return rc;
    auto report = [&](int bad_rank) {
        if (bad_rank == rank) {
            const std::string prefix = "rtmi_film_gather: "; // index_frame's messages carry it already
            return fail(local_code != RTMI_OK ? local_code : RTMI_ERR_DEVICE,
                        local_error.compare(0, prefix.size(), prefix) == 0 ? local_error : prefix + local_error);
        }
        return fail(RTMI_ERR_STATE, "rtmi_film_gather: rank " + std::to_string(bad_rank) + " could not take part; nothing was exchanged");
    };
    // Agrees on the outcome of a step only this rank can see the result of (an allocation, the root's indexing).
    auto agree = [&]() -> int {
        std::vector<unsigned long long> ok;
        int rc = status_round(comm, stream, 0, local_code != RTMI_OK ? FLAG_ERROR : 0, 0, 0, ok);
        if (rc != RTMI_OK) return rc;
        const int bad = first_with(ok, FLAG_ERROR);
        return bad >= 0 ? report(local_code != RTMI_OK ? rank : bad) : RTMI_OK;
    };
    int bad_rank = first_with(all, FLAG_ERROR);
    if (bad_rank >= 0) return report(local_code != RTMI_OK ? rank : bad_rank);

    // every rank's plan must have been made for exactly the films present now; all ranks evaluate the same words
    const unsigned long long epoch = epoch_of(all, root);
    bool exchange = false;
    for (int r = 0; r < world; ++r)
        if (all[STATUS_WORDS * (size_t)r + 3] != epoch) exchange = true;

    if (exchange) {
        // some rank is new to this combination of films: every peer sends the root its pixel list again
        rtmi_gather_plan_free(film);
        plan = film->plan = new rtmi_gather_plan();
        plan->comm = comm;
        plan->root = root;
        if (is_root) {
            plan->owns_xy = true;
            plan->counts.resize((size_t)world);
            plan->d_xy.assign((size_t)world, nullptr);
            plan->d_stage.assign((size_t)world, nullptr);
            hipError_t e = hipSuccess;
            for (int r = 0; r < world && e == hipSuccess; ++r) {
                const size_t n = plan->counts[(size_t)r] = (size_t)all[STATUS_WORDS * (size_t)r];
                if (r == root) {
                    plan->d_xy[(size_t)r] = film->d_xy; // the root's own pixels never leave its film
                    continue;
                }
                if (n == 0) continue;
                e = hipMalloc((void **)&plan->d_xy[(size_t)r], n * 2 * sizeof(uint32_t));
                if (e == hipSuccess) e = hipMalloc((void **)&plan->d_stage[(size_t)r], n * RTMI_ACCUM_PLANES * sizeof(float));
            }
            if (e != hipSuccess) {
                local_code = RTMI_ERR_DEVICE;
                local_error = std::string("staging allocation on the root: ") + hipGetErrorString(e);
            }
        }
        if (int rc = agree()) {
            rtmi_gather_plan_free(film);
            return rc;
        }
        ncclResult_t r = rccl()->GroupStart();
        if (r == ncclSuccess && !is_root && film->n_pixels)
            r = rccl()->Send(film->d_xy, film->n_pixels * 2, ncclUint32, root, comm->comm, stream);
        if (is_root)
            for (int p = 0; p < world && r == ncclSuccess; ++p)
                if (p != root && plan->counts[(size_t)p])
                    r = rccl()->Recv(plan->d_xy[(size_t)p], plan->counts[(size_t)p] * 2, ncclUint32, p, comm->comm, stream);
        ncclResult_t r2 = rccl()->GroupEnd();
        if (r == ncclSuccess) r = r2;
        hipError_t e = hipStreamSynchronize(stream);
        if (r != ncclSuccess || e != hipSuccess) {
            rtmi_gather_plan_free(film);
            return fail(RTMI_ERR_DEVICE, std::string("rtmi_film_gather, pixel lists: ") +
                                             (r != ncclSuccess ? rccl()->GetErrorString(r) : hipGetErrorString(e)));
        }
        plan->epoch = epoch; // the lists are in place; a frame that does not fit them leaves them valid
    }
    if (exchange || first_with(all, FLAG_REINDEX) >= 0) {
        // the root indexes its frame with the lists it holds; whether that worked is agreed on before anything moves
        if (is_root) {
            local_code = index_frame(plan, frame, stream);
            if (local_code != RTMI_OK) local_error = rtmi_last_error();
        }
        if (int rc = agree()) return rc;
    }

    // every peer sends its planes to the root, which receives one message per peer; the root's own planes are
    // scattered straight from its film
    ncclResult_t r = ncclSuccess;
    if (world > 1) {
        r = rccl()->GroupStart();
        if (r == ncclSuccess && !is_root && film->n_pixels)
            r = rccl()->Send(film->d_accum, film->n_pixels * RTMI_ACCUM_PLANES, ncclFloat32, root, comm->comm, stream);
        if (is_root)
            for (int p = 0; p < world && r == ncclSuccess; ++p)
                if (p != root && plan->counts[(size_t)p])
                    r = rccl()->Recv(plan->d_stage[(size_t)p], plan->counts[(size_t)p] * RTMI_ACCUM_PLANES, ncclFloat32, p,
                                     comm->comm, stream);
        ncclResult_t r2 = rccl()->GroupEnd();
        if (r == ncclSuccess) r = r2;
    }
    if (r != ncclSuccess) return fail(RTMI_ERR_DEVICE, std::string("rtmi_film_gather: ") + rccl()->GetErrorString(r));
    if (is_root)
        for (int p = 0; p < world; ++p)
            if (rtk::launch_scatter_planes(p == root ? film->d_accum : plan->d_stage[(size_t)p],
                                           (uint32_t)plan->counts[(size_t)p], plan->d_lin[(size_t)p], frame->d_accum,
                                           (uint32_t)frame->n_pixels, stream))
                return fail(RTMI_ERR_DEVICE, "scatter kernel launch failed");
    HIP_TRY(hipStreamSynchronize(stream));
    return RTMI_OK;
}

int rtmi_comm_gather_abort(rtmi_comm *comm) {
    if (!comm) return fail(RTMI_ERR_ARG, "rtmi_comm_gather_abort: comm is NULL");
    rtmi_device *dev = comm->dev;
    std::lock_guard<std::mutex> g(dev->lock);
    USE_DEVICE(dev);
    std::vector<unsigned long long> all;
    return status_round(comm, dev->stream, 0, FLAG_ERROR, 0, 0, all);
}

} // extern "C"
