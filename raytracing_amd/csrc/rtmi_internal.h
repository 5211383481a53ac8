// rtmi_internal.h -- what the translation units behind include/rtmi.h share: the handle structures, the error
// convention and the device scope.  Not installed; callers see only include/rtmi.h.
#pragma once

#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rtmi.h"
#include "kernels.h"
#include "scene_build.h"

namespace rti {

// Records the message rtmi_last_error() returns on the calling thread; returns `code`.
int fail(int code, const std::string &msg);

} // namespace rti

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return rti::fail(RTMI_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)

namespace rti {

// Makes a device current for the duration of one entry point and puts the caller's device back on the way out, so a
// host that shares the thread with other HIP users (PyTorch, its own kernels) keeps its current device.
struct DeviceScope {
    int previous = -1;
    hipError_t status = hipSuccess;
    explicit DeviceScope(int ordinal) {
        if (hipGetDevice(&previous) != hipSuccess) previous = -1;
        if (previous != ordinal) status = hipSetDevice(ordinal);
        else previous = -1; // already current: nothing to restore
    }
    ~DeviceScope() {
        if (previous >= 0) (void)hipSetDevice(previous);
    }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

#define USE_DEVICE(dev)                                                                                  \
    rti::DeviceScope device_scope_((dev)->ordinal);                                                      \
    if (device_scope_.status != hipSuccess)                                                              \
    return rti::fail(RTMI_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(device_scope_.status))

} // namespace rti

struct rtmi_comm;

struct rtmi_device {
    int ordinal = 0;
    hipStream_t stream = nullptr;
    std::mutex lock;
    bool count_visits = false;
    uint32_t *d_visits = nullptr; // two 64-bit counters
    int n_cu = 256;
    uint32_t *static_spill = nullptr; // this device's copy of the statically allocated stack-overflow buffer
    size_t static_spill_words = 0;
    // Survivor counters come back to the host on a stream of their own, so that the copy is not between two bounces on
    // the render stream: ring of pinned slots, "bounce shaded" / "counters copied" event pairs (created on first use).
    static constexpr int COUNT_RING = 4;
    hipStream_t copy_stream = nullptr;
    uint32_t *h_counts = nullptr; // COUNT_RING slots of SEGMENTS * COUNTER_STRIDE words, pinned
    hipEvent_t ev_shaded[COUNT_RING] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_copied[COUNT_RING] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<rtmi_bounce_profile> profile; // of the last render (under `lock`)
    std::atomic<int> refs{1}; // the handle itself + one per scene and film created on it
};

// A device outlives the scenes and films created on it, whatever order the caller destroys them in: the last
// reference releases the stream and the counters.
inline void rtmi_device_release(rtmi_device *dev) {
    if (dev->refs.fetch_sub(1) != 1) return;
    rti::DeviceScope scope(dev->ordinal);
    if (dev->stream) {
        (void)hipStreamSynchronize(dev->stream);
        (void)hipStreamDestroy(dev->stream);
    }
    if (dev->d_visits) (void)hipFree(dev->d_visits);
    if (dev->copy_stream) {
        (void)hipStreamSynchronize(dev->copy_stream);
        (void)hipStreamDestroy(dev->copy_stream);
    }
    if (dev->h_counts) (void)hipHostFree(dev->h_counts);
    for (int k = 0; k < rtmi_device::COUNT_RING; ++k) {
        if (dev->ev_shaded[k]) (void)hipEventDestroy(dev->ev_shaded[k]);
        if (dev->ev_copied[k]) (void)hipEventDestroy(dev->ev_copied[k]);
    }
    delete dev;
}

struct rtmi_scene {
    rtmi_device *dev = nullptr;
    std::vector<rtd::Material> materials;
    std::vector<float> lights;
    std::vector<rts::HostPrim> prims;
    std::vector<uint32_t> geom_material;
    bool committed = false;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    rtmi_scene_info info{};
    rtd::F4 *d_nodes = nullptr;
    rtd::F4 *d_prims = nullptr;
    uint32_t *d_geom_material = nullptr;
    rtd::Material *d_materials = nullptr;
    uint32_t *d_spill = nullptr; // traversal stack overflow for a full resident grid; used on dev->stream only, under
                                 // dev->lock (renders and rtmi_intersect)
    size_t spill_bytes = 0;
    uint32_t grid_blocks = 0;    // resident grid the spill buffers are sized for (the larger of the two traversal kernels')
    uint32_t grid_blocks_vote = 0;  // resident grid of extend_kernel
    uint32_t grid_blocks_queue = 0; // resident grid of extend_queue_kernel
    uint32_t grid_blocks_tail = 0;  // resident grid of tail_kernel
    rtd::SceneView view{};
    // rtmi_intersect_device launches on caller-chosen streams, possibly several at once: each launch takes the next
    // of a few spill buffers of its own and leaves an event behind; the launch that takes the buffer next waits for
    // that event on its stream.  Buffers are created on first use and only for trees deeper than the LDS stack.
    static constexpr int SPILL_SLOTS = 4;
    struct SpillSlot {
        uint32_t *d = nullptr;
        hipEvent_t done = nullptr;
        bool used = false;
    };
    std::mutex query_lock; // guards slots / next_slot
    SpillSlot slots[SPILL_SLOTS];
    unsigned next_slot = 0;
    // staging of rtmi_intersect (host arrays), kept between calls and grown on demand; used under dev->lock
    size_t stage_cap = 0;      // rays the device pair holds
    rtmi_ray *d_stage_rays = nullptr;
    rtmi_hit *d_stage_hits = nullptr;
    size_t pinned_cap = 0;     // rays the pinned host pair holds (small batches only)
    rtmi_ray *h_stage_rays = nullptr;
    rtmi_hit *h_stage_hits = nullptr;
};

struct rtmi_film {
    rtmi_device *dev = nullptr;
    unsigned long long id = 0; // unique among the films this process has made; tells a gather's peers which film a rank brings
    size_t n_pixels = 0;
    uint32_t *d_xy = nullptr;
    float *d_accum = nullptr;
    bool own_accum = false;
    // workspace, grown on demand
    size_t cap_paths = 0;
    float *d_state = nullptr;
    float *d_hits = nullptr;
    float *d_queue[2] = {nullptr, nullptr}; // ray payload planes, ping-pong by bounce parity
    size_t cap_counters = 0;
    uint32_t *d_counts = nullptr; // counts + work, cap_counters entries each
    float *d_resolve = nullptr;
    uint32_t *d_done = nullptr; // n_pixels flags + 1 count, see rtmi_render
    // a frame film (rtmi_film_create_frame): pixel i is (frame_x + i % frame_w, frame_y + i / frame_w)
    bool is_frame = false;
    uint32_t frame_x = 0, frame_y = 0, frame_w = 0, frame_h = 0;
    struct rtmi_gather_plan *plan = nullptr; // what rtmi_film_gather keeps between calls (comm.cpp)
};

// Releases a film's gather plan (device index lists and staging buffers); the film's device must be current.
void rtmi_gather_plan_free(rtmi_film *film);

