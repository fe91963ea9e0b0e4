// multi.hip — single-process lane split over several devices (include/idsp_hip.h, "idsp_multi_*").
//
// Lanes never interact (`Lanes::process` touches state[i], x[i] only, dsp-process/src/compose.rs:468-476), so G
// devices take G contiguous lane blocks and there is NO data-path exchange: this file is host-side bookkeeping —
// which device owns which lanes, one stream per device, launch everywhere, wait for everyone.  It exists for hosts
// without a process launcher of their own (the Rust shim, a C program); the Python benchmark uses one process per
// GPU with torch.distributed instead, and both forms call the same per-device entry points.
#include <vector>

#include "common.h"

struct idsp_multi {
    std::vector<int> devices;
    std::vector<hipStream_t> streams;
};

namespace {

// GPU g of G gets lanes [g L / G, (g + 1) L / G) — SURVEY.md 8e, idsp_amd/sharding.py lane_shard
inline void shard(size_t lanes, size_t g, size_t G, size_t &lo, size_t &hi)
{
    lo = lanes * g / G;
    hi = lanes * (g + 1) / G;
}

struct DeviceGuard {  // restores the caller's current device
    int prev = 0;
    bool ok;
    DeviceGuard() : ok(hipGetDevice(&prev) == hipSuccess) {}
    ~DeviceGuard()
    {
        if (ok) (void)hipSetDevice(prev);
    }
};

// index of the lane block at which the last idsp_multi_for_each / idsp_multi_biquad_* of this thread stopped (-1: none)
inline int &last_block()
{
    static thread_local int b = -1;
    return b;
}

inline bool mul_overflows(size_t a, size_t b) { return b != 0 && a > SIZE_MAX / b; }

}  // namespace

using namespace idsp;

extern "C" {

int idsp_multi_create(const int *devices, int n_devices, idsp_multi **out)
{
    if (!out) return fail(IDSP_EINVAL, "out is NULL");
    *out = nullptr;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) return fail(IDSP_ENODEV, "no HIP device visible");
    if (n_devices < 1) {
        if (devices) return fail(IDSP_EINVAL, "n_devices = %d with a device list", n_devices);
        n_devices = visible;  // all visible devices
    }
    DeviceGuard guard;
    auto *m = new idsp_multi;
    for (int i = 0; i < n_devices; i++) {
        const int d = devices ? devices[i] : i;
        if (d < 0 || d >= visible) {
            delete m;
            return fail(IDSP_EINVAL, "device %d not in 0..%d", d, visible - 1);
        }
        m->devices.push_back(d);
    }
    for (int d : m->devices) {
        hipStream_t s = nullptr;
        if (hipSetDevice(d) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
            for (size_t j = 0; j < m->streams.size(); j++) {
                (void)hipSetDevice(m->devices[j]);
                (void)hipStreamDestroy(m->streams[j]);
            }
            delete m;
            return fail(IDSP_EHIP, "stream creation on device %d failed", d);
        }
        m->streams.push_back(s);
    }
    *out = m;
    return IDSP_OK;
}

int idsp_multi_destroy(idsp_multi *m)
{
    if (!m) return IDSP_OK;
    DeviceGuard guard;
    for (size_t j = 0; j < m->streams.size(); j++) {
        (void)hipSetDevice(m->devices[j]);
        (void)hipStreamSynchronize(m->streams[j]);
        (void)hipStreamDestroy(m->streams[j]);
    }
    delete m;
    return IDSP_OK;
}

int idsp_multi_size(const idsp_multi *m) { return m ? int(m->devices.size()) : fail(IDSP_EINVAL, "m is NULL"); }

int idsp_multi_device(const idsp_multi *m, int index)
{
    if (!m || index < 0 || size_t(index) >= m->devices.size()) return fail(IDSP_EINVAL, "bad shard index %d", index);
    return m->devices[size_t(index)];
}

void *idsp_multi_stream(const idsp_multi *m, int index)
{
    if (!m || index < 0 || size_t(index) >= m->devices.size()) return nullptr;
    return m->streams[size_t(index)];
}

int idsp_multi_shard(const idsp_multi *m, size_t lanes, int index, size_t *lane_lo, size_t *lane_hi)
{
    if (!m || !lane_lo || !lane_hi || index < 0 || size_t(index) >= m->devices.size())
        return fail(IDSP_EINVAL, "bad arguments (index %d)", index);
    shard(lanes, size_t(index), m->devices.size(), *lane_lo, *lane_hi);
    return IDSP_OK;
}

int idsp_multi_for_each(idsp_multi *m, size_t lanes, idsp_shard_fn fn, void *user)
{
    last_block() = -1;
    if (!m || !fn) return fail(IDSP_EINVAL, "m or fn is NULL");
    DeviceGuard guard;
    const size_t G = m->devices.size();
    for (size_t g = 0; g < G; g++) {
        size_t lo, hi;
        shard(lanes, g, G, lo, hi);
        IDSP_HIP_TRY(hipSetDevice(m->devices[g]));
        const int rc = fn(user, int(g), lo, hi, m->streams[g]);
        // Any non-zero status stops the walk and is returned as it is (negative: the callee's idsp_status, with its
        // idsp_last_error() text; positive: the callback's own "stop" code).  Blocks 0..g-1 have been handed to fn
        // already and whatever they launched is in flight on their streams: after a failure the per-block state is
        // indeterminate — idsp_multi_sync() and reload it.  idsp_multi_last_block() tells which block stopped.
        if (rc != 0) {
            last_block() = int(g);
            return rc;
        }
    }
    last_block() = -1;
    return IDSP_OK;
}

int idsp_multi_last_block(void) { return last_block(); }

int idsp_multi_sync(idsp_multi *m)
{
    if (!m) return fail(IDSP_EINVAL, "m is NULL");
    DeviceGuard guard;
    for (size_t g = 0; g < m->devices.size(); g++) {
        IDSP_HIP_TRY(hipSetDevice(m->devices[g]));
        IDSP_HIP_TRY(hipStreamSynchronize(m->streams[g]));
    }
    return IDSP_OK;
}

int idsp_multi_alloc(idsp_multi *m, size_t lanes, size_t bytes_per_lane, void **ptrs)
{
    if (!m || !ptrs) return fail(IDSP_EINVAL, "m or ptrs is NULL");
    if (mul_overflows(lanes, bytes_per_lane)) return fail(IDSP_EINVAL, "lanes * bytes_per_lane overflows size_t");
    DeviceGuard guard;
    const size_t G = m->devices.size();
    for (size_t g = 0; g < G; g++) ptrs[g] = nullptr;
    for (size_t g = 0; g < G; g++) {
        size_t lo, hi;
        shard(lanes, g, G, lo, hi);
        IDSP_HIP_TRY(hipSetDevice(m->devices[g]));
        const size_t bytes = (hi - lo) * bytes_per_lane;
        hipError_t e = hipMalloc(&ptrs[g], bytes ? bytes : 1);
        if (e == hipSuccess) e = hipMemsetAsync(ptrs[g], 0, bytes, m->streams[g]);
        if (e != hipSuccess) {
            for (size_t j = 0; j <= g; j++)
                if (ptrs[j]) {
                    (void)hipSetDevice(m->devices[j]);
                    (void)hipFree(ptrs[j]);
                    ptrs[j] = nullptr;
                }
            return fail(IDSP_EHIP, "allocation of %zu bytes on device %d: %s", bytes, m->devices[g], hipGetErrorString(e));
        }
    }
    return idsp_multi_sync(m);
}

int idsp_multi_free(idsp_multi *m, void **ptrs)
{
    if (!m || !ptrs) return fail(IDSP_EINVAL, "m or ptrs is NULL");
    DeviceGuard guard;
    for (size_t g = 0; g < m->devices.size(); g++)
        if (ptrs[g]) {
            IDSP_HIP_TRY(hipSetDevice(m->devices[g]));
            IDSP_HIP_TRY(hipFree(ptrs[g]));
            ptrs[g] = nullptr;
        }
    return IDSP_OK;
}

int idsp_multi_copy(idsp_multi *m, size_t lanes, size_t bytes_per_lane, void *const *dev_ptrs, void *host, int to_device)
{
    if (!m || !dev_ptrs || (!host && lanes)) return fail(IDSP_EINVAL, "m, dev_ptrs or host is NULL");
    if (mul_overflows(lanes, bytes_per_lane)) return fail(IDSP_EINVAL, "lanes * bytes_per_lane overflows size_t");
    DeviceGuard guard;
    const size_t G = m->devices.size();
    for (size_t g = 0; g < G; g++) {  // nothing is queued unless every non-empty block has a buffer
        size_t lo, hi;
        shard(lanes, g, G, lo, hi);
        if (hi > lo && !dev_ptrs[g]) return fail(IDSP_EINVAL, "dev_ptrs[%zu] is NULL for a block of %zu lanes", g, hi - lo);
    }
    for (size_t g = 0; g < G; g++) {
        size_t lo, hi;
        shard(lanes, g, G, lo, hi);
        if (hi == lo) continue;
        IDSP_HIP_TRY(hipSetDevice(m->devices[g]));
        char *h = static_cast<char *>(host) + lo * bytes_per_lane;
        if (to_device)
            IDSP_HIP_TRY(hipMemcpyAsync(dev_ptrs[g], h, (hi - lo) * bytes_per_lane, hipMemcpyHostToDevice, m->streams[g]));
        else
            IDSP_HIP_TRY(hipMemcpyAsync(h, dev_ptrs[g], (hi - lo) * bytes_per_lane, hipMemcpyDeviceToHost, m->streams[g]));
    }
    return IDSP_OK;
}

}  // extern "C"

// ---- the two headline operators over a lane split (BASELINE.json configs[1] and configs[4]) -------------------
namespace {
template <class Cfg, class T, class Fn>
int multi_biquad(idsp_multi *m, Fn entry, const Cfg *cfg, size_t n, void *const *state, const T *const *x, T *const *y, size_t lanes,
                 size_t frames, int layout)
{
    last_block() = -1;  // reset before ANY return: a stale value must not be read as this call's block
    if (!m || !state || !x || !y) return fail(IDSP_EINVAL, "m, state, x or y is NULL");
    DeviceGuard guard;
    const size_t G = m->devices.size();
    // Validate EVERY block before launching any: a zero-frame call of the entry runs all of its argument checks
    // (configuration, section count, layout, NULL buffers) and launches nothing, so a bad block is reported before
    // any block's state has advanced.
    for (size_t g = 0; g < G; g++) {
        size_t lo, hi;
        shard(lanes, g, G, lo, hi);
        if (hi == lo) continue;
        if (!state[g] || !x[g] || !y[g]) {
            last_block() = int(g);
            return fail(IDSP_EINVAL, "state, x or y of block %zu is NULL (%zu lanes)", g, hi - lo);
        }
        // the probe runs with block g's device current, like the launch below: an entry is free to touch HIP state in its checks
        IDSP_HIP_TRY(hipSetDevice(m->devices[g]));
        const int rc = entry(cfg, n, state[g], x[g], y[g], hi - lo, 0, layout, m->streams[g]);
        if (rc < 0) {
            last_block() = int(g);
            return rc;
        }
    }
    for (size_t g = 0; g < G; g++) {
        size_t lo, hi;
        shard(lanes, g, G, lo, hi);
        if (hi == lo) continue;
        IDSP_HIP_TRY(hipSetDevice(m->devices[g]));
        const int rc = entry(cfg, n, state[g], x[g], y[g], hi - lo, frames, layout, m->streams[g]);
        if (rc < 0) {  // a launch failure after validation: blocks 0..g-1 are in flight, state indeterminate
            last_block() = int(g);
            return rc;
        }
    }
    return IDSP_OK;
}
}  // namespace

extern "C" {

int idsp_multi_biquad_i32_df1(idsp_multi *m, const idsp_biquad_i32 *cfg, size_t n, void *const *state, const int32_t *const *x,
                              int32_t *const *y, size_t lanes, size_t frames, int layout)
{
    return multi_biquad(m, idsp_biquad_i32_df1, cfg, n, state, x, y, lanes, frames, layout);
}

int idsp_multi_biquad_f32_df2t(idsp_multi *m, const idsp_biquad_f32 *cfg, size_t n, void *const *state, const float *const *x,
                               float *const *y, size_t lanes, size_t frames, int layout)
{
    return multi_biquad(m, idsp_biquad_f32_df2t, cfg, n, state, x, y, lanes, frames, layout);
}

}  // extern "C"
