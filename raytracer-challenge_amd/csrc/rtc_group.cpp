// rtc_group.cpp — row tiles across the GPUs of one node (include/rtc.h, "rtc_group"): the multi-GPU form
// of Camera::render_async (camera.rs:144-160 — every pixel an independent work item). Member r renders
// the 8-row bands r, r+N, ... of each frame (rtc_render_views, one launch per member and batch), then ONE
// exchange step: an RCCL gather (ncclGather over xGMI) of the packed f64 tiles to member 0, where one
// kernel un-deals the bands into the reference's row-major Canvas (canvas.rs:43-51).
//
// RCCL is bound at run time (dlopen) — librtc.so has no link dependency on it, a single-GPU caller never
// loads it, and inside a PyTorch process the RCCL torch already loaded is reused (one copy per process).
// No CUDA/NCCL compatibility layer is involved: these are RCCL's own entry points (rccl/rccl.h).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "rtc.h"
#include "rtc_bands.h"
#include "rtc_device.h"
#include "rtc_internal.h"

namespace {

#define HIP_TRY(expr)                                   \
    do {                                                \
        if ((expr) != hipSuccess) return RTC_ERR_DEVICE; \
    } while (0)

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    bool ok = false;
};

Rccl &rccl() {
    static Rccl R = [] {
        Rccl r;
        // a copy that is already mapped (PyTorch's) first, then the system one
        const char *names[] = {"librccl.so", "librccl.so.1"};
        for (const char *n : names)
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char *n : names)
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) r.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) return r;
#define RTC_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name))
        RTC_SYM(GetUniqueId, "ncclGetUniqueId");
        RTC_SYM(CommInitRank, "ncclCommInitRank");
        RTC_SYM(CommInitAll, "ncclCommInitAll");
        RTC_SYM(CommDestroy, "ncclCommDestroy");
        RTC_SYM(Gather, "ncclGather");
        RTC_SYM(GroupStart, "ncclGroupStart");
        RTC_SYM(GroupEnd, "ncclGroupEnd");
#undef RTC_SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommInitAll && r.CommDestroy && r.Gather && r.GroupStart && r.GroupEnd;
        return r;
    }();
    return R;
}

static_assert(RTC_GROUP_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "rtc_group id size must be RCCL's");

struct Member {
    int device = -1;
    uint32_t rank = 0;              // rank in the whole group
    hipStream_t s_render = nullptr; // renders (the member's rtc_context runs on it)
    hipStream_t s_comm = nullptr;   // gathers / peer copies, and on member 0 the un-deal kernel
    rtc_context *ctx = nullptr;
    ncclComm_t comm = nullptr;
    // two tile buffers: the exchange of batch j (s_comm) overlaps the render of batch j+1 (s_render)
    double *tile[2] = {nullptr, nullptr};
    unsigned char *tile8[2] = {nullptr, nullptr};
    size_t tile_cap = 0, tile8_cap = 0; // bytes per buffer
    hipEvent_t rendered[2] = {nullptr, nullptr}, sent[2] = {nullptr, nullptr};
};

} // namespace

struct rtc_group {
    uint32_t nranks = 0;
    uint32_t exchange = RTC_EXCHANGE_RCCL;
    bool in_process = false;
    std::vector<Member> m; // local members, ascending rank
    uint64_t batches = 0;  // rtc_group_render calls so far (buffer parity)
    // member 0's gather destination: nranks chunks of one batch's packed tiles (grow-only)
    double *staging = nullptr;
    unsigned char *staging8 = nullptr;
    size_t staging_cap = 0, staging8_cap = 0;
    bool has_root() const { return !m.empty() && m[0].rank == 0; }
};

struct rtc_group_world {
    rtc_group *g = nullptr;
    std::vector<rtc_world *> w; // one per local member
};

namespace {

rtc_status member_init(Member &mb, int device, uint32_t rank) {
    mb.device = device;
    mb.rank = rank;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&mb.s_render, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&mb.s_comm, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b) {
        HIP_TRY(hipEventCreateWithFlags(&mb.rendered[b], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&mb.sent[b], hipEventDisableTiming));
    }
    return rtc_context_create(device, mb.s_render, &mb.ctx);
}

void member_release(Member &mb) {
    if (mb.device >= 0) (void)hipSetDevice(mb.device);
    if (mb.s_render) (void)hipStreamSynchronize(mb.s_render);
    if (mb.s_comm) (void)hipStreamSynchronize(mb.s_comm);
    if (mb.comm && rccl().ok) (void)rccl().CommDestroy(mb.comm);
    if (mb.ctx) rtc_context_destroy(mb.ctx);
    for (int b = 0; b < 2; ++b) {
        if (mb.tile[b]) (void)hipFree(mb.tile[b]);
        if (mb.tile8[b]) (void)hipFree(mb.tile8[b]);
        if (mb.rendered[b]) (void)hipEventDestroy(mb.rendered[b]);
        if (mb.sent[b]) (void)hipEventDestroy(mb.sent[b]);
    }
    if (mb.s_render) (void)hipStreamDestroy(mb.s_render);
    if (mb.s_comm) (void)hipStreamDestroy(mb.s_comm);
    mb = Member{};
}

template <class T> rtc_status grow(T *&p, size_t &cap, size_t bytes) {
    if (cap >= bytes) return RTC_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), bytes);
    if (e == hipErrorOutOfMemory) return RTC_ERR_NOMEM;
    if (e != hipSuccess) return RTC_ERR_DEVICE;
    cap = bytes;
    return RTC_OK;
}

static_assert(RTC_BANDS_ROWS == RTC_BAND_ROWS, "rtc_bands.h and include/rtc.h disagree");
// the dealing of bands over members lives in rtc_bands.h (shared with k_undeal and the [host] entries below)
uint32_t bands_of(uint32_t vsize) { return rtc_bands_of(vsize); }
uint32_t packed_rows(uint32_t vsize, uint32_t nranks) { return rtc_packed_rows(vsize, nranks); }
uint32_t bands_owned(uint32_t vsize, uint32_t nranks, uint32_t rank) { return rtc_bands_owned(vsize, nranks, rank); }

} // namespace

extern "C" {

rtc_status rtc_group_unique_id(uint8_t id[RTC_GROUP_ID_BYTES]) {
    if (!id) return RTC_ERR_ARG;
    if (!rccl().ok) return RTC_ERR_UNSUPPORTED;
    ncclUniqueId u;
    if (rccl().GetUniqueId(&u) != ncclSuccess) return RTC_ERR_DEVICE;
    std::memcpy(id, u.internal, RTC_GROUP_ID_BYTES);
    return RTC_OK;
}

void rtc_group_destroy(rtc_group *g) {
    if (!g) return;
    for (Member &mb : g->m) member_release(mb);
    if (g->staging) (void)hipFree(g->staging);
    if (g->staging8) (void)hipFree(g->staging8);
    delete g;
}

rtc_status rtc_group_create(const int32_t *devices, uint32_t ndev, uint32_t exchange, rtc_group **out) {
    if (!out) return RTC_ERR_ARG;
    *out = nullptr;
    if (!devices || ndev == 0 || ndev > 64 || exchange > RTC_EXCHANGE_P2P) return RTC_ERR_ARG;
    if (exchange == RTC_EXCHANGE_RCCL) {
        if (!rccl().ok) return RTC_ERR_UNSUPPORTED;
        for (uint32_t i = 0; i < ndev; ++i) // RCCL refuses a device that appears twice; say so as an argument error
            for (uint32_t j = 0; j < i; ++j)
                if (devices[i] == devices[j]) return RTC_ERR_ARG;
    }
    rtc_group *g = new (std::nothrow) rtc_group;
    if (!g) return RTC_ERR_NOMEM;
    g->nranks = ndev;
    g->exchange = exchange;
    g->in_process = true;
    g->m.resize(ndev);
    for (uint32_t i = 0; i < ndev; ++i) {
        const rtc_status st = member_init(g->m[i], devices[i], i);
        if (st != RTC_OK) {
            rtc_group_destroy(g);
            return st;
        }
    }
    if (exchange == RTC_EXCHANGE_RCCL) {
        std::vector<ncclComm_t> comms(ndev, nullptr);
        std::vector<int> devs(devices, devices + ndev);
        if (rccl().CommInitAll(comms.data(), (int)ndev, devs.data()) != ncclSuccess) {
            rtc_group_destroy(g);
            return RTC_ERR_DEVICE;
        }
        for (uint32_t i = 0; i < ndev; ++i) g->m[i].comm = comms[i];
    } else {
        // peer copies into member 0's staging buffer: let every other device reach device 0's memory
        for (uint32_t i = 1; i < ndev; ++i) {
            if (devices[i] == devices[0]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[i], devices[0]) != hipSuccess || !can) {
                rtc_group_destroy(g);
                return RTC_ERR_DEVICE;
            }
            (void)hipSetDevice(devices[i]);
            const hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                rtc_group_destroy(g);
                return RTC_ERR_DEVICE;
            }
            (void)hipGetLastError();
        }
    }
    *out = g;
    return RTC_OK;
}

rtc_status rtc_group_create_rank(int32_t device, uint32_t nranks, uint32_t rank, const uint8_t id[RTC_GROUP_ID_BYTES],
                                 rtc_group **out) {
    if (!out) return RTC_ERR_ARG;
    *out = nullptr;
    if (!id || nranks == 0 || rank >= nranks) return RTC_ERR_ARG;
    if (!rccl().ok) return RTC_ERR_UNSUPPORTED;
    rtc_group *g = new (std::nothrow) rtc_group;
    if (!g) return RTC_ERR_NOMEM;
    g->nranks = nranks;
    g->exchange = RTC_EXCHANGE_RCCL;
    g->in_process = false;
    g->m.resize(1);
    rtc_status st = member_init(g->m[0], device, rank);
    if (st == RTC_OK) {
        ncclUniqueId u;
        std::memcpy(u.internal, id, RTC_GROUP_ID_BYTES);
        if (hipSetDevice(device) != hipSuccess || rccl().CommInitRank(&g->m[0].comm, (int)nranks, u, (int)rank) != ncclSuccess)
            st = RTC_ERR_DEVICE;
    }
    if (st != RTC_OK) {
        rtc_group_destroy(g);
        return st;
    }
    *out = g;
    return RTC_OK;
}

uint32_t rtc_group_size(const rtc_group *g) { return g ? g->nranks : 0u; }
uint32_t rtc_group_local_size(const rtc_group *g) { return g ? (uint32_t)g->m.size() : 0u; }

rtc_context *rtc_group_context(rtc_group *g, uint32_t i) { return (g && i < g->m.size()) ? g->m[i].ctx : nullptr; }

rtc_status rtc_group_synchronize(rtc_group *g) {
    if (!g) return RTC_ERR_ARG;
    for (Member &mb : g->m) {
        HIP_TRY(hipSetDevice(mb.device));
        HIP_TRY(hipStreamSynchronize(mb.s_render));
        HIP_TRY(hipStreamSynchronize(mb.s_comm));
    }
    return RTC_OK;
}

rtc_status rtc_group_world_create(rtc_group *g, const rtc_shape *shapes, uint32_t n_shapes, const rtc_light *light,
                                  rtc_group_world **out) {
    if (!g || !out) return RTC_ERR_ARG;
    *out = nullptr;
    rtc_group_world *gw = new (std::nothrow) rtc_group_world;
    if (!gw) return RTC_ERR_NOMEM;
    gw->g = g;
    for (Member &mb : g->m) {
        rtc_world *w = nullptr;
        const rtc_status st = rtc_world_create(mb.ctx, shapes, n_shapes, light, &w);
        if (st != RTC_OK) {
            rtc_group_world_destroy(gw);
            return st;
        }
        gw->w.push_back(w);
    }
    *out = gw;
    return RTC_OK;
}

void rtc_group_world_destroy(rtc_group_world *w) {
    if (!w) return;
    for (rtc_world *x : w->w) rtc_world_destroy(x);
    delete w;
}

// One launch per local member: its bands of `nframes` frames into tile buffer `b` (after the exchange that
// last read that buffer); the member's exchange stream is made to wait for the launch.
static rtc_status render_members(rtc_group *g, const rtc_group_world *w, const rtc_camera *cams, uint32_t nframes, uint32_t mode,
                                 uint32_t flags, bool want8, int b, bool want64 = true) {
    const uint32_t W = cams[0].hsize, H = cams[0].vsize, N = g->nranks;
    const uint32_t rows = packed_rows(H, N);
    const size_t tile_bytes = (size_t)nframes * rows * W * 3u * sizeof(double), tile8_bytes = (size_t)nframes * rows * W * 3u;
    for (size_t i = 0; i < g->m.size(); ++i) {
        Member &mb = g->m[i];
        HIP_TRY(hipSetDevice(mb.device));
        if (want64 && mb.tile_cap < tile_bytes) { // both buffers grow together (hipFree waits for the device)
            size_t c0 = mb.tile[0] ? mb.tile_cap : 0, c1 = mb.tile[1] ? mb.tile_cap : 0;
            rtc_status st = grow(mb.tile[0], c0, tile_bytes);
            if (st == RTC_OK) st = grow(mb.tile[1], c1, tile_bytes);
            if (st != RTC_OK) return st;
            mb.tile_cap = tile_bytes;
        }
        if (want8 && mb.tile8_cap < tile8_bytes) {
            size_t c0 = mb.tile8[0] ? mb.tile8_cap : 0, c1 = mb.tile8[1] ? mb.tile8_cap : 0;
            rtc_status st = grow(mb.tile8[0], c0, tile8_bytes);
            if (st == RTC_OK) st = grow(mb.tile8[1], c1, tile8_bytes);
            if (st != RTC_OK) return st;
            mb.tile8_cap = tile8_bytes;
        }
        HIP_TRY(hipStreamWaitEvent(mb.s_render, mb.sent[b], 0)); // an event never recorded does not block
        const rtc_status st = rtc_render_views(mb.ctx, w->w[i], cams, nframes, mode, mb.rank, N, want64 ? mb.tile[b] : nullptr,
                                               want8 ? mb.tile8[b] : nullptr, rows, flags);
        if (st != RTC_OK) return st;
        HIP_TRY(hipEventRecord(mb.rendered[b], mb.s_render));
        HIP_TRY(hipStreamWaitEvent(mb.s_comm, mb.rendered[b], 0));
    }
    return RTC_OK;
}

rtc_status rtc_group_render(rtc_group *g, const rtc_group_world *w, const rtc_camera *cams, uint32_t nframes, uint32_t mode,
                            uint32_t flags, uint32_t what, void *d_canvas, void *d_rgb8) {
    if (!g || !w || !cams || w->g != g || w->w.size() != g->m.size()) return RTC_ERR_ARG;
    if (nframes == 0 || nframes > RTC_MAX_VIEWS_PER_LAUNCH || what > (RTC_GATHER_F64 | RTC_GATHER_U8)) return RTC_ERR_ARG;
    const uint32_t W = cams[0].hsize, H = cams[0].vsize, N = g->nranks;
    if (W == 0 || H == 0) return RTC_ERR_ARG;
    const bool f64 = (what & RTC_GATHER_F64) != 0, u8 = (what & RTC_GATHER_U8) != 0;
    if (g->has_root() && ((f64 && !d_canvas) || (u8 && !d_rgb8))) return RTC_ERR_ARG;
    const uint32_t rows = packed_rows(H, N);
    const size_t row_bytes = (size_t)W * 3u * sizeof(double), row8 = (size_t)W * 3u;
    const size_t tile_bytes = (size_t)nframes * rows * row_bytes, tile8_bytes = (size_t)nframes * rows * row8;
    const int b = (int)(g->batches & 1u);
    {
        const rtc_status st = render_members(g, w, cams, nframes, mode, flags, u8, b);
        if (st != RTC_OK) return st;
    }
    ++g->batches;
    if (what == RTC_GATHER_NONE) {
        for (Member &mb : g->m) {
            HIP_TRY(hipSetDevice(mb.device));
            HIP_TRY(hipEventRecord(mb.sent[b], mb.s_comm));
        }
        return RTC_OK;
    }

    // ---- exchange: gather the packed tiles to member 0's staging buffer
    if (g->has_root()) {
        HIP_TRY(hipSetDevice(g->m[0].device));
        rtc_status st = RTC_OK;
        if (f64) st = grow(g->staging, g->staging_cap, (size_t)N * tile_bytes);
        if (st == RTC_OK && u8) st = grow(g->staging8, g->staging8_cap, (size_t)N * tile8_bytes);
        if (st != RTC_OK) return st;
    }
    if (g->exchange == RTC_EXCHANGE_RCCL) {
        Rccl &R = rccl();
        for (int pass = 0; pass < 2; ++pass) { // pass 0: f64 tiles, pass 1: 8-bit tiles
            if ((pass == 0 && !f64) || (pass == 1 && !u8)) continue;
            if (g->m.size() > 1 && R.GroupStart() != ncclSuccess) return RTC_ERR_DEVICE;
            bool ok = true;
            for (Member &mb : g->m) {
                if (hipSetDevice(mb.device) != hipSuccess) ok = false;
                void *recv = mb.rank == 0 ? (pass == 0 ? (void *)g->staging : (void *)g->staging8) : nullptr;
                const void *send = pass == 0 ? (const void *)mb.tile[b] : (const void *)mb.tile8[b];
                const size_t count = pass == 0 ? tile_bytes / sizeof(double) : tile8_bytes;
                if (R.Gather(send, recv, count, pass == 0 ? ncclDouble : ncclUint8, 0, mb.comm, mb.s_comm) != ncclSuccess) ok = false;
            }
            if (g->m.size() > 1 && R.GroupEnd() != ncclSuccess) ok = false;
            if (!ok) return RTC_ERR_DEVICE;
        }
    } else { // RTC_EXCHANGE_P2P (in-process): SDMA peer copies, each on the sending member's exchange stream
        Member &root = g->m[0];
        for (Member &mb : g->m) {
            HIP_TRY(hipSetDevice(mb.device));
            // the staging chunk may still be read by the previous batch's un-deal kernel (root's s_comm)
            if (mb.rank != 0) HIP_TRY(hipStreamWaitEvent(mb.s_comm, root.sent[b ^ 1], 0));
            if (f64)
                HIP_TRY(hipMemcpyPeerAsync(reinterpret_cast<char *>(g->staging) + (size_t)mb.rank * tile_bytes, root.device, mb.tile[b],
                                           mb.device, tile_bytes, mb.s_comm));
            if (u8)
                HIP_TRY(hipMemcpyPeerAsync(g->staging8 + (size_t)mb.rank * tile8_bytes, root.device, mb.tile8[b], mb.device, tile8_bytes,
                                           mb.s_comm));
            HIP_TRY(hipEventRecord(mb.sent[b], mb.s_comm));
        }
        HIP_TRY(hipSetDevice(root.device));
        for (size_t i = 1; i < g->m.size(); ++i) HIP_TRY(hipStreamWaitEvent(root.s_comm, g->m[i].sent[b], 0));
    }
    // ---- member 0: un-deal the bands into row-major frames
    if (g->has_root()) {
        Member &root = g->m[0];
        HIP_TRY(hipSetDevice(root.device));
        if (f64) HIP_TRY(rtc_launch_undeal(g->staging, d_canvas, N, nframes, H, rows, row_bytes, root.s_comm));
        if (u8) HIP_TRY(rtc_launch_undeal(g->staging8, d_rgb8, N, nframes, H, rows, row8, root.s_comm));
    }
    for (Member &mb : g->m) {
        HIP_TRY(hipSetDevice(mb.device));
        HIP_TRY(hipEventRecord(mb.sent[b], mb.s_comm));
    }
    return RTC_OK;
}

rtc_status rtc_group_render_host(rtc_group *g, const rtc_group_world *w, const rtc_camera *cam, uint32_t mode, uint32_t flags,
                                 double *rgb, rtc_stats *stats) {
    if (!g || !w || !cam || !rgb || w->g != g || w->w.size() != g->m.size()) return RTC_ERR_ARG;
    const uint32_t W = cam->hsize, H = cam->vsize, N = g->nranks;
    if (W == 0 || H == 0) return RTC_ERR_ARG;
    if (cam->samples > 255u) return RTC_ERR_ARG;
    const int b = (int)(g->batches & 1u);
    rtc_status st = RTC_OK;
    if (stats) st = rtc_group_stats_reset(g);
    if (st == RTC_OK) st = render_members(g, w, cam, 1, mode, flags, false, b);
    if (st != RTC_OK) return st;
    ++g->batches;
    const size_t row_bytes = (size_t)W * 3u * sizeof(double), band_bytes = row_bytes * RTC_BAND_ROWS;
    const uint32_t nb = bands_of(H);
    // every member's bands go straight to their rows of the host canvas: one strided DMA per member
    // (band k of member r = image rows (r + k*N)*8 ..), each over that GPU's own PCIe link
    for (Member &mb : g->m) {
        if (mb.rank >= nb) continue;
        HIP_TRY(hipSetDevice(mb.device));
        const uint32_t mine = bands_owned(H, N, mb.rank);
        const bool last_short = (H % RTC_BAND_ROWS) != 0 && (mb.rank + (mine - 1u) * N) == nb - 1u;
        const uint32_t whole = last_short ? mine - 1u : mine;
        char *dst = reinterpret_cast<char *>(rgb) + (size_t)rtc_packed_row_to_image(mb.rank, 0u, N) * row_bytes; // = rank * band_bytes
        if (whole)
            HIP_TRY(hipMemcpy2DAsync(dst, (size_t)N * band_bytes, mb.tile[b], band_bytes, band_bytes, whole, hipMemcpyDeviceToHost, mb.s_comm));
        if (last_short)
            HIP_TRY(hipMemcpyAsync(dst + (size_t)whole * N * band_bytes, reinterpret_cast<char *>(mb.tile[b]) + (size_t)whole * band_bytes,
                                   (size_t)(H % RTC_BAND_ROWS) * row_bytes, hipMemcpyDeviceToHost, mb.s_comm));
        HIP_TRY(hipEventRecord(mb.sent[b], mb.s_comm));
    }
    st = rtc_group_synchronize(g);
    if (st == RTC_OK && stats) st = rtc_group_stats_read(g, stats);
    return st;
}

rtc_status rtc_group_render_host_rgb8(rtc_group *g, const rtc_group_world *w, const rtc_camera *cam, uint32_t mode, uint32_t flags,
                                      uint8_t *rgb8, rtc_stats *stats) {
    if (!g || !w || !cam || !rgb8 || w->g != g || w->w.size() != g->m.size()) return RTC_ERR_ARG;
    const uint32_t W = cam->hsize, H = cam->vsize, N = g->nranks;
    if (W == 0 || H == 0) return RTC_ERR_ARG;
    if (cam->samples > 255u) return RTC_ERR_ARG;
    const int b = (int)(g->batches & 1u);
    rtc_status st = RTC_OK;
    if (stats) st = rtc_group_stats_reset(g);
    if (st == RTC_OK) st = render_members(g, w, cam, 1, mode, flags, true, b, false); // 8-bit rows only
    if (st != RTC_OK) return st;
    ++g->batches;
    const size_t row_bytes = (size_t)W * 3u, band_bytes = row_bytes * RTC_BAND_ROWS;
    const uint32_t nb = bands_of(H);
    for (Member &mb : g->m) { // band k of member r = image rows (r + k*N)*8 ..: one strided DMA per member, as for the f64 canvas
        if (mb.rank >= nb) continue;
        HIP_TRY(hipSetDevice(mb.device));
        const uint32_t mine = bands_owned(H, N, mb.rank);
        const bool last_short = (H % RTC_BAND_ROWS) != 0 && (mb.rank + (mine - 1u) * N) == nb - 1u;
        const uint32_t whole = last_short ? mine - 1u : mine;
        unsigned char *dst = rgb8 + (size_t)mb.rank * band_bytes;
        if (whole)
            HIP_TRY(hipMemcpy2DAsync(dst, (size_t)N * band_bytes, mb.tile8[b], band_bytes, band_bytes, whole, hipMemcpyDeviceToHost, mb.s_comm));
        if (last_short)
            HIP_TRY(hipMemcpyAsync(dst + (size_t)whole * N * band_bytes, mb.tile8[b] + (size_t)whole * band_bytes,
                                   (size_t)(H % RTC_BAND_ROWS) * row_bytes, hipMemcpyDeviceToHost, mb.s_comm));
        HIP_TRY(hipEventRecord(mb.sent[b], mb.s_comm));
    }
    st = rtc_group_synchronize(g);
    if (st == RTC_OK && stats) st = rtc_group_stats_read(g, stats);
    return st;
}

rtc_status rtc_group_stats_read(rtc_group *g, rtc_stats *out) {
    if (!g || !out) return RTC_ERR_ARG;
    std::memset(out, 0, sizeof *out);
    for (Member &mb : g->m) {
        rtc_stats s;
        const rtc_status st = rtc_stats_read(mb.ctx, &s);
        if (st != RTC_OK) return st;
        out->rays_primary += s.rays_primary;
        out->rays_shadow += s.rays_shadow;
        out->rays_reflect += s.rays_reflect;
        out->rays_refract += s.rays_refract;
        out->pixels += s.pixels;
        out->pixels_resample += s.pixels_resample;
    }
    return RTC_OK;
}

// ---- [host] the dealing itself, for callers that lay out their own buffers and for the CPU tests ----------------
uint32_t rtc_group_packed_rows(uint32_t vsize, uint32_t nranks) { return nranks ? rtc_packed_rows(vsize, nranks) : 0u; }
uint32_t rtc_group_bands_owned(uint32_t vsize, uint32_t nranks, uint32_t rank) { return nranks ? rtc_bands_owned(vsize, nranks, rank) : 0u; }
void rtc_group_row_owner(uint32_t y, uint32_t nranks, uint32_t *member, uint32_t *packed_row) {
    if (!nranks || !member || !packed_row) return;
    rtc_row_owner(y, nranks, member, packed_row);
}
uint32_t rtc_group_packed_row_to_image(uint32_t member, uint32_t packed_row, uint32_t nranks) {
    return nranks ? rtc_packed_row_to_image(member, packed_row, nranks) : 0u;
}
rtc_status rtc_group_undeal_host(const void *staging, void *canvas, uint32_t nranks, uint32_t nframes, uint32_t vsize, size_t row_bytes) {
    if (!staging || !canvas || nranks == 0) return RTC_ERR_ARG;
    const uint32_t rows_max = rtc_packed_rows(vsize, nranks);
    const unsigned char *src = static_cast<const unsigned char *>(staging);
    unsigned char *dst = static_cast<unsigned char *>(canvas);
    for (uint32_t f = 0; f < nframes; ++f)
        for (uint32_t y = 0; y < vsize; ++y) // k_undeal with one unit per row: the same index function (rtc_bands.h)
            std::memcpy(dst + ((size_t)f * vsize + y) * row_bytes, src + rtc_staging_index(f, y, 0u, nranks, nframes, rows_max, 1u) * row_bytes, row_bytes);
    return RTC_OK;
}

rtc_status rtc_group_stats_reset(rtc_group *g) {
    if (!g) return RTC_ERR_ARG;
    for (Member &mb : g->m) {
        const rtc_status st = rtc_stats_reset(mb.ctx);
        if (st != RTC_OK) return st;
    }
    return RTC_OK;
}

} // extern "C"
