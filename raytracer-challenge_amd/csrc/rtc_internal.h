// rtc_internal.h — host-side objects behind the opaque handles of include/rtc.h; shared by rtc_api.cpp
// (one GPU) and rtc_group.cpp (row tiles across GPUs). Not part of the ABI.
#ifndef RTC_INTERNAL_H
#define RTC_INTERNAL_H

#include <hip/hip_runtime.h>

#include <vector>

#include "rtc.h"
#include "rtc_device.h"

enum { SRC_SMEM = 0, SRC_LDS1 = 1, SRC_LDSN = 2, SRC_CULL = 3, SRC_CULL2 = 4 };

struct rtc_context {
    int device = -1;
    unsigned long long render_allocs = 0; // hipMalloc calls made by render entry points (rtc_debug_render_allocs)
    unsigned long long pixels = 0; // rtc_stats::pixels of the launches since the last reset (counted by render_launch)
    hipStream_t stream = nullptr;
    unsigned long long *d_counters = nullptr;
    // ring of (begin, end) event pairs, one per timed k_trace launch. Created on demand, EV_CHUNK pairs
    // at a time (a context that never renders creates none; creating all 2048 up front made
    // rtc_context_create the slowest call of a one-frame render)
    static constexpr uint32_t EV_RING = 1024, EV_CHUNK = 16;
    hipEvent_t ev[EV_RING][2] = {};
    hipEvent_t ev_bin[EV_RING][2] = {}; // the same for the launch's binning kernel (k_bin_tiles), when it has one
    bool bin_timed[EV_RING] = {};       // ... whether slot k's launch had one
    uint32_t ev_created = 0; // pairs [0, ev_created) exist
    uint64_t launches = 0; // render launches so far
    uint64_t timed = 0;    // ... of which carried an event pair (ring position)
    uint32_t time_every = 1; // rtc_context_set_timing
    // device canvas of rtc_render (host-canvas entry point): grow-only, reused between frames
    double *d_canvas = nullptr;
    size_t canvas_bytes = 0;
    unsigned char *d_canvas8 = nullptr; // the same for rtc_render_rgb8 (3 B/pixel)
    size_t canvas8_bytes = 0;
    int force_src = -1;   // RTC_SRC env override (experiments)
    uint32_t tiles_per_wg = 1; // tiles one workgroup renders in sequence (RTC_TILES_PER_WG)
    uint32_t tiles_guided_tenths = 20; // guided chunks: tiles per chunk level in tenths of the resident workgroups (RTC_TILES_GUIDED; 0 = off)
    uint32_t tiles_slots = 0;          // 0: from the kernel's occupancy; else the number of resident workgroups to assume (RTC_TILES_SLOTS, tests)
    uint32_t tiles_kmax = 8;           // ... largest chunk (RTC_TILES_KMAX: 1, 2, 3, 4 or 8)
    uint32_t tile_cap = 512;
    hipStream_t side_stream = nullptr; // created on demand: per-render binning kernels run here, beside the previous launch's render
    // Pipelined launches (rtc_context_set_pipeline, include/rtc.h): `lanes` > 1 deals consecutive render launches round-robin
    // over that many streams of the context's own, so launch i+1 fills the CUs launch i's last waves leave idle (Camera::
    // render_async returns a NEW Canvas per call, canvas.rs:26-41: consecutive frames never alias). A lane is in order:
    // [k_bin_tiles ->] k_trace, its own set of tile lists (rtc_world::bin[lane]), no cross-stream events at all.
    static constexpr uint32_t MAX_LANES = 4;
    hipStream_t lane[MAX_LANES] = {};
    uint32_t lanes = 1;       // 1 = every launch in order on `stream` (the default)
    uint64_t lane_next = 0;   // launches dealt so far
    // the source / lists the most recent render launch ran with (rtc_context_last_launch_info)
    rtc_launch_info last{};
    uint64_t launches_total = 0; // render launches since the context was created (never reset)
    hipEvent_t fence_ev = nullptr; // rtc_context_fence
    bool light_lists = true; // RTC_LIGHT_LISTS=0: shadow passes of two-level worlds walk the groups (A/B)
    bool binning = true;  // RTC_BINNING=0: primary rays take the wave-level cull / group walk too (A/B)
    bool sky_rows = true; // RTC_SKY_ROWS=0: tile rows the binning kernel proved black are traced like any other (A/B)
    // one-level worlds (<= 256 objects) are binned only in launches of at least this many views (RTC_BIN_SMALL_VIEWS): the
    // binning kernels run on the side stream beside the previous launch's render, which hides them when launches follow each
    // other (north star, 8 views per launch: 0.0745 -> 0.0685 ms per frame; 4 views: 0.0732 -> 0.0704; C4 1.48 -> 1.42;
    // C2 unchanged) but not in front of a lone one-view launch (0.0824 -> 0.0895). Before the side stream the same binning
    // gave the render kernel its 12 % and took it all back in launch latency (profiles/r02_exp_binned_small_worlds.log).
    // one-level worlds (n <= 256) are binned when the launch is long enough for the extra kernel and its two cross-stream
    // events to pay: views x pixels >= this (RTC_BIN_SMALL_PIXELS; 1080p: from 3 views per launch, 4096^2: always)
    unsigned long long bin_small_pixels = 6000000ull;
    // ... and in a pipelined context (lanes > 1), where the binning kernel of launch i+1 runs beside launch i's render on the
    // other lane without any event: from this many pixels (RTC_BIN_SMALL_PIXELS_PIPELINED)
    unsigned long long bin_small_pixels_pipelined = 1500000ull;
};

struct rtc_world {
    rtc_context *ctx = nullptr; // identity check only; never dereferenced at destroy time
    int device = -1;
    uint32_t n = 0;
    DevIsect *d_isect = nullptr;
    uint32_t *d_kind = nullptr;
    DevShade *d_shade = nullptr;
    DevPrim *d_prim = nullptr;
    DevBound *d_bound = nullptr;
    DevIsect *d_isect_s = nullptr; // Morton-sorted copies for the two-level cull
    uint32_t *d_kind_s = nullptr;
    DevBound *d_bound_s = nullptr;
    uint32_t *d_orig_s = nullptr;
    DevBound *d_gbound = nullptr;
    DevIdEntry *d_idtab = nullptr;
    DevPre *d_pre = nullptr, *d_pre_s = nullptr; // per-lane prefilter records (insertion / sorted order)
    double pre_limit = 0.;
    // binned primary pass: per-render scratch, grow-only, TWO sets — the binning of launch k+1 runs on the context's side
    // stream while launch k's render kernel still reads set k (rtc_render_* take the World as const: mutable)
    struct BinSet {
        uint32_t *tile_cnt = nullptr, *tile_list = nullptr; // per (view, tile): entries used, RTC_TILE_LIST_CAP entry slots (tile_cnt: RTC_BIN_ROW_WORDS row words first)
        size_t tiles_cap = 0;            // capacity in (view, tile) entries
        DevPrim *prim = nullptr;         // per (view, object): the primary rays' constants, written by the set's binning kernel
        size_t prim_cap = 0;             // capacity in records
        hipEvent_t binned = nullptr;     // recorded on the side stream after the set's binning kernel
        hipEvent_t traced = nullptr;     // recorded on the render stream after the render kernel that read the set
    };
    mutable BinSet bin[rtc_context::MAX_LANES]; // in-order contexts alternate between [0] and [1]; a pipelined context's lane l owns [l]
    mutable uint32_t bin_next = 0;
    // light-space shadow lists (two-level worlds), built once at rtc_world_create
    DevTileBundle *d_light_cells = nullptr;
    uint32_t *d_light_cnt = nullptr, *d_light_list = nullptr;
    double light_reach = 0.;
    uint32_t light_cap = 0;
    uint32_t n_unb = 0;               // unbounded objects: the first n_unb entries of the Morton-sorted tables
    uint32_t ngroups = 0;
    rtc_light light{};
    bool any_refl = false, any_refr = false;
};


extern "C" hipError_t rtc_launch_binning(const DevCamera *views, uint32_t nviews, uint32_t W, uint32_t H, uint32_t n, const DevBound *bound_s,
                                         const DevBound *gbound, const uint32_t *orig_s, uint32_t ngroups, uint32_t *cnt, uint32_t *list,
                                         uint32_t row0, uint32_t row_stride, hipStream_t stream, hipEvent_t e0, hipEvent_t e1,
                                         const DevIsect *isect_s, const uint32_t *kind_s, uint32_t n_unb, uint32_t *rows,
                                         const DevIsect *isect, DevPrim *prim);
enum { RTC_BIN_ROW_WORDS = 2 * RTC_MAX_VIEWS }; // a BinSet's tile_cnt buffer starts with the views' row words (RenderParams::tile_rows)
extern "C" hipError_t rtc_launch_light_lists(uint32_t n, uint32_t cap, const DevBound *bound, const double light[3], double reach, DevTileBundle *cells,
                                             DevTileBundle *macros, uint32_t *cnt, uint32_t *list, hipStream_t stream);
extern "C" hipError_t rtc_launch_undeal(const void *staging, void *canvas, uint32_t nranks, uint32_t nframes, uint32_t H,
                                        uint32_t rows_max, size_t row_bytes, hipStream_t stream);

#endif
