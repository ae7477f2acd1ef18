// rtc_api.cpp — [device] half of the C-ABI in include/rtc.h: context, HBM-resident World,
// render / color_at launches. Compiled by hipcc together with rtc_kernels.hip.
//
// There is deliberately no CPU fallback here: without a usable gfx950 device every entry
// point returns RTC_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "rtc.h"
#include "rtc_device.h"
#include "rtc_internal.h"

extern "C" hipError_t rtc_launch_trace(const RenderParams *P, int src, int refl, int refr, uint32_t nblocks,
                                       size_t lds_bytes, hipStream_t stream, hipEvent_t e0, hipEvent_t e1);
extern "C" hipError_t rtc_launch_prep(const DevIsect *isect, DevPrim *prim, uint32_t n, const double vinv[12],
                                      hipStream_t stream);
extern "C" hipError_t rtc_launch_arith(uint32_t op, const double *a, const double *b, uint32_t n, double *out,
                                       hipStream_t stream);

namespace {

#define HIP_TRY(expr)                                   \
    do {                                                \
        if ((expr) != hipSuccess) return RTC_ERR_DEVICE; \
    } while (0)

struct DeviceGuard { // make ctx->device current for the calling thread
    explicit DeviceGuard(int dev) { ok = hipSetDevice(dev) == hipSuccess; }
    bool ok;
};

// Pipelined context: wait for every lane (in-order contexts have none).
hipError_t drain_lanes(rtc_context *ctx) {
    for (uint32_t l = 0; l < rtc_context::MAX_LANES; ++l)
        if (ctx->lane[l]) {
            const hipError_t e = hipStreamSynchronize(ctx->lane[l]);
            if (e != hipSuccess) return e;
        }
    return hipSuccess;
}

void choose_source(const rtc_context *ctx, uint32_t n, uint32_t flags, int *src, uint32_t *tile_cap, size_t *lds_bytes) {
    // per object in LDS: 96 B inverse rows + 32 B primary prologue + 4 B kind
    const uint32_t per_obj = 96 + 32 + 4;
    int s;
    if (ctx->force_src >= 0) s = ctx->force_src;
    else if (!(flags & RTC_FLAG_NO_CULL)) s = (n > 256) ? SRC_CULL2 : SRC_CULL; // default: per-wave conservative cull,
                                                                              // two-level above 4 groups of 64
    else if (flags & RTC_FLAG_LDS_TABLE) s = SRC_LDS1; // brute force over the LDS-staged object table (LDS tiles when it does not fit)
    else if (n <= 128) s = SRC_SMEM;
    else if (n <= 448) s = SRC_LDS1;
    else s = SRC_LDSN;
    uint32_t cap = 0;
    if (s == SRC_LDS1) {
        if ((size_t)n * per_obj > 150 * 1024) s = SRC_LDSN;
        else cap = n ? n : 1;
    }
    if (s == SRC_LDSN) cap = ctx->tile_cap;
    if (s == SRC_SMEM || s == SRC_CULL || s == SRC_CULL2) cap = 0;
    *src = s;
    *tile_cap = cap;
    // kinds sit behind cap*16 doubles; round the block up to 16 bytes
    *lds_bytes = cap ? (((size_t)cap * per_obj + 15) & ~(size_t)15) : 0;
}

// 63-bit Morton key of a point inside the box [lo, hi]^3 (21 bits per axis).
uint64_t spread21(uint64_t v) {
    v &= 0x1fffffULL;
    v = (v | (v << 32)) & 0x1f00000000ffffULL;
    v = (v | (v << 16)) & 0x1f0000ff0000ffULL;
    v = (v | (v << 8)) & 0x100f00f00f00f00fULL;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ULL;
    v = (v | (v << 2)) & 0x1249249249249249ULL;
    return v;
}
uint64_t morton_key(const DevBound &b, const double lo[3], const double hi[3]) {
    uint64_t k = 0;
    const double c[3] = {b.cx, b.cy, b.cz};
    for (int a = 0; a < 3; ++a) {
        const double ext = hi[a] - lo[a];
        double u = ext > 0. ? (c[a] - lo[a]) / ext : 0.;
        if (!(u >= 0.)) u = 0.;
        if (u > 1.) u = 1.;
        k |= spread21((uint64_t)(u * 2097151.0)) << a;
    }
    return k;
}

// World-space bounding sphere of a shape, derived from the stored inverse transform only (that is
// what the kernels intersect with): the surface is { F p : p on the unit sphere / cube } with
// F = inv^-1 (affine part). Conservative: radius = largest singular value of F's 3x3 (spheres) or
// the farthest transformed corner (cubes), inflated by 1e-6; anything not clearly well-conditioned
// gets r = +inf and is simply never culled.
DevBound bound_of(const rtc_shape &s) {
    DevBound b{0., 0., 0., INFINITY, 0., 0.};
    if (s.kind == RTC_PLANE) return b;
    const double *m = s.inv;
    const double a[3][3] = {{m[0], m[1], m[2]}, {m[4], m[5], m[6]}, {m[8], m[9], m[10]}};
    const double t[3] = {m[3], m[7], m[11]};
    for (int i = 0; i < 3; ++i) {
        if (!std::isfinite(t[i])) return b;
        for (int j = 0; j < 3; ++j)
            if (!std::isfinite(a[i][j])) return b;
    }
    const double det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                       a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
    if (!(std::fabs(det) > 1e-300) || !std::isfinite(det)) return b;
    double f[3][3]; // F3 = a^-1 by cofactors
    f[0][0] = (a[1][1] * a[2][2] - a[1][2] * a[2][1]) / det;
    f[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) / det;
    f[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) / det;
    f[1][0] = (a[1][2] * a[2][0] - a[1][0] * a[2][2]) / det;
    f[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) / det;
    f[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) / det;
    f[2][0] = (a[1][0] * a[2][1] - a[1][1] * a[2][0]) / det;
    f[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) / det;
    f[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) / det;
    double resid = 0., fmaxabs = 0.; // || a F - I ||_max: refuse to trust an ill-conditioned inverse
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double v = (i == j) ? -1. : 0.;
            for (int k = 0; k < 3; ++k) v += a[i][k] * f[k][j];
            resid = std::fmax(resid, std::fabs(v));
            fmaxabs = std::fmax(fmaxabs, std::fabs(f[i][j]));
            if (!std::isfinite(f[i][j])) return b;
        }
    if (!(resid < 1e-9)) return b;
    double c[3];
    for (int i = 0; i < 3; ++i) c[i] = -(f[i][0] * t[0] + f[i][1] * t[1] + f[i][2] * t[2]);
    double r2;
    if (s.kind == RTC_SPHERE) {
        // lambda_max of S = F F^T by cyclic Jacobi
        double S[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) S[i][j] = f[i][0] * f[j][0] + f[i][1] * f[j][1] + f[i][2] * f[j][2];
        for (int sweep = 0; sweep < 30; ++sweep) {
            const double off = std::fabs(S[0][1]) + std::fabs(S[0][2]) + std::fabs(S[1][2]);
            if (off < 1e-300) break;
            for (int p = 0; p < 2; ++p)
                for (int q = p + 1; q < 3; ++q) {
                    if (S[p][q] == 0.) continue;
                    const double th = (S[q][q] - S[p][p]) / (2. * S[p][q]);
                    const double tt = (th >= 0. ? 1. : -1.) / (std::fabs(th) + std::sqrt(th * th + 1.));
                    const double cs = 1. / std::sqrt(tt * tt + 1.), sn = tt * cs;
                    for (int k = 0; k < 3; ++k) { // S <- S J
                        const double skp = S[k][p], skq = S[k][q];
                        S[k][p] = cs * skp - sn * skq;
                        S[k][q] = sn * skp + cs * skq;
                    }
                    for (int k = 0; k < 3; ++k) { // S <- J^T S
                        const double spk = S[p][k], sqk = S[q][k];
                        S[p][k] = cs * spk - sn * sqk;
                        S[q][k] = sn * spk + cs * sqk;
                    }
                }
        }
        // Gershgorin guard on whatever off-diagonal mass is left
        const double g = std::fabs(S[0][1]) + std::fabs(S[0][2]) + std::fabs(S[1][2]);
        r2 = std::fmax(S[0][0], std::fmax(S[1][1], S[2][2])) + 2. * g;
    } else { // cube: farthest of the 8 transformed corners
        r2 = 0.;
        for (int k = 0; k < 8; ++k) {
            const double px = (k & 1) ? 1. : -1., py = (k & 2) ? 1. : -1., pz = (k & 4) ? 1. : -1.;
            double q = 0.;
            for (int i = 0; i < 3; ++i) {
                const double v = f[i][0] * px + f[i][1] * py + f[i][2] * pz;
                q += v * v;
            }
            r2 = std::fmax(r2, q);
        }
    }
    if (!std::isfinite(r2) || !(r2 >= 0.)) return b;
    const double cn = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    const double r = std::sqrt(r2) * (1. + 1e-6) + 1e-9 * (1. + cn) + 1e-7 * fmaxabs;
    if (!std::isfinite(r) || !std::isfinite(cn)) return b;
    double na2 = 0.; // ||A||_F^2 >= ||A||_2^2
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) na2 += a[i][j] * a[i][j];
    const double k = 0.75e-14 * na2;
    if (!std::isfinite(k)) return b;
    b.cx = c[0]; b.cy = c[1]; b.cz = c[2]; b.r = r;
    b.k = k;
    b.cn = cn;
    return b;
}

void fill_camera(RenderParams &P, const rtc_camera *cam, uint32_t view = 0) {
    if (view == 0) {
        P.W = cam->hsize;
        P.H = cam->vsize;
        // antialiasing_samples == 1 is the one-ray branch; every other value (0 included) the sub-sample
        // branch (camera.rs:96-99)
        P.samples = cam->samples == 1u ? 1u : 4u;
    }
    DevCamera &c = P.views[view];
    c.half_width = cam->half_width;
    c.half_height = cam->half_height;
    c.pixel_size = cam->pixel_size;
    std::memcpy(c.vinv, cam->view_inv, sizeof(double) * 12);
}

void fill_world(RenderParams &P, const rtc_world *w) {
    P.isect = w->d_isect;
    P.kind = w->d_kind;
    P.shade = w->d_shade;
    P.prim = w->d_prim;
    P.bound = w->d_bound;
    P.isect_s = w->d_isect_s;
    P.kind_s = w->d_kind_s;
    P.bound_s = w->d_bound_s;
    P.orig_s = w->d_orig_s;
    P.gbound = w->d_gbound;
    P.idtab = w->d_idtab;
    P.pre = w->d_pre;
    P.pre_s = w->d_pre_s;
    P.pre_limit = w->pre_limit;
    P.light_cnt = w->d_light_cnt;
    P.light_list = w->d_light_list;
    P.light_reach = w->light_reach;
    P.light_cap = w->light_cap;
    P.n_unb = w->n_unb;
    P.ngroups = w->ngroups;
    P.n = w->n;
    for (int i = 0; i < 3; ++i) {
        P.light_pos[i] = w->light.position[i];
        P.light_int[i] = w->light.intensity[i];
    }
}

} // namespace

extern "C" {

rtc_status rtc_context_create(int32_t device, void *stream, rtc_context **out) {
    if (!out) return RTC_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return RTC_ERR_DEVICE;
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return RTC_ERR_DEVICE; // kernels are gfx950-only
    rtc_context *ctx = new (std::nothrow) rtc_context;
    if (!ctx) return RTC_ERR_NOMEM;
    ctx->device = device;
    ctx->stream = static_cast<hipStream_t>(stream); // NULL = the device's default stream

    if (hipMalloc(&ctx->d_counters, sizeof(unsigned long long) * CNT_N * CNT_SLOTS) != hipSuccess ||
        hipMemsetAsync(ctx->d_counters, 0, sizeof(unsigned long long) * CNT_N * CNT_SLOTS, ctx->stream) != hipSuccess) {
        rtc_context_destroy(ctx);
        return RTC_ERR_DEVICE;
    }
    if (const char *e = std::getenv("RTC_SRC")) {
        const int v = std::atoi(e);
        if (v >= 0 && v <= 4) ctx->force_src = v;
    }
    if (const char *e = std::getenv("RTC_BINNING")) ctx->binning = std::atoi(e) != 0;
    if (const char *e = std::getenv("RTC_LIGHT_LISTS")) ctx->light_lists = std::atoi(e) != 0;
    if (const char *e = std::getenv("RTC_SKY_ROWS")) ctx->sky_rows = std::atoi(e) != 0;
    if (const char *e = std::getenv("RTC_BIN_SMALL_PIXELS")) ctx->bin_small_pixels = std::strtoull(e, nullptr, 10);
    if (const char *e = std::getenv("RTC_BIN_SMALL_PIXELS_PIPELINED")) ctx->bin_small_pixels_pipelined = std::strtoull(e, nullptr, 10);
    if (const char *e = std::getenv("RTC_TILES_PER_WG")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= 16) ctx->tiles_per_wg = (uint32_t)v;
    }
    if (const char *e = std::getenv("RTC_TILES_GUIDED")) { // tenths of `slots` tiles per chunk level (RenderParams::chunk_wgs); 0 = off
        const long v = std::strtol(e, nullptr, 10);
        if (v >= 0 && v <= 1000) ctx->tiles_guided_tenths = (uint32_t)v;
    }
    if (const char *e = std::getenv("RTC_TILES_SLOTS")) { // tests: pretend this many workgroups are resident, so that small launches get every chunk level
        const long v = std::strtol(e, nullptr, 10);
        if (v >= 0 && v <= 1000000) ctx->tiles_slots = (uint32_t)v;
    }
    if (const char *e = std::getenv("RTC_TILES_KMAX")) {
        const long v = std::strtol(e, nullptr, 10);
        if (v >= 1 && v <= 8) ctx->tiles_kmax = (uint32_t)v;
    }
    if (const char *e = std::getenv("RTC_TILE_CAP")) {
        const int v = std::atoi(e);
        if (v >= 16 && v <= 1024) ctx->tile_cap = (uint32_t)v;
    }
    *out = ctx;
    return RTC_OK;
}

void rtc_context_destroy(rtc_context *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)drain_lanes(ctx);
    for (hipStream_t &l : ctx->lane)
        if (l) { (void)hipStreamDestroy(l); l = nullptr; }
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->d_canvas) (void)hipFree(ctx->d_canvas);
    if (ctx->d_canvas8) (void)hipFree(ctx->d_canvas8);
    if (ctx->side_stream) { (void)hipStreamSynchronize(ctx->side_stream); (void)hipStreamDestroy(ctx->side_stream); }
    if (ctx->fence_ev) (void)hipEventDestroy(ctx->fence_ev);
    for (auto &pair : ctx->ev)
        for (hipEvent_t e : pair)
            if (e) (void)hipEventDestroy(e);
    for (auto &pair : ctx->ev_bin)
        for (hipEvent_t e : pair)
            if (e) (void)hipEventDestroy(e);
    delete ctx;
}

rtc_status rtc_context_synchronize(rtc_context *ctx) {
    if (!ctx) return RTC_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(drain_lanes(ctx));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RTC_OK;
}

rtc_status rtc_context_set_pipeline(rtc_context *ctx, uint32_t depth) {
    if (!ctx || depth == 0 || depth > rtc_context::MAX_LANES) return RTC_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(drain_lanes(ctx));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->side_stream) HIP_TRY(hipStreamSynchronize(ctx->side_stream));
    if (depth > 1)
        for (uint32_t l = 0; l < depth; ++l)
            if (!ctx->lane[l]) HIP_TRY(hipStreamCreateWithFlags(&ctx->lane[l], hipStreamNonBlocking));
    ctx->lanes = depth;
    ctx->lane_next = 0;
    return RTC_OK;
}

rtc_status rtc_context_fence(rtc_context *ctx) {
    if (!ctx) return RTC_ERR_ARG;
    if (ctx->lanes <= 1) return RTC_OK; // in order on the stream already
    HIP_TRY(hipSetDevice(ctx->device));
    if (!ctx->fence_ev) HIP_TRY(hipEventCreateWithFlags(&ctx->fence_ev, hipEventDisableTiming));
    for (uint32_t l = 0; l < rtc_context::MAX_LANES; ++l)
        if (ctx->lane[l]) {
            HIP_TRY(hipEventRecord(ctx->fence_ev, ctx->lane[l]));
            HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->fence_ev, 0)); // (the wait captures the record made just above)
        }
    return RTC_OK;
}

rtc_status rtc_context_last_launch_info(rtc_context *ctx, rtc_launch_info *out) {
    if (!ctx || !out) return RTC_ERR_ARG;
    if (ctx->launches_total == 0) return RTC_ERR_ARG; // nothing launched yet
    *out = ctx->last;
    return RTC_OK;
}

rtc_status rtc_context_device_info(rtc_context *ctx, char *name, size_t cap, int32_t *compute_units, int32_t *clock_mhz) {
    if (!ctx) return RTC_ERR_ARG;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
    if (name && cap) {
        std::strncpy(name, prop.name, cap - 1);
        name[cap - 1] = 0;
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (clock_mhz) *clock_mhz = prop.clockRate / 1000;
    return RTC_OK;
}

rtc_status rtc_world_create(rtc_context *ctx, const rtc_shape *shapes, uint32_t n, const rtc_light *light, rtc_world **out) {
    if (!ctx || !out || !light || (n && !shapes)) return RTC_ERR_ARG;
    *out = nullptr;
    // Material::lighting panics when a material has neither colour nor pattern (material.rs:328-331)
    for (uint32_t i = 0; i < n; ++i) {
        if (shapes[i].kind > RTC_CUBE || shapes[i].material.pattern_kind > RTC_PATTERN_GRID) return RTC_ERR_ARG;
        if (shapes[i].material.pattern_kind == RTC_PATTERN_NONE && !shapes[i].material.has_color) return RTC_ERR_NO_COLOR;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t na = n ? n : 1;
    std::vector<DevIsect> isect(na);
    std::vector<uint32_t> kind(na, 0);
    std::vector<DevShade> shade(na);
    std::vector<DevBound> bound(na);
    std::memset(isect.data(), 0, sizeof(DevIsect) * na);
    std::memset(shade.data(), 0, sizeof(DevShade) * na);
    bool any_refl = false, any_refr = false;
    // World::add_shape numbers the shapes last_world_id + 1 (shape.rs:661-667). A caller that leaves every id
    // 0 (rtc_shape_init does) gets exactly that numbering; ids that were given are honoured as they are —
    // equal ids are ONE container to compute_refractive (shape.rs:127), which is also what the reference's
    // own u8 ids do beyond 255 shapes.
    bool all_zero = true;
    for (uint32_t i = 0; i < n; ++i) all_zero = all_zero && shapes[i].world_id == 0u;
    std::vector<DevIdEntry> idtab(na, DevIdEntry{0u, 0u});
    for (uint32_t i = 0; i < n; ++i) idtab[i] = DevIdEntry{i, all_zero ? i + 1u : shapes[i].world_id};
    std::stable_sort(idtab.begin(), idtab.begin() + n, [](const DevIdEntry &a, const DevIdEntry &b) { return a.id < b.id; });
    for (uint32_t i = 0; i < n; ++i) {
        const rtc_shape &s = shapes[i];
        const rtc_material &m = s.material;
        std::memcpy(isect[i].m, s.inv, sizeof(double) * 12);
        kind[i] = s.kind;
        DevShade &d = shade[i];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                // Shape::normal_at uses transform_transpose (shape.rs:38); Cube::normal_at
                // transposes its inverse on the fly (shape.rs:627)
                d.nt[r * 3 + c] = (s.kind == RTC_CUBE) ? s.inv[c * 4 + r] : s.inv_t[r * 4 + c];
        for (int c = 0; c < 3; ++c) {
            d.color[c] = m.color[c];
            d.pat_a[c] = m.pat_a[c];
            d.pat_b[c] = m.pat_b[c];
        }
        d.ambient = m.ambient;
        d.diffuse = m.diffuse;
        d.specular = m.specular;
        d.shininess = m.shininess;
        d.reflective = m.reflective;
        d.transparency = m.transparency;
        d.refractive_index = m.refractive_index;
        std::memcpy(d.pat_inv, m.pat_inv, sizeof(double) * 12);
        if (s.kind == RTC_PLANE) {
            // world_normal = local_normal.transform(inverse_transpose).normalize() with local_normal
            // (0,1,0) (shape.rs:37-39, 481-483; transform.rs:114-117; vec.rs:65-76): mul/add/sqrt/div
            // in the reference's order, correctly rounded on the host exactly as on the device
            // (this file is compiled with -ffp-contract=off)
            const double *t = d.nt;
            const double wx = t[0] * 0. + t[1] * 1. + t[2] * 0.;
            const double wy = t[3] * 0. + t[4] * 1. + t[5] * 0.;
            const double wz = t[6] * 0. + t[7] * 1. + t[8] * 0.;
            const double mag = std::sqrt(wx * wx + wy * wy + wz * wz);
            d.plane_n[0] = wx / mag;
            d.plane_n[1] = wy / mag;
            d.plane_n[2] = wz / mag;
        }
        d.kind = s.kind;
        d.pattern_kind = m.pattern_kind;
        d.world_id = all_zero ? i + 1u : s.world_id;
        if (m.reflective > 0.) any_refl = true;     // reflected_color shape.rs:730
        if (m.transparency != 0.0) any_refr = true; // refracted_color shape.rs:752
        bound[i] = bound_of(s);
    }
    // ---- two-level cull tables: unbounded objects first, the rest in Morton order of their centres;
    // groups of 64 consecutive entries get a sphere around their members (inf if any is unbounded)
    std::vector<uint32_t> order(na, 0);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    {
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = 0; i < n; ++i)
            if (std::isfinite(bound[i].r)) {
                const double c[3] = {bound[i].cx, bound[i].cy, bound[i].cz};
                for (int a = 0; a < 3; ++a) { lo[a] = std::fmin(lo[a], c[a]); hi[a] = std::fmax(hi[a], c[a]); }
            }
        std::vector<uint64_t> key(na, 0);
        for (uint32_t i = 0; i < n; ++i) key[i] = std::isfinite(bound[i].r) ? (1ULL << 63) | morton_key(bound[i], lo, hi) : 0ULL;
        std::stable_sort(order.begin(), order.begin() + n, [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
    }
    const uint32_t ngroups = (n + 63u) / 64u;
    std::vector<DevIsect> isect_s(na);
    std::vector<uint32_t> kind_s(na, 0), orig_s(na, 0);
    std::vector<DevBound> bound_s(na), gbound(ngroups ? ngroups : 1);
    std::memset(isect_s.data(), 0, sizeof(DevIsect) * na);
    for (uint32_t i = 0; i < na; ++i) bound_s[i] = DevBound{0., 0., 0., INFINITY, 0., 0.};
    for (uint32_t i = 0; i < n; ++i) {
        isect_s[i] = isect[order[i]];
        kind_s[i] = kind[order[i]];
        bound_s[i] = bound[order[i]];
        orig_s[i] = order[i];
    }
    gbound[0] = DevBound{0., 0., 0., INFINITY, 0., 0.};
    for (uint32_t g = 0; g < ngroups; ++g) {
        const uint32_t a = g * 64u, b = (a + 64u < n) ? a + 64u : n;
        DevBound gb{0., 0., 0., INFINITY, 0., 0.};
        bool finite = true;
        double cx = 0., cy = 0., cz = 0., kmax = 0., cnmax = 0.;
        for (uint32_t i = a; i < b; ++i) {
            if (!std::isfinite(bound_s[i].r)) { finite = false; break; }
            cx += bound_s[i].cx; cy += bound_s[i].cy; cz += bound_s[i].cz;
            kmax = std::fmax(kmax, bound_s[i].k);
            cnmax = std::fmax(cnmax, bound_s[i].cn);
        }
        if (finite && b > a) {
            const double cnt = (double)(b - a);
            cx /= cnt; cy /= cnt; cz /= cnt;
            double r = 0.;
            for (uint32_t i = a; i < b; ++i) {
                const double dx = bound_s[i].cx - cx, dy = bound_s[i].cy - cy, dz = bound_s[i].cz - cz;
                r = std::fmax(r, std::sqrt(dx * dx + dy * dy + dz * dz) + bound_s[i].r);
            }
            r = r * (1. + 1e-9) + 1e-12;
            // rounding inflation of the group: members' radii inflate by at most
            // r_i*k_i*D_i*(cn_i + D_i) with D_i <= D_group + r_group; folded into the group's k/cn
            // conservatively: k = max k_i, cn = max cn_i + r (so that D_group + cn covers D_i + cn_i)
            if (std::isfinite(r) && std::isfinite(cx) && std::isfinite(cy) && std::isfinite(cz))
                gb = DevBound{cx, cy, cz, r, kmax * 4., cnmax + 2. * r};
        }
        gbound[g] = gb;
    }

    // per-lane prefilter records (DevPre, rtc_device.h): pre-inflated for every ray origin within 64x the extent of the bounded
    // objects around the world origin (secondary rays start on surfaces; a floor hit near the horizon can be farther: such a pass
    // takes the general test). The inflation is relative ~k*D^2 (1e-13 * 1e4..1e8 for ordinary scenes): generous limits cost nothing.
    double extent = 1.;
    for (uint32_t i = 0; i < n; ++i)
        if (std::isfinite(bound[i].r)) extent = std::fmax(extent, std::fabs(bound[i].cx) + std::fabs(bound[i].cy) + std::fabs(bound[i].cz) + bound[i].r);
    const double pre_limit = 64. * extent;
    auto pre_of = [&](const DevBound &b) {
        DevPre q{b.cx, b.cy, b.cz, INFINITY};
        if (std::isfinite(b.r) && std::isfinite(pre_limit)) {
            const double Dw = (std::fabs(b.cx) + std::fabs(b.cy) + std::fabs(b.cz) + pre_limit) * (1. + 1e-12);
            const double R = ((b.r + b.r * (b.k * Dw * (b.cn + Dw))) * 1.000001 + 1e-12) * (1. + 1e-12);
            const double R2 = R * R * (1. + 1e-12);
            if (std::isfinite(R2)) q.R2 = R2;
        }
        return q;
    };
    std::vector<DevPre> pre(na), pre_s(na);
    for (uint32_t i = 0; i < na; ++i) pre[i] = pre_s[i] = DevPre{0., 0., 0., INFINITY};
    for (uint32_t i = 0; i < n; ++i) { pre[i] = pre_of(bound[i]); pre_s[i] = pre_of(bound_s[i]); }

    rtc_world *w = new (std::nothrow) rtc_world;
    if (!w) return RTC_ERR_NOMEM;
    w->ctx = ctx;
    w->pre_limit = std::isfinite(pre_limit) ? pre_limit : 0.;
    w->ngroups = ngroups;
    w->device = ctx->device;
    w->n = n;
    w->light = *light;
    w->any_refl = any_refl;
    w->any_refr = any_refr;
    bool ok = hipMalloc(&w->d_isect, sizeof(DevIsect) * na) == hipSuccess &&
              hipMalloc(&w->d_kind, sizeof(uint32_t) * na) == hipSuccess &&
              hipMalloc(&w->d_shade, sizeof(DevShade) * na) == hipSuccess &&
              hipMalloc(&w->d_prim, sizeof(DevPrim) * na) == hipSuccess &&
              hipMalloc(&w->d_bound, sizeof(DevBound) * na) == hipSuccess &&
              hipMalloc(&w->d_isect_s, sizeof(DevIsect) * na) == hipSuccess &&
              hipMalloc(&w->d_kind_s, sizeof(uint32_t) * na) == hipSuccess &&
              hipMalloc(&w->d_bound_s, sizeof(DevBound) * na) == hipSuccess &&
              hipMalloc(&w->d_orig_s, sizeof(uint32_t) * na) == hipSuccess &&
              hipMalloc(&w->d_gbound, sizeof(DevBound) * gbound.size()) == hipSuccess &&
              hipMalloc(&w->d_idtab, sizeof(DevIdEntry) * na) == hipSuccess &&
              hipMalloc(&w->d_pre, sizeof(DevPre) * na) == hipSuccess &&
              hipMalloc(&w->d_pre_s, sizeof(DevPre) * na) == hipSuccess;
    for (uint32_t i = 0; i < n; ++i)
        if (!std::isfinite(bound_s[i].r)) w->n_unb = i + 1u; // unbounded objects sort first (key 0)
    ok = ok && hipMemcpy(w->d_isect, isect.data(), sizeof(DevIsect) * na, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(w->d_kind, kind.data(), sizeof(uint32_t) * na, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(w->d_shade, shade.data(), sizeof(DevShade) * na, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(w->d_bound, bound.data(), sizeof(DevBound) * na, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(w->d_isect_s, isect_s.data(), sizeof(DevIsect) * na, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(w->d_kind_s, kind_s.data(), sizeof(uint32_t) * na, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(w->d_bound_s, bound_s.data(), sizeof(DevBound) * na, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(w->d_orig_s, orig_s.data(), sizeof(uint32_t) * na, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(w->d_gbound, gbound.data(), sizeof(DevBound) * gbound.size(), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(w->d_idtab, idtab.data(), sizeof(DevIdEntry) * na, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(w->d_pre, pre.data(), sizeof(DevPre) * na, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(w->d_pre_s, pre_s.data(), sizeof(DevPre) * na, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemset(w->d_prim, 0, sizeof(DevPrim) * na) == hipSuccess;
    // light-space shadow lists: every shadow segment ends at the light, so the objects a segment can meet
    // are listed per direction cell of a cube map around the light, once per World. Reach = twice the far side of the
    // farthest bounded object as seen from the light (longer segments fall back to the group walk).
    if (ok && n >= 32) { // (a handful of objects: one cull step is cheaper than finding the cells — Criterion scene 33.5 vs 37.8 us)
        double far = 0.;
        for (uint32_t i = 0; i < n; ++i)
            if (std::isfinite(bound[i].r)) {
                const double dx = bound[i].cx - light->position[0], dy = bound[i].cy - light->position[1], dz = bound[i].cz - light->position[2];
                far = std::fmax(far, std::sqrt(dx * dx + dy * dy + dz * dz) + bound[i].r);
            }
        const double reach = 2. * far;
        if (std::isfinite(reach) && reach > 0. && reach < 1e30 && std::isfinite(light->position[0]) && std::isfinite(light->position[1]) &&
            std::isfinite(light->position[2])) {
            const uint32_t cap = n > 256 ? RTC_LIGHT_LIST_CAP : RTC_LIGHT_LIST_CAP_SMALL;
            w->light_cap = cap;
            const size_t cells = 6u * (size_t)RTC_LIGHT_R * RTC_LIGHT_R, macros = 6u * (size_t)(RTC_LIGHT_R / 8u) * (RTC_LIGHT_R / 8u);
            // the lists are an optimisation (the shadow pass walks without them): a failed allocation must not fail the upload
            const bool got = hipMalloc(&w->d_light_cells, sizeof(DevTileBundle) * (cells + macros)) == hipSuccess &&
                             hipMalloc(&w->d_light_cnt, sizeof(uint32_t) * cells) == hipSuccess &&
                             hipMalloc(&w->d_light_list, sizeof(uint32_t) * cells * cap) == hipSuccess;
            if (got) {
                ok = rtc_launch_light_lists(n, cap, w->d_bound, light->position, reach, w->d_light_cells, w->d_light_cells + cells, w->d_light_cnt,
                                            w->d_light_list, ctx->stream) == hipSuccess &&
                     hipStreamSynchronize(ctx->stream) == hipSuccess;
                w->light_reach = reach;
            } else {
                (void)hipGetLastError();
                if (w->d_light_cells) (void)hipFree(w->d_light_cells);
                if (w->d_light_cnt) (void)hipFree(w->d_light_cnt);
                if (w->d_light_list) (void)hipFree(w->d_light_list);
                w->d_light_cells = nullptr; w->d_light_cnt = nullptr; w->d_light_list = nullptr;
                w->light_cap = 0;
            }
        }
    }
    if (!ok) {
        const hipError_t e = hipGetLastError();
        rtc_world_destroy(w);
        return e == hipErrorOutOfMemory ? RTC_ERR_NOMEM : RTC_ERR_DEVICE;
    }
    *out = w;
    return RTC_OK;
}

void rtc_world_destroy(rtc_world *w) {
    if (!w) return;
    if (w->device >= 0) { // the context may already be gone: synchronise the device, not its stream
        (void)hipSetDevice(w->device);
        (void)hipDeviceSynchronize();
    }
    if (w->d_isect) (void)hipFree(w->d_isect);
    if (w->d_kind) (void)hipFree(w->d_kind);
    if (w->d_shade) (void)hipFree(w->d_shade);
    if (w->d_prim) (void)hipFree(w->d_prim);
    if (w->d_bound) (void)hipFree(w->d_bound);
    if (w->d_isect_s) (void)hipFree(w->d_isect_s);
    if (w->d_kind_s) (void)hipFree(w->d_kind_s);
    if (w->d_bound_s) (void)hipFree(w->d_bound_s);
    if (w->d_orig_s) (void)hipFree(w->d_orig_s);
    if (w->d_gbound) (void)hipFree(w->d_gbound);
    if (w->d_idtab) (void)hipFree(w->d_idtab);
    if (w->d_pre) (void)hipFree(w->d_pre);
    if (w->d_pre_s) (void)hipFree(w->d_pre_s);
    for (rtc_world::BinSet &b : w->bin) {
        if (b.tile_cnt) (void)hipFree(b.tile_cnt);
        if (b.tile_list) (void)hipFree(b.tile_list);
        if (b.prim) (void)hipFree(b.prim);
        if (b.binned) (void)hipEventDestroy(b.binned);
        if (b.traced) (void)hipEventDestroy(b.traced);
    }
    if (w->d_light_cells) (void)hipFree(w->d_light_cells);
    if (w->d_light_cnt) (void)hipFree(w->d_light_cnt);
    if (w->d_light_list) (void)hipFree(w->d_light_list);
    delete w;
}

// rows [y0, y1) in tile rows of 8, tile row k at image rows y0 + 8*k*band_stride; grid_y tile rows
static rtc_status render_launch(rtc_context *ctx, const rtc_world *w, const rtc_camera *cam, uint32_t mode, uint32_t y0,
                                uint32_t y1, uint32_t band_stride, uint32_t grid_y, void *d_rgb, void *d_rgb8,
                                uint32_t flags, uint32_t nviews = 1, uint32_t view_rows = 0) {
    HIP_TRY(hipSetDevice(ctx->device));
    RenderParams P;
    std::memset(&P, 0, sizeof P);
    fill_world(P, w);
    if (!ctx->light_lists) P.light_cnt = nullptr;
    for (uint32_t v = 0; v < nviews; ++v) fill_camera(P, cam + v, v);
    P.nviews = nviews;
    P.view_rows = view_rows;
    P.y0 = y0;
    P.y1 = y1;
    P.mode = mode;
    P.out = static_cast<double *>(d_rgb);
    P.out8 = static_cast<unsigned char *>(d_rgb8);
    P.counters = ctx->d_counters;
    P.rays = nullptr;
    P.remaining = RTC_MAX_REFLECTIONS; // render_pixel passes Camera::MAX_REFLECTIONS camera.rs:98
    int src;
    size_t lds_bytes;
    choose_source(ctx, w->n, flags, &src, &P.tile_cap, &lds_bytes);
    const int cull = src == SRC_CULL2 ? 2 : src == SRC_CULL ? 1 : 0; // RTC_BLOCK_FOR's cull level
    const bool refl = w->any_refl || w->any_refr;
    const uint32_t block = RTC_BLOCK_FOR(cull, refl, w->any_refr, false), tile_w = RTC_TILE_W_FOR(cull, refl, w->any_refr, false);
    P.grid_x = (cam->hsize + tile_w - 1u) / tile_w;
    P.grid_y = grid_y;
    P.band_stride = band_stride;
    P.flags = flags;
    if (P.samples != 1u) { // the 4 sub-samples of every pixel wait in LDS for the resample test (camera.rs:108)
        P.aa_lds_off = (uint32_t)lds_bytes;
        lds_bytes += (size_t)block * 15u * sizeof(double); // + the running sums
        // Camera::resample traces `antialiasing_samples` more rays (camera.rs:87); u8 in the reference
        P.resample_n = (flags & RTC_FLAG_AA_RESAMPLE) ? (cam->samples & 0xffu) : 0u;
    }
    // start/stop events cost ~9 us of host time and ~5 us of GPU time per launch (measured): callers
    // that are launch-bound sample every n-th launch instead (rtc_context_set_timing)
    const bool timed = ctx->time_every != 0 && ctx->launches % ctx->time_every == 0;
    const uint32_t slot = (uint32_t)(ctx->timed % rtc_context::EV_RING);
    if (timed && slot >= ctx->ev_created) { // next chunk of the ring
        const uint32_t upto = std::min<uint32_t>(rtc_context::EV_RING, ctx->ev_created + rtc_context::EV_CHUNK);
        for (uint32_t k = ctx->ev_created; k < upto; ++k) {
            HIP_TRY(hipEventCreate(&ctx->ev[k][0]));
            HIP_TRY(hipEventCreate(&ctx->ev[k][1]));
            HIP_TRY(hipEventCreate(&ctx->ev_bin[k][0]));
            HIP_TRY(hipEventCreate(&ctx->ev_bin[k][1]));
            ctx->ev_created = k + 1;
        }
    }
    hipEvent_t *pair = ctx->ev[slot], *pair_bin = ctx->ev_bin[slot];
    if (timed) ctx->bin_timed[slot] = false;
    // Which stream. A pipelined context deals the launches round-robin over its lanes; the brute-force variants share one
    // per-render table (w->d_prim) and stay in order on lane 0.
    const bool piped = ctx->lanes > 1;
    const uint32_t lane = piped ? ((src == SRC_CULL || src == SRC_CULL2) ? (uint32_t)(ctx->lane_next++ % ctx->lanes) : 0u) : 0u;
    hipStream_t stream = piped ? ctx->lane[lane] : ctx->stream;
    // the part of the frame this launch renders, in pixels (a rank's bands: its share)
    const unsigned long long launch_pixels = (unsigned long long)nviews * cam->hsize * std::min<unsigned long long>((unsigned long long)grid_y * 8u, cam->vsize);
    // binned primary pass (tile rows aligned with the image's): one small kernel puts every object on the list of each 8x8
    // tile its bounding sphere can touch (same conservative predicate as the wave-level cull), so the render kernel's primary
    // pass runs exact tests on a short list instead of walking the groups. Two-level worlds always; one-level worlds when the
    // launch is long enough for the extra kernel (and, in order on one stream, its two cross-stream events) to pay:
    // bin_small_pixels counts the pixels THIS launch renders — whole frames or one rank's bands (k_bin_tiles lists only the
    // tile rows the launch renders).
    const bool bin_this = (src == SRC_CULL2) ||
                          (src == SRC_CULL && (piped ? launch_pixels >= ctx->bin_small_pixels_pipelined
                                                     : launch_pixels >= ctx->bin_small_pixels));
    rtc_world::BinSet *binset = nullptr;
    bool bin_ok = bin_this && ctx->binning && (y0 % 8u) == 0u && w->n != 0u;
    const uint32_t tiles_x = (cam->hsize + 7u) / 8u, tiles_y = (cam->vsize + 7u) / 8u;
    const size_t tiles = (size_t)tiles_x * tiles_y * nviews;
    if (bin_ok && piped) {
        // lane-local lists: the binning kernel precedes the render kernel on the lane's own stream and runs beside the other
        // lanes' render kernels — no events. Sized for the largest launch seen (grow-only; growing waits for the lane).
        rtc_world::BinSet &B = w->bin[lane];
        if (B.tiles_cap < tiles) {
            (void)hipStreamSynchronize(stream);
            if (B.tile_cnt) (void)hipFree(B.tile_cnt);
            if (B.tile_list) (void)hipFree(B.tile_list);
            B.tile_cnt = nullptr; B.tile_list = nullptr; B.tiles_cap = 0;
            ++ctx->render_allocs;
            const bool got = hipMalloc(&B.tile_cnt, sizeof(uint32_t) * (tiles + RTC_BIN_ROW_WORDS)) == hipSuccess &&
                             (++ctx->render_allocs, hipMalloc(&B.tile_list, sizeof(uint32_t) * tiles * RTC_TILE_LIST_CAP) == hipSuccess);
            if (got) B.tiles_cap = tiles;
            else { // the lists are an optimisation: without memory for them the launch walks (same pixels)
                if (B.tile_cnt) (void)hipFree(B.tile_cnt);
                B.tile_cnt = nullptr; B.tile_list = nullptr;
                (void)hipGetLastError();
                bin_ok = false;
            }
        }
        if (bin_ok && B.prim_cap < (size_t)w->n * nviews) {
            (void)hipStreamSynchronize(stream);
            if (B.prim) (void)hipFree(B.prim);
            B.prim = nullptr; B.prim_cap = 0;
            ++ctx->render_allocs;
            if (hipMalloc(&B.prim, sizeof(DevPrim) * (size_t)w->n * nviews) == hipSuccess) B.prim_cap = (size_t)w->n * nviews;
            else { (void)hipGetLastError(); bin_ok = false; }
        }
        if (bin_ok) {
            HIP_TRY(rtc_launch_binning(P.views, nviews, cam->hsize, cam->vsize, w->n, w->d_bound_s, w->d_gbound, w->d_orig_s, w->ngroups,
                                       B.tile_cnt + RTC_BIN_ROW_WORDS, B.tile_list, y0 / 8u, band_stride, stream, timed ? pair_bin[0] : nullptr,
                                       timed ? pair_bin[1] : nullptr, w->d_isect_s, w->d_kind_s, w->n_unb, B.tile_cnt, w->d_isect, B.prim));
            if (timed) ctx->bin_timed[slot] = true;
            P.prim = B.prim;
            P.tile_rows = ctx->sky_rows ? B.tile_cnt : nullptr;
            P.tile_cnt = B.tile_cnt + RTC_BIN_ROW_WORDS;
            P.tile_list = B.tile_list;
            P.tiles_x = tiles_x;
            P.tiles_y = tiles_y;
            P.bin_packed = RTC_BIN_PACKED(w->n) ? 1u : 0u;
            P.n_unb = w->n_unb;
        }
    } else if (bin_ok) {
        if (!ctx->side_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
        // Capacity. Both sets are made ready by the FIRST binned launch, and for RTC_MAX_VIEWS views while that stays within
        // 128 MB per set (1080p: 67 MB; larger frames: exactly the launch's views, growing once if a later launch has more): a
        // launch sequence must not allocate after its first launch — hipMalloc / hipFree wait for the device, 0.2-3 ms in the
        // middle of a frame sequence (a 5-frame warm-up launch followed by 8-frame launches did exactly that: 0.09-0.22 ms per
        // frame instead of 0.07). The lists are an optimisation: when there is no memory for them the launch walks instead.
        const size_t per_view_bytes = (size_t)tiles_x * tiles_y * sizeof(uint32_t) * (1u + RTC_TILE_LIST_CAP);
        const uint32_t alloc_views = per_view_bytes * RTC_MAX_VIEWS <= ((size_t)128 << 20) ? (uint32_t)RTC_MAX_VIEWS : nviews;
        const size_t tiles_alloc = (size_t)tiles_x * tiles_y * alloc_views;
        for (uint32_t k = 0; k < 2u && bin_ok; ++k) {
            rtc_world::BinSet &S = w->bin[k];
            if (!S.binned) {
                HIP_TRY(hipEventCreateWithFlags(&S.binned, hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&S.traced, hipEventDisableTiming));
            }
            if (S.tiles_cap < tiles) { // (hipFree waits for the device: nothing reads the old buffers any more)
                if (S.tile_cnt) (void)hipFree(S.tile_cnt);
                if (S.tile_list) (void)hipFree(S.tile_list);
                S.tile_cnt = nullptr; S.tile_list = nullptr;
                S.tiles_cap = 0;
                ++ctx->render_allocs;
                const bool got = hipMalloc(&S.tile_cnt, sizeof(uint32_t) * (tiles_alloc + RTC_BIN_ROW_WORDS)) == hipSuccess &&
                                 (++ctx->render_allocs, hipMalloc(&S.tile_list, sizeof(uint32_t) * tiles_alloc * RTC_TILE_LIST_CAP) == hipSuccess);
                if (got) S.tiles_cap = tiles_alloc;
                else {
                    if (S.tile_cnt) (void)hipFree(S.tile_cnt);
                    S.tile_cnt = nullptr; S.tile_list = nullptr;
                    (void)hipGetLastError();
                    bin_ok = false;
                }
            }
            if (bin_ok && S.prim_cap < (size_t)w->n * nviews) {
                if (S.prim) (void)hipFree(S.prim);
                S.prim = nullptr; S.prim_cap = 0;
                ++ctx->render_allocs;
                const size_t want = (size_t)w->n * std::max(alloc_views, nviews);
                if (hipMalloc(&S.prim, sizeof(DevPrim) * want) == hipSuccess) S.prim_cap = want;
                else { (void)hipGetLastError(); bin_ok = false; }
            }
        }
    }
    if (bin_ok && !piped) {
        rtc_world::BinSet &B = w->bin[w->bin_next++ & 1u];
        // The binning depends on the World (resident since rtc_world_create) and on this launch's cameras only, so it goes
        // to the side stream: it runs beside the PREVIOUS launch's render kernel, which still reads the other set. It must
        // wait for the render kernel that last read THIS set (two launches ago); the render stream waits for the binning.
        HIP_TRY(hipStreamWaitEvent(ctx->side_stream, B.traced, 0)); // never recorded: no wait
        HIP_TRY(rtc_launch_binning(P.views, nviews, cam->hsize, cam->vsize, w->n, w->d_bound_s, w->d_gbound, w->d_orig_s, w->ngroups,
                                   B.tile_cnt + RTC_BIN_ROW_WORDS, B.tile_list, y0 / 8u, band_stride, ctx->side_stream, timed ? pair_bin[0] : nullptr,
                                   timed ? pair_bin[1] : nullptr, w->d_isect_s, w->d_kind_s, w->n_unb, B.tile_cnt, w->d_isect, B.prim));
        if (timed) ctx->bin_timed[slot] = true;
        P.prim = B.prim;
        HIP_TRY(hipEventRecord(B.binned, ctx->side_stream));
        HIP_TRY(hipStreamWaitEvent(ctx->stream, B.binned, 0));
        P.tile_rows = ctx->sky_rows ? B.tile_cnt : nullptr;
        P.tile_cnt = B.tile_cnt + RTC_BIN_ROW_WORDS;
        P.tile_list = B.tile_list;
        P.tiles_x = tiles_x;
        P.tiles_y = tiles_y;
        P.bin_packed = RTC_BIN_PACKED(w->n) ? 1u : 0u;
        P.n_unb = w->n_unb;
        binset = &B;
    }
    // per-render prologue table of the brute-force variants (the culled kernels do not use it)
    if (src != SRC_CULL && src != SRC_CULL2) HIP_TRY(rtc_launch_prep(w->d_isect, w->d_prim, w->n, P.views[0].vinv, stream));
    P.total_blocks = P.grid_x * P.grid_y * nviews;
    P.reps = ctx->tiles_per_wg;
    // Guided chunks (RenderParams::chunk_wgs): with `slots` workgroups resident at once, the launch's last f x slots tiles go one
    // per workgroup, the f x slots before them two, then three, four, and everything earlier eight (f = RTC_TILES_GUIDED
    // tenths, default 2.0; 0 = off; the largest chunk = RTC_TILES_KMAX). Launches of fewer than 3 rounds of workgroups are left alone.
    P.chunk_wgs[0] = P.chunk_wgs[1] = P.chunk_wgs[2] = P.chunk_wgs[3] = 0u;
    uint32_t grid_wgs = (P.total_blocks + P.reps - 1u) / P.reps;
    {
        static const uint32_t sizes[5] = {1u, 2u, 3u, 4u, 8u}; // chunk sizes from the END of the launch backwards
        const uint32_t slots = ctx->tiles_slots ? ctx->tiles_slots : (1024u * ((w->any_refl || w->any_refr) ? 4u : 5u) / std::max(1u, block / 64u));
        const unsigned long long per_level = (unsigned long long)slots * ctx->tiles_guided_tenths / 10u;
        uint32_t nlevels = 1;
        while (nlevels < 5u && sizes[nlevels] <= ctx->tiles_kmax) ++nlevels;
        // Not for a large world on a small frame (C3: 10 000 spheres at 1080p): there the NEXT launch's binning kernel is as long
        // as this render kernel, and its few waves wait for slots that long-lived workgroups free late — the solo kernel gains
        // 6 %, the pipelined frame loses 9 % (profiles/r03_exp_tiles_per_workgroup.log).
        const bool heavy_binning = w->n > 4096u && launch_pixels < 8000000ull;
        if (P.reps == 1u && per_level != 0u && nlevels > 1u && P.total_blocks >= 3u * slots && !heavy_binning) {
            unsigned long long rest = P.total_blocks, tiles[5] = {0, 0, 0, 0, 0};
            for (uint32_t l = 0; l < nlevels && rest; ++l) {
                unsigned long long tk = (l + 1u == nlevels) ? rest : std::min<unsigned long long>(rest, per_level);
                if (l) tk -= tk % sizes[l];          // whole workgroups; what does not divide joins the single-tile level
                tiles[l] = tk;
                rest -= tk;
            }
            tiles[0] += rest;
            P.chunk_wgs[0] = (uint32_t)(tiles[4] / 8u); P.chunk_wgs[1] = (uint32_t)(tiles[3] / 4u);
            P.chunk_wgs[2] = (uint32_t)(tiles[2] / 3u); P.chunk_wgs[3] = (uint32_t)(tiles[1] / 2u);
            grid_wgs = P.chunk_wgs[0] + P.chunk_wgs[1] + P.chunk_wgs[2] + P.chunk_wgs[3] + (uint32_t)tiles[0];
        }
    }
    HIP_TRY(rtc_launch_trace(&P, src, w->any_refl || w->any_refr, w->any_refr, grid_wgs, lds_bytes, stream,
                             timed ? pair[0] : nullptr, timed ? pair[1] : nullptr));
    if (binset) HIP_TRY(hipEventRecord(binset->traced, ctx->stream));
    ctx->last = rtc_launch_info{(uint32_t)src, (w->any_refl || w->any_refr) ? 1u : 0u, w->any_refr ? 1u : 0u, P.tile_cnt ? 1u : 0u,
                                P.light_cnt ? 1u : 0u, lane, block, (uint32_t)lds_bytes, P.reps, P.chunk_wgs[0] + P.chunk_wgs[1] + P.chunk_wgs[2] + P.chunk_wgs[3], {0u, 0u}};
    ++ctx->launches_total;
    // rtc_stats::pixels is known here (the kernel traces exactly the pixels of this launch's rows; Camera::render leaves the
    // last row and column alone, camera.rs:120-121): counted on the host, one atomic per wave less
    {
        const bool serial = mode == RTC_MODE_RENDER;
        unsigned long long rows = 0;
        for (uint32_t k = 0; k < grid_y; ++k) {
            const unsigned long long py0 = (unsigned long long)y0 + (unsigned long long)k * band_stride * 8u;
            if (py0 >= y1) break;
            unsigned long long r = y1 - py0 < 8u ? y1 - py0 : 8u;
            if (serial && py0 + r == cam->vsize) --r; // the image's last row
            rows += r;
        }
        ctx->pixels += rows * (cam->hsize - (serial ? 1u : 0u)) * nviews;
    }
    ++ctx->launches;
    if (timed) ++ctx->timed;
    return RTC_OK;
}

rtc_status rtc_render_rows(rtc_context *ctx, const rtc_world *w, const rtc_camera *cam, uint32_t mode, uint32_t y0,
                           uint32_t y1, void *d_rgb, void *d_rgb8, uint32_t flags) {
    if (!ctx || !w || !cam || (!d_rgb && !d_rgb8) || w->ctx != ctx) return RTC_ERR_ARG;
    if (mode > RTC_MODE_RENDER_ASYNC || cam->hsize == 0 || cam->vsize == 0 || y0 > y1 || y1 > cam->vsize) return RTC_ERR_ARG;
    if (cam->samples > 255u) return RTC_ERR_ARG; // antialiasing_samples is a u8 (camera.rs:24)
    if (y0 == y1) return RTC_OK;
    return render_launch(ctx, w, cam, mode, y0, y1, 1u, (y1 - y0 + 7u) / 8u, d_rgb, d_rgb8, flags);
}

rtc_status rtc_render_bands(rtc_context *ctx, const rtc_world *w, const rtc_camera *cam, uint32_t mode,
                            uint32_t first_band, uint32_t band_stride, void *d_rgb, void *d_rgb8, uint32_t flags) {
    if (!ctx || !w || !cam || (!d_rgb && !d_rgb8) || w->ctx != ctx) return RTC_ERR_ARG;
    if (mode > RTC_MODE_RENDER_ASYNC || cam->hsize == 0 || cam->vsize == 0 || band_stride == 0) return RTC_ERR_ARG;
    if (cam->samples > 255u) return RTC_ERR_ARG;
    const uint32_t nbands = (cam->vsize + RTC_BAND_ROWS - 1u) / RTC_BAND_ROWS;
    if (first_band >= nbands) return RTC_OK; // this caller owns no band of so small a canvas
    const uint32_t mine = (nbands - first_band + band_stride - 1u) / band_stride;
    return render_launch(ctx, w, cam, mode, first_band * RTC_BAND_ROWS, cam->vsize, band_stride, mine, d_rgb, d_rgb8, flags);
}

rtc_status rtc_stats_read(rtc_context *ctx, rtc_stats *out) {
    if (!ctx || !out) return RTC_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<unsigned long long> slots((size_t)CNT_N * CNT_SLOTS);
    HIP_TRY(drain_lanes(ctx));
    HIP_TRY(hipMemcpyAsync(slots.data(), ctx->d_counters, sizeof(unsigned long long) * slots.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    unsigned long long h[CNT_N] = {0};
    for (int sl = 0; sl < CNT_SLOTS; ++sl)
        for (int k = 0; k < CNT_N; ++k) h[k] += slots[(size_t)sl * CNT_N + k];
    std::memset(out, 0, sizeof *out);
    out->rays_primary = h[CNT_PRIMARY];
    out->rays_shadow = h[CNT_SHADOW];
    out->rays_reflect = h[CNT_REFLECT];
    out->rays_refract = h[CNT_REFRACT];
    out->pixels = ctx->pixels; // counted at launch time (render_launch)
    out->pixels_resample = h[CNT_RESAMPLE];
    out->rays_primary_proven_miss = h[CNT_SKY];
    return RTC_OK;
}

// Diagnostic: the raw replicated counter block summed over replicas (CNT_N values); not part of
// include/rtc.h. Used by tools/phase_shares.py with a -DRTC_STAMPS build.
extern "C" rtc_status rtc_debug_counters(rtc_context *ctx, unsigned long long *out, uint32_t n) {
    if (!ctx || !out) return RTC_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<unsigned long long> slots((size_t)CNT_N * CNT_SLOTS);
    HIP_TRY(hipMemcpyAsync(slots.data(), ctx->d_counters, sizeof(unsigned long long) * slots.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (uint32_t k = 0; k < n && k < CNT_N; ++k) {
        out[k] = 0;
        for (int sl = 0; sl < CNT_SLOTS; ++sl) out[k] += slots[(size_t)sl * CNT_N + k];
    }
    return RTC_OK;
}

// Diagnostic (not part of include/rtc.h): device allocations made by the render entry points of this context so far — the
// binning sets and rtc_render's scratch canvas. A frame sequence must stop allocating after its first launch
// (tests/test_gpu_group.py::test_render_paths_stop_allocating_after_the_first_launch).
extern "C" rtc_status rtc_debug_render_allocs(rtc_context *ctx, unsigned long long *out) {
    if (!ctx || !out) return RTC_ERR_ARG;
    *out = ctx->render_allocs;
    return RTC_OK;
}

rtc_status rtc_stats_reset(rtc_context *ctx) {
    if (!ctx) return RTC_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(drain_lanes(ctx));
    HIP_TRY(hipMemsetAsync(ctx->d_counters, 0, sizeof(unsigned long long) * CNT_N * CNT_SLOTS, ctx->stream));
    if (ctx->lanes > 1) HIP_TRY(hipStreamSynchronize(ctx->stream)); // the lanes are not ordered behind the stream
    ctx->pixels = 0;
    return RTC_OK;
}

rtc_status rtc_context_set_timing(rtc_context *ctx, uint32_t every) {
    if (!ctx) return RTC_ERR_ARG;
    if (every != 0) { // a caller that asks for timings gets the first 64 event pairs now, not 16 at a time in the middle of its timed loop
        HIP_TRY(hipSetDevice(ctx->device));
        const uint32_t upto = std::min<uint32_t>(rtc_context::EV_RING, 64u);
        for (uint32_t k = ctx->ev_created; k < upto; ++k) {
            HIP_TRY(hipEventCreate(&ctx->ev[k][0]));
            HIP_TRY(hipEventCreate(&ctx->ev[k][1]));
            HIP_TRY(hipEventCreate(&ctx->ev_bin[k][0]));
            HIP_TRY(hipEventCreate(&ctx->ev_bin[k][1]));
            ctx->ev_created = k + 1;
        }
    }
    ctx->time_every = every;
    ctx->launches = 0; // the next launch is sampled (if any is), and the ring starts afresh
    ctx->timed = 0;
    return RTC_OK;
}

rtc_status rtc_kernel_times_ms(rtc_context *ctx, float *out, uint32_t cap, uint32_t *n) {
    if (!ctx || !n || (cap && !out)) return RTC_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    const uint64_t have = ctx->timed < rtc_context::EV_RING ? ctx->timed : rtc_context::EV_RING;
    const uint64_t take = have < cap ? have : cap;
    *n = (uint32_t)take;
    if (take == 0) return RTC_OK;
    HIP_TRY(hipEventSynchronize(ctx->ev[(ctx->timed - 1) % rtc_context::EV_RING][1]));
    for (uint64_t k = 0; k < take; ++k) {
        hipEvent_t *pair = ctx->ev[(ctx->timed - take + k) % rtc_context::EV_RING];
        HIP_TRY(hipEventElapsedTime(&out[k], pair[0], pair[1]));
    }
    return RTC_OK;
}

rtc_status rtc_binning_times_ms(rtc_context *ctx, float *out, uint32_t cap, uint32_t *n) {
    if (!ctx || !n || (cap && !out)) return RTC_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    const uint64_t have = ctx->timed < rtc_context::EV_RING ? ctx->timed : rtc_context::EV_RING;
    const uint64_t take = have < cap ? have : cap;
    *n = (uint32_t)take;
    if (take == 0) return RTC_OK;
    HIP_TRY(hipEventSynchronize(ctx->ev[(ctx->timed - 1) % rtc_context::EV_RING][1])); // the render kernel follows its binning
    for (uint64_t k = 0; k < take; ++k) {
        const uint64_t sl = (ctx->timed - take + k) % rtc_context::EV_RING;
        out[k] = 0.f;
        if (ctx->bin_timed[sl]) HIP_TRY(hipEventElapsedTime(&out[k], ctx->ev_bin[sl][0], ctx->ev_bin[sl][1]));
    }
    return RTC_OK;
}

rtc_status rtc_last_kernel_ms(rtc_context *ctx, float *ms) {
    uint32_t n = 0;
    if (!ctx || !ms) return RTC_ERR_ARG;
    const rtc_status st = rtc_kernel_times_ms(ctx, ms, 1, &n);
    if (st != RTC_OK) return st;
    return n == 1 ? RTC_OK : RTC_ERR_ARG;
}

static_assert(RTC_MAX_VIEWS == RTC_MAX_VIEWS_PER_LAUNCH, "include/rtc.h and rtc_device.h disagree");

rtc_status rtc_render_views(rtc_context *ctx, const rtc_world *w, const rtc_camera *cams, uint32_t nviews, uint32_t mode,
                            uint32_t first_band, uint32_t band_stride, void *d_rgb, void *d_rgb8, uint32_t view_rows,
                            uint32_t flags) {
    if (!ctx || !w || !cams || (!d_rgb && !d_rgb8) || w->ctx != ctx) return RTC_ERR_ARG;
    if (nviews == 0 || nviews > RTC_MAX_VIEWS_PER_LAUNCH || band_stride == 0 || mode > RTC_MODE_RENDER_ASYNC) return RTC_ERR_ARG;
    if (cams[0].hsize == 0 || cams[0].vsize == 0) return RTC_ERR_ARG;
    for (uint32_t v = 1; v < nviews; ++v)
        if (cams[v].hsize != cams[0].hsize || cams[v].vsize != cams[0].vsize || cams[v].samples != cams[0].samples)
            return RTC_ERR_ARG; // one grid, one sampling pattern per launch
    if (cams[0].samples > 255u) return RTC_ERR_ARG;
    const uint32_t nbands = (cams[0].vsize + RTC_BAND_ROWS - 1u) / RTC_BAND_ROWS;
    if (first_band >= nbands) return RTC_OK;
    const uint32_t mine = (nbands - first_band + band_stride - 1u) / band_stride;
    if (view_rows < mine * RTC_BAND_ROWS) return RTC_ERR_ARG;
    int src;
    uint32_t cap;
    size_t lds;
    choose_source(ctx, w->n, flags, &src, &cap, &lds);
    if (src != SRC_CULL && src != SRC_CULL2) {
        // the brute-force variants keep a per-render table of the camera origin in object space: one view per launch
        for (uint32_t v = 0; v < nviews; ++v) {
            const rtc_status st = render_launch(ctx, w, cams + v, mode, first_band * RTC_BAND_ROWS, cams[0].vsize, band_stride, mine,
                                                d_rgb ? static_cast<double *>(d_rgb) + (size_t)v * view_rows * cams[0].hsize * 3u : nullptr,
                                                d_rgb8 ? static_cast<unsigned char *>(d_rgb8) + (size_t)v * view_rows * cams[0].hsize * 3u : nullptr,
                                                flags);
            if (st != RTC_OK) return st;
        }
        return RTC_OK;
    }
    return render_launch(ctx, w, cams, mode, first_band * RTC_BAND_ROWS, cams[0].vsize, band_stride, mine, d_rgb, d_rgb8, flags,
                         nviews, view_rows);
}

rtc_status rtc_render(rtc_context *ctx, const rtc_world *w, const rtc_camera *cam, uint32_t mode, uint32_t flags,
                      double *rgb, rtc_stats *stats) {
    if (!ctx || !w || !cam || !rgb || w->ctx != ctx) return RTC_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t bytes = sizeof(double) * 3 * (size_t)cam->hsize * cam->vsize;
    if (bytes == 0) return RTC_ERR_ARG;
    if (ctx->canvas_bytes < bytes) { // grow-only scratch canvas: no hipMalloc/hipFree (a device sync) per frame
        if (ctx->d_canvas) (void)hipFree(ctx->d_canvas);
        ctx->d_canvas = nullptr;
        ctx->canvas_bytes = 0;
        ++ctx->render_allocs;
        const hipError_t e = hipMalloc(&ctx->d_canvas, bytes);
        if (e != hipSuccess) { (void)hipGetLastError(); return e == hipErrorOutOfMemory ? RTC_ERR_NOMEM : RTC_ERR_DEVICE; }
        ctx->canvas_bytes = bytes;
    }
    double *d = ctx->d_canvas;
    rtc_status st = RTC_OK;
    if (stats) st = rtc_stats_reset(ctx);
    if (st == RTC_OK) st = rtc_render_rows(ctx, w, cam, mode, 0, cam->vsize, d, nullptr, flags);
    if (st == RTC_OK && drain_lanes(ctx) != hipSuccess) st = RTC_ERR_DEVICE; // pipelined context: the copy below is on the stream
    // `rgb` from rtc_host_alloc (page-locked) is filled by one DMA at link speed; pageable memory
    // goes through the runtime's bounce buffers (several times slower)
    if (st == RTC_OK && hipMemcpyAsync(rgb, d, bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) st = RTC_ERR_DEVICE;
    if (st == RTC_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = RTC_ERR_DEVICE;
    if (st == RTC_OK && stats) st = rtc_stats_read(ctx, stats);
    return st;
}

rtc_status rtc_render_rgb8(rtc_context *ctx, const rtc_world *w, const rtc_camera *cam, uint32_t mode, uint32_t flags,
                           uint8_t *rgb8, rtc_stats *stats) {
    if (!ctx || !w || !cam || !rgb8 || w->ctx != ctx) return RTC_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)3 * cam->hsize * cam->vsize;
    if (bytes == 0) return RTC_ERR_ARG;
    if (ctx->canvas8_bytes < bytes) { // grow-only, like rtc_render's f64 scratch
        if (ctx->d_canvas8) (void)hipFree(ctx->d_canvas8);
        ctx->d_canvas8 = nullptr;
        ctx->canvas8_bytes = 0;
        ++ctx->render_allocs;
        const hipError_t e = hipMalloc(&ctx->d_canvas8, bytes);
        if (e != hipSuccess) { (void)hipGetLastError(); return e == hipErrorOutOfMemory ? RTC_ERR_NOMEM : RTC_ERR_DEVICE; }
        ctx->canvas8_bytes = bytes;
    }
    rtc_status st = RTC_OK;
    if (stats) st = rtc_stats_reset(ctx);
    // only the 8-bit rows leave the kernel: no f64 canvas is written (d_rgb = NULL)
    if (st == RTC_OK) st = rtc_render_rows(ctx, w, cam, mode, 0, cam->vsize, nullptr, ctx->d_canvas8, flags);
    if (st == RTC_OK && drain_lanes(ctx) != hipSuccess) st = RTC_ERR_DEVICE;
    if (st == RTC_OK && hipMemcpyAsync(rgb8, ctx->d_canvas8, bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) st = RTC_ERR_DEVICE;
    if (st == RTC_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = RTC_ERR_DEVICE;
    if (st == RTC_OK && stats) st = rtc_stats_read(ctx, stats);
    return st;
}

// render_lua (lua.rs:50-91) for a program rtc_lua_run has interpreted: every job is one render launch, 8-bit rows only. The
// launches go through the context's lanes (pipeline depth 3 unless the caller chose one), each followed on its own lane by
// the copy of its frame into a page-locked host buffer — the shape of the reference's AddFrame loop, one camera per call —
// and the frames are handed to `fn` in job order while later ones are still being rendered.
rtc_status rtc_lua_program_render(rtc_context *ctx, const rtc_lua_program *prog, uint32_t mode, uint32_t flags, rtc_lua_frame_fn fn,
                                  void *user, rtc_stats *stats) {
    if (!ctx || !prog || mode > RTC_MODE_RENDER_ASYNC) return RTC_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    constexpr uint32_t RING = rtc_context::MAX_LANES + 1u; // a frame's buffers are reused only after `depth` later launches
    struct Slot {
        unsigned char *d = nullptr, *h = nullptr;
        size_t cap = 0;
        hipEvent_t done = nullptr;
        bool pending = false;
        uint32_t job = 0;
    } ring[RING];
    const uint32_t njobs = rtc_lua_program_jobs(prog);
    const uint32_t lanes_before = ctx->lanes;
    rtc_world *world = nullptr;
    rtc_status st = RTC_OK;
    bool stop = false;
    auto deliver = [&](Slot &sl) -> rtc_status { // wait for the slot's frame and hand it over
        if (!sl.pending) return RTC_OK;
        sl.pending = false;
        if (hipEventSynchronize(sl.done) != hipSuccess) return RTC_ERR_DEVICE;
        rtc_lua_job job;
        const rtc_status js = rtc_lua_program_job(prog, sl.job, &job);
        if (js != RTC_OK) return js;
        if (fn && !stop && fn(user, &job, sl.job, sl.h) != 0) stop = true;
        return RTC_OK;
    };
    auto drain = [&](uint32_t next_job) -> rtc_status { // every frame in flight, oldest first
        rtc_status r = RTC_OK;
        for (uint32_t k = 0; k < RING; ++k) {
            const rtc_status d = deliver(ring[(next_job + k) % RING]);
            if (r == RTC_OK) r = d;
        }
        return r;
    };
    if (stats) st = rtc_stats_reset(ctx);
    if (st == RTC_OK && lanes_before == 1u && njobs > 1u) st = rtc_context_set_pipeline(ctx, 3u);
    uint32_t i = 0;
    for (; st == RTC_OK && !stop && i < njobs; ++i) {
        rtc_lua_job job;
        st = rtc_lua_program_job(prog, i, &job);
        if (st != RTC_OK) break;
        const size_t bytes = (size_t)3 * job.camera.hsize * job.camera.vsize;
        if (bytes == 0) { st = RTC_ERR_ARG; break; }
        Slot &sl = ring[i % RING];
        st = deliver(sl);
        if (st != RTC_OK || stop) break;
        if (!world || !job.same_world_as_previous) { // a new World: nothing may still read the old one
            st = drain(i);
            if (st == RTC_OK) st = rtc_context_synchronize(ctx);
            if (st != RTC_OK || stop) break;
            if (world) rtc_world_destroy(world);
            world = nullptr;
            st = rtc_world_create(ctx, job.shapes, job.n_shapes, &job.light, &world);
            if (st != RTC_OK) break;
        }
        if (sl.cap < bytes) {
            if (sl.d) (void)hipFree(sl.d);
            if (sl.h) (void)hipHostFree(sl.h);
            sl.d = sl.h = nullptr;
            sl.cap = 0;
            hipError_t e = hipMalloc(&sl.d, bytes);
            if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&sl.h), bytes, hipHostMallocDefault);
            if (e != hipSuccess) { (void)hipGetLastError(); st = e == hipErrorOutOfMemory ? RTC_ERR_NOMEM : RTC_ERR_DEVICE; break; }
            sl.cap = bytes;
        }
        if (!sl.done && hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) != hipSuccess) { st = RTC_ERR_DEVICE; break; }
        st = rtc_render_rows(ctx, world, &job.camera, mode, 0, job.camera.vsize, nullptr, sl.d, flags);
        if (st != RTC_OK) break;
        hipStream_t s = ctx->lanes > 1u ? ctx->lane[ctx->last.lane] : ctx->stream; // the stream that launch went to
        if (hipMemcpyAsync(sl.h, sl.d, bytes, hipMemcpyDeviceToHost, s) != hipSuccess || hipEventRecord(sl.done, s) != hipSuccess) { st = RTC_ERR_DEVICE; break; }
        sl.pending = true;
        sl.job = i;
    }
    {
        const rtc_status d = drain(i);
        if (st == RTC_OK) st = d;
    }
    const rtc_status sy = rtc_context_synchronize(ctx);
    if (st == RTC_OK) st = sy;
    if (world) rtc_world_destroy(world);
    for (Slot &sl : ring) {
        if (sl.d) (void)hipFree(sl.d);
        if (sl.h) (void)hipHostFree(sl.h);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (ctx->lanes != lanes_before) {
        const rtc_status r = rtc_context_set_pipeline(ctx, lanes_before);
        if (st == RTC_OK) st = r;
    }
    if (st == RTC_OK && stats) st = rtc_stats_read(ctx, stats);
    return st;
}

rtc_status rtc_host_alloc(size_t bytes, void **out) {
    if (!out || bytes == 0) return RTC_ERR_ARG;
    *out = nullptr;
    void *p = nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e == hipErrorOutOfMemory) return RTC_ERR_NOMEM;
    if (e != hipSuccess) return RTC_ERR_DEVICE;
    *out = p;
    return RTC_OK;
}

void rtc_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

rtc_status rtc_host_register(void *p, size_t bytes) {
    if (!p || bytes == 0) return RTC_ERR_ARG;
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (e == hipErrorOutOfMemory) return RTC_ERR_NOMEM;
    if (e == hipErrorHostMemoryAlreadyRegistered) { (void)hipGetLastError(); return RTC_OK; }
    return e == hipSuccess ? RTC_OK : RTC_ERR_DEVICE;
}

rtc_status rtc_host_unregister(void *p) {
    if (!p) return RTC_ERR_ARG;
    return hipHostUnregister(p) == hipSuccess ? RTC_OK : RTC_ERR_DEVICE;
}

rtc_status rtc_color_at(rtc_context *ctx, const rtc_world *w, const double *rays, uint32_t n, uint32_t remaining,
                        uint32_t flags, double *rgb, rtc_hit *hits) {
    if (!ctx || !w || !rays || !rgb || w->ctx != ctx) return RTC_ERR_ARG;
    if (remaining > RTC_MAX_REFLECTIONS) return RTC_ERR_ARG; // frame stack depth of the kernel = Camera::MAX_REFLECTIONS (camera.rs:31)
    if (n == 0) return RTC_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    double *d_rays = nullptr, *d_rgb = nullptr;
    rtc_hit *d_hits = nullptr;
    rtc_status st = RTC_OK;
    if (hipMalloc(&d_rays, sizeof(double) * 6 * n) != hipSuccess || hipMalloc(&d_rgb, sizeof(double) * 3 * n) != hipSuccess ||
        (hits && hipMalloc(&d_hits, sizeof(rtc_hit) * n) != hipSuccess))
        st = RTC_ERR_DEVICE;
    if (st == RTC_OK && hipMemcpyAsync(d_rays, rays, sizeof(double) * 6 * n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        st = RTC_ERR_DEVICE;
    if (st == RTC_OK) {
        RenderParams P;
        std::memset(&P, 0, sizeof P);
        fill_world(P, w);
        if (!ctx->light_lists) P.light_cnt = nullptr;
        P.W = n; P.H = 1; P.y0 = 0; P.y1 = 1; P.mode = RTC_MODE_RENDER_ASYNC; P.samples = 1;
        P.out = d_rgb;
        P.counters = nullptr;
        P.rays = d_rays;
        P.nrays = n;
        P.remaining = remaining;
        P.hits = d_hits;
        int src;
        size_t lds_bytes;
        choose_source(ctx, w->n, flags, &src, &P.tile_cap, &lds_bytes);
        const uint32_t blk = RTC_BLOCK_FOR(src == SRC_CULL2 ? 2 : src == SRC_CULL ? 1 : 0, w->any_refl || w->any_refr, w->any_refr, true);
        P.grid_x = (n + blk - 1u) / blk;
        P.grid_y = 1;
        P.band_stride = 1;
        P.flags = flags;
        P.total_blocks = P.grid_x;
        P.reps = 1;
        P.chunk_wgs[0] = P.chunk_wgs[1] = P.chunk_wgs[2] = P.chunk_wgs[3] = 0;
        if (rtc_launch_trace(&P, src, w->any_refl || w->any_refr, w->any_refr, P.grid_x, lds_bytes, ctx->stream, nullptr,
                             nullptr) != hipSuccess)
            st = RTC_ERR_DEVICE;
    }
    if (st == RTC_OK && hipMemcpyAsync(rgb, d_rgb, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        st = RTC_ERR_DEVICE;
    if (st == RTC_OK && hits && hipMemcpyAsync(hits, d_hits, sizeof(rtc_hit) * n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        st = RTC_ERR_DEVICE;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) st = RTC_ERR_DEVICE;
    if (d_rays) (void)hipFree(d_rays);
    if (d_rgb) (void)hipFree(d_rgb);
    if (d_hits) (void)hipFree(d_hits);
    return st;
}

rtc_status rtc_device_arith(rtc_context *ctx, uint32_t op, const double *a, const double *b, uint32_t n, double *out) {
    if (!ctx || !a || !out || op > 6 || (op == 1 && !b) || (op == 2 && !b)) return RTC_ERR_ARG;
    if (n == 0) return RTC_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    double *da = nullptr, *db = nullptr, *dout = nullptr;
    rtc_status st = RTC_OK;
    const size_t bytes = sizeof(double) * n;
    if (hipMalloc(&da, bytes) != hipSuccess || hipMalloc(&db, bytes) != hipSuccess || hipMalloc(&dout, bytes) != hipSuccess)
        st = RTC_ERR_DEVICE;
    if (st == RTC_OK && hipMemcpy(da, a, bytes, hipMemcpyHostToDevice) != hipSuccess) st = RTC_ERR_DEVICE;
    if (st == RTC_OK && hipMemcpy(db, b ? b : a, bytes, hipMemcpyHostToDevice) != hipSuccess) st = RTC_ERR_DEVICE;
    if (st == RTC_OK && rtc_launch_arith(op, da, db, n, dout, ctx->stream) != hipSuccess) st = RTC_ERR_DEVICE;
    if (st == RTC_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = RTC_ERR_DEVICE;
    if (st == RTC_OK && hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost) != hipSuccess) st = RTC_ERR_DEVICE;
    if (da) (void)hipFree(da);
    if (db) (void)hipFree(db);
    if (dout) (void)hipFree(dout);
    return st;
}

} // extern "C"
