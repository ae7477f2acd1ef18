// rtc_kernels.hip — hand-written CDNA4 (gfx950, wave64) kernels for the hot path
//   Camera::render -> World::color_at -> World::intersect -> shade_hit
// of joedane/raytracer-challenge (ch1/src/camera.rs:94-160, shape.rs:677-781,
// material.rs:319-361).
//
// Design:
//  * one thread per pixel; a wave owns an 8x8 pixel tile (coherent hit / miss / shadow
//    decisions), a workgroup 2 such tiles side by side (1 for the frame-stack kernels, rtc_device.h);
//    tile = workgroup id, which the hardware
//    deals round-robin over the 8 XCDs (an even share of the image for each, RTC_TILE_ORDER).
//  * every value is IEEE f64 evaluated in the reference's operation order; the file is
//    compiled with -ffp-contract=off, so results are bit-identical to the CPU path apart
//    from pow() (material.rs:355).
//  * the object loop is wave-uniform. Default (SRC_CULL, SRC_CULL2 above 256 objects): a
//    conservative per-wave cull — 64 objects (or groups of 64) at a time, one bounding sphere per
//    lane against the wave's ray bundle, __ballot, exact tests only for the survivors, their
//    records fetched by uniform index through the scalar cache into SGPRs. Brute-force variants
//    kept for A/B (RTC_FLAG_NO_CULL): records through the scalar cache (SRC_SMEM) or from LDS
//    tiles shared by the workgroup (SRC_LDS1 one tile, SRC_LDSN many tiles with a barrier each).
//  * the sorted Intersections list of the reference (shape.rs:167-237) is replaced by its
//    streaming equivalent: running minimum over t >= 0 with first-inserted-wins ties
//    (closest hit), any-hit with early exit (shadow), and an "open set" pass for n1/n2
//    (compute_refractive, shape.rs:115-141) taken only when the hit material is transparent.
//  * recursion (reflected_color / refracted_color, depth <= 5) is an explicit per-thread
//    frame stack; colours are folded back in the reference's nesting order.
//  * no MFMA anywhere: there is no dense contraction on this path.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "rtc.h"
#include "rtc_bands.h"
#include "rtc_device.h"

// Depth of the reflection / refraction frame stack. A frame is pushed only by a color_at call whose
// `remaining` is not 0, and each level passes remaining - 1 on (shape.rs:730,735,752,765): a chain started with
// remaining = 5 (Camera::MAX_REFLECTIONS, camera.rs:31) suspends at most 5 shade_hit calls.
#define RTC_MAX_STACK 5
// Reflection-only Worlds keep their 32-byte frames (surface, kr) in LDS instead of scratch memory: 5 levels x
// 4 doubles x 64 lanes = 10 KB per wave, SoA ([level][component][lane]: lane-consecutive 8-byte accesses, no bank
// conflicts). The scratch stack was HBM traffic: C4 wrote 941 MB and fetched 236 MB per frame against 453 MB
// algorithmic (profiles/r02_a_c4_pmc.json). The output tile is staged in the same LDS (the stack is dead by then),
// so a one-wave workgroup still fits 16 times in a CU's 160 KB. 0 = the scratch stack (A/B).
#ifndef RTC_LDS_STACK
#define RTC_LDS_STACK 1
#endif
// Tile output: true = each wave stores its own 8x8 part as soon as it is done, false = workgroup
// barrier + cooperative store of the workgroup's whole tile (full 128-byte lines). Measured: the barrier
// form wins for the flat kernel (0.0738 vs 0.0784 ms, 192-byte row pieces straddle lines), the
// per-wave form for the frame-stack kernels, whose waves finish far apart (0.662 vs 0.690 ms).
// Binned primary pass: request the tile's list at kernel entry, ahead of the ray generation (1), or where the pass needs
// it (0). Measured: the early request is SLOWER (north star +3 %, C5 +4 %, where it also reads all 64 slots of a million
// tiles; profiles/r02_exp_packed_tile_lists.log), so it is off.
#ifndef RTC_BIN_HOIST
#define RTC_BIN_HOIST 0
#endif
// Binned primary pass: the exact tests take the camera origin in object space and the sphere's c from the table the binning
// kernel wrote for the view (1: closest_prim, 24 VALU instructions fewer per sphere test) or transform the origin again (0).
#ifndef RTC_BIN_PRIM
#define RTC_BIN_PRIM 1
#endif
// Canvas stores (written once, never read by the kernel): plain (0) or non-temporal (1).
#ifndef RTC_NT_STORE
#define RTC_NT_STORE 1
#endif
#if RTC_NT_STORE
#define RTC_CANVAS_STORE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define RTC_CANVAS_STORE(p, v) (*(p) = (v))
#endif
#ifndef RTC_WAVE_OUTPUT
#define RTC_WAVE_OUTPUT(REFL) (REFL)
#endif
// Per-lane prefilter (ray_touches) in front of the exact test, beyond the incoherent secondary rays
// that always get it. Shadow segments in large worlds (two-level cull): the wave-level bundle of a
// dense world keeps ~17 candidates per pass of which each lane's own segment touches few — 1000
// spheres 0.535 -> 0.423 ms, 10 000 spheres 0.560 -> 0.462 ms; at 100 objects (one-level cull, ~2
// candidates per pass) the filter costs more than it saves (0.0754 -> 0.0767 ms). The frame-stack
// kernels take it too: shadow segments from scattered secondary hits (6.7 exact tests per pass
// without it), reflective 1080p 0.551 -> 0.536 ms.
#ifndef RTC_SHADOW_LANE_FILTER
#define RTC_SHADOW_LANE_FILTER(SRC, REFL) ((SRC) == SRC_CULL2 || (REFL))
#endif
// Apex of a secondary bundle: 0 = centroid of the origins (tighter origin spread), 1 = the axis lane's origin (four wave
// reductions less per secondary pass: C4 -1.4 %, reflective 1080p -1 %; profiles/r03_exp_small_steps.log). Either is conservative: rho is
// measured from whatever apex is chosen.
#ifndef RTC_BUNDLE_APEX_LANE
#define RTC_BUNDLE_APEX_LANE 1
#endif
#ifndef RTC_PRIMARY_LANE_FILTER
#define RTC_PRIMARY_LANE_FILTER(SRC) false
#endif
// Shape of the cull walks. A round of the group level tests RTC_GROUP_SLOTS x 64 group spheres at once (ordered walks
// then take the nearest key across the whole round), a round of the one-level cull RTC_OBJ_SLOTS x 64 object spheres,
// and RTC_EXPAND_K surviving groups are expanded together. Measured on MI355X (ms per frame, 8 frames per launch):
// slots/objslots/K = 1/1/1: C3 0.234, C5 5.07, north star 0.0696; 4/2/2: 0.259 / 5.63 / 0.0724; 4/4/4: 0.274 / 5.99 / 0.0718.
// Batching the walks buys nothing: these kernels are bound by VALU issue slots, not by the length of the
// load -> test -> ballot dependency chains, and the wider rounds cost instructions and registers. Defaults 1/1/1.
#ifndef RTC_GROUP_SLOTS
#define RTC_GROUP_SLOTS 1
#endif
#ifndef RTC_OBJ_SLOTS
#define RTC_OBJ_SLOTS 1
#endif
#ifndef RTC_EXPAND_K
#define RTC_EXPAND_K 1
#endif
#ifndef RTC_TILE_ORDER
// Workgroup id -> tile. The hardware deals consecutive workgroup ids round-robin over the 8 XCDs.
// 1 (default): tile = workgroup id, so every XCD gets every 8th tile of the image — an even share
//    of sky, floor and spheres. 0: remap so that each XCD owns one contiguous band of rows (the
//    usual "XCD-aware" advice, for L2 locality): here there is nothing to share (the scene is 50 KB,
//    the canvas is write-once) and the bands are wildly uneven (sky above, everything below), so
//    the top XCDs idle: north star 0.099 ms vs 0.073 ms, reflective 1.07 vs 0.69 ms, 10 000
//    spheres 0.66 vs 0.54 ms. 2: round-robin, bottom rows first (no better than 1).
#define RTC_TILE_ORDER 1
#endif
// 2nd argument of __launch_bounds__ = minimum waves per SIMD (caps VGPRs: 5 -> 96, 4 -> 128,
// 3 -> 168, 2 -> 256). Measured on the north-star scene (culled flat kernel, 90 VGPRs):
// 4 -> 0.0784 ms, 5 -> 0.0736 ms, 6 -> 0.0882 ms with 256-thread workgroups and one frame per launch; with
// 128-thread workgroups and 8 frames per launch 4, 5 and 6 measure the same (0.5365 ms per launch).
// Kernels that carry the reflection/refraction frame stack: see RTC_WAVES_PER_SIMD_STACK.
#ifndef RTC_WAVES_PER_SIMD
#define RTC_WAVES_PER_SIMD 5
#endif
// Frame-stack kernels (124 VGPRs) on a reflective 2048x2048 / 100-sphere scene: 4 -> 0.871 ms,
// 5 -> 0.870, 6 -> 0.862 (forcing more waves trades spills for occupancy: a wash).
#ifndef RTC_WAVES_PER_SIMD_STACK
#define RTC_WAVES_PER_SIMD_STACK 4
#endif

enum { SRC_SMEM = 0, SRC_LDS1 = 1, SRC_LDSN = 2, SRC_CULL = 3, SRC_CULL2 = 4 };
#define IS_CULL(S) ((S) == SRC_CULL || (S) == SRC_CULL2)
#define CULL_LEVEL(S) ((S) == SRC_CULL2 ? 2 : (S) == SRC_CULL ? 1 : 0)

struct V3 {
    double x, y, z;
};

#define DEVI __device__ __forceinline__

// Diagnostic build only (-DRTC_STAMPS): s_memtime stamps around the phases of k_trace, summed per
// phase into counters CNT_STAMP0.. (read with rtc_debug_counters). Never enabled in the shipped
// library; the stamped build's run time is not meaningful, only the shares are.
#ifdef RTC_STAMPS
#define DIAG(i, v) do { diag_c[i] += (v); } while (0)
#define DIAG_FILTER(p) do { if (p) *(p) += 1u; } while (0)
#define DIAG_PTR(i) (&diag_c[i])
// STAMP(i): the time since the previous stamp (whichever it was) is added to phase i — summed over all passes of the wave in
// stamp_t[i], and over its secondary passes only in stamp_t[8 + i].
#define STAMP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
                      stamp_t[i] += now_ - stamp_last; if (stamp_secondary) stamp_t[8 + (i)] += now_ - stamp_last; stamp_last = now_; } while (0)
#else
#define STAMP(i) do { } while (0)
#define DIAG(i, v) do { } while (0)
#define DIAG_FILTER(p) do { } while (0)
#define DIAG_PTR(i) ((unsigned *)nullptr)
#endif

DEVI V3 mk(double x, double y, double z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
DEVI V3 vadd(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DEVI V3 vsub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DEVI V3 vmul(V3 a, double m) { return mk(a.x * m, a.y * m, a.z * m); }
DEVI V3 vmulv(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
DEVI V3 vneg(V3 a) { return mk(-a.x, -a.y, -a.z); }
// Vector::dot vec.rs:78-82: (x*x' + y*y') + z*z'
DEVI double vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// Vector::normalize vec.rs:65-76: magnitude, then three divisions
DEVI V3 vnormalize_plain(V3 a) {
    const double mag = sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return mk(a.x / mag, a.y / mag, a.z / mag);
}
// The three IEEE divisions share their divisor. hipcc expands x / y (f64) into v_div_scale x2, v_rcp, four v_fma refining
// 1/y, v_mul, v_fma, v_div_fmas, v_div_fixup (11 instructions); everything up to the refined reciprocal depends on y alone
// as long as v_div_scale leaves y unscaled, which it does unless y, 1/y or x/y leave the normal range or x is tiny
// (ISA: V_DIV_SCALE_F64). RTC_SHARED_NORMALIZE = 1 evaluates that part once when |mag| is in [2^-400, 2^400] and every
// component is +-0 or at least 2^-500 in magnitude (then no scaling happens, VCC is clear and v_div_fmas is a plain fma):
// 5 + 3 x 4 instructions + the guards instead of 33, bit-identical (tests/test_gpu_round3.py: rtc_device_arith op 5 vs op 6
// vs the host, 2 M triples). Measured (profiles/r03_exp_shared_normalize.log): no gain in k_trace (north star equal, C4 +1 %), so 0 there;
// the binning kernel uses it (primary_dir).
#ifndef RTC_SHARED_NORMALIZE
#define RTC_SHARED_NORMALIZE 0
#endif
DEVI V3 vnormalize_shared(V3 a) {
    const double mag = sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    auto in_range = [](double v) { return __builtin_amdgcn_class(v, 0x060 /* +-0 */) || fabs(v) >= 0x1p-500; };
    if (fabs(mag) >= 0x1p-400 && fabs(mag) <= 0x1p400 && in_range(a.x) && in_range(a.y) && in_range(a.z)) {
        const double r0 = __builtin_amdgcn_rcp(mag);
        const double e0 = __builtin_fma(-mag, r0, 1.0);
        const double r1 = __builtin_fma(r0, e0, r0);
        const double e1 = __builtin_fma(-mag, r1, 1.0);
        const double r = __builtin_fma(r1, e1, r1);
        auto quot = [&](double x) {
            const double q = x * r;
            const double rem = __builtin_fma(-mag, q, x);
            return __builtin_amdgcn_div_fixup(__builtin_fma(rem, r, q), mag, x);
        };
        return mk(quot(a.x), quot(a.y), quot(a.z));
    }
    return mk(a.x / mag, a.y / mag, a.z / mag);
}
DEVI V3 vnormalize(V3 a) {
#if RTC_SHARED_NORMALIZE
    return vnormalize_shared(a);
#else
    return vnormalize_plain(a);
#endif
}
// Vector::reflect vec.rs:106-108: self - n*(2*(self.n))
DEVI V3 vreflect(V3 v, V3 n) { return vsub(v, vmul(n, 2. * vdot(v, n))); }

// Matrix::transform_point transform.rs:122-128 (rows of 4, row 3 never read)
template <class P> DEVI V3 xpoint(P m, V3 p) {
    return mk(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
              m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
// Matrix::transform_vector transform.rs:107-120
template <class P> DEVI V3 xvector(P m, V3 v) {
    return mk(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
              m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
// transform_vector with a packed 3x3
DEVI V3 xvector3(const double *m, V3 v) {
    return mk(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z,
              m[6] * v.x + m[7] * v.y + m[8] * v.z);
}

DEVI unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
DEVI uint32_t popc64(unsigned long long m) { return (uint32_t)__popcll(m); }

// Cube::check_axis shape.rs:540-564
DEVI void check_axis(double origin, double direction, double &tmin, double &tmax) {
    const double tmin_numerator = -1.0 - origin;
    const double tmax_numerator = 1. - origin;
    if (fabs(direction) >= RTC_EPSILON) {
        tmin = tmin_numerator / direction;
        tmax = tmax_numerator / direction;
    } else {
        tmin = (tmin_numerator >= 0.0) ? __builtin_inf() : -__builtin_inf();
        tmax = (tmax_numerator >= 0.0) ? __builtin_inf() : -__builtin_inf();
    }
    if (tmin > tmax) {
        const double s = tmin;
        tmin = tmax;
        tmax = s;
    }
}

// Entries of one shape for a ray already in object space, in the order the reference's
// per-shape list holds them (shape.rs:361-375, 462-471, 577-591). Returns their count.
// `c_pre`: the sphere's `c` when the caller already has it (primary rays).
template <bool HAVE_C> DEVI int shape_entries(uint32_t kind, V3 o, V3 d, double c_pre, double &t0, double &t1) {
    if (kind == RTC_SPHERE) {
        const double a = vdot(d, d);
        const double b = 2. * vdot(d, o);
        const double c = HAVE_C ? c_pre : (vdot(o, o) - 1.);
        const double disc = (b * b) - 4. * a * c;
        if (disc < 0.) return 0;
        const double sq = sqrt(disc);
        const double den = 2. * a;
        t0 = (-b - sq) / den;
        t1 = (-b + sq) / den;
        return 2;
    } else if (kind == RTC_PLANE) {
        if (fabs(d.y) < RTC_EPSILON) return 0;
        t0 = -o.y / d.y;
        return 1;
    } else {
        double xmin, xmax, ymin, ymax, zmin, zmax;
        check_axis(o.x, d.x, xmin, xmax);
        check_axis(o.y, d.y, ymin, ymax);
        check_axis(o.z, d.z, zmin, zmax);
        const double tmin = fmax(xmin, fmax(ymin, zmin));
        const double tmax = fmin(xmax, fmin(ymax, zmax));
        if (tmin < tmax) {
            t0 = tmin;
            t1 = tmax;
            return 2;
        }
        return 0;
    }
}

// Does entry (t, object j) come before the current best entry (best, hidx) in the reference's
// merged sorted list? Smaller t first; at equal t the shape inserted first (lower index); an
// object's own second root never precedes its first (t1 <= t2 and j == hidx fails the test).
// Visiting objects in insertion order makes the index comparison redundant but harmless; the
// culled path visits them in Morton order and needs it.
DEVI bool closer(double t, int j, double best, int hidx) { return t < best || (t == best && j < hidx); }

// Closest-hit update for one object: Intersections::get_hit (shape.rs:220-232) over the merged
// sorted list == smallest t >= 0.0, ties to the entry inserted first (lower object index, then
// first root). For a sphere t1 <= t2, so the second root is only needed when the first is not
// a candidate (t1 < 0 or NaN).
// Object-space y row only: all a plane needs (transform.rs:116,125 for the y component).
template <class P> DEVI double xpoint_y(P m, V3 p) { return m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7]; }
template <class P> DEVI double xvector_y(P m, V3 v) { return m[4] * v.x + m[5] * v.y + m[6] * v.z; }

// Plane::intersect_local computes t = -o.y / d.y (shape.rs:467) and every caller then asks whether
// t >= 0.0. When o.y and d.y are both positive or both negative the quotient is strictly negative —
// provided it cannot underflow to -0.0 (which WOULD satisfy t >= 0.0): with |o.y| >= 2^-500 and
// |d.y| <= 2^500 its magnitude is >= 2^-1000. In that case the division (~12 instructions, several
// of them quarter-rate) can be skipped without changing any result.
DEVI bool plane_t_certainly_negative(double oy, double dy) {
    const double lo = 0x1p-500, hi = 0x1p500;
    return ((oy >= lo && dy > 0. && dy <= hi) || (oy <= -lo && dy < 0. && dy >= -hi));
}

// World-space ray in, per-kind transform inside (a plane only needs the y row of the inverse).
template <class P> DEVI void closest_world(uint32_t kind, P m, V3 ro, V3 rd, int j, double &best, int &hidx, int &hroot);
template <class P> DEVI bool occludes_world(uint32_t kind, P m, V3 ro, V3 rd, double dist);

template <bool HAVE_C>
DEVI void closest_update(uint32_t kind, V3 o, V3 d, double c_pre, int j, double &best, int &hidx, int &hroot) {
    if (kind == RTC_SPHERE) {
        const double a = vdot(d, d);
        const double b = 2. * vdot(d, o);
        const double c = HAVE_C ? c_pre : (vdot(o, o) - 1.);
        const double disc = (b * b) - 4. * a * c;
        if (!(disc < 0.)) {
            const double sq = sqrt(disc);
            const double den = 2. * a;
            const double t1 = (-b - sq) / den;
            if (t1 >= 0.0) {
                if (closer(t1, j, best, hidx)) { best = t1; hidx = j; hroot = 0; }
            } else {
                const double t2 = (-b + sq) / den;
                if (t2 >= 0.0 && closer(t2, j, best, hidx)) { best = t2; hidx = j; hroot = 1; }
            }
        }
    } else {
        double t0 = 0., t1 = 0.;
        const int cnt = shape_entries<false>(kind, o, d, 0., t0, t1);
        if (cnt >= 1 && t0 >= 0.0 && closer(t0, j, best, hidx)) { best = t0; hidx = j; hroot = 0; }
        if (cnt == 2 && t1 >= 0.0 && closer(t1, j, best, hidx)) { best = t1; hidx = j; hroot = 1; }
    }
}

// is_shadowed_by_light (shape.rs:716-727): hit.t < distance  <=>  some entry has 0 <= t < distance.
DEVI bool occludes(uint32_t kind, V3 o, V3 d, double dist) {
    if (kind == RTC_SPHERE) {
        const double a = vdot(d, d);
        const double b = 2. * vdot(d, o);
        const double c = vdot(o, o) - 1.;
        const double disc = (b * b) - 4. * a * c;
        if (disc < 0.) return false;
        const double sq = sqrt(disc);
        const double den = 2. * a;
        const double t1 = (-b - sq) / den;
        if (t1 >= 0.0) return t1 < dist;
        const double t2 = (-b + sq) / den;
        return t2 >= 0.0 && t2 < dist;
    }
    double t0 = 0., t1 = 0.;
    const int cnt = shape_entries<false>(kind, o, d, 0., t0, t1);
    if (cnt >= 1 && t0 >= 0.0 && t0 < dist) return true;
    if (cnt == 2 && t1 >= 0.0 && t1 < dist) return true;
    return false;
}

template <class P> DEVI void closest_world(uint32_t kind, P m, V3 ro, V3 rd, int j, double &best, int &hidx, int &hroot) {
    if (kind == RTC_PLANE) { // shape.rs:462-471
        const double oy = xpoint_y(m, ro), dy = xvector_y(m, rd);
        if (!(fabs(dy) < RTC_EPSILON) && !plane_t_certainly_negative(oy, dy)) {
            const double t = -oy / dy;
            if (t >= 0.0 && closer(t, j, best, hidx)) { best = t; hidx = j; hroot = 0; }
        }
    } else {
        closest_update<false>(kind, xpoint(m, ro), xvector(m, rd), 0., j, best, hidx, hroot);
    }
}
template <class P> DEVI bool occludes_world(uint32_t kind, P m, V3 ro, V3 rd, double dist) {
    if (kind == RTC_PLANE) {
        const double oy = xpoint_y(m, ro), dy = xvector_y(m, rd);
        if (fabs(dy) < RTC_EPSILON || plane_t_certainly_negative(oy, dy)) return false;
        const double t = -oy / dy;
        return t >= 0.0 && t < dist;
    }
    return occludes(kind, xpoint(m, ro), xvector(m, rd), dist);
}

// Primary rays: the object-space origin `po` (= transform_point(inv, camera origin)) and the sphere's
// c = po.po - 1 are the same for every pixel; they come from the per-render table `prim`
// (k_prep_primary, same arithmetic), so only the direction is transformed here.
template <class P, class Q> DEVI void closest_prim(uint32_t kind, P m, Q pr, V3 rd, int j, double &best, int &hidx, int &hroot) {
    if (kind == RTC_PLANE) {
        const double oy = pr[1], dy = xvector_y(m, rd);
        if (!(fabs(dy) < RTC_EPSILON) && !plane_t_certainly_negative(oy, dy)) {
            const double t = -oy / dy;
            if (t >= 0.0 && closer(t, j, best, hidx)) { best = t; hidx = j; hroot = 0; }
        }
    } else {
        closest_update<true>(kind, mk(pr[0], pr[1], pr[2]), xvector(m, rd), pr[3], j, best, hidx, hroot);
    }
}

// ---- wave64 reductions (DPP): inclusive scan inside each row of 16 lanes, then two row
// broadcasts; the total lands in lane 63. All 64 lanes must execute these (converged code);
// lanes that do not take part pass the identity.
template <int CTRL, int ROW_MASK> DEVI float dpp_f32(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                                 CTRL, ROW_MASK, 0xf, false));
}
DEVI float lane63(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63)); }
DEVI float wave_sum(float v) {
    v += dpp_f32<0x111, 0xf>(0.f, v); // row_shr:1
    v += dpp_f32<0x112, 0xf>(0.f, v); // row_shr:2
    v += dpp_f32<0x114, 0xf>(0.f, v); // row_shr:4
    v += dpp_f32<0x118, 0xf>(0.f, v); // row_shr:8
    v += dpp_f32<0x142, 0xa>(0.f, v); // row_bcast:15 into rows 1,3
    v += dpp_f32<0x143, 0xc>(0.f, v); // row_bcast:31 into rows 2,3
    return lane63(v);
}
// Maximum of NON-NEGATIVE floats (NaN-free): their bit patterns order like unsigned integers, and
// v_max_u32 with identity 0 folds into the DPP instruction (one instruction per step).
template <int CTRL, int ROW_MASK> DEVI unsigned dpp_u32(unsigned src) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)src, CTRL, ROW_MASK, 0xf, false);
}
DEVI unsigned wave_max_u32(unsigned v) {
    v = max(v, dpp_u32<0x111, 0xf>(v));
    v = max(v, dpp_u32<0x112, 0xf>(v));
    v = max(v, dpp_u32<0x114, 0xf>(v));
    v = max(v, dpp_u32<0x118, 0xf>(v));
    v = max(v, dpp_u32<0x142, 0xa>(v));
    v = max(v, dpp_u32<0x143, 0xc>(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
DEVI float wave_max_nonneg(float f) { return __builtin_bit_cast(float, wave_max_u32(__builtin_bit_cast(unsigned, f))); }

// ---- conservative per-wave cull ------------------------------------------------------------
// The rays a wave is about to trace form a bundle: a cone (apex, unit axis, half-angle theta)
// that contains every active lane's ray, optionally fattened by the spread `rho` of the ray
// origins around the apex and cut at distance `tmax`. An object whose world-space bounding
// sphere (centre C, radius R, DevBound) cannot touch the bundle cannot produce an intersection
// with t >= 0 for any lane, so skipping it leaves the closest hit / shadow bit unchanged — the
// objects that survive are visited in insertion order and tested with the exact f64 arithmetic.
// The bundle is built in f32 (cheap DPP reductions) and widened by far more than the f32 error;
// the per-object test is f64 and square-root free.
struct Bundle {
    double px, py, pz;   // apex
    double ax, ay, az;   // axis (unit to ~1e-6)
    double cosT, sinT;   // half-angle, already widened
    double rho;          // origin spread around the apex
    double tmax;         // reach along the rays (inf: unbounded)
    double spread;       // upper bound of the distance apex -> the ORIGIN the exact test uses (0 for
                         // primary rays, rho for secondary rays, tmax for shadow segments, whose
                         // exact rays start at the far end)
    bool off;            // bundle could not be bounded: every object is a candidate
};

// A wave-uniform double that the compiler would otherwise keep in two VGPRs: move it to SGPRs.
DEVI double uniform_f64(double x) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

DEVI double lane_f64(double x, uint32_t l) { // lane l's value in every lane (l wave-uniform)
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, (int)l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), (int)l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

DEVI bool finite3(V3 v) { return fabs(v.x) < __builtin_inf() && fabs(v.y) < __builtin_inf() && fabs(v.z) < __builtin_inf(); }

// SHARED: every active lane's ray starts at `apex` (the camera origin, or the light for shadow
// segments walked backwards); otherwise the apex is the centroid of the lanes' origins and `rho`
// their spread. REACH: the rays end after `reach` (shadow segments). The axis is the direction of
// one active lane near the tile centre (no reduction needed); the half-angle is the largest
// deviation from it.
template <bool SHARED, bool REACH>
DEVI Bundle make_bundle(bool active, V3 apex, V3 o, V3 d, double reach) {
#pragma clang fp contract(fast) // cull arithmetic (see bundle_touches)
    Bundle B;
    float fx = 0.f, fy = 0.f, fz = 0.f;
    bool good = active && finite3(o) && finite3(d);
    if (good) {
        const float x = (float)d.x, y = (float)d.y, z = (float)d.z;
        const float l2 = x * x + y * y + z * z;
        good = l2 > 1e-30f && l2 < 1e30f;
        const float il = __builtin_amdgcn_rsqf(l2);
        fx = x * il; fy = y * il; fz = z * il;
    }
    const unsigned long long gmask = ballot(good);
    bool off = ballot(active && !good) != 0ull || gmask == 0ull;
    // axis = direction of a lane near the middle of the 8x8 tile (lane 27 = pixel (3,3)) when it is
    // active, else the nearest active lane: a centred axis halves the cone's half-angle compared
    // with a corner lane, i.e. ~4x fewer objects survive the cull
    const unsigned long long ghi = gmask & ~((1ull << 27) - 1ull);
    const int lane0 = ghi ? __builtin_ctzll(ghi) : (gmask ? 63 - __builtin_clzll(gmask) : 0);
    const float ax = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fx), lane0));
    const float ay = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fy), lane0));
    const float az = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fz), lane0));
    float dotv = 1.f, q2 = 0.f; // cos and sin^2 of the angle to the axis
    if (good) {
        dotv = ax * fx + ay * fy + az * fz;
        const float cx = ay * fz - az * fy, cy = az * fx - ax * fz, cz = ax * fy - ay * fx;
        q2 = cx * cx + cy * cy + cz * cz;
    }
    const float q2max = wave_max_nonneg(q2);
    // narrow bundle (every lane within ~45 degrees of the axis): the common case needs only the
    // sin^2 reduction; the cosine minimum is reduced only for wide bundles
    const bool narrow = ballot(good && !(dotv > 0.7f)) == 0ull;
    float cmin = 1.f;
    if (!narrow) cmin = 1.f - wave_max_nonneg(good ? fmaxf(0.f, 1.f - dotv) : 0.f);
    float rho = 0.f;
    if constexpr (!SHARED) {
#if RTC_BUNDLE_APEX_LANE
        // apex = the origin of the axis lane (three readlanes) instead of the centroid of the origins (four wave sums)
        const float mx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (float)o.x), lane0));
        const float my = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (float)o.y), lane0));
        const float mz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (float)o.z), lane0));
#else
        const float cnt = fmaxf(wave_sum(good ? 1.f : 0.f), 1.f);
        const float mx = wave_sum(good ? (float)o.x : 0.f) / cnt;
        const float my = wave_sum(good ? (float)o.y : 0.f) / cnt;
        const float mz = wave_sum(good ? (float)o.z : 0.f) / cnt;
#endif
        apex = mk((double)mx, (double)my, (double)mz);
        float e2 = 0.f;
        if (good) {
            const double ex = o.x - apex.x, ey = o.y - apex.y, ez = o.z - apex.z;
            e2 = (float)(ex * ex + ey * ey + ez * ez) * 1.0001f;
        }
        const float e2max = wave_max_nonneg(e2);
        rho = __builtin_sqrtf(e2max) * 1.0001f + 1e-30f;
        if (!(e2max < 1e30f) || !(fabsf(mx) < 1e30f && fabsf(my) < 1e30f && fabsf(mz) < 1e30f)) off = true;
    }
    float tmax = __builtin_inff();
    if constexpr (REACH) {
        float tm = 0.f;
        if (good) tm = (reach < 1e30) ? fmaxf(0.f, (float)reach) * 1.0001f + 1e-30f : __builtin_inff();
        tmax = wave_max_nonneg(tm);
    }
    if (!(cmin > 0.2f)) off = true;
    float sinT, cosT;
    if (narrow) { // the cross product resolves small angles, the dot does not
        sinT = __builtin_sqrtf(q2max) * 1.001f + 4e-6f;
        cosT = __builtin_sqrtf(fmaxf(0.f, 1.f - sinT * sinT));
    } else {
        // widen through the cosine only and derive the sine from it: (cosT, sinT) must stay a unit
        // pair. (Inflating the sine on its own SHRINKS the computed distance budget for centres
        // behind the apex plane, wa < 0 — found by the stress campaign, 1 world in ~25 000.)
        cosT = cmin - 1e-3f;
        sinT = __builtin_sqrtf(fmaxf(0.f, 1.f - cosT * cosT));
    }
    if (!(sinT < 0.98f)) off = true;
    B.off = off;
    B.px = uniform_f64(apex.x); B.py = uniform_f64(apex.y); B.pz = uniform_f64(apex.z);
    B.ax = uniform_f64((double)ax); B.ay = uniform_f64((double)ay); B.az = uniform_f64((double)az);
    B.cosT = uniform_f64((double)cosT); B.sinT = uniform_f64((double)sinT);
    B.rho = uniform_f64((double)rho);
    B.tmax = uniform_f64((double)tmax);
    B.spread = REACH ? B.tmax : B.rho;
    if (!(B.spread < 1e300)) off = true; // shadow segment of unbounded length: do not cull
    B.off = off;
    return B;
}

// Can the bounding sphere touch the bundle? (false => provably no intersection with t >= 0.)
// Let (wa, perp) be the centre's axial / radial coordinates about the axis. The distance from the
// centre to the cone's side is perp*cos(theta) - wa*sin(theta); the sphere (fattened by rho) can
// touch the solid cone only if that is <= Re, it is not wholly behind the apex plane, and it
// is within reach. Squares instead of square roots.
// `key` (KEYED only): a lower bound of the distance from the apex to any point of the (inflated)
// sphere, 0 when there is none to give. For rays that START at the apex with a unit direction
// (primary rays) every intersection of the object has t >= key, which is what lets the ordered walk
// of the two-level cull stop early (for_each_object, Skip).
template <bool KEYED = false> DEVI bool bundle_touches(const Bundle &B, const DevBound &b, float *key = nullptr) {
#pragma clang fp contract(fast) // cull arithmetic, not reference arithmetic: the file is compiled with
                                // -ffp-contract=off for the exact tests; here FMA halves the dot products and only
                                // tightens the rounding that the slacks below already cover
    if constexpr (KEYED) *key = 0.f;
    if (B.off) return true;
    if (!(b.r < __builtin_inf())) return true;
    const double wx = b.cx - B.px, wy = b.cy - B.py, wz = b.cz - B.pz;
    // rounding inflation of the radius (rtc_device.h, DevBound): D <= |C - apex|_1 + spread
    const double l1 = fabs(wx) + fabs(wy) + fabs(wz);
    const double Dub = l1 + B.spread;
    const double r_eff = b.r + b.r * (b.k * Dub * (b.cn + Dub));
    // slack: 1e-5 relative on the radius, plus 1e-6 of the centre's L1 distance (>= its Euclidean
    // distance) for the f32 origin of (axis, cosT, sinT): axis length and cos^2+sin^2 are 1 to ~2e-6
    const double Re = (r_eff + B.rho) * 1.00001 + 1e-6 * l1 + 1e-12;
    const double d2 = wx * wx + wy * wy + wz * wz;
    if (d2 <= Re * Re) return true;
    if constexpr (KEYED) { // |C - apex| - Re, rounded down with 1e-6 margins (f32 conversions and sqrt err ~1e-7)
        const float dist = __builtin_sqrtf((float)d2) * 0.999999f, rad = (float)Re * 1.000001f;
        *key = fmaxf(0.f, (dist - rad) * 0.999999f); // NaN / inf - inf -> 0: no bound
    }
    const double wa = wx * B.ax + wy * B.ay + wz * B.az;
    if (wa < -Re) return false;                       // wholly behind the apex plane (theta < 90 deg)
    const double far = B.tmax + Re;
    if (d2 > far * far) return false;                  // beyond the reach of every ray (inf*inf = inf: never)
    // the axis is unit only to ~2e-6: take an upper bound of the axial coordinate on the right-hand
    // side and a lower bound of perp^2 on the left, so the inequality can only err towards "keep"
    const double rhs = Re + (wa + fabs(wa) * 1e-5) * B.sinT;
    if (rhs < 0.) return false;
    const double perp2 = d2 - wa * wa * 1.00001;
    return !(perp2 * (B.cosT * B.cosT) > rhs * rhs);   // NaN-safe: keep the object unless provably far
}

// The KEY of bundle_touches<true> on its own: a lower bound of the distance from `apex` to the bound's inflated sphere
// (0 when the apex may be inside or nothing can be said). Rays that start at the apex with unit directions meet the object
// only at t >= key.
DEVI float bound_key(V3 apex, const DevBound &b) {
#pragma clang fp contract(fast)
    if (!(b.r < __builtin_inf())) return 0.f;
    const double wx = b.cx - apex.x, wy = b.cy - apex.y, wz = b.cz - apex.z;
    const double l1 = fabs(wx) + fabs(wy) + fabs(wz);
    const double r_eff = b.r + b.r * (b.k * l1 * (b.cn + l1));
    const double Re = r_eff * 1.00001 + 1e-6 * l1 + 1e-12;
    const double d2 = wx * wx + wy * wy + wz * wz;
    if (d2 <= Re * Re) return 0.f;
    const float dist = __builtin_sqrtf((float)d2) * 0.999999f, rad = (float)Re * 1.000001f;
    return fmaxf(0.f, (dist - rad) * 0.999999f); // NaN / inf - inf -> 0: no bound
}

// Direction (from the light) -> cube-map cell. Face = the axis of the largest component (ties: x before y before z) and
// its sign; (u, v) = the other two components over it, each in [-1, 1]; cell = floor((u + 1) * R / 2), clamped.
// Evaluated in f32 (a third of the f64 form's instructions, one of them a division): the direction is unit, so u/m and v/m carry
// an absolute error below 3e-7 — 2e-5 of a cell — which k_light_cells' cones cover fifty times over (their half-angles are
// widened by 1e-5 rad for exactly this: a direction on a cell border may land in either cell); a tie between two components
// resolved differently than in f64 is a direction on a face border, covered by both faces' border cells the same way.
DEVI uint32_t light_cell(V3 d) {
    const float x = (float)d.x, y = (float)d.y, z = (float)d.z;
    const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
    uint32_t a;
    float m, u, v;
    if (ax >= ay && ax >= az) { a = 0u; m = x; u = y; v = z; }
    else if (ay >= az) { a = 1u; m = y; u = z; v = x; }
    else { a = 2u; m = z; u = x; v = y; }
    const float im = __builtin_amdgcn_rcpf(fabsf(m));
    const float fu = (u * im + 1.0f) * (0.5f * RTC_LIGHT_R), fv = (v * im + 1.0f) * (0.5f * RTC_LIGHT_R);
    const int iu = (int)fminf(fmaxf(fu, 0.0f), (float)(RTC_LIGHT_R - 1u)), iv = (int)fminf(fmaxf(fv, 0.0f), (float)(RTC_LIGHT_R - 1u)); // NaN -> 0
    return ((a * 2u + (m < 0.f ? 1u : 0u)) * RTC_LIGHT_R + (uint32_t)iv) * RTC_LIGHT_R + (uint32_t)iu;
}
// Per-lane prefilter for INCOHERENT rays (reflection / refraction): can THIS lane's ray, for some
// t >= 0, touch the object's bounding sphere? false => the exact test would find no entry with
// t >= 0 for this lane. ~22 f64 instructions against 54+ for the exact test; the exact test is
// then issued only if some lane of the wave passes. (For coherent primary / shadow bundles the
// wave-level cull already leaves 1-2 candidates per pass and this filter would only add work.)
DEVI bool ray_touches(V3 o, V3 d, const DevBound &b) {
#pragma clang fp contract(fast) // cull arithmetic, not reference arithmetic: FMA only tightens the rounding the slacks cover
    if (!(b.r < __builtin_inf())) return true;
    const double wx = b.cx - o.x, wy = b.cy - o.y, wz = b.cz - o.z;
    const double Dub = fabs(wx) + fabs(wy) + fabs(wz); // >= |C - o|
    const double R = (b.r + b.r * (b.k * Dub * (b.cn + Dub))) * 1.000001 + 1e-12; // rounding inflation, see DevBound
    const double ww = wx * wx + wy * wy + wz * wz;
    const double R2 = R * R;
    if (ww <= R2) return true;                       // origin inside the sphere
    const double proj = wx * d.x + wy * d.y + wz * d.z;
    if (proj < 0.) return false;                      // sphere wholly behind the origin (ww > R^2)
    const double dd = d.x * d.x + d.y * d.y + d.z * d.z;
    // perp^2 * dd = ww*dd - proj^2 <= R^2 * dd ; the slack covers the cancellation error
    return !(ww * dd - proj * proj > R2 * dd + 1e-11 * ww * dd); // NaN-safe: keep unless provably far
}

// The same question against a pre-inflated record (DevPre, rtc_device.h) — valid for origins with |o|_1 <= pre_limit, which the
// caller checks once per pass. One inequality: with pm = max(proj, 0) the squared distance from the centre to the RAY (t >= 0)
// is ww - pm^2 / dd, so the sphere can be touched only if ww*dd - pm^2 <= R2*dd (+ the cancellation slack); proj < 0 turns it
// into ww <= R2 (origin inside), as the general test has it. `dd` = d.d, computed once per pass.
DEVI bool ray_touches_pre(V3 o, V3 d, double dd, const DevPre &b) {
#pragma clang fp contract(fast) // cull arithmetic
    const double wx = b.cx - o.x, wy = b.cy - o.y, wz = b.cz - o.z;
    const double ww = wx * wx + wy * wy + wz * wz;
    const double pm = fmax(wx * d.x + wy * d.y + wz * d.z, 0.);
    return !(ww * dd - pm * pm > (b.R2 + 1e-11 * ww) * dd); // NaN-safe: keep unless provably far; R2 = inf: never culled
}
// The same test as the MASK of the active lanes it keeps: the compare writes the wave's mask directly (v_cmp -> SGPR pair), where
// ballot(bool) of a value that lives across blocks costs a v_cndmask 0/1 and a second compare. Used in the batched walk of the
// one-level kernels (C4 -1.1 %, reflective 1080p -1.4 %); the same form in the two-level kernels' walks and in the binned walk's
// early-out measured 3 % SLOWER on C3 / C5 (profiles/r03_exp_small_steps.log) and is not used there.
DEVI unsigned long long ray_touches_pre_mask(V3 o, V3 d, double dd, const DevPre &b) {
#pragma clang fp contract(fast) // cull arithmetic
    const double wx = b.cx - o.x, wy = b.cy - o.y, wz = b.cz - o.z;
    const double ww = wx * wx + wy * wy + wz * wz;
    const double pm = fmax(wx * d.x + wy * d.y + wz * d.z, 0.);
    return __builtin_amdgcn_fcmp(ww * dd - pm * pm, (b.R2 + 1e-11 * ww) * dd, 13 /* ULE: !(lhs > rhs) */);
}
// Candidates of a per-lane-filtered walk are taken RTC_PRE_BATCH at a time: their records are requested together and their
// prefilters evaluated back to back before any exact test, so that a pass over many candidates — a secondary pass whose bundle
// cannot be bounded visits all 101 objects of the reflective north star — is not one scalar-load round trip per object
// (profiles/r03_exp_unbounded_walks.log: those walks, 0.27 passes per wave, were a third of C4's kernel time).
#ifndef RTC_PRE_BATCH
#define RTC_PRE_BATCH 2
#endif

// ---- wave-uniform object loop ------------------------------------------------------------
// f(j, m, kind, prim) is called for objects j = 0..n-1 in insertion order (World::intersect,
// shape.rs:679-681) and returns true while some lane of the wave still needs objects.
// SRC_SMEM : uniform global loads (scalar cache -> SGPR operands); no barriers.
// SRC_LDS1 : the whole table was staged into LDS once by stage_all(); no barriers.
// SRC_LDSN : tiles of P.tile_cap objects staged by the whole workgroup; EVERY thread of the
//            workgroup must call this the same number of times (barriers inside).
// The World tables are passed as separate `const T *__restrict__` kernel arguments (not inside the
// parameter struct): only then may the compiler treat them as read-only for the whole launch and
// fetch wave-uniform records with scalar loads (s_load through the scalar cache into SGPRs)
// instead of 64 identical vector loads.
struct Tables {
    const DevIsect *__restrict__ isect;
    const uint32_t *__restrict__ kind;
    const DevShade *__restrict__ shade;
    const DevPrim *__restrict__ prim;
    const DevBound *__restrict__ bound;
    // two-level cull tables (Morton order, groups of 64)
    const DevIsect *__restrict__ isect_s;
    const uint32_t *__restrict__ kind_s;
    const DevBound *__restrict__ bound_s;
    const uint32_t *__restrict__ orig_s;
    const DevBound *__restrict__ gbound;
    const DevIdEntry *__restrict__ idtab; // shapes in stable order of world_id (n1/n2 pass)
    const DevPre *__restrict__ pre;       // per-lane prefilter records, insertion order / sorted order
    const DevPre *__restrict__ pre_s;
};

struct LdsView {
    double *m;      // [cap][12]
    double *prim;   // [cap][4]
    uint32_t *kind; // [cap]
};

DEVI LdsView lds_view(double *base, uint32_t cap) {
    LdsView v;
    v.m = base;
    v.prim = base + (size_t)cap * 12;
    v.kind = reinterpret_cast<uint32_t *>(base + (size_t)cap * 16);
    return v;
}

DEVI void stage_tile(const Tables &T, const LdsView &L, uint32_t base, uint32_t cnt) {
    const double *gm = reinterpret_cast<const double *>(T.isect + base);
    for (uint32_t e = threadIdx.x; e < cnt * 12; e += blockDim.x) L.m[e] = gm[e];
    const double *gp = reinterpret_cast<const double *>(T.prim + base);
    for (uint32_t e = threadIdx.x; e < cnt * 4; e += blockDim.x) L.prim[e] = gp[e];
    for (uint32_t e = threadIdx.x; e < cnt; e += blockDim.x) L.kind[e] = T.kind[base + e];
}

// Skip (two-level cull, closest-hit passes of rays that start at the bundle's apex with unit
// directions): skip(key) is wave-uniform and true when no lane can still be improved by an object
// whose intersections all have t >= key. With it the groups of a 64-group step, and the objects of a
// group, are visited in ascending key order and the walk stops at the first key that is out of
// reach for every lane. Results do not depend on the visiting order: closer() compares (t, index).
struct NoSkip {
    DEVI bool operator()(float) const { return false; }
};
// From the lanes flagged in `mask` take the one with the smallest key (ties: lowest lane).
DEVI int take_min_key(unsigned long long &mask, float key, float &kmin) {
    const uint32_t lane = threadIdx.x & 63u;
    const bool in = (mask >> lane) & 1ull;
    const unsigned kb = in ? __builtin_bit_cast(unsigned, key) : 0xffffffffu; // keys are >= 0: bits order like the values
    const unsigned minbits = ~wave_max_u32(~kb);
    const int sel = (int)__builtin_ctzll(ballot(in && kb == minbits));
    mask &= ~(1ull << sel);
    kmin = __builtin_bit_cast(float, minbits);
    return sel;
}

template <int SRC, bool LANE_FILTER = false, class PP, class F, class SK = NoSkip>
DEVI void for_each_object(const PP &P, const Tables &T, const LdsView &L, bool lane_needs, const Bundle &B, F &&f,
                          V3 fro = V3{0., 0., 0.}, V3 frd = V3{0., 0., 0.}, SK skip = SK{}, unsigned *nfilt = nullptr,
                          unsigned *ngrp = nullptr, unsigned *nobj = nullptr) {
    constexpr bool ORDERED = !__is_same(SK, NoSkip);
    // per-lane prefilter: the pre-inflated records hold while every origin of the pass is within their limit
    bool pre_ok = false;
    double fdd = 0.;
    if constexpr (LANE_FILTER) {
#pragma clang fp contract(fast)
        pre_ok = ballot(lane_needs && !(fabs(fro.x) + fabs(fro.y) + fabs(fro.z) <= P.pre_limit)) == 0ull;
        fdd = frd.x * frd.x + frd.y * frd.y + frd.z * frd.z;
    }
    if constexpr (SRC == SRC_CULL) {
        const unsigned long long needs_mask = ballot(lane_needs); // (lane_needs does not change during the walk)
        // One-level cull (small worlds): up to RTC_OBJ_SLOTS x 64 objects per round, each lane tests one object's sphere of
        // every slot against the wave's bundle (the slots' loads are in flight together: one load -> test -> ballot
        // dependency chain per round instead of one per 64 objects); the ballot masks are walked in ascending
        // (= insertion) order and the survivors get the exact test, their records fetched by uniform index (scalar cache).
        if (ballot(lane_needs) == 0ull) return;
        const uint32_t lane = threadIdx.x & 63u;
        constexpr uint32_t OS = RTC_OBJ_SLOTS;
        for (uint32_t base = 0; base < P.n; base += 64u * OS) {
            unsigned long long masks[OS];
#pragma unroll
            for (uint32_t sl = 0; sl < OS; ++sl) {
                const uint32_t j = base + sl * 64u + lane;
                bool cand = false;
                if (base + sl * 64u < P.n && j < P.n) cand = bundle_touches(B, T.bound[j]);
                masks[sl] = ballot(cand);
            }
#pragma unroll
            for (uint32_t sl = 0; sl < OS; ++sl) {
                unsigned long long mask = masks[sl];
                if constexpr (LANE_FILTER) {
                    if (pre_ok) { // RTC_PRE_BATCH candidates at a time (see ray_touches_pre)
                        while (mask) {
                            uint32_t jx[RTC_PRE_BATCH];
                            unsigned long long bx[RTC_PRE_BATCH];
                            uint32_t cnt = 0;
#pragma unroll
                            for (uint32_t k = 0; k < RTC_PRE_BATCH; ++k) {
                                jx[k] = k ? jx[0] : base + sl * 64u; // (slots past the last candidate repeat the first: a harmless second look)
                                if (mask) {
                                    jx[k] = base + sl * 64u + (uint32_t)__builtin_ctzll(mask);
                                    mask &= mask - 1ull;
                                    cnt = k + 1u;
                                }
                            }
                            DevPre q[RTC_PRE_BATCH]; // all requested before the first is used: one round trip for the batch
#pragma unroll
                            for (uint32_t k = 0; k < RTC_PRE_BATCH; ++k) q[k] = T.pre[jx[k]];
#pragma unroll
                            for (uint32_t k = 0; k < RTC_PRE_BATCH; ++k) {
                                DIAG_FILTER(nfilt);
                                bx[k] = ray_touches_pre_mask(fro, frd, fdd, q[k]) & needs_mask; // (evaluated by every lane: no branch around the loads)
                            }
#pragma unroll
                            for (uint32_t k = 0; k < RTC_PRE_BATCH; ++k) {
                                if (k >= cnt || bx[k] == 0ull) continue;
                                const DevIsect rec = T.isect[jx[k]]; // record and kind requested together, ahead of the callback's branches
                                const uint32_t kd = T.kind[jx[k]];
                                if (!f((int)jx[k], rec.m, kd, (const double *)nullptr)) return;
                            }
                        }
                        continue;
                    }
                }
                while (mask) {
                    const uint32_t jj = base + sl * 64u + (uint32_t)__builtin_ctzll(mask);
                    mask &= mask - 1ull;
                    if constexpr (LANE_FILTER) {
                        DIAG_FILTER(nfilt);
                        if (ballot(lane_needs && ray_touches(fro, frd, T.bound[jj])) == 0ull) continue;
                    }
                    const DevIsect *rec = T.isect + jj;
                    if (!f((int)jj, rec->m, T.kind[jj], (const double *)nullptr)) return;
                }
            }
        }
    } else if constexpr (SRC == SRC_CULL2) {
        // Two-level cull (large worlds) over the Morton-sorted tables. Level 1: group spheres against the wave's
        // bundle, one per lane and slot. Level 2: the 64 objects of surviving groups, one object sphere per lane —
        // RTC_EXPAND_K groups per round, their bounds loaded and tested together (one dependency chain per round).
        // Objects are not visited in insertion order here, so the callback receives the insertion index and the
        // tie-break compares it (closer()).
        if (ballot(lane_needs) == 0ull) return;
        const uint32_t lane = threadIdx.x & 63u;
        constexpr uint32_t SLOTS = RTC_GROUP_SLOTS, K = RTC_EXPAND_K;
        // level 2 for the `cnt` (<= K) groups gidx[0..cnt); returns false when the walk is over (callback said so)
        auto expand = [&](const uint32_t (&gidx)[K], uint32_t cnt) -> bool {
            unsigned long long masks[K];
            float okey[K];
#pragma unroll
            for (uint32_t k = 0; k < K; ++k) {
                bool cand = false;
                okey[k] = 0.f;
                if (k < cnt) {
                    DIAG_FILTER(ngrp);
                    const uint32_t j = gidx[k] * 64u + lane;
                    if (j < P.n) cand = bundle_touches<ORDERED>(B, T.bound_s[j], &okey[k]);
                }
                masks[k] = ballot(cand);
            }
#pragma unroll
            for (uint32_t k = 0; k < K; ++k) {
                unsigned long long mask = masks[k];
                const uint32_t base = (k < cnt ? gidx[k] : 0u) * 64u;
                while (mask) {
                    uint32_t jj;
                    if constexpr (ORDERED) {
                        float kmin;
                        jj = base + (uint32_t)take_min_key(mask, okey[k], kmin);
                        if (skip(kmin)) break;
                    } else {
                        jj = base + (uint32_t)__builtin_ctzll(mask);
                        mask &= mask - 1ull;
                    }
                    DIAG_FILTER(nobj);
                    if constexpr (LANE_FILTER) {
                        DIAG_FILTER(nfilt);
                        if (ballot(lane_needs && (pre_ok ? ray_touches_pre(fro, frd, fdd, T.pre_s[jj]) : ray_touches(fro, frd, T.bound_s[jj]))) == 0ull) continue;
                    }
                    const DevIsect *rec = T.isect_s + jj;
                    if (!f((int)T.orig_s[jj], rec->m, T.kind_s[jj], (const double *)nullptr)) return false;
                }
            }
            return true;
        };
        for (uint32_t gbase = 0; gbase < P.ngroups; gbase += 64u * SLOTS) {
            // one round of level 1: group (gbase + s*64 + lane) for s = 0..SLOTS-1, all loads in flight together
            unsigned kb[SLOTS];               // ordered: key bits (keys are >= 0: they order like unsigned); ~0u = no candidate
            unsigned long long gm[SLOTS];     // unordered: candidate masks
#pragma unroll
            for (uint32_t sl = 0; sl < SLOTS; ++sl) {
                const uint32_t g = gbase + sl * 64u + lane;
                bool gc = false;
                float gkey = 0.f;
                if (gbase + sl * 64u < P.ngroups && g < P.ngroups)
                    gc = bundle_touches<ORDERED>(B, T.gbound[g], &gkey);
                kb[sl] = gc ? __builtin_bit_cast(unsigned, gkey) : 0xffffffffu;
                gm[sl] = ballot(gc);
            }
            if constexpr (ORDERED) {
                // the round's queue: nearest key first over ALL its slots. Each lane keeps its own minimum; the wave
                // minimum picks the lane, that lane's slot is read back and struck out. K groups are taken per expansion.
                for (bool more = true; more;) {
                    uint32_t gidx[K];
                    uint32_t cnt = 0;
#pragma unroll
                    for (uint32_t k = 0; k < K; ++k) {
                        gidx[k] = 0u;
                        if (!more) continue;
                        unsigned mine = kb[0];
                        uint32_t msl = 0;
#pragma unroll
                        for (uint32_t sl = 1; sl < SLOTS; ++sl)
                            if (kb[sl] < mine) { mine = kb[sl]; msl = sl; }
                        const unsigned minbits = ~wave_max_u32(~mine);
                        // queue empty, or ascending keys: the rest is out of reach too
                        if (minbits == 0xffffffffu || skip(__builtin_bit_cast(float, minbits))) { more = false; continue; }
                        const int sel = (int)__builtin_ctzll(ballot(mine == minbits));
                        const uint32_t ssl = (uint32_t)__builtin_amdgcn_readlane((int)msl, sel);
#pragma unroll
                        for (uint32_t sl = 0; sl < SLOTS; ++sl)
                            if (sl == ssl && lane == (uint32_t)sel) kb[sl] = 0xffffffffu;
                        gidx[k] = gbase + ssl * 64u + (uint32_t)sel;
                        cnt = k + 1u;
                    }
                    if (cnt == 0u) break;
                    if (!expand(gidx, cnt)) return;
                }
            } else {
                uint32_t gidx[K];
                uint32_t cnt = 0;
#pragma unroll
                for (uint32_t sl = 0; sl < SLOTS; ++sl) {
                    unsigned long long gmask = gm[sl];
                    while (gmask) {
                        gidx[cnt++] = gbase + sl * 64u + (uint32_t)__builtin_ctzll(gmask);
                        gmask &= gmask - 1ull;
                        if (cnt == K) {
                            if (!expand(gidx, cnt)) return;
                            cnt = 0;
                        }
                    }
                }
                if (cnt != 0u && !expand(gidx, cnt)) return;
            }
        }
    } else if constexpr (SRC == SRC_SMEM) {
        if (ballot(lane_needs) == 0ull) return;
        for (uint32_t j = 0; j < P.n; ++j) {
            const DevIsect *rec = T.isect + j;
            if (!f((int)j, rec->m, T.kind[j], reinterpret_cast<const double *>(T.prim + j))) break;
        }
    } else if constexpr (SRC == SRC_LDS1) {
        if (ballot(lane_needs) == 0ull) return;
        for (uint32_t j = 0; j < P.n; ++j) {
            const uint32_t kind = __builtin_amdgcn_readfirstlane(L.kind[j]);
            if (!f((int)j, L.m + j * 12, kind, L.prim + j * 4)) break;
        }
    } else {
        bool wave_live = ballot(lane_needs) != 0ull;
        for (uint32_t base = 0; base < P.n; base += P.tile_cap) {
            const uint32_t cnt = (P.n - base < P.tile_cap) ? (P.n - base) : P.tile_cap;
            __syncthreads(); // previous tile fully consumed
            stage_tile(T, L, base, cnt);
            __syncthreads();
            if (wave_live) {
                for (uint32_t j = 0; j < cnt; ++j) {
                    const uint32_t kind = __builtin_amdgcn_readfirstlane(L.kind[j]);
                    if (!f((int)(base + j), L.m + j * 12, kind, L.prim + j * 4)) { wave_live = false; break; }
                }
            }
        }
    }
}

// ---- patterns (material.rs:41-45 and the six pattern_at bodies) --------------------------
DEVI V3 pattern_color(const DevShade *S, const double *m_obj, V3 world_point) {
    const V3 op = xpoint(m_obj, world_point);
    const V3 pp = xpoint(S->pat_inv, op);
    const V3 a = mk(S->pat_a[0], S->pat_a[1], S->pat_a[2]);
    const V3 b = mk(S->pat_b[0], S->pat_b[1], S->pat_b[2]);
    switch (S->pattern_kind) {
    case RTC_PATTERN_TEST: return pp;
    case RTC_PATTERN_STRIPE: return (fmod(floor(pp.x), 2.0) == 0.) ? a : b;
    case RTC_PATTERN_GRADIENT: {
        const V3 diff = vsub(b, a); // GradientPattern::new: _diff = b.sub(a)
        return vadd(a, vmul(diff, pp.x - floor(pp.x)));
    }
    case RTC_PATTERN_RING: return (fmod(floor(sqrt(pp.x * pp.x + pp.y * pp.y)), 2.0) == 0.0) ? a : b;
    case RTC_PATTERN_CHECKER: return (fmod(floor(pp.x) + floor(pp.y) + floor(pp.z), 2.0) == 0.0) ? a : b;
    case RTC_PATTERN_GRID:
        return (fabs(pp.x - floor(pp.x)) < 0.01 || fabs(pp.z - floor(pp.z)) < 0.01) ? b : a;
    default: return mk(0., 0., 0.);
    }
}

// Material::lighting material.rs:319-361. lightv = (light.position - point).normalize() is
// the shadow ray's direction (same expression, shape.rs:717-719), passed in.
template <class PP>
DEVI V3 lighting(const PP &P, const DevShade *S, const double *m_obj, V3 point, V3 eyev, V3 normal,
                 V3 lightv, bool in_shadow) {
    // The reference's statement order is effective_color, lightv, ambient, [shadow?] light_dot_normal,
    // diffuse, reflectv, reflect_dot_eye, factor, specular. The values do not depend on that order
    // (no shared rounding), so the geometric scalars and the one expensive call (pow) are done FIRST,
    // while little else is live, and the colour arithmetic afterwards.
    double ldn = 0., kspec = 0.;
    bool lit = false, spec = false;
    if (!in_shadow) {
        ldn = vdot(lightv, normal);
        if (!(ldn < 0.)) {
            lit = true;
            const V3 reflectv = vreflect(vneg(lightv), normal);
            const double rde = vdot(reflectv, eyev);
            if (!(rde <= 0.)) {
                // factor = rde.powf(shininess) (material.rs:355). When specular == 0 and the power is
                // certainly finite and positive (0 < rde <= 1, 0 <= shininess < inf), specular*factor is
                // exactly specular (a signed zero) whatever the power is: skip the ~200-instruction pow.
                const double ks = S->specular, sh = S->shininess;
                const bool trivial = (ks == 0.0) && (rde <= 1.0) && (sh >= 0.0) && (sh < __builtin_inf());
                const double factor = trivial ? 1.0 : pow(rde, sh);
                kspec = ks * factor;
                spec = true;
            }
        }
    }
    asm volatile("" ::: "memory"); // the material colour / pattern loads start here, not above the pow
    const V3 I = mk(P.light_int[0], P.light_int[1], P.light_int[2]);
    const V3 base = (S->pattern_kind != RTC_PATTERN_NONE) ? pattern_color(S, m_obj, point)
                                                          : mk(S->color[0], S->color[1], S->color[2]);
    const V3 eff = vmulv(base, I);
    const V3 ambient = vmul(eff, S->ambient);
    if (in_shadow) return ambient;
    V3 diffuse = mk(0., 0., 0.), specular = mk(0., 0., 0.);
    if (lit) {
        diffuse = vmul(eff, S->diffuse * ldn);
        if (spec) specular = vmul(I, kspec);
    }
    return vadd(vadd(ambient, diffuse), specular);
}

// World::reflectance (Schlick) shape.rs:768-781
DEVI double reflectance(V3 eyev, V3 normal, double n1, double n2) {
    double cosv = vdot(eyev, normal);
    if (n1 > n2) {
        const double n = n1 / n2;
        const double sin2_t = (n * n) * (1.0 - cosv * cosv);
        if (sin2_t > 1.0) return 1.0;
        cosv = sqrt(1.0 - sin2_t);
    }
    const double q = (n1 - n2) / (n1 + n2);
    const double r0 = q * q;
    const double x = 1.0 - cosv;
    const double x2 = x * x;
    return r0 + (1.0 - r0) * (x * (x2 * x2)); // powi(5)
}

// The parameter block lives in the kernarg segment. Its loads are invariant, so LLVM hoists every
// one of them to the kernel entry and then has to carry ~80 SGPRs through the whole kernel (they
// spill into VGPR lanes). KP(P_arg) hands out a fresh, opaque constant-address-space view of the
// same block; loads through it stay in the phase that asked for it.
// (RenderParams is the kernel's FIRST argument, i.e. offset 0 of the kernarg segment.)
typedef const __attribute__((address_space(4))) RenderParams *ParamPtr;
DEVI ParamPtr param_view() {
    unsigned long long a = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(a));
    return (ParamPtr)a;
}
#define KP(arg) (*param_view())

// One suspended shade_hit call (shape.rs:685-700) on a lane's stack. The stack lives in scratch
// (dynamic index), so its size is HBM/L2 traffic: Worlds without transparent materials (REFR ==
// false) recurse along a single chain `surface + color_at(reflected ray) * kr` and need only
// (surface, kr) = 32 bytes per level instead of 152 (1080p reflective north star: 1.67 GB of
// scratch writes per frame with the full frame).
template <bool REFR> struct FrameT;
template <> struct FrameT<true> {
    V3 surface;
    V3 a;           // state 0: origin of the pending refraction ray; state 1: the reflected colour (the ray has left)
    V3 rd;          // state 0: direction of the pending refraction ray
    double kr, tr, R;
    uint8_t rem;    // `remaining` of the color_at call that owns the frame
    uint8_t state;  // 0 waiting for the reflected child, 1 waiting for the refracted child
    uint8_t schlick;
    uint8_t has_refr;
};
template <> struct FrameT<false> {
    V3 surface;
    double kr;
};

// Color::scale(component, 255) color.rs:100-114: `(c * 255.0) as i32` (truncating, saturating,
// NaN -> 0) then clamp to [0, 255].
DEVI unsigned char scale255(double c) {
    const double v = c * 255.0;
    if (!(v >= 0.0)) return 0;   // negative (truncates to <= 0, clamps to 0) or NaN
    if (v >= 255.0) return 255;  // saturates / clamps
    return (unsigned char)(int)v; // v_cvt_i32_f64 truncates toward zero
}

// Offsets of Camera::resample's extra rays (camera.rs:84-92). The reference draws them from
// thread_rng; the documented counter-based stand-in (include/rtc.h, rtc_camera.samples): SplitMix64 of
// ((y*hsize + x) << 16 | draw), top 53 bits -> [0, 1).
DEVI double resample_offset(uint32_t W, uint32_t x, uint32_t y, uint32_t draw) {
    unsigned long long z = ((((unsigned long long)y * W + x) << 16) | draw) + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * 0x1p-53;
}

// shade_hit's final combination shape.rs:692-699
DEVI V3 combine(V3 surface, V3 reflected, V3 refracted, bool schlick, double R) {
    if (schlick) return vadd(surface, vadd(vmul(reflected, R), vmul(refracted, 1.0 - R)));
    return vadd(vadd(surface, reflected), refracted);
}

// ---- the kernel -------------------------------------------------------------------------
// PROBE = true is the rtc_color_at flavour (arbitrary rays in, colours + hit records out); the
// render flavour (PROBE = false) never carries the hit record's extra vectors in registers.
template <int SRC, bool REFL, bool REFR, bool PROBE>
__global__ void __launch_bounds__(RTC_BLOCK_FOR(CULL_LEVEL(SRC), REFL, REFR, PROBE), (REFL ? RTC_WAVES_PER_SIMD_STACK : RTC_WAVES_PER_SIMD))
k_trace(const RenderParams P_arg, const DevIsect *__restrict__ t_isect, const uint32_t *__restrict__ t_kind,
        const DevShade *__restrict__ t_shade, const DevPrim *__restrict__ t_prim, const DevBound *__restrict__ t_bound,
        const DevIsect *__restrict__ t_isect_s, const uint32_t *__restrict__ t_kind_s, const DevBound *__restrict__ t_bound_s,
        const uint32_t *__restrict__ t_orig_s, const DevBound *__restrict__ t_gbound, const DevIdEntry *__restrict__ t_idtab,
        const DevPre *__restrict__ t_pre, const DevPre *__restrict__ t_pre_s) {
    constexpr uint32_t BLOCK = RTC_BLOCK_FOR(CULL_LEVEL(SRC), REFL, REFR, PROBE), TILE_W = RTC_TILE_W_FOR(CULL_LEVEL(SRC), REFL, REFR, PROBE);
    constexpr bool COMPACT = RTC_COMPACT_FOR(CULL_LEVEL(SRC), REFL, REFR, PROBE); // K3: two waves, live rays merged between bounces
    extern __shared__ double lds_raw[];
    constexpr bool LDS_STACK = RTC_LDS_STACK && REFL && !REFR; // 32-byte frames in LDS (one wave per workgroup)
    static_assert(!LDS_STACK || BLOCK == 64 || COMPACT, "the tile is staged over the LDS frame stack: one wave per workgroup, or a barrier first");
    // K3 exchange area: live counts per wave (double-buffered by pass parity) and up to 32 rays in transit
    __shared__ uint32_t k3_cnt[COMPACT ? 4 : 1];
    __shared__ __attribute__((aligned(16))) double k3_xfer[COMPACT ? 7 * 32 : 1];
    __shared__ __attribute__((aligned(16))) double stack_lds[LDS_STACK ? RTC_MAX_STACK * 4 * BLOCK : 2];
    __shared__ __attribute__((aligned(16))) double stage_own[(PROBE || LDS_STACK) ? 2 : 8 * TILE_W * 3];     // the tile, canvas layout
    __shared__ __attribute__((aligned(16))) unsigned char stage_own8[(PROBE || LDS_STACK) ? 16 : 8 * TILE_W * 3];
    // with the LDS stack the tile is staged over it (every lane's stack is empty when its sample is done)
    double *const stage_f64 = LDS_STACK ? stack_lds : stage_own;
    unsigned char *const stage_u8 = LDS_STACK ? reinterpret_cast<unsigned char *>(stack_lds + 8 * TILE_W * 3) : stage_own8;
    const auto &P = KP(P_arg); // set-up view: grid, sizes, mode
    const LdsView L = lds_view(lds_raw, P.tile_cap);
    Tables T;
    T.isect = t_isect; T.kind = t_kind; T.shade = t_shade; T.prim = t_prim; T.bound = t_bound;
    T.isect_s = t_isect_s; T.kind_s = t_kind_s; T.bound_s = t_bound_s; T.orig_s = t_orig_s; T.gbound = t_gbound;
    T.idtab = t_idtab;
    T.pre = t_pre; T.pre_s = t_pre_s;

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    constexpr bool probe = PROBE;

    if constexpr (SRC == SRC_LDS1) { // the whole object table, once per workgroup
        stage_tile(T, L, 0, P.n);
        __syncthreads();
    }
    uint32_t c_primary = 0, c_shadow = 0, c_reflect = 0, c_refract = 0; // wave-uniform
    uint32_t c_resample = 0;            // wave-uniform: pixels that tripped the resample test
    uint32_t c_sky = 0;                 // wave-uniform: primary rays of tiles proven black (sky_tile)
    // A workgroup renders `reps` tiles, one after the other: tile ids blockIdx.x, blockIdx.x + gridDim.x, ... (consecutive
    // workgroups still land on consecutive XCDs). The canvas stores of tile k then drain while tile k+1 is traced (a wave cannot
    // retire before its stores are acknowledged), and the per-wave set-up and the counter atomics are paid once per `reps` tiles.
    uint32_t wg_reps = PROBE ? 1u : P.reps, wg_base = blockIdx.x, wg_stride = gridDim.x;
    if (!PROBE && (P.chunk_wgs[0] | P.chunk_wgs[1] | P.chunk_wgs[2] | P.chunk_wgs[3])) { // guided chunks (RenderParams::chunk_wgs)
        const uint32_t n8 = P.chunk_wgs[0], n4 = P.chunk_wgs[1], n3 = P.chunk_wgs[2], n2 = P.chunk_wgs[3];
        uint32_t b = blockIdx.x;
        if (b < n8) { wg_reps = 8u; wg_base = b; wg_stride = n8; }
        else if ((b -= n8) < n4) { wg_reps = 4u; wg_base = 8u * n8 + b; wg_stride = n4; }
        else if ((b -= n4) < n3) { wg_reps = 3u; wg_base = 8u * n8 + 4u * n4 + b; wg_stride = n3; }
        else if ((b -= n3) < n2) { wg_reps = 2u; wg_base = 8u * n8 + 4u * n4 + 3u * n3 + b; wg_stride = n2; }
        else { wg_reps = 1u; wg_base = 8u * n8 + 4u * n4 + 3u * n3 + 2u * n2 + (b - n2); wg_stride = 0u; }
    }
    const uint32_t reps = wg_reps;
    for (uint32_t rep = 0; rep < reps; ++rep) {
    // workgroup id -> tile: see RTC_TILE_ORDER (XCD balance beats XCD locality here)
    uint32_t bid = wg_base + rep * wg_stride;
    if (!PROBE && bid >= P.total_blocks) break; // (workgroup-uniform)
#if RTC_TILE_ORDER == 0
    {
        const uint32_t nb = gridDim.x, q = nb / 8u, r = nb % 8u, xcd = bid % 8u, k = bid / 8u;
        bid = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + k;
    }
#elif RTC_TILE_ORDER == 2
    bid = gridDim.x - 1u - bid;
#endif

    uint32_t px = 0, py = 0, ray_index = 0;
    uint32_t view = 0, tbid = bid; // which camera of the launch, and the tile's index inside that view
    bool in_range, traced;
    if (probe) {
        ray_index = bid * BLOCK + threadIdx.x;
        in_range = ray_index < P.nrays;
        traced = in_range;
    } else {
        const uint32_t tiles = P.grid_x * P.grid_y;
        if (P.nviews > 1u) { view = bid / tiles; tbid = bid % tiles; }
        const uint32_t bx = tbid % P.grid_x, by = tbid / P.grid_x;
        px = bx * TILE_W + wave * 8u + (lane & 7u);
        py = P.y0 + by * P.band_stride * 8u + (lane >> 3);
        in_range = px < P.W && py < P.y1;
        // Camera::render leaves the last row and column untouched (camera.rs:120-121)
        traced = in_range && !(P.mode == RTC_MODE_RENDER && (px + 1u >= P.W || py + 1u >= P.H));
    }

#ifdef RTC_STAMPS
    unsigned long long stamp_t[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
    bool stamp_secondary = false;
    // per wave: [0] closest passes, [1] closest passes with an unbounded bundle, [2] exact tests in
    // closest passes, [3] shadow passes, [4] shadow passes unbounded, [5] exact tests in shadow passes
    // [8] groups expanded in closest passes, [9] in shadow passes, [10] object-level cull survivors (closest), [11] (shadow)
    unsigned diag_c[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    STAMP(0);

    // Tile rows the binning kernel PROVED black for this view (k_bin_tiles, `rows`): no ray is generated, no pass is run, the
    // tile is stored as Canvas::new left it; the primary rays the reference would have cast are still counted (and reported
    // separately, rtc_stats::rays_primary_proven_miss). One-sample renders only.
    bool sky_tile = false;
    if constexpr (IS_CULL(SRC) && !PROBE) {
        const auto &Pt = KP(P_arg);
        if (Pt.tile_rows != nullptr && Pt.samples == 1u) {
            const uint32_t ity = (Pt.y0 >> 3) + (tbid / Pt.grid_x) * Pt.band_stride;
            const uint32_t first = Pt.tile_rows[2u * view], lastinv = Pt.tile_rows[2u * view + 1u];
            sky_tile = ity < first || ity > ~lastinv; // (no tile of the view is non-empty: first = 0xffffffff)
        }
    }

    // Binned primary pass: this wave's 8x8 tile has a list of the objects its primary rays can touch (k_bin_tiles): lane e
    // holds entry e. (RTC_BIN_HOIST: the 64 entry slots of a tile always exist, the ones past the
    // count hold garbage and are masked in the walk.)
    bool binned = false;
    uint32_t bin_cnt = 0, bin_ent = 0xffffffffu;
    auto tile_lookup = [&]() {
        const auto &Pt = KP(P_arg);
        if (Pt.tile_cnt != nullptr) {
            const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave); // wave-uniform by construction
            const uint32_t itx = (tbid % Pt.grid_x) * (TILE_W / 8u) + wv, ity = (Pt.y0 >> 3) + (tbid / Pt.grid_x) * Pt.band_stride;
            if (itx < Pt.tiles_x && ity < Pt.tiles_y) {
                const size_t tile = (size_t)(view * Pt.tiles_y + ity) * Pt.tiles_x + itx;
#if RTC_BIN_HOIST
                bin_ent = Pt.tile_list[tile * RTC_TILE_LIST_CAP + lane];
                bin_cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)Pt.tile_cnt[tile]);
#else
                bin_cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)Pt.tile_cnt[tile]);
                if (lane < bin_cnt) bin_ent = Pt.tile_list[tile * RTC_TILE_LIST_CAP + lane];
#endif
                binned = bin_cnt <= RTC_TILE_LIST_CAP; // an overflowing list is incomplete: walk instead
            }
        }
    };
#if RTC_BIN_HOIST
    if constexpr (IS_CULL(SRC) && !PROBE) tile_lookup();
#endif

    // render_pixel (camera.rs:94-114): one ray, or the 4 fixed sub-samples followed — for the pixels whose
    // samples differ by more than 0.01 from their mean — by `resample_n` more rays (Camera::resample)
    const bool aa = !probe && P.samples != 1u;
    uint32_t nsamples = aa ? 4u : 1u;   // grows to 4 + resample_n after sample 3 when some lane resamples
    bool lane_resample = false;         // this lane's pixel tripped the test AND the resample is enabled
    // per thread: the four sub-samples (12 doubles) and Color::average_over's running sums (3 doubles)
    double *aa_store = reinterpret_cast<double *>(reinterpret_cast<char *>(lds_raw) + P.aa_lds_off) + threadIdx.x * 15u;
    V3 result = mk(0., 0., 0.);

    typedef FrameT<REFR> Frame;
    Frame stack[(REFL && !LDS_STACK) ? RTC_MAX_STACK : 1];
    // LDS stack: level l, component c of the ray OWNED by home thread `owner` at stack_lds[(l*4 + c) * BLOCK + owner]. Without
    // K3 a ray never leaves its home lane (owner == threadIdx.x); with it the lane that traces a ray pushes onto, and finally
    // unwinds, the owner's stack and leaves the pixel's colour in the owner's level-0 slot.
    uint32_t owner = threadIdx.x;
    double *lstk = stack_lds + owner;

    for (uint32_t s = 0; s < nsamples; ++s) {
        V3 ro, rd;
        bool shared_origin;
        const auto &Pr = KP(P_arg).views[view]; // ray-generation view: this workgroup's camera block
        // ray origin of every primary ray: transform_point(view_inv, (0,0,0)) camera.rs:72
        V3 cam_origin = xpoint(Pr.vinv, mk(0., 0., 0.));
        cam_origin = mk(uniform_f64(cam_origin.x), uniform_f64(cam_origin.y), uniform_f64(cam_origin.z)); // same in every lane
        if (probe) {
            const double *rp = P.rays + (size_t)(in_range ? ray_index : 0u) * 6;
            ro = mk(rp[0], rp[1], rp[2]);
            rd = mk(rp[3], rp[4], rp[5]);
            shared_origin = false;
        } else if (sky_tile) { // proven black: no ray is needed
            ro = cam_origin; rd = mk(0., 0., 1.);
            shared_origin = true;
        } else {
            // Camera::ray_for_pixel_offset camera.rs:64-76; sub-sample offsets camera.rs:98,102-105
            double xo = 0.5, yo = 0.5;
            if (aa) {
                if (s < 4u) { xo = (s & 1u) ? 0.75 : 0.25; yo = (s & 2u) ? 0.75 : 0.25; }
                else { xo = resample_offset(P.W, px, py, 2u * (s - 4u)); yo = resample_offset(P.W, px, py, 2u * (s - 4u) + 1u); }
            }
            const double xoffset = ((double)px + xo) * Pr.pixel_size;
            const double yoffset = ((double)py + yo) * Pr.pixel_size;
            const double world_x = Pr.half_width - xoffset;
            const double world_y = Pr.half_height - yoffset;
            const V3 pixel = xpoint(Pr.vinv, mk(world_x, world_y, -1.));
            ro = cam_origin;
            rd = vnormalize(vsub(pixel, cam_origin));
            shared_origin = true;
        }

        bool tracing = traced && (s < 4u || lane_resample);
        bool first = true; // this ray is the one color_at was called with (depth 0)
        int rem = (int)P.remaining;
        int sp = 0;
        c_primary += popc64(ballot(tracing));
        if (sky_tile) { c_sky += popc64(ballot(tracing)); tracing = false; } // counted as cast (the reference casts them), answered by the proof

        // Worlds without reflective / transparent materials (REFL == false) need exactly one pass per
        // primary ray; otherwise loop until every lane's frame stack has unwound.
        if constexpr (COMPACT) { // every pixel's colour is collected from its owner's level-0 slot; BLACK until a ray says otherwise
            owner = threadIdx.x;
            lstk = stack_lds + owner;
            lstk[0] = 0.; lstk[BLOCK] = 0.; lstk[2 * BLOCK] = 0.;
        }
        uint32_t k3_pass = 0;
        for (bool pass_again = !sky_tile; pass_again;) {
            if constexpr (COMPACT) {
                // K3 — wavefront compaction between bounces. The workgroup's two waves publish how many rays each still
                // traces (__ballot + popcount); once both have some and together they fit one wave, the wave with fewer
                // hands its rays (origin, direction, remaining, stack depth, owner) over through LDS to the idle lanes of the
                // other and stops issuing passes. A ray's frames and its pixel's colour stay with its OWNER (LDS stack,
                // owner-indexed), so results are bit-identical whichever lane traces it. One barrier per pass (the counts
                // are double-buffered by pass parity), one more at the hand-over, which happens at most once.
                const unsigned long long live = ballot(tracing);
                uint32_t *cnt = k3_cnt + ((k3_pass & 1u) << 1);
                ++k3_pass;
                if (lane == 0) cnt[wave] = popc64(live);
                __syncthreads();
                const uint32_t c0 = cnt[0], c1 = cnt[1];
                if (c0 + c1 == 0u) break;
                if (c0 != 0u && c1 != 0u && c0 + c1 <= 64u) {
                    const uint32_t donor = (c1 <= c0) ? 1u : 0u, moving = donor ? c1 : c0;
                    if (wave == donor) {
                        if (tracing) {
                            const uint32_t r = (uint32_t)__builtin_amdgcn_mbcnt_hi((unsigned)(live >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)live, 0u));
                            k3_xfer[r] = ro.x; k3_xfer[32 + r] = ro.y; k3_xfer[64 + r] = ro.z;
                            k3_xfer[96 + r] = rd.x; k3_xfer[128 + r] = rd.y; k3_xfer[160 + r] = rd.z;
                            k3_xfer[192 + r] = __builtin_bit_cast(double, (unsigned long long)(uint32_t)rem | ((unsigned long long)(uint32_t)sp << 8) |
                                                                              ((unsigned long long)owner << 16));
                        }
                        tracing = false;
                    }
                    __syncthreads();
                    if (wave != donor) {
                        const unsigned long long idle = ~live;
                        const uint32_t r = (uint32_t)__builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));
                        if (!tracing && r < moving) {
                            ro = mk(k3_xfer[r], k3_xfer[32 + r], k3_xfer[64 + r]);
                            rd = mk(k3_xfer[96 + r], k3_xfer[128 + r], k3_xfer[160 + r]);
                            const unsigned long long pk = __builtin_bit_cast(unsigned long long, k3_xfer[192 + r]);
                            rem = (int)(pk & 0xffu);
                            sp = (int)((pk >> 8) & 0xffu);
                            owner = (uint32_t)(pk >> 16);
                            lstk = stack_lds + owner;
                            tracing = true;
                        }
                    }
                }
                if (ballot(tracing) == 0ull) continue; // nothing to trace in this wave (handed over, or done): next barrier
            } else if constexpr (REFL) {
                bool any_tracing;
                if constexpr (SRC == SRC_LDSN) any_tracing = __syncthreads_or(tracing ? 1 : 0) != 0;
                else any_tracing = ballot(tracing) != 0ull;
                if (!any_tracing) break;
            } else {
                pass_again = false;
            }

#ifdef RTC_STAMPS
            stamp_secondary = !first; // [1] of a secondary pass = the previous pass's lighting + frame push
#endif
            STAMP(1); // ray generated
            // ---- World::intersect + get_hit (shape.rs:677-683, 220-232), streaming form ----
            double best = __builtin_inf();
            int hidx = -1, hroot = 0;
            Bundle B{}; // every field defined: an undefined field turns into a value carried around the pass loop
            B.off = true;
#if !RTC_BIN_HOIST
            if constexpr (IS_CULL(SRC) && !PROBE) {
                if (shared_origin && first) tile_lookup();
            }
#endif
            const bool use_bins = binned && shared_origin && first; // (binned is false in the probe and brute-force variants)
            if constexpr (IS_CULL(SRC)) {
                if (ballot(tracing) != 0ull && !use_bins) {
                    if (shared_origin && first) B = make_bundle<true, false>(tracing, cam_origin, ro, rd, 0.);
                    else B = make_bundle<false, false>(tracing, cam_origin, ro, rd, 0.);
#ifdef RTC_NO_SECONDARY_CULL
                    if (!(shared_origin && first)) B.off = true;
#endif
#ifdef RTC_NO_PRIMARY_CULL
                    if (shared_origin && first) B.off = true;
#endif
                }
            }
            STAMP(2); // primary bundle built
            DIAG(0, ballot(tracing) != 0ull ? 1u : 0u);
            DIAG(1, (ballot(tracing) != 0ull && B.off) ? 1u : 0u);
            DIAG(12, (ballot(tracing) != 0ull && !first) ? 1u : 0u); // secondary closest passes
#ifdef RTC_STAMPS
            const unsigned diag_c2_before = diag_c[2], diag_c5_before = diag_c[5];
            const bool diag_secondary = !first;
#endif
            if (!IS_CULL(SRC) && shared_origin && first) {
                for_each_object<SRC>(P, T, L, tracing, B, [&](int j, auto m, uint32_t kind, auto pr) {
                    DIAG(2, 1u);
                    if (tracing) closest_prim(kind, m, pr, rd, j, best, hidx, hroot);
                    return true;
                });
#ifndef RTC_NO_LANE_FILTER
            } else if (IS_CULL(SRC) && ((REFL && !(shared_origin && first)) || RTC_PRIMARY_LANE_FILTER(SRC))) {
                // reflection / refraction rays: incoherent, per-lane prefilter before the exact test
#ifdef RTC_EXP_SKIP_UNBOUNDED // (elimination build, profiles/r03_exp_unbounded_walks.log: wrong image, never shipped)
                if (!B.off)
#endif
                for_each_object<SRC, true>(P, T, L, tracing, B, [&](int j, auto m, uint32_t kind, auto pr) {
                    DIAG(2, 1u);
                    if (tracing) closest_world(kind, m, ro, rd, j, best, hidx, hroot);
                    return true;
                }, ro, rd, NoSkip{}, DIAG_PTR(6), DIAG_PTR(8), DIAG_PTR(10));
#endif
            } else if (IS_CULL(SRC) && !PROBE && use_bins) {
                // binned primary pass: the unbounded objects, then the tile's own list (k_bin_tiles) — together
                // every object this tile's rays can touch
                // (the view's primary-ray constants come from the binning kernel's table: only the direction is transformed)
                const auto &Pb = KP(P_arg);
                const DevPrim *__restrict__ vprim = T.prim + (size_t)view * Pb.n;
#if RTC_BIN_PRIM
#define RTC_BINNED_TEST(KIND, M, J) closest_prim(KIND, M, reinterpret_cast<const double *>(vprim + (J)), rd, (int)(J), best, hidx, hroot)
#else
#define RTC_BINNED_TEST(KIND, M, J) closest_world(KIND, M, ro, rd, (int)(J), best, hidx, hroot)
#endif
                for (uint32_t k = 0; k < Pb.n_unb; ++k) {
                    DIAG(2, 1u);
                    const uint32_t jo = T.orig_s[k];
                    if (tracing) RTC_BINNED_TEST(T.kind_s[k], T.isect_s[k].m, jo);
                }
                // the tile's list, nearest first: lane e holds entry e and the lower bound of the distance from the camera to
                // its (inflated) bounding sphere — every intersection of that object has t >= key for these unit-direction
                // rays — and the walk stops at the first key no lane can use any more (as the ordered group walk does)
                if (Pb.bin_packed) {
                    // entry = key's upper 16 bits (truncated: still a lower bound) above the object index: the wave's
                    // minimum IS the next object and its key
                    uint32_t ent = lane < bin_cnt ? bin_ent : 0xffffffffu;
                    for (;;) {
                        const uint32_t e = ~wave_max_u32(~ent);
                        if (e == 0xffffffffu) break;
                        const float kmin = __builtin_bit_cast(float, e & 0xffff0000u);
                        if (ballot(tracing && !(best < (double)kmin)) == 0ull) break;
                        if (ent == e) ent = 0xffffffffu;
                        const uint32_t j = e & 0xffffu;
                        DIAG(2, 1u);
                        if (tracing) RTC_BINNED_TEST(T.kind[j], T.isect[j].m, j);
                    }
                } else { // more than 65 536 objects: plain indices, keys from the bounds
                    uint32_t my_j = 0u;
                    float my_key = 0.f;
                    if (lane < bin_cnt) {
                        my_j = bin_ent;
                        my_key = bound_key(cam_origin, T.bound[my_j]);
                    }
                    unsigned long long lmask = bin_cnt >= 64u ? ~0ull : ((1ull << bin_cnt) - 1ull);
                    while (lmask) {
                        float kmin;
                        const int sel = take_min_key(lmask, my_key, kmin);
                        if (ballot(tracing && !(best < (double)kmin)) == 0ull) break;
                        const uint32_t j = (uint32_t)__builtin_amdgcn_readlane((int)my_j, sel);
                        DIAG(2, 1u);
                        if (tracing) RTC_BINNED_TEST(T.kind[j], T.isect[j].m, j);
                    }
                }
            } else if (SRC == SRC_CULL2 && !PROBE && shared_origin && first) {
                // primary rays of a large world: start at the apex, unit direction -> ordered walk with early stop
                for_each_object<SRC, false>(P, T, L, tracing, B, [&](int j, auto m, uint32_t kind, auto pr) {
                    DIAG(2, 1u);
                    if (tracing) closest_world(kind, m, ro, rd, j, best, hidx, hroot);
                    return true;
                }, ro, rd, [&](float key) { return ballot(tracing && !(best < (double)key)) == 0ull; }, nullptr, DIAG_PTR(8), DIAG_PTR(10));
            } else {
                for_each_object<SRC>(P, T, L, tracing, B, [&](int j, auto m, uint32_t kind, auto pr) {
                    DIAG(2, 1u);
                    if (tracing) closest_world(kind, m, ro, rd, j, best, hidx, hroot);
                    return true;
                }, ro, rd, NoSkip{}, nullptr, DIAG_PTR(8), DIAG_PTR(10));
            }
            const bool hit = tracing && hidx >= 0;
#ifdef RTC_STAMPS
            if (stamp_secondary && B.off) { // [8] (unused otherwise): closest-hit walks of secondary passes WITHOUT a bounded bundle (also part of [3])
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                stamp_t[8] += __builtin_amdgcn_s_memtime() - stamp_last;
            }
#endif
            STAMP(3); // closest hit found
            const auto &Ph = KP(P_arg); // shading view: light
            const V3 lightp = mk(Ph.light_pos[0], Ph.light_pos[1], Ph.light_pos[2]);

            // ---- Intersection::compute_vectors (shape.rs:144-152, 75-96) --------------------
            V3 point = mk(0, 0, 0), eyev = mk(0, 0, 0), normal = mk(0, 0, 0), over = mk(0, 0, 0), under = mk(0, 0, 0),
               reflectv = mk(0, 0, 0), sdir = mk(0, 0, 0);
            double sdist = 0., n1 = 1.0, n2 = 1.0, m_kr = 0., m_tr = 0.;
            bool inside = false;
            const DevShade *S = T.shade + (hit ? hidx : 0);
            const double *m_obj = T.isect[hit ? hidx : 0].m;
            if (hit) {
                point = vadd(ro, vmul(rd, best)); // Ray::position vec.rs:207-209
                eyev = vneg(rd);
                // Shape::normal_at shape.rs:34-40. A plane's normal does not depend on the point: it was
                // evaluated once per object at rtc_world_create (DevShade::plane_n).
                V3 ln = mk(0., 1., 0.);
                const uint32_t kind = S->kind;
                if (kind == RTC_SPHERE) {
                    const V3 lp = xpoint(m_obj, point);
                    ln = mk(lp.x - 0., lp.y - 0., lp.z - 0.);
                } else if (kind == RTC_CUBE) { // Cube::normal_at_local shape.rs:601-610
                    const V3 lp = xpoint(m_obj, point);
                    const double ax = fabs(lp.x), ay = fabs(lp.y), az = fabs(lp.z);
                    const double maxc = fmax(ax, fmax(ay, az));
                    if (maxc == ax) ln = mk(lp.x, 0., 0.);
                    else if (maxc == ay) ln = mk(0., lp.y, 0.);
                    else ln = mk(0., 0., lp.z);
                }
                if (kind == RTC_PLANE) normal = mk(S->plane_n[0], S->plane_n[1], S->plane_n[2]);
                else normal = vnormalize(xvector3(S->nt, ln));
                inside = vdot(normal, eyev) < 0.0;
                if (inside) normal = vneg(normal);
                over = vadd(point, vmul(normal, RTC_EPSILON));
                under = vsub(point, vmul(normal, RTC_EPSILON));
                reflectv = vreflect(rd, normal);
                m_kr = S->reflective;
                m_tr = S->transparency;
                // shadow ray: is_shadowed_by_light shape.rs:716-720
                const V3 v = vsub(lightp, over);
                sdist = sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
                sdir = mk(v.x / sdist, v.y / sdist, v.z / sdist);
            }

            // ---- compute_refractive (shape.rs:115-141), open-set form, transparent hits only ----
            if constexpr (REFR) {
                const bool need = hit && m_tr != 0.0;
                bool any_need;
                if constexpr (SRC == SRC_LDSN) any_need = __syncthreads_or(need ? 1 : 0) != 0;
                else any_need = ballot(need) != 0ull;
                if (any_need) {
                    // `containers` is keyed by world_id (shape.rs:127): every entry that precedes the hit
                    // entry in list order toggles its id. An id is present iff an odd number of its
                    // entries precede the hit; the element standing for it is the LAST of them (the one
                    // that pushed it back in); containers.last = the present id whose such entry is
                    // latest in list order (max (t, shape index)). Shapes are walked grouped by id
                    // (idtab: stable order of world_id, wave-uniform scalar loads), one class
                    // accumulator at a time. Unique ids (World::add_shape's own numbering up to 255
                    // shapes): a class is one shape and this is the open set of SURVEY.md App. A.6;
                    // shared ids (the reference's u8 counter wraps, shape.rs:287,661-667): still the
                    // literal walk. Never culled: entries with t < 0 count.
                    bool have_all = false, have_oth = false;
                    double key_all = 0., key_oth = 0.;
                    int idx_all = -1, idx_oth = -1;
                    const uint32_t hid = S->world_id;
                    uint32_t cnt_h = 0, cnt = 0;
                    double last_t = 0.;
                    int last_s = -1;
                    auto fold = [&](uint32_t id) { // class `id` is complete
                        if (id == hid) cnt_h = cnt;
                        if (cnt & 1u) {
                            if (!have_all || last_t > key_all || (last_t == key_all && last_s > idx_all)) { have_all = true; key_all = last_t; idx_all = last_s; }
                            if (id != hid && (!have_oth || last_t > key_oth || (last_t == key_oth && last_s > idx_oth))) { have_oth = true; key_oth = last_t; idx_oth = last_s; }
                        }
                        cnt = 0;
                        last_s = -1;
                    };
                    const uint32_t nobj = P.n;
                    uint32_t cur_id = 0;
                    for (uint32_t k = 0; k < nobj; ++k) {
                        const DevIdEntry e = T.idtab[k];
                        if (k > 0 && e.id != cur_id) fold(cur_id);
                        cur_id = e.id;
                        if (need) {
                            const int j = (int)e.index;
                            const double *m = T.isect[j].m;
                            const V3 o = xpoint(m, ro);
                            const V3 d = xvector(m, rd);
                            double t0 = 0., t1 = 0.;
                            const int cnt_e = shape_entries<false>(T.kind[j], o, d, 0., t0, t1);
                            bool p1, p2;
                            if (j == hidx) { p1 = (hroot == 1); p2 = false; }
                            else {
                                p1 = t0 < best || (t0 == best && j < hidx);
                                p2 = t1 < best || (t1 == best && j < hidx);
                            }
                            p1 = p1 && cnt_e >= 1;
                            p2 = p2 && cnt_e == 2;
                            if (p1 || p2) {
                                const double te = p2 ? t1 : t0; // this shape's last entry before the hit
                                cnt += (p1 ? 1u : 0u) + (p2 ? 1u : 0u);
                                if (last_s < 0 || te > last_t || (te == last_t && j >= last_s)) { last_t = te; last_s = j; }
                            }
                        }
                    }
                    if (nobj) fold(cur_id);
                    if (need) {
                        n1 = have_all ? T.shade[idx_all].refractive_index : 1.0;
                        if (cnt_h & 1u) n2 = have_oth ? T.shade[idx_oth].refractive_index : 1.0; // the hit entry removes its id
                        else n2 = S->refractive_index;                                            // the hit entry pushes its id
                    }
                }
            }

            // ---- is_shadowed (shape.rs:712-727): any-hit with early exit ----------------------
            STAMP(4); // hit record + shadow ray
            bool sh_pending = hit, shadowed = false;
#ifdef RTC_EXP_SKIP_SECONDARY_SHADOW // (elimination build, as above)
            if (!first) sh_pending = false;
#endif
            c_shadow += popc64(ballot(hit));
            Bundle Bs{};
            Bs.off = true;
            // Light-space shadow lists, small worlds (one-level cull): the cells' lists are a handful of objects, so they
            // are the candidates themselves — no shadow bundle, no bound tests: per listed object (each once: a 256-bit
            // wave-uniform "done" set, cells next to each other list the same objects) the per-lane prefilter and the
            // exact test.
            bool listed = false;
            if constexpr (SRC == SRC_CULL) {
                const auto &Pl = KP(P_arg);
                // (Pl.n <= 256: the "done" set below has one bit per object; a larger World only gets here with the one-level
                // cull FORCED, RTC_SRC=3, and then takes the bundle walk)
                if (Pl.light_cnt != nullptr && Pl.n <= 256u && ballot(hit) != 0ull && ballot(hit && !(sdist <= Pl.light_reach)) == 0ull) {
                    static_assert(RTC_LIGHT_LIST_CAP_SMALL <= 16u, "a quarter of the wave holds one cell's list");
                    // Round trips, not instructions, are what this path costs: the cells' counters come with ONE
                    // per-lane load (every lane asks for its own cell's), the lists of up to four distinct cells with ONE more
                    // (lanes 16c..16c+15 hold cell c's entries); only the listed objects' own records are fetched one by one.
                    const uint32_t cid = hit ? light_cell(vneg(sdir)) : 0u;
                    const uint32_t ccnt = hit ? Pl.light_cnt[cid] : 0u;
                    // a cell whose list overflowed is incomplete; hit points scattered over more than four cells (shadow rays
                    // of secondary hits) are served better by the bundle cull: fall back
                    listed = ballot(hit && ccnt > Pl.light_cap) == 0ull && Pl.light_cap <= 16u;
                    uint32_t cells[4] = {0u, 0u, 0u, 0u}, cnts[4] = {0u, 0u, 0u, 0u}, ncells = 0;
                    for (unsigned long long todo = ballot(hit); todo && listed;) {
                        const int l = (int)__builtin_ctzll(todo);
                        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cid, l);
                        todo &= ~ballot(hit && cid == c);
                        if (ncells == 4u) { listed = false; break; }
                        const uint32_t cn = (uint32_t)__builtin_amdgcn_readlane((int)ccnt, l);
#pragma unroll
                        for (uint32_t k = 0; k < 4u; ++k)
                            if (k == ncells) { cells[k] = c; cnts[k] = cn; }
                        ++ncells;
                    }
                    if (listed) {
                        for (uint32_t k = 0; k < Pl.n_unb && ballot(sh_pending) != 0ull; ++k) { // unbounded objects: never listed
                            DIAG(5, 1u);
                            if (sh_pending && occludes_world(T.kind_s[k], T.isect_s[k].m, over, sdir, sdist)) { shadowed = true; sh_pending = false; }
                        }
                        const uint32_t ci = lane >> 4, ei = lane & 15u;
                        const uint32_t myc = ci == 0u ? cells[0] : ci == 1u ? cells[1] : ci == 2u ? cells[2] : cells[3];
                        const uint32_t mycnt = ci == 0u ? cnts[0] : ci == 1u ? cnts[1] : ci == 2u ? cnts[2] : cnts[3];
                        const bool have = ci < ncells && ei < mycnt;
                        const uint32_t ent = have ? Pl.light_list[(size_t)myc * Pl.light_cap + ei] : 0u;
                        // pre-inflated prefilter records hold while every origin is within their limit (DevPre)
                        bool pre_ok;
                        double sdd;
                        {
#pragma clang fp contract(fast)
                            pre_ok = ballot(hit && !(fabs(over.x) + fabs(over.y) + fabs(over.z) <= Pl.pre_limit)) == 0ull;
                            sdd = sdir.x * sdir.x + sdir.y * sdir.y + sdir.z * sdir.z;
                        }
                        unsigned long long done[4] = {0ull, 0ull, 0ull, 0ull}; // one bit per object (n <= 256): neighbouring cells list the same objects
                        for (unsigned long long vm = ballot(have); vm && ballot(sh_pending) != 0ull;) {
                            const uint32_t j = (uint32_t)__builtin_amdgcn_readlane((int)ent, (int)__builtin_ctzll(vm));
                            vm &= vm - 1ull;
                            const unsigned long long bit = 1ull << (j & 63u);
                            unsigned long long &word = done[(j >> 6) & 3u];
                            if (word & bit) continue;
                            word |= bit;
                            DIAG_FILTER(DIAG_PTR(7));
                            bool t;
                            if (pre_ok) { const DevPre q = T.pre[j]; t = ray_touches_pre(over, sdir, sdd, q); }
                            else t = ray_touches(over, sdir, T.bound[j]);
                            if (ballot(sh_pending & t) == 0ull) continue;
                            DIAG(5, 1u);
                            const DevIsect rec = T.isect[j]; // record and kind requested together
                            const uint32_t kd = T.kind[j];
                            if (sh_pending && occludes_world(kd, rec.m, over, sdir, sdist)) { shadowed = true; sh_pending = false; }
                        }
                    }
                }
            }
            if constexpr (IS_CULL(SRC)) {
                // the segment over_point -> light, walked from the light: apex = light (shared)
                if (ballot(hit) != 0ull && !listed) Bs = make_bundle<true, true>(hit, lightp, lightp, vneg(sdir), sdist);
#ifdef RTC_NO_SHADOW_CULL
                Bs.off = true;
#endif
            }
            STAMP(5); // shadow bundle built
            DIAG(3, ballot(hit) != 0ull ? 1u : 0u);
            DIAG(4, (ballot(hit) != 0ull && Bs.off) ? 1u : 0u);
            // Light-space shadow lists (two-level worlds): instead of walking every group sphere, filter the lists of the
            // direction cells (around the light) this wave's segments fall in with the wave's own shadow bundle.
            if constexpr (SRC == SRC_CULL2) {
                const auto &Pl = KP(P_arg);
                if (Pl.light_cnt != nullptr && !Bs.off && Bs.tmax <= Pl.light_reach && ballot(hit) != 0ull) {
                    const uint32_t cid = hit ? light_cell(vneg(sdir)) : 0u;
                    const uint32_t ccnt = hit ? Pl.light_cnt[cid] : 0u;          // every lane its own cell's counter: one round trip
                    listed = ballot(hit && ccnt > Pl.light_cap) == 0ull;       // a cell whose list overflowed is incomplete: walk instead
                    if (listed) {
                        for (uint32_t k = 0; k < Pl.n_unb && ballot(sh_pending) != 0ull; ++k) { // unbounded objects: never listed
                            DIAG(5, 1u);
                            if (sh_pending && occludes_world(T.kind_s[k], T.isect_s[k].m, over, sdir, sdist)) { shadowed = true; sh_pending = false; }
                        }
                        bool pre_ok; // pre-inflated prefilter records hold while every origin is within their limit (DevPre)
                        double sdd;
                        {
#pragma clang fp contract(fast)
                            pre_ok = ballot(hit && !(fabs(over.x) + fabs(over.y) + fabs(over.z) <= Pl.pre_limit)) == 0ull;
                            sdd = sdir.x * sdir.x + sdir.y * sdir.y + sdir.z * sdir.z;
                        }
                        for (unsigned long long todo = ballot(hit); todo && ballot(sh_pending) != 0ull;) {
                            const int cl = (int)__builtin_ctzll(todo);
                            const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cid, cl);
                            const uint32_t nl = (uint32_t)__builtin_amdgcn_readlane((int)ccnt, cl);
                            todo &= ~ballot(hit && cid == c);
                            const uint32_t *ll = Pl.light_list + (size_t)c * Pl.light_cap;
                            for (uint32_t base = 0; base < nl && ballot(sh_pending) != 0ull; base += 64u) {
                                const uint32_t e = base + lane;
                                uint32_t idx = 0u;
                                bool cand = false;
                                if (e < nl) { idx = ll[e]; cand = bundle_touches(Bs, T.bound[idx]); }
                                DIAG_FILTER(DIAG_PTR(9));
                                unsigned long long mask = ballot(cand);
                                while (mask && ballot(sh_pending) != 0ull) { // survivors two at a time (RTC_PRE_BATCH, as for_each_object)
                                    const uint32_t j0 = (uint32_t)__builtin_amdgcn_readlane((int)idx, (int)__builtin_ctzll(mask));
                                    mask &= mask - 1ull;
                                    uint32_t j1 = j0;
                                    const bool two = mask != 0ull;
                                    if (two) { j1 = (uint32_t)__builtin_amdgcn_readlane((int)idx, (int)__builtin_ctzll(mask)); mask &= mask - 1ull; }
                                    DIAG_FILTER(DIAG_PTR(11));
                                    unsigned long long b0, b1;
                                    if (pre_ok) {
                                        const DevPre q0 = T.pre[j0], q1 = T.pre[j1];
                                        const bool t0 = ray_touches_pre(over, sdir, sdd, q0), t1 = ray_touches_pre(over, sdir, sdd, q1);
                                        b0 = ballot(sh_pending & t0);
                                        b1 = two ? ballot(sh_pending & t1) : 0ull;
                                    } else {
                                        b0 = ballot(sh_pending && ray_touches(over, sdir, T.bound[j0]));
                                        b1 = two ? ballot(sh_pending && ray_touches(over, sdir, T.bound[j1])) : 0ull;
                                    }
                                    if (b0 != 0ull) {
                                        DIAG(5, 1u);
                                        const DevIsect rec = T.isect[j0];
                                        const uint32_t kd = T.kind[j0];
                                        if (sh_pending && occludes_world(kd, rec.m, over, sdir, sdist)) { shadowed = true; sh_pending = false; }
                                    }
                                    if (b1 != 0ull && ballot(sh_pending) != 0ull) {
                                        DIAG(5, 1u);
                                        const DevIsect rec = T.isect[j1];
                                        const uint32_t kd = T.kind[j1];
                                        if (sh_pending && occludes_world(kd, rec.m, over, sdir, sdist)) { shadowed = true; sh_pending = false; }
                                    }
                                }
                            }
                        }
                    }
                }
            }
            if (!listed)
            for_each_object<SRC, RTC_SHADOW_LANE_FILTER(SRC, REFL)>(P, T, L, sh_pending, Bs, [&](int j, auto m, uint32_t kind, auto pr) {
                DIAG(5, 1u);
                if (sh_pending) {
                    if (occludes_world(kind, m, over, sdir, sdist)) { shadowed = true; sh_pending = false; }
                }
                return ballot(sh_pending) != 0ull;
            }, over, sdir, NoSkip{}, DIAG_PTR(7), DIAG_PTR(9), DIAG_PTR(11));

            STAMP(6); // shadow resolved
#ifdef RTC_STAMPS
            if (diag_secondary) { // [13] exact tests in secondary closest passes, [14] their shadow passes, [15] exact tests in those
                diag_c[13] += diag_c[2] - diag_c2_before;
                diag_c[14] += ballot(hit) != 0ull ? 1u : 0u;
                diag_c[15] += diag_c[5] - diag_c5_before;
            }
#endif
            // keep the material / pattern loads of the lighting stage BELOW the shadow loop: hoisted
            // above it they stay live through the loop and cost a wave per SIMD in occupancy
            asm volatile("" ::: "memory");
            if constexpr (PROBE) if (first && in_range && P.hits) {
                rtc_hit *H = reinterpret_cast<rtc_hit *>(P.hits) + ray_index;
                H->hit_index = hit ? hidx : -1;
                H->inside = inside ? 1u : 0u;
                H->shadowed = shadowed ? 1u : 0u;
                H->_pad = 0u;
                H->t = hit ? best : 0.;
                H->point[0] = point.x; H->point[1] = point.y; H->point[2] = point.z;
                H->over_point[0] = over.x; H->over_point[1] = over.y; H->over_point[2] = over.z;
                H->under_point[0] = under.x; H->under_point[1] = under.y; H->under_point[2] = under.z;
                H->eyev[0] = eyev.x; H->eyev[1] = eyev.y; H->eyev[2] = eyev.z;
                H->normal[0] = normal.x; H->normal[1] = normal.y; H->normal[2] = normal.z;
                H->reflectv[0] = reflectv.x; H->reflectv[1] = reflectv.y; H->reflectv[2] = reflectv.z;
                H->n1 = n1; H->n2 = n2;
            }

            // ---- shade_hit (shape.rs:685-700) ------------------------------------------------
            V3 val = mk(0., 0., 0.); // value returned by the color_at call that just finished
            bool have_val = false;
            bool l_refl = false, l_refr = false; // this lane launched a reflection / refraction ray
            if (tracing && !hit) { // background_color: BLACK shape.rs:652-653
                have_val = true;
            } else if (hit) {
                const V3 surface = lighting(KP(P_arg), S, m_obj, over, eyev, normal, sdir, shadowed);
                bool want_refl = false, want_refr = false;
                V3 fr_o = mk(0, 0, 0), fr_d = mk(0, 0, 0);
                if constexpr (REFL) want_refl = (rem != 0) && (m_kr > 0.); // reflected_color shape.rs:730
                if constexpr (REFR) {
                    if (rem != 0 && m_tr != 0.0) { // refracted_color shape.rs:751-766
                        const double n_ratio = n1 / n2;
                        const double cos_i = vdot(eyev, normal);
                        const double sin2_t = (n_ratio * n_ratio) * (1.0 - cos_i * cos_i);
                        if (!(sin2_t > 1.0)) {
                            const double cos_t = sqrt(1.0 - sin2_t);
                            fr_d = vsub(vmul(normal, n_ratio * cos_i - cos_t), vmul(eyev, n_ratio));
                            fr_o = under;
                            want_refr = true;
                        }
                    }
                }
                bool schlick = false;
                double R = 0.;
                if constexpr (REFR) {
                    if (m_kr > 0.0 && m_tr > 0.0) { schlick = true; R = reflectance(eyev, normal, n1, n2); }
                }
                if (!want_refl && !want_refr) {
                    val = combine(surface, mk(0., 0., 0.), mk(0., 0., 0.), schlick, R);
                    have_val = true;
                } else {
                    if constexpr (REFR) {
                        Frame &F = stack[sp];
                        F.surface = surface;
                        F.a = fr_o; F.rd = fr_d;
                        F.kr = m_kr; F.tr = m_tr; F.R = R;
                        F.rem = (uint8_t)rem;
                        F.state = want_refl ? 0 : 1;
                        F.schlick = schlick ? 1 : 0;
                        F.has_refr = want_refr ? (want_refl ? 1 : 2) : 0; // 2: refraction only, no reflected colour will arrive
                        ++sp;
                        if (want_refl) { ro = over; rd = reflectv; l_refl = true; } // Ray::new(over_point, reflectv) shape.rs:734
                        else { ro = fr_o; rd = fr_d; l_refr = true; }               // Ray::new(under_point, direction) shape.rs:764
                        rem = rem - 1;
                    } else if constexpr (REFL) { // reflection only: want_refl holds here
                        if constexpr (LDS_STACK) {
                            double *f = lstk + (uint32_t)sp * 4u * BLOCK;
                            f[0] = surface.x; f[BLOCK] = surface.y; f[2 * BLOCK] = surface.z; f[3 * BLOCK] = m_kr;
                        } else {
                            Frame &F = stack[sp];
                            F.surface = surface;
                            F.kr = m_kr;
                        }
                        ++sp;
                        ro = over; rd = reflectv; l_refl = true;                     // shape.rs:734
                        rem = rem - 1;
                    }
                }
            }
            first = false;

            // ---- return `val` to the callers (unwind) ------------------------------------------
            if (have_val) {
                bool relaunched = false;
                if constexpr (REFL && !REFR) {
                    while (sp > 0) { // surface + reflected + refracted(BLACK), innermost call first shape.rs:692-699
                        if constexpr (LDS_STACK) {
                            const double *f = lstk + (uint32_t)(sp - 1) * 4u * BLOCK;
                            val = combine(mk(f[0], f[BLOCK], f[2 * BLOCK]), vmul(val, f[3 * BLOCK]), mk(0., 0., 0.), false, 0.);
                        } else {
                            const Frame &F = stack[sp - 1];
                            val = combine(F.surface, vmul(val, F.kr), mk(0., 0., 0.), false, 0.);
                        }
                        --sp;
                    }
                }
                if constexpr (REFR) {
                    while (sp > 0) {
                        Frame &F = stack[sp - 1];
                        if (F.state == 0) {
                            const V3 reflected = vmul(val, F.kr); // c.mul_f64(reflectiveness) shape.rs:736
                            if (F.has_refr) { // now refracted_color's recursion shape.rs:764-765
                                F.state = 1;
                                ro = F.a; rd = F.rd;
                                F.a = reflected;     // the pending ray has left the frame: its slot keeps the reflected colour
                                rem = (int)F.rem - 1;
                                relaunched = true;
                                l_refr = true;
                                break;
                            }
                            val = combine(F.surface, reflected, mk(0., 0., 0.), F.schlick != 0, F.R);
                            --sp;
                        } else {
                            // a frame that never launched a reflection ray (state 1 from the start) holds the refraction
                            // ray's origin in `a`; its reflected colour is BLACK (reflected_color shape.rs:730-732)
                            const V3 reflected = F.has_refr == 2 ? mk(0., 0., 0.) : F.a;
                            const V3 refracted = vmul(val, F.tr); // shape.rs:765
                            val = combine(F.surface, reflected, refracted, F.schlick != 0, F.R);
                            --sp;
                        }
                    }
                }
                if constexpr (REFL) {
                    if (!relaunched) {
                        if constexpr (COMPACT) { // the owner's stack is unwound: its level-0 slot now carries the pixel's colour
                            lstk[0] = val.x; lstk[BLOCK] = val.y; lstk[2 * BLOCK] = val.z;
                        } else {
                            result = val;
                        }
                        tracing = false;
                    }
                }
            }
            if constexpr (REFL) {
                c_reflect += popc64(ballot(l_refl));
                c_refract += popc64(ballot(l_refr));
            } else {
                result = val; // BLACK for lanes that traced nothing or missed (val starts at 0)
            }
        }

        if constexpr (COMPACT) {
            // both waves are out of the pass loop (the break is taken on the same counts): collect every pixel's colour
            // from its own slot; the second barrier frees the stack area for the tile that is staged over it
            const double *mine = stack_lds + threadIdx.x;
            result = mk(mine[0], mine[BLOCK], mine[2 * BLOCK]);
            __syncthreads();
        }
        if constexpr (PROBE) {
            if (in_range) {
                double *o = KP(P_arg).out + (size_t)ray_index * 3;
                o[0] = result.x;
                o[1] = result.y;
                o[2] = result.z;
            }
        } else {
            // The workgroup's (TILE_W x 8)-pixel tile is staged in LDS (row-major, exactly the canvas layout of
            // the tile) and written out by the whole workgroup: with TILE_W = 16 each tile row is 384 contiguous
            // bytes of the f64 canvas (three full 128-byte lines) and 48 contiguous bytes of the 8-bit frame,
            // stored 16 bytes per lane. Direct per-pixel stores (3 x 8 B at a 24 B stride, 3 single
            // bytes) cost 1.6x the algorithmic bytes in HBM write traffic (rocprofv3 WRITE_SIZE).
            // Between AA samples Color::average_over's running sums (color.rs:128-139: reds = ((0 + c0) + c1) + ...)
            // wait in the thread's part of the dynamic LDS block (the tile may be staged over the frame stack).
            const uint32_t tx = wave * 8u + (lane & 7u), ty = lane >> 3; // position inside the tile
            double *slot = stage_f64 + (ty * TILE_W + tx) * 3u;
            if (!traced) result = mk(0., 0., 0.); // Canvas::new BLACK canvas.rs:37-41
            if (aa) {
                // aa_store[12..14] hold Color::average_over's running sums (reds = ((0 + c0) + c1) + ..., color.rs:128-139)
                // while a lane still collects samples, and the pixel's final colour afterwards
                const bool collecting = s < 4u || lane_resample;
                V3 acc = (s == 0u) ? mk(0., 0., 0.) : mk(aa_store[12], aa_store[13], aa_store[14]);
                if (collecting) acc = vadd(acc, result);
                if (s < 4u) { aa_store[s * 3u] = result.x; aa_store[s * 3u + 1u] = result.y; aa_store[s * 3u + 2u] = result.z; }
                if (s == 3u) {
                    // average = Color::average_over(&sample); resample if any |c - average| > 0.01 (camera.rs:106-111,
                    // Color::distance_from color.rs:122-126: sqrt of the sum of squares)
                    const V3 mean = mk(acc.x / 4., acc.y / 4., acc.z / 4.);
                    bool trip = false;
                    for (uint32_t i = 0; i < 4u; ++i) {
                        const double dr = aa_store[i * 3u] - mean.x, dg = aa_store[i * 3u + 1u] - mean.y, db = aa_store[i * 3u + 2u] - mean.z;
                        trip = trip || (sqrt(dr * dr + dg * dg + db * db) > 0.01);
                    }
                    trip = trip && traced;
                    c_resample += popc64(ballot(trip));
                    const auto &Pa = KP(P_arg);
                    lane_resample = trip && Pa.resample_n != 0u;
                    bool more;
                    if constexpr (SRC == SRC_LDSN || COMPACT) more = __syncthreads_or(lane_resample ? 1 : 0) != 0; // same pass count for every wave
                    else more = ballot(lane_resample) != 0ull;
                    if (more) nsamples = 4u + Pa.resample_n;
                    if (!lane_resample) acc = mean; // final: the mean of the four
                }
                if (s >= 4u && s + 1u == nsamples && lane_resample) { // Color::average_over(&samples) over 4 + n
                    const double l = (double)nsamples;
                    acc = mk(acc.x / l, acc.y / l, acc.z / l);
                }
                result = acc;
                aa_store[12] = acc.x; aa_store[13] = acc.y; aa_store[14] = acc.z;
            }
            if (s + 1u == nsamples) {
                slot[0] = result.x;
                slot[1] = result.y;
                slot[2] = result.z;
                const auto &Po = KP(P_arg); // output view
                const bool want8 = Po.out8 != nullptr;
                if (want8) {
                    unsigned char *q = stage_u8 + (ty * TILE_W + tx) * 3u;
                    q[0] = scale255(result.x);
                    q[1] = scale255(result.y);
                    q[2] = scale255(result.z);
                }
#ifdef RTC_DIAG_NO_STORE
                if (Po.W == 0xffffffffu) // never true: keeps the code, skips the stores (diagnosis builds only)
#endif
                if constexpr (RTC_WAVE_OUTPUT(REFL)) {
                // Each wave stores its own 8x8 part of the tile (no workgroup barrier: a wave that is
                // done retires without waiting for the slowest of its three neighbours). Its LDS
                // region is written and read by this wave only; LDS operations of one wave execute
                // in order, the fences only stop the compiler from reordering them.
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t px0 = (tbid % Po.grid_x) * TILE_W + wave * 8u, py0 = Po.y0 + (tbid / Po.grid_x) * Po.band_stride * 8u,
                               orow0 = view * Po.view_rows + (tbid / Po.grid_x) * 8u; // first image row / first output row of the tile
                const uint32_t cols = (px0 >= Po.W) ? 0u : ((Po.W - px0 < 8u) ? (Po.W - px0) : 8u); // valid pixels per row
                const uint32_t rows = (Po.y1 - py0 < 8u) ? (Po.y1 - py0) : 8u;                       // valid tile rows
                const size_t row_bytes = (size_t)Po.W * 24u;
                const double *src = stage_f64 + wave * 24u;
                // a row of the wave's part is 192 contiguous bytes of the canvas: 12 pieces of 16 bytes
                const bool wide = cols == 8u && (row_bytes % 16u) == 0 && ((size_t)Po.out % 16u) == 0;
                if (Po.out == nullptr) { // 8-bit frame only (rtc_render_rgb8): nothing of the f64 canvas leaves the chip
                } else if (wide) {
                    typedef double __attribute__((ext_vector_type(2))) d2;
                    for (uint32_t c = lane; c < rows * 12u; c += 64u) {
                        const uint32_t r = c / 12u, k = c % 12u;
                        const d2 v = *reinterpret_cast<const d2 *>(src + r * (TILE_W * 3u) + k * 2u);
                        char *dst = reinterpret_cast<char *>(Po.out) + (size_t)(orow0 + r) * row_bytes + (size_t)px0 * 24u + k * 16u;
                        RTC_CANVAS_STORE(reinterpret_cast<d2 *>(dst), v);
                    }
                } else {
                    for (uint32_t c = lane; c < rows * cols * 3u; c += 64u) {
                        const uint32_t r = c / (cols * 3u), k = c % (cols * 3u);
                        Po.out[((size_t)(orow0 + r) * Po.W + px0) * 3u + k] = src[r * (TILE_W * 3u) + k];
                    }
                }
                if (want8) {
                    const size_t row8 = (size_t)Po.W * 3u;
                    const unsigned char *src8 = stage_u8 + wave * 24u;
                    // 24 contiguous bytes per row: 3 pieces of 8 bytes
                    const bool wide8 = cols == 8u && (row8 % 8u) == 0 && ((size_t)Po.out8 % 8u) == 0;
                    if (wide8) {
                        typedef unsigned __attribute__((ext_vector_type(2))) u2;
                        for (uint32_t c = lane; c < rows * 3u; c += 64u) {
                            const uint32_t r = c / 3u, k = c % 3u;
                            const u2 v = *reinterpret_cast<const u2 *>(src8 + r * (TILE_W * 3u) + k * 8u);
                            RTC_CANVAS_STORE(reinterpret_cast<u2 *>(Po.out8 + (size_t)(orow0 + r) * row8 + (size_t)px0 * 3u + k * 8u), v);
                        }
                    } else {
                        for (uint32_t c = lane; c < rows * cols * 3u; c += 64u) {
                            const uint32_t r = c / (cols * 3u), k = c % (cols * 3u);
                            Po.out8[((size_t)(orow0 + r) * Po.W + px0) * 3u + k] = src8[r * (TILE_W * 3u) + k];
                        }
                    }
                }
                } else {
                __syncthreads();
                const uint32_t px0 = (tbid % Po.grid_x) * TILE_W, py0 = Po.y0 + (tbid / Po.grid_x) * Po.band_stride * 8u,
                               orow0 = view * Po.view_rows + (tbid / Po.grid_x) * 8u; // first image row / first output row of the tile
                const uint32_t cols = (Po.W - px0 < TILE_W) ? (Po.W - px0) : TILE_W;      // valid pixels per tile row
                const uint32_t rows = (Po.y1 - py0 < 8u) ? (Po.y1 - py0) : 8u;      // valid tile rows
                // f64 canvas: 16-byte pieces when every tile row is whole and 16-byte aligned
                const size_t row_bytes = (size_t)Po.W * 24u;
                const bool wide = cols == TILE_W && (row_bytes % 16u) == 0 && ((size_t)Po.out % 16u) == 0;
                if (Po.out == nullptr) { // 8-bit frame only (rtc_render_rgb8)
                } else if (wide) {
                    typedef double __attribute__((ext_vector_type(2))) d2;
                    for (uint32_t c = threadIdx.x; c < rows * (TILE_W * 3u / 2u); c += BLOCK) {
                        const uint32_t r = c / (TILE_W * 3u / 2u), k = c % (TILE_W * 3u / 2u);
                        const d2 v = *reinterpret_cast<const d2 *>(stage_f64 + r * (TILE_W * 3u) + k * 2u);
                        char *dst = reinterpret_cast<char *>(Po.out) + (size_t)(orow0 + r) * row_bytes + (size_t)px0 * 24u + k * 16u;
                        RTC_CANVAS_STORE(reinterpret_cast<d2 *>(dst), v);
                    }
                } else {
                    for (uint32_t c = threadIdx.x; c < rows * cols * 3u; c += BLOCK) {
                        const uint32_t r = c / (cols * 3u), k = c % (cols * 3u);
                        Po.out[((size_t)(orow0 + r) * Po.W + px0) * 3u + k] = stage_f64[r * (TILE_W * 3u) + k];
                    }
                }
                if (want8) {
                    const size_t row8 = (size_t)Po.W * 3u;
                    // a tile row is 24 bytes per wave: 16-byte pieces for an even number of waves, 8-byte pieces for one
                    constexpr uint32_t PIECE = (TILE_W * 3u) % 16u == 0 ? 16u : 8u;
                    const bool wide8 = cols == TILE_W && (row8 % PIECE) == 0 && ((size_t)Po.out8 % PIECE) == 0;
                    if (wide8) {
                        typedef unsigned __attribute__((ext_vector_type(PIECE / 4u))) piece_t;
                        for (uint32_t c = threadIdx.x; c < rows * (TILE_W * 3u / PIECE); c += BLOCK) {
                            const uint32_t r = c / (TILE_W * 3u / PIECE), k = c % (TILE_W * 3u / PIECE);
                            const piece_t v = *reinterpret_cast<const piece_t *>(stage_u8 + r * (TILE_W * 3u) + k * PIECE);
                            RTC_CANVAS_STORE(reinterpret_cast<piece_t *>(Po.out8 + (size_t)(orow0 + r) * row8 + (size_t)px0 * 3u + k * PIECE), v);
                        }
                    } else {
                        for (uint32_t c = threadIdx.x; c < rows * cols * 3u; c += BLOCK) {
                            const uint32_t r = c / (cols * 3u), k = c % (cols * 3u);
                            Po.out8[((size_t)(orow0 + r) * Po.W + px0) * 3u + k] = stage_u8[r * (TILE_W * 3u) + k];
                        }
                    }
                }
                }
            }
        }
    }

    if (rep + 1u < reps) {
        // the next tile is staged over the same LDS: the workgroup's cooperative store must have read it (a wave's own LDS
        // operations are in order, so the per-wave output form needs nothing)
        if constexpr (!PROBE && !RTC_WAVE_OUTPUT(REFL)) __syncthreads();
    }
    } // rep
    STAMP(7); // shaded, stored
    const auto &Pc = KP(P_arg);
#ifdef RTC_DIAG_NO_COUNTERS
    if (Pc.counters && Pc.W == 0xffffffffu) {
#else
    if (Pc.counters) {
#endif
        if (lane == 0) { // (rtc_stats::pixels is counted by the host, render_launch)
            unsigned long long *slot = Pc.counters + (size_t)((blockIdx.x * (BLOCK / 64u) + wave) % CNT_SLOTS) * CNT_N;
            if (c_primary) atomicAdd(slot + CNT_PRIMARY, (unsigned long long)c_primary);
            if (c_shadow) atomicAdd(slot + CNT_SHADOW, (unsigned long long)c_shadow);
            if (c_reflect) atomicAdd(slot + CNT_REFLECT, (unsigned long long)c_reflect);
            if (c_refract) atomicAdd(slot + CNT_REFRACT, (unsigned long long)c_refract);
            if (c_resample) atomicAdd(slot + CNT_RESAMPLE, (unsigned long long)c_resample);
            if (c_sky) atomicAdd(slot + CNT_SKY, (unsigned long long)c_sky);
#ifdef RTC_STAMPS
            for (int i = 0; i < 8; ++i) atomicAdd(slot + CNT_STAMP0 + i, stamp_t[i]);
            for (int i = 0; i < 8; ++i) atomicAdd(slot + CNT_STAMP2 + i, stamp_t[8 + i]);
            for (int i = 0; i < 16; ++i) atomicAdd(slot + CNT_DIAG0 + i, (unsigned long long)diag_c[i]);
#endif
        }
    }
}

// Per-render prologue: camera origin in each object's space and the sphere quadratic's `c`
// (shape.rs:363,366) — the part of every primary-ray test that does not depend on the pixel.
DEVI DevPrim prim_of(const double *m_obj, V3 origin) { // (file scope: never under a contraction pragma — parity arithmetic)
    const V3 o = xpoint(m_obj, origin);
    DevPrim r;
    r.ox = o.x;
    r.oy = o.y;
    r.oz = o.z;
    r.c = vdot(o, o) - 1.;
    return r;
}
__global__ void __launch_bounds__(256) k_prep_primary(const DevIsect *isect, DevPrim *prim, uint32_t n,
                                                            double v0, double v1, double v2, double v3, double v4,
                                                            double v5, double v6, double v7, double v8, double v9,
                                                            double v10, double v11) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= n) return;
    const double vinv[12] = {v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11};
    prim[j] = prim_of(isect[j].m, xpoint(vinv, mk(0., 0., 0.)));
}

// Device arithmetic probe (rtc_device_arith).
__global__ void k_arith(uint32_t op, const double *a, const double *b, uint32_t n, double *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (op == 5u || op == 6u) { // Vector::normalize of the triple (a[3t], a[3t+1], a[3t+2]): 5 = the shared-divisor form, 6 = three divisions
        if (i % 3u == 0u && i + 2u < n) {
            const V3 v = mk(a[i], a[i + 1], a[i + 2]);
            const V3 r3 = op == 5u ? vnormalize_shared(v) : vnormalize_plain(v);
            out[i] = r3.x; out[i + 1] = r3.y; out[i + 2] = r3.z;
        }
        return;
    }
    double r;
    switch (op) {
    case 0: r = sqrt(a[i]); break;
    case 1: r = a[i] / b[i]; break;
    case 2: r = pow(a[i], b[i]); break;
    case 3: r = floor(a[i]); break;
    default: r = fmod(a[i], 2.0); break;
    }
    out[i] = r;
}

// ---- binned primary pass: per-view tile lists (DevTileBundle, rtc_device.h) ---------------------------------------
// Primary ray of pixel (px, py) of one camera: Camera::ray_for_pixel_offset(x, 0.5, y, 0.5) camera.rs:64-76, the
// expression k_trace evaluates.
DEVI V3 primary_dir(const DevCamera &C, V3 cam_origin, uint32_t px, uint32_t py, double xo = 0.5, double yo = 0.5) {
    const double xoffset = ((double)px + xo) * C.pixel_size;
    const double yoffset = ((double)py + yo) * C.pixel_size;
    const double world_x = C.half_width - xoffset;
    const double world_y = C.half_height - yoffset;
    const V3 pixel = xpoint(C.vinv, mk(world_x, world_y, -1.));
    // the shared-divisor form (bit-identical, and a cone does not even need that): the binning kernel's lanes normalise ten
    // corner rays each on one latency-bound chain — C3's pipelined frame -7 % (profiles/r03_exp_shared_normalize.log)
    return vnormalize_shared(vsub(pixel, cam_origin));
}

struct BinParams {
    DevCamera views[RTC_MAX_VIEWS];
    uint32_t nviews, W, H, n;
    uint32_t tiles_x, tiles_y, macros_x, macros_y; // tiles of 8x8 pixels, macro tiles of 8x8 tiles
    uint32_t packed;           // n <= 65 536: list entries carry the key (bin_entry)
    uint32_t row0, row_stride; // the launch renders tile rows row0, row0 + row_stride, ... only (one rank's bands): the others get no lists
};

DEVI Bundle bundle_of(const DevTileBundle &t, V3 apex) { // a stored cone as the Bundle bundle_touches takes (rho 0, no reach)
    Bundle B{};
    B.px = apex.x; B.py = apex.y; B.pz = apex.z;
    B.ax = (double)t.ax; B.ay = (double)t.ay; B.az = (double)t.az;
    B.cosT = (double)t.cosT; B.sinT = (double)t.sinT;
    B.rho = 0.; B.tmax = __builtin_inf(); B.spread = 0.;
    B.off = t.off != 0u;
    return B;
}

// f32 unit direction as make_bundle takes it from a lane's f64 direction
DEVI bool dir_f32(V3 d, float &fx, float &fy, float &fz) {
    const float x = (float)d.x, y = (float)d.y, z = (float)d.z;
    const float l2 = x * x + y * y + z * z;
    const float il = __builtin_amdgcn_rsqf(l2);
    fx = x * il; fy = y * il; fz = z * il;
    return finite3(d) && l2 > 1e-30f && l2 < 1e30f;
}

// The cone of the primary rays of the `cell` x `cell` pixel block at (x0, y0), clipped to the image, with make_bundle's
// arithmetic and margins but from five rays instead of all: the axis ray (a pixel near the block's centre) and the four
// corners. The angle between a ray through the image plane and a fixed axis is a quasi-convex function of the pixel
// position (its sub-level sets are the interiors of conic sections), so over the block's rectangle it peaks at a corner;
// the reference arithmetic's rounding (1e-16) and the f32 conversion (6e-8) sit far inside make_bundle's margins
// (sinT * 1.001 + 4e-6). All pixels inside the image count: a superset of the lanes Camera::render traces.
DEVI DevTileBundle cell_cone(const BinParams &Q, uint32_t view, uint32_t x0, uint32_t y0, uint32_t cell) {
#pragma clang fp contract(fast)
    const uint32_t x1 = min(x0 + cell - 1u, Q.W - 1u), y1 = min(y0 + cell - 1u, Q.H - 1u);
    const DevCamera &C = Q.views[view];
    const V3 o = xpoint(C.vinv, mk(0., 0., 0.));
    float ax, ay, az;
    bool good = finite3(o) && dir_f32(primary_dir(C, o, min(x0 + cell / 2u - 1u, x1), min(y0 + cell / 2u - 1u, y1)), ax, ay, az);
    float q2max = 0.f;
    bool narrow = true;
    // corners of the cell's pixel AREA (offsets 0 and 1, not the pixel centres): the cone then also holds every
    // anti-aliasing sub-sample and resample ray of its pixels (camera.rs:98-105: offsets in [0, 1))
    const uint32_t cx[4] = {x0, x1, x0, x1}, cy[4] = {y0, y0, y1, y1};
    for (int k = 0; k < 4; ++k) {
        float fx, fy, fz;
        good = dir_f32(primary_dir(C, o, cx[k], cy[k], (k & 1) ? 1.0 : 0.0, (k & 2) ? 1.0 : 0.0), fx, fy, fz) && good;
        const float dotv = ax * fx + ay * fy + az * fz;
        const float ux = ay * fz - az * fy, uy = az * fx - ax * fz, uz = ax * fy - ay * fx;
        q2max = fmaxf(q2max, ux * ux + uy * uy + uz * uz);
        narrow = narrow && dotv > 0.7f;
    }
    DevTileBundle r;
    r.ax = ax; r.ay = ay; r.az = az;
    r.sinT = __builtin_sqrtf(q2max) * 1.001f + 4e-6f;
    r.cosT = __builtin_sqrtf(fmaxf(0.f, 1.f - r.sinT * r.sinT));
    r.off = (good && narrow && r.sinT < 0.98f) ? 0u : 1u; // wide or odd cells: every object is a candidate
    return r;
}

// A tile-list entry: the object's index and, while the indices fit 16 bits (BinParams::packed), the upper half of its
// key above it — bound_key from the view's camera, truncated towards zero (keys are >= 0), so still a lower bound.
DEVI uint32_t bin_entry(const BinParams &Q, V3 o, const DevBound &b, uint32_t j) {
    if (!Q.packed) return j;
    return (__builtin_bit_cast(uint32_t, bound_key(o, b)) & 0xffff0000u) | j;
}

// The binning kernel: one wave per (view, macro tile of 64x64 pixels), lane = one of its 8x8 tiles of 8x8 pixels. Every
// object goes on the list of each tile whose cone (cell_cone) its bounding sphere can touch — bundle_touches, the wave-level
// cull's own conservative predicate. The wave first finds the objects its MACRO tile's cone can touch through the World's
// two-level tables (groups of 64 Morton-ordered objects, a sphere around each group: lane = group, then lane = member);
// every lane then tests those survivors — fetched by uniform index — against its own tile's cone and appends to its own
// list: no atomics, entries in table order, one launch per render. (Round 2 first PUSHED the objects into the tiles — one
// wave per (view, object), atomic appends, a second kernel for objects that cover many tiles, a third for the cones: the
// same lists at 3x the launches; C3 -2 %, C5 -3 %, one camera per launch C3 -11 % for this form,
// profiles/r02_exp_pull_binning.log.)
// Can NO ray of the cone `t` around apex `o` hit the plane whose stored inverse has the rows m[0..11]? Plane::intersect_local
// (shape.rs:462-471): none when |d'.y| < EPSILON, else t = -o'.y / d'.y, a hit iff t >= 0 — i.e. iff o'.y and d'.y differ in
// sign (or the quotient underflows to -0.0, which `t >= 0.0` accepts: |o'.y| >= 2^-500 and |r| <= 2^500 rule that out, as in
// plane_t_certainly_negative). o'.y is the same for every ray of the view (shared origin; evaluated as the kernel does); d'.y
// = r . d with r = (m4, m5, m6) and d a unit vector within theta of the axis a: s * (r . d) >= s*(r.a) cos(theta) - |r x a|
// sin(theta) when s*(r.a) > 0. Proven a miss when that lower bound clears 1e-4 |r| (the axis is unit to 2e-6, cos / sin carry
// cell_cone's margins, the reference's own roundings are 1e-16).
DEVI bool cone_misses_plane(const double *m, V3 o, const DevTileBundle &t) {
#pragma clang fp contract(fast) // cull arithmetic
    if (t.off != 0u) return false;
    const double oy = m[4] * o.x + m[5] * o.y + m[6] * o.z + m[7];
    if (!(fabs(oy) >= 0x1p-500) || !(fabs(oy) < __builtin_inf())) return false; // (0, NaN, inf: no proof)
    const double s = oy > 0. ? 1. : -1.;
    const double rx = m[4], ry = m[5], rz = m[6];
    const double rr = rx * rx + ry * ry + rz * rz;
    if (!(rr > 0x1p-900) || !(rr < 0x1p900)) return false;
    const double ra = s * (rx * (double)t.ax + ry * (double)t.ay + rz * (double)t.az);
    if (!(ra > 0.)) return false;
    const double perp = __builtin_sqrt(fmax(0., rr - ra * ra) * 1.00001 + 1e-10 * rr);
    const double lower = ra * (double)t.cosT * 0.99999 - perp * (double)t.sinT;
    return lower > 1e-4 * __builtin_sqrt(rr);
}

__global__ void __launch_bounds__(64) k_bin_tiles(const BinParams Q, const DevBound *__restrict__ bound_s, const DevBound *__restrict__ gbound,
                                                   const uint32_t *__restrict__ orig_s, uint32_t ngroups, uint32_t *__restrict__ cnt,
                                                   uint32_t *__restrict__ list, const DevIsect *__restrict__ isect_s,
                                                   const uint32_t *__restrict__ kind_s, uint32_t n_unb, uint32_t *__restrict__ rows,
                                                   const DevIsect *__restrict__ isect, DevPrim *__restrict__ prim) {
    const uint32_t macros = Q.macros_x * Q.macros_y;
    const uint32_t view = blockIdx.x / macros, m = blockIdx.x % macros, lane = threadIdx.x;
    // The view's per-object primary-ray constants (camera origin in object space, the sphere's c: k_prep_primary's table,
    // same arithmetic) for the render kernel's binned pass, spread over all the view's binning threads.
    if (prim != nullptr) {
        const V3 cam = xpoint(Q.views[view].vinv, mk(0., 0., 0.));
        for (uint32_t j = m * 64u + lane; j < Q.n; j += macros * 64u) prim[(size_t)view * Q.n + j] = prim_of(isect[j].m, cam);
    }
    const uint32_t mx = m % Q.macros_x, my = m / Q.macros_x;
    const uint32_t tx = mx * 8u + (lane & 7u), ty = my * 8u + (lane >> 3);
    const bool mine = tx < Q.tiles_x && ty < Q.tiles_y && ty >= Q.row0 && (ty - Q.row0) % Q.row_stride == 0u;
    if (ballot(mine) == 0ull) return; // the launch renders none of this macro tile's rows
    const DevCamera &C = Q.views[view];
    V3 o = xpoint(C.vinv, mk(0., 0., 0.));
    o = mk(uniform_f64(o.x), uniform_f64(o.y), uniform_f64(o.z));
    const Bundle MB = bundle_of(cell_cone(Q, view, mx * 64u, my * 64u, 64u), o);
    const DevTileBundle tcone = cell_cone(Q, view, min(tx, Q.tiles_x - 1u) * 8u, min(ty, Q.tiles_y - 1u) * 8u, 8u);
    const Bundle TB = bundle_of(tcone, o);
    const size_t tile = (size_t)(view * Q.tiles_y + min(ty, Q.tiles_y - 1u)) * Q.tiles_x + min(tx, Q.tiles_x - 1u);
    uint32_t *my_list = list + tile * RTC_TILE_LIST_CAP;
    uint32_t my_cnt = 0;
    // is this tile's cone clear of every unbounded object? (uniform loop, scalar loads, ahead of the group walk so that their latency
    // is not on the kernel's tail; used below for the tile-row proof)
    bool clear_of_planes = n_unb <= 4u;
    for (uint32_t k = 0; k < n_unb && n_unb <= 4u; ++k)
        clear_of_planes = clear_of_planes && (kind_s[k] == RTC_PLANE) && cone_misses_plane(isect_s[k].m, o, tcone);
    for (uint32_t gbase = 0; gbase < ngroups; gbase += 64u) {
        const uint32_t g = gbase + lane;
        unsigned long long gmask = ballot(g < ngroups && bundle_touches(MB, gbound[g])); // (a group with an unbounded member: r = inf, kept)
        // the members' bounds of the NEXT touched group are requested before the current group's survivors are appended: the
        // expansions are a chain of dependent 48-byte-per-lane gathers otherwise (C3: 92 -> see profiles/r03_exp_small_steps.log)
        const DevBound none = DevBound{0., 0., 0., __builtin_inf(), 0., 0.};
        auto fetch = [&](uint32_t gs, DevBound &bb, uint32_t &oo) {
            const uint32_t kk = gs * 64u + lane;
            bb = none; oo = 0u;
            if (kk < Q.n) { bb = bound_s[kk]; oo = orig_s[kk]; }
        };
        DevBound b_next = none;
        uint32_t o_next = 0u;
        if (gmask) fetch(gbase + (uint32_t)__builtin_ctzll(gmask), b_next, o_next);
        while (gmask) {
            const uint32_t gsel = gbase + (uint32_t)__builtin_ctzll(gmask);
            gmask &= gmask - 1ull;
            const DevBound b = b_next;
            const uint32_t orig = o_next;
            if (gmask) fetch(gbase + (uint32_t)__builtin_ctzll(gmask), b_next, o_next);
            const uint32_t k = gsel * 64u + lane;
            bool to = false;
            uint32_t entry = 0u;
            if (k < Q.n) {
                to = b.r < __builtin_inf() && bundle_touches(MB, b); // unbounded objects are never listed: every tile tests them anyway
                if (to) entry = bin_entry(Q, o, b, orig);
            }
            unsigned long long omask = ballot(to);
            while (omask) { // the survivor's record comes from the lane that tested it (readlane: no second round trip to memory)
                const uint32_t l = (uint32_t)__builtin_ctzll(omask);
                omask &= omask - 1ull;
                const DevBound bj = DevBound{lane_f64(b.cx, l), lane_f64(b.cy, l), lane_f64(b.cz, l), lane_f64(b.r, l), lane_f64(b.k, l), lane_f64(b.cn, l)};
                const uint32_t ej = (uint32_t)__builtin_amdgcn_readlane((int)entry, (int)l);
                if (mine && bundle_touches(TB, bj)) {
                    if (my_cnt < RTC_TILE_LIST_CAP) my_list[my_cnt] = ej;
                    ++my_cnt;
                }
            }
        }
    }
    if (mine) cnt[tile] = my_cnt; // > RTC_TILE_LIST_CAP: the list is incomplete and the render wave walks instead
    // Tile rows that are PROVABLY BLACK for this view: a tile whose list is empty and whose cone misses every unbounded object
    // (planes: cone_misses_plane; anything else unbounded: no proof) has no primary hit at any pixel — Camera::render writes
    // background_color there (shape.rs:702-710, 652-653) whatever the rays' exact directions. rows[2v] = the smallest,
    // ~rows[2v + 1] the largest tile row with a tile that is NOT proven empty (both preset to 0xffffffff): the render kernel
    // skips ray generation and every pass for the tile rows outside that range (the sky of a floor scene: a third of the north star).
    const bool nonempty = mine && !(my_cnt == 0u && clear_of_planes);
    const uint32_t rmin = ~wave_max_u32(nonempty ? ~ty : 0u), rmaxinv = ~wave_max_u32(nonempty ? ty : 0u);
    const bool any_nonempty = ballot(nonempty) != 0ull; // (evaluated by the whole wave, not under the lane-0 branch)
    if (lane == 0u && any_nonempty) {
        // look before the atomic: the words only ever decrease, so a (possibly stale) value that is already small enough makes it
        // redundant — at 8192^2 sixteen thousand waves would otherwise queue on these two words (C5: the binning kernel 0.38 ms)
        volatile uint32_t *rw = rows + 2u * view;
        if (rmin < rw[0]) atomicMin(rows + 2u * view, rmin);
        if (rmaxinv < rw[1]) atomicMin(rows + 2u * view + 1u, rmaxinv);
    }
}

extern "C" hipError_t rtc_launch_binning(const DevCamera *views, uint32_t nviews, uint32_t W, uint32_t H, uint32_t n, const DevBound *bound_s,
                                         const DevBound *gbound, const uint32_t *orig_s, uint32_t ngroups, uint32_t *cnt, uint32_t *list,
                                         uint32_t row0, uint32_t row_stride, hipStream_t stream, hipEvent_t e0, hipEvent_t e1,
                                         const DevIsect *isect_s, const uint32_t *kind_s, uint32_t n_unb, uint32_t *rows,
                                         const DevIsect *isect, DevPrim *prim) {
    if (n == 0) return hipSuccess;
    if (hipMemsetAsync(rows, 0xff, sizeof(uint32_t) * 2u * RTC_MAX_VIEWS, stream) != hipSuccess) return hipGetLastError();
    BinParams Q;
    Q.row0 = row0; Q.row_stride = row_stride ? row_stride : 1u;
    for (uint32_t v = 0; v < nviews; ++v) Q.views[v] = views[v];
    for (uint32_t v = nviews; v < RTC_MAX_VIEWS; ++v) Q.views[v] = views[0];
    Q.nviews = nviews; Q.W = W; Q.H = H; Q.n = n;
    Q.packed = RTC_BIN_PACKED(n) ? 1u : 0u;
    Q.tiles_x = (W + 7u) / 8u; Q.tiles_y = (H + 7u) / 8u;
    Q.macros_x = (Q.tiles_x + 7u) / 8u; Q.macros_y = (Q.tiles_y + 7u) / 8u;
    // e0/e1 (may be NULL): the dispatch's own begin/end timestamps, as for k_trace
    hipExtLaunchKernelGGL(k_bin_tiles, dim3(nviews * Q.macros_x * Q.macros_y), dim3(64), 0, stream, e0, e1, 0, Q, bound_s, gbound, orig_s, ngroups, cnt, list,
                          isect_s, kind_s, n_unb, rows, isect, prim);
    return hipGetLastError();
}

// ---- light-space shadow lists (rtc_device.h) -------------------------------------------------------------------------
// Unit direction through cube-map coordinates (u, v) of face `face`.
DEVI void light_dir(uint32_t face, float u, float v, float &x, float &y, float &z) {
    const float s = (face & 1u) ? -1.f : 1.f;
    const uint32_t a = face >> 1;
    float c[3];
    c[a] = s; c[(a + 1u) % 3u] = u; c[(a + 2u) % 3u] = v;
    const float il = __builtin_amdgcn_rsqf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    x = c[0] * il; y = c[1] * il; z = c[2] * il;
}
// One thread per cell or per macro cell (8x8 cells): the cone of its directions (axis through the centre, half-angle to
// the farthest of the four corners — the angle to the axis is quasi-convex over the square — with make_bundle's margins).
__global__ void __launch_bounds__(256) k_light_cells(DevTileBundle *__restrict__ cells, DevTileBundle *__restrict__ macros, uint32_t *__restrict__ cnt) {
#pragma clang fp contract(fast)
    constexpr uint32_t R = RTC_LIGHT_R, M = RTC_LIGHT_R / 8u;
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint32_t g, span;
    DevTileBundle *out;
    if (i < 6u * R * R) { g = R; span = 1u; out = cells; cnt[i] = 0u; }
    else if ((i -= 6u * R * R) < 6u * M * M) { g = M; span = 8u; out = macros; }
    else return;
    const uint32_t face = i / (g * g), ix = (i % g) * span, iy = ((i / g) % g) * span;
    const float u0 = (float)ix * (2.f / R) - 1.f, u1 = (float)(ix + span) * (2.f / R) - 1.f;
    const float v0 = (float)iy * (2.f / R) - 1.f, v1 = (float)(iy + span) * (2.f / R) - 1.f;
    float ax, ay, az;
    light_dir(face, 0.5f * (u0 + u1), 0.5f * (v0 + v1), ax, ay, az);
    const float cu[4] = {u0, u1, u0, u1}, cv[4] = {v0, v0, v1, v1};
    float q2max = 0.f;
    for (int k = 0; k < 4; ++k) {
        float fx, fy, fz;
        light_dir(face, cu[k], cv[k], fx, fy, fz);
        const float ux = ay * fz - az * fy, uy = az * fx - ax * fz, uz = ax * fy - ay * fx;
        q2max = fmaxf(q2max, ux * ux + uy * uy + uz * uz);
    }
    DevTileBundle r;
    r.ax = ax; r.ay = ay; r.az = az;
    r.sinT = __builtin_sqrtf(q2max) * 1.001f + 1e-5f; // + the f64 -> cell rounding of a direction on a cell border
    r.cosT = __builtin_sqrtf(fmaxf(0.f, 1.f - r.sinT * r.sinT));
    r.off = r.sinT < 0.7f ? 0u : 1u;
    out[i] = r;
}
DEVI Bundle light_bundle_of(const DevTileBundle &t, V3 apex, double reach) { // rays FROM the light, reach-limited (rho 0)
    Bundle B = bundle_of(t, apex);
    B.tmax = reach;
    B.spread = reach; // the exact rays start at the far end (DevBound rounding note)
    return B;
}
// One wave per object: macro cells (8x8 cells) 64 per step, then the cells of the touched macro cells.
__global__ void __launch_bounds__(64) k_light_bin(uint32_t n, uint32_t cap, const DevBound *__restrict__ bound, double lx, double ly, double lz, double reach,
                                                   const DevTileBundle *__restrict__ cells, const DevTileBundle *__restrict__ macros,
                                                   uint32_t *__restrict__ cnt, uint32_t *__restrict__ list) {
    constexpr uint32_t R = RTC_LIGHT_R, M = RTC_LIGHT_R / 8u;
    const uint32_t j = blockIdx.x, lane = threadIdx.x;
    if (j >= n) return;
    const DevBound b = bound[j];
    if (!(b.r < __builtin_inf())) return; // unbounded: every shadow pass tests it anyway
    const V3 o = mk(lx, ly, lz);
    for (uint32_t mbase = 0; mbase < 6u * M * M; mbase += 64u) {
        const uint32_t m = mbase + lane;
        bool tm = false;
        if (m < 6u * M * M) tm = bundle_touches(light_bundle_of(macros[m], o, reach), b);
        unsigned long long mmask = ballot(tm);
        while (mmask) {
            const uint32_t msel = mbase + (uint32_t)__builtin_ctzll(mmask);
            mmask &= mmask - 1ull;
            const uint32_t face = msel / (M * M), mx = msel % M, my = (msel / M) % M;
            const uint32_t cell = (face * R + my * 8u + (lane >> 3)) * R + mx * 8u + (lane & 7u);
            if (bundle_touches(light_bundle_of(cells[cell], o, reach), b)) {
                const uint32_t slot = atomicAdd(cnt + cell, 1u);
                if (slot < cap) list[(size_t)cell * cap + slot] = j;
            }
        }
    }
}

extern "C" hipError_t rtc_launch_light_lists(uint32_t n, uint32_t cap, const DevBound *bound, const double light[3], double reach, DevTileBundle *cells,
                                             DevTileBundle *macros, uint32_t *cnt, uint32_t *list, hipStream_t stream) {
    constexpr uint32_t R = RTC_LIGHT_R, M = RTC_LIGHT_R / 8u;
    hipLaunchKernelGGL(k_light_cells, dim3((6u * R * R + 6u * M * M + 255u) / 256u), dim3(256), 0, stream, cells, macros, cnt);
    if (n) hipLaunchKernelGGL(k_light_bin, dim3(n), dim3(64), 0, stream, n, cap, bound, light[0], light[1], light[2], reach, (const DevTileBundle *)cells,
                              (const DevTileBundle *)macros, cnt, list);
    return hipGetLastError();
}

// Un-deal (rtc_group_render, member 0): the gather leaves N chunks, chunk p = member p's packed bands of
// `nframes` frames ([nframes][rows_max][row_units] units each); this puts band k of member p at image rows
// (p + k*N)*8.. of its frame — the reference's row-major Canvas (canvas.rs:43-51). One unit = UNIT bytes
// (16 for the f64 canvas of an even-width image), one lane per unit, 256-B contiguous per wave-instruction.
template <class U>
__global__ void __launch_bounds__(256) k_undeal(const U *__restrict__ staging, U *__restrict__ canvas, uint32_t nranks,
                                                 uint32_t nframes, uint32_t H, uint32_t rows_max, uint32_t row_units) {
    const size_t total = (size_t)nframes * H * row_units;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < total; i += (size_t)gridDim.x * 256u) {
        const uint32_t u = (uint32_t)(i % row_units);
        const size_t fy = i / row_units;
        const uint32_t y = (uint32_t)(fy % H), f = (uint32_t)(fy / H);
        canvas[i] = staging[rtc_staging_index(f, y, u, nranks, nframes, rows_max, row_units)]; // rtc_bands.h: THE mapping
    }
}

extern "C" hipError_t rtc_launch_undeal(const void *staging, void *canvas, uint32_t nranks, uint32_t nframes, uint32_t H,
                                        uint32_t rows_max, size_t row_bytes, hipStream_t stream) {
    if (nframes == 0 || H == 0 || row_bytes == 0) return hipSuccess;
    const bool a16 = row_bytes % 16u == 0 && ((size_t)staging % 16u) == 0 && ((size_t)canvas % 16u) == 0;
    const bool a8 = row_bytes % 8u == 0 && ((size_t)staging % 8u) == 0 && ((size_t)canvas % 8u) == 0;
    const uint32_t unit = a16 ? 16u : a8 ? 8u : 1u;
    const size_t total = (size_t)nframes * H * (row_bytes / unit);
    const uint32_t blocks = (uint32_t)((total + 255u) / 256u < 16384u ? (total + 255u) / 256u : 16384u);
    typedef unsigned __attribute__((ext_vector_type(4))) u4;
    if (unit == 16u)
        hipLaunchKernelGGL(k_undeal<u4>, dim3(blocks), dim3(256), 0, stream, (const u4 *)staging, (u4 *)canvas, nranks, nframes, H, rows_max, (uint32_t)(row_bytes / 16u));
    else if (unit == 8u)
        hipLaunchKernelGGL(k_undeal<unsigned long long>, dim3(blocks), dim3(256), 0, stream, (const unsigned long long *)staging, (unsigned long long *)canvas, nranks, nframes, H, rows_max, (uint32_t)(row_bytes / 8u));
    else
        hipLaunchKernelGGL(k_undeal<unsigned char>, dim3(blocks), dim3(256), 0, stream, (const unsigned char *)staging, (unsigned char *)canvas, nranks, nframes, H, rows_max, (uint32_t)row_bytes);
    return hipGetLastError();
}

// ---- launchers (called from rtc_api.cpp) --------------------------------------------------
template <int SRC, bool REFL, bool REFR, bool PROBE>
static hipError_t launch_kernel(const RenderParams &P, dim3 grid, size_t lds_bytes, hipStream_t stream, hipEvent_t e0,
                                hipEvent_t e1) {
    if (lds_bytes > 48 * 1024) { // more dynamic LDS than the default limit: opt in (up to 160 KiB per CU on gfx950)
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_trace<SRC, REFL, REFR, PROBE>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    // e0/e1 (may be NULL) receive the dispatch's own begin/end timestamps: no marker packets on the stream
    hipExtLaunchKernelGGL((k_trace<SRC, REFL, REFR, PROBE>), grid, dim3(RTC_BLOCK_FOR(CULL_LEVEL(SRC), REFL, REFR, PROBE)), lds_bytes, stream, e0, e1, 0, P, P.isect,
                          P.kind, P.shade, P.prim, P.bound, P.isect_s, P.kind_s, P.bound_s, P.orig_s, P.gbound, P.idtab, P.pre, P.pre_s);
    return hipGetLastError();
}
template <int SRC, bool REFL, bool REFR>
static hipError_t launch_one(const RenderParams &P, dim3 grid, size_t lds_bytes, hipStream_t stream, hipEvent_t e0,
                             hipEvent_t e1) {
    if (P.rays != nullptr) return launch_kernel<SRC, REFL, REFR, true>(P, grid, lds_bytes, stream, e0, e1);
    return launch_kernel<SRC, REFL, REFR, false>(P, grid, lds_bytes, stream, e0, e1);
}

extern "C" hipError_t rtc_launch_trace(const RenderParams *P, int src, int refl, int refr, uint32_t nblocks,
                                       size_t lds_bytes, hipStream_t stream, hipEvent_t e0, hipEvent_t e1) {
    const dim3 grid(nblocks);
#define RTC_CASE(S)                                                                            \
    if (src == S) {                                                                            \
        if (refr) return launch_one<S, true, true>(*P, grid, lds_bytes, stream, e0, e1);               \
        if (refl) return launch_one<S, true, false>(*P, grid, lds_bytes, stream, e0, e1);              \
        return launch_one<S, false, false>(*P, grid, lds_bytes, stream, e0, e1);                       \
    }
    RTC_CASE(SRC_SMEM)
    RTC_CASE(SRC_LDS1)
    RTC_CASE(SRC_LDSN)
    RTC_CASE(SRC_CULL)
    RTC_CASE(SRC_CULL2)
#undef RTC_CASE
    return hipErrorInvalidValue;
}

extern "C" hipError_t rtc_launch_prep(const DevIsect *isect, DevPrim *prim, uint32_t n, const double vinv[12],
                                      hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_prep_primary, dim3((n + 255u) / 256u), dim3(256), 0, stream, isect, prim,
                       n, vinv[0], vinv[1], vinv[2], vinv[3], vinv[4], vinv[5], vinv[6], vinv[7], vinv[8], vinv[9],
                       vinv[10], vinv[11]);
    return hipGetLastError();
}

extern "C" hipError_t rtc_launch_arith(uint32_t op, const double *a, const double *b, uint32_t n, double *out,
                                       hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_arith, dim3((n + 255) / 256), dim3(256), 0, stream, op, a, b, n, out);
    return hipGetLastError();
}
