// rtc_device.h — layout of the flattened World in HBM and the kernel parameter block.
// Shared by rtc_api.cpp (host side of the C-ABI) and rtc_kernels.hip.
//
// HBM layout (all f64, written once per rtc_world_create, read-only afterwards):
//   isect[n]  96 B  rows 0..2 of the shape's stored inverse transform — everything
//                   Shape::intersect needs (vec.rs:211-214, transform.rs:107-128);
//                   read wave-uniformly through the scalar cache (or staged in LDS tiles).
//   kind[n]    4 B  RTC_SPHERE / RTC_PLANE / RTC_CUBE.
//   bound[n]  48 B  f64 world-space bounding sphere (centre, radius) + rounding-inflation terms.
//   shade[n] 344 B  what shade_hit needs for the ONE object a ray hit: inverse-transpose 3x3,
//                   material scalars, pattern; gathered per lane after the hit is known.
//   isect_s/kind_s/bound_s/orig_s[n], gbound[ceil(n/64)]: the same records in Morton order of the
//                   bound centres with one bounding sphere per group of 64 — the two-level cull.
//   prim[n]   32 B  per-render scratch: the camera origin in object space and `c` of the sphere
//                   quadratic — identical for every primary ray, so computed once per object
//                   (with the reference's arithmetic) instead of once per pixel x object.
#ifndef RTC_DEVICE_H
#define RTC_DEVICE_H

#include <stdint.h>

struct DevIsect {
    double m[12]; // inv[0][0..3], inv[1][0..3], inv[2][0..3]
};

struct DevShade {
    double nt[9];       // 3x3 of transform_transpose (shape.rs:285,301); cubes use inv^T (shape.rs:627)
    double color[3];
    double ambient, diffuse, specular, shininess, reflective, transparency, refractive_index;
    double pat_inv[12]; // rows 0..2 of Pattern.xf_inv
    double pat_a[3], pat_b[3];
    double plane_n[3];  // planes only: normalize(transform_vector(nt, (0,1,0))) — Shape::normal_at is the
                        // same for every point of a plane (shape.rs:34-40,481-483), so it is evaluated
                        // once per object (same operations, same IEEE results) instead of once per hit
    uint32_t kind, pattern_kind, world_id, _pad;
};

struct DevPrim {
    double ox, oy, oz; // transform_point(inv, camera origin)
    double c;          // (o.o) - 1.   (shape.rs:366)
};

// One shape of the id-ordered walk of the n1/n2 pass: World.shapes index + its world_id
// (shape.rs:127 compares ids; shapes sharing an id are one container).
struct DevIdEntry {
    uint32_t index, id;
};

struct DevBound {
    double cx, cy, cz, r; // world-space bounding sphere; r = +inf: unbounded (planes) or not
                          // computable -> never culled
    double k;             // 0.75e-14 * ||A||_F^2 (A = 3x3 of the stored inverse): rounding inflation
    double cn;            // |centre|: see the note on rounding below
};
// The per-lane prefilter's own record (ray_touches_pre, rtc_kernels.hip): centre and SQUARED radius of the bounding sphere with
// the rounding inflation of the note below already applied for every ray origin within |o|_1 <= pre_limit (RenderParams) —
// D <= |C - o|_1 <= |C|_1 + pre_limit, so R = r * (1 + k * Dw * (cn + Dw)) * 1.000001 + 1e-12 with Dw = |C|_1 + pre_limit is
// an upper bound of what ray_touches computes per lane. 32 bytes instead of 48, and 15 instead of 26 instructions per test;
// a pass with an origin beyond the limit takes the general test. R2 = +inf: unbounded, never culled.
struct DevPre {
    double cx, cy, cz, R2;
};

// Binned primary pass. Per render and view one kernel (k_bin_tiles) puts every object on the list of each 8x8-pixel tile
// whose primary-ray cone its bounding sphere can touch — the SAME conservative predicate the wave-level cull applies
// (bundle_touches), against a cone built from the tile's corner rays (cell_cone). The render kernel's primary pass then runs
// the exact test on its tile's list (1.8 objects on average at 10 000 spheres) instead of walking 157 group spheres and
// expanding 17 groups. A tile whose list overflows RTC_TILE_LIST_CAP falls back to the walk. Unbounded objects (planes) are
// never listed: they are the first `n_unb` entries of the Morton-sorted tables and every tile tests them.
struct DevTileBundle {
    float ax, ay, az, cosT, sinT; // cone around the tile's (or macro tile's) rays; apex = the view's camera origin
    uint32_t off;                 // 1: could not be bounded — every object is a candidate
};
#define RTC_TILE_LIST_CAP 64u // one lane per entry in the render kernel's nearest-first walk

// Light-space shadow lists (two-level worlds). Every shadow segment ends at the light (is_shadowed_by_light shape.rs:716-720),
// so seen from the light it is a ray in some direction: a cube map of RTC_LIGHT_R x RTC_LIGHT_R direction cells per face
// around the light, built ONCE per World, holds for each cell the objects a ray from the light in that cell's directions can
// touch within `light_reach` (bundle_touches against the cell's cone, as everywhere). A wave's shadow pass then filters the
// lists of the (typically one to four) cells its lanes' directions fall in with its own shadow bundle, instead of walking
// every group sphere of the World. Cells whose list overflows, segments longer than light_reach and unbounded objects fall
// back to the walk / are tested always.
#define RTC_LIGHT_R 128u
#define RTC_LIGHT_LIST_CAP 128u      // entries per cell, two-level worlds (filtered 64 at a time)
#define RTC_LIGHT_LIST_CAP_SMALL 16u // one-level worlds: the lists are the candidates themselves

// Rounding note. The cull must never drop an object for which the REFERENCE ARITHMETIC reports an
// intersection — including intersections that exist only because of rounding. The sphere test
// evaluates disc = b*b - 4*a*c with b^2 and 4ac of size ~4a|o'|^2 (o' = object-space ray origin);
// far origins and thin objects (large ||A||) make the f64 error exceed the true discriminant and
// the reference "hits" an object the ray geometrically misses (seen: origin 1e6 away, ||A||_F^2 =
// 1.7e5, hit reported 2.1 radii from the centre). Error analysis of the reference's expression
// order gives: a root can be reported only if the line passes within object-space distance
//   R_eff^2 <= 1 + eps*|o'|*(16*S + 13*|o'|),  |o'| <= ||A||*D,  S = ||A||*(|o| + |c|),
// D = distance of the ray origin from the centre; every reported root lies inside that inflated
// sphere. With eps = 2^-52, |o| <= |c| + D and sqrt(1+x) <= 1 + x/2 the world-space radius to test
// is r * (1 + k * D * (cn + D)) with k = 0.75e-14 * ||A||_F^2 (any upper bound of D may be used).
// For ordinary scenes the factor is 1 + 1e-9.

enum { CNT_PRIMARY = 0, CNT_SHADOW = 1, CNT_REFLECT = 2, CNT_REFRACT = 3, CNT_PIXELS = 4, CNT_RESAMPLE = 5, CNT_SKY = 6, CNT_STAMP0 = 8, CNT_DIAG0 = 16, CNT_STAMP2 = 32 /* the same phases, secondary passes only */, CNT_N = 40 };
// Ray counters are kept in CNT_SLOTS replicas (one 64-byte line each); a wave adds to the replica
// picked by its workgroup id, so no single word sees more than 1/CNT_SLOTS of the atomics. The host
// sums the replicas (rtc_stats_read).
enum { CNT_SLOTS = 256 };

// One camera of a launch (camera.rs:17-27, the part ray_for_pixel reads).
struct DevCamera {
    double half_width, half_height, pixel_size;
    double vinv[12];
};
enum { RTC_MAX_VIEWS = 8 };

// Workgroup = waves side by side, each an 8x8 pixel tile: the workgroup's tile is (threads/64)*8 x 8
// pixels (host grid and kernel must agree). The flat kernel uses 128 threads (its workgroup stores the
// 16x8 tile cooperatively, three full 128-byte lines per row; against 256 threads: the same at 100
// objects, 5 % faster at 1000 and 10 000; 64 is within 2 % either way, 512 is 16 % slower); the
// frame-stack kernels, whose waves finish far apart and store their parts on their own, use ONE wave per
// workgroup — a workgroup's slots are only handed on when its last wave is done (reflective 1080p:
// 256 threads 0.535 ms, 128 0.475, 64 0.463; 4096x4096: 2.27 / 1.82 / 1.63 ms).
#ifndef RTC_BLOCK
#define RTC_BLOCK 128
#endif
#ifndef RTC_BLOCK_STACK
#define RTC_BLOCK_STACK 64
#endif
// K3, ray compaction between bounces (rtc_kernels.hip, COMPACT): reflection-only Worlds on the culled variants render with
// TWO waves per workgroup, which merge their live rays into one wave once they fit (RTC_COMPACT = 1); 0 = one wave per
// workgroup, no merging. Host grid and kernel must agree on the workgroup size: RTC_BLOCK_FOR.
#ifndef RTC_COMPACT
#define RTC_COMPACT 0
#endif
#define RTC_COMPACT_FOR(cull, refl, refr, probe) (RTC_COMPACT && (cull) && (refl) && !(refr) && !(probe))
// `cull` is the cull level of the variant: 0 = none, 1 = one level (n <= 256), 2 = two levels (RTC_CULL_LEVEL).
#ifndef RTC_BLOCK_CULL2
#define RTC_BLOCK_CULL2 64 // large worlds, flat kernel: C3 0.106 -> 0.102 ms against 128 (profiles/r02_exp_block_size.log)
#endif
#ifndef RTC_BLOCK_CULL1
#define RTC_BLOCK_CULL1 64 // small worlds, flat kernel, since the guided chunks (a workgroup renders several tiles): one wave per workgroup
                           // 0.0576 -> 0.0561 ms pipelined, 0.0581 -> 0.0566 solo against 128 (north star; C2 -2.4 %); before the chunks: -1 %
#endif
#define RTC_BLOCK_FOR(cull, refl, refr, probe) \
    (((refl) || (refr)) ? (RTC_COMPACT_FOR(cull, refl, refr, probe) ? 128 : RTC_BLOCK_STACK) \
                        : ((cull) == 2 && !(probe) ? RTC_BLOCK_CULL2 : ((cull) == 1 && !(probe) ? RTC_BLOCK_CULL1 : RTC_BLOCK)))
#define RTC_TILE_W_FOR(cull, refl, refr, probe) ((RTC_BLOCK_FOR(cull, refl, refr, probe) / 64u) * 8u)

// Tile-list entries hold (key >> 16) << 16 | index while every index fits 16 bits
#define RTC_BIN_PACKED(n) ((n) <= 65536u)

struct RenderParams {
    const DevIsect *isect;
    const uint32_t *kind;
    const DevShade *shade;
    const DevPrim *prim;
    const DevBound *bound;
    // cull tables: objects in Morton order of their bound centres (unbounded ones first), grouped by 64
    const DevIsect *isect_s;   // [n] records in sorted order
    const uint32_t *kind_s;    // [n]
    const DevBound *bound_s;   // [n]
    const uint32_t *orig_s;    // [n] sorted position -> insertion index (World.shapes order)
    const DevBound *gbound;    // [ngroups] sphere around each group of 64 sorted objects
    const DevIdEntry *idtab;   // [n] shapes in stable order of world_id (compute_refractive's container key)
    const DevPre *pre;         // [n] prefilter records, insertion order
    const DevPre *pre_s;       // [n] ... in sorted order
    double pre_limit;          // the records hold for ray origins with |o|_1 <= pre_limit
    // binned primary pass (nullptr: not available for this launch): per view `tiles_x * tiles_y` counters and lists of
    // RTC_TILE_LIST_CAP insertion indices, image tile (tx, ty) of view v at (v * tiles_y + ty) * tiles_x + tx
    const uint32_t *tile_cnt;
    const uint32_t *tile_list;
    const uint32_t *tile_rows; // per view two words: the smallest, and ~ the largest, tile row with a tile NOT proven black (k_bin_tiles)
    uint32_t tiles_x, tiles_y;
    // light-space shadow lists (nullptr: none): per direction cell a counter and RTC_LIGHT_LIST_CAP insertion indices
    const uint32_t *light_cnt;
    const uint32_t *light_list;
    uint32_t light_cap;          // entries per cell of THIS World's lists
    double light_reach;
    uint32_t bin_packed;        // tile-list entries carry the upper half of the object's key above its index (RTC_BIN_PACKED)
    uint32_t n_unb;             // unbounded objects = the first n_unb entries of isect_s / kind_s / orig_s
    uint32_t ngroups;
    uint32_t n;
    uint32_t tile_cap; // objects per LDS tile (LDS variants)
    double light_pos[3], light_int[3];
    // camera (camera.rs:17-27)
    uint32_t W, H, y0, y1, mode;
    uint32_t samples;     // 1: one ray per pixel (camera.rs:96-98); anything else: the 4 fixed sub-samples
                          // + the resample test (camera.rs:99-113) — 0 is normalised to 4 by the host
    uint32_t resample_n;  // RTC_FLAG_AA_RESAMPLE: rays added by Camera::resample (antialiasing_samples), else 0
    uint32_t aa_lds_off;  // samples != 1: byte offset of the sub-sample store inside the dynamic LDS block
    // A launch renders `nviews` cameras of the same size onto the same World (rtc_render_views: the
    // frames of a camera move, a stereo pair): workgroups [v*tiles, (v+1)*tiles) belong to view v and
    // write `view_rows` rows further down the output buffers. One camera = views[0], nviews = 1.
    DevCamera views[RTC_MAX_VIEWS];
    uint32_t nviews, view_rows;
    double *out;                 // rows x W x 3, rows = y1-y0 (band_stride 1) or 8*grid_y (packed bands)
    unsigned char *out8;         // optional: the same rows quantised by Color::scale(c, 255)
    unsigned long long *counters; // CNT_N
    // probe mode (rtc_color_at): rays != nullptr
    const double *rays;
    uint32_t nrays, remaining;
    void *hits; // rtc_hit[nrays] or nullptr
    uint32_t grid_x, grid_y; // logical block grid of the render
    uint32_t flags;          // RTC_FLAG_*
    uint32_t total_blocks;   // tiles (x views) of the launch; a workgroup renders tiles blockIdx.x + k * gridDim.x, k < reps
    uint32_t reps;
    uint32_t chunk_wgs[4];   // guided chunks: the first chunk_wgs[0] workgroups render EIGHT tiles each, the next chunk_wgs[1] four, then
                             // chunk_wgs[2] three, chunk_wgs[3] two, the rest one — tile ids ascending in that order, a workgroup's tiles one level-stride apart. A
                             // tile's stores drain under the workgroup's next tile and the per-wave set-up is shared, while the launch ENDS
                             // with single-tile workgroups: a lone launch's tail stays one tile long. All zero: `reps` tiles for everyone.
    uint32_t band_stride;    // tile row k of the grid renders image rows y0 + 8*k*band_stride .. (+8) and
                             // writes output rows 8*k .. (+8): 1 = a contiguous range of rows, N = every
                             // N-th band of 8 rows (interleaved row tiles, rtc_render_bands)
};

#endif
