/*
 * hsr_oracle.c — CPU restatement of the Hier-SLAM differentiable Gaussian rasterizer.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this library, and there only
 * as the checker / reported CPU baseline.  The product path (hier-slam_amd/) never links, imports
 * or calls it and fails loudly when its HIP library is missing.
 *
 * PARITY UNPINNED.  The reference (LeeBY68/Hier-SLAM, hierslam-diff-gaussian-rasterization-w-depth)
 * holds no tests, golden vectors or fixtures for this path and its kernels are
 * CUDA (nvcc, cooperative_groups, cub): unbuildable in this image, so no reference output exists to
 * pin this file against.  It is an independent plain-C restatement written from the behaviour of the
 * reference source; every function cites the reference file:line it follows
 * (paths relative to hierslam-diff-gaussian-rasterization-w-depth/cuda_rasterizer/).  What pins it
 * instead (tests/): a float64 dense torch.autograd restatement of the forward maths, finite
 * differences, and the algebraic identities the compositing satisfies.
 *
 * Two builds of this one file (oracle/Makefile):
 *   libhsr_oracle.so      `real` = float: THE ORACLE.  fp32 wherever the reference is fp32, in its operation order.
 *   libhsr_oracle_f64.so  -DHSRO_TRUTH, `real` = double: the "truth" build.  The per-Gaussian preprocess — everything that
 *                         decides an integer (radii, tile rects, keys, sort, ranges) and the fp32 state the tile kernels
 *                         read (means2D, conic, depth) — is the SAME fp32 code, so both builds walk identical lists; the
 *                         compositing loop, its backward and the per-Gaussian chain rule run in double on that state.
 *                         Where the oracle and another fp32 implementation (the HIP kernels) differ by rounding, the truth
 *                         build says who is closer to exact arithmetic (tests/test_oracle.py, tests/test_gpu_truth.py).
 *
 * Floating-point policy: everything that decides an INTEGER output (radii, tile rects, tiles_touched,
 * sort keys, ranges) is evaluated in the reference's exact operation order, fp32 (double where the
 * reference promotes), with contraction disabled (build with -ffp-contract=off).
 *
 * Gradient accumulation: the reference sums per-(pixel, Gaussian) terms with fp32 atomicAdd in
 * arbitrary order (backward.cu:616-663, :828-896).  Here each per-pair term is computed in fp32
 * exactly as the reference does and the per-Gaussian sum is kept in double, rounded to fp32 once:
 * order-independent, and within the reference's own fp32 noise of any of its summation orders.
 *
 * Median-depth gradient: the reference's backward finds the splat at which T crossed 0.5 again, from the
 * T it reconstructs by dividing (backward.cu:623-626, :854-857).  That reconstruction is exact only up to
 * rounding, so on a pixel whose T passes within an ulp of 0.5 the backward can pick the neighbouring
 * splat, none, or two — an artefact of the arithmetic that differs between any two implementations
 * (CUDA's expf / division included).  hsro_set_median_rule(0) (default) keeps the reference's rule;
 * rule 1 uses the list position the FORWARD recorded for the crossing (field "median_pos"), which is what
 * the HIP product does.  Every backward call counts the pixels on which the two rules disagree
 * (hsro_last_median_rule_disagreements()), so a comparison can say how many there were.
 *
 * Threshold ties (test aid, not in the reference).  The compositing loop takes hard decisions on computed floats —
 * power > 0, alpha >= 1/255, T(1-alpha) < 1e-4, T crossing 0.5 — and two correct fp32 evaluations (glibc expf here,
 * v_exp_f32 in the HIP kernels, CUDA's expf in the reference) can take a decision that falls within ulps of its
 * threshold differently; the pixel then differs by that splat's whole contribution.  The forward flags such pixels
 * (tie_pixels) and the splats involved (tie_gaussians), and BOUNDS the difference: for every flagged decision the pixel
 * is evaluated again with that one decision taken the other way; the sum of |difference| over the flagged decisions of a
 * pixel is its image bound (tie_img_bound), and the backward does the same for the gradient rows (HsroBounds: per-splat
 * differences of the accumulated sums, carried through the per-Gaussian chain rule — which is linear in them — and
 * summed in absolute value per Gaussian).  A comparison may then allow exactly that much on exactly those entries,
 * instead of leaving them out.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BLOCK_X 16
#define BLOCK_Y 16
#define BLOCK_SIZE (BLOCK_X * BLOCK_Y)
#define NUM_CHANNELS 3 /* config.h:15 */

#ifdef HSRO_TRUTH
typedef double real;
#define R_EXP exp
#define R_FABS fabs
static inline double sqrt_r(double x) { return sqrt(x); }
#else
typedef float real;
#define R_EXP expf
#define R_FABS fabsf
static inline float sqrt_r(float x) { return sqrtf(x); }
#endif
static inline float sqrt_f(float x) { return sqrtf(x); }
int hsro_real_bytes(void) { return (int)sizeof(real); }

/* auxiliary.h:21-37 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                              -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                              0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                              -0.5900435899266435f};

/* the vector / matrix helpers, once in fp32 (forward preprocess: always fp32) and once in `real` (backward chain) */
#define RT float
#define LA(x) x##_f
#include "hsr_oracle_la.h"
#undef RT
#undef LA
#define RT real
#define LA(x) x##_r
#include "hsr_oracle_la.h"
#undef RT
#undef LA

/* auxiliary.h:41-44: double arithmetic, rounded to float on return */
static float ndc2Pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

/* auxiliary.h:46-56 */
static void getRect(float px, float py, int max_radius, uint32_t* rminx, uint32_t* rminy,
                    uint32_t* rmaxx, uint32_t* rmaxy, uint32_t gx, uint32_t gy)
{
    int a;
    a = (int)((px - max_radius) / BLOCK_X); if (a < 0) a = 0; *rminx = (uint32_t)a < gx ? (uint32_t)a : gx;
    a = (int)((py - max_radius) / BLOCK_Y); if (a < 0) a = 0; *rminy = (uint32_t)a < gy ? (uint32_t)a : gy;
    a = (int)((px + max_radius + BLOCK_X - 1) / BLOCK_X); if (a < 0) a = 0; *rmaxx = (uint32_t)a < gx ? (uint32_t)a : gx;
    a = (int)((py + max_radius + BLOCK_Y - 1) / BLOCK_Y); if (a < 0) a = 0; *rmaxy = (uint32_t)a < gy ? (uint32_t)a : gy;
}

/* rasterizer_impl.cu:35-50 */
uint32_t hsro_get_higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

/* auxiliary.h:139-164 (prefiltered trap omitted: the oracle has no device to trap) */
static int in_frustum(int idx, const float* pts, const float* view, const float* proj, v3_f* p_view)
{
    v3_f p = {pts[3 * idx], pts[3 * idx + 1], pts[3 * idx + 2]};
    (void)proj; /* p_hom / p_proj are computed but unused by the reference's test */
    *p_view = transformPoint4x3_f(p, view);
    return !(p_view->z <= 0.2f);
}

/* rasterizer_impl.cu:54-66, :141-153 */
void hsro_mark_visible(int P, const float* means3D, const float* view, const float* proj, uint8_t* present)
{
    for (int i = 0; i < P; i++) { v3_f pv; present[i] = (uint8_t)in_frustum(i, means3D, view, proj, &pv); }
}

/* forward.cu:20-71 */
static v3_f computeColorFromSH(int idx, int deg, int max_coeffs, const float* means, const float* campos,
                               const float* shs, uint8_t* clamped)
{
    v3_f pos = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
    v3_f dir = {pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]};
    float len = sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
    dir.x = dir.x / len; dir.y = dir.y / len; dir.z = dir.z / len;
    const float* sh = shs + (size_t)idx * max_coeffs * 3;
    float res[3];
    float x = dir.x, y = dir.y, z = dir.z;
    for (int c = 0; c < 3; c++) {
#define SH(i) sh[(i) * 3 + c]
        float r = SH_C0 * SH(0);
        if (deg > 0) {
            r = r - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) + SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) +
                    SH_C2[3] * xz * SH(7) + SH_C2[4] * (xx - yy) * SH(8);
                if (deg > 2) {
                    r = r + SH_C3[0] * y * (3.0f * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
                        SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
                        SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                        SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) + SH_C3[5] * z * (xx - yy) * SH(14) +
                        SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
                }
            }
        }
#undef SH
        r += 0.5f;
        clamped[3 * idx + c] = (r < 0);
        res[c] = r > 0.0f ? r : 0.0f;
    }
    v3_f out = {res[0], res[1], res[2]};
    return out;
}

/* forward.cu:118-152 */
static void computeCov3D(v3_f scale, float mod, v4_f rot, float* cov3D)
{
    m3_f S = m3_make_f(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.c[0][0] = mod * scale.x; S.c[1][1] = mod * scale.y; S.c[2][2] = mod * scale.z;
    float r = rot.x, x = rot.y, y = rot.z, z = rot.w; /* not normalised: forward.cu:127 */
    m3_f R = m3_make_f(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                       2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                       2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    m3_f M = m3_mul_f(&S, &R);
    m3_f Mt = m3_transpose_f(&M);
    m3_f Sigma = m3_mul_f(&Mt, &M);
    cov3D[0] = Sigma.c[0][0]; cov3D[1] = Sigma.c[0][1]; cov3D[2] = Sigma.c[0][2];
    cov3D[3] = Sigma.c[1][1]; cov3D[4] = Sigma.c[1][2]; cov3D[5] = Sigma.c[2][2];
}

#define HSRO_IMG_BOUNDS 6   /* tie_img_bound planes: colour (max over channels), depth, opacity, semantic (max over channels), final_T, mask */

typedef struct HsroState {
    int P, W, H, K, semantic, R, tiles_x, tiles_y, has_sh, own_cov3d;
    float* depths;          /* [P]   view-space z                         (GeometryState.depths)        */
    float* means2D;         /* [P,2] pixel centre                         (GeometryState.means2D)       */
    float* conic_opacity;   /* [P,4] conic xyz + opacity                  (GeometryState.conic_opacity) */
    float* cov3D;           /* [P,6]                                      (GeometryState.cov3D)         */
    float* rgb;             /* [P,3] SH colours                           (GeometryState.rgb)           */
    uint8_t* clamped;       /* [P,3]                                      (GeometryState.clamped)       */
    int* radii;             /* [P]                                                                      */
    uint32_t* tiles_touched;/* [P]                                                                      */
    uint32_t* point_offsets;/* [P] inclusive scan                                                       */
    uint64_t *keys_unsorted, *keys; /* [R]                                (BinningState)                */
    uint32_t *vals_unsorted, *vals; /* [R]                                                              */
    uint32_t* ranges;       /* [T,2]                                      (ImageState.ranges)           */
    real* final_T;          /* [N]                                        (ImageState.accum_alpha)      */
    uint32_t* n_contrib;    /* [N]                                        (ImageState.n_contrib)        */
    uint32_t* median_pos;   /* [N] 1 + list position of the splat at which T crossed 0.5 (0: never); not in the reference */
    /* Test aid, not in the reference (file header, "Threshold ties"): tie_pixels[pix] != 0: some decision of that pixel was taken
     * within HSRO_TIE_EPS (relative) of its threshold; tie_gaussians[id] != 0: the splat contributes to such a pixel (its gradient rows
     * see the difference); tie_img_bound[plane][pix]: how far the pixel's outputs move when the flagged decisions are taken the other
     * way (INFINITY where a pixel had more flagged decisions than are tracked). */
    uint8_t* tie_pixels;    /* [N] */
    uint8_t* tie_gaussians; /* [P] */
    real* tie_img_bound;    /* [HSRO_IMG_BOUNDS, N] */
    const float* feat;      /* colours the forward blended (caller's colors_precomp or rgb); valid while the caller keeps its inputs */
} HsroState;

#define HSRO_TIE_EPS 2e-6f   /* a few ulps of alpha; T drifts by less over a tile's list (same fp32 products, alpha differing in the last ulp) */

static int g_median_rule = 0;
static long g_median_disagree = 0;
/* Accumulation model of the backward's per-Gaussian sums.  0 (default, THE ORACLE): double, rounded once — order-independent.
 * 1: fp32, the tiles visited in a seeded random order, one thread: ONE of the orders in which the reference's fp32 atomicAdds
 * (backward.cu:616-663, :828-896) — or the HIP kernels' — can arrive.  Different seeds give different, equally valid fp32 results;
 * their spread is the noise floor no fp32-atomic implementation can go below (tests/harness.truth_report uses it to tell an
 * ill-conditioned gradient from a defect). */
static int g_accum_fp32 = 0;
static unsigned g_accum_seed = 0;
/* In the fp32 model the exponential may also carry the error a GPU's own exp has: G is multiplied by 1 + u * ulps * 2^-23 with u
 * uniform in [-1, 1], hashed from (seed, pixel, list position).  glibc's expf is within ~0.5 ulp; CUDA documents 2 ulp for expf (the
 * reference is built without fast-math: setup.py has no such flag), gfx950's v_exp_f32 / v_rcp_f32 1 ulp each.  Decisions (alpha >= 1/255
 * ...) keep using the unperturbed value: the model varies the arithmetic, not the lists. */
static float g_exp_ulps = 0.0f;
/* ... and the exponent's ARGUMENT the rounding of another correct fp32 formulation.  power = -0.5 (a dx^2 + c dy^2) - b dx dy is a sum of
 * three products; each fp32 evaluation order (the reference's with nvcc's FMA contraction, the oracle's IEEE order, the HIP kernels' pre-scaled
 * base-2 form) rounds each product and each partial sum once, i.e. errs by up to ~1.5 * 2^-24 * S with S = 0.5 (|a| dx^2 + |c| dy^2) + |b dx dy|
 * — NOT relative to power itself: for an elongated, rotated splat the three terms cancel (S >> |power|), and exp turns the absolute error of its argument
 * into a relative error of G of that size: dozens of ulps on a 20:1 needle, where one ulp of exp itself (above) is the smaller effect.
 * g_arg_roundings = the coefficient (0: off); G is multiplied by 1 + u2 * g_arg_roundings * 2^-24 * S, u2 uniform in [-1, 1], hashed. */
static float g_arg_roundings = 0.0f;
void hsro_set_exp_argument_error(float roundings) { g_arg_roundings = roundings > 0 ? roundings : 0.0f; }
void hsro_set_accumulation(int fp32, unsigned seed) { g_accum_fp32 = fp32 ? 1 : 0; g_accum_seed = seed; }
void hsro_set_exp_error(float ulps) { g_exp_ulps = ulps > 0 ? ulps : 0.0f; }
void hsro_set_median_rule(int rule) { g_median_rule = rule ? 1 : 0; }
long hsro_last_median_rule_disagreements(void) { return g_median_disagree; }

void hsro_free(HsroState* s)
{
    if (!s) return;
    free(s->depths); free(s->means2D); free(s->conic_opacity); free(s->cov3D); free(s->rgb); free(s->clamped);
    free(s->radii); free(s->tiles_touched); free(s->point_offsets); free(s->keys_unsorted); free(s->keys);
    free(s->vals_unsorted); free(s->vals); free(s->ranges); free(s->final_T); free(s->n_contrib);
    free(s->median_pos); free(s->tie_pixels); free(s->tie_gaussians); free(s->tie_img_bound);
    free(s);
}

int hsro_num_rendered(const HsroState* s) { return s->R; }
/* field ids for tests */
const void* hsro_field(const HsroState* s, int id)
{
    switch (id) {
    case 0: return s->depths; case 1: return s->means2D; case 2: return s->conic_opacity; case 3: return s->cov3D;
    case 4: return s->rgb; case 5: return s->clamped; case 6: return s->radii; case 7: return s->tiles_touched;
    case 8: return s->point_offsets; case 9: return s->keys_unsorted; case 10: return s->keys;
    case 11: return s->vals_unsorted; case 12: return s->vals; case 13: return s->ranges; case 14: return s->final_T;
    case 15: return s->n_contrib; case 16: return s->median_pos; case 17: return s->tie_pixels; case 18: return s->tie_gaussians;
    case 19: return s->tie_img_bound;
    default: return 0;
    }
}

/* stable LSD radix sort of (u64 key, u32 value) pairs on bits [0, end_bit) — the contract of
 * cub::DeviceRadixSort::SortPairs as called at rasterizer_impl.cu:307-312 / :570-575 */
static void stable_sort_pairs(const uint64_t* kin, const uint32_t* vin, uint64_t* kout, uint32_t* vout, size_t n, int end_bit)
{
    uint64_t* ka = (uint64_t*)malloc(sizeof(uint64_t) * (n ? n : 1));
    uint32_t* va = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint64_t* kb = (uint64_t*)malloc(sizeof(uint64_t) * (n ? n : 1));
    uint32_t* vb = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    memcpy(ka, kin, n * sizeof(uint64_t)); memcpy(va, vin, n * sizeof(uint32_t));
    size_t* cnt = (size_t*)malloc(sizeof(size_t) * 65537);
    for (int shift = 0; shift < end_bit; shift += 16) {
        int bits = end_bit - shift < 16 ? end_bit - shift : 16;
        uint64_t mask = ((uint64_t)1 << bits) - 1;
        memset(cnt, 0, sizeof(size_t) * 65537);
        for (size_t i = 0; i < n; i++) cnt[((ka[i] >> shift) & mask) + 1]++;
        for (size_t d = 0; d < 65536; d++) cnt[d + 1] += cnt[d];
        for (size_t i = 0; i < n; i++) { size_t d = (ka[i] >> shift) & mask; size_t o = cnt[d]++; kb[o] = ka[i]; vb[o] = va[i]; }
        uint64_t* tk = ka; ka = kb; kb = tk; uint32_t* tv = va; va = vb; vb = tv;
    }
    memcpy(kout, ka, n * sizeof(uint64_t)); memcpy(vout, va, n * sizeof(uint32_t));
    free(ka); free(va); free(kb); free(vb); free(cnt);
}

/* ---- one pixel of renderCUDA (forward.cu:261-398) / renderCUDA_SEM (forward.cu:400-538) ----
 * A flagged decision: list index + which test.  kind 1: power > 0; 2: alpha < 1/255; 3: T(1 - alpha) < 1e-4; 4: T crossing 0.5.
 * `ovr` (pos, kind 1..3) takes that one test the other way (the tie bounds); kind 0: none. */
typedef struct { uint32_t pos; int kind; } HsroDecision;
#define HSRO_MAX_DECISIONS 24
typedef struct {
    real C[NUM_CHANNELS], D, M, median_D, T;
    uint32_t last_contributor, median_at, i_end;
    int ndec, overflow;
    HsroDecision dec[HSRO_MAX_DECISIONS];
} HsroPixFwd;

static void pixel_forward(const HsroState* s, const float* feat, const float* semantics, int K, uint32_t r0, uint32_t r1,
                          float pfx, float pfy, HsroDecision ovr, int collect, real* Sacc, HsroPixFwd* o)
{
    real T = 1.0f; uint32_t contributor = 0, last_contributor = 0;
    real C[NUM_CHANNELS] = {0, 0, 0}; real Dd = 0; real median_D = 15.0f; real Mm = 0;
    uint32_t median_at = 0;
    for (int ch = 0; ch < K; ch++) Sacc[ch] = 0;
    o->ndec = 0; o->overflow = 0; o->i_end = r1;
#define FLAG(kind_) do { if (collect) { if (o->ndec < HSRO_MAX_DECISIONS) { o->dec[o->ndec].pos = i; o->dec[o->ndec].kind = (kind_); o->ndec++; } else o->overflow = 1; } } while (0)
    for (uint32_t i = r0; i < r1; i++) {
        contributor++;
        uint32_t id = s->vals[i];
        real dx = (real)s->means2D[2 * id] - (real)pfx, dy = (real)s->means2D[2 * id + 1] - (real)pfy;
        const float* co = s->conic_opacity + 4 * (size_t)id;
        real power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        /* how far two correct fp32 evaluations of `power` can be apart: a few roundings of its three terms (the reference's
         * own nvcc build contracts them into FMAs; the HIP path evaluates a pre-scaled form) — relative to alpha that is an
         * ABSOLUTE difference in power, which for elongated splats (large cancelling terms) is far more than an ulp of alpha */
        const real pw_slack = 4.0f * 5.96e-8f * (0.5f * (R_FABS(co[0] * dx * dx) + R_FABS(co[2] * dy * dy)) + R_FABS(co[1] * dx * dy));
        if (R_FABS(power) <= 1e-6f + pw_slack && co[3] >= 1.0f / 255.0f) FLAG(1);
        int skip = power > 0.0f;
        if (ovr.kind == 1 && ovr.pos == i) skip = !skip;
        if (skip) continue;
        real alpha = fmin_r(0.99f, co[3] * R_EXP(power));
        if (R_FABS(alpha - 1.0f / 255.0f) <= (HSRO_TIE_EPS + pw_slack) * (1.0f / 255.0f)) FLAG(2);
        skip = alpha < 1.0f / 255.0f;
        if (ovr.kind == 2 && ovr.pos == i) skip = !skip;
        if (skip) continue;
        real test_T = T * (1 - alpha);
        if (R_FABS(test_T - 0.0001f) <= 4.0f * HSRO_TIE_EPS * 0.0001f) FLAG(3);
        if (R_FABS(T - 0.5f) <= 4.0f * HSRO_TIE_EPS * 0.5f || R_FABS(test_T - 0.5f) <= 4.0f * HSRO_TIE_EPS * 0.5f) FLAG(4);
        int done = test_T < 0.0001f;
        if (ovr.kind == 3 && ovr.pos == i) done = !done;
        if (done) { o->i_end = i + 1; break; } /* done = true (forward.cu:358-362, :496-500) */
        for (int ch = 0; ch < NUM_CHANNELS; ch++) C[ch] += feat[(size_t)id * NUM_CHANNELS + ch] * alpha * T;
        Dd += s->depths[id] * alpha * T;
        if (s->semantic) { for (int ch = 0; ch < K; ch++) Sacc[ch] += semantics[(size_t)id * K + ch] * alpha * T; }
        else Mm += alpha * T;
        if (T > 0.5f && test_T < 0.5) { median_D = s->depths[id]; median_at = contributor; } /* forward.cu:371-376, :511-515 */
        T = test_T;
        last_contributor = contributor;
    }
#undef FLAG
    for (int ch = 0; ch < NUM_CHANNELS; ch++) o->C[ch] = C[ch];
    o->D = Dd; o->M = Mm; o->median_D = median_D; o->T = T; o->last_contributor = last_contributor; o->median_at = median_at;
}

/*
 * Forward.  Follows Rasterizer::forward (rasterizer_impl.cu:198-345) when `semantics == NULL`
 * (outputs colour, depth, median depth, opacity, mask) and Rasterizer::forward_semantic
 * (rasterizer_impl.cu:460-610) otherwise (colour, semantic[K], depth, median depth, opacity).
 * All pointers are host pointers; absent optionals are NULL (the reference's `data_ptr()==nullptr`
 * switches, rasterizer_impl.cu:588, forward.cu:205, :241).  Image outputs are `real` (float, or double in the truth build).
 * Returns a state handle (the reference's geom/binning/img buffers) or NULL on allocation failure.
 */
HsroState* hsro_forward(int P, int D, int M, int K, const float* background, int width, int height,
                        const float* means3D, const float* shs, const float* colors_precomp, const float* semantics,
                        const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                        const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                        const float* cam_pos, float tan_fovx, float tan_fovy, real* out_color, real* out_semantic,
                        real* out_depth, real* out_median_depth, real* out_opacity, real* out_mask, int* radii_out)
{
    (void)background; /* forward never composites the background (forward.cu:391-392, :530-531) */
    HsroState* s = (HsroState*)calloc(1, sizeof(HsroState));
    if (!s) return 0;
    const int W = width, H = height;
    const size_t N = (size_t)W * H;
    s->P = P; s->W = W; s->H = H; s->K = K; s->semantic = semantics != 0;
    s->tiles_x = (W + BLOCK_X - 1) / BLOCK_X; s->tiles_y = (H + BLOCK_Y - 1) / BLOCK_Y;
    const uint32_t gx = (uint32_t)s->tiles_x, gy = (uint32_t)s->tiles_y;
    const size_t Tn = (size_t)gx * gy;
    size_t Pa = P ? (size_t)P : 1;
    s->depths = (float*)calloc(Pa, 4); s->means2D = (float*)calloc(Pa * 2, 4); s->conic_opacity = (float*)calloc(Pa * 4, 4);
    s->cov3D = (float*)calloc(Pa * 6, 4); s->rgb = (float*)calloc(Pa * 3, 4); s->clamped = (uint8_t*)calloc(Pa * 3, 1);
    s->radii = (int*)calloc(Pa, 4); s->tiles_touched = (uint32_t*)calloc(Pa, 4); s->point_offsets = (uint32_t*)calloc(Pa, 4);
    s->ranges = (uint32_t*)calloc(Tn * 2, 4); s->final_T = (real*)calloc(N, sizeof(real)); s->n_contrib = (uint32_t*)calloc(N, 4);
    s->median_pos = (uint32_t*)calloc(N, 4);
    s->tie_pixels = (uint8_t*)calloc(N, 1); s->tie_gaussians = (uint8_t*)calloc(Pa, 1);
    s->tie_img_bound = (real*)calloc(N * HSRO_IMG_BOUNDS, sizeof(real));
    s->has_sh = colors_precomp == 0; s->own_cov3d = cov3D_precomp == 0;

    /* rasterizer_impl.cu:226-227 */
    const float focal_y = height / (2.0f * tan_fovy);
    const float focal_x = width / (2.0f * tan_fovx);

    /* ---- preprocessCUDA, forward.cu:155-256 (fp32 in every build) ---- */
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        s->radii[idx] = 0; s->tiles_touched[idx] = 0;
        v3_f p_view;
        if (!in_frustum(idx, means3D, viewmatrix, projmatrix, &p_view)) continue;
        v3_f p_orig = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
        v4_f p_hom = transformPoint4x4_f(p_orig, projmatrix);
        float p_w = 1.0f / (p_hom.w + 0.0000001f);
        v3_f p_proj = {p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w};
        const float* cov3D;
        if (cov3D_precomp) cov3D = cov3D_precomp + (size_t)idx * 6;
        else {
            v3_f sc = {scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]};
            v4_f q = {rotations[4 * idx], rotations[4 * idx + 1], rotations[4 * idx + 2], rotations[4 * idx + 3]};
            computeCov3D(sc, scale_modifier, q, s->cov3D + (size_t)idx * 6);
            cov3D = s->cov3D + (size_t)idx * 6;
        }
        v3_f t; float txtz, tytz; m3_f T, Vrk, cv;
        cov2d_core_f(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, &t, &txtz, &tytz, &T, &Vrk, &cv);
        cv.c[0][0] += 0.3f; cv.c[1][1] += 0.3f;                       /* forward.cu:110-111 */
        const float cx = cv.c[0][0], cy = cv.c[0][1], cz = cv.c[1][1];
        float det = (cx * cz - cy * cy);                                /* forward.cu:219 */
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conx = cz * det_inv, cony = -cy * det_inv, conz = cx * det_inv;
        float mid = 0.5f * (cx + cz);
        float lambda1 = mid + sqrtf(fmax_f(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmax_f(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmax_f(lambda1, lambda2)));
        float pix = ndc2Pix(p_proj.x, W), piy = ndc2Pix(p_proj.y, H);
        uint32_t rminx, rminy, rmaxx, rmaxy;
        getRect(pix, piy, (int)my_radius, &rminx, &rminy, &rmaxx, &rmaxy, gx, gy);
        if ((rmaxx - rminx) * (rmaxy - rminy) == 0) continue;
        if (colors_precomp == 0) {
            v3_f c = computeColorFromSH(idx, D, M, means3D, cam_pos, shs, s->clamped);
            s->rgb[3 * idx] = c.x; s->rgb[3 * idx + 1] = c.y; s->rgb[3 * idx + 2] = c.z;
        }
        s->depths[idx] = p_view.z;
        s->radii[idx] = (int)my_radius;
        s->means2D[2 * idx] = pix; s->means2D[2 * idx + 1] = piy;
        s->conic_opacity[4 * idx] = conx; s->conic_opacity[4 * idx + 1] = cony; s->conic_opacity[4 * idx + 2] = conz;
        s->conic_opacity[4 * idx + 3] = opacities[idx];
        s->tiles_touched[idx] = (rmaxy - rminy) * (rmaxx - rminx);
    }
    if (radii_out) memcpy(radii_out, s->radii, sizeof(int) * (size_t)P);

    /* ---- InclusiveSum, rasterizer_impl.cu:281 / :544 ---- */
    { uint32_t acc = 0; for (int i = 0; i < P; i++) { acc += s->tiles_touched[i]; s->point_offsets[i] = acc; } }
    const int R = P > 0 ? (int)s->point_offsets[P - 1] : 0;          /* rasterizer_impl.cu:285 */
    s->R = R;
    size_t Ra = R ? (size_t)R : 1;
    s->keys_unsorted = (uint64_t*)calloc(Ra, 8); s->keys = (uint64_t*)calloc(Ra, 8);
    s->vals_unsorted = (uint32_t*)calloc(Ra, 4); s->vals = (uint32_t*)calloc(Ra, 4);

    /* ---- duplicateWithKeys, rasterizer_impl.cu:70-111 ---- */
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (s->radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : s->point_offsets[idx - 1];
            uint32_t rminx, rminy, rmaxx, rmaxy;
            getRect(s->means2D[2 * idx], s->means2D[2 * idx + 1], s->radii[idx], &rminx, &rminy, &rmaxx, &rmaxy, gx, gy);
            uint32_t dbits; memcpy(&dbits, &s->depths[idx], 4);
            for (uint32_t y = rminy; y < rmaxy; y++)
                for (uint32_t x = rminx; x < rmaxx; x++) {
                    uint64_t key = (uint64_t)(y * gx + x);
                    key <<= 32; key |= dbits;
                    s->keys_unsorted[off] = key; s->vals_unsorted[off] = (uint32_t)idx; off++;
                }
        }
    }
    /* ---- SortPairs on bits [0, 32+bit), rasterizer_impl.cu:304-312 ---- */
    const int bit = (int)hsro_get_higher_msb(gx * gy);
    stable_sort_pairs(s->keys_unsorted, s->vals_unsorted, s->keys, s->vals, (size_t)R, 32 + bit);

    /* ---- memset + identifyTileRanges, rasterizer_impl.cu:314-322, :116-138 ---- */
    for (int i = 0; i < R; i++) {
        uint32_t cur = (uint32_t)(s->keys[i] >> 32);
        if (i == 0) s->ranges[2 * cur] = 0;
        else {
            uint32_t prev = (uint32_t)(s->keys[i - 1] >> 32);
            if (cur != prev) { s->ranges[2 * prev + 1] = (uint32_t)i; s->ranges[2 * cur] = (uint32_t)i; }
        }
        if (i == R - 1) s->ranges[2 * cur + 1] = (uint32_t)R;
    }

    /* ---- renderCUDA (forward.cu:261-398) / renderCUDA_SEM (forward.cu:400-538), one pixel per "thread" ---- */
    const float* feat = colors_precomp ? colors_precomp : s->rgb;
    s->feat = feat;
    const HsroDecision no_ovr = {0, 0};
#pragma omp parallel for schedule(dynamic, 1)
    for (long tile = 0; tile < (long)Tn; tile++) {
        const uint32_t ty = (uint32_t)(tile / gx), tx = (uint32_t)(tile % gx);
        const uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
        real* Sacc = (real*)malloc(sizeof(real) * (size_t)(K > 0 ? 3 * K : 1));
        real* Salt = (real*)malloc(sizeof(real) * (size_t)(K > 0 ? 3 * K : 1));
        for (int tyy = 0; tyy < BLOCK_Y; tyy++)
            for (int txx = 0; txx < BLOCK_X; txx++) {
                uint32_t px = tx * BLOCK_X + txx, py = ty * BLOCK_Y + tyy;
                if (!(px < (uint32_t)W && py < (uint32_t)H)) continue;
                size_t pix_id = (size_t)W * py + px;
                float pfx = (float)px, pfy = (float)py;
                HsroPixFwd f;
                pixel_forward(s, feat, semantics, K, r0, r1, pfx, pfy, no_ovr, 1, Sacc, &f);
                s->final_T[pix_id] = f.T; s->n_contrib[pix_id] = f.last_contributor; s->median_pos[pix_id] = f.median_at;
                if (f.ndec || f.overflow) {
                    s->tie_pixels[pix_id] = 1;
                    /* every splat that reaches (or nearly reaches) alpha >= 1/255 on this pixel sees the difference */
                    for (uint32_t i = r0; i < f.i_end; i++) {
                        uint32_t id = s->vals[i];
                        real dx = (real)s->means2D[2 * id] - (real)pfx, dy = (real)s->means2D[2 * id + 1] - (real)pfy;
                        const float* co = s->conic_opacity + 4 * (size_t)id;
                        real power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        const real pw_slack = 4.0f * 5.96e-8f * (0.5f * (R_FABS(co[0] * dx * dx) + R_FABS(co[2] * dy * dy)) + R_FABS(co[1] * dx * dy));
                        if (power > 1e-6f + pw_slack) continue;
                        if (fmin_r(0.99f, co[3] * R_EXP(power)) >= (1.0f / 255.0f) * (1.0f - HSRO_TIE_EPS - pw_slack)) s->tie_gaussians[id] = 1;   /* benign race: all writers store 1 */
                    }
                    /* the bound: each flagged decision taken the other way, one at a time */
                    real* b = s->tie_img_bound;
                    for (int d = 0; d < f.ndec; d++) {
                        if (f.dec[d].kind == 4) continue;   /* the crossing moves the median depth only: counted, not bounded (median_pos) */
                        HsroPixFwd g;
                        pixel_forward(s, feat, semantics, K, r0, r1, pfx, pfy, f.dec[d], 0, Salt, &g);
                        real dc = 0, ds = 0;
                        for (int ch = 0; ch < NUM_CHANNELS; ch++) dc = fmax_r(dc, R_FABS(g.C[ch] - f.C[ch]));
                        if (s->semantic) for (int ch = 0; ch < K; ch++) ds = fmax_r(ds, R_FABS(Salt[ch] - Sacc[ch]));
                        b[0 * N + pix_id] += dc; b[1 * N + pix_id] += R_FABS(g.D - f.D); b[2 * N + pix_id] += R_FABS(g.T - f.T);
                        b[3 * N + pix_id] += ds; b[4 * N + pix_id] += R_FABS(g.T - f.T); b[5 * N + pix_id] += R_FABS(g.M - f.M);
                    }
                    if (f.overflow) for (int pl = 0; pl < HSRO_IMG_BOUNDS; pl++) b[pl * N + pix_id] = INFINITY;
                }
                for (int ch = 0; ch < NUM_CHANNELS; ch++) out_color[(size_t)ch * N + pix_id] = f.C[ch];
                out_depth[pix_id] = f.D; out_median_depth[pix_id] = f.median_D; out_opacity[pix_id] = 1 - f.T;
                if (s->semantic) { for (int ch = 0; ch < K; ch++) out_semantic[(size_t)ch * N + pix_id] = Sacc[ch]; }
                else if (out_mask) out_mask[pix_id] = f.M;
            }
        free(Sacc); free(Salt);
    }
    return s;
}

/* backward.cu:20-139 */
static void computeColorFromSH_bwd(int idx, int deg, int max_coeffs, const float* means, const float* campos, const float* shs,
                                   const uint8_t* clamped, const real* dL_dcolor3, real* dL_dmeans3, real* dsh)
{
    v3_r pos = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
    v3_r dir_orig = {pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]};
    real len = sqrt_r(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
    real x = dir_orig.x / len, y = dir_orig.y / len, z = dir_orig.z / len;
    const float* sh = shs + (size_t)idx * max_coeffs * 3;
    real dRGB[3];
    for (int c = 0; c < 3; c++) dRGB[c] = dL_dcolor3[c] * (clamped[3 * idx + c] ? 0.f : 1.f);
    real dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
#define SH(i) sh[(i) * 3 + c]
#define DSH(i, v) for (int c = 0; c < 3; c++) dsh[(i) * 3 + c] = (v) * dRGB[c]
    DSH(0, SH_C0);
    if (deg > 0) {
        DSH(1, -SH_C1 * y); DSH(2, SH_C1 * z); DSH(3, -SH_C1 * x);
        for (int c = 0; c < 3; c++) { dRGBdx[c] = -SH_C1 * SH(3); dRGBdy[c] = -SH_C1 * SH(1); dRGBdz[c] = SH_C1 * SH(2); }
        if (deg > 1) {
            real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            DSH(4, SH_C2[0] * xy); DSH(5, SH_C2[1] * yz); DSH(6, SH_C2[2] * (2.f * zz - xx - yy));
            DSH(7, SH_C2[3] * xz); DSH(8, SH_C2[4] * (xx - yy));
            for (int c = 0; c < 3; c++) {
                dRGBdx[c] += SH_C2[0] * y * SH(4) + SH_C2[2] * 2.f * -x * SH(6) + SH_C2[3] * z * SH(7) + SH_C2[4] * 2.f * x * SH(8);
                dRGBdy[c] += SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * 2.f * -y * SH(6) + SH_C2[4] * 2.f * -y * SH(8);
                dRGBdz[c] += SH_C2[1] * y * SH(5) + SH_C2[2] * 2.f * 2.f * z * SH(6) + SH_C2[3] * x * SH(7);
            }
            if (deg > 2) {
                DSH(9, SH_C3[0] * y * (3.f * xx - yy)); DSH(10, SH_C3[1] * xy * z); DSH(11, SH_C3[2] * y * (4.f * zz - xx - yy));
                DSH(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)); DSH(13, SH_C3[4] * x * (4.f * zz - xx - yy));
                DSH(14, SH_C3[5] * z * (xx - yy)); DSH(15, SH_C3[6] * x * (xx - 3.f * yy));
                for (int c = 0; c < 3; c++) {
                    dRGBdx[c] += (SH_C3[0] * SH(9) * 3.f * 2.f * xy + SH_C3[1] * SH(10) * yz + SH_C3[2] * SH(11) * -2.f * xy +
                                  SH_C3[3] * SH(12) * -3.f * 2.f * xz + SH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                                  SH_C3[5] * SH(14) * 2.f * xz + SH_C3[6] * SH(15) * 3.f * (xx - yy));
                    dRGBdy[c] += (SH_C3[0] * SH(9) * 3.f * (xx - yy) + SH_C3[1] * SH(10) * xz +
                                  SH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * SH(12) * -3.f * 2.f * yz +
                                  SH_C3[4] * SH(13) * -2.f * xy + SH_C3[5] * SH(14) * -2.f * yz + SH_C3[6] * SH(15) * -3.f * 2.f * xy);
                    dRGBdz[c] += (SH_C3[1] * SH(10) * xy + SH_C3[2] * SH(11) * 4.f * 2.f * yz +
                                  SH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * SH(13) * 4.f * 2.f * xz +
                                  SH_C3[5] * SH(14) * (xx - yy));
                }
            }
        }
    }
#undef SH
#undef DSH
    v3_r dL_ddir = {dRGBdx[0] * dRGB[0] + dRGBdx[1] * dRGB[1] + dRGBdx[2] * dRGB[2],
                    dRGBdy[0] * dRGB[0] + dRGBdy[1] * dRGB[1] + dRGBdy[2] * dRGB[2],
                    dRGBdz[0] * dRGB[0] + dRGBdz[1] * dRGB[1] + dRGBdz[2] * dRGB[2]};
    v3_r dm = dnormvdv3_r(dir_orig, dL_ddir);
    dL_dmeans3[0] += dm.x; dL_dmeans3[1] += dm.y; dL_dmeans3[2] += dm.z;
}

/* ---- the per-Gaussian chain rule: computeCov2DCUDA (backward.cu:144-274) then preprocessCUDA (backward.cu:346-412, also the
 * semantic path, :1117) for ONE Gaussian, as a function of the sums the tile pass accumulated for it.  It is linear in those
 * sums, which is what lets the tie bounds carry a per-splat DIFFERENCE of the sums through it. ---- */
typedef struct {
    const HsroState* s;
    int D, M;
    const float *means3D, *shs, *scales, *rotations, *cov3Ds, *viewmatrix, *projmatrix, *campos;
    float scale_modifier, focal_x, focal_y, tan_fovx, tan_fovy;
} HsroChain;
/* in: dL_dmean2D.xy, dL_dconic.{x,y,w}, dL_ddepth, dL_dcolor.rgb (the last only feeds the SH path).
 * out: dL_dmean3D[3], dL_dcov3D[6], dL_dscale[3] / dL_drot[4] (when scales given), dL_dsh[M,3] (when shs given; may be NULL) */
static void gauss_chain(const HsroChain* c, int idx, const real in[9], real* dmean3D, real* dcov, real* dscale, real* drot, real* dsh)
{
    const float* cov3D = c->cov3Ds + 6 * (size_t)idx;
    const float* means3D = c->means3D;
    v3_r mean = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    const real dcx = in[2], dcy = in[3], dcz = in[4];
    v3_r t; real txtz, tytz; m3_r T, Vrk, c2;
    cov2d_core_r(mean, c->focal_x, c->focal_y, c->tan_fovx, c->tan_fovy, cov3D, c->viewmatrix, &t, &txtz, &tytz, &T, &Vrk, &c2);
    const real limx = 1.3f * c->tan_fovx, limy = 1.3f * c->tan_fovy;
    const real x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    const real y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    real a = c2.c[0][0] + 0.3f, b = c2.c[0][1], cc = c2.c[1][1] + 0.3f;
    real denom = a * cc - b * b;
    real dL_da = 0, dL_db = 0, dL_dc = 0;
    real denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
#define TT(i, j) T.c[i][j]
#define VV(i, j) Vrk.c[i][j]
    if (denom2inv != 0) {
        dL_da = denom2inv * (-cc * cc * dcx + 2 * b * cc * dcy + (denom - a * cc) * dcz);
        dL_dc = denom2inv * (-a * a * dcz + 2 * a * b * dcy + (denom - a * cc) * dcx);
        dL_db = denom2inv * 2 * (b * cc * dcx - (denom + 2 * b * b) * dcy + a * b * dcz);
        dcov[0] = (TT(0, 0) * TT(0, 0) * dL_da + TT(0, 0) * TT(1, 0) * dL_db + TT(1, 0) * TT(1, 0) * dL_dc);
        dcov[3] = (TT(0, 1) * TT(0, 1) * dL_da + TT(0, 1) * TT(1, 1) * dL_db + TT(1, 1) * TT(1, 1) * dL_dc);
        dcov[5] = (TT(0, 2) * TT(0, 2) * dL_da + TT(0, 2) * TT(1, 2) * dL_db + TT(1, 2) * TT(1, 2) * dL_dc);
        dcov[1] = 2 * TT(0, 0) * TT(0, 1) * dL_da + (TT(0, 0) * TT(1, 1) + TT(0, 1) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 1) * dL_dc;
        dcov[2] = 2 * TT(0, 0) * TT(0, 2) * dL_da + (TT(0, 0) * TT(1, 2) + TT(0, 2) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 2) * dL_dc;
        dcov[4] = 2 * TT(0, 2) * TT(0, 1) * dL_da + (TT(0, 1) * TT(1, 2) + TT(0, 2) * TT(1, 1)) * dL_db + 2 * TT(1, 1) * TT(1, 2) * dL_dc;
    } else {
        for (int i = 0; i < 6; i++) dcov[i] = 0;
    }
    real dL_dT00 = 2 * (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_da +
                   (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_db;
    real dL_dT01 = 2 * (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_da +
                   (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_db;
    real dL_dT02 = 2 * (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_da +
                   (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_db;
    real dL_dT10 = 2 * (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_dc +
                   (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_db;
    real dL_dT11 = 2 * (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_dc +
                   (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_db;
    real dL_dT12 = 2 * (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_dc +
                   (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_db;
#undef TT
#undef VV
    /* W as built at backward.cu:182-185: W[c][r] */
    const float* vm = c->viewmatrix;
    m3_r Wm = m3_make_r(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
    real dL_dJ00 = Wm.c[0][0] * dL_dT00 + Wm.c[0][1] * dL_dT01 + Wm.c[0][2] * dL_dT02;
    real dL_dJ02 = Wm.c[2][0] * dL_dT00 + Wm.c[2][1] * dL_dT01 + Wm.c[2][2] * dL_dT02;
    real dL_dJ11 = Wm.c[1][0] * dL_dT10 + Wm.c[1][1] * dL_dT11 + Wm.c[1][2] * dL_dT12;
    real dL_dJ12 = Wm.c[2][0] * dL_dT10 + Wm.c[2][1] * dL_dT11 + Wm.c[2][2] * dL_dT12;
    real tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
    const float h_x = c->focal_x, h_y = c->focal_y;
    real dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
    real dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
    real dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12;
    v3_r dt = {dL_dtx, dL_dty, dL_dtz};
    v3_r dm = transformVec4x3Transpose_r(dt, c->viewmatrix);
    dmean3D[0] = dm.x; dmean3D[1] = dm.y; dmean3D[2] = dm.z; /* assignment, :273 */

    /* ---- preprocessCUDA (backward), backward.cu:346-412 ---- */
    v3_r m = mean;
    const float* proj = c->projmatrix; const float* view = c->viewmatrix;
    v4_r m_hom = transformPoint4x4_r(m, proj);
    real m_w = 1.0f / (m_hom.w + 0.0000001f);
    real mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
    real mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
    const real d2x = in[0], d2y = in[1];
    real gxm = (proj[0] * m_w - proj[3] * mul1) * d2x + (proj[1] * m_w - proj[3] * mul2) * d2y;
    real gym = (proj[4] * m_w - proj[7] * mul1) * d2x + (proj[5] * m_w - proj[7] * mul2) * d2y;
    real gzm = (proj[8] * m_w - proj[11] * mul1) * d2x + (proj[9] * m_w - proj[11] * mul2) * d2y;
    dmean3D[0] += gxm; dmean3D[1] += gym; dmean3D[2] += gzm;
    real mul3 = view[2] * m.x + view[6] * m.y + view[10] * m.z + view[14];
    const real dd = in[5];
    dmean3D[0] += (view[2] - view[3] * mul3) * dd;
    dmean3D[1] += (view[6] - view[7] * mul3) * dd;
    dmean3D[2] += (view[10] - view[11] * mul3) * dd;
    if (c->shs && dsh) computeColorFromSH_bwd(idx, c->D, c->M, means3D, c->campos, c->shs, c->s->clamped, in + 6, dmean3D, dsh);
    if (c->scales && dscale && drot) {
        /* computeCov3D backward, backward.cu:278-341 */
        const float* rotations = c->rotations; const float* scales = c->scales;
        real r = rotations[4 * idx], x = rotations[4 * idx + 1], y = rotations[4 * idx + 2], z = rotations[4 * idx + 3];
        m3_r Rm = m3_make_r(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                            2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                            2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
        m3_r S = m3_make_r(1, 0, 0, 0, 1, 0, 0, 0, 1);
        real sx = c->scale_modifier * scales[3 * idx], sy = c->scale_modifier * scales[3 * idx + 1], sz = c->scale_modifier * scales[3 * idx + 2];
        S.c[0][0] = sx; S.c[1][1] = sy; S.c[2][2] = sz;
        m3_r Mm = m3_mul_r(&S, &Rm);
        const real* dc = dcov;
        m3_r dSig = m3_make_r(dc[0], 0.5f * dc[1], 0.5f * dc[2], 0.5f * dc[1], dc[3], 0.5f * dc[4], 0.5f * dc[2], 0.5f * dc[4], dc[5]);
        m3_r M2 = Mm;
        for (int k = 0; k < 3; k++) for (int rr = 0; rr < 3; rr++) M2.c[k][rr] = 2.0f * Mm.c[k][rr]; /* 2.0f * M */
        m3_r dL_dM = m3_mul_r(&M2, &dSig);
        m3_r Rt = m3_transpose_r(&Rm);
        m3_r dMt = m3_transpose_r(&dL_dM);
        dscale[0] = Rt.c[0][0] * dMt.c[0][0] + Rt.c[0][1] * dMt.c[0][1] + Rt.c[0][2] * dMt.c[0][2];
        dscale[1] = Rt.c[1][0] * dMt.c[1][0] + Rt.c[1][1] * dMt.c[1][1] + Rt.c[1][2] * dMt.c[1][2];
        dscale[2] = Rt.c[2][0] * dMt.c[2][0] + Rt.c[2][1] * dMt.c[2][1] + Rt.c[2][2] * dMt.c[2][2];
        for (int k = 0; k < 3; k++) { dMt.c[0][k] *= sx; dMt.c[1][k] *= sy; dMt.c[2][k] *= sz; }
#define DM(i, j) dMt.c[i][j]
        real qx = 2 * z * (DM(0, 1) - DM(1, 0)) + 2 * y * (DM(2, 0) - DM(0, 2)) + 2 * x * (DM(1, 2) - DM(2, 1));
        real qy = 2 * y * (DM(1, 0) + DM(0, 1)) + 2 * z * (DM(2, 0) + DM(0, 2)) + 2 * r * (DM(1, 2) - DM(2, 1)) - 4 * x * (DM(2, 2) + DM(1, 1));
        real qz = 2 * x * (DM(1, 0) + DM(0, 1)) + 2 * r * (DM(2, 0) - DM(0, 2)) + 2 * z * (DM(1, 2) + DM(2, 1)) - 4 * y * (DM(2, 2) + DM(0, 0));
        real qw = 2 * r * (DM(0, 1) - DM(1, 0)) + 2 * x * (DM(2, 0) + DM(0, 2)) + 2 * y * (DM(1, 2) + DM(2, 1)) - 4 * z * (DM(1, 1) + DM(0, 0));
#undef DM
        drot[0] = qx; drot[1] = qy; drot[2] = qz; drot[3] = qw;
    }
}

/* ---- one pixel of renderCUDA (backward.cu:472-666) / renderCUDA_SEM (backward.cu:669-899) ----
 * Adds the pixel's per-splat terms into rows of NA = 10 + K doubles: mean2D xy, conic xyw, opacity, colour rgb, depth, sem[K].
 * local == NULL: into acc[id] (shared, atomic); else into local[list position - r0] (the tie bounds' private copy). */
typedef struct {
    const HsroState* s;
    const float *colors, *background, *dL_dpix, *dL_dpix_sem, *dL_dpix_depth, *dL_dpix_median, *dL_dpix_opacity;
    const float* semantics;   /* [P,K] features: read only in sem_alpha_mode 1 */
    int sem_alpha_mode;       /* 0: reference as observed (the semantic loss never reaches alpha); 1: "exact" (what backward.cu:778-779, :834-845 intended) */
    size_t N; int K, NA;
    real ddelx_dx, ddely_dy;
    int median_rule;
} HsroBwdCtx;

static int pixel_backward(const HsroBwdCtx* c, size_t pix_id, float pfx, float pfy, uint32_t r0, uint32_t r1, real T_final,
                          int last_contributor, uint32_t median_at, HsroDecision ovr, real* dsem, double* acc, double* local, float* acc32)
{
    const HsroState* s = c->s;
    const size_t N = c->N; const int K = c->K, NA = c->NA;
    real T = T_final;
    uint32_t contributor = r1 - r0;
    real accum_rec[NUM_CHANNELS] = {0, 0, 0}, dpx[NUM_CHANNELS], last_color[NUM_CHANNELS] = {0, 0, 0};
    for (int ch = 0; ch < NUM_CHANNELS; ch++) dpx[ch] = c->dL_dpix[(size_t)ch * N + pix_id];
    const real dpd = c->dL_dpix_depth[pix_id], dpm = c->dL_dpix_median[pix_id], dpo = c->dL_dpix_opacity[pix_id];
    for (int ch = 0; ch < K; ch++) dsem[ch] = c->dL_dpix_sem[(size_t)ch * N + pix_id];
    real accum_depth_rec = 0, accum_op_rec = 0, last_alpha = 0, last_depth = 0, last_op = 0;
    int pixel_disagrees = 0;
    /* exact mode: the semantic channels' accum_rec / last value, as the colour channels have (dsem holds 3 K reals then) */
    real* accum_rec_sem = dsem + K;
    real* last_sem = dsem + 2 * K;
    if (c->sem_alpha_mode) for (int ch = 0; ch < 2 * K; ch++) accum_rec_sem[ch] = 0;
#define ADD(col, v) do { if (local) a[col] += (double)(v); else if (acc32) a32[col] += (float)(v); else { _Pragma("omp atomic") a[col] += (double)(v); } } while (0)
    for (uint32_t ii = r1; ii > r0; ii--) { /* back to front, backward.cu:562, :771 */
        contributor--;
        if ((int64_t)contributor >= (int64_t)last_contributor) continue;
        uint32_t id = s->vals[ii - 1];
        real dx = (real)s->means2D[2 * id] - (real)pfx, dy = (real)s->means2D[2 * id + 1] - (real)pfy;
        const float* co = s->conic_opacity + 4 * (size_t)id;
        real power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        int skip = power > 0.0f;
        if (ovr.kind == 1 && ovr.pos == ii - 1) skip = !skip;
        if (skip) continue;
        real G = R_EXP(power);
        real alpha = fmin_r(0.99f, co[3] * G);
        skip = alpha < 1.0f / 255.0f;
        if (ovr.kind == 2 && ovr.pos == ii - 1) skip = !skip;
        if (skip) continue;
        if (acc32 && g_exp_ulps > 0.0f) {   /* fp32 model: an exp that is off by up to g_exp_ulps ulps (hsro_set_exp_error) */
            uint64_t hsh = ((uint64_t)pix_id * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)ii * 0xC2B2AE3D27D4EB4Full) ^ ((uint64_t)g_accum_seed << 32);
            hsh ^= hsh >> 29; hsh *= 0xBF58476D1CE4E5B9ull; hsh ^= hsh >> 32;
            const float u = (float)((double)(hsh & 0xFFFFFFu) / 8388607.5 - 1.0);
            G = G * (1.0f + u * g_exp_ulps * 1.1920929e-7f);
            alpha = fmin_r(0.99f, co[3] * G);
        }
        if (acc32 && g_arg_roundings > 0.0f) {   /* fp32 model: the exponent's argument as another evaluation order rounds it (hsro_set_exp_argument_error) */
            uint64_t hsh = ((uint64_t)pix_id * 0xD6E8FEB86659FD93ull) ^ ((uint64_t)ii * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)g_accum_seed << 32) ^ 0x5851F42D4C957F2Dull;
            hsh ^= hsh >> 29; hsh *= 0xBF58476D1CE4E5B9ull; hsh ^= hsh >> 32;
            const float u2 = (float)((double)(hsh & 0xFFFFFFu) / 8388607.5 - 1.0);
            const float S = 0.5f * (fabsf(co[0]) * (float)(dx * dx) + fabsf(co[2]) * (float)(dy * dy)) + fabsf(co[1] * (float)(dx * dy));
            G = G * (1.0f + u2 * g_arg_roundings * 5.9604645e-8f * S);
            alpha = fmin_r(0.99f, co[3] * G);
        }
        real test_T = T / (1.f - alpha);
        const real w = alpha * test_T;
        double* a = local ? local + (size_t)(ii - 1 - r0) * NA : acc + (size_t)id * NA;
        float* a32 = acc32 ? acc32 + (size_t)id * NA : 0;
        real dL_dalpha = 0.0f;
        for (int ch = 0; ch < NUM_CHANNELS; ch++) {
            const real cval = c->colors[(size_t)id * NUM_CHANNELS + ch];
            accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
            last_color[ch] = cval;
            dL_dalpha += (cval - accum_rec[ch]) * dpx[ch];
            real v = w * dpx[ch];
            ADD(6 + ch, v);
        }
        /* semantic: s == 0 (unwritten scratch), accum_rec_sem and last_semantic stay 0, so the
           dL_dalpha term (backward.cu:840) is exactly 0; only dL_dsemantics accumulates (:845) */
        for (int ch = 0; ch < K; ch++) {
            if (c->sem_alpha_mode) {   /* backward.cu:834-845 with the staging of :778-779 in place: the channel's own (c - accum_rec) term */
                const real cval = c->semantics[(size_t)id * K + ch];
                accum_rec_sem[ch] = last_alpha * last_sem[ch] + (1.f - last_alpha) * accum_rec_sem[ch];
                last_sem[ch] = cval;
                dL_dalpha += (cval - accum_rec_sem[ch]) * dsem[ch];
            }
            real v = w * dsem[ch];
            ADD(10 + ch, v);
        }
        const real c_d = s->depths[id];
        accum_depth_rec = last_alpha * last_depth + (1.f - last_alpha) * accum_depth_rec;
        last_depth = c_d;
        dL_dalpha += (c_d - accum_depth_rec) * dpd;
        {
            real v = w * dpd;
            ADD(9, v);
        }
        {
            const int by_reference_rule = (test_T > 0.5f && T < 0.5); /* backward.cu:623-626, :854-857 */
            const int by_forward_record = (contributor + 1 == median_at);
            if (by_reference_rule != by_forward_record) pixel_disagrees = 1;
            if (c->median_rule ? by_forward_record : by_reference_rule) ADD(9, dpm);
        }
        accum_op_rec = last_alpha * last_op + (1.f - last_alpha) * accum_op_rec;
        last_op = 1.f;
        dL_dalpha += (1.f - accum_op_rec) * dpo;
        {
            real v = w * dpo;
            ADD(5, v);
        }
        dL_dalpha *= test_T;
        T = test_T;
        last_alpha = alpha;
        real bg_dot = 0;
        for (int i = 0; i < NUM_CHANNELS; i++) bg_dot += c->background[i] * dpx[i];
        dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
        const real dL_dG = co[3] * dL_dalpha;
        const real gdx = G * dx, gdy = G * dy;
        const real dG_ddelx = -gdx * co[0] - gdy * co[1];
        const real dG_ddely = -gdy * co[2] - gdx * co[1];
        real v0 = dL_dG * dG_ddelx * c->ddelx_dx, v1 = dL_dG * dG_ddely * c->ddely_dy;
        real v2 = -0.5f * gdx * dx * dL_dG, v3_ = -0.5f * gdx * dy * dL_dG, v4_ = -0.5f * gdy * dy * dL_dG;
        real v5 = G * dL_dalpha;
        ADD(0, v0); ADD(1, v1); ADD(2, v2); ADD(3, v3_); ADD(4, v4_); ADD(5, v5);
    }
#undef ADD
    return pixel_disagrees;
}

/* Tie bounds of the gradients (file header): all `real`, each either NULL or sized like the gradient of the same name. */
typedef struct {
    real *means2D /* [P,3] */, *opacities /* [P] */, *colors /* [P,3] */, *semantics /* [P,K] */, *means3D /* [P,3] */,
        *cov3D /* [P,6] */, *shs /* [P,M,3] */, *scales /* [P,3] */, *rotations /* [P,4] */;
    long tie_pixels, tie_decisions, overflow_pixels;   /* out: what was evaluated */
} HsroBounds;

static void bound_add(real* arr, size_t i, real v)
{
    if (!arr) return;
    v = R_FABS(v);
    if (v == 0) return;
#pragma omp atomic
    arr[i] += v;
}

/* |J_id * delta| into the bound arrays: delta = a per-splat difference of the NA accumulated sums */
static void bounds_add_delta(const HsroChain* ch, HsroBounds* b, uint32_t id, const double* delta, int K, int M, int inf)
{
    const HsroState* s = ch->s;
    real in[9] = {(real)delta[0], (real)delta[1], (real)delta[2], (real)delta[3], (real)delta[4], (real)delta[9],
                  (real)delta[6], (real)delta[7], (real)delta[8]};
    real dm[3], dc[6], dsx[3], dr[4];
    real* dsh = (ch->shs && M > 0) ? (real*)calloc((size_t)M * 3, sizeof(real)) : 0;
    gauss_chain(ch, (int)id, in, dm, dc, ch->scales ? dsx : 0, ch->scales ? dr : 0, dsh);
    const real big = inf ? (real)INFINITY : 0;
    bound_add(b->means2D, 3 * (size_t)id, inf ? big : (real)delta[0]); bound_add(b->means2D, 3 * (size_t)id + 1, inf ? big : (real)delta[1]);
    bound_add(b->opacities, id, inf ? big : (real)delta[5]);
    if (!s->has_sh) for (int k = 0; k < 3; k++) bound_add(b->colors, 3 * (size_t)id + k, inf ? big : (real)delta[6 + k]);
    for (int k = 0; k < K; k++) bound_add(b->semantics, (size_t)id * K + k, inf ? big : (real)delta[10 + k]);
    for (int k = 0; k < 3; k++) bound_add(b->means3D, 3 * (size_t)id + k, inf ? big : dm[k]);
    for (int k = 0; k < 6; k++) bound_add(b->cov3D, 6 * (size_t)id + k, inf ? big : dc[k]);
    if (ch->scales) {
        for (int k = 0; k < 3; k++) bound_add(b->scales, 3 * (size_t)id + k, inf ? big : dsx[k]);
        for (int k = 0; k < 4; k++) bound_add(b->rotations, 4 * (size_t)id + k, inf ? big : dr[k]);
    }
    if (dsh) { for (int k = 0; k < 3 * M; k++) bound_add(b->shs, (size_t)id * M * 3 + k, inf ? big : dsh[k]); free(dsh); }
}

/*
 * Backward.  Follows Rasterizer::backward (rasterizer_impl.cu:349-454) / backward_semantic
 * (:614-731) according to how the state was produced.  Gradient outputs must be caller-allocated
 * and are fully overwritten (the reference zero-fills them first, rasterize_points.cu:378-388); they are `real`.
 * dL_dconic is [P,4] (only [0],[1],[3] written, backward.cu:658-660).  dL_dmean2D is [P,3].
 * sem_alpha_mode 0 = reference-as-observed: the semantic->alpha term reads a scratch buffer the
 * reference never writes (backward.cu:834, rasterizer_impl.cu:673-674), i.e. zeros.  1 = "exact": the term as the
 * commented-out staging (backward.cu:778-779) would have made it — each semantic channel contributes (c - accum_rec) * dL_dchannel
 * to dL_dalpha like a colour channel; pinned by tests/dense_ref.py(sem_alpha_exact=True) (tests/test_oracle.py).
 * bounds (may be NULL): the tie bounds of the gradients, arrays zero-filled by the caller.
 */
int hsro_backward(const HsroState* s, int D, int M, const float* background, const float* means3D, const float* shs,
                  const float* colors_precomp, const float* semantics, const float* scales, float scale_modifier,
                  const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                  const float* campos, float tan_fovx, float tan_fovy, const float* dL_dpix, const float* dL_dpix_sem,
                  const float* dL_dpix_depth, const float* dL_dpix_median, const float* dL_dpix_opacity,
                  real* dL_dmean2D, real* dL_dconic, real* dL_dopacity, real* dL_dcolor, real* dL_dsemantics,
                  real* dL_ddepth, real* dL_dmean3D, real* dL_dcov3D, real* dL_dsh, real* dL_dscale, real* dL_drot,
                  int sem_alpha_mode, HsroBounds* bounds)
{
    if (sem_alpha_mode != 0 && sem_alpha_mode != 1) return -1;
    if (sem_alpha_mode == 1 && s->semantic && s->K > 0 && !semantics) return -1;
    const int P = s->P, W = s->W, H = s->H, K = s->semantic ? s->K : 0;
    const size_t N = (size_t)W * H;
    const uint32_t gx = (uint32_t)s->tiles_x, gy = (uint32_t)s->tiles_y;
    const size_t Tn = (size_t)gx * gy;
    const float focal_y = H / (2.0f * tan_fovy), focal_x = W / (2.0f * tan_fovx);
    const float* colors = colors_precomp ? colors_precomp : s->rgb;
    const int NA = 10 + K; /* per-Gaussian accumulators: mean2D xy, conic xyw, opacity, colour rgb, depth, sem[K] */
    size_t Pa = P ? (size_t)P : 1;
    double* acc = (double*)calloc(Pa * (size_t)NA, sizeof(double));
    if (!acc) return -2;
    HsroBwdCtx bc;
    bc.s = s; bc.colors = colors; bc.background = background; bc.dL_dpix = dL_dpix; bc.dL_dpix_sem = dL_dpix_sem;
    bc.dL_dpix_depth = dL_dpix_depth; bc.dL_dpix_median = dL_dpix_median; bc.dL_dpix_opacity = dL_dpix_opacity;
    bc.N = N; bc.K = K; bc.NA = NA;
    bc.semantics = semantics; bc.sem_alpha_mode = (K > 0) ? sem_alpha_mode : 0;
    bc.ddelx_dx = (float)(0.5 * W); bc.ddely_dy = (float)(0.5 * H); /* backward.cu:550-551, :759-760 */
    bc.median_rule = g_median_rule;
    long median_disagree = 0;
    const HsroDecision no_ovr = {0, 0};

    /* fp32 accumulation model (hsro_set_accumulation): float sums, tiles in a seeded random order, one thread */
    const int fp32_acc = g_accum_fp32;
    float* acc32 = fp32_acc ? (float*)calloc(Pa * (size_t)NA, sizeof(float)) : 0;
    long* order = (long*)malloc(sizeof(long) * (Tn ? Tn : 1));
    for (size_t i = 0; i < Tn; i++) order[i] = (long)i;
    if (fp32_acc) {
        uint64_t st = 0x9E3779B97F4A7C15ull ^ ((uint64_t)g_accum_seed * 0xD1B54A32D192ED03ull + 1ull);
        for (size_t i = Tn; i > 1; i--) {
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            const size_t j = (size_t)(st % i);
            const long t_ = order[i - 1]; order[i - 1] = order[j]; order[j] = t_;
        }
    }
    /* ---- renderCUDA (backward.cu:472-666) / renderCUDA_SEM (backward.cu:669-899) ---- */
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : median_disagree) if (!fp32_acc)
    for (long ti = 0; ti < (long)Tn; ti++) {
        const long tile = order[ti];
        const uint32_t ty = (uint32_t)(tile / gx), tx = (uint32_t)(tile % gx);
        const uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
        real* dsem = (real*)malloc(sizeof(real) * (size_t)(K > 0 ? 3 * K : 1));
        /* fp32 model: the pixels of a tile in a seeded random order too — the reference's 256 threads of a block add with atomicAdd in
         * whatever order the hardware serialises them (backward.cu:616-663), not in raster order, which would add neighbouring (similar)
         * terms one after the other and understate the spread */
        int pix_order[BLOCK_X * BLOCK_Y];
        for (int i = 0; i < BLOCK_X * BLOCK_Y; i++) pix_order[i] = i;
        if (fp32_acc) {
            uint64_t st = 0xA0761D6478BD642Full ^ ((uint64_t)g_accum_seed * 0xE7037ED1A0B428DBull + (uint64_t)tile * 0x8EBC6AF09C88C6E3ull + 1ull);
            for (int i = BLOCK_X * BLOCK_Y; i > 1; i--) {
                st ^= st << 13; st ^= st >> 7; st ^= st << 17;
                const int j = (int)(st % (uint64_t)i);
                const int t_ = pix_order[i - 1]; pix_order[i - 1] = pix_order[j]; pix_order[j] = t_;
            }
        }
        for (int pi = 0; pi < BLOCK_X * BLOCK_Y; pi++) {
                const int tyy = pix_order[pi] / BLOCK_X, txx = pix_order[pi] % BLOCK_X;
                uint32_t px = tx * BLOCK_X + txx, py = ty * BLOCK_Y + tyy;
                if (!(px < (uint32_t)W && py < (uint32_t)H)) continue;
                size_t pix_id = (size_t)W * py + px;
                median_disagree += pixel_backward(&bc, pix_id, (float)px, (float)py, r0, r1, s->final_T[pix_id], (int)s->n_contrib[pix_id],
                                                  s->median_pos[pix_id], no_ovr, dsem, acc, 0, acc32);
            }
        free(dsem);
    }
    free(order);
    g_median_disagree = median_disagree;
    if (acc32) { for (size_t i = 0; i < Pa * (size_t)NA; i++) acc[i] = (double)acc32[i]; free(acc32); }
    for (int i = 0; i < P; i++) {
        const double* a = acc + (size_t)i * NA;
        dL_dmean2D[3 * i] = (real)a[0]; dL_dmean2D[3 * i + 1] = (real)a[1]; dL_dmean2D[3 * i + 2] = 0.f;
        dL_dconic[4 * i] = (real)a[2]; dL_dconic[4 * i + 1] = (real)a[3]; dL_dconic[4 * i + 2] = 0.f; dL_dconic[4 * i + 3] = (real)a[4];
        dL_dopacity[i] = (real)a[5];
        for (int c = 0; c < 3; c++) dL_dcolor[3 * i + c] = (real)a[6 + c];
        dL_ddepth[i] = (real)a[9];
        for (int c = 0; c < K; c++) dL_dsemantics[(size_t)i * K + c] = (real)a[10 + c];
    }
    free(acc);

    memset(dL_dmean3D, 0, sizeof(real) * 3 * (size_t)P);
    memset(dL_dcov3D, 0, sizeof(real) * 6 * (size_t)P);
    if (dL_dscale) memset(dL_dscale, 0, sizeof(real) * 3 * (size_t)P);
    if (dL_drot) memset(dL_drot, 0, sizeof(real) * 4 * (size_t)P);
    if (dL_dsh && M > 0) memset(dL_dsh, 0, sizeof(real) * 3 * (size_t)M * (size_t)P);

    HsroChain ch;
    ch.s = s; ch.D = D; ch.M = M; ch.means3D = means3D; ch.shs = shs; ch.scales = scales; ch.rotations = rotations;
    ch.cov3Ds = cov3D_precomp ? cov3D_precomp : s->cov3D; ch.viewmatrix = viewmatrix; ch.projmatrix = projmatrix; ch.campos = campos;
    ch.scale_modifier = scale_modifier; ch.focal_x = focal_x; ch.focal_y = focal_y; ch.tan_fovx = tan_fovx; ch.tan_fovy = tan_fovy;

    /* ---- computeCov2DCUDA (backward.cu:144-274) + preprocessCUDA (backward.cu:346-412), per Gaussian ---- */
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (!(s->radii[idx] > 0)) continue;
        const real in[9] = {dL_dmean2D[3 * idx], dL_dmean2D[3 * idx + 1], dL_dconic[4 * idx], dL_dconic[4 * idx + 1], dL_dconic[4 * idx + 3],
                            dL_ddepth[idx], dL_dcolor[3 * idx], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2]};
        gauss_chain(&ch, idx, in, dL_dmean3D + 3 * (size_t)idx, dL_dcov3D + 6 * (size_t)idx, scales ? dL_dscale + 3 * (size_t)idx : 0,
                    scales ? dL_drot + 4 * (size_t)idx : 0, (shs && dL_dsh) ? dL_dsh + (size_t)idx * M * 3 : 0);
    }

    /* ---- tie bounds of the gradients: every flagged decision of every flagged pixel taken the other way ---- */
    if (bounds) {
        long n_pix = 0, n_dec = 0, n_ovf = 0;
        const float* feat = colors;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : n_pix, n_dec, n_ovf)
        for (long pix = 0; pix < (long)N; pix++) {
            if (!s->tie_pixels[pix]) continue;
            const uint32_t px = (uint32_t)(pix % W), py = (uint32_t)(pix / W);
            const size_t tile = (size_t)(py / BLOCK_Y) * gx + px / BLOCK_X;
            const uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
            const size_t nl = (size_t)(r1 - r0);
            real* Sacc = (real*)malloc(sizeof(real) * (size_t)(K > 0 ? 3 * K : 1));
            real* dsem = (real*)malloc(sizeof(real) * (size_t)(K > 0 ? 3 * K : 1));
            double* base = (double*)calloc((nl ? nl : 1) * (size_t)NA, sizeof(double));
            double* alt = (double*)malloc((nl ? nl : 1) * (size_t)NA * sizeof(double));
            HsroPixFwd f;
            pixel_forward(s, feat, semantics, K, r0, r1, (float)px, (float)py, no_ovr, 1, Sacc, &f);
            pixel_backward(&bc, (size_t)pix, (float)px, (float)py, r0, r1, f.T, (int)f.last_contributor, f.median_at, no_ovr, dsem, 0, base, 0);
            n_pix++; n_dec += f.ndec; n_ovf += f.overflow;
            for (int d = 0; d < f.ndec; d++) {
                if (f.dec[d].kind == 4) {
                    /* T passed within ulps of 0.5 here: dL_dmedian_depth may go to this splat or not */
                    double* dl = (double*)calloc((size_t)NA, sizeof(double));
                    dl[9] = (double)dL_dpix_median[pix];
                    bounds_add_delta(&ch, bounds, s->vals[f.dec[d].pos], dl, K, M, 0);
                    free(dl);
                    continue;
                }
                HsroPixFwd g;
                pixel_forward(s, feat, semantics, K, r0, r1, (float)px, (float)py, f.dec[d], 0, Sacc, &g);
                memset(alt, 0, (nl ? nl : 1) * (size_t)NA * sizeof(double));
                pixel_backward(&bc, (size_t)pix, (float)px, (float)py, r0, r1, g.T, (int)g.last_contributor, g.median_at, f.dec[d], dsem, 0, alt, 0);
                for (size_t j = 0; j < nl; j++) {
                    double* da = alt + j * NA; const double* ba = base + j * NA;
                    int any = 0;
                    for (int k = 0; k < NA; k++) { da[k] -= ba[k]; any |= (da[k] != 0.0); }
                    if (any) bounds_add_delta(&ch, bounds, s->vals[r0 + j], da, K, M, 0);
                }
            }
            if (f.overflow) {   /* more flagged decisions than tracked: no bound for the splats of this pixel */
                double* dl = (double*)calloc((size_t)NA, sizeof(double));
                for (uint32_t i = r0; i < f.i_end; i++) bounds_add_delta(&ch, bounds, s->vals[i], dl, K, M, 1);
                free(dl);
            }
            free(Sacc); free(dsem); free(base); free(alt);
        }
        bounds->tie_pixels = n_pix; bounds->tie_decisions = n_dec; bounds->overflow_pixels = n_ovf;
    }
    return 0;
}

void hsro_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int hsro_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
