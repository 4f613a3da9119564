/*
 * hsr_oracle_la.h — the small linear algebra of the oracle, written once over a scalar type.
 *
 * TEST INFRASTRUCTURE ONLY (see hsr_oracle.c).  Included twice by hsr_oracle.c: with RT = float, LA(x) = x##_f for the
 * forward's per-Gaussian preprocess — everything that decides an INTEGER output stays fp32 in the reference's operation
 * order in EVERY build — and with RT = real, LA(x) = x##_r for the per-Gaussian backward chain, where `real` is float in
 * libhsr_oracle.so (the oracle proper) and double in libhsr_oracle_f64.so (the "truth" build that adjudicates between two
 * fp32 evaluations, tests/test_oracle.py / tests/test_gpu_truth.py).
 *
 * Each function cites the reference lines it follows (paths relative to
 * hierslam-diff-gaussian-rasterization-w-depth/cuda_rasterizer/).
 */
typedef struct { RT x, y, z; } LA(v3);
typedef struct { RT x, y, z, w; } LA(v4);
/* GLM-style column-major 3x3: m.c[col][row] (glm/detail/type_mat3x3.inl) */
typedef struct { RT c[3][3]; } LA(m3);

static inline RT LA(fmin)(RT a, RT b) { return a < b ? a : b; }
static inline RT LA(fmax)(RT a, RT b) { return a > b ? a : b; }

/* glm mat3*mat3, element sums left to right (third_party/glm/glm/detail/type_mat3x3.inl:486-520) */
static LA(m3) LA(m3_mul)(const LA(m3)* a, const LA(m3)* b)
{
    LA(m3) r;
    for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++)
            r.c[c][rr] = a->c[0][rr] * b->c[c][0] + a->c[1][rr] * b->c[c][1] + a->c[2][rr] * b->c[c][2];
    return r;
}
static LA(m3) LA(m3_transpose)(const LA(m3)* a)
{
    LA(m3) r;
    for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++) r.c[c][rr] = a->c[rr][c];
    return r;
}
/* glm::mat3(a0..a8): column-major fill */
static LA(m3) LA(m3_make)(RT a0, RT a1, RT a2, RT a3, RT a4, RT a5, RT a6, RT a7, RT a8)
{
    LA(m3) r;
    r.c[0][0] = a0; r.c[0][1] = a1; r.c[0][2] = a2;
    r.c[1][0] = a3; r.c[1][1] = a4; r.c[1][2] = a5;
    r.c[2][0] = a6; r.c[2][1] = a7; r.c[2][2] = a8;
    return r;
}

/* auxiliary.h:58-66 */
static LA(v3) LA(transformPoint4x3)(LA(v3) p, const float* m)
{
    LA(v3) t = {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
                m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
    return t;
}
/* auxiliary.h:68-77 */
static LA(v4) LA(transformPoint4x4)(LA(v3) p, const float* m)
{
    LA(v4) t = {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
                m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14],
                m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]};
    return t;
}
/* auxiliary.h:89-97 */
static LA(v3) LA(transformVec4x3Transpose)(LA(v3) p, const float* m)
{
    LA(v3) t = {m[0] * p.x + m[1] * p.y + m[2] * p.z,
                m[4] * p.x + m[5] * p.y + m[6] * p.z,
                m[8] * p.x + m[9] * p.y + m[10] * p.z};
    return t;
}

/* auxiliary.h:107-118 */
static LA(v3) LA(dnormvdv3)(LA(v3) v, LA(v3) dv)
{
    RT sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    RT invsum32 = (RT)1.0f / LA(sqrt)(sum2 * sum2 * sum2);
    LA(v3) r;
    r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return r;
}

/* shared by forward.cu:74-113 and backward.cu:166-199: T = W*J and cov2D (before the +0.3) */
static void LA(cov2d_core)(LA(v3) mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy, const float* cov3D,
                           const float* view, LA(v3)* t_out, RT* txtz_o, RT* tytz_o, LA(m3)* T_out, LA(m3)* Vrk_out, LA(m3)* cov_out)
{
    LA(v3) t = LA(transformPoint4x3)(mean, view);
    const RT limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    const RT txtz = t.x / t.z, tytz = t.y / t.z;
    t.x = LA(fmin)(limx, LA(fmax)(-limx, txtz)) * t.z;
    t.y = LA(fmin)(limy, LA(fmax)(-limy, tytz)) * t.z;
    LA(m3) J = LA(m3_make)(focal_x / t.z, 0.0f, -(focal_x * t.x) / (t.z * t.z),
                           0.0f, focal_y / t.z, -(focal_y * t.y) / (t.z * t.z), 0, 0, 0);
    LA(m3) W = LA(m3_make)(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    LA(m3) T = LA(m3_mul)(&W, &J);
    LA(m3) Vrk = LA(m3_make)(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    LA(m3) Tt = LA(m3_transpose)(&T), Vt = LA(m3_transpose)(&Vrk);
    LA(m3) A = LA(m3_mul)(&Tt, &Vt);
    *cov_out = LA(m3_mul)(&A, &T);
    *t_out = t; *txtz_o = txtz; *tytz_o = tytz; *T_out = T; *Vrk_out = Vrk;
}
