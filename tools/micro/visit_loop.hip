// Micro-benchmark (round 4): what bounds the backward's visit loop?  The real loop (hsr_render_bwd_q.hip) costs ~650-750 cycles per visit
// and wave at 3 or 4 waves per SIMD whatever is taken out of it (profiles/r04_e_ablate_geo.log).  This reproduces its instruction stream on
// synthetic LDS contents, with pieces switched off by template flags, at 1..4 workgroups of 4 waves per CU (LDS padding sets the occupancy),
// and reports shader cycles per visit and wave.
// Build: hipcc --offload-arch=gfx950 -O3 visit_loop.hip -o visit_loop ; prints one JSON line per (variant, occupancy).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

constexpr int NREC = 224;

// flags: 1 LDS record reads, 2 panel stores, 4 exp / rcp, 8 the three compares + masks (else unconditional), 16 median branch,
//        32 list element read, 64 software pipelined (two stages) vs straight
template <int F>
__global__ void __launch_bounds__(256, 4) k(float* out, unsigned long long* cyc, int iters, int pad_words)
{
    extern __shared__ float4 s_dyn[];
    __shared__ float4 s_ent[3 * (NREC + 1)];
    __shared__ float s_pan[4][2 * 42 * 17];
    __shared__ uint2 s_ord[4][4][20];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, gq = lane >> 4, l16 = lane & 15;
    for (int i = t; i < 3 * (NREC + 1); i += 256) s_ent[i] = make_float4(0.3f + 0.001f * i, 0.2f + 0.002f * i, -0.05f, 0.01f);
    if (pad_words > 0 && t == 0) s_dyn[0] = make_float4(0, 0, 0, 0);
    for (int i = lane; i < 20; i += 64) s_ord[wv][gq][i % 20] = make_uint2(0, 0);
    if (l16 < 20) s_ord[wv][gq][l16] = make_uint2((uint32_t)((l16 * 13 + gq * 7) % NREC) * 48u, (uint32_t)((l16 * 3 + gq) % 40) * 68u);
    if (l16 < 4) s_ord[wv][gq][16 + l16] = make_uint2((uint32_t)NREC * 48u, 40u * 68u);
    __syncthreads();
    float pfx = (float)(lane & 7), pfy = (float)(lane >> 3);
    asm volatile("" : "+v"(pfx), "+v"(pfy));
    float dpx0 = 0.1f + 0.01f * lane, dpx1 = 0.2f, dpx2 = 0.3f, dpd = 0.4f, dpo = 0.5f, dpm = 0.25f, ntfbg = 0.f;
    asm volatile("" : "+v"(dpx0), "+v"(dpx1), "+v"(dpx2), "+v"(dpd), "+v"(dpo), "+v"(dpm), "+v"(ntfbg));
    float T = 0.5f, Racc = 0.f;
    int jf_off = 0, jm_off = 48 * 5 + (lane == 3 ? 0 : 100000);
    asm volatile("" : "+v"(jf_off), "+v"(jm_off));
    char* const panb = reinterpret_cast<char*>(&s_pan[0][0]);
    const char* const entb = reinterpret_cast<const char*>(&s_ent[0]);
    const uint32_t lane16_off = (uint32_t)(wv * 2 * 42 * 17 * 4 + l16 * 4);
    uint2* const ordp = &s_ord[wv][gq][0];

    struct StageA { float am, Gm, inv, nb, h; uint32_t ro; bool med; };
    auto stage_a = [&](const float4& ra, const float4& rb, const float2& rc, uint2 e) -> StageA {
        StageA o;
        const int eo = (int)e.x;
        o.ro = lane16_off + e.y;
        const float dx = ra.x - pfx, dy = ra.y - pfy;
        const float dxx = dx * dx, dxy = dx * dy, dyy = dy * dy;
        const float power2 = fmaf(rc.x, dyy, fmaf(ra.w, dxy, ra.z * dxx));
        const float G = (F & 4) ? __builtin_amdgcn_exp2f(power2) : power2 + 1.0f;
        const float alpha = fminf(0.99f, rc.y * G);
        const bool active = (F & 8) ? (eo >= jf_off && power2 <= 0.0f && alpha >= 1.0f / 255.0f) : true;
        o.am = active ? alpha : 0.f;
        o.Gm = active ? G : 0.f;
        o.inv = (F & 4) ? __builtin_amdgcn_rcpf(1.0f - o.am) : 1.0f + o.am;
        o.nb = ntfbg * o.inv;
        o.h = fmaf(rb.x, dpx0, fmaf(rb.y, dpx1, fmaf(rb.z, dpx2, fmaf(rb.w, dpd, dpo))));
        o.med = (F & 16) ? (active && eo == jm_off) : false;
        return o;
    };
    auto stage_b = [&](const StageA& c) {
        const float test_T = T * c.inv;
        const float w = c.am * test_T;
        const float d = c.h - Racc;
        const float dL_dalpha = fmaf(d, test_T, c.nb);
        const float gda = c.Gm * dL_dalpha;
        if (F & 2) {
            *reinterpret_cast<float*>(panb + c.ro) = w;
            *reinterpret_cast<float*>(panb + c.ro + 42 * 17 * 4) = gda;
        } else {
            asm volatile("" ::"v"(w), "v"(gda));
        }
        if ((F & 16) && c.med) {
            asm volatile("");
            atomicAdd(reinterpret_cast<float*>(panb + (c.ro - (uint32_t)(l16 * 4)) + 42 * 17 * 4 + 64), dpm);
        }
        Racc = fmaf(c.am, d, Racc);
        T = test_T * 0.999f + 0.0005f;
    };
    auto load_rec = [&](uint2 e, float4& ra, float4& rb, float2& rc) {
        const float4* ent = reinterpret_cast<const float4*>(entb + e.x);
        ra = ent[0];
        rb = ent[1];
        rc = *reinterpret_cast<const float2*>(&ent[2]);
    };
    unsigned long long total = 0;
    for (int rep = 0; rep < iters; rep++) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (F & 64) {
            uint2 e1 = ordp[1], e2 = ordp[2];
            float4 ra1, rb1;
            float2 rc1;
            StageA cur;
            {
                const uint2 e0 = ordp[0];
                float4 ra0, rb0;
                float2 rc0;
                load_rec(e0, ra0, rb0, rc0);
                load_rec(e1, ra1, rb1, rc1);
                cur = stage_a(ra0, rb0, rc0, e0);
            }
#pragma nounroll
            for (int it = 0; it < 16; it += 2) {
                const uint2 e3 = (F & 32) ? ordp[(it + 3) & 15] : e1;
                float4 ra2, rb2;
                float2 rc2;
                if (F & 1) load_rec(e2, ra2, rb2, rc2);
                else { ra2 = ra1; rb2 = rb1; rc2 = rc1; }
                const StageA nxt = stage_a(ra1, rb1, rc1, e1);
                stage_b(cur);
                const uint2 e4 = (F & 32) ? ordp[(it + 4) & 15] : e2;
                if (F & 1) load_rec(e3, ra1, rb1, rc1);
                cur = stage_a(ra2, rb2, rc2, e2);
                stage_b(nxt);
                e1 = e3; e2 = e4;
            }
        } else {
            uint2 e = ordp[0];
            float4 ra, rb;
            float2 rc;
            load_rec(e, ra, rb, rc);
#pragma nounroll
            for (int it = 0; it < 16; it++) {
                if (F & 32) e = ordp[(it + 1) & 15];
                if (F & 1) load_rec(e, ra, rb, rc);
                const StageA c = stage_a(ra, rb, rc, e);
                stage_b(c);
            }
        }
        total += __builtin_amdgcn_s_memtime() - t0;
    }
    if (lane == 0) cyc[blockIdx.x * 4 + wv] = total;
    out[blockIdx.x * 256 + t] = T + Racc;
}

template <int F>
void run(const char* name, int wg_per_cu)
{
    const int blocks = 256 * wg_per_cu, iters = 200;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 4);
    // static LDS of the kernel: 10.8 KB + 22.8 KB + 2.5 KB = 36.2 KB; pad so that exactly wg_per_cu workgroups fit 160 KB
    const int per_wg = 160 * 1024 / wg_per_cu;
    int dyn = per_wg - 37 * 1024;
    if (dyn < 0) dyn = 0;
    hipFuncSetAttribute((const void*)k<F>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<F><<<blocks, 256, dyn>>>(out, cyc, 5, dyn / 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<F><<<blocks, 256, dyn>>>(out, cyc, iters, dyn / 4);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2] / (iters * 16.0);
    printf("{\"variant\": \"%s\", \"flags\": %d, \"wg_per_cu\": %d, \"cycles_per_visit_per_wave\": %.1f, \"simd_cycles_per_wave_visit\": %.1f, \"kernel_ms\": %.4f, \"err\": \"%s\"}\n",
           name, F, wg_per_cu, med, med / wg_per_cu, ms, hipGetErrorString(hipGetLastError()));
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    for (int w = 1; w <= 4; w++) {
        run<127>("full, pipelined", w);
        run<63>("full, straight", w);
        run<127 - 16>("no median branch", w);
        run<127 - 8>("no compares / masks", w);
        run<127 - 4>("no exp / rcp", w);
        run<127 - 2>("no panel stores", w);
        run<127 - 1>("no record reads", w);
        run<127 - 32>("no list reads", w);
        run<64>("arithmetic only, pipelined", w);
        run<0>("arithmetic only, straight", w);
        run<64 + 8>("arithmetic + compares", w);
        run<64 + 4>("arithmetic + exp / rcp", w);
    }
    return 0;
}
