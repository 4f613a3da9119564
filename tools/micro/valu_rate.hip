// Micro-benchmark: how many SIMD cycles does one wave64 VALU instruction cost on gfx950, as a
// function of the number of waves resident on the SIMD?  DESIGN.md (round 1) assumed 4 cycles always; the MI355X guide
// says 2 with >= 2 waves resident (SIMD-32) and 4 only for a wave alone.  Also: v_pk_fma_f32, v_exp_f32, fp32 MFMA
// (16x16x4) alone, and a VALU stream interleaved with fp32 MFMAs — the mix the backward tile kernel issues.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box; prints one JSON line per row.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

// MODE 0: 16 independent v_fma_f32 per iteration     1: 8 independent v_pk_fma_f32 (16 lanes-worth of fma each... 2 fma/lane)
// MODE 2: 16 independent v_exp_f32                   3: 4 independent v_mfma_f32_16x16x4_f32
// MODE 4: 4 MFMA + 16 v_fma interleaved 1:4          5: 4 MFMA + 32 v_fma interleaved 1:8
// MODE 8: 16 independent v_fmac_f32_dpp (quad_perm)   9: 16 independent v_fmac_f32
// MODE 6: 4 independent v_mfma_f32_16x16x32_bf16     7: 4 bf16 MFMA + 32 v_fma interleaved 1:8   (the 3 x bf16 split of an fp32 product)
template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, int iters)
{
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = 1.0f + 0.001f * (threadIdx.x + i);
    v4f acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
    const float m = 0.999f, c = 0.0001f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                float2 v = make_float2(a[i], a[i + 1]);
                const float2 mm = make_float2(m, m), cc = make_float2(c, c);
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(mm), "v"(cc));
                a[i] = v.x; a[i + 1] = v.y;
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        } else if (MODE == 8) {   // the wide forward's accumulate: the feature comes from a quad lane through the FMA's DPP operand
#pragma unroll
            for (int i = 0; i < 16; i++)
                asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(m), "v"(c));
        } else if (MODE == 9) {   // the same without DPP
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], a[i + 4], acc[i], 0, 0, 0);
        } else if (MODE == 6 || MODE == 7) {
            union { float f[4]; v8bf v; } ua, ub;
#pragma unroll
            for (int i = 0; i < 4; i++) { ua.f[i] = a[12 + i]; ub.f[i] = a[12 + ((i + 1) & 3)]; }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, acc[i], 0, 0, 0);
                if (MODE == 7) {
#pragma unroll
                    for (int j = 0; j < 8; j++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(i * 8 + j) % 12]) : "v"(m), "v"(c));
                }
            }
        } else {
            constexpr int PER = MODE == 4 ? 4 : 8;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[12 + (i & 3)], m, acc[i], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < PER; j++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(i * PER + j) % 12]) : "v"(m), "v"(c));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i];
#pragma unroll
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE> void run(const char* name, int waves_per_simd, double valu_per_iter, double mfma_per_iter)
{
    // one workgroup per CU of 4 * waves_per_simd waves: the hardware deals a workgroup's waves round-robin over the 4 SIMDs
    const int threads = 256 * waves_per_simd, nb = 256, iters = 20000;
    if (threads > 1024) return;
    float* out; unsigned long long* cyc;
    const int nw = nb * threads / 64;
    hipMalloc(&out, (size_t)nb * threads * 4); hipMalloc(&cyc, (size_t)nw * 8);
    k<MODE><<<nb, threads>>>(out, cyc, 200);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    k<MODE><<<nb, threads>>>(out, cyc, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(nw);
    hipMemcpy(h.data(), cyc, (size_t)nw * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double wave_cycles = (double)h[nw / 2];      // s_memtime ticks = shader cycles
    const double instr = (valu_per_iter + mfma_per_iter) * iters;
    printf("{\"bench\": \"%s\", \"waves_per_simd\": %d, \"median_wave_cycles\": %.0f, \"cycles_per_instr_per_wave\": %.3f, "
           "\"simd_cycles_per_wave_instr\": %.3f, \"valu_per_iter\": %.0f, \"mfma_per_iter\": %.0f, \"kernel_ms\": %.4f, "
           "\"clock_GHz_est\": %.3f}\n",
           name, waves_per_simd, wave_cycles, wave_cycles / instr, wave_cycles / instr / waves_per_simd, valu_per_iter, mfma_per_iter, ms,
           wave_cycles / (ms * 1e6));
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int w : {1, 2, 3, 4}) {
        run<8>("v_fmac_f32_dpp quad_perm x16 independent", w, 16, 0);
        run<9>("v_fmac_f32 x16 independent", w, 16, 0);
        run<0>("v_fma_f32 x16 independent", w, 16, 0);
        run<1>("v_pk_fma_f32 x8 independent", w, 8, 0);
        run<2>("v_exp_f32 x16 independent", w, 16, 0);
        run<3>("v_mfma_f32_16x16x4_f32 x4 independent", w, 0, 4);
        run<4>("mfma16x16x4 + 4 v_fma each", w, 16, 4);
        run<5>("mfma16x16x4 + 8 v_fma each", w, 32, 4);
        run<6>("v_mfma_f32_16x16x32_bf16 x4 independent", w, 0, 4);
        run<7>("mfma16x16x32_bf16 + 8 v_fma each", w, 32, 4);
    }
    return 0;
}
