// Micro-benchmark (round 4): issue cost of FMA forms by where the multiplier comes from.  Each wave runs REPS x 28 multiply-adds per lane with
// FOUR independent accumulator chains (no dependency stalls), the 28 multipliers either wave-uniform in SGPRs (as the leaf head's weight row after
// scalar loads) or per lane in VGPRs; scalar v_fmac_f32 or packed v_pk_fma_f32.  No loads inside the loop.  4 workgroups of 256 per CU.
// Prints SIMD cycles per instruction (kernel time x clock / instructions issued per SIMD).   hipcc --offload-arch=gfx950 -O3 fma_operand.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int KU = 28, REPS = 8192;

template <int MODE>
__global__ void __launch_bounds__(256, 4) k(const float* __restrict__ wt, const float* __restrict__ x, float* out)
{
    float sv[KU], wv[KU];
#pragma unroll
    for (int i = 0; i < KU; i++) sv[i] = x[(threadIdx.x * KU + i) & 4095];
#pragma unroll
    for (int i = 0; i < KU; i++) wv[i] = wt[i];                 // wave-uniform address: scalar loads, values live in SGPRs
    if (MODE >= 2) {
#pragma unroll
        for (int i = 0; i < KU; i++) asm volatile("v_mov_b32 %0, %0" : "+v"(wv[i]));   // MODE >= 2: the multipliers in vector registers
    }
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    f32x2 p0 = {0.f, 0.f}, p1 = {0.f, 0.f}, p2 = {0.f, 0.f}, p3 = {0.f, 0.f};
    for (int r = 0; r < REPS; r++) {
        if ((MODE & 1) == 0) {
#pragma unroll
            for (int i = 0; i < KU; i += 4) {
                a0 = fmaf(wv[i], sv[i], a0); a1 = fmaf(wv[i + 1], sv[i + 1], a1);
                a2 = fmaf(wv[i + 2], sv[i + 2], a2); a3 = fmaf(wv[i + 3], sv[i + 3], a3);
            }
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else {
#pragma unroll
            for (int i = 0; i < KU; i += 8) {
                p0 = __builtin_elementwise_fma(f32x2{wv[i], wv[i + 1]}, f32x2{sv[i], sv[i + 1]}, p0);
                p1 = __builtin_elementwise_fma(f32x2{wv[i + 2], wv[i + 3]}, f32x2{sv[i + 2], sv[i + 3]}, p1);
                if (i + 4 < KU) {
                    p2 = __builtin_elementwise_fma(f32x2{wv[i + 4], wv[i + 5]}, f32x2{sv[i + 4], sv[i + 5]}, p2);
                    p3 = __builtin_elementwise_fma(f32x2{wv[i + 6], wv[i + 7]}, f32x2{sv[i + 6], sv[i + 7]}, p3);
                }
            }
            asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (a0 + a1) + (a2 + a3) + (p0.x + p0.y) + (p1.x + p1.y) + (p2.x + p2.y) + (p3.x + p3.y);
}

template <int MODE>
void run(const char* name, const float* wt, const float* x, float* out, int instr_per_rep)
{
    const int blocks = 1024;   // 4 per CU: 4 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(wt, x, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(wt, x, out);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("{\"form\": \"%s\", \"ms\": %.4f, \"simd_cycles_per_instruction\": %.2f, \"macs_per_simd_cycle\": %.1f, \"err\": \"%s\"}\n", name, ms,
           cyc / (4.0 * REPS * instr_per_rep), 4.0 * REPS * 28 * 64 / cyc, hipGetErrorString(hipGetLastError()));
}

int main()
{
    float *wt, *x, *out;
    hipMalloc(&wt, 4096 * 4); hipMalloc(&x, 4096 * 4); hipMalloc(&out, 1024 * 256 * 4);
    hipMemset(wt, 0, 4096 * 4); hipMemset(x, 0, 4096 * 4);
    run<0>("v_fmac_f32, multiplier in an SGPR", wt, x, out, 28);
    run<1>("v_pk_fma_f32, multipliers in an SGPR pair", wt, x, out, 14);
    run<2>("v_fmac_f32, multiplier in a VGPR", wt, x, out, 28);
    run<3>("v_pk_fma_f32, multipliers in a VGPR pair", wt, x, out, 14);
    return 0;
}
