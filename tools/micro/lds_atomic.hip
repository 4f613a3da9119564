// Micro-benchmark (round 4): what does a float atomic add into LDS (ds_add_f32, no return) cost next to a plain ds_write_b32 / ds_read_b32?
// One workgroup of 256 threads per CU x 1..3 per CU; each wave issues ITERS x 8 operations back to back; prints CU-cycles per wave-instruction.
// Patterns: 0 distinct addresses, consecutive words (no conflict); 1 four lanes per address (the four 16-lane groups add to the same 16 words);
//           2 rows of 16 words at a 48-word stride (4 rows per instruction: the packed-row cache of hsr_render_bwd_q.hip's MRG experiment)
// Build: hipcc --offload-arch=gfx950 -O3 lds_atomic.hip -o lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int OP, int PAT>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int iters)
{
    __shared__ float s[4][4096];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    for (int i = t; i < 4 * 4096; i += 256) (&s[0][0])[i] = 0.f;
    __syncthreads();
    int addr;
    if (PAT == 0) addr = lane;
    else if (PAT == 1) addr = lane & 15;
    else addr = ((lane >> 4) * 5 + 3) * 48 + (lane & 15);
    float* p = &s[wv][addr];
    float acc = 0.f;
    const float v = 1.0f + lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float* q = p + j * 192;
            if (OP == 0) atomicAdd(q, v);
            else if (OP == 1) *reinterpret_cast<volatile float*>(q) = v;
            else acc += *reinterpret_cast<volatile float*>(q);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 4 + wv] = t1 - t0;
    out[blockIdx.x * 256 + t] = acc + s[wv][addr];
}

template <int OP, int PAT>
void run(const char* name, int wg_per_cu)
{
    const int blocks = 256 * wg_per_cu, iters = 2000;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 4);
    k<OP, PAT><<<blocks, 256>>>(out, cyc, 10);
    hipDeviceSynchronize();
    k<OP, PAT><<<blocks, 256>>>(out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    // s_memtime counts at 100 MHz on gfx950: report in shader cycles at 2.4 GHz as well as raw
    const double raw = (double)h[h.size() / 2] / (iters * 8.0);
    printf("{\"op\": \"%s\", \"pattern\": %d, \"wg_per_cu\": %d, \"memtime_ticks_per_wave_instr\": %.3f, \"per_cu_ticks_per_wave_instr\": %.4f, \"err\": \"%s\"}\n", name, PAT, wg_per_cu, raw,
           raw / (4.0 * wg_per_cu), hipGetErrorString(hipGetLastError()));
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int w = 1; w <= 3; w++) {
        run<0, 0>("ds_add_f32", w); run<0, 1>("ds_add_f32", w); run<0, 2>("ds_add_f32", w);
        run<1, 0>("ds_write_b32", w); run<1, 1>("ds_write_b32", w); run<1, 2>("ds_write_b32", w);
        run<2, 0>("ds_read_b32", w); run<2, 2>("ds_read_b32", w);
    }
    return 0;
}
