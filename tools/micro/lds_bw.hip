// Micro-benchmark: LDS cost of wave-uniform (broadcast) reads vs per-lane reads on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 lds_bw.hip -o lds_bw ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters)
{
    __shared__ float4 s[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) s[i] = make_float4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float4 acc = make_float4(0, 0, 0, 0);
    int base = 0;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            int idx;
            if (MODE == 0) idx = (base + u) & 2047;                       // uniform b128
            else if (MODE == 1) idx = (base + u * 64 + lane) & 2047;      // per-lane distinct b128, consecutive
            else if (MODE == 2) idx = (base + u + (lane >> 4) * 37) & 2047;  // 4 addresses per wave (16-lane groups)
            else idx = (base + u) & 2047;
            if (MODE == 3) { float v = reinterpret_cast<float*>(s)[idx * 4]; acc.x += v; }                    // uniform b32
            else if (MODE == 4) { float2 v = reinterpret_cast<float2*>(s)[idx * 2]; acc.x += v.x; acc.y += v.y; }  // uniform b64
            else { float4 v = s[idx]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        }
        base += 8;
    }
    long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* name, int waves_per_cu_blocks)
{
    float* out; long long* cyc;
    const int nb = 256 * waves_per_cu_blocks;
    hipMalloc(&out, nb * 256 * 4); hipMalloc(&cyc, nb * 8);
    const int iters = 4096;
    k<MODE><<<nb, 256>>>(out, cyc, 64);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    k<MODE><<<nb, 256>>>(out, cyc, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    // reads per CU = blocks/CU * 4 waves * iters * 8
    double reads_per_cu = (double)waves_per_cu_blocks * 4 * iters * 8;
    printf("%-28s blocks/CU=%d  %.3f ms  ns per wave-read per CU = %.3f  (clock64 delta block0 %lld)\n", name, waves_per_cu_blocks, ms,
           ms * 1e6 / reads_per_cu, h[0]);
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int b : {1, 2, 4}) {
        run<0>("uniform b128", b);
        run<1>("per-lane b128", b);
        run<2>("4-address b128", b);
        run<3>("uniform b32", b);
        run<4>("uniform b64", b);
    }
    return 0;
}
