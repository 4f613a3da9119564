// Micro-benchmark: what does it cost to run T short workgroups (a tile kernel: 256 threads, ~39 KB LDS, 128 VGPRs -> 4 per CU)
// through the hardware dispatcher, against persistent workgroups that take tiles from an atomic counter?
// Every "tile" spins for a fixed number of shader cycles (s_memtime), so the ideal time is T / slots * spin.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void spin(long long cycles, float* lds, int t)
{
    const long long t0 = __builtin_amdgcn_s_memtime();
    float a = lds[t];
    for (int guard = 0; guard < 200000 && (long long)__builtin_amdgcn_s_memtime() - t0 < cycles; guard++) {   // bounded: can never hang
#pragma unroll
        for (int i = 0; i < 32; i++) a = a * 1.0001f + 0.5f;
    }
    lds[t] = a;
}

template <bool PERSIST>
__global__ void __launch_bounds__(256, 4) k(int tiles, long long cycles, int jitter, unsigned* counter, float* out)
{
    __shared__ float lds[9800];   // 39.2 KB: four workgroups per CU
    __shared__ int s_tile;
    const int t = threadIdx.x;
    lds[t] = (float)t;
    if (!PERSIST) {
        const int tile = blockIdx.x;
        if (tile >= tiles) return;
        spin(cycles + (long long)((tile * 2654435761u) >> 16) % (jitter + 1), lds, t);
        __syncthreads();
        if (t == 0) out[tile] = lds[0];
        return;
    }
    for (int guard = 0; guard < 64; guard++) {   // bounded: at most 64 tiles per workgroup
        if (t == 0) s_tile = (int)atomicAdd(counter, 1u);
        __syncthreads();
        const int tile = s_tile;
        if (tile >= tiles) return;
        spin(cycles + (long long)((tile * 2654435761u) >> 16) % (jitter + 1), lds, t);
        __syncthreads();
        if (t == 0) out[tile] = lds[0];
    }
}

int main()
{
    const int tiles = 3225;
    unsigned* counter; float* out;
    hipMalloc(&counter, 4); hipMalloc(&out, tiles * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (long long cycles : {20000ll, 40000ll, 80000ll}) {
        for (int jitter : {0, 20000}) {
            for (int mode = 0; mode < 3; mode++) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; rep++) {
                    hipMemset(counter, 0, 4);
                    hipDeviceSynchronize();
                    hipEventRecord(a);
                    if (mode == 0) k<false><<<tiles, 256>>>(tiles, cycles, jitter, counter, out);
                    else k<true><<<mode == 1 ? 1024 : 2048, 256>>>(tiles, cycles, jitter, counter, out);
                    hipEventRecord(b); hipEventSynchronize(b);
                    float ms; hipEventElapsedTime(&ms, a, b);
                    if (ms < best) best = ms;
                }
                printf("{\"spin_cycles\": %lld, \"jitter\": %d, \"mode\": \"%s\", \"tiles\": %d, \"ms\": %.4f, \"ns_per_tile\": %.2f}\n", cycles, jitter,
                       mode == 0 ? "dispatcher" : (mode == 1 ? "persistent_1024" : "persistent_2048"), tiles, best, 1e6 * best / tiles);
                fflush(stdout);
            }
        }
    }
    return 0;
}
