// Calibration of rocprofv3's FETCH_SIZE for the access patterns of the tile kernels' staging phase.
//
// MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads exactly HALF the bytes of a wide coalesced stream (16 B per lane, 128-byte requests
// tallied at 64 B) and is "uncalibrated" for other widths.  The forward / backward tile kernels do not stream: every lane GATHERS one
// splat — three 16-byte loads inside one 64-byte record (GeomState::rec) and the splat's semantic row, 4 K bytes at an arbitrary 4-byte
// alignment, as float2 loads — so "2 x FETCH_SIZE" (tools/make_traffic_json.py, round 2) is an upper bound of unknown slack.  This
// program issues exactly those patterns over tables far larger than the Infinity Cache, each row touched ONCE, so the bytes that must
// cross the fabric are known (64-byte lines touched), and is run under `rocprofv3 --pmc FETCH_SIZE`: the ratio per kernel is the
// correction for that pattern.
//   stream16   : lane i reads 16 B at base + 16 i            (the guide's case: expect FETCH_SIZE = 1/2 bytes)
//   rec_gather : lane reads rec[4 id + {0,1,2}], id random   (one 64-byte line per lane)
//   row_gather : lane reads K floats of row id as float2     (4 K bytes = ceil-ish 2.6 lines at K = 26)
// Prints the known bytes per kernel; tools/fetch_calib.sh divides the counter by them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void __launch_bounds__(256) stream16(const float4* __restrict__ src, size_t n, float* __restrict__ sink)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 1.2345e-30f) sink[0] = acc;
}

// 4 bytes per lane, consecutive lanes consecutive words: how the tile kernels read the upstream gradient planes and write their outputs
__global__ void __launch_bounds__(256) stream4(const float* __restrict__ src, size_t n, float* __restrict__ sink)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += src[i];
    if (acc == 1.2345e-30f) sink[0] = acc;
}

__global__ void __launch_bounds__(256) rec_gather(const float4* __restrict__ rec, const uint32_t* __restrict__ ids, size_t n, float* __restrict__ sink)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4* r = rec + 4 * (size_t)ids[i];
        const float4 a = r[0], b = r[1], c = r[2];
        acc += a.x + b.y + c.z;
    }
    if (acc == 1.2345e-30f) sink[0] = acc;
}

template <int K>
__global__ void __launch_bounds__(256) row_gather(const float* __restrict__ sem, const uint32_t* __restrict__ ids, size_t n, float* __restrict__ sink)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float2* row = reinterpret_cast<const float2*>(sem + (size_t)ids[i] * K);
#pragma unroll
        for (int q = 0; q < K / 2; q++) {
            const float2 v = row[q];
            acc += v.x + v.y;
        }
    }
    if (acc == 1.2345e-30f) sink[0] = acc;
}

int main()
{
    const size_t NREC = 12u << 20;                 // 12 Mi records of 64 B = 768 MiB, rows of 104 B = 1.2 GiB: far beyond the 256 MiB Infinity Cache
    constexpr int K = 26;
    float4* rec; float* sem; uint32_t* ids; float* sink;
    if (hipMalloc(&rec, NREC * 64) != hipSuccess || hipMalloc(&sem, NREC * K * 4) != hipSuccess || hipMalloc(&ids, NREC * 4) != hipSuccess ||
        hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(rec, 0, NREC * 64); hipMemset(sem, 0, NREC * K * 4);
    // a random permutation of the record numbers: every row exactly once, in random order
    std::vector<uint32_t> h(NREC);
    for (size_t i = 0; i < NREC; i++) h[i] = (uint32_t)i;
    uint64_t s = 88172645463325252ull;
    for (size_t i = NREC - 1; i > 0; i--) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const size_t j = (size_t)(s % (i + 1));
        const uint32_t t = h[i]; h[i] = h[j]; h[j] = t;
    }
    hipMemcpy(ids, h.data(), NREC * 4, hipMemcpyHostToDevice);
    // lines of 64 B touched by the rows: row r spans bytes [104 r, 104 r + 104)
    size_t row_lines = 0;
    for (size_t r = 0; r < NREC; r++) row_lines += (K * 4 * (r + 1) - 1) / 64 - (K * 4 * r) / 64 + 1;
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; rep++) {
        stream16<<<2048, 256>>>(rec, NREC * 4, sink);
        stream4<<<2048, 256>>>(reinterpret_cast<const float*>(rec), NREC * 16, sink);
        rec_gather<<<2048, 256>>>(rec, ids, NREC, sink);
        row_gather<K><<<2048, 256>>>(sem, ids, NREC, sink);
    }
    hipDeviceSynchronize();
    printf("{\"stream4_bytes\": %zu, \"stream16_bytes\": %zu, \"rec_gather_bytes\": %zu, \"rec_gather_ids_bytes\": %zu, \"row_gather_line_bytes\": %zu, \"row_gather_row_bytes\": %zu, \"row_gather_ids_bytes\": %zu, \"launches_each\": 3}\n",
           NREC * 64, NREC * 64, NREC * 64, NREC * 4, row_lines * 64, NREC * (size_t)K * 4, NREC * 4);
    return 0;
}
