// Developer microbenchmark (GPU box): HBM bandwidth of 4096 wavefronts (256 workgroups x 16) each reading ITS OWN
// sequential stream, as a function of the bytes one wave-instruction brings (the strand pass reads 128 B - 768 B pieces of
// four arrays per wavefront and phase).  Buffer 2 GB (beyond the Infinity Cache), every byte read once.
// build: hipcc --offload-arch=gfx950 -O3 -o variants/stream_bursts scripts/micro/stream_bursts.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <unistd.h>

template <int BPL, int INFLIGHT>   // bytes per lane and load, loads kept in flight
__global__ __launch_bounds__(1024) void rd(const unsigned char* __restrict__ buf, size_t per_wave, uint32_t* out) {
    const int lane = threadIdx.x & 63;
    const size_t w = (size_t)blockIdx.x * 16 + (threadIdx.x >> 6);
    const unsigned char* p = buf + w * per_wave + (size_t)lane * BPL;
    uint32_t acc = 0;
    constexpr int STEP = 64 * BPL;
    for (size_t o = 0; o + (size_t)STEP * INFLIGHT <= per_wave; o += (size_t)STEP * INFLIGHT) {
        uint32_t v[INFLIGHT][BPL / 4 > 0 ? BPL / 4 : 1];
#pragma unroll
        for (int i = 0; i < INFLIGHT; ++i) {
            if constexpr (BPL == 2) v[i][0] = *reinterpret_cast<const uint16_t*>(p + o + (size_t)i * STEP);
            if constexpr (BPL == 4) v[i][0] = *reinterpret_cast<const uint32_t*>(p + o + (size_t)i * STEP);
            if constexpr (BPL == 8) { const uint2 t = *reinterpret_cast<const uint2*>(p + o + (size_t)i * STEP); v[i][0] = t.x, v[i][1] = t.y; }
            if constexpr (BPL == 16) { const uint4 t = *reinterpret_cast<const uint4*>(p + o + (size_t)i * STEP); v[i][0] = t.x, v[i][1] = t.y, v[i][2] = t.z, v[i][3] = t.w; }
        }
#pragma unroll
        for (int i = 0; i < INFLIGHT; ++i)
#pragma unroll
            for (int j = 0; j < (BPL / 4 > 0 ? BPL / 4 : 1); ++j) acc += v[i][j];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int BPL, int INFLIGHT>
void run(const unsigned char* d, size_t total, uint32_t* d_out) {
    const size_t per_wave = total / 4096;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((rd<BPL, INFLIGHT>), dim3(256), dim3(1024), 0, 0, d, per_wave, d_out);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("%5d B per wave-instruction, %d in flight per wavefront (%6d B): %.2f TB/s\n", 64 * BPL, INFLIGHT, 64 * BPL * INFLIGHT,
           (double)(per_wave / (64 * BPL * INFLIGHT) * (64 * BPL * INFLIGHT)) * 4096 / (best * 1e-3) / 1e12);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t total = (size_t)2 << 30;
    unsigned char* d; uint32_t* d_out;
    if (hipMalloc(&d, total) != hipSuccess || hipMalloc(&d_out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 1, total);
    run<2, 4>(d, total, d_out);  run<2, 16>(d, total, d_out);
    run<4, 4>(d, total, d_out);  run<4, 16>(d, total, d_out);
    run<8, 4>(d, total, d_out);  run<8, 8>(d, total, d_out);
    run<16, 1>(d, total, d_out); run<16, 2>(d, total, d_out); run<16, 4>(d, total, d_out); run<16, 8>(d, total, d_out);
    return 0;
}
