// Developer microbenchmark (GPU box): 4096 wavefronts, each reading per step a 128-byte, a 256-byte and a 768-byte piece
// (what a strand1_kernel phase reads: depth row, mask row, changed bytes) -- from THREE arrays (each wavefront a sequential
// stream in each), or as ONE 1152-byte record of a single interleaved stream.  Every byte read once, 2.3 GB in all.
// build: hipcc --offload-arch=gfx950 -O3 -o variants/stream_layouts scripts/micro/stream_layouts.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <bool ONE, int AHEAD>
__global__ __launch_bounds__(1024) void rd(const unsigned char* __restrict__ a128, const unsigned char* __restrict__ a256,
                                           const unsigned char* __restrict__ a768, int steps, uint32_t* out) {
    const int lane = threadIdx.x & 63;
    const size_t w = (size_t)blockIdx.x * 16 + (threadIdx.x >> 6);
    uint32_t acc = 0;
    // per step and lane: 2 B + 4 B + 12 B (three 4-byte loads)
    const unsigned char *p0, *p1, *p2;
    size_t s0, s1, s2;
    if (ONE) {   // record = [128][256][768]
        p0 = a768 + w * (size_t)steps * 1152 + lane * 2;
        p1 = a768 + w * (size_t)steps * 1152 + 128 + lane * 4;
        p2 = a768 + w * (size_t)steps * 1152 + 384 + lane * 12;
        s0 = s1 = s2 = 1152;
    } else {
        p0 = a128 + w * (size_t)steps * 128 + lane * 2, s0 = 128;
        p1 = a256 + w * (size_t)steps * 256 + lane * 4, s1 = 256;
        p2 = a768 + w * (size_t)steps * 768 + lane * 12, s2 = 768;
    }
    for (int i = 0; i + AHEAD <= steps; i += AHEAD) {
        uint32_t v[AHEAD][5];
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) {
            v[k][0] = *reinterpret_cast<const uint16_t*>(p0 + (size_t)(i + k) * s0);
            v[k][1] = *reinterpret_cast<const uint32_t*>(p1 + (size_t)(i + k) * s1);
            v[k][2] = *reinterpret_cast<const uint32_t*>(p2 + (size_t)(i + k) * s2);
            v[k][3] = *reinterpret_cast<const uint32_t*>(p2 + (size_t)(i + k) * s2 + 4);
            v[k][4] = *reinterpret_cast<const uint32_t*>(p2 + (size_t)(i + k) * s2 + 8);
        }
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) acc += v[k][0] + v[k][1] + v[k][2] + v[k][3] + v[k][4];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <bool ONE, int AHEAD>
void run(const unsigned char* a, const unsigned char* b, const unsigned char* c, int steps, uint32_t* d_out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((rd<ONE, AHEAD>), dim3(256), dim3(1024), 0, 0, a, b, c, steps, d_out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%s, %d steps in flight per wavefront: %.2f TB/s\n", ONE ? "one interleaved stream of 1152-byte records" : "three arrays (128 / 256 / 768 bytes per step)",
           AHEAD, 4096.0 * steps * 1152 / (best * 1e-3) / 1e12);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int steps = 480;   // 4096 wavefronts x 480 steps x 1152 B = 2.26 GB
    unsigned char *a, *b, *c; uint32_t* d_out;
    const size_t n = (size_t)4096 * steps;
    if (hipMalloc(&a, n * 128 + 4096) != hipSuccess || hipMalloc(&b, n * 256 + 4096) != hipSuccess || hipMalloc(&c, n * 1152 + 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&d_out, 64);
    hipMemset(a, 1, n * 128); hipMemset(b, 1, n * 256); hipMemset(c, 1, n * 1152);
    run<false, 1>(a, b, c, steps, d_out); run<false, 2>(a, b, c, steps, d_out); run<false, 4>(a, b, c, steps, d_out);
    run<true, 1>(a, b, c, steps, d_out);  run<true, 2>(a, b, c, steps, d_out);  run<true, 4>(a, b, c, steps, d_out);
    return 0;
}
