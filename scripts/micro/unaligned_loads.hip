// Developer microbenchmark (GPU box): cost of a wave's per-lane global loads by size, stride and misalignment, data in
// L2 / L1 (every wavefront sweeps its own 16 KB over and over): cycles per wave-instruction and CU at 16 wavefronts per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o variants/unaligned_loads scripts/micro/unaligned_loads.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

constexpr int kIters = 512;

template <int BYTES, int STRIDE, int MIS>
__global__ __launch_bounds__(1024) void k(const unsigned char* __restrict__ buf, uint32_t* out, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    const size_t w = (size_t)blockIdx.x * 16 + (threadIdx.x >> 6);
    const unsigned char* p = buf + w * 16384 + (size_t)lane * STRIDE + MIS;
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
        const unsigned char* q = p + (size_t)(it & 7) * 1024;   // eight spots of the wavefront's 16 KB
        uint32_t v[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (BYTES == 4) __builtin_memcpy(&v[i][0], q + i * 2048, 4);
            if constexpr (BYTES == 8) __builtin_memcpy(&v[i][0], q + i * 2048, 8);
            if constexpr (BYTES == 16) __builtin_memcpy(&v[i][0], q + i * 2048, 16);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < BYTES / 4; ++j) acc += v[i][j];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 0x12345678u) out[0] = acc;
    if (lane == 0) cyc[w] = t1 - t0;
}

template <int BYTES, int STRIDE, int MIS>
void run(const unsigned char* d, uint32_t* d_out, unsigned long long* d_cyc) {
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<BYTES, STRIDE, MIS>), dim3(256), dim3(1024), 0, 0, d, d_out, d_cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(4096);
    hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%2d B per lane, lane stride %2d B, offset %d: %6.1f cycles per wave-instruction and CU (16 wavefronts per CU)\n", BYTES, STRIDE, MIS,
           (double)h[2048] / (kIters * 4.0 * 16.0));
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned char* d; uint32_t* d_out; unsigned long long* d_cyc;
    hipMalloc(&d, (size_t)4096 * 16384 + 65536); hipMalloc(&d_out, 64); hipMalloc(&d_cyc, 4096 * 8);
    hipMemset(d, 1, (size_t)4096 * 16384 + 65536);
    run<4, 4, 0>(d, d_out, d_cyc);   run<8, 8, 0>(d, d_out, d_cyc);   run<16, 16, 0>(d, d_out, d_cyc);
    run<8, 12, 0>(d, d_out, d_cyc);  run<8, 12, 1>(d, d_out, d_cyc);  run<8, 11, 0>(d, d_out, d_cyc);
    run<16, 12, 0>(d, d_out, d_cyc); run<16, 12, 1>(d, d_out, d_cyc); run<16, 11, 0>(d, d_out, d_cyc);
    run<4, 12, 0>(d, d_out, d_cyc);  run<4, 12, 1>(d, d_out, d_cyc);  run<4, 3, 0>(d, d_out, d_cyc);
    run<8, 16, 0>(d, d_out, d_cyc);  run<8, 16, 4>(d, d_out, d_cyc);  run<8, 16, 1>(d, d_out, d_cyc);
    return 0;
}
