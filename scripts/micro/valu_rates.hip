// Developer microbenchmark (GPU box): issue cost of the instructions strand1_kernel is made of, at its occupancy
// (1024-thread workgroups, one per CU = 4 wavefronts per SIMD), 8 independent chains per wavefront.
// build: hipcc --offload-arch=gfx950 -O3 -o variants/valu_rates scripts/micro/valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <cstdint>
#include <unistd.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
constexpr int kIters = 2048;

template <int OP>
__global__ __launch_bounds__(1024) void k(uint32_t* out, unsigned long long* cyc, uint32_t seed) {
    __shared__ uint64_t lds[8192];
    uint32_t a[8], b = seed + threadIdx.x, c = seed * 3u + 1u;
    uint64_t w[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + i * 17u + threadIdx.x, w[i] = ((uint64_t)a[i] << 32) | b;
    for (int i = threadIdx.x; i < 8192; i += 1024) lds[i] = i;
    __syncthreads();
    const uint32_t laddr = (threadIdx.x & 31) * 8u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
#define OPX(i)                                                                                                        \
    if constexpr (OP == 0) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                   \
    if constexpr (OP == 1) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));                \
    if constexpr (OP == 2) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(w[i]) : "v"(c));                            \
    if constexpr (OP == 3) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[i]));                                       \
    if constexpr (OP == 4) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                   \
    if constexpr (OP == 5) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));                     \
    if constexpr (OP == 6) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(a[i]) : "v"(b)); \
    if constexpr (OP == 7) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[i]) : "s"(c));                       \
    if constexpr (OP == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");               \
    if constexpr (OP == 9) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));                           \
    if constexpr (OP == 10) asm volatile("v_lshl_add_u32 %0, %0, 7, %1" : "+v"(a[i]) : "v"(b));                       \
    if constexpr (OP == 11) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                \
    if constexpr (OP == 12) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                               \
    if constexpr (OP == 13) asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));             \
    if constexpr (OP == 14) asm volatile("v_cmp_le_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");                   \
    if constexpr (OP == 15) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                               \
    if constexpr (OP == 16) asm volatile("v_add_f64 %0, %0, %0" : "+v"(w[i]));                                        \
    if constexpr (OP == 17) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));                                   \
    if constexpr (OP == 18) asm volatile("ds_read_b64 %0, %1 offset:" #i "*256\n s_waitcnt lgkmcnt(7)" : "=v"(w[i]) : "v"(laddr) : "memory"); \
    if constexpr (OP == 19) asm volatile("ds_read_b32 %0, %1 offset:" #i "*256\n s_waitcnt lgkmcnt(7)" : "=v"(a[i]) : "v"(laddr) : "memory"); \
    if constexpr (OP == 20) asm volatile("ds_write_b64 %1, %0 offset:" #i "*256" : : "v"(w[i]), "v"(laddr) : "memory"); \
    if constexpr (OP == 21) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(c) : "v"(a[i]));
        REP8(OPX)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t r = c;
    for (int i = 0; i < 8; ++i) r ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
    out[blockIdx.x * 1024 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(const char* name, uint32_t* d_out, unsigned long long* d_cyc) {
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, d_out, d_cyc, 12345u);
    for (int ms = 0; hipStreamQuery(0) == hipErrorNotReady; ++ms) {  // a kernel that does not end within 10 s: say so and leave
        if (ms > 10000) {
            printf("%s: kernel did not finish\n", name);
            fflush(stdout);
            _exit(3);
        }
        usleep(1000);
    }
    std::vector<unsigned long long> h(256 * 16);
    hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    // 4 wavefronts per SIMD each issue kIters * 8 instructions in `med` cycles
    printf("%-22s %7.2f cycles per wave-instruction and SIMD (wavefront alone would see %.2f)\n", name, med / (kIters * 8.0 * 4.0), med / (kIters * 8.0));
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    uint32_t* d_out;
    unsigned long long* d_cyc;
    hipMalloc(&d_out, 256 * 1024 * 4);
    hipMalloc(&d_cyc, 256 * 16 * 8);
    run<0>("v_perm_b32", d_out, d_cyc);
    run<1>("v_dot4_u32_u8", d_out, d_cyc);
    run<2>("v_lshrrev_b64", d_out, d_cyc);
    run<3>("v_bfe_u32", d_out, d_cyc);
    run<4>("v_add3_u32", d_out, d_cyc);
    run<5>("v_sad_u8", d_out, d_cyc);
    run<6>("v_add_u32_sdwa", d_out, d_cyc);
    run<7>("v_mbcnt_lo", d_out, d_cyc);
    run<8>("v_cndmask_b32", d_out, d_cyc);
    run<9>("v_bcnt_u32_b32", d_out, d_cyc);
    run<10>("v_lshl_add_u32", d_out, d_cyc);
    run<11>("v_and_or_b32", d_out, d_cyc);
    run<12>("v_add_u32", d_out, d_cyc);
    run<13>("v_alignbyte_b32", d_out, d_cyc);
    run<14>("v_cmp_le_u32", d_out, d_cyc);
    run<15>("v_min_u32", d_out, d_cyc);
    run<16>("v_add_f64", d_out, d_cyc);
    run<17>("v_mov_b32", d_out, d_cyc);
    run<18>("ds_read_b64 (own banks)", d_out, d_cyc);
    run<19>("ds_read_b32 (2 per bank)", d_out, d_cyc);
    run<20>("ds_write_b64", d_out, d_cyc);
    run<21>("v_readlane_b32", d_out, d_cyc);
    return 0;
}
