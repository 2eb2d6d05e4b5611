// micro-benchmark: bytes per clock a CU moves from an L2-resident buffer into LDS, by LDS-DMA (global_load_lds_dwordx4) and by
// register staging (global_load_dwordx4 + ds_write_b128), with 4 or 8 filling waves per CU.  Every wave-instruction moves 1 KiB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kPieces = 8;            // pieces in flight per wave
template <int MODE>                   // 0: LDS-DMA, 1: registers + ds_write_b128, 2: LDS-DMA of GEMM-shaped pieces (8 rows x 128 B, row stride LD)
__global__ __launch_bounds__(256) void fill(const char* src, size_t window, int iters, unsigned long long* out, float* sink, int ld) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // each workgroup walks its own 64 KiB stretch of a window that all workgroups share (so it stays in the L2s)
    const size_t base = ((size_t)blockIdx.x * 65536) % window;
    char* my_lds = lds + wave * (kPieces * 1024);
    float acc = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const char* p = src + ((base + (size_t)(it & 7) * 8192 + wave * 16384) % window) + lane * 16;
        if (MODE == 2) {
            // a 64-row x 64-halves K tile per wave and iteration: pieces of 8 rows x 128 B out of rows `ld` bytes apart; the K
            // offset advances by 128 B per iteration (K = ld / 2 halves, wrapping), the row panel is the workgroup's
            const char* q = src + ((size_t)blockIdx.x * 64 * ld + (size_t)wave * 16 * ld) % window + (size_t)((it * 128) % ld);
#pragma unroll
            for (int k = 0; k < kPieces; ++k)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(q + (size_t)((k & 1) * 8 + (lane >> 3)) * ld + (lane & 7) * 16 + (k >> 1) * 128 % ld),
                                                 (__attribute__((address_space(3))) void*)(my_lds + k * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < kPieces; ++k)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + k * 1024),
                                                 (__attribute__((address_space(3))) void*)(my_lds + k * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            f4 v[kPieces];
#pragma unroll
            for (int k = 0; k < kPieces; ++k) v[k] = *reinterpret_cast<const f4*>(p + k * 1024);
#pragma unroll
            for (int k = 0; k < kPieces; ++k) *reinterpret_cast<f4*>(my_lds + k * 1024 + lane * 16) = v[k];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    acc += *reinterpret_cast<float*>(my_lds + lane * 4);
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, const char* src, size_t window, int wgs, unsigned long long* d, float* sink, int ld = 0) {
    const int iters = 200;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((fill<MODE>), dim3(wgs), dim3(256), 4 * kPieces * 1024, 0, src, window, iters, d, sink, ld);
    (void)hipDeviceSynchronize();
    static unsigned long long h[1024];
    (void)hipMemcpy(h, d, wgs * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < wgs; ++i) mean += (double)h[i]; mean /= wgs;
    const double bytes = (double)iters * 4 * kPieces * 1024;      // per workgroup
    printf("%-28s %4d workgroups (%d per CU): %.1f B/clk per workgroup, %.1f B/clk per CU\n", name, wgs, wgs / 256, bytes / mean, bytes / mean * (wgs / 256));
}
int main() {
    const size_t window = 2u << 20;
    char* src; unsigned long long* d; float* sink;
    (void)hipMalloc(&src, (16u << 20)); (void)hipMemset(src, 1, (16u << 20));
    (void)hipMalloc(&d, 1024 * 8); (void)hipMalloc(&sink, 1024 * 256 * 4);
    run<0>("LDS-DMA", src, window, 256, d, sink);
    run<1>("registers + ds_write_b128", src, window, 256, d, sink);
    run<0>("LDS-DMA", src, window, 512, d, sink);
    run<1>("registers + ds_write_b128", src, window, 512, d, sink);
    run<0>("LDS-DMA", src, window, 768, d, sink);
    run<1>("registers + ds_write_b128", src, window, 768, d, sink);
    for (int wg : {256, 512, 768}) {
        run<2>("LDS-DMA, rows 640 B apart", src, window, wg, d, sink, 640);
        run<2>("LDS-DMA, rows 2560 B apart", src, 8u << 20, wg, d, sink, 2560);
    }
    return 0;
}
