// micro-benchmark: issue pace of v_mfma_f32_32x32x16_f16 under different accumulator dependency patterns (one wave per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f16x_t __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* out, float* sink) {
    h8_t a[8], b[4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (_Float16)(threadIdx.x * 0.001f + i + j);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (_Float16)(threadIdx.x * 0.002f + i - j);
    f16x_t o0 = {}, o1 = {}, o2 = {}, o3 = {};
    __builtin_amdgcn_s_barrier();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < 64; ++it) {
        if (MODE == 0) {          // two chains, alternating (the P.V pattern)
            o0 = MFMA(a[0], b[0], o0); o1 = MFMA(a[4], b[0], o1); o0 = MFMA(a[1], b[1], o0); o1 = MFMA(a[5], b[1], o1);
            o0 = MFMA(a[2], b[2], o0); o1 = MFMA(a[6], b[2], o1); o0 = MFMA(a[3], b[3], o0); o1 = MFMA(a[7], b[3], o1);
        } else if (MODE == 1) {   // two chains, one after the other
            o0 = MFMA(a[0], b[0], o0); o0 = MFMA(a[1], b[1], o0); o0 = MFMA(a[2], b[2], o0); o0 = MFMA(a[3], b[3], o0);
            o1 = MFMA(a[4], b[0], o1); o1 = MFMA(a[5], b[1], o1); o1 = MFMA(a[6], b[2], o1); o1 = MFMA(a[7], b[3], o1);
        } else if (MODE == 2) {   // four chains round robin
            o0 = MFMA(a[0], b[0], o0); o1 = MFMA(a[4], b[0], o1); o2 = MFMA(a[1], b[1], o2); o3 = MFMA(a[5], b[1], o3);
            o0 = MFMA(a[2], b[2], o0); o1 = MFMA(a[6], b[2], o1); o2 = MFMA(a[3], b[3], o2); o3 = MFMA(a[7], b[3], o3);
        } else {                  // one chain
            o0 = MFMA(a[0], b[0], o0); o0 = MFMA(a[4], b[0], o0); o0 = MFMA(a[1], b[1], o0); o0 = MFMA(a[5], b[1], o0);
            o0 = MFMA(a[2], b[2], o0); o0 = MFMA(a[6], b[2], o0); o0 = MFMA(a[3], b[3], o0); o0 = MFMA(a[7], b[3], o0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    float s = 0;
    for (int i = 0; i < 16; ++i) s += o0[i] + o1[i] + o2[i] + o3[i];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[MODE] = t1 - t0;
}
int main() {
    unsigned long long* d; float* sink; hipMalloc(&d, 64); hipMalloc(&sink, 256 * 256 * 4); hipMemset(d, 0, 64);
    for (int r = 0; r < 3; ++r) {
        k<0><<<256, 256>>>(d, sink); k<1><<<256, 256>>>(d, sink); k<2><<<256, 256>>>(d, sink); k<3><<<256, 256>>>(d, sink);
    }
    hipDeviceSynchronize();
    unsigned long long h[4]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    const char* names[4] = {"two chains alternating", "two chains in sequence", "four chains round robin", "one chain"};
    for (int i = 0; i < 4; ++i) printf("%-26s %6.1f cycles (s_memtime, 100 MHz ticks x clock ratio not applied: raw %llu) per MFMA\n", names[i], h[i] / 512.0, h[i]);
    return 0;
}
