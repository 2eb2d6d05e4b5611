// micro-benchmark: v_exp_f32 throughput of one wave, alone and beside a partner wave on the same SIMD that issues MFMAs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f16x_t __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
// MODE 0: 32 independent v_exp_f32 per iteration; 1: 32 v_exp + 16 v_cvt_pk + 16 v_max3-like; 2: 64 v_fma only
// PARTNER 0: workgroup of 4 waves (one per SIMD); 1: 8 waves, waves 4-7 issue MFMAs in a loop; 2: 8 waves, waves 4-7 run the same VALU loop
template <int MODE, int PARTNER>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int iters) {
    const int wave = threadIdx.x >> 6;
    float x[32];
    for (int i = 0; i < 32; ++i) x[i] = -1.f - 0.01f * (threadIdx.x & 63) - i;
    h8_t a, b; for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * threadIdx.x + j); b[j] = (_Float16)(j * 0.5f); }
    f16x_t o0 = {}, o1 = {};
    float acc = 0.f;
    h8_t hsink = {}, hsink2 = {};
    __builtin_amdgcn_s_barrier();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if (PARTNER == 1 && wave >= 4) {
        for (int it = 0; it < iters; ++it) {
            o0 = MFMA(a, b, o0); o1 = MFMA(a, b, o1); o0 = MFMA(a, b, o0); o1 = MFMA(a, b, o1);
            o0 = MFMA(a, b, o0); o1 = MFMA(a, b, o1); o0 = MFMA(a, b, o0); o1 = MFMA(a, b, o1);
            o0 = MFMA(a, b, o0); o1 = MFMA(a, b, o1); o0 = MFMA(a, b, o0); o1 = MFMA(a, b, o1); o0 = MFMA(a, b, o0); o1 = MFMA(a, b, o1);
        }
    } else if (wave < 4 || PARTNER == 2) {
        for (int it = 0; it < iters; ++it) {
            float e[32];
            if (MODE == 3 || MODE == 4) {       // the fused stream: 8 x [MFMA, 4 v_exp, 2 v_cvt_pk]  (MODE 4: 2 v_exp, 1 cvt per gap)
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    if (g & 1) o1 = MFMA(a, b, o1); else o0 = MFMA(a, b, o0);
                    __builtin_amdgcn_sched_barrier(0);
                    const int n = MODE == 3 ? 4 : 2;
#pragma unroll
                    for (int i = 0; i < n; i += 2) {
                        const float e0 = __builtin_amdgcn_exp2f(x[4 * g + i]), e1 = __builtin_amdgcn_exp2f(x[4 * g + i + 1]);
                        h2 pk = __builtin_convertvector(f2{e0, e1}, h2);
                        hsink[(g + i / 2) & 7] = pk[0]; hsink2[(g + i / 2) & 7] = pk[1];     // (not an MFMA operand: no WAR on a / b)
                        x[4 * g + i] -= 1e-6f; x[4 * g + i + 1] -= 1e-6f;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 32; ++i) { e[i] = __builtin_fmaf(x[i], 1.0001f, acc); }
#pragma unroll
                for (int i = 0; i < 32; ++i) { x[i] = __builtin_fmaf(e[i], 0.999f, -0.5f); }
            } else {
#pragma unroll
                for (int i = 0; i < 32; ++i) e[i] = __builtin_amdgcn_exp2f(x[i]);
                if (MODE == 1) {
#pragma unroll
                    for (int i = 0; i < 32; i += 2) {
                        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                        typedef float f2 __attribute__((ext_vector_type(2)));
                        h2 pk = __builtin_convertvector(f2{e[i], e[i + 1]}, h2);
                        float m; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(e[i]), "v"(e[i + 1]), "v"(acc));
                        acc = m + (float)pk[0];
                    }
                }
#pragma unroll
                for (int i = 0; i < 32; ++i) x[i] = x[i] - e[i] * 1e-6f;      // keeps the exps live and loop-carried (32 more VALU)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    float s = acc;
    for (int j = 0; j < 8; ++j) s += (float)hsink[j] + (float)hsink2[j];
    for (int i = 0; i < 32; ++i) s += x[i];
    for (int i = 0; i < 16; ++i) s += o0[i] + o1[i];
    sink[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x == 0 || threadIdx.x == 256) && blockIdx.x == 0) out[(MODE * 3 + PARTNER) * 2 + (threadIdx.x ? 1 : 0)] = t1 - t0;
}
template <int MODE, int PARTNER> void run(unsigned long long* d, float* sink) {
    const int threads = PARTNER ? 512 : 256;
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<MODE, PARTNER>), dim3(256), dim3(threads), 0, 0, d, sink, 64);
}
int main() {
    unsigned long long* d; float* sink;
    (void)hipMalloc(&d, 512); (void)hipMalloc(&sink, 256 * 512 * 4); (void)hipMemset(d, 0, 512);
    run<0, 0>(d, sink); run<0, 1>(d, sink); run<0, 2>(d, sink);
    run<1, 0>(d, sink); run<1, 1>(d, sink); run<1, 2>(d, sink);
    run<2, 0>(d, sink); run<2, 1>(d, sink); run<2, 2>(d, sink);
    run<3, 0>(d, sink); run<3, 1>(d, sink); run<3, 2>(d, sink);
    run<4, 0>(d, sink); run<4, 1>(d, sink); run<4, 2>(d, sink);
    (void)hipDeviceSynchronize();
    unsigned long long h[30]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* mn[5] = {"32 v_exp + 32 v_fma", "32 v_exp + 16 cvt_pk + 16 max3 + 16 add + 32 fma", "64 v_fma", "8 x [MFMA, 4 v_exp, 2 cvt_pk, 4 v_sub]", "8 x [MFMA, 2 v_exp, 1 cvt_pk, 2 v_sub]"};
    const char* pn[3] = {"alone on its SIMD", "partner issues 14 MFMAs / iteration", "partner runs the same VALU loop"};
    for (int m = 0; m < 5; ++m)
        for (int p = 0; p < 3; ++p)
            printf("%-50s | %-36s | wave 0: %7.1f cycles / iteration   wave 4: %7.1f\n", mn[m], pn[p], h[(m * 3 + p) * 2] / 64.0, h[(m * 3 + p) * 2 + 1] / 64.0);
    return 0;
}
