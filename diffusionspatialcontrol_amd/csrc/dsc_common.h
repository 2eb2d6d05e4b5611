// Shared device-side helpers for the gfx950 kernels of libdsc_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef float f16x_t __attribute__((ext_vector_type(16)));
typedef float f4x_t __attribute__((ext_vector_type(4)));
typedef float f2x_t __attribute__((ext_vector_type(2)));

#define DSC_WAVE 64

// Division of a workgroup index by a launch constant without the ~30-instruction, ~130-cycle dependent chain of a runtime integer
// division (v_rcp_iflag + fix-ups + v_readfirstlane): most kernels start with one to five of them - tile row / column from the
// workgroup id, image / group / chunk - on the critical path in front of their first load.  mg = floor(2^32 / d) + 1 gives
// floor(n / d) = umulhi(n, mg) exactly while n * d < 2^32 (checked on the host, else mg = 0 and the kernel divides).
struct FastDiv { int d; unsigned mg; };
inline FastDiv make_fastdiv(long long d, long long n_max) {
    FastDiv f;
    f.d = (int)(d > 0 ? d : 1);
    f.mg = (f.d > 1 && n_max >= 0 && (unsigned long long)n_max * (unsigned long long)f.d < (1ull << 32))
               ? (unsigned)((1ull << 32) / (unsigned)f.d + 1ull) : 0u;
    return f;
}
__device__ __forceinline__ int fdiv(int n, const FastDiv& f) {       // n >= 0
    return f.d == 1 ? n : (f.mg ? (int)__umulhi((unsigned)n, f.mg) : n / f.d);
}


// v_mfma_f32_32x32x16_f16: D[32x32] += A[32x16] * B[16x32].  Lane l (r = l & 31, hh = l >> 5) holds
// A[row r][k = 8hh + j] and B[k = 8hh + j][col r] in element j; D element i of lane l is
// D[row (i & 3) + 8 (i >> 2) + 4 hh][col r]   (cdna_hip_programming.md section 3).
__device__ __forceinline__ f16x_t mfma_32x32x16(h8_t a, h8_t b, f16x_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// Kernel launch whose status is THIS launch's: hipGetLastError() is sticky per thread, and other libraries in the process
// leave benign errors behind (hipBLASLt probing for kernel names: hipErrorNotFound), which the entry points' post-launch
// check would otherwise report as a failed launch
#define DSC_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

__device__ __forceinline__ float round_f16(float x) { return (float)(half_t)x; }

// SiLU x / (1 + e^-x) on v_exp_f32 + v_rcp_f32 (relative error ~2e-7) instead of the IEEE division sequence
// (v_div_scale / v_div_fmas / v_div_fixup: ~10 VALU instructions per element of every GroupNorm+SiLU output)
__device__ __forceinline__ float silu_f(float x) {
    return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

// GELU with the exact (erf) definition diffusers' GEGLU uses, branch-free: erfc(|z|), z = x / sqrt 2, from Abramowitz &
// Stegun 7.1.26 (|error| <= 1.5e-7, three orders below the fp16 resolution of the result) - 5 FMAs, one v_rcp_f32 and one
// v_exp_f32 instead of the device library's two-branch erff (~45 VALU instructions per element with both branches live
// in a wave: a third of the GEGLU GEMM's time).  The negative tail uses erfc directly: no 1 - erf cancellation.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
    float pl = fmaf(1.061405429f, t, -1.453152027f);
    pl = fmaf(pl, t, 1.421413741f);
    pl = fmaf(pl, t, -0.284496736f);
    pl = fmaf(pl, t, 0.254829592f);
    const float e = pl * t * __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);     // erfc(|z|)
    return 0.5f * x * (x >= 0.f ? 2.f - e : e);                                          // x/2 (1 + erf z)
}

__device__ __forceinline__ double wave_sum_f64(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}
__device__ __forceinline__ float wave_sum_f32(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}
