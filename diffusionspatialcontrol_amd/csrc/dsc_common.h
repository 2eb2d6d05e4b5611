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

#define DSC_WAVE 64

// v_mfma_f32_32x32x16_f16: D[32x32] += A[32x16] * B[16x32].  Lane l (r = l & 31, hh = l >> 5) holds
// A[row r][k = 8hh + j] and B[k = 8hh + j][col r] in element j; D element i of lane l is
// D[row (i & 3) + 8 (i >> 2) + 4 hh][col r]   (cdna_hip_programming.md section 3).
__device__ __forceinline__ f16x_t mfma_32x32x16(h8_t a, h8_t b, f16x_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float round_f16(float x) { return (float)(half_t)x; }

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
