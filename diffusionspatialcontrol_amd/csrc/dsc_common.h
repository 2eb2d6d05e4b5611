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

// One 32-channel block of an attention output tile to global memory.  After D^T = V^T P^T on v_mfma_f32_32x32x16_f16 lane (r, hh)
// holds, of query row r, channels 32 dm + 8 g + 4 hh + {0..3} for g = 0..3 in o[4 g .. 4 g + 3]: 8-byte pieces at a row stride of
// hundreds of bytes.  `wide` (the row pointer and strides 16-byte aligned): one v_permlane32_swap per register hands lane hh = 0
// its partner's half of the EVEN 8-channel groups and lane hh = 1 its partner's half of the ODD ones, so every lane stores whole
// groups as 16-byte pieces - half the store instructions (the store tail of these kernels is issue-bound; measured: region
// cross-attention forward -3 % at d = 40, -6 % at d = 80, -9 % at d = 160; flash self-attention -1.3 % / -3.3 %).  The exchange
// runs on ALL lanes (call it from wave-uniform code); only the store is predicated by `ok`.  Same bytes either way.
__device__ __forceinline__ void store_o_block(half_t* row, const f16x_t& o, float scale, int dm, int hh, int d, bool ok, bool wide) {
    if (wide) {
        typedef unsigned u2_t __attribute__((ext_vector_type(2)));
        typedef unsigned u4_t __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            const h4_t ev = {(half_t)(o[8 * gp] * scale), (half_t)(o[8 * gp + 1] * scale), (half_t)(o[8 * gp + 2] * scale), (half_t)(o[8 * gp + 3] * scale)};
            const h4_t od = {(half_t)(o[8 * gp + 4] * scale), (half_t)(o[8 * gp + 5] * scale), (half_t)(o[8 * gp + 6] * scale), (half_t)(o[8 * gp + 7] * scale)};
            const u2_t e2 = __builtin_bit_cast(u2_t, ev), o2 = __builtin_bit_cast(u2_t, od);
            const auto s0 = __builtin_amdgcn_permlane32_swap(e2[0], o2[0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(e2[1], o2[1], false, false);
            const u4_t w = {s0[0], s1[0], s0[1], s1[1]};                  // lower 4 channels of the group, then the upper 4
            const int dd0 = 32 * dm + 16 * gp + 8 * hh;
            if (ok && dd0 < d) *reinterpret_cast<u4_t*>(row + dd0) = w;
        }
    } else {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int dd0 = 32 * dm + 8 * g4 + 4 * hh;
            if (ok && dd0 < d) {
                const h4_t ov = {(half_t)(o[4 * g4] * scale), (half_t)(o[4 * g4 + 1] * scale), (half_t)(o[4 * g4 + 2] * scale), (half_t)(o[4 * g4 + 3] * scale)};
                *reinterpret_cast<h4_t*>(row + dd0) = ov;
            }
        }
    }
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
