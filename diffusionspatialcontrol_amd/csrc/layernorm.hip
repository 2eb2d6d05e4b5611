// Residual add + LayerNorm (dsc_add_layernorm): one wave per token row, 16-byte accesses, row kept in registers
// between the statistics and the normalisation (one read of x / a, one write of the sum, one write of y).
// HBM-bound: algorithmic bytes = rows * C * 2 * (2 reads + 2 writes).
#include "dsc_common.h"
#include "dsc_hip.h"

namespace {

constexpr int kMaxV = 8;         // 16-byte vectors per lane: C <= 64 * 8 * 8 = 4096

template <int NV>
__global__ __launch_bounds__(256) void add_ln_kernel(const half_t* x, const half_t* a, const half_t* gamma,
                                                     const half_t* beta, half_t* sum_out, half_t* y, long long rows,
                                                     int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int cv = C >> 3;
    const half_t* xr = x + row * C;
    float v[NV][8];
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c8 = lane + 64 * i;
        if (c8 < cv) {
            h8_t xv = *reinterpret_cast<const h8_t*>(xr + c8 * 8);
            if (a) {
                const h8_t av = *reinterpret_cast<const h8_t*>(a + row * C + c8 * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[j] = (half_t)((float)xv[j] + (float)av[j]);      // the sum is an fp16 tensor
                if (sum_out) *reinterpret_cast<h8_t*>(sum_out + row * C + c8 * 8) = xv;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[i][j] = (float)xv[j]; s1 += v[i][j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
    }
    const float mean = wave_sum_f32(s1) / (float)C;
    float s2 = 0.f;                                             // two-pass variance on the register copy: no cancellation
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c8 = lane + 64 * i;
        if (c8 < cv) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float dlt = v[i][j] - mean; s2 += dlt * dlt; }
        }
    }
    const float rstd = rsqrtf(wave_sum_f32(s2) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c8 = lane + 64 * i;
        if (c8 < cv) {
            const h8_t g = *reinterpret_cast<const h8_t*>(gamma + c8 * 8);
            const h8_t bt = *reinterpret_cast<const h8_t*>(beta + c8 * 8);
            h8_t o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (half_t)((v[i][j] - mean) * rstd * (float)g[j] + (float)bt[j]);
            *reinterpret_cast<h8_t*>(y + row * C + c8 * 8) = o;
        }
    }
}

bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

extern "C" int dsc_add_layernorm(const void* x, const void* a, const void* gamma, const void* beta, void* sum_out, void* y,
                                 int64_t rows, int C, float eps, int dtype, void* stream) {
    if (!x || !gamma || !beta || !y || rows <= 0 || C <= 0) return DSC_ERR_BAD_ARG;
    if (dtype != DSC_F16 || C % 8 != 0 || C > 64 * 8 * kMaxV) return DSC_ERR_UNSUPPORTED;
    if (!al16(x) || !al16(y) || !al16(gamma) || !al16(beta) || (a && !al16(a)) || (sum_out && !al16(sum_out)))
        return DSC_ERR_UNSUPPORTED;
    const int nv = (C / 8 + 63) / 64;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const half_t* xp = static_cast<const half_t*>(x); const half_t* ap = static_cast<const half_t*>(a);
    const half_t* gp = static_cast<const half_t*>(gamma); const half_t* bp = static_cast<const half_t*>(beta);
    half_t* sp = static_cast<half_t*>(sum_out); half_t* yp = static_cast<half_t*>(y);
    switch (nv) {
        case 1: DSC_LAUNCH(add_ln_kernel<1>, grid, block, 0, st, xp, ap, gp, bp, sp, yp, (long long)rows, C, eps); break;
        case 2: DSC_LAUNCH(add_ln_kernel<2>, grid, block, 0, st, xp, ap, gp, bp, sp, yp, (long long)rows, C, eps); break;
        case 3: DSC_LAUNCH(add_ln_kernel<3>, grid, block, 0, st, xp, ap, gp, bp, sp, yp, (long long)rows, C, eps); break;
        case 4: DSC_LAUNCH(add_ln_kernel<4>, grid, block, 0, st, xp, ap, gp, bp, sp, yp, (long long)rows, C, eps); break;
        default: DSC_LAUNCH(add_ln_kernel<kMaxV>, grid, block, 0, st, xp, ap, gp, bp, sp, yp, (long long)rows, C, eps); break;
    }
    return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
}
