// Few-row linear layers of the time-embedding path (dsc_linear_rows_f16): y[m, n] = act(sum_k x[m, k] w[n, k] + bias[n])
// for M <= 8 rows - `Timesteps` + `TimestepEmbedding` (reference u_net_condition_modify.py:554-560, diffusers
// embeddings [recalled]) and the 22 `ResnetBlock2D.time_emb_proj(silu(temb))` projections that the UNet module runs as
// one concatenated GEMM.  These are weight-streaming GEMVs (M = 2 with CFG): one wave per output feature, lanes split
// K in 16-byte chunks (coalesced weight rows), fp32 accumulation, wave reduction.  Flags fuse what surrounded them
// as ~10 separate elementwise launches per step:
//   DSC_ROWS_SINUSOID_IN  x is fp32 t[M]; the input row is the sinusoidal embedding [cos(t f_k), sin(t f_k)] (K = 2 * half,
//                         f_k = exp(-ln(10000) k / half), flip_sin_to_cos, freq_shift 0), rounded to fp16 like the
//                         `.to(sample.dtype)` of the torch path
//   DSC_ROWS_SILU_OUT     SiLU on the fp16-rounded result (F.silu of an fp16 tensor), one more fp16 rounding
#include "dsc_common.h"
#include "dsc_hip.h"

namespace {

constexpr int kMaxRows = 8;

template <int M>
__global__ __launch_bounds__(256) void linear_rows_kernel(const void* xin, const half_t* w, const half_t* bias, half_t* out,
                                                          int N, int K, long long ldx, long long ldo, int flags) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int nchunk = K >> 3;
    float acc[M];
#pragma unroll
    for (int m = 0; m < M; ++m) acc[m] = 0.f;
    const half_t* wr = w + (long long)n * K;
    const int half_k = K >> 1;
    for (int c = lane; c < nchunk; c += 64) {
        const h8_t wv = *reinterpret_cast<const h8_t*>(wr + c * 8);
#pragma unroll
        for (int m = 0; m < M; ++m) {
            h8_t xv;
            if (flags & DSC_ROWS_SINUSOID_IN) {
                const float t = static_cast<const float*>(xin)[m];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = c * 8 + j;
                    const int kk = k < half_k ? k : k - half_k;
                    const float f = expf(-9.210340371976184f * (float)kk / (float)half_k);
                    const float a = t * f;
                    xv[j] = (half_t)(k < half_k ? cosf(a) : sinf(a));
                }
            } else {
                xv = *reinterpret_cast<const h8_t*>(static_cast<const half_t*>(xin) + m * ldx + c * 8);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[m] += (float)xv[j] * (float)wv[j];
        }
    }
#pragma unroll
    for (int m = 0; m < M; ++m) acc[m] = wave_sum_f32(acc[m]);
    if (lane == 0) {
        const float b = bias ? (float)bias[n] : 0.f;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float y = round_f16(acc[m] + b);
            if (flags & DSC_ROWS_SILU_OUT) y = silu_f(y);
            out[m * ldo + n] = (half_t)y;
        }
    }
}

}  // namespace

extern "C" int dsc_linear_rows_f16(const void* x, const void* w, const void* bias, void* out, int M, int N, int K,
                                   int64_t ldx, int64_t ldo, int flags, int dtype, void* stream) {
    if (!x || !w || !out || M <= 0 || N <= 0 || K <= 0) return DSC_ERR_BAD_ARG;
    if (dtype != DSC_F16 || M > kMaxRows || K % 8 != 0) return DSC_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(w) & 15) || (!(flags & DSC_ROWS_SINUSOID_IN) && ((reinterpret_cast<uintptr_t>(x) & 15) || ldx % 8 != 0)))
        return DSC_ERR_UNSUPPORTED;
    if ((flags & DSC_ROWS_SINUSOID_IN) && K % 16 != 0) return DSC_ERR_UNSUPPORTED;
    const dim3 grid((N + 3) / 4), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const half_t* wp = static_cast<const half_t*>(w); const half_t* bp = static_cast<const half_t*>(bias);
    half_t* op = static_cast<half_t*>(out);
#define DSC_ROWS(M_) DSC_LAUNCH(linear_rows_kernel<M_>, grid, block, 0, st, x, wp, bp, op, N, K, (long long)ldx, (long long)ldo, flags)
    switch (M) {
        case 1: DSC_ROWS(1); break;
        case 2: DSC_ROWS(2); break;
        case 3: DSC_ROWS(3); break;
        case 4: DSC_ROWS(4); break;
        case 5: DSC_ROWS(5); break;
        case 6: DSC_ROWS(6); break;
        case 7: DSC_ROWS(7); break;
        default: DSC_ROWS(8); break;
    }
#undef DSC_ROWS
    return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
}
