// Row softmax over fp16 score rows (dsc_softmax_rows_f16): p[r, :] = softmax(scale * s[r, :]) in fp32 arithmetic, one fp16
// rounding.  The middle step of the VAE's single wide attention head (512 channels over h*w tokens: reference
// `source/modules/model_k_diffusion.py:291-299` reaches it through `vae.decode`), whose two contractions run on the GEMM
// kernel (gemm.hip): scores = q.k^T, out = p.v.  The flash self-attention kernel holds a head of at most 160 channels in
// registers; a 512-wide head goes GEMM -> this -> GEMM instead (scores materialised once in fp16: 32 MiB at 64x64 tokens).
//
// One workgroup of 256 threads per row; the row (<= 256 * 8 * kMaxV halves) stays in registers between the maximum, the
// sum and the normalisation: one 16-byte read and one 16-byte write per element group.  HBM / L2-bound:
// algorithmic bytes = rows * n * 4.
#include "dsc_common.h"
#include "dsc_hip.h"

namespace {

constexpr int kMaxV = 8;          // 16-byte vectors per thread: n <= 256 * 8 * 8 = 16384

template <int NV>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const half_t* s, half_t* p, int n, long long lds_, long long ldp,
                                                           float scale_log2e) {
    __shared__ float red[8];
    const long long row = blockIdx.x;
    const half_t* sr = s + row * lds_;
    const int nv = n >> 3, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float v[NV][8];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c8 = threadIdx.x + 256 * i;
        if (c8 < nv) {
            const h8_t x = *reinterpret_cast<const h8_t*>(sr + (long long)c8 * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[i][j] = (float)x[j] * scale_log2e; mx = fmaxf(mx, v[i][j]); }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = -INFINITY;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[i][j] = __builtin_amdgcn_exp2f(v[i][j] - mx); sum += v[i][j]; }   // exp2(-inf) = 0
    sum = wave_sum_f32(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);          // fixed order: bit-reproducible
    half_t* pr = p + row * ldp;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c8 = threadIdx.x + 256 * i;
        if (c8 < nv) {
            h8_t o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (half_t)(v[i][j] * inv);
            *reinterpret_cast<h8_t*>(pr + (long long)c8 * 8) = o;
        }
    }
}

}  // namespace

extern "C" int dsc_softmax_rows_f16(const void* scores, void* probs, int64_t rows, int n, int64_t ld_scores, int64_t ld_probs,
                                    float scale, int dtype, void* stream) {
    if (!scores || !probs || rows <= 0 || n <= 0) return DSC_ERR_BAD_ARG;
    if (dtype != DSC_F16 || n % 8 != 0 || n > 256 * 8 * kMaxV || ld_scores % 8 != 0 || ld_probs % 8 != 0 || ld_scores < n ||
        ld_probs < n || rows > 0x7fffffffll || (reinterpret_cast<uintptr_t>(scores) & 15) || (reinterpret_cast<uintptr_t>(probs) & 15))
        return DSC_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float sl = scale * 1.4426950408889634f;
    const dim3 grid((unsigned)rows), block(256);
    const int nv = (n / 8 + 255) / 256;
    const half_t* s = static_cast<const half_t*>(scores);
    half_t* p = static_cast<half_t*>(probs);
    if (nv <= 1) DSC_LAUNCH(softmax_rows_kernel<1>, grid, block, 0, st, s, p, n, (long long)ld_scores, (long long)ld_probs, sl);
    else if (nv <= 2) DSC_LAUNCH(softmax_rows_kernel<2>, grid, block, 0, st, s, p, n, (long long)ld_scores, (long long)ld_probs, sl);
    else if (nv <= 4) DSC_LAUNCH(softmax_rows_kernel<4>, grid, block, 0, st, s, p, n, (long long)ld_scores, (long long)ld_probs, sl);
    else DSC_LAUNCH(softmax_rows_kernel<8>, grid, block, 0, st, s, p, n, (long long)ld_scores, (long long)ld_probs, sl);
    return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
}
