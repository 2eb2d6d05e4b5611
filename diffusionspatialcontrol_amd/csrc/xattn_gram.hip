// dsc_xattn_gram_pack: what the region cross-attention's std needs of the text keys, once per generation (round 4).
//
// The reference takes `qk.std()` over the scores a = scale * q.k^T (attention_modify.py:90,96).  Per head and query row
//     sum_s a = scale q . (sum_s k_s)            sum_s a^2 = scale^2 q^T (K^T K) q
// so the statistics can be taken where q is produced (dsc_linear_q_gram_f16: the to_q projection's epilogue) once the keys are
// reduced to their d x d Gram matrix and their sum.  The keys are step-invariant (the text), so this runs next to
// dsc_xattn_kv_pack: one workgroup per (text row, head), K_h [S, d] staged in LDS as fp32 (the fp16 values the score kernels
// multiply, exactly), G = K^T K accumulated in fp32 in key order (fixed: reproducible), then
//     gram   fp16 [Bt, H, JP, KP]   G / gscale, zero padded to MFMA tiles (JP = d up to 32, KP = d up to 16)
//     gscale fp32 [Bt, H]           max |G| / 1024: fp16 then keeps 11 bits of every entry whatever the keys' range
//     ksum   fp32 [Bt, H * d]       sum of the keys over s
#include "dsc_common.h"
#include "dsc_hip.h"

namespace {

constexpr int kGT = 256;
constexpr int kGSMax = 96, kGDMax = 160;

__global__ __launch_bounds__(kGT) void xattn_gram_kernel(const half_t* k, long long ksb, long long kss, long long ksh, int H, int S, int d,
                                                          int JP, int KP, half_t* gram, float* gscale, float* ksum) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ks = reinterpret_cast<float*>(smem);                 // [S][d]
    float* red = ks + kGSMax * kGDMax;                          // [kGT] max reduction
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const half_t* src = k + b * ksb + h * ksh;
    for (int i = threadIdx.x; i < S * d; i += kGT) {
        const int s = i / d, j = i - s * d;
        ks[i] = (float)src[s * kss + j];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < d; j += kGT) {                // key sums, in key order
        float t = 0.f;
        for (int s = 0; s < S; ++s) t += ks[s * d + j];
        ksum[(long long)b * H * d + h * d + j] = t;
    }
    // G[j'][j] for this thread's entries (d * d of them, strided over the block); two passes: the maximum, then the scaled store
    float mx = 0.f;
    for (int e = threadIdx.x; e < d * d; e += kGT) {
        const int jp = e / d, j = e - jp * d;
        float t = 0.f;
        for (int s = 0; s < S; ++s) t += ks[s * d + jp] * ks[s * d + j];
        mx = fmaxf(mx, fabsf(t));
    }
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int o = kGT / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    const float sc = fmaxf(red[0] * (1.0f / 1024.0f), 1e-30f);
    if (threadIdx.x == 0) gscale[bh] = sc;
    const float inv = 1.0f / sc;
    half_t* dst = gram + (long long)bh * JP * KP;
    for (int e = threadIdx.x; e < JP * KP; e += kGT) {
        const int jp = e / KP, j = e - jp * KP;
        float t = 0.f;
        if (jp < d && j < d)
            for (int s = 0; s < S; ++s) t += ks[s * d + jp] * ks[s * d + j];
        dst[e] = (half_t)(t * inv);
    }
}

}  // namespace

extern "C" size_t dsc_xattn_gram_bytes(int n_text_rows, int H, int d) {
    if (n_text_rows <= 0 || H <= 0 || d <= 0) return 0;
    const size_t JP = (size_t)(d + 31) / 32 * 32, KP = (size_t)(d + 15) / 16 * 16;
    return (size_t)n_text_rows * H * JP * KP * sizeof(half_t);
}

extern "C" int dsc_xattn_gram_pack(const void* k, void* gram, float* gscale, float* ksum, int n_text_rows, int H, int S, int d,
                                   const int64_t k_strides[3], int dtype, void* stream) {
    if (!k || !gram || !gscale || !ksum || !k_strides || n_text_rows <= 0 || H <= 0 || S <= 0 || d <= 0) return DSC_ERR_BAD_ARG;
    if (dtype != DSC_F16 || S > kGSMax || d > kGDMax || d % 8 != 0 || k_strides[0] < 0 || k_strides[1] < 0 || k_strides[2] < 0)
        return DSC_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(gram) & 15) || (reinterpret_cast<uintptr_t>(ksum) & 15)) return DSC_ERR_UNSUPPORTED;
    const int JP = (d + 31) / 32 * 32, KP = (d + 15) / 16 * 16;
    const size_t lds = (size_t)(kGSMax * kGDMax + kGT) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_gram_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    DSC_LAUNCH(xattn_gram_kernel, dim3((unsigned)(n_text_rows * H)), dim3(kGT), lds, static_cast<hipStream_t>(stream),
               static_cast<const half_t*>(k), (long long)k_strides[0], (long long)k_strides[1], (long long)k_strides[2], H, S, d, JP, KP,
               static_cast<half_t*>(gram), gscale, ksum);
    return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
}
