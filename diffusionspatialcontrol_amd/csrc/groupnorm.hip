// GroupNorm (+SiLU) and GEGLU for NCHW fp16 activations (include/dsc_hip.h: dsc_groupnorm_silu, dsc_geglu).
//
// In NCHW the (C/groups) channels of a group are adjacent, so group g of row b is ONE contiguous span of
// n = (C/groups)*hw halves.  The op is HBM-bound (read x, write y) and the UNet has only B*32 groups per
// tensor (64 at batch 1), far fewer than the 256 CUs, so every span is cut into `nsplit` chunks:
//   launch 1  gn_stats : one workgroup per (group, chunk): 16-B loads, fp32 lane sums -> fp64 (sum, sumsq)
//   launch 2  gn_apply : same grid; each workgroup re-adds its group's nsplit partials (fixed order ->
//                        bit-reproducible), then normalises its chunk with the per-channel affine and SiLU.
// The second read of x comes from L2 / Infinity Cache (the tensors are 0.3 - 10 MB).
#include "dsc_common.h"
#include "dsc_hip.h"

namespace {

constexpr int kT = 256;

struct GnParams {
    const half_t* x; half_t* y; const half_t* gamma; const half_t* beta;
    double* partials;        // [B*groups][nsplit][2]
    int C, hw, groups, cpg, nsplit, chunk8;   // chunk8: 16-byte vectors per chunk
    long long n8;            // vectors per group
    float eps; int silu;
};

__global__ __launch_bounds__(kT) void gn_stats(GnParams p) {
    __shared__ double red[2 * (kT / 64)];
    const int bg = blockIdx.x / p.nsplit, sp = blockIdx.x % p.nsplit;
    const half_t* base = p.x + (long long)bg * p.n8 * 8;
    const long long v0 = (long long)sp * p.chunk8;
    const long long v1 = min(v0 + p.chunk8, p.n8);
    double d1 = 0.0, d2 = 0.0;
    float s1 = 0.f, s2 = 0.f;
    int cnt = 0;
    for (long long v = v0 + threadIdx.x; v < v1; v += kT) {
        const h8_t val = *reinterpret_cast<const h8_t*>(base + v * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = (float)val[j]; s1 += f; s2 += f * f; }
        if (++cnt == 16) { d1 += s1; d2 += s2; s1 = s2 = 0.f; cnt = 0; }    // fp32 only over 128 values
    }
    d1 += s1; d2 += s2;
    d1 = wave_sum_f64(d1); d2 = wave_sum_f64(d2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[2 * wave] = d1; red[2 * wave + 1] = d2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a1 = 0.0, a2 = 0.0;
        for (int w = 0; w < kT / 64; ++w) { a1 += red[2 * w]; a2 += red[2 * w + 1]; }
        double* dst = p.partials + ((long long)bg * p.nsplit + sp) * 2;
        dst[0] = a1; dst[1] = a2;
    }
}

__global__ __launch_bounds__(kT) void gn_apply(GnParams p) {
    const int bg = blockIdx.x / p.nsplit, sp = blockIdx.x % p.nsplit;
    const int g = bg % p.groups;
    const double* src = p.partials + (long long)bg * p.nsplit * 2;
    double a1 = 0.0, a2 = 0.0;
    for (int i = 0; i < p.nsplit; ++i) { a1 += src[2 * i]; a2 += src[2 * i + 1]; }   // every thread, same order
    const double n = (double)p.n8 * 8.0;
    const double mean_d = a1 / n;
    double var = a2 / n - mean_d * mean_d;                     // biased, as torch.nn.functional.group_norm
    var = var > 0.0 ? var : 0.0;
    const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + (double)p.eps));
    const half_t* base = p.x + (long long)bg * p.n8 * 8;
    half_t* out = p.y + (long long)bg * p.n8 * 8;
    const long long v0 = (long long)sp * p.chunk8;
    const long long v1 = min(v0 + p.chunk8, p.n8);
    const int hw8 = p.hw / 8;
    for (long long v = v0 + threadIdx.x; v < v1; v += kT) {
        const int c = g * p.cpg + (int)(v / hw8);              // hw % 8 == 0: a vector never straddles channels
        const float ga = (float)p.gamma[c] * rstd, be = (float)p.beta[c] - mean * ga;
        const h8_t val = *reinterpret_cast<const h8_t*>(base + v * 8);
        h8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = (float)val[j] * ga + be;
            if (p.silu) f = silu_f(f);
            o[j] = (half_t)f;
        }
        *reinterpret_cast<h8_t*>(out + v * 8) = o;
    }
}

// generic-shape fallback (hw % 8 != 0, e.g. the 2x2 / 4x4 levels of toy UNets): one element per lane-step
__global__ __launch_bounds__(kT) void gn_stats_scalar(GnParams p, long long n, long long chunk) {
    __shared__ double red[2 * (kT / 64)];
    const int bg = blockIdx.x / p.nsplit, sp = blockIdx.x % p.nsplit;
    const half_t* base = p.x + (long long)bg * n;
    const long long e0 = (long long)sp * chunk, e1 = min(e0 + chunk, n);
    double d1 = 0.0, d2 = 0.0;
    for (long long e = e0 + threadIdx.x; e < e1; e += kT) { const double f = (double)(float)base[e]; d1 += f; d2 += f * f; }
    d1 = wave_sum_f64(d1); d2 = wave_sum_f64(d2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[2 * wave] = d1; red[2 * wave + 1] = d2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a1 = 0.0, a2 = 0.0;
        for (int w = 0; w < kT / 64; ++w) { a1 += red[2 * w]; a2 += red[2 * w + 1]; }
        double* dst = p.partials + ((long long)bg * p.nsplit + sp) * 2;
        dst[0] = a1; dst[1] = a2;
    }
}

__global__ __launch_bounds__(kT) void gn_apply_scalar(GnParams p, long long n, long long chunk) {
    const int bg = blockIdx.x / p.nsplit, sp = blockIdx.x % p.nsplit;
    const int g = bg % p.groups;
    const double* src = p.partials + (long long)bg * p.nsplit * 2;
    double a1 = 0.0, a2 = 0.0;
    for (int i = 0; i < p.nsplit; ++i) { a1 += src[2 * i]; a2 += src[2 * i + 1]; }
    const double mean_d = a1 / (double)n;
    double var = a2 / (double)n - mean_d * mean_d;
    var = var > 0.0 ? var : 0.0;
    const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + (double)p.eps));
    const half_t* base = p.x + (long long)bg * n;
    half_t* out = p.y + (long long)bg * n;
    const long long e0 = (long long)sp * chunk, e1 = min(e0 + chunk, n);
    for (long long e = e0 + threadIdx.x; e < e1; e += kT) {
        const int c = g * p.cpg + (int)(e / p.hw);
        const float ga = (float)p.gamma[c] * rstd, be = (float)p.beta[c] - mean * ga;
        float f = (float)base[e] * ga + be;
        if (p.silu) f = silu_f(f);
        out[e] = (half_t)f;
    }
}

void plan(GnParams& p, int B) {
    const int ng = B * p.groups;
    int ns = (1024 + ng - 1) / ng;                             // aim for >= 1024 workgroups
    const long long max_by_size = p.n8 / 256 > 0 ? p.n8 / 256 : 1;     // >= 2048 halves per chunk
    if (ns > max_by_size) ns = (int)max_by_size;
    if (ns < 1) ns = 1;
    if (ns > 64) ns = 64;
    p.nsplit = ns;
    p.chunk8 = (int)((p.n8 + ns - 1) / ns);
}

__global__ __launch_bounds__(kT) void geglu_kernel(const half_t* x, half_t* y, long long rows, int n8) {
    const long long total = rows * n8;
    for (long long i = blockIdx.x * (long long)kT + threadIdx.x; i < total; i += (long long)gridDim.x * kT) {
        const long long r = i / n8;
        const int j8 = (int)(i - r * n8);
        const half_t* row = x + r * (long long)n8 * 16;
        const h8_t hv = *reinterpret_cast<const h8_t*>(row + j8 * 8);
        const h8_t gv = *reinterpret_cast<const h8_t*>(row + (long long)n8 * 8 + j8 * 8);
        h8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gte = (float)gv[j];
            const float gel = gelu_erf(gte);
            // diffusers GEGLU: hidden_states * gelu(gate); gelu output is an fp16 tensor before the product
            o[j] = (half_t)((float)hv[j] * (float)(half_t)gel);
        }
        *reinterpret_cast<h8_t*>(y + i * 8) = o;
    }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" size_t dsc_groupnorm_workspace_bytes(int B, int C, int hw, int groups) {
    if (B <= 0 || C <= 0 || hw <= 0 || groups <= 0 || C % groups) return 0;
    return (size_t)B * groups * 64 * 2 * sizeof(double);
}

extern "C" int dsc_groupnorm_silu(const void* x, void* y, const void* gamma, const void* beta, int B, int C, int hw,
                                  int groups, float eps, int apply_silu, int dtype, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    if (!x || !y || !gamma || !beta || B <= 0 || C <= 0 || hw <= 0 || groups <= 0 || C % groups) return DSC_ERR_BAD_ARG;
    if (dtype != DSC_F16 || !al16(x) || !al16(y)) return DSC_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < dsc_groupnorm_workspace_bytes(B, C, hw, groups) ||
        (reinterpret_cast<uintptr_t>(workspace) & 7))
        return DSC_ERR_WORKSPACE;
    GnParams p{};
    p.x = static_cast<const half_t*>(x); p.y = static_cast<half_t*>(y);
    p.gamma = static_cast<const half_t*>(gamma); p.beta = static_cast<const half_t*>(beta);
    p.partials = static_cast<double*>(workspace);
    p.C = C; p.hw = hw; p.groups = groups; p.cpg = C / groups;
    p.n8 = (long long)p.cpg * hw / 8;
    p.eps = eps; p.silu = apply_silu;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hw % 8 != 0) {                       // scalar fallback: correctness path for odd spatial sizes
        const long long n = (long long)p.cpg * hw;
        p.nsplit = 1;
        const dim3 grid(B * groups), block(kT);
        DSC_LAUNCH(gn_stats_scalar, grid, block, 0, st, p, n, n);
        DSC_LAUNCH(gn_apply_scalar, grid, block, 0, st, p, n, n);
        return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
    }
    plan(p, B);
    const dim3 grid(B * groups * p.nsplit), block(kT);
    DSC_LAUNCH(gn_stats, grid, block, 0, st, p);
    DSC_LAUNCH(gn_apply, grid, block, 0, st, p);
    return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
}

extern "C" int dsc_geglu(const void* x, void* y, int64_t rows, int n, int dtype, void* stream) {
    if (!x || !y || rows <= 0 || n <= 0) return DSC_ERR_BAD_ARG;
    if (dtype != DSC_F16 || n % 8 != 0 || !al16(x) || !al16(y)) return DSC_ERR_UNSUPPORTED;
    const long long total = rows * (n / 8);
    long long g = (total + kT - 1) / kT;
    if (g > 4096) g = 4096;
    DSC_LAUNCH(geglu_kernel, dim3((int)g), dim3(kT), 0, static_cast<hipStream_t>(stream),
                       static_cast<const half_t*>(x), static_cast<half_t*>(y), (long long)rows, n / 8);
    return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
}
