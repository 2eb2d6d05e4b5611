// Fused sampler-step kernels (include/dsc_hip.h: dsc_prepare_unet_input, dsc_cfg_dpmpp2m_step, dsc_dpmpp2m_update).
// Pure HBM-bound elementwise work on [n_img, 4, h, w] latents (32 KB per 512x512 image): the point is launch
// count - one launch per step instead of ~10 - and keeping sigma / t on the device for the captured UNet graph.
#include "dsc_common.h"
#include "dsc_hip.h"

namespace {

__device__ __forceinline__ void unpack8(const h8_t v, float (&f)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (float)v[j];
}
__device__ __forceinline__ h8_t pack8(const float (&f)[8]) {
    h8_t v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (half_t)f[j];
    return v;
}

// the step's row of a per-generation table (the time-embedding projections of every ResNet block for this step's timestep,
// computed for all steps before the loop) copied to every row of the static buffer the captured UNet step reads
__device__ __forceinline__ void copy_row(const half_t* src, half_t* dst, int halfs, int copies) {
    if (!src) return;
    const int v8 = halfs / 8;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < (long long)v8 * copies; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / v8), c = (int)(i - (long long)r * v8);
        *reinterpret_cast<h8_t*>(dst + (long long)r * halfs + c * 8) = *reinterpret_cast<const h8_t*>(src + c * 8);
    }
}

__global__ __launch_bounds__(256) void prepare_kernel(const half_t* x, float c_in, float t, float sigma, half_t* x_in,
                                                      float* t_buf, float* sigma_buf, int n_img, int chw,
                                                      const half_t* row_src, half_t* row_dst, int row_halfs, int row_copies) {
    copy_row(row_src, row_dst, row_halfs, row_copies);
    const long long n8 = (long long)n_img * chw / 8;
    const long long half_elems = (long long)n_img * chw;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        float f[8];
        unpack8(*reinterpret_cast<const h8_t*>(x + i * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] *= c_in;
        const h8_t o = pack8(f);
        *reinterpret_cast<h8_t*>(x_in + i * 8) = o;
        *reinterpret_cast<h8_t*>(x_in + half_elems + i * 8) = o;
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x < 2 * n_img) t_buf[threadIdx.x] = t;
        for (int i = threadIdx.x + 256; i < 2 * n_img; i += 256) t_buf[i] = t;
        if (threadIdx.x == 0) sigma_buf[0] = sigma;
    }
}

__global__ __launch_bounds__(256) void step_kernel(half_t* x, const half_t* eps, half_t* old, float sigma, float g,
                                                   float a, float b, float c, float c_in_next, float t_next,
                                                   float sigma_next, half_t* x_in, float* t_buf, float* sigma_buf,
                                                   int n_img, int chw,
                                                   const half_t* row_src, half_t* row_dst, int row_halfs, int row_copies) {
    copy_row(row_src, row_dst, row_halfs, row_copies);
    const long long half_elems = (long long)n_img * chw;
    const long long n8 = half_elems / 8;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        float xv[8], eu[8], ec[8], ov[8], dn[8], xn[8], xi[8];
        unpack8(*reinterpret_cast<const h8_t*>(x + i * 8), xv);
        unpack8(*reinterpret_cast<const h8_t*>(eps + i * 8), eu);
        unpack8(*reinterpret_cast<const h8_t*>(eps + half_elems + i * 8), ec);
        unpack8(*reinterpret_cast<const h8_t*>(old + i * 8), ov);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float e = eu[j] + g * (ec[j] - eu[j]);          // model_k_diffusion.py:1162-1166 (affine in eps)
            dn[j] = (float)(half_t)(xv[j] - sigma * e);           // external_k_diffusion.py:114, stored as fp16
            xn[j] = (float)(half_t)(a * xv[j] + b * dn[j] + c * ov[j]);
            xi[j] = xn[j] * c_in_next;
        }
        *reinterpret_cast<h8_t*>(old + i * 8) = pack8(dn);
        *reinterpret_cast<h8_t*>(x + i * 8) = pack8(xn);
        const h8_t o = pack8(xi);
        *reinterpret_cast<h8_t*>(x_in + i * 8) = o;
        *reinterpret_cast<h8_t*>(x_in + half_elems + i * 8) = o;
    }
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < 2 * n_img; i += 256) t_buf[i] = t_next;
        if (threadIdx.x == 0) sigma_buf[0] = sigma_next;
    }
}

__global__ __launch_bounds__(256) void update_kernel(const half_t* x, const half_t* den, const half_t* old, float a,
                                                     float b, float c, half_t* out, long long n8) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        float xv[8], dv[8], ov[8], r[8];
        unpack8(*reinterpret_cast<const h8_t*>(x + i * 8), xv);
        unpack8(*reinterpret_cast<const h8_t*>(den + i * 8), dv);
        if (old) unpack8(*reinterpret_cast<const h8_t*>(old + i * 8), ov);
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = a * xv[j] + b * dv[j] + (old ? c * ov[j] : 0.f);
        *reinterpret_cast<h8_t*>(out + i * 8) = pack8(r);
    }
}

int grid_for(long long n8) {
    long long g = (n8 + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}
bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

namespace {
int row_args_status(const void* row_src, const void* row_dst, int row_halfs, int row_copies) {
    if (!row_src) return DSC_OK;
    if (!row_dst || row_halfs <= 0 || row_copies <= 0) return DSC_ERR_BAD_ARG;
    if (row_halfs % 8 != 0 || !al16(row_src) || !al16(row_dst)) return DSC_ERR_UNSUPPORTED;
    return DSC_OK;
}
}  // namespace

extern "C" int dsc_prepare_unet_input(const void* x, float c_in, float t, float sigma, void* x_in, float* t_buf,
                                      float* sigma_buf, int n_img, int chw, int dtype,
                                      const void* row_src, void* row_dst, int row_halfs, int row_copies, void* stream) {
    if (!x || !x_in || !t_buf || !sigma_buf || n_img <= 0 || chw <= 0) return DSC_ERR_BAD_ARG;
    if (dtype != DSC_F16 || chw % 8 != 0 || !al16(x) || !al16(x_in)) return DSC_ERR_UNSUPPORTED;
    if (const int rs = row_args_status(row_src, row_dst, row_halfs, row_copies)) return rs;
    const long long n8 = (long long)n_img * chw / 8;
    DSC_LAUNCH(prepare_kernel, dim3(grid_for(n8)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const half_t*>(x), c_in, t, sigma, static_cast<half_t*>(x_in), t_buf, sigma_buf,
                       n_img, chw, static_cast<const half_t*>(row_src), static_cast<half_t*>(row_dst), row_halfs, row_copies);
    return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
}

extern "C" int dsc_cfg_dpmpp2m_step(void* x, const void* eps, void* old, float sigma, float guidance, float a, float b,
                                    float c, float c_in_next, float t_next, float sigma_next, void* x_in, float* t_buf,
                                    float* sigma_buf, int n_img, int chw, int dtype,
                                    const void* row_src, void* row_dst, int row_halfs, int row_copies, void* stream) {
    if (!x || !eps || !old || !x_in || !t_buf || !sigma_buf || n_img <= 0 || chw <= 0) return DSC_ERR_BAD_ARG;
    if (dtype != DSC_F16 || chw % 8 != 0 || !al16(x) || !al16(eps) || !al16(old) || !al16(x_in)) return DSC_ERR_UNSUPPORTED;
    if (const int rs = row_args_status(row_src, row_dst, row_halfs, row_copies)) return rs;
    const long long n8 = (long long)n_img * chw / 8;
    DSC_LAUNCH(step_kernel, dim3(grid_for(n8)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<half_t*>(x), static_cast<const half_t*>(eps), static_cast<half_t*>(old), sigma,
                       guidance, a, b, c, c_in_next, t_next, sigma_next, static_cast<half_t*>(x_in), t_buf, sigma_buf,
                       n_img, chw, static_cast<const half_t*>(row_src), static_cast<half_t*>(row_dst), row_halfs, row_copies);
    return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
}

extern "C" int dsc_dpmpp2m_update(const void* x, const void* denoised, const void* old, float a, float b, float c,
                                  void* out, int64_t n, int dtype, void* stream) {
    if (!x || !denoised || !out || n <= 0 || (!old && c != 0.f)) return DSC_ERR_BAD_ARG;
    if (dtype != DSC_F16 || n % 8 != 0 || !al16(x) || !al16(denoised) || !al16(out) || (old && !al16(old)))
        return DSC_ERR_UNSUPPORTED;
    DSC_LAUNCH(update_kernel, dim3(grid_for(n / 8)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const half_t*>(x), static_cast<const half_t*>(denoised),
                       static_cast<const half_t*>(old), a, b, c, static_cast<half_t*>(out), (long long)(n / 8));
    return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
}
