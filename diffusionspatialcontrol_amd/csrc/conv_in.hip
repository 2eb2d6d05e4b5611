// 3x3 / pad 1 convolution with a handful of input channels (dsc_conv3x3_fewcin_f16): the UNet's `conv_in` (4 -> 320,
// reference u_net_condition_modify.py:352-356 instantiates it, :1187 calls it).  K = 9 * Cin = 36 is far below an MFMA
// tile, and MIOpen has no NHWC kernel for 4 channels (the module ran it NCHW and converted: 23 + 8 us); this is a
// direct kernel: input NCHW (the sampler's latent layout) -> output NHWC (the layout the UNet keeps), bias fused.
//   block = 8 consecutive pixels of a row x all output channels; thread = (pixel, 8 output channels)
//   the 3 x 10 x Cin input patch is staged in LDS as fp32 (zero padding applied there), every thread walks the
//   9 * Cin taps with one 16-byte load of the transposed weights w_t[k][Cout] per tap (coalesced over the 8-channel groups)
#include "dsc_common.h"
#include "dsc_hip.h"

namespace {

constexpr int kPix = 8, kMaxCin = 16;      // 16: the 9-channel inpainting UNet's conv_in fits too

__global__ __launch_bounds__(512) void conv_fewcin_kernel(const half_t* x, const half_t* wt, const half_t* bias, half_t* out,
                                                          int B, int Cin, int H, int W, int Cout) {
    __shared__ float patch[kMaxCin][3][kPix + 2];
    const int groups = Cout >> 3;                               // 8-channel groups
    const int wblk = W / kPix;
    const int bx = blockIdx.x % wblk, y = (blockIdx.x / wblk) % H, b = blockIdx.x / (wblk * H);
    const int x0 = bx * kPix;
    for (int i = threadIdx.x; i < Cin * 3 * (kPix + 2); i += blockDim.x) {
        const int c = i / (3 * (kPix + 2)), r = (i / (kPix + 2)) % 3, col = i % (kPix + 2);
        const int yy = y - 1 + r, xx = x0 - 1 + col;
        float v = 0.f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = (float)x[(((long long)b * Cin + c) * H + yy) * W + xx];
        patch[c][r][col] = v;
    }
    __syncthreads();
    const int g = threadIdx.x % groups, px = threadIdx.x / groups;
    if (px >= kPix) return;
    float acc[8];
    {
        h8_t bv = {0, 0, 0, 0, 0, 0, 0, 0};
        if (bias) bv = *reinterpret_cast<const h8_t*>(bias + g * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = (float)bv[j];
    }
    for (int c = 0; c < Cin; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float xv = patch[c][r][px + dx];
                const h8_t wv = *reinterpret_cast<const h8_t*>(wt + (long long)((c * 3 + r) * 3 + dx) * Cout + g * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += xv * (float)wv[j];
            }
    h8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)acc[j];
    *reinterpret_cast<h8_t*>(out + (((long long)b * H + y) * W + x0 + px) * Cout + g * 8) = o;
}

}  // namespace

extern "C" int dsc_conv3x3_fewcin_f16(const void* x_nchw, const void* w_t, const void* bias, void* out_nhwc,
                                      int B, int Cin, int H, int W, int Cout, int dtype, void* stream) {
    if (!x_nchw || !w_t || !out_nhwc || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return DSC_ERR_BAD_ARG;
    if (dtype != DSC_F16 || Cin > kMaxCin || Cout % 8 != 0 || W % kPix != 0 || (Cout / 8) * kPix > 512 ||
        (reinterpret_cast<uintptr_t>(w_t) & 15) || (reinterpret_cast<uintptr_t>(out_nhwc) & 15) ||
        (bias && (reinterpret_cast<uintptr_t>(bias) & 15)))
        return DSC_ERR_UNSUPPORTED;
    const int threads = (Cout / 8) * kPix;
    DSC_LAUNCH(conv_fewcin_kernel, dim3((unsigned)((long long)B * H * (W / kPix))), dim3(threads), 0,
                       static_cast<hipStream_t>(stream), static_cast<const half_t*>(x_nchw), static_cast<const half_t*>(w_t),
                       static_cast<const half_t*>(bias), static_cast<half_t*>(out_nhwc), B, Cin, H, W, Cout);
    return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
}
