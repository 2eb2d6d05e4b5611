// Region-biased cross-attention for gfx950 (MI355X): the kernel behind dsc_region_xattn_fwd / _std.
//
// Replaces the reference's op chain  q@k^T -> *scale -> std() -> w*sigma*std -> repeat_interleave -> += ->
// softmax -> @v  (source/modules/attention_modify.py:74-103, source/app.py:1004).
//
// The std is GLOBAL over a std group (all heads, all rows of the group), so the work is two launches:
//   phase 1  xattn_stats : scores on MFMA, per-workgroup (sum, sum of squares) in fp64 -> workspace partials
//   phase 2  xattn_fwd   : every workgroup re-reduces its group's partials (a few KB, L2-resident) to the
//                          std, recomputes its scores, adds region*sigma*std, softmax, PV, stores fp16 out
// Nothing is zero-initialised, no atomics: results are bit-reproducible run to run.
//
// Decomposition (one wave = 32 query rows of one (b, h)):
//   S^T[s, l] = K[s, :] . Q[l, :]   "swapped" product: A = K (rows s), B = Q^T, so that after the MFMA every
//   lane owns ONE query row l = lane & 31 and half of its S key columns in registers -> the softmax row
//   reduction is lane-local plus one exchange with lane ^ 32 (no LDS, no 5-stage shuffle tree).
//   O^T[dd, l] = sum_s V^T[dd, s] P^T[s, l]: the fp16-packed score registers are already the B operand of this
//   MFMA (k order permuted inside each 16-step); the A operand V^T is read from an LDS image transposed at
//   staging time in exactly that permuted order.
// K (pad columns zeroed) and V (row-major; the V^T operand is read with ds_read_b64_tr_b16, the hardware transpose)
// are staged once per workgroup in LDS and shared by its 4 waves; the fp32 region tile of a wave (32 rows x S,
// contiguous in HBM) is staged through LDS with coalesced 16-byte loads.  ALL global loads of the prologue (K, V,
// Q fragments, region tile, std partials) are issued up front into registers with compile-time trip counts, so the
// workgroup pays ONE memory round trip before its first MFMA instead of one per staging-loop iteration.
// blockIdx is mapped so that the H heads of one (b, row chunk) share an XCD: they re-read the same region
// rows and the same 128-B lines of Q from that XCD's L2.
#include "xattn_shared.h"

using namespace dsc_xattn;

namespace {

// scores of one 32-row tile: acc[m] element i <-> s = 32m + (i & 3) + 8 (i >> 2) + 4 hh, l = lane & 31
template <int NK, bool REF16, bool RAW = false>
__device__ __forceinline__ void scores(const XattnParams& p, const half_t* Ks, const h8_t (&qf)[NK], f16x_t (&acc)[3],
                                       int r, int hh, float scale) {
    constexpr int KP = XCfg<NK>::KP;
    const int mt = (p.S + 31) >> 5;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
        if (m < mt) {
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const h8_t kf = *reinterpret_cast<const h8_t*>(Ks + (32 * m + r) * KP + 16 * ks + 8 * hh);
                acc[m] = mfma_32x32x16(kf, qf[ks], acc[m]);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            // attention_modify.py:90 - the matmul result is an fp16 tensor, then * scale_factor rounds again
            if (REF16) acc[m][i] = round_f16(pin_f32(round_f16(acc[m][i]) * scale));
            else if (!RAW) acc[m][i] = acc[m][i] * scale;
        }
    }
}

// ------------------------------------------------------------------------------------------ phase 1
template <int NK, bool REF16>
__global__ __launch_bounds__(kThreads, 2) void xattn_stats(XattnParams p) {   // 2 waves per SIMD: score accumulators stay in VGPRs
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KP = XCfg<NK>::KP;
    half_t* Ks = reinterpret_cast<half_t*>(smem);
    double* red = reinterpret_cast<double*>(smem + kSMax * KP * 2);
    int b, h, chunk;
    block_to_work(p, b, h, chunk);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, hh = lane >> 5;
    h8_t kr[XCfg<NK>::CH], vr[XCfg<NK>::CH];
    kv_load<NK, false>(p, b, h, kr, vr);
    h8_t qf[NK];
    {   // first tile's Q fragments ride the same round trip as K
        const int l0f = (chunk * kWaves * p.tiles_per_wave + wave) * 32;
        load_q_frags<NK>(p, qf, b, h, min(l0f + r, p.L - 1), hh);
    }
    kv_store<NK, false>(p, Ks, nullptr, kr, vr);
    __syncthreads();
    float s1 = 0.f, s2 = 0.f;
    double d1 = 0.0, d2 = 0.0;
    for (int t = 0; t < p.tiles_per_wave; ++t) {
        const int l0 = (chunk * kWaves * p.tiles_per_wave + t * kWaves + wave) * 32;
        if (l0 >= p.L) break;
        const int row = min(l0 + r, p.L - 1);
        const bool row_ok = l0 + r < p.L;
        if (t > 0) load_q_frags<NK>(p, qf, b, h, row, hh);
        f16x_t acc[3];
        scores<NK, REF16>(p, Ks, qf, acc, r, hh, p.scale);
        const float* mrow = p.mask ? p.mask + (long long)(b * p.H + h) * p.msbh + (long long)row * p.msl : nullptr;   // wave-uniform test
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int s = 32 * m + (i & 3) + 8 * (i >> 2) + 4 * hh;
                float a = acc[m][i];
                if (mrow) {                                   // attn_weight += attn_bias (:91): an fp16 tensor in the fp16 pipeline
                    a += mrow[min(s, p.S - 1)];
                    if (REF16) a = round_f16(a);
                }
                a = (row_ok && s < p.S) ? a : 0.f;
                s1 += a;
                s2 += a * a;
            }
        d1 += (double)s1; d2 += (double)s2;          // fp32 only within one tile (48 values per lane)
        s1 = 0.f; s2 = 0.f;
    }
    d1 = wave_sum_f64(d1);
    d2 = wave_sum_f64(d2);
    if (lane == 0) { red[2 * wave] = d1; red[2 * wave + 1] = d2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a1 = 0.0, a2 = 0.0;
        for (int w = 0; w < kWaves; ++w) { a1 += red[2 * w]; a2 += red[2 * w + 1]; }
        const int bg = fdiv(b, p.fd_ngroups), g = b - bg * p.n_groups;
        const int idx = (bg * p.H + h) * p.nchunks + chunk;
        double* dst = p.partials + ((long long)g * p.npart + idx) * 2;
        dst[0] = a1; dst[1] = a2;
    }
}

// finalises std_out[g] for dsc_region_xattn_std (one block per group)
template <bool REF16>
__global__ __launch_bounds__(kThreads) void xattn_std_finalize(XattnParams p) {
    __shared__ double red[kRedBytes / 8];
    const float sd = group_std(p, blockIdx.x, red, REF16);
    if (threadIdx.x == 0) p.std_out[blockIdx.x] = sd;
}

// ------------------------------------------------------------------------------------------ phase 2
constexpr int kWCH = (32 * kSMax / 4 + 63) / 64;             // float4 chunks per lane of one 32 x S region tile (12)

template <int NK, bool REF16>
__global__ __launch_bounds__(kThreads, (NK <= 5 ? 2 : 1)) void xattn_fwd(XattnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using C = XCfg<NK>;
    constexpr int KP = C::KP, VP = C::VP, DM = C::DM;
    half_t* Ks = reinterpret_cast<half_t*>(smem);
    half_t* Vs = Ks + kSMax * KP;
    double* red = reinterpret_cast<double*>(Vs + kSMax * VP);
    float* Wt_all = reinterpret_cast<float*>(red + kRedBytes / 8);
    int b, h, chunk;
    block_to_work(p, b, h, chunk);
    if (p.flags & 8u) return;                                // DEBUG timing probe: launch floor
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, hh = lane >> 5;
    const bool has_bias = p.region != nullptr;
    const bool need_std = has_bias && !(p.flags & DSC_FLAG_BIAS_IS_FINAL);
    const int wt_stride = (32 * p.S + 3) & ~3;
    float* Wt = Wt_all + wave * wt_stride;
    const int bw = has_bias ? fdiv(b * p.H + h, p.fd_rep) : 0;   // repeat_interleave, :96-99
    const int mt = (p.S + 31) >> 5, nt = (p.S + 15) >> 4;

    // ---- prologue: every global load of the workgroup is issued before anything waits
    h8_t kr[C::CH], vr[C::CH];
    kv_load<NK, true>(p, b, h, kr, vr);
    double pa1 = 0.0, pa2 = 0.0;
    if (need_std) group_partials(p, b - fdiv(b, p.fd_ngroups) * p.n_groups, pa1, pa2);
    float sig = 1.f;
    if (need_std) sig = p.sigma_dev ? *p.sigma_dev : p.sigma_host;

    h8_t qf[NK];
    f4x_t wreg[kWCH];
    auto tile_loads = [&](int l0, int nrows, int row, bool& vec) {
        load_q_frags<NK>(p, qf, b, h, row, hh);
        vec = false;
        if (has_bias) {                                      // fp32 region rows l0 .. l0+nrows-1 are contiguous
            const long long off = ((long long)bw * p.L + l0) * p.S;
            const int cnt = nrows * p.S;
            vec = ((off & 3) == 0) && ((cnt & 3) == 0);
            if (vec) {
                const f4x_t* src = reinterpret_cast<const f4x_t*>(p.region + off);
#pragma unroll
                for (int c = 0; c < kWCH; ++c) {
                    const int i = lane + 64 * c;
                    if (i * 4 < cnt) wreg[c] = src[i];
                }
            }
        }
    };
    auto tile_store_w = [&](int l0, int nrows, bool vec) {
        if (!has_bias) return;
        const int cnt = nrows * p.S;
        if (vec) {
#pragma unroll
            for (int c = 0; c < kWCH; ++c) {
                const int i = lane + 64 * c;
                if (i * 4 < cnt) *reinterpret_cast<f4x_t*>(Wt + 4 * i) = wreg[c];
            }
        } else {                                             // odd alignment / ragged tail: plain dword loop
            const float* src = p.region + ((long long)bw * p.L + l0) * p.S;
            for (int i = lane; i < cnt; i += 64) Wt[i] = src[i];
        }
    };

    int l0 = (chunk * kWaves * p.tiles_per_wave + wave) * 32;
    bool tile_ok = l0 < p.L, vec = false;
    int nrows = tile_ok ? min(32, p.L - l0) : 0;
    int row = tile_ok ? min(l0 + r, p.L - 1) : 0;
    if (tile_ok) tile_loads(l0, nrows, row, vec);

    kv_store<NK, true>(p, Ks, Vs, kr, vr);
    float sd = 1.f;
    if (need_std) sd = group_std_finish(p, pa1, pa2, red, REF16);          // contains a __syncthreads()
    if (tile_ok) tile_store_w(l0, nrows, vec);

    // transposed-read lane offset: row 4 hh + ((lane & 15) >> 2), column 16 ((lane >> 4) & 1) + 4 (lane & 3)
    const int tr_off = (4 * hh + ((lane & 15) >> 2)) * VP + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

    for (int t = 0; t < p.tiles_per_wave; ++t) {
        if (t > 0) {
            l0 = (chunk * kWaves * p.tiles_per_wave + t * kWaves + wave) * 32;
            tile_ok = l0 < p.L;
            nrows = tile_ok ? min(32, p.L - l0) : 0;
            row = tile_ok ? min(l0 + r, p.L - 1) : 0;
            if (tile_ok) { tile_loads(l0, nrows, row, vec); tile_store_w(l0, nrows, vec); }
        }
        __syncthreads();          // t == 0: K / V images complete; every t: this wave's Wt tile is written
        if (p.flags & 16u) return;                           // DEBUG timing probe: prologue only
        if (tile_ok) {
            f16x_t acc[3];
            scores<NK, REF16, !REF16>(p, Ks, qf, acc, r, hh, p.scale);
            h8_t pf[6];
            float oscale = 1.f;
            const float* brow = has_bias ? Wt + min(r, nrows - 1) * p.S : nullptr;
            if (REF16) softmax_tile<REF16, false>(acc, pf, brow, sig, sd, p.S, hh);
            else oscale = softmax_tile_lean<false>(acc, pf, brow, sig, sd, p.scale * 1.4426950408889634f, p.S, hh);

            half_t* ob = p.out + b * p.osb + h * p.osh + (long long)row * p.osl;
            const bool row_ok = l0 + r < p.L;
#pragma unroll
            for (int dm = 0; dm < DM; ++dm) {
                if (32 * dm < p.d) {                         // wave-uniform: EXEC stays full for the transposed reads
                    f16x_t o;
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[i] = 0.f;
#pragma unroll
                    for (int tt = 0; tt < 6; ++tt) {
                        if (tt < nt) {
                            const half_t* vp = Vs + tr_off + (16 * tt) * VP + 32 * dm;
                            const h4_t lo = tr_read(vp);
                            const h4_t hi = tr_read(vp + 8 * VP);
                            const h8_t vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                            o = mfma_32x32x16(vf, pf[tt], o);
                        }
                    }
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int dd0 = 32 * dm + 8 * g4 + 4 * hh;
                        if (row_ok && dd0 < p.d) {
                            const h4_t ov = {(half_t)(o[4 * g4] * oscale), (half_t)(o[4 * g4 + 1] * oscale),
                                             (half_t)(o[4 * g4 + 2] * oscale), (half_t)(o[4 * g4 + 3] * oscale)};
                            *reinterpret_cast<h4_t*>(ob + dd0) = ov;
                        }
                    }
                }
            }
        }
        if (t + 1 < p.tiles_per_wave) __syncthreads();       // Wt is rewritten by the next tile
    }
}

template <int NK>
size_t fwd_lds_bytes(int S) {
    const int wt_stride = (32 * S + 3) & ~3;
    return (size_t)kSMax * XCfg<NK>::KP * 2 + (size_t)kSMax * XCfg<NK>::VP * 2 + kRedBytes +
           (size_t)kWaves * wt_stride * 4;
}
template <int NK>
size_t stats_lds_bytes() {
    return (size_t)kSMax * XCfg<NK>::KP * 2 + kRedBytes;
}

int pick_nk(int d) {
    static const int opts[] = {2, 3, 4, 5, 6, 8, 10};
    for (int nk : opts) if (16 * nk >= d) return nk;
    return 0;
}

void plan(XattnParams& p) { plan_tiles(p); }

template <int NK, bool REF16>
int launch_stats(const XattnParams& p, hipStream_t st) {
    const dim3 grid = xattn_grid(p), block(kThreads);
    DSC_LAUNCH((xattn_stats<NK, REF16>), grid, block, stats_lds_bytes<NK>(), st, p);
    return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
}
template <int NK, bool REF16>
int launch_fwd(const XattnParams& p, hipStream_t st) {
    const size_t lds = fwd_lds_bytes<NK>(p.S);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_fwd<NK, REF16>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const dim3 grid = xattn_grid(p), block(kThreads);
    DSC_LAUNCH((xattn_fwd<NK, REF16>), grid, block, lds, st, p);
    return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
}

#define DSC_DISPATCH_NK(nk, FN, ...)                                   \
    switch (nk) {                                                      \
        case 2: return FN<2, __VA_ARGS__;                              \
        case 3: return FN<3, __VA_ARGS__;                              \
        case 4: return FN<4, __VA_ARGS__;                              \
        case 5: return FN<5, __VA_ARGS__;                              \
        case 6: return FN<6, __VA_ARGS__;                              \
        case 8: return FN<8, __VA_ARGS__;                              \
        case 10: return FN<10, __VA_ARGS__;                            \
        default: return DSC_ERR_UNSUPPORTED;                           \
    }

int dispatch_stats(int nk, bool ref16, const XattnParams& p, hipStream_t st) {
    if (ref16) { DSC_DISPATCH_NK(nk, launch_stats, true>(p, st)) }
    DSC_DISPATCH_NK(nk, launch_stats, false>(p, st))
}
int dispatch_fwd(int nk, bool ref16, const XattnParams& p, hipStream_t st) {
    if (ref16) { DSC_DISPATCH_NK(nk, launch_fwd, true>(p, st)) }
    DSC_DISPATCH_NK(nk, launch_fwd, false>(p, st))
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
bool strides_ok(const int64_t s[3]) { return s[0] % 8 == 0 && s[1] % 8 == 0 && s[2] % 8 == 0; }

int check_common(const void* q, const void* k, int Bc, int H, int L, int S, int d, int n_groups,
                 const int64_t* qs, const int64_t* ks, int dtype) {
    if (!q || !k || !qs || !ks) return DSC_ERR_BAD_ARG;
    if (Bc <= 0 || H <= 0 || L <= 0 || S <= 0 || d <= 0 || n_groups <= 0 || Bc % n_groups != 0) return DSC_ERR_BAD_ARG;
    if (dtype != DSC_F16) return DSC_ERR_UNSUPPORTED;
    if (d % 8 != 0 || d > 160 || S > kSMax) return DSC_ERR_UNSUPPORTED;
    if (!aligned16(q) || !aligned16(k) || !strides_ok(qs) || !strides_ok(ks)) return DSC_ERR_UNSUPPORTED;
    return DSC_OK;
}

}  // namespace

extern "C" size_t dsc_region_xattn_workspace_bytes(int Bc, int H, int L, int S, int d, int n_std_groups) {
    (void)d;
    if (Bc <= 0 || H <= 0 || L <= 0 || n_std_groups <= 0) return 0;
    XattnParams p{};
    p.Bc = Bc; p.H = H; p.L = L; p.n_groups = n_std_groups;
    plan_tiles(p, S > kSMax ? 1 : 0);        // the long-prompt kernels (packed path, S > 96) run one tile per wave
    return (size_t)n_std_groups * p.npart * 2 * sizeof(double);
}

extern "C" int dsc_region_xattn_fwd(const void* q, const void* k, const void* v, void* out, const float* region,
                                    int Bc, int H, int L, int S, int d, int Bw, int n_std_groups,
                                    const int64_t q_strides[3], const int64_t k_strides[3],
                                    const int64_t v_strides[3], const int64_t o_strides[3],
                                    float sigma_host, const float* sigma_dev, float scale, int dtype, unsigned flags,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_common(q, k, Bc, H, L, S, d, n_std_groups, q_strides, k_strides, dtype);
    if (rc != DSC_OK) return rc;
    if (!v || !out || !v_strides || !o_strides) return DSC_ERR_BAD_ARG;
    if (!aligned16(v) || !strides_ok(v_strides) || (reinterpret_cast<uintptr_t>(out) & 7) || !strides_ok(o_strides))
        return DSC_ERR_UNSUPPORTED;
    const bool need_stats = region != nullptr && !(flags & DSC_FLAG_BIAS_IS_FINAL);
    if (region) {
        if (Bw <= 0 || (Bc * H) % Bw != 0) return DSC_ERR_BAD_ARG;
        if (reinterpret_cast<uintptr_t>(region) & 3) return DSC_ERR_UNSUPPORTED;
    }
    XattnParams p{};
    p.q = static_cast<const half_t*>(q); p.k = static_cast<const half_t*>(k);
    p.v = static_cast<const half_t*>(v); p.out = static_cast<half_t*>(out);
    p.region = region; p.sigma_dev = sigma_dev; p.sigma_host = sigma_host;
    p.scale = scale > 0.f ? scale : 1.0f / sqrtf((float)d);
    p.Bc = Bc; p.H = H; p.L = L; p.S = S; p.d = d; p.Bw = region ? Bw : 1; p.n_groups = n_std_groups;
    p.qsb = q_strides[0]; p.qsl = q_strides[1]; p.qsh = q_strides[2];
    p.ksb = k_strides[0]; p.kss = k_strides[1]; p.ksh = k_strides[2];
    p.vsb = v_strides[0]; p.vss = v_strides[1]; p.vsh = v_strides[2];
    p.osb = o_strides[0]; p.osl = o_strides[1]; p.osh = o_strides[2];
    p.flags = flags;
    plan(p);
    if (need_stats) {
        const size_t need = (size_t)n_std_groups * p.npart * 2 * sizeof(double);
        if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 7)) return DSC_ERR_WORKSPACE;
        p.partials = static_cast<double*>(workspace);
    }
    const int nk = pick_nk(d);
    const bool ref16 = (flags & DSC_FLAG_REF_FP16_ROUNDING) != 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (need_stats && !(flags & DSC_FLAG_REUSE_STATS)) {
        rc = dispatch_stats(nk, ref16, p, st);
        if (rc != DSC_OK) return rc;
    }
    return dispatch_fwd(nk, ref16, p, st);
}

extern "C" int dsc_region_xattn_std(const void* q, const void* k, int Bc, int H, int L, int S, int d, int n_std_groups,
                                    const int64_t q_strides[3], const int64_t k_strides[3], float scale, int dtype,
                                    unsigned flags, float* std_out, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    return dsc_region_xattn_std_masked(q, k, Bc, H, L, S, d, n_std_groups, q_strides, k_strides, scale, dtype, flags, nullptr,
                                       nullptr, std_out, workspace, workspace_bytes, stream);
}

extern "C" int dsc_region_xattn_std_masked(const void* q, const void* k, int Bc, int H, int L, int S, int d, int n_std_groups,
                                           const int64_t q_strides[3], const int64_t k_strides[3], float scale, int dtype,
                                           unsigned flags, const float* mask, const int64_t mask_strides[2], float* std_out,
                                           void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_common(q, k, Bc, H, L, S, d, n_std_groups, q_strides, k_strides, dtype);
    if (rc != DSC_OK) return rc;
    if (!std_out || (mask && !mask_strides)) return DSC_ERR_BAD_ARG;
    if (mask && ((reinterpret_cast<uintptr_t>(mask) & 3) || mask_strides[0] < 0 || mask_strides[1] < 0)) return DSC_ERR_UNSUPPORTED;
    XattnParams p{};
    p.q = static_cast<const half_t*>(q); p.k = static_cast<const half_t*>(k);
    p.scale = scale > 0.f ? scale : 1.0f / sqrtf((float)d);
    p.Bc = Bc; p.H = H; p.L = L; p.S = S; p.d = d; p.Bw = 1; p.n_groups = n_std_groups;
    p.qsb = q_strides[0]; p.qsl = q_strides[1]; p.qsh = q_strides[2];
    p.ksb = k_strides[0]; p.kss = k_strides[1]; p.ksh = k_strides[2];
    p.flags = flags; p.std_out = std_out;
    if (mask) { p.mask = mask; p.msbh = mask_strides[0]; p.msl = mask_strides[1]; }
    plan(p);
    const size_t need = (size_t)n_std_groups * p.npart * 2 * sizeof(double);
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 7)) return DSC_ERR_WORKSPACE;
    p.partials = static_cast<double*>(workspace);
    const int nk = pick_nk(d);
    const bool ref16 = (flags & DSC_FLAG_REF_FP16_ROUNDING) != 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = dispatch_stats(nk, ref16, p, st);
    if (rc != DSC_OK) return rc;
    if (ref16) DSC_LAUNCH(xattn_std_finalize<true>, dim3(n_std_groups), dim3(kThreads), 0, st, p);
    else DSC_LAUNCH(xattn_std_finalize<false>, dim3(n_std_groups), dim3(kThreads), 0, st, p);
    return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
}
