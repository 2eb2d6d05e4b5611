// Device helpers shared by the region cross-attention kernels (region_xattn.hip: generic operands;
// region_xattn_packed.hip: pre-packed K/V + compressed region table).
#pragma once
#include "dsc_common.h"
#include "dsc_hip.h"
#include <type_traits>

namespace dsc_xattn {

constexpr int kSMax = 96;        // key length padded to 3 MFMA row tiles
typedef short s4_t __attribute__((__vector_size__(4 * sizeof(short))));
constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kRedBytes = (2 * kThreads + 32) * 8;       // LDS scratch of the fp64 reductions

struct XattnParams {
    const half_t* q; const half_t* k; const half_t* v; half_t* out;
    const float* region;
    const float* sigma_dev;
    double* partials;            // [n_groups][npart][2]
    float* std_out;              // optional [n_groups]
    float sigma_host, scale;
    int Bc, H, L, S, d, Bw, n_groups;
    int nchunks, tiles_per_wave, npart, xcd_map;
    long long qsb, qsl, qsh, ksb, kss, ksh, vsb, vss, vsh, osb, osl, osh;
    unsigned flags;
    int wide_store;              // out and its strides are 16-byte aligned: 16-byte row pieces (store_o_block, dsc_common.h)
    // additive attention mask of the statistics pass (dsc_region_xattn_std_masked; reference attention_modify.py:85-95:
    // the std is taken over scale * q.k^T + mask): fp32, element (bh, l, s) at mask[bh * msbh + l * msl + s]; strides 0 = broadcast
    const float* mask; long long msbh, msl;
    double inv_n, inv_nm1;       // 1 / n and 1 / (n - 1), n = scores per std group (plan_tiles)
    FastDiv fd_nchunks, fd_ngroups, fd_rep;      // by nchunks, n_groups, (Bc H) / Bw (plan_tiles)
};

// grid = (8, H, Bc*nchunks/8) when Bc*nchunks % 8 == 0, else (1, H, Bc*nchunks).  The hardware deals linear workgroup
// ids x + 8*(y + H*z) round-robin over the 8 XCDs, so the H heads (y) of one (b, row chunk) = (z, x) share an XCD and
// re-read the same region rows / Q lines from that XCD's L2 - with no integer division except b = cg / nchunks (a multiply: FastDiv).
__device__ __forceinline__ void block_to_work(const XattnParams& p, int& b, int& h, int& chunk) {
    h = blockIdx.y;
    const int cg = blockIdx.z * gridDim.x + blockIdx.x;
    b = fdiv(cg, p.fd_nchunks);
    chunk = cg - b * p.nchunks;
}
inline dim3 xattn_grid(const XattnParams& p) {
    const int ncg = p.Bc * p.nchunks;
    return (ncg % 8 == 0) ? dim3(8, p.H, ncg / 8) : dim3(1, p.H, ncg);
}

template <int NK>
struct XCfg {
    static constexpr int DM = (NK + 1) / 2;
    static constexpr int KP = 16 * NK + 8;                  // K row stride (halves): odd multiple of 16 B
    static constexpr int VP = (DM <= 3) ? 96 : 160;         // V row stride: tr reads conflict-free ((VP/2) % 64 in {16,48})
    static constexpr int CH = (kSMax * 2 * NK + kThreads - 1) / kThreads;   // 16-B chunks per thread per operand
};

__device__ __forceinline__ h4_t tr_read(const half_t* p) {
    const s4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s4_t __attribute__((address_space(3)))*)(const_cast<half_t*>(p)));
    return __builtin_bit_cast(h4_t, r);
}

// K[b, :, h, :] (and V) -> registers: chunk idx = s * d8 + c covers 8 halves; all loads issued back to back
template <int NK, bool WITH_V>
__device__ __forceinline__ void kv_load(const XattnParams& p, int b, int h, h8_t (&kr)[XCfg<NK>::CH], h8_t (&vr)[XCfg<NK>::CH]) {
    const half_t* kb = p.k + b * p.ksb + h * p.ksh;
    const half_t* vb = p.v + b * p.vsb + h * p.vsh;
    const int d8 = p.d >> 3, n = p.S * d8;
#pragma unroll
    for (int c = 0; c < XCfg<NK>::CH; ++c) {
        const int idx = threadIdx.x + c * kThreads;
        if (idx < n) {
            const int s = idx / d8, col = idx - s * d8;
            kr[c] = *reinterpret_cast<const h8_t*>(kb + s * p.kss + col * 8);
            if (WITH_V) vr[c] = *reinterpret_cast<const h8_t*>(vb + s * p.vss + col * 8);
        }
    }
}

// registers -> LDS images Ks[96][KP] / Vs[96][VP] (row-major), plus the few zeros the MFMAs need:
//   K columns [d, 16*NK): the Q fragment is zero there, but 0 * (NaN garbage) would poison the score;
//   V rows [S, 16*ceil(S/16)): P is exactly 0 there, same reason.  K rows >= S only feed scores that are replaced by
//   -inf (a select, not arithmetic) and V columns >= d only feed output rows that are never stored: left as they are.
template <int NK, bool WITH_V>
__device__ __forceinline__ void kv_store(const XattnParams& p, half_t* Ks, half_t* Vs, const h8_t (&kr)[XCfg<NK>::CH],
                                         const h8_t (&vr)[XCfg<NK>::CH]) {
    constexpr int KP = XCfg<NK>::KP, VP = XCfg<NK>::VP;
    const int d8 = p.d >> 3, n = p.S * d8;
#pragma unroll
    for (int c = 0; c < XCfg<NK>::CH; ++c) {
        const int idx = threadIdx.x + c * kThreads;
        if (idx < n) {
            const int s = idx / d8, col = idx - s * d8;
            *reinterpret_cast<h8_t*>(Ks + s * KP + col * 8) = kr[c];
            if (WITH_V) *reinterpret_cast<h8_t*>(Vs + s * VP + col * 8) = vr[c];
        }
    }
    const h8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
    {
        const int padc = 2 * NK - d8;                        // 16-byte pad chunks per row (0 when d == 16*NK)
        for (int idx = threadIdx.x; idx < kSMax * padc; idx += kThreads)
            *reinterpret_cast<h8_t*>(Ks + (idx / padc) * KP + p.d + (idx % padc) * 8) = z;
    }
    if (WITH_V) {
        const int zrows = ((p.S + 15) & ~15) - p.S, v8 = VP / 8;
        for (int idx = threadIdx.x; idx < zrows * v8; idx += kThreads)
            *reinterpret_cast<h8_t*>(Vs + (p.S + idx / v8) * VP + (idx % v8) * 8) = z;
    }
}

template <int NK>
__device__ __forceinline__ void load_q_frags(const XattnParams& p, h8_t (&qf)[NK], int b, int h, int row, int hh) {
    const half_t* qb = p.q + b * p.qsb + h * p.qsh + (long long)row * p.qsl;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        const int col = 16 * ks + 8 * hh;
        h8_t val = {0, 0, 0, 0, 0, 0, 0, 0};
        if (col < p.d) val = *reinterpret_cast<const h8_t*>(qb + col);
        qf[ks] = val;
    }
}

// group std from the partials: every thread of the block gets the same value (fixed summation order).
// Split in two so that the partial loads are in flight together with the K / V / Q loads of the prologue.
__device__ __forceinline__ void group_partials(const XattnParams& p, int g, double& a1, double& a2) {
    const double* src = p.partials + (long long)g * p.npart * 2;
    a1 = 0.0; a2 = 0.0;
    // pairs threadIdx.x, threadIdx.x + 256, ... added in that order; the loads go out four at a time (a loop of load - wait -
    // add made every pair a memory round trip of its own: the partials come from the other XCDs' workgroups of the statistics
    // launch, i.e. from beyond this XCD's L2)
    constexpr int kPL = 4;
    for (int i0 = threadIdx.x; i0 < p.npart; i0 += kPL * kThreads) {
        double v1[kPL], v2[kPL];
#pragma unroll
        for (int u = 0; u < kPL; ++u) {
            const int i = i0 + u * kThreads;
            v1[u] = 0.0; v2[u] = 0.0;
            if (i < p.npart) { v1[u] = src[2 * i]; v2[u] = src[2 * i + 1]; }
        }
#pragma unroll
        for (int u = 0; u < kPL; ++u) { a1 += v1[u]; a2 += v2[u]; }     // (+ 0.0 for the pairs past the end: exact)
    }
}
// the same sum in two halves: issue() starts this thread's first four loads and returns without using them, finish() - called
// once every other load of the caller's prologue is in flight - adds them in group_partials' order (and walks any further
// partials, npart > 4 * kThreads, with the plain loop).  Same bits as group_partials.
struct PartialLoads {
    static constexpr int kPL = 4;
    double v1[kPL], v2[kPL];
    const double* src;
    __device__ __forceinline__ void issue(const XattnParams& p, int g) {
        src = p.partials + (long long)g * p.npart * 2;
#pragma unroll
        for (int u = 0; u < kPL; ++u) {
            const int i = threadIdx.x + u * kThreads;
            v1[u] = 0.0; v2[u] = 0.0;
            if (i < p.npart) { v1[u] = src[2 * i]; v2[u] = src[2 * i + 1]; }
        }
    }
    __device__ __forceinline__ void finish(const XattnParams& p, double& a1, double& a2) const {
        a1 = 0.0; a2 = 0.0;
#pragma unroll
        for (int u = 0; u < kPL; ++u) { a1 += v1[u]; a2 += v2[u]; }
        for (int i0 = threadIdx.x + kPL * kThreads; i0 < p.npart; i0 += kPL * kThreads) {
            double w1[kPL], w2[kPL];
#pragma unroll
            for (int u = 0; u < kPL; ++u) {
                const int i = i0 + u * kThreads;
                w1[u] = 0.0; w2[u] = 0.0;
                if (i < p.npart) { w1[u] = src[2 * i]; w2[u] = src[2 * i + 1]; }
            }
#pragma unroll
            for (int u = 0; u < kPL; ++u) { a1 += w1[u]; a2 += w2[u]; }
        }
    }
};

// `red` needs 2 * kThreads doubles of LDS, 16-byte aligned.  Two halves around ONE workgroup barrier (which a kernel can share
// with the barrier that publishes its LDS images): every thread stores its pair; after the barrier each WAVE adds all 256 pairs
// itself - lane i the pairs i, i + 64, i + 128, i + 192 in that order, then a fixed xor butterfly - so every lane of every wave
// of every workgroup holds the same bits, with no second barrier and no single-thread tail.  (Round 2: 256 -> 16 -> 1 through
// LDS with two barriers, the second one exposed in front of the softmax of every workgroup.)
__device__ __forceinline__ void group_std_stage(double a1, double a2, double* red) {
    red[2 * threadIdx.x] = a1;
    red[2 * threadIdx.x + 1] = a2;
}
// all-lanes sum of a double over the wave with the first four exchange steps on DPP (a few cycles each: lane ^ 1, lane ^ 2, the
// other quad of the 8-lane group by row_half_mirror, the other half of the 16-lane row by row_mirror) and only the last two
// (lanes 16 and 32 apart) through the LDS crossbar (__shfl_xor: ~100 cycles of dependent latency each - six of those, twice,
// in front of every workgroup's softmax were ~1100 cycles).  Every step adds two values that are uniform over the group they
// come from, and a + b = b + a bit for bit: all 64 lanes end with the same bits.
template <int CTRL>
__device__ __forceinline__ double dpp_move_f64(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_allsum_f64(double x) {
    x += dpp_move_f64<0xB1>(x);          // quad_perm [1, 0, 3, 2]
    x += dpp_move_f64<0x4E>(x);          // quad_perm [2, 3, 0, 1]
    x += dpp_move_f64<0x141>(x);         // row_half_mirror
    x += dpp_move_f64<0x140>(x);         // row_mirror
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    return x;
}
__device__ __forceinline__ float group_std_reduce(const XattnParams& p, const double* red, bool ref16) {
    const int lane = threadIdx.x & 63;
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) { t1 += red[2 * (lane + 64 * w)]; t2 += red[2 * (lane + 64 * w) + 1]; }
    t1 = wave_allsum_f64(t1);
    t2 = wave_allsum_f64(t2);
    // unbiased variance (torch.std default) with the two reciprocals from the host (fp64 divides are ~30-instruction
    // sequences and this runs in every wave of every workgroup), the cancellation-prone subtraction still in fp64; the square
    // root in fp32 (the oracle's std is an fp32 tensor: one more rounding at 6e-8)
    double var = (t2 - t1 * t1 * p.inv_n) * p.inv_nm1;
    var = var > 0.0 ? var : 0.0;
    float sd = sqrtf((float)var);
    if (ref16) sd = round_f16(sd);                           // std of an fp16 tensor is a 0-dim fp16 tensor
    return sd;
}
__device__ __forceinline__ float group_std_finish(const XattnParams& p, double a1, double a2, double* red, bool ref16) {
    group_std_stage(a1, a2, red);
    __syncthreads();
    return group_std_reduce(p, red, ref16);
}
__device__ __forceinline__ float group_std(const XattnParams& p, int g, double* red, bool ref16) {
    double a1, a2;
    group_partials(p, g, a1, a2);
    return group_std_finish(p, a1, a2, red, ref16);
}


// an fp32 value the compiler must materialise as computed: no contraction into the consumer (emulation mode only)
__device__ __forceinline__ float pin_f32(float x) { asm volatile("" : "+v"(x)); return x; }

// Key-validity lane masks.  Element (m, i) of a score tile is key c = 32 m + (i & 3) + 8 (i >> 2) in lanes 0-31 and c + 4 in lanes
// 32-63, so "key < S" is two SCALAR compares per element.  Written as a per-lane compare (`s < S` with s built from hh) each
// element is a loop-invariant v_cmp whose 64-bit result the compiler computes once per kernel and keeps: 16-48 SGPR pairs,
// spilled to VGPR lanes and read back two v_readlane at a time (xp_fwd<3>: 58 spilled SGPRs, xp_stats: 35-41).
// `S_now` must come from opaque_s() inside the tile loop, or the masks are hoisted out of it all the same.
__device__ __forceinline__ int opaque_s(int S) { asm volatile("" : "+s"(S)); return S; }
__device__ __forceinline__ unsigned long long key_keep_mask(int c, int S_now) {
    return (c < S_now ? 0x00000000FFFFFFFFull : 0ull) | (c + 4 < S_now ? 0xFFFFFFFF00000000ull : 0ull);
}
__device__ __forceinline__ float keep_or_zero(float a, unsigned long long keep) {
    asm("v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(a) : "s"(keep));
    return a;
}
__device__ __forceinline__ float keep_or(float a, float other, unsigned long long keep) {
    asm("v_cndmask_b32_e64 %0, %2, %0, %1" : "+v"(a) : "s"(keep), "v"(other));
    return a;
}

// Bias of the 4 consecutive keys 32m + 8g + 4hh + {0..3} of this lane's row.  VEC: the row lives in LDS with a
// 16-byte-aligned base and zero padding up to key 99 (packed kernel: one ds_read_b128); else scalar, address clamped.
template <bool VEC>
__device__ __forceinline__ void bias4(const float* brow, int s0, int smax, float (&out)[4]) {
    if (VEC) {
        const f4x_t v = *reinterpret_cast<const f4x_t*>(brow + s0);
        out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3];
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = brow[min(s0 + j, smax)];
    }
}

// (Emulation mode pins every fp32 result that is rounded to fp16 next, or feeds an add, with pin_f32: left to the compiler,
// `(half)(a * b)` and `a + b * c` become a v_fma_mix / fma with ONE rounding in some instantiations and two operations in
// others - one-ulp differences between kernels that are meant to give the same bits.  __fmul_rn does not prevent it.)
// Biased softmax of one 32-row score tile, branch-free.  acc[m][i] holds score (key s = 32m + (i&3) + 8(i>>2) + 4hh,
// query row = lane & 31) on entry; on exit pf[] holds the fp16 probabilities packed as the B operand of the PV MFMA.
// brow: this lane's bias row in LDS (nullptr = no bias, wave-uniform); bias = (brow[s] * mul1) * mul2 (mul1 = sigma,
// mul2 = std for a raw table; 1, 1 for a table that already holds the final bias).
template <bool REF16, bool VEC>
__device__ __forceinline__ void softmax_tile(f16x_t (&acc)[3], h8_t (&pf)[6], const float* brow, float mul1, float mul2,
                                             int S, int hh) {
    const int smax = S - 1;
    float mx = -INFINITY;
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float bias[4] = {0.f, 0.f, 0.f, 0.f};
            if (brow) bias4<VEC>(brow, 32 * m + 8 * g + 4 * hh, smax, bias);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = 4 * g + j, s = 32 * m + 8 * g + 4 * hh + j;
                float a = acc[m][i];
                if (brow) {
                    // w * sigma * std, then the add: three separately rounded fp32 operations, as the reference's tensors are
                    // (app.py:1004, :97) - never contracted into an fma, so every instantiation gives the same bits
                    a = pin_f32(a + pin_f32(pin_f32(bias[j] * mul1) * mul2));
                    if (REF16) a = round_f16(a);
                }
                a = s < S ? a : -INFINITY;                          // select, not a branch
                acc[m][i] = a;
                mx = fmaxf(mx, a);
            }
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float e = __expf(acc[m][i] - mx);
            acc[m][i] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) pf[2 * m + (i >> 3)][i & 7] = (half_t)pin_f32(acc[m][i] * inv);   // fp16 tensor (:101)
}

// Lean variant for fp32 scores (no fp16-rounding emulation): acc holds the RAW q.k products and they STAY raw through the
// maximum - the bias is brought into raw-score units instead (bias * sigma * std / scale: ONE fma per score), and the softmax
// scale rides in the exponent's fma: p = exp2(a' * c - max(a') * c), c = scale * log2 e.  Per score: fma, max, fma, exp, add
// (round 2: two multiplies, fma, compare + select, max, subtract, exp, add).  The key mask (s < S) is applied only in the
// 32-key tile that S actually cuts (wave-uniform test per tile).  p is left UNNORMALISED in pf (values <= 1); the caller
// multiplies the 16 PV outputs per channel tile by the returned 1/sum instead of 48 probabilities.
template <bool VEC>
__device__ __forceinline__ float softmax_tile_lean(f16x_t (&acc)[3], h8_t (&pf)[6], const float* brow, float mul1,
                                                   float mul2, float scale_log2e, int S, int hh) {
    const int smax = S - 1;
    const float bmul = (mul1 * mul2) * (1.4426950408889634f / scale_log2e);     // sigma * std / scale
    float mx = -INFINITY;
    auto pass1 = [&](int m, auto masked) {
        float ninf = 0.f;
        int S_now = S;
        if (decltype(masked)::value) {                         // (-inf lives in a register for this tile only)
            asm volatile("v_mov_b32 %0, 0xff800000" : "=v"(ninf));
            S_now = opaque_s(S);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float bias[4] = {0.f, 0.f, 0.f, 0.f};
            if (brow) bias4<VEC>(brow, 32 * m + 8 * g + 4 * hh, smax, bias);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = 4 * g + j;
                float a = acc[m][i];
                if (brow) a = fmaf(bias[j], bmul, a);
                if (decltype(masked)::value) a = keep_or(a, ninf, key_keep_mask(32 * m + 8 * g + j, S_now));
                acc[m][i] = a;
                mx = fmaxf(mx, a);
            }
        }
    };
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        if (32 * m + 32 <= S) pass1(m, std::false_type{});     // every key of the tile is real: no select
        else pass1(m, std::true_type{});
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float nm = -mx * scale_log2e;
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float e = __builtin_amdgcn_exp2f(fmaf(acc[m][i], scale_log2e, nm));   // exp2(-inf) = 0 for masked keys
            acc[m][i] = e;
            sum += e;
        }
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) pf[2 * m + (i >> 3)][i & 7] = (half_t)acc[m][i];
    sum += __shfl_xor(sum, 32, 64);
    return 1.f / sum;
}

// plan shared by all entry points: one wave = one 32-row tile; a workgroup = 4 waves x tiles_per_wave tiles of one (b, h)
inline void plan_tiles(XattnParams& p, int tiles_per_wave_hint = 0) {
    const int tiles = (p.L + 31) / 32;
    int tpw = 1;
    if (tiles_per_wave_hint > 0) tpw = tiles_per_wave_hint;
    else while (tpw < 4 && (long long)p.Bc * p.H * ((tiles + 4 * tpw - 1) / (4 * tpw)) > 2048) tpw *= 2;
    p.tiles_per_wave = tpw;
    p.nchunks = (tiles + 4 * tpw - 1) / (4 * tpw);
    p.npart = (p.Bc / p.n_groups) * p.H * p.nchunks;
    {
        const double n = (double)(p.Bc / p.n_groups) * p.H * (double)p.L * p.S;
        p.inv_n = 1.0 / n;
        p.inv_nm1 = 1.0 / (n - 1.0);                          // (a single score: inf, 0 * inf = nan - what torch.std gives)
    }
    p.xcd_map = ((p.Bc * p.nchunks) % 8 == 0) ? 1 : 0;
    p.fd_nchunks = make_fastdiv(p.nchunks, (long long)p.Bc * p.nchunks);
    p.fd_ngroups = make_fastdiv(p.n_groups, p.Bc);
    p.fd_rep = make_fastdiv(((long long)p.Bc * p.H) / (p.Bw > 0 ? p.Bw : 1), (long long)p.Bc * p.H);
}

}  // namespace dsc_xattn
