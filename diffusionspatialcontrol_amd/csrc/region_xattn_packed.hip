// Region cross-attention, "prepared operands" path (dsc_xattn_kv_pack, dsc_region_xattn_fwd_packed).
//
// Two facts of the denoising loop make this the fast path (generic operands: region_xattn.hip):
//   * the text keys / values of a cross-attention layer do not change over the 25 steps, so they are packed ONCE per
//     generation into the exact register image the MFMAs consume: K as A-fragments [m][ks][lane][8], V^T as A-fragments
//     in the permuted k order of the chained product [dm][tt][lane][8], zero padding included.  The forward kernel
//     DMAs that image global -> LDS (global_load_lds_dwordx4: no VGPR staging, no transpose, no zero fill) and reads
//     every fragment back with one conflict-free ds_read_b128 (lane-linear);
//   * the region table has only a handful of DISTINCT rows (<= 2^regions; the reference rasterises a few masks,
//     encode_region_map_function.py:49-69): it is passed as uint16 row ids [Bw, L] + the distinct rows [NU, S].  The
//     workgroup folds sigma * std into an LDS table bias[id][s] once, so a score costs one LDS read + one add
//     instead of a 4-byte HBM read and two multiplies, and 4*L*S bytes per row leave the HBM traffic.
// Same arithmetic, rounding points and std-group semantics as the generic kernel (parity-tested against it and the
// oracle).
#include "xattn_shared.h"

using namespace dsc_xattn;

namespace {

constexpr int kNUMax = 32;       // distinct region rows held in LDS
constexpr int kBP = 100;         // bias-table row stride (floats): 16-byte aligned rows, zero padded past S (b128 reads)

struct XpParams {
    XattnParams x;               // q/out/strides/shape/plan; x.k, x.v, x.region unused here
    const half_t* img;           // [Bc*H][IMG halves]
    const unsigned short* ids;   // [Bw, L] or null (no bias)
    const float* rows;           // [NU, S]
    int NU;
};

template <int NK>
struct PCfg {
    static constexpr int DM = (NK + 1) / 2;
    static constexpr int KFR = 3 * NK, VFR = 6 * DM;
    static constexpr int IMG = (KFR + VFR) * 512;            // halves per (b, h)
};

// ---------------------------------------------------------------------------------------------- pack (once per generation)
template <int NK>
__global__ __launch_bounds__(256) void xp_pack(const half_t* k, const half_t* v, half_t* img, int H, int S, int d,
                                               long long ksb, long long kss, long long ksh, long long vsb, long long vss,
                                               long long vsh) {
    using P = PCfg<NK>;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    // key chunk blockIdx.y of gridDim.y (prompts longer than 96 keys: images [bh][chunk]); keys [96 c, min(S, 96 c + 96))
    const int kc = blockIdx.y;
    const half_t* kb = k + b * ksb + h * ksh + (long long)kc * kSMax * kss;
    const half_t* vb = v + b * vsb + h * vsh + (long long)kc * kSMax * vss;
    half_t* dst = img + ((long long)bh * gridDim.y + kc) * P::IMG;
    S = min(S - kc * kSMax, kSMax);
    for (int idx = threadIdx.x; idx < (P::KFR + P::VFR) * 64; idx += 256) {
        const int f = idx >> 6, lane = idx & 63, r = lane & 31, hh = lane >> 5;
        h8_t val = {0, 0, 0, 0, 0, 0, 0, 0};
        if (f < P::KFR) {                                    // K[s = 32 m + r][16 ks + 8 hh + j]
            const int m = f / NK, ks = f - m * NK, s = 32 * m + r, col = 16 * ks + 8 * hh;
            if (s < S && col < d) val = *reinterpret_cast<const h8_t*>(kb + s * kss + col);
        } else {                                             // V[s = 16 tt + 8 (j >> 2) + 4 hh + (j & 3)][32 dm + r]
            const int g = f - P::KFR, dm = g / 6, tt = g - dm * 6, col = 32 * dm + r;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int s = 16 * tt + 8 * (j >> 2) + 4 * hh + (j & 3);
                if (s < S && col < d) val[j] = vb[s * vss + col];
            }
        }
        *reinterpret_cast<h8_t*>(dst + (long long)idx * 8) = val;
    }
}

// global -> LDS DMA of `pieces` 1-KiB pieces (one wave-instruction each), spread over the 4 waves
__device__ __forceinline__ void dma_image(const half_t* src, char* lds, int pieces, int wave, int lane) {
    const char* s = reinterpret_cast<const char*>(src) + lane * 16;
    for (int pc = wave; pc < pieces; pc += kWaves)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + pc * 1024),
                                         (__attribute__((address_space(3))) void*)(lds + pc * 1024), 16, 0, 0);
}

// global -> registers -> LDS copy of FR 1-KiB fragments by the whole workgroup: thread t moves the 16-byte chunks t, t + 256, ...
// Two halves so that every load of the prologue is in flight before anything waits.  (Round 2 moved the image by LDS-DMA: no
// register staging, but an LDS-DMA instruction holds its wave's issue port for 100-200 cycles - 5-6 of them per wave were
// ~1300 cycles of the prologue before the first byte was even waited for; a 16-byte register load issues in a few cycles and
// the ds_write_b128 behind it in ~13.)
template <int FR>
struct ImgCopy {
    static constexpr int NCH = (FR * 64 + kThreads - 1) / kThreads;
    h8_t st[NCH];
    __device__ __forceinline__ void load(const half_t* src) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int idx = threadIdx.x + c * kThreads;
            if (NCH * kThreads == FR * 64 || idx < FR * 64) st[c] = *reinterpret_cast<const h8_t*>(src + (long long)idx * 8);
        }
    }
    __device__ __forceinline__ void store(half_t* lds) const {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int idx = threadIdx.x + c * kThreads;
            if (NCH * kThreads == FR * 64 || idx < FR * 64) *reinterpret_cast<h8_t*>(lds + idx * 8) = st[c];
        }
    }
};

// K fragments of one (b, h) straight from the packed image into registers (statistics pass: nothing else of the image is
// needed, and four waves reading 9-30 KiB each from L2 is cheaper than staging + a workgroup barrier)
template <int NK>
__device__ __forceinline__ void load_k_frags(const half_t* img, h8_t (&kf)[3 * NK], int lane) {
#pragma unroll
    for (int f = 0; f < 3 * NK; ++f) kf[f] = *reinterpret_cast<const h8_t*>(img + ((long long)f * 64 + lane) * 8);
}

template <int NK, bool REF16>
__device__ __forceinline__ void scores_reg(const h8_t (&kf)[3 * NK], const h8_t (&qf)[NK], f16x_t (&acc)[3], float scale) {
#pragma unroll
    for (int m = 0; m < 3; ++m) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) acc[m] = mfma_32x32x16(kf[m * NK + ks], qf[ks], acc[m]);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (REF16) acc[m][i] = round_f16(pin_f32(round_f16(acc[m][i]) * scale));   // attention_modify.py:90
            else acc[m][i] = acc[m][i] * scale;
        }
    }
}

template <int NK, bool REF16, bool RAW = false>
__device__ __forceinline__ void scores_img(const XattnParams& p, const half_t* img, const h8_t (&qf)[NK], f16x_t (&acc)[3],
                                           int lane, float scale) {
    // all three 32-key tiles, no branch on the key count: the image holds ZERO fragments beyond S (xp_pack), so a tile past the
    // last key costs three MFMAs on zeros and its scores are masked anyway - while a wave-uniform `if (m < mt)` around each
    // tile made every fragment read wait for its own s_waitcnt in front of its MFMA (one LDS latency exposed per MFMA)
#pragma unroll
    for (int m = 0; m < 3; ++m) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const h8_t kf = *reinterpret_cast<const h8_t*>(img + ((m * NK + ks) * 64 + lane) * 8);
            acc[m] = mfma_32x32x16(kf, qf[ks], acc[m]);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (REF16) acc[m][i] = round_f16(pin_f32(round_f16(acc[m][i]) * scale));   // attention_modify.py:90
            else if (!RAW) acc[m][i] = acc[m][i] * scale;
        }
    }
}

// ---------------------------------------------------------------------------------------------- statistics
// K fragments: straight into registers while they fit beside two waves per SIMD (NK <= 5: d <= 80, 9-15 fragments); the wide
// heads (d = 160: 30 fragments = 120 registers, and their launches have few workgroups: four waves each pulling 30 KiB through
// L2 took 10.1 us against 6.6) stage the image once per workgroup in LDS
template <int NK, bool REF16>
__global__ __launch_bounds__(kThreads, 2) void xp_stats(XpParams pp) {   // 2 waves per SIMD: score accumulators stay in VGPRs
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using P = PCfg<NK>;
    constexpr bool KREG = NK <= 5;
    const XattnParams& p = pp.x;
    half_t* img = reinterpret_cast<half_t*>(smem);
    double* red = reinterpret_cast<double*>(smem + (KREG ? 0 : P::KFR * 1024));
    int b, h, chunk;
    block_to_work(p, b, h, chunk);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, hh = lane >> 5;
    h8_t qf[NK], kf[KREG ? 3 * NK : 1];
    int l0 = (chunk * kWaves * p.tiles_per_wave + wave) * 32;
    load_q_frags<NK>(p, qf, b, h, min(l0 + r, p.L - 1), hh);
    if constexpr (KREG) {
        load_k_frags<NK>(pp.img + (long long)(b * p.H + h) * P::IMG, kf, lane);
    } else {
        ImgCopy<P::KFR> imgc;
        imgc.load(pp.img + (long long)(b * p.H + h) * P::IMG);
        imgc.store(img);
        __syncthreads();
    }
    double d1 = 0.0, d2 = 0.0;
    for (int t = 0; t < p.tiles_per_wave; ++t) {
        l0 = (chunk * kWaves * p.tiles_per_wave + t * kWaves + wave) * 32;
        if (l0 >= p.L) break;
        if (t > 0) load_q_frags<NK>(p, qf, b, h, min(l0 + r, p.L - 1), hh);
        const bool row_ok = l0 + r < p.L;
        f16x_t acc[3];
        if constexpr (KREG) scores_reg<NK, REF16>(kf, qf, acc, p.scale);
        else scores_img<NK, REF16>(p, img, qf, acc, lane, p.scale);
        // keys >= S count as 0 (only the 32-key tile S cuts needs a select: scalar lane masks, xattn_shared.h), and a lane whose
        // query row is beyond L drops its sums as a whole - same additions in the same order as a select per element
        float s1 = 0.f, s2 = 0.f;
        const int S_now = dsc_xattn::opaque_s(p.S);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            if (32 * m + 32 <= S_now) {
#pragma unroll
                for (int i = 0; i < 16; ++i) { const float a = acc[m][i]; s1 += a; s2 += a * a; }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float a = dsc_xattn::keep_or_zero(acc[m][i], dsc_xattn::key_keep_mask(32 * m + (i & 3) + 8 * (i >> 2), S_now));
                    s1 += a;
                    s2 += a * a;
                }
            }
        }
        if (!row_ok) { s1 = 0.f; s2 = 0.f; }
        d1 += (double)s1; d2 += (double)s2;
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

// diagnostic stamps (flag 32): shader-clock and 100 MHz real-time stamps of workgroup 0 go to x.std_out (never an
// output tensor); no stamp executes in a normal call
#define XP_STAMP(n)                                                                                          \
    if constexpr (STAMPS) if (dbg) {                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(), r_ = __builtin_amdgcn_s_memrealtime();   \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                  \
        if (lane == 0) { dbgp[2 * (n)] = t_; dbgp[2 * (n) + 1] = r_; }                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    }

// ---------------------------------------------------------------------------------------------- forward
// STAMPS: the diagnostic instantiation (debug flag 32, tools/stamps_xattn.py); the stamps' scheduling barriers and branches cut
// the kernel into blocks the compiler cannot overlap, so the production instantiation carries none of them
template <int NK, bool REF16, bool STAMPS = false>
__global__ __launch_bounds__(kThreads, (NK <= 5 ? 2 : 1)) void xp_fwd(XpParams pp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using P = PCfg<NK>;
    constexpr int DM = P::DM;
    const XattnParams& p = pp.x;
    half_t* img = reinterpret_cast<half_t*>(smem);
    float* biasT = reinterpret_cast<float*>(smem + P::IMG * 2);
    double* red = reinterpret_cast<double*>(biasT + kNUMax * kBP);
    int b, h, chunk;
    block_to_work(p, b, h, chunk);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, hh = lane >> 5;
    const bool has_bias = pp.ids != nullptr;
    const int bw = has_bias ? fdiv(b * p.H + h, p.fd_rep) : 0;   // repeat_interleave, :96-99
    const bool dbg = STAMPS && (p.flags & 32u) && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && p.std_out != nullptr;
    unsigned long long* dbgp = reinterpret_cast<unsigned long long*>(p.std_out) + wave * 64;
    (void)dbg; (void)dbgp;
    XP_STAMP(0)

    // ---- prologue: image DMA, first tile's Q / row id, std partials - all in flight together
    // The statistics partials first: they were written by the OTHER XCDs' workgroups of the launch before this one, i.e. they
    // come from beyond this XCD's L2 - the longest latency of the prologue, so it starts first and everything else hides under it
    PartialLoads pl;
    float sd = 1.f, sig = 1.f;
    if (has_bias) {
        pl.issue(p, b - fdiv(b, p.fd_ngroups) * p.n_groups);
        sig = p.sigma_dev ? *p.sigma_dev : p.sigma_host;
    }
    ImgCopy<P::KFR + P::VFR> imgc;
    imgc.load(pp.img + (long long)(b * p.H + h) * P::IMG);
    h8_t qf[NK];
    int l0 = (chunk * kWaves * p.tiles_per_wave + wave) * 32;
    bool tile_ok = l0 < p.L;
    int row = tile_ok ? min(l0 + r, p.L - 1) : 0;
    int id = 0;
    if (tile_ok) {
        load_q_frags<NK>(p, qf, b, h, row, hh);
        if (has_bias) id = pp.ids[(long long)bw * p.L + row];
    }
    // distinct rows -> LDS.  DSC_FLAG_ROWS_PADDED (what the pipeline passes): the caller's table already has the LDS table's
    // shape - [NU][kBP] fp32, 16-byte aligned - so it is a flat copy of 16-byte chunks, at most four per thread.  Otherwise
    // ([NU][S], S odd in general): thread t owns key column t % 128 of rows t / 128, t / 128 + 2, ... - ONE lane mask (s < S)
    // for all sixteen loads and a wave-uniform row test.  Columns [S, kBP) may stay unwritten: a score that reads them is
    // beyond the last key and is replaced by -inf with a select, never used in arithmetic.
    constexpr int kRowRegs = kNUMax / 2;
    constexpr int kRowVec = (kNUMax * kBP / 4 + kThreads - 1) / kThreads;      // 4
    if (has_bias) {
        const bool padded = (p.flags & DSC_FLAG_ROWS_PADDED) != 0;
        float rowv[kRowRegs];
        f4x_t rvec[kRowVec];
        const int bs = threadIdx.x & 127, bu = threadIdx.x >> 7;
        if (padded) {
#pragma unroll
            for (int c = 0; c < kRowVec; ++c) {
                const int idx = threadIdx.x + c * kThreads;
                if (idx * 4 < pp.NU * kBP) rvec[c] = reinterpret_cast<const f4x_t*>(pp.rows)[idx];
            }
        } else {
#pragma unroll
            for (int c = 0; c < kRowRegs; ++c) {
                const int u = bu + 2 * c;
                rowv[c] = 0.f;
                if (u < pp.NU && bs < p.S) rowv[c] = pp.rows[u * p.S + bs];
            }
        }
        XP_STAMP(7)
        XP_STAMP(8)
        if (padded) {
#pragma unroll
            for (int c = 0; c < kRowVec; ++c) {
                const int idx = threadIdx.x + c * kThreads;
                if (idx * 4 < pp.NU * kBP) reinterpret_cast<f4x_t*>(biasT)[idx] = rvec[c];
            }
        } else {
#pragma unroll
            for (int c = 0; c < kRowRegs; ++c) {
                const int u = bu + 2 * c;
                if (u < pp.NU && bs < p.S) biasT[u * kBP + bs] = rowv[c];
            }
        }
        double a1, a2;
        pl.finish(p, a1, a2);
        group_std_stage(a1, a2, red);
        XP_STAMP(9)
    }
    XP_STAMP(1)
    imgc.store(img);                                         // (waits for this thread's chunks)
    XP_STAMP(2)
    __syncthreads();                                         // the image, the raw rows and the partial pairs are in LDS
    XP_STAMP(3)
    if (has_bias) sd = group_std_reduce(p, red, REF16);      // per wave, no barrier: overlaps the first tile's score MFMAs

    for (int t = 0; t < p.tiles_per_wave; ++t) {
        // prefetch the next tile's Q fragments / row id behind this tile's MFMAs
        h8_t qn[NK];
        int idn = 0, l0n = 0, rown = 0;
        bool okn = false;
        if (t + 1 < p.tiles_per_wave) {
            l0n = (chunk * kWaves * p.tiles_per_wave + (t + 1) * kWaves + wave) * 32;
            okn = l0n < p.L;
            rown = okn ? min(l0n + r, p.L - 1) : 0;
            if (okn) {
                load_q_frags<NK>(p, qn, b, h, rown, hh);
                if (has_bias) idn = pp.ids[(long long)bw * p.L + rown];
            }
        }
        if (tile_ok) {
            f16x_t acc[3];
            scores_img<NK, REF16, !REF16>(p, img, qf, acc, lane, p.scale);
            XP_STAMP(4)
            h8_t pf[6];
            float oscale = 1.f;
            const float* brow = has_bias ? biasT + id * kBP : nullptr;
            if (REF16) softmax_tile<REF16, true>(acc, pf, brow, sig, sd, p.S, hh);          // bias = (w * sigma) * std, fp32 (app.py:1004)
            else oscale = softmax_tile_lean<true>(acc, pf, brow, sig, sd, p.scale * 1.4426950408889634f, p.S, hh);

            XP_STAMP(5)
            half_t* ob = p.out + b * p.osb + h * p.osh + (long long)row * p.osl;
            const bool row_ok = l0 + r < p.L;
            // every channel tile (32 dm < d holds for the NK that pick_nk gives d) and all six 16-key steps, no branch: V^T
            // fragments beyond S are zeros in the image and so are the probabilities of masked keys
#pragma unroll
            for (int dm = 0; dm < DM; ++dm) {
                f16x_t o;
#pragma unroll
                for (int i = 0; i < 16; ++i) o[i] = 0.f;
#pragma unroll
                for (int tt = 0; tt < 6; ++tt) {
                    const h8_t vf = *reinterpret_cast<const h8_t*>(img + ((P::KFR + dm * 6 + tt) * 64 + lane) * 8);
                    o = mfma_32x32x16(vf, pf[tt], o);
                }
                store_o_block(ob, o, oscale, dm, hh, p.d, row_ok, p.wide_store != 0);
            }
        }
        XP_STAMP(6)
        if (t + 1 < p.tiles_per_wave) {
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) qf[ks] = qn[ks];
            id = idn; l0 = l0n; row = rown; tile_ok = okn;
        }
    }
}


// ---------------------------------------------------------------------------------------------- long prompts (S > 96)
// Prompts of several 77-token chunks (A1111-style / lpw-style long prompts: S = 154, 231, ...) keep the same two-phase
// method: the key axis is cut into chunks of 96 keys, each with its own packed image [bh][chunk]; the statistics kernel
// sums over all chunks, the forward kernel walks the chunks with an ONLINE softmax (running max / sum per query row, the
// output accumulators rescaled when the max grows) - scores stay fp32 (no fp16-rounding emulation on this path).
// One 32-row tile per wave (tiles_per_wave = 1); the image of a chunk is DMA'd into one LDS buffer per chunk step.
constexpr int kChunksMax = 4;    // S <= 384

template <int NK>
__global__ __launch_bounds__(kThreads, 2) void xp_stats_long(XpParams pp, int nkc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using P = PCfg<NK>;
    const XattnParams& p = pp.x;
    half_t* img = reinterpret_cast<half_t*>(smem);
    double* red = reinterpret_cast<double*>(smem + P::KFR * 1024);
    int b, h, chunk;
    block_to_work(p, b, h, chunk);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, hh = lane >> 5;
    const int l0 = (chunk * kWaves + wave) * 32;
    const bool tile_ok = l0 < p.L, row_ok = l0 + r < p.L;
    h8_t qf[NK];
    load_q_frags<NK>(p, qf, b, h, min(l0 + r, p.L - 1), hh);
    double d1 = 0.0, d2 = 0.0;
    for (int kc = 0; kc < nkc; ++kc) {
        if (kc > 0) __syncthreads();                         // every wave is done reading the previous chunk's image
        dma_image(pp.img + ((long long)(b * p.H + h) * nkc + kc) * P::IMG, smem, P::KFR, wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tile_ok) {
            const int valid = min(p.S - kc * kSMax, kSMax);
            XattnParams pc = p;
            pc.S = valid;
            f16x_t acc[3];
            scores_img<NK, false>(pc, img, qf, acc, lane, p.scale);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int s_ = 32 * m + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    const float a = (row_ok && s_ < valid) ? acc[m][i] : 0.f;
                    s1 += a;
                    s2 += a * a;
                }
            d1 += (double)s1; d2 += (double)s2;
        }
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

// one chunk of the online softmax: acc holds the raw scores of <= 96 keys (`valid` of them real); m_run / l_run the running
// row maximum (log2 domain) and sum; returns the factor the output accumulators are rescaled by; pf = exp2(a - m_new)
__device__ __forceinline__ float softmax_chunk_online(f16x_t (&acc)[3], h8_t (&pf)[6], const float* brow, float scale_log2e,
                                                      int valid, int hh, float& m_run, float& l_run) {
    constexpr float kLog2e = 1.4426950408889634f;
    float mx = -INFINITY;
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float bias[4] = {0.f, 0.f, 0.f, 0.f};
            if (brow) bias4<true>(brow, 32 * m + 8 * g + 4 * hh, valid - 1, bias);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = 4 * g + j, s_ = 32 * m + 8 * g + 4 * hh + j;
                float a = acc[m][i] * scale_log2e;
                if (brow) a = fmaf(bias[j], kLog2e, a);
                a = s_ < valid ? a : -INFINITY;
                acc[m][i] = a;
                mx = fmaxf(mx, a);
            }
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);                    // finite: every chunk holds at least one real key
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // first chunk: exp2(-inf) = 0
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float e = __builtin_amdgcn_exp2f(acc[m][i] - m_new);
            acc[m][i] = e;
            sum += e;
        }
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) pf[2 * m + (i >> 3)][i & 7] = (half_t)acc[m][i];
    sum += __shfl_xor(sum, 32, 64);
    l_run = l_run * alpha + sum;
    m_run = m_new;
    return alpha;
}

template <int NK>
__global__ __launch_bounds__(kThreads, (NK <= 5 ? 2 : 1)) void xp_fwd_long(XpParams pp, int nkc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using P = PCfg<NK>;
    constexpr int DM = P::DM;
    const XattnParams& p = pp.x;
    half_t* img = reinterpret_cast<half_t*>(smem);
    float* biasT = reinterpret_cast<float*>(smem + P::IMG * 2);          // [NU][nkc * kBP]
    double* red = reinterpret_cast<double*>(biasT + kNUMax * kChunksMax * kBP);
    int b, h, chunk;
    block_to_work(p, b, h, chunk);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, hh = lane >> 5;
    const bool has_bias = pp.ids != nullptr;
    const int bw = has_bias ? fdiv(b * p.H + h, p.fd_rep) : 0;
    const int l0 = (chunk * kWaves + wave) * 32;
    const bool tile_ok = l0 < p.L, row_ok = l0 + r < p.L;
    const int row = tile_ok ? min(l0 + r, p.L - 1) : 0;
    const int rs = nkc * kBP;                                             // bias-table row stride (floats)
    h8_t qf[NK];
    int id = 0;
    if (tile_ok) {
        load_q_frags<NK>(p, qf, b, h, row, hh);
        if (has_bias) id = pp.ids[(long long)bw * p.L + row];
    }
    if (has_bias) {
        const float sig = p.sigma_dev ? *p.sigma_dev : p.sigma_host;
        const float sd = group_std(p, b - fdiv(b, p.fd_ngroups) * p.n_groups, red, false);        // contains __syncthreads()
        for (int idx = threadIdx.x; idx < pp.NU * rs; idx += kThreads) {  // w * sigma * std, zero padded past each chunk's keys
            const int u = idx / rs, rem = idx - u * rs, kc = rem / kBP, s_ = rem - kc * kBP;
            const int key = kc * kSMax + s_;
            biasT[idx] = (s_ < kSMax && key < p.S) ? (pp.rows[(long long)u * p.S + key] * sig) * sd : 0.f;
        }
    }
    f16x_t o[DM];
#pragma unroll
    for (int dm = 0; dm < DM; ++dm)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dm][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    for (int kc = 0; kc < nkc; ++kc) {
        if (kc > 0) __syncthreads();                         // the previous chunk's fragments are all in registers / consumed
        dma_image(pp.img + ((long long)(b * p.H + h) * nkc + kc) * P::IMG, smem, P::KFR + P::VFR, wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                     // image landed; (first pass) bias table complete
        if (!tile_ok) continue;
        const int valid = min(p.S - kc * kSMax, kSMax);
        XattnParams pc = p;
        pc.S = valid;
        f16x_t acc[3];
        scores_img<NK, false, true>(pc, img, qf, acc, lane, p.scale);
        h8_t pf[6];
        const float* brow = has_bias ? biasT + id * rs + kc * kBP : nullptr;
        const float alpha = softmax_chunk_online(acc, pf, brow, p.scale * 1.4426950408889634f, valid, hh, m_run, l_run);
        const int nt = (valid + 15) >> 4;
#pragma unroll
        for (int dm = 0; dm < DM; ++dm) {
            if (32 * dm < p.d) {
#pragma unroll
                for (int i = 0; i < 16; ++i) o[dm][i] *= alpha;
#pragma unroll
                for (int tt = 0; tt < 6; ++tt) {
                    if (tt < nt) {
                        const h8_t vf = *reinterpret_cast<const h8_t*>(img + ((P::KFR + dm * 6 + tt) * 64 + lane) * 8);
                        o[dm] = mfma_32x32x16(vf, pf[tt], o[dm]);
                    }
                }
            }
        }
    }
    if (tile_ok && row_ok) {
        const float inv = 1.f / l_run;
        half_t* ob = p.out + b * p.osb + h * p.osh + (long long)row * p.osl;
#pragma unroll
        for (int dm = 0; dm < DM; ++dm)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dd0 = 32 * dm + 8 * g4 + 4 * hh;
                if (dd0 < p.d) {
                    const h4_t ov = {(half_t)(o[dm][4 * g4] * inv), (half_t)(o[dm][4 * g4 + 1] * inv),
                                     (half_t)(o[dm][4 * g4 + 2] * inv), (half_t)(o[dm][4 * g4 + 3] * inv)};
                    *reinterpret_cast<h4_t*>(ob + dd0) = ov;
                }
            }
    }
}

template <int NK>
int launch_packed_long(const XpParams& pp, int nkc, bool need_stats, hipStream_t st) {
    using P = PCfg<NK>;
    const dim3 grid = xattn_grid(pp.x), block(kThreads);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&xp_fwd_long<NK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&xp_stats_long<NK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (need_stats) DSC_LAUNCH((xp_stats_long<NK>), grid, block, (size_t)P::KFR * 1024 + kRedBytes, st, pp, nkc);
    const size_t lds = (size_t)P::IMG * 2 + (size_t)kNUMax * kChunksMax * kBP * 4 + kRedBytes;
    DSC_LAUNCH((xp_fwd_long<NK>), grid, block, lds, st, pp, nkc);
    return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
}

int pick_nk(int d) {
    static const int opts[] = {2, 3, 4, 5, 6, 8, 10};
    for (int nk : opts) if (16 * nk >= d) return nk;
    return 0;
}

template <int NK>
size_t img_bytes() { return (size_t)PCfg<NK>::IMG * sizeof(half_t); }

size_t img_bytes_nk(int nk) {
    switch (nk) {
        case 2: return img_bytes<2>(); case 3: return img_bytes<3>(); case 4: return img_bytes<4>();
        case 5: return img_bytes<5>(); case 6: return img_bytes<6>(); case 8: return img_bytes<8>();
        case 10: return img_bytes<10>(); default: return 0;
    }
}

template <int NK>
int launch_pack(const half_t* k, const half_t* v, half_t* img, int Bc, int H, int S, int d, const int64_t* ks,
                const int64_t* vs, hipStream_t st) {
    DSC_LAUNCH(xp_pack<NK>, dim3(Bc * H, (S + kSMax - 1) / kSMax), dim3(256), 0, st, k, v, img, H, S, d, (long long)ks[0], (long long)ks[1],
                       (long long)ks[2], (long long)vs[0], (long long)vs[1], (long long)vs[2]);
    return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
}

template <int NK, bool REF16>
int launch_packed(const XpParams& pp, bool need_stats, hipStream_t st) {
    using P = PCfg<NK>;
    const dim3 grid = xattn_grid(pp.x), block(kThreads);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&xp_fwd<NK, REF16>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&xp_stats<NK, REF16>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (need_stats)
        DSC_LAUNCH((xp_stats<NK, REF16>), grid, block, (size_t)(NK <= 5 ? 0 : P::KFR * 1024) + kRedBytes, st, pp);
    const size_t lds = (size_t)P::IMG * 2 + (size_t)kNUMax * kBP * 4 + kRedBytes;
    if constexpr (!REF16 && (NK == 3 || NK == 10)) {           // the diagnostic instantiation (tools/stamps_xattn.py's two shapes)
        if (pp.x.flags & 32u) {
            static bool sattr = false;
            if (!sattr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&xp_fwd<NK, REF16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); sattr = true; }
            DSC_LAUNCH((xp_fwd<NK, REF16, true>), grid, block, lds, st, pp);
            return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
        }
    }
    DSC_LAUNCH((xp_fwd<NK, REF16>), grid, block, lds, st, pp);
    return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
}

#define COMMA_TRUE , true
#define COMMA_FALSE , false
#define DSC_NK_SWITCH(nk, EXPR_PREFIX, EXPR_SUFFIX)                    \
    switch (nk) {                                                      \
        case 2: return EXPR_PREFIX<2 EXPR_SUFFIX;                      \
        case 3: return EXPR_PREFIX<3 EXPR_SUFFIX;                      \
        case 4: return EXPR_PREFIX<4 EXPR_SUFFIX;                      \
        case 5: return EXPR_PREFIX<5 EXPR_SUFFIX;                      \
        case 6: return EXPR_PREFIX<6 EXPR_SUFFIX;                      \
        case 8: return EXPR_PREFIX<8 EXPR_SUFFIX;                      \
        case 10: return EXPR_PREFIX<10 EXPR_SUFFIX;                    \
        default: return DSC_ERR_UNSUPPORTED;                           \
    }

int dispatch_pack(int nk, const half_t* k, const half_t* v, half_t* img, int Bc, int H, int S, int d, const int64_t* ks,
                  const int64_t* vs, hipStream_t st) {
    DSC_NK_SWITCH(nk, launch_pack, >(k, v, img, Bc, H, S, d, ks, vs, st))
}
int dispatch_packed_long(int nk, const XpParams& pp, int nkc, bool need_stats, hipStream_t st) {
    DSC_NK_SWITCH(nk, launch_packed_long, >(pp, nkc, need_stats, st))
}
int dispatch_packed(int nk, bool ref16, const XpParams& pp, bool need_stats, hipStream_t st) {
    if (ref16) { DSC_NK_SWITCH(nk, launch_packed, COMMA_TRUE>(pp, need_stats, st)) }
    DSC_NK_SWITCH(nk, launch_packed, COMMA_FALSE>(pp, need_stats, st))
}

bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
bool strides_ok(const int64_t s[3]) { return s[0] % 8 == 0 && s[1] % 8 == 0 && s[2] % 8 == 0; }

}  // namespace

static void* g_debug_stamps = nullptr;
extern "C" void dsc_debug_set_stamp_buffer(void* device_buffer_2KiB) { g_debug_stamps = device_buffer_2KiB; }

extern "C" size_t dsc_xattn_kv_pack_bytes(int Bc, int H, int S, int d) {
    if (Bc <= 0 || H <= 0 || S <= 0 || S > kSMax * kChunksMax || d <= 0 || d % 8 != 0 || d > 160) return 0;
    return (size_t)Bc * H * ((S + kSMax - 1) / kSMax) * img_bytes_nk(pick_nk(d));
}

extern "C" int dsc_xattn_kv_pack(const void* k, const void* v, void* packed, int Bc, int H, int S, int d,
                                 const int64_t k_strides[3], const int64_t v_strides[3], int dtype, void* stream) {
    if (!k || !v || !packed || !k_strides || !v_strides || Bc <= 0 || H <= 0 || S <= 0 || d <= 0) return DSC_ERR_BAD_ARG;
    if (dtype != DSC_F16 || d % 8 != 0 || d > 160 || S > kSMax * kChunksMax) return DSC_ERR_UNSUPPORTED;
    if (!aligned16(k) || !aligned16(v) || !aligned16(packed) || !strides_ok(k_strides) || !strides_ok(v_strides))
        return DSC_ERR_UNSUPPORTED;
    return dispatch_pack(pick_nk(d), static_cast<const half_t*>(k), static_cast<const half_t*>(v),
                         static_cast<half_t*>(packed), Bc, H, S, d, k_strides, v_strides, static_cast<hipStream_t>(stream));
}

extern "C" int dsc_region_xattn_fwd_packed(const void* q, const void* packed_kv, void* out,
                                           const uint16_t* region_ids, const float* region_rows, int n_rows,
                                           int Bc, int H, int L, int S, int d, int Bw, int n_std_groups,
                                           const int64_t q_strides[3], const int64_t o_strides[3],
                                           float sigma_host, const float* sigma_dev, float scale, int dtype,
                                           unsigned flags, void* workspace, size_t workspace_bytes, void* stream) {
    if (!q || !packed_kv || !out || !q_strides || !o_strides) return DSC_ERR_BAD_ARG;
    if (Bc <= 0 || H <= 0 || L <= 0 || S <= 0 || d <= 0 || n_std_groups <= 0 || Bc % n_std_groups != 0) return DSC_ERR_BAD_ARG;
    if (dtype != DSC_F16 || d % 8 != 0 || d > 160 || S > kSMax * kChunksMax) return DSC_ERR_UNSUPPORTED;
    const int nkc = (S + kSMax - 1) / kSMax;                 // key chunks: > 1 = the long-prompt kernels (fp32 scores only)
    if (nkc > 1 && (flags & DSC_FLAG_REF_FP16_ROUNDING)) return DSC_ERR_UNSUPPORTED;
    if (!aligned16(q) || !aligned16(packed_kv) || (reinterpret_cast<uintptr_t>(out) & 7) || !strides_ok(q_strides) ||
        !strides_ok(o_strides))
        return DSC_ERR_UNSUPPORTED;
    const bool has_bias = region_ids != nullptr;
    if (has_bias) {
        if (!region_rows || n_rows <= 0 || Bw <= 0 || (Bc * H) % Bw != 0) return DSC_ERR_BAD_ARG;
        if (n_rows > kNUMax) return DSC_ERR_UNSUPPORTED;
        if ((flags & DSC_FLAG_ROWS_PADDED) && (nkc > 1 || !aligned16(region_rows))) return DSC_ERR_UNSUPPORTED;
    }
    XpParams pp{};
    XattnParams& p = pp.x;
    p.q = static_cast<const half_t*>(q); p.out = static_cast<half_t*>(out);
    p.sigma_dev = sigma_dev; p.sigma_host = sigma_host;
    p.scale = scale > 0.f ? scale : 1.0f / sqrtf((float)d);
    p.Bc = Bc; p.H = H; p.L = L; p.S = S; p.d = d; p.Bw = has_bias ? Bw : 1; p.n_groups = n_std_groups;
    p.qsb = q_strides[0]; p.qsl = q_strides[1]; p.qsh = q_strides[2];
    p.osb = o_strides[0]; p.osl = o_strides[1]; p.osh = o_strides[2];
    p.flags = flags;
    plan_tiles(p, nkc > 1 ? 1 : 0);
    p.std_out = static_cast<float*>(g_debug_stamps);
    pp.img = static_cast<const half_t*>(packed_kv);
    pp.ids = region_ids; pp.rows = region_rows; pp.NU = has_bias ? n_rows : 0;
    p.wide_store = (aligned16(out) && o_strides[0] % 8 == 0 && o_strides[1] % 8 == 0 && o_strides[2] % 8 == 0 && !(flags & 1024u)) ? 1 : 0;   // (flag 1024: A/B, 8-byte pieces)
    if (has_bias) {
        const size_t need = (size_t)n_std_groups * p.npart * 2 * sizeof(double);
        if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 7)) return DSC_ERR_WORKSPACE;
        p.partials = static_cast<double*>(workspace);
    }
    const bool need_stats = has_bias && !(flags & DSC_FLAG_REUSE_STATS);
    if (nkc > 1) return dispatch_packed_long(pick_nk(d), pp, nkc, need_stats, static_cast<hipStream_t>(stream));
    return dispatch_packed(pick_nk(d), (flags & DSC_FLAG_REF_FP16_ROUNDING) != 0, pp, need_stats,
                           static_cast<hipStream_t>(stream));
}
