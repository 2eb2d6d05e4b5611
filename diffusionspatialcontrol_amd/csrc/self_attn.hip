// Flash self-attention forward for gfx950 (MI355X): the kernel behind dsc_self_attn_fwd.
//
// Replaces `F.scaled_dot_product_attention(query, key, value)` on the self-attention branch of the processors
// (source/modules/attention_modify.py:483-485) for the UNet's head dims 40 / 64 / 80 / 160.
// The L x L score matrix is never materialised: per 32-query-row wave, KV tiles of 64 keys stream through LDS with an
// online softmax (running max per query row, lazily updated).
//
// MI355X mapping (64-wide waves, MFMA 32x32x16 f16):
//   * K / V tiles go HBM/L2 -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`: 1 KiB per wave instruction, NO register
//     destination - nothing for the compiler to move, copy or spill between issue and wait; the first version staged
//     through registers with asm loads whose destinations the compiler could not see and faulted at 3 waves per SIMD).
//     The LDS image is lane-linear, so the padded row layout is made on the SOURCE side: lane l of piece j fetches the
//     16-byte chunk that belongs at LDS chunk 64 j + l; pad chunks and keys beyond S carry an out-of-range buffer
//     offset and the hardware deposits zeros.  Tile t+1 is in flight while tile t is multiplied: one `s_waitcnt
//     vmcnt(0)` + ONE raw s_barrier per tile (it publishes tile t and proves every wave has left tile t-1's buffer).
//   * S^T[kv, q] = K[kv, :] . Q[q, :]  ("swapped" product: A = K rows from LDS, B = Q^T kept in registers for the whole
//     kernel, pre-multiplied by scale * log2 e) puts ONE query row on each lane (q = lane & 31) with half of the tile's
//     64 scores in its registers.  The chain's initial accumulator is -m (the row's running max, 16 registers holding the
//     same value, rewritten only when the max moves): the MFMA result IS  s - m  and  p = exp2(result)  - no scale, no
//     subtract, 32 v_exp_f32 + 16 v_max3_f32 + 16 v_cvt_pk_f16_f32 per lane and tile;
//   * lazy rescale (tau = 8): the running max moves only when some row's scores exceed it by 2^8 - then (rare, wave-
//     uniform branch) the accumulators are rescaled; P <= 256 in fp16, sums in fp32;
//   * O^T[d, q] += V^T[d, kv] . P^T[kv, q]: the fp16-packed probabilities are already the B operand (k order permuted
//     inside each 16-step, cdna_hip_programming.md section 3 "accumulator tile as the next MFMA's operand"); the A operand
//     V^T comes from the ROW-MAJOR V tile in LDS through ds_read_b64_tr_b16 (hardware transpose);
//   * row sums for free where the head dim leaves padding in the last 32-channel tile (d = 40, 80): the lanes that would
//     read V's padding columns 16 NK .. 16 NK + 15 read a constant LDS region [1, 0, ..., 0] instead, so output channel
//     16 NK of the PV product is sum_s p[s] - of the same fp16-rounded p as the numerator;
//   * blockIdx is remapped so that all query blocks of one (b, h) run on one XCD: its K / V (L*d*4 bytes) is fetched
//     from HBM once and then served by that XCD's L2.
// Bound: MFMA (arithmetic intensity 4*L*L*C / 8*L*C = L/2 FLOP/B); head dim 40 pads to 48 (QK^T) and 64 (PV).
#include "dsc_common.h"
#include "dsc_hip.h"

namespace {

typedef short s4_t __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) char lds_char_t;

constexpr int kKV = 64;                          // keys per tile
constexpr unsigned kOob = 0x80000000u;           // buffer offset beyond any supported tensor: the DMA deposits zeros
constexpr float kTau = 8.f;                      // lazy rescale threshold (log2 units)

struct SaParams {
    const half_t* q; const half_t* k; const half_t* v; half_t* out;
    int Bc, H, L, S, d;          // S = number of keys (== L for self-attention)
    int nqb, xcd_map;
    float scale_log2e;
    long long qsb, qsl, qsh, ksb, kss, ksh, vsb, vss, vsh, osb, osl, osh;
    unsigned k_bytes, v_bytes;   // extent of one (b, h) slice of K / V: ((S - 1) * row stride + d) * 2
    unsigned long long* stamps;  // diagnostics; the stamping wave's index rides in the pointer's low three bits (no extra SGPR)
    int wide_store;              // out and its strides are 16-byte aligned: the epilogue writes 16-byte row pieces
};

// D8 = 0: generic image - K rows of 2 NK + 1 chunks (2 NK operand chunks, zero beyond the head dim, + 1 chunk that makes the
// row stride an odd number of 16 B: the 16 rows of a ds_read_b128 lane group then hit 16 distinct 4-bank groups), V rows of
// 2 NK chunks.  D8 > 0 (= d / 8, odd): COMPACT image for exactly that head dim - rows of d8 chunks, no padding at all: an odd
// chunk count is conflict-free by itself, the K operand's columns d .. 16 NK - 1 then read the next row's first halves, which
// only ever multiply Q's zero padding, and the V operand's read the next row too, which only reaches output channels >= d that
// are never stored.  d = 40: 10 DMA pieces per tile instead of 13.
template <int NK, int D8>
struct SaCfg {
    static constexpr int DM = (NK + 1) / 2;
    static constexpr bool ONES = (NK & 1) != 0;             // the last 32-channel PV tile has 16 spare channels at 16 NK
    static constexpr int KC = D8 ? D8 : 2 * NK + 1;         // 16-byte chunks per K row
    static constexpr int VC = D8 ? D8 : 2 * NK;             // chunks per V row
    static constexpr int KP = 8 * KC, VP = 8 * VC;          // row strides (halves)
    static constexpr int K_BYTES = kKV * KP * 2, V_BYTES = kKV * VP * 2;
    static constexpr int TILE_BYTES = K_BYTES + V_BYTES;    // one (K, V) buffer
    static constexpr int ONES_BYTES = ONES ? kKV * VP * 2 + 64 : 0;
    static constexpr int lds_bytes(int nbuf) { return nbuf * TILE_BYTES + ONES_BYTES + 64; }   // + slack: the compact image's over-reads
    static_assert(D8 == 0 || ((D8 & 1) && D8 > 2 * NK - 2 && D8 <= 2 * NK), "compact rows: odd d/8 that needs exactly NK k-steps");
};

__device__ __forceinline__ float max3(float a, float b, float c) {
    // one v_max3_f32; fmaxf() costs a canonicalising v_max_f32 per input.  Scores are finite or -inf, never NaN.
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ h8_t lds_read_h8(unsigned byte_addr) {
    return *reinterpret_cast<const __attribute__((address_space(3))) h8_t*>((uintptr_t)byte_addr);
}
__device__ __forceinline__ h4_t lds_read_tr(unsigned byte_addr) {
    const s4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4_t __attribute__((address_space(3)))*)(uintptr_t)byte_addr);
    return __builtin_bit_cast(h4_t, r);
}
// LDS-DMA of 16 bytes per lane through a buffer descriptor; the 64 lanes' chunks land at lds_byte + 16 * lane
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_byte) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(uintptr_t)lds_byte, 16, voff, soff, 0, 0);
}

unsigned long long* g_sa_stamps = nullptr;
int g_sa_stamp_wave = 0;        // diagnostic: per-segment cycle sums of workgroup 0 / wave 0

#define SA_STAMP(slot)                                                                          \
    if (dbg) {                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                             \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                     \
        seg[slot] += t_ - tprev; tprev = t_;                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    }

// The online softmax of the tile whose S'^T sits in s[2] (element i of s[m] is key 32 m + (i & 3) + 8 (i >> 2) + 4 hh of the lane's
// query row), as two pieces of TEXT: the tile loop has two shapes (plain and rotated, below) and both paste them.  They were
// lambdas for one commit: a lambda around this block - even one the DMA lambda `stage_part` is not called from - kept 10-30 more
// SGPRs live in EVERY instantiation (SGPR spills, a 36-byte scratch frame in 15 kernels; tests/test_cabi.py).
// HEAD: the ragged last tile's mask, the tile maximum, the (rare) rescale of the running state.  T = tile index.
#define SA_SOFTMAX_HEAD(T)                                                                                              \
    {                                                                                                                   \
        const int t_ = (T);                                                                                             \
        const int kv_left = p.S - t_ * kKV;                  /* keys valid in this tile (>= 64 except the last) */       \
        if (kv_left < kKV) {                                 /* wave-uniform: only the ragged last tile masks */         \
            /* element i of s[m] is key c = 32 m + (i & 3) + 8 (i >> 2) in lanes 0-31 and c + 4 in lanes 32-63: the keep mask  */  \
            /* is two scalar compares per element - a per-lane bound and a hoisted register of -inf cost every instantiation */  \
            /* two VGPRs for the whole loop (the rotated kernels spilled at 168)                                             */  \
            float ninf;                                      /* (materialised HERE: hoisted, it is a register for the whole loop) */ \
            asm volatile("v_mov_b32 %0, 0xff800000" : "=v"(ninf));                                                      \
            _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                               \
                _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                        \
                    const int c = 32 * m + (i & 3) + 8 * (i >> 2);                                                      \
                    const unsigned long long keep = (c < kv_left ? 0x00000000FFFFFFFFull : 0ull) |                      \
                                                    (c + 4 < kv_left ? 0xFFFFFFFF00000000ull : 0ull);                   \
                    asm("v_cndmask_b32_e64 %0, %2, %0, %1" : "+v"(s[m][i]) : "s"(keep), "v"(ninf));                     \
                }                                                                                                       \
        }                                                                                                               \
        float mxa = max3(s[0][0], s[0][1], s[0][2]), mxb = max3(s[1][0], s[1][1], s[1][2]);                             \
        _Pragma("unroll") for (int i = 3; i < 15; i += 2) {                                                             \
            mxa = max3(mxa, s[0][i], s[0][i + 1]); mxb = max3(mxb, s[1][i], s[1][i + 1]);                               \
        }                                                                                                               \
        float mx = max3(mxa, mxb, s[0][15]);                                                                            \
        mx = max3(mx, s[1][15], s[1][15]);                                                                              \
        if (t_ == 0 || __any(mx > kTau)) {                   /* wave-uniform and rare after the first tiles */           \
            /* the row's maximum over all 64 keys (the other half of the keys lives in lane ^ 32) */                    \
            const float mrow = fmaxf(mx, __shfl_xor(mx, 32, 64));                                                       \
            const float delta = t_ == 0 ? mrow : fmaxf(mrow, 0.f);         /* the running max only grows */              \
            const float alpha = t_ == 0 ? 1.f : __builtin_amdgcn_exp2f(-delta);                                         \
            l_run *= alpha;                                                                                             \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) negm[i] -= delta;                                            \
            _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                               \
                _Pragma("unroll") for (int i = 0; i < 16; ++i) s[m][i] -= delta;                                        \
            _Pragma("unroll") for (int dm = 0; dm < DM; ++dm)                                                           \
                _Pragma("unroll") for (int i = 0; i < 16; ++i) o[dm][i] *= alpha;                                       \
        }                                                                                                               \
        psum = 0.f;                                                                                                     \
    }
// HALF: the exponentials of keys 32 M .. 32 M + 31 -> pf[2 M], pf[2 M + 1] (and, without the ones column, their sum)
#define SA_SOFTMAX_HALF(M)                                                                                              \
    {                                                                                                                   \
        const int m_ = (M);                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 16; i += 2) {                                                             \
            const float e0 = __builtin_amdgcn_exp2f(s[m_][i]), e1 = __builtin_amdgcn_exp2f(s[m_][i + 1]);               \
            if (!ONES) psum += e0 + e1;                                                                                 \
            const h2_t pk = __builtin_convertvector(f2x_t{e0, e1}, h2_t);          /* v_cvt_pk_f16_f32 */                \
            pf[2 * m_ + (i >> 3)][i & 7] = pk[0];                                                                       \
            pf[2 * m_ + (i >> 3)][(i & 7) + 1] = pk[1];                                                                 \
        }                                                                                                               \
    }

// MINW = minimum waves per SIMD the register allocation must allow (HIP's second launch bound).  With it the compiler
// keeps the S / O accumulators in VGPRs; without it it parks them in AGPRs and copies them back and forth for the softmax.
//
// NLOAD > 0: the workgroup has NLOAD waves more than WAVES, and those do nothing but the LDS-DMA of the K / V tiles (one
// `buffer_load ... lds` per 1-KiB piece; a wave gets ~one piece per 100-200 cycles through: 13 of them cost each of 4 computing
// waves ~450 of ~1900 cycles per tile when they issued their share themselves).  The computing waves then contain no vector
// memory instruction between their Q loads and their output stores; all WAVES + NLOAD waves meet at ONE s_barrier per tile.
//
// STAGGER (8 computing waves + loaders): the two waves of a SIMD come from ONE workgroup and would run the same phase at the
// same time - both in their MFMA chains, then both in their exponentials.  Waves 4-7 therefore run P.V one tile late (its
// V^T fragments and probabilities stay in registers across the barrier), so that while waves 0-3 are in the softmax their SIMD
// partners are in QK^T, and while the partners are in the softmax waves 0-3 are in P.V.
template <int NK, int WAVES, int MINW, int D8, int NLOAD, int STAGGER>
__global__ __launch_bounds__(64 * (WAVES + NLOAD), MINW) void self_attn_fwd(SaParams p) {
    constexpr bool LOADER = NLOAD > 0;
    static_assert(!STAGGER || (LOADER && WAVES == 8 && NK <= 5), "the stagger is for 8 computing waves with loader waves");
    static_assert(STAGGER >= 0 && STAGGER <= 2, "0: none, 1: waves 4-7 run P.V one tile late, 2: and waves 0-3 the softmax");
    using C = SaCfg<NK, D8>;
    constexpr int DM = C::DM, KP = C::KP, VP = C::VP, KC = C::KC, VC = C::VC;
    constexpr bool ONES = C::ONES;
    constexpr int NISSUE = LOADER ? NLOAD : WAVES;           // waves that issue DMA
    constexpr int KPW = (KC + NISSUE - 1) / NISSUE, VPW = (VC + NISSUE - 1) / NISSUE;   // pieces per issuing wave and tile
    constexpr bool HOIST = NK <= 5;                          // all LDS reads of a tile issued before its first MFMA
    // 168-register kernels (three waves on a SIMD) with loader waves: the V^T fragments are read BEHIND the QK^T MFMAs, into
    // the registers the K fragments leave (their latency hides under the softmax); nothing orders LDS reads against DMA in
    // the computing waves of a loader kernel
    constexpr bool V_LATE = HOIST && LOADER && (MINW >= 3 || WAVES + NLOAD > 8);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_char_t*)smem;
    // loader kernels keep a ring of THREE tile buffers: tile t+2 is issued behind the barrier of tile t, so a tile has two
    // tile times to arrive (with two buffers the computing waves waited ~590 of ~1900 cycles per tile at that barrier for the
    // loader's issue + L2 latency chain); the others double-buffer
    constexpr int NBUF = LOADER ? 3 : 2;
    const unsigned ones0 = lds0 + NBUF * C::TILE_BYTES;

    int bh, qb;
    {
        const int bid = blockIdx.x, nbh = p.Bc * p.H;
        if (p.xcd_map) { const int x = bid & 7, j = bid >> 3, nb8 = nbh >> 3; bh = x + 8 * (j % nb8); qb = j / nb8; }
        else { bh = bid % nbh; qb = bid / nbh; }
    }
    const int b = bh / p.H, h = bh % p.H;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, hh = lane >> 5;
    const int q0 = (qb * WAVES + wave) * 32;
    const int d8 = p.d >> 3;
    const int ntiles = (p.S + kKV - 1) / kKV;

    // ---- DMA plan: piece j of the K image = LDS chunks 64 j .. 64 j + 63 (chunk c of row s sits at s * KC + c).  Without a
    // loader wave, wave w issues pieces w, w + WAVES, ...; the per-lane source offsets are tile-invariant (the tile advances
    // in the scalar offset)
    const bool issues_dma = !LOADER || wave >= WAVES;        // wave-uniform
    const int j0 = LOADER ? wave - WAVES : wave, jstep = NISSUE;
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<half_t*>(p.k + b * p.ksb + h * p.ksh), 0, p.k_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<half_t*>(p.v + b * p.vsb + h * p.vsh), 0, p.v_bytes, 0x00020000);
    unsigned kvo[KPW], vvo[VPW];
    int krow[KPW], vrow[VPW];
    if (issues_dma) {
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            const int ci = 64 * (j0 + i * jstep) + lane;
            krow[i] = ci / KC;
            const int col = ci - krow[i] * KC;
            kvo[i] = col < d8 ? (unsigned)(krow[i] * (int)p.kss + col * 8) * 2u : kOob;
        }
#pragma unroll
        for (int i = 0; i < VPW; ++i) {
            const int ci = 64 * (j0 + i * jstep) + lane;
            vrow[i] = ci / VC;
            const int col = ci - vrow[i] * VC;
            vvo[i] = col < d8 ? (unsigned)(vrow[i] * (int)p.vss + col * 8) * 2u : kOob;
        }
    }
    const unsigned ktile = (unsigned)(kKV * p.kss * 2), vtile = (unsigned)(kKV * p.vss * 2);
    // which = 0: the K image, 1: the V image (issued at two different points of the tile loop)
    auto stage_part = [&](int tile, int buf, int which) {
        const unsigned kb = lds0 + buf * C::TILE_BYTES, vb = kb + C::K_BYTES;
        const bool full = (tile + 1) * kKV <= p.S;           // wave-uniform; the ragged last tile fetches keys >= S as zeros
        if (which == 0) {
#pragma unroll
            for (int i = 0; i < KPW; ++i) {
                const int j = j0 + i * jstep;
                if (j < KC) {                                // wave-uniform
                    if (full) dma16(krs, kvo[i], tile * ktile, kb + j * 1024);
                    else dma16(krs, tile * kKV + krow[i] < p.S ? kvo[i] : kOob, tile * ktile, kb + j * 1024);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < VPW; ++i) {
                const int j = j0 + i * jstep;
                if (j < VC) {
                    if (full) dma16(vrs, vvo[i], tile * vtile, vb + j * 1024);
                    else dma16(vrs, tile * kKV + vrow[i] < p.S ? vvo[i] : kOob, tile * vtile, vb + j * 1024);
                }
            }
        }
    };
    auto stage = [&](int tile, int buf) { stage_part(tile, buf, 0); stage_part(tile, buf, 1); };
    if constexpr (LOADER) {
        if (wave >= WAVES) {
            // a loader wave (with two of them, each takes every other piece): tile t+1 goes out right behind the barrier that proves every computing wave has left the buffer
            // it overwrites (they arrive there after tile t-1); tile t is complete - `vmcnt(0)` - before the barrier publishes it
            stage(0, 0);
            if (ntiles > 1) stage(1, 1);
            // pieces this wave issues per tile, at least: what may stay outstanding while the OLDER tile is known to be complete
            constexpr int kKeep = KC / NISSUE + VC / NISSUE;
            for (int t = 0; t < ntiles; ++t) {
                // tile t has landed once only (a part of) tile t+1's pieces are outstanding: vmcnt retires in issue order
                if (t + 1 < ntiles) __builtin_amdgcn_s_waitcnt((kKeep & 15) | (7 << 4) | (15 << 8) | ((kKeep >> 4) << 14));
                else __builtin_amdgcn_s_waitcnt((0 & 15) | (7 << 4) | (15 << 8));
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();          // publishes tile t; every computing wave has left tile t-1's buffer
                if (t + 2 < ntiles) stage(t + 2, (t + 2) % 3);
            }
            return;
        }
    } else {
        stage(0, 0);
    }

    // ---- constant LDS region of the row-sum trick: rows of VP halves, [row][0] = 1, [row][1..15] = 0 (the rest is never read)
    if (ONES) {
        for (int idx = threadIdx.x; idx < kKV * 2; idx += 64 * WAVES) {           // (a loader wave has left by now)
            const h8_t one = {(half_t)1, 0, 0, 0, 0, 0, 0, 0}, zero = {0, 0, 0, 0, 0, 0, 0, 0};
            *reinterpret_cast<__attribute__((address_space(3))) h8_t*>((uintptr_t)(ones0 + (idx >> 1) * VP * 2 + (idx & 1) * 16)) =
                (idx & 1) ? zero : one;
        }
    }

    // ---- Q^T fragments stay in registers, pre-multiplied by scale * log2(e): qf[ks] = Q[q0 + r][16 ks + 8 hh .. +8]
    h8_t qf[NK];
    {
        const int row = min(q0 + r, p.L - 1);
        const half_t* qp = p.q + b * p.qsb + h * p.qsh + (long long)row * p.qsl;
        const float c2 = p.scale_log2e;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int col = 16 * ks + 8 * hh;
            h8_t val = {0, 0, 0, 0, 0, 0, 0, 0};
            if (col < p.d) val = *reinterpret_cast<const h8_t*>(qp + col);
#pragma unroll
            for (int j = 0; j < 8; ++j) val[j] = (half_t)((float)val[j] * c2);
            qf[ks] = val;
        }
    }

    f16x_t o[DM], negm;                                      // negm: -(running max) of this lane's query row, 16 copies
    float l_run = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) negm[i] = 0.f;
#pragma unroll
    for (int dm = 0; dm < DM; ++dm)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dm][i] = 0.f;

    // per-lane parts of the LDS read addresses (bytes), relative to a buffer's K / V image
    const unsigned k_lane = (unsigned)(r * KP + 8 * hh) * 2u;
    const int g = (lane >> 4) & 1;
    const unsigned v_lane = (unsigned)((4 * hh + ((lane & 15) >> 2)) * VP + 16 * g + 4 * (lane & 3)) * 2u;
    // last channel tile: the g = 1 lanes (channels 16 NK ..) read the constant region; its address is biased so that the
    // tile's common immediate (32 (DM-1) halves) lands on column 0
    const unsigned ones_lane = ones0 + (unsigned)((4 * hh + ((lane & 15) >> 2)) * VP + 4 * (lane & 3)) * 2u - 64u * (DM - 1);

    const bool dbg = p.stamps != nullptr && blockIdx.x == 0 && wave == (int)(reinterpret_cast<uintptr_t>(p.stamps) & 7);
    unsigned long long seg[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tprev = dbg ? __builtin_amdgcn_s_memtime() : 0;

    h8_t pf[4];                                              // probabilities of a tile, packed as the P.V B operand
    h8_t vf[HOIST ? DM : 1][HOIST ? 4 : 1];                  // its V^T fragments (HOIST kernels)
    const bool late = STAGGER && wave >= 4;                  // wave-uniform
    auto pv_hoisted = [&]() {
#pragma unroll
        for (int dm = 0; dm < DM; ++dm)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) o[dm] = mfma_32x32x16(vf[HOIST ? dm : 0][HOIST ? tt : 0], pf[tt], o[dm]);
    };
    // ---- softmax, base 2, of the tile whose S'^T sits in s[]: probabilities into pf[], the running max / sums updated
    f16x_t s[2];
    float psum = 0.f;
    // STAGGER == 2: every wave runs the same sequence QK^T(t), softmax(t), P.V(t), QK^T(t+1), ... - what differs is where in
    // it the tile barrier falls.  Waves 0-3 meet it between QK^T(t) and softmax(t), waves 4-7 between softmax(t) and P.V(t):
    // within a period the first group runs [softmax(t-1) | P.V(t-1), QK^T(t)] = [VALU | 14 MFMAs] while its SIMD partners run
    // [P.V(t-1), QK^T(t) | softmax(t)] = [14 MFMAs | VALU].  (STAGGER == 1 leaves waves 0-3 in the plain order, so their
    // QK^T runs beside the partners' P.V + QK^T: matrix beside matrix on one SIMD.)  Scores and V^T fragments cross the barrier
    // in registers; both groups read K(t) and V(t) from LDS in period t, as the loader's ring of three assumes.
    const bool rotated = STAGGER == 2 && !late;              // wave-uniform
    // (static s_setprio 1 for either group: no gain on the rotated kernel, 69 -> 77 us on STAGGER == 1 with waves 4-7 raised)
    if constexpr (STAGGER == 2) {
        if (rotated) {
            int rb = NBUF - 1;
            for (int t = 0; t < ntiles; ++t) {
                rb = rb + 1 == NBUF ? 0 : rb + 1;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                const unsigned kb = lds0 + rb * C::TILE_BYTES + k_lane;
                const unsigned vb = lds0 + rb * C::TILE_BYTES + C::K_BYTES + v_lane;
                const unsigned vb_last = (ONES && g == 1) ? ones_lane : vb;
                SA_STAMP(0)
                if (t > 0) {
                    SA_SOFTMAX_HEAD(t - 1)
                    SA_SOFTMAX_HALF(0)
                    SA_SOFTMAX_HALF(1)
                    if (!ONES) l_run += psum;
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (dbg) asm volatile("" :: "v"(pf[0][0]), "v"(pf[3][7]));
                SA_STAMP(2)
                // K(t)'s fragments are read in the middle of P.V(t-1): behind its key steps 0 and 1, whose probabilities and V^T
                // fragments are dead by then (read in front of P.V they made the kernel spill at 168 registers), and with key
                // steps 2 and 3 - four MFMAs - to arrive under
                if (t > 0) {
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int dm = 0; dm < DM; ++dm) o[dm] = mfma_32x32x16(vf[HOIST ? dm : 0][HOIST ? tt : 0], pf[tt], o[dm]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                h8_t kf[2][NK];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int ks = 0; ks < NK; ++ks) kf[m][ks] = lds_read_h8(kb + (32 * m * KP + 16 * ks) * 2);
                __builtin_amdgcn_sched_barrier(0);
                if (t > 0) {
#pragma unroll
                    for (int tt = 2; tt < 4; ++tt)
#pragma unroll
                        for (int dm = 0; dm < DM; ++dm) o[dm] = mfma_32x32x16(vf[HOIST ? dm : 0][HOIST ? tt : 0], pf[tt], o[dm]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (dbg) asm volatile("" :: "v"(o[0][0]), "v"(o[DM - 1][15]));
                SA_STAMP(3)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int ks = 0; ks < NK; ++ks) s[m] = mfma_32x32x16(kf[m][ks], qf[ks], ks == 0 ? negm : s[m]);
                __builtin_amdgcn_sched_barrier(0);
                if (dbg) asm volatile("" :: "v"(s[0][0]), "v"(s[1][15]));
                SA_STAMP(1)
#pragma unroll
                for (int dm = 0; dm < DM; ++dm)
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        const unsigned a = ((ONES && dm == DM - 1) ? vb_last : vb) + (16 * tt * VP + 32 * dm) * 2;
                        const h4_t lo = lds_read_tr(a);
                        const h4_t hi = lds_read_tr(a + 8 * VP * 2);
                        vf[HOIST ? dm : 0][HOIST ? tt : 0] = h8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            // (the tile index goes through an opaque move: known before the loop, the ragged-tile mask bound `lim` of the last tile
            // was computed up there and kept - spilled - across the whole loop)
            int t_last = ntiles - 1;
            asm volatile("" : "+s"(t_last));
            SA_SOFTMAX_HEAD(t_last)
            SA_SOFTMAX_HALF(0)
            SA_SOFTMAX_HALF(1)
            if (!ONES) l_run += psum;
            pv_hoisted();
        }
    }
    // (an `else` of the rotated loop, not a loop of zero trips behind it: what only this path needs - the LDS lane offsets - would
    // otherwise stay live, i.e. spilled, across the rotated loop)
    if (!rotated) {
    int buf = NBUF - 1;
    for (int t = 0; t < ntiles; ++t) {
        buf = buf + 1 == NBUF ? 0 : buf + 1;                 // t % NBUF
        const int nbuf1 = buf + 1 == NBUF ? 0 : buf + 1;     // (t + 1) % NBUF: where a non-loader kernel stages the next tile
        // tile t: this wave's pieces have landed (the only DMAs outstanding), then everyone's - and every wave is done
        // reading buffer buf ^ 1 (tile t-1), so tile t+1 may go there
        if (LOADER) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the ones region's stores, first tile)
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // The compiler puts an `s_waitcnt vmcnt(0)` in front of every ds_read_b64_tr_b16 that follows an LDS-DMA it has
        // seen (it cannot tell the transposed read from the DMA's destination).  HOIST kernels therefore issue ALL LDS reads
        // of tile t first and the DMA of tile t+1 behind them; the others issue the DMA here and pay part of its latency
        // at their first transposed read.
        if (!LOADER && !HOIST && t + 1 < ntiles) stage(t + 1, nbuf1);
        SA_STAMP(0)
        const unsigned kb = lds0 + buf * C::TILE_BYTES + k_lane;
        const unsigned vb = lds0 + buf * C::TILE_BYTES + C::K_BYTES + v_lane;
        const unsigned vb_last = (ONES && g == 1) ? ones_lane : vb;

        auto k_frag = [&](int m, int ks) { return lds_read_h8(kb + (32 * m * KP + 16 * ks) * 2); };
        auto v_frag = [&](int dm, int tt) {
            const unsigned a = ((ONES && dm == DM - 1) ? vb_last : vb) + (16 * tt * VP + 32 * dm) * 2;
            const h4_t lo = lds_read_tr(a);
            const h4_t hi = lds_read_tr(a + 8 * VP * 2);
            return h8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        if (STAGGER && late && t > 0) {                      // the previous tile's P.V, beside the partner wave's QK^T
            pv_hoisted();
            __builtin_amdgcn_sched_barrier(0);
        }
        h8_t kf[HOIST ? 2 : 1][HOIST ? NK : 1];
        if constexpr (HOIST) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) kf[m][ks] = k_frag(m, ks);
            if constexpr (!V_LATE) {
#pragma unroll
                for (int dm = 0; dm < DM; ++dm)
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) vf[dm][tt] = v_frag(dm, tt);
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- S'^T = K . Q'^T - m  (2 row tiles of 32 keys): the chain starts from negm
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const h8_t kfr = HOIST ? kf[HOIST ? m : 0][HOIST ? ks : 0] : k_frag(m, ks);
                s[m] = mfma_32x32x16(kfr, qf[ks], ks == 0 ? negm : s[m]);
            }
        }
        if constexpr (V_LATE) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dm = 0; dm < DM; ++dm)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) vf[dm][tt] = v_frag(dm, tt);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (dbg) asm volatile("" :: "v"(s[0][0]), "v"(s[1][15]));
        SA_STAMP(1)

        SA_SOFTMAX_HEAD(t)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            // HOIST kernels: the DMA of tile t+1 is issued HERE, behind every LDS read of tile t (see above) and in the
            // VALU-only stretch of the tile (an LDS-DMA instruction issued among LDS reads costs 2-3x as much issue time):
            // the K image before the first half of the exponentials, the V image before the second
            if (!LOADER && HOIST && t + 1 < ntiles) {
                __builtin_amdgcn_sched_barrier(0);
                stage_part(t + 1, nbuf1, m);
                __builtin_amdgcn_sched_barrier(0);
            }
            SA_SOFTMAX_HALF(m)
        }
        if (!ONES) l_run += psum;
        SA_STAMP(2)

        // ---- O^T += V^T . P^T  (4 k-steps of 16 keys, DM row tiles of 32 channels)
        if constexpr (STAGGER) {
            if (!late) pv_hoisted();
        } else {
#pragma unroll
            for (int dm = 0; dm < DM; ++dm) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const h8_t vfr = HOIST ? vf[HOIST ? dm : 0][HOIST ? tt : 0] : v_frag(dm, tt);
                    o[dm] = mfma_32x32x16(vfr, pf[tt], o[dm]);
                }
            }
        }
        if (dbg) asm volatile("" :: "v"(o[0][0]), "v"(o[DM - 1][15]));
        SA_STAMP(3)
    }
    if (STAGGER && late) pv_hoisted();                       // the last tile's P.V of the late waves
    }
    if (dbg && lane == 0) {
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(reinterpret_cast<uintptr_t>(p.stamps) & ~uintptr_t(7));
        for (int i = 0; i < 6; ++i) dst[i] = seg[i];
    }

    // ---- epilogue: O / l, fp16, out[b, q, h, :].  The lane id is re-derived here (v_mbcnt) so that nothing the epilogue
    // addresses with has to stay live - or be spilled - across the tile loop of the 168-register kernels
    const int lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int r_e = lane_e & 31, hh_e = lane_e >> 5;
    float l_tot;
    if (ONES) {
        // channel 16 NK of O^T is sum_s p: row tile DM-1, element 8 of the hh = 0 lanes (16 = (i & 3) + 8 (i >> 2) + 4 hh)
        l_tot = __shfl(o[DM - 1][8], r_e, 64);               // lane r (hh = 0) holds it for query row r
    } else {
        l_tot = l_run + __shfl_xor(l_run, 32, 64);
    }
    const float inv = 1.f / l_tot;
    const int qrow = q0 + r_e;
    half_t* op = p.out + b * p.osb + h * p.osh + (long long)qrow * p.osl;
#pragma unroll
    for (int dm = 0; dm < DM; ++dm) store_o_block(op, o[dm], inv, dm, hh_e, p.d, qrow < p.L, p.wide_store != 0);
}

template <int NK, int WAVES, int MINW, int D8 = 0, int NLOAD = 0, int STAGGER = 0>
int launch(const SaParams& p0, hipStream_t st) {
    SaParams p = p0;
    p.nqb = (p.L + 32 * WAVES - 1) / (32 * WAVES);
    p.xcd_map = ((p.Bc * p.H) % 8 == 0) ? 1 : 0;
    const size_t lds = (size_t)SaCfg<NK, D8>::lds_bytes(NLOAD > 0 ? 3 : 2);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&self_attn_fwd<NK, WAVES, MINW, D8, NLOAD, STAGGER>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    DSC_LAUNCH((self_attn_fwd<NK, WAVES, MINW, D8, NLOAD, STAGGER>), dim3(p.Bc * p.H * p.nqb), dim3(64 * (WAVES + NLOAD)), lds, st, p);
    return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
}

int g_sa_variant = 0;            // tuning knob (dsc_debug_set_self_attn_variant): 0 = auto
// Not built (examined, round 2): two query tiles per wave (64 rows: half the LDS reads and tile bytes per MFMA).  It needs
// ~300 registers.  With a loader wave two waves share a SIMD, i.e. 256 registers each: 33 spills.  Without one, at 512, the
// compiler parks the accumulators in AGPRs and copies them around every branch (~290 v_accvgpr per tile), and the DMA must
// then be issued by the computing waves, where every ds_read_b64_tr_b16 behind a pending DMA gets a compiler-inserted
// vmcnt(0).

// Tilings (dsc_debug_set_self_attn_variant forces one; 0 = choose from the shape):
//   1  4 waves of 32 query rows, every wave issues its share of the DMA
//   2  8 waves (256 query rows share each K / V tile), likewise
//   3  4 waves under a three-waves-per-SIMD register budget (the configuration that faulted with register-staged tiles)
//   4 / 5  4 / 8 computing waves + 1 loader wave        6 / 7  4 / 8 computing waves + 2 loader waves
//   8 / 9 / 10  as 6 / 7 / 5 with the compact d = 40 image (10 DMA pieces per tile instead of 13); 11 / 12  as 1 / 2 with it
//   13 / 14  8 computing waves + 1 / 2 loader waves, waves 4-7 staggered by one P.V (compact image at d = 40)
//   17 / 18  as 14, and waves 0-3 staggered by one softmax + P.V (STAGGER == 2: [VALU | MFMA] beside [MFMA | VALU]); d = 40 only
//   15 / 16  2 / 4 computing waves + 4 loader waves (any head dim)
template <int NK>
int launch_nk(const SaParams& p, hipStream_t st) {
    constexpr int MW = NK <= 5 ? 2 : 1;                       // waves per SIMD the registers allow: 256 / 512 per wave
    constexpr int NL = NK <= 4 ? NK : 3;                      // (instantiation guards: the loader kernels exist for NK <= 4)
    // (A/B in the step: 100 + v applies v to the d = 40 level only, 200 = the other levels without their loader kernels)
    const bool d40 = NK == 3 && p.d == 40;
    const int v = g_sa_variant >= 200 ? 0 : (g_sa_variant >= 100 ? (d40 ? g_sa_variant - 100 : 0) : g_sa_variant);
    const bool small_loaders = g_sa_variant != 200;
    const long long wg8 = (long long)p.Bc * p.H * ((p.L + 255) / 256);
    const long long wg4 = (long long)p.Bc * p.H * ((p.L + 127) / 128);
    if (NK <= 4) {
        const bool compact = NK == 3 && p.d == 40;
        // (NK = 4 does not fit 168 registers next to its own DMA plan: its three-waves-per-SIMD kernel is a loader one,
        // and one loader wave alone cannot carry its 17 pieces per tile: NK = 4 always gets two)
        constexpr int N3 = NK <= 3 ? NK : 3;
        if (v == 3) return NK <= 3 ? launch<N3, 4, 3>(p, st) : launch<NL, 4, 3, 0, 2>(p, st);
        if (v == 4) return NK <= 3 ? launch<N3, 4, 3, 0, 1>(p, st) : launch<NL, 4, 3, 0, 2>(p, st);
        if (v == 5) return NK <= 3 ? launch<N3, 8, 2, 0, 1>(p, st) : launch<NL, 8, 2, 0, 2>(p, st);
        if (v == 6) return launch<NL, 4, 3, 0, 2>(p, st);
        if (v == 7) return launch<NL, 8, 2, 0, 2>(p, st);
        if (v == 8 && compact) return launch<3, 4, 3, 5, 2>(p, st);
        if (v == 9 && compact) return launch<3, 8, 2, 5, 2>(p, st);
        if (v == 10 && compact) return launch<3, 8, 2, 5, 1>(p, st);
        if (v == 11 && compact) return launch<3, 4, 2, 5, 0>(p, st);
        if (v == 12 && compact) return launch<3, 8, 2, 5, 0>(p, st);
        constexpr int NS = NK <= 3 ? NK : 3;                  // (the staggered kernel of NK = 4 spills at 168 registers: not built)
        if (v == 13) return compact ? launch<3, 8, 2, 5, 1, 1>(p, st) : (NK <= 3 ? launch<NS, 8, 2, 0, 2, 1>(p, st) : launch<NL, 8, 2, 0, 2>(p, st));
        if (v == 14) return compact ? launch<3, 8, 2, 5, 2, 1>(p, st) : (NK <= 3 ? launch<NS, 8, 2, 0, 2, 1>(p, st) : launch<NL, 8, 2, 0, 2>(p, st));
        // measured (tools/mb_sa.py, head-major K / V): at batch 1 (+CFG) the SD1.5 64x64 level offers 256 workgroups of 256
        // query rows - one per CU, 8 computing waves + a loader wave: 64 us against 70 for two 4-wave workgroups per CU
        // that issue their own DMA; at 8 images the 4-wave workgroups win (480 vs 505 us)
        // (staggered: 64.0 us against 66.6 for the same kernel with all eight waves in phase)
        // (the rotated stagger exists for the compact d = 40 image with two loader waves only: with one loader wave, or with the
        // generic image, it spills at 168 registers; every other head dim keeps STAGGER == 1 under 17 / 18)
        if (v == 17 || v == 18) return compact ? launch<3, 8, 2, 5, 2, 2>(p, st) : (NK <= 3 ? launch<NS, 8, 2, 0, 2, 1>(p, st) : launch<NL, 8, 2, 0, 2>(p, st));
        // (round 3, rotated stagger: 60.3 us against 62.3 for STAGGER == 1 at batch 1 + CFG, 472 against 479 for the 4-wave
        // workgroups at 8 images - the same bits as every other tiling)
        // (in the step, 64x64 level: 73.6 us STAGGER == 1 -> 69.1 rotated, one loader wave -> 67.2 with two)
        if (v == 0 && compact && wg8 >= 256) return launch<3, 8, 2, 5, 2, 2>(p, st);
        if (v == 0 && NK == 3 && wg8 >= 256 && wg8 < 512) return launch<3, 8, 2, 0, 2, 1>(p, st);
        // one image's 8 heads (the shared CFG prefix runs the first self-attention once per image): 256 workgroups of 128 query
        // rows, one per CU - four computing waves + two loader waves 43.4 us against 51.8 for the four waves issuing their own DMA
        if (v == 0 && compact && wg4 > 128 && wg4 <= 256) return launch<3, 4, 3, 5, 2>(p, st);
        if (v == 0 && compact && wg4 >= 256) return launch<3, 4, 2, 5, 0>(p, st);
    }
    // 15 / 16: 2 / 4 computing waves + FOUR loader waves, any head dim - the small-L levels (d = 80 at 32x32, d = 160 at 16x16)
    // have too few query rows to share a tile among many waves, so without loaders each of the 2 computing waves issues 10-20
    // DMA pieces per tile (1300-2700 cycles against 700-1300 of MFMA)
    // (d = 160 does not fit the 256 registers that six waves per workgroup leave each wave: two loaders, four waves, 512)
    if (v == 15) return NK <= 8 ? launch<(NK <= 8 ? NK : 3), 2, 2, 0, 4>(p, st) : launch<NK, 2, 1, 0, 2>(p, st);
    if (v == 16) return NK <= 8 ? launch<(NK <= 8 ? NK : 3), 4, 2, 0, 4>(p, st) : launch<NK, 2, 1, 0, 2>(p, st);
    // measured (tools/mb_sa.py): d = 80 @ L = 1024 21.4 -> 15.1 us, d = 160 @ L = 256 16.5 -> 11.9 us, @ L = 64 8.6 -> 6.8 us
    if (v == 0 && NK >= 5 && wg4 < 256 && small_loaders) return NK <= 8 ? launch<(NK <= 8 ? NK : 3), 2, 2, 0, 4>(p, st) : launch<NK, 2, 1, 0, 2>(p, st);
    if (NK == 5 && ((v == 0 && wg8 >= 512) || v == 2)) return launch<(NK <= 5 ? NK : 3), 8, 2>(p, st);
    if (NK <= 4 && v == 2) return launch<(NK <= 5 ? NK : 3), 8, 2>(p, st);
    if (wg4 >= 256 || v == 1) return launch<NK, 4, MW>(p, st);
    const long long wg2 = (long long)p.Bc * p.H * ((p.L + 63) / 64);
    if (wg2 >= 128 || NK >= 6) return launch<NK, 2, MW>(p, st);      // (NK >= 6: one wave alone would carry 40 DMA offsets)
    return launch<(NK <= 5 ? NK : 3), 1, MW>(p, st);
}

bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
bool strides_ok(const int64_t s[3]) { return s[0] % 8 == 0 && s[1] % 8 == 0 && s[2] % 8 == 0; }

}  // namespace

extern "C" void dsc_debug_set_self_attn_variant(int v) { g_sa_variant = v; }
extern "C" void dsc_debug_set_self_attn_stamp_wave(int w) { g_sa_stamp_wave = w; }
extern "C" void dsc_debug_set_self_attn_stamps(void* device_buffer_64B) { g_sa_stamps = static_cast<unsigned long long*>(device_buffer_64B); }

extern "C" int dsc_self_attn_fwd(const void* q, const void* k, const void* v, void* out, int Bc, int H, int L, int S,
                                 int d, const int64_t q_strides[3], const int64_t k_strides[3],
                                 const int64_t v_strides[3], const int64_t o_strides[3], float scale, int dtype,
                                 void* stream) {
    if (!q || !k || !v || !out || !q_strides || !k_strides || !v_strides || !o_strides) return DSC_ERR_BAD_ARG;
    if (Bc <= 0 || H <= 0 || L <= 0 || S <= 0 || d <= 0) return DSC_ERR_BAD_ARG;
    if (dtype != DSC_F16 || d % 8 != 0 || d > 160) return DSC_ERR_UNSUPPORTED;
    if (!aligned16(q) || !aligned16(k) || !aligned16(v) || (reinterpret_cast<uintptr_t>(out) & 7) ||
        !strides_ok(q_strides) || !strides_ok(k_strides) || !strides_ok(v_strides) || !strides_ok(o_strides))
        return DSC_ERR_UNSUPPORTED;
    // the K / V tiles are addressed through 32-bit buffer offsets: key-row strides must be positive and one (b, h) slice
    // must stay below 2 GiB (an SD / SDXL slice is a few MB)
    if (k_strides[1] <= 0 || v_strides[1] <= 0) return DSC_ERR_UNSUPPORTED;
    const long long kext = ((long long)(S - 1) * k_strides[1] + d) * 2, vext = ((long long)(S - 1) * v_strides[1] + d) * 2;
    const long long kreach = ((long long)((S + kKV - 1) / kKV) * kKV * k_strides[1] + d) * 2;
    const long long vreach = ((long long)((S + kKV - 1) / kKV) * kKV * v_strides[1] + d) * 2;
    if (kext >= (1ll << 31) || vext >= (1ll << 31) || kreach >= (1ll << 31) || vreach >= (1ll << 31)) return DSC_ERR_UNSUPPORTED;
    SaParams p{};
    p.q = static_cast<const half_t*>(q); p.k = static_cast<const half_t*>(k);
    p.v = static_cast<const half_t*>(v); p.out = static_cast<half_t*>(out);
    p.Bc = Bc; p.H = H; p.L = L; p.S = S; p.d = d;
    p.scale_log2e = (scale > 0.f ? scale : 1.0f / sqrtf((float)d)) * 1.4426950408889634f;
    p.qsb = q_strides[0]; p.qsl = q_strides[1]; p.qsh = q_strides[2];
    p.ksb = k_strides[0]; p.kss = k_strides[1]; p.ksh = k_strides[2];
    p.vsb = v_strides[0]; p.vss = v_strides[1]; p.vsh = v_strides[2];
    p.osb = o_strides[0]; p.osl = o_strides[1]; p.osh = o_strides[2];
    p.wide_store = ((reinterpret_cast<uintptr_t>(out) & 15) == 0 && o_strides[0] % 8 == 0 && o_strides[1] % 8 == 0 && o_strides[2] % 8 == 0) ? 1 : 0;
    p.k_bytes = (unsigned)kext; p.v_bytes = (unsigned)vext;
    p.stamps = g_sa_stamps ? reinterpret_cast<unsigned long long*>(reinterpret_cast<uintptr_t>(g_sa_stamps) | (uintptr_t)(g_sa_stamp_wave & 7)) : nullptr;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (d <= 16) return launch_nk<1>(p, st);
    if (d <= 32) return launch_nk<2>(p, st);
    if (d <= 48) return launch_nk<3>(p, st);
    if (d <= 64) return launch_nk<4>(p, st);
    if (d <= 80) return launch_nk<5>(p, st);
    if (d <= 96) return launch_nk<6>(p, st);
    if (d <= 128) return launch_nk<8>(p, st);
    return launch_nk<10>(p, st);
}
