// Flash self-attention forward for gfx950 (MI355X): the kernel behind dsc_self_attn_fwd.
//
// Replaces `F.scaled_dot_product_attention(query, key, value)` on the self-attention branch of the processors
// (source/modules/attention_modify.py:483-485) for the UNet's head dims 40 / 64 / 80 / 160.
// The L x L score matrix is never materialised: per 32-query-row wave, KV tiles of 64 keys stream through LDS with an
// online softmax (running max m, running sum l per query row).
//
// MI355X mapping (64-wide waves, MFMA 32x32x16 f16):
//   * S^T[kv, q] = K[kv, :] . Q[q, :]  ("swapped" product: A = K rows from LDS, B = Q^T kept in registers for the whole
//     kernel) puts ONE query row on each lane (q = lane & 31) with half of the tile's 64 scores in its registers, so
//     max / exp2 / sum are lane-local plus one exchange with lane ^ 32;
//   * O^T[d, q] += V^T[d, kv] . P^T[kv, q]: the fp16-packed probabilities are already the B operand (k order
//     permuted inside each 16-step, cdna_hip_programming.md section 3 "accumulator tile as the next MFMA's operand"); the A
//     operand V^T comes from the ROW-MAJOR V tile in LDS through ds_read_b64_tr_b16 (hardware transpose), two reads
//     per MFMA, conflict-free with a row stride of 96 / 160 halves;
//   * K / V tiles are double-buffered in LDS: the global loads of tile t+1 are issued before the MFMAs of tile t and
//     written after them (one barrier per tile);
//   * blockIdx is remapped so that all query blocks of one (b, h) run on one XCD: its K / V (L*d*4 bytes) is fetched
//     from HBM once and then served by that XCD's L2.
// Bound: MFMA (arithmetic intensity 4*L*L*C / 8*L*C = L/2 FLOP/B); head dim 40 pads to 48 (QK^T) and 64 (PV).
#include "dsc_common.h"
#include "dsc_hip.h"

namespace {

typedef short s4_t __attribute__((__vector_size__(4 * sizeof(short))));

constexpr int kKV = 64;          // keys per tile

struct SaParams {
    const half_t* q; const half_t* k; const half_t* v; half_t* out;
    int Bc, H, L, S, d;          // S = number of keys (== L for self-attention)
    int nqb, xcd_map;
    float scale_log2e;
    long long qsb, qsl, qsh, ksb, kss, ksh, vsb, vss, vsh, osb, osl, osh;
    unsigned long long* stamps;
};

template <int NK>
struct SaCfg {
    static constexpr int DM = (NK + 1) / 2;
    static constexpr int KP = 16 * NK + 8;                  // K row stride (halves): odd multiple of 16 B
    static constexpr int VP = (DM <= 3) ? 96 : 160;         // V row stride: (VP/2) % 64 in {16, 48} -> tr reads conflict-free
    static constexpr int TILE_HALVES = kKV * KP + kKV * VP; // one (K, V) buffer
};

// Prefetch load that hipcc's s_waitcnt insertion does not see (cdna_hip_programming.md section 5.7 form (ii)): the compiler
// otherwise waits vmcnt(0) before the FIRST MFMA after the loads - exposing the full memory latency every tile - because
// it reuses the loads' address registers.  The matching wait is stage_wait() right before the LDS write.
__device__ __forceinline__ void hidden_load(h8_t& dst, const half_t* src) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
}

// one v_max3_f32; fmaxf() costs a canonicalising v_max_f32 per input on top of the max itself (53 + 8 instructions per
// tile instead of 16) and the kernel is VALU-issue bound.  Scores are finite or -inf, never NaN.
__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ h4_t tr_read(const half_t* p) {
    const s4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s4_t __attribute__((address_space(3)))*)(const_cast<half_t*>(p)));
    return __builtin_bit_cast(h4_t, r);
}

// QT query tiles (32 rows each) per wave share every K fragment read and every V^T transposed read.
// ONES (NK odd: the PV row tile has >= 16 padding channels): V's first padding column holds 1.0, so the PV MFMA
// returns sum_s p[s] in output channel 16*NK for free - no per-score adds, and the sum uses the same fp16-rounded p as
// the numerator.
unsigned long long* g_sa_stamps = nullptr;        // diagnostic: per-segment cycle sums of workgroup 0 / wave 0

#define SA_STAMP(slot)                                                                          \
    if (dbg) {                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                             \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                     \
        seg[slot] += t_ - tprev; tprev = t_;                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    }

// Head dims <= 80: ask for two waves per SIMD (256 registers per wave).  Without the second launch-bound the compiler
// parks the S / O accumulators in AGPRs and moves them to VGPRs and back for the softmax and the rescale:
// 191 of the 277 VALU instructions per tile were v_accvgpr_read/write (rocprofv3 PMC: the kernel is VALU-issue bound).
template <int NK, int WAVES, int QT>
__global__ __launch_bounds__(64 * WAVES, (NK <= 5 && QT == 1 && WAVES >= 2 ? 2 : 1)) void self_attn_fwd(SaParams p) {   // HIP: 2nd = min waves per SIMD
    using C = SaCfg<NK>;
    constexpr int T = 64 * WAVES, DM = C::DM, KP = C::KP, VP = C::VP;
    constexpr int CH = (kKV * 2 * NK + T - 1) / T;          // 16-byte chunks per thread per operand per tile
    constexpr bool ONES = (NK & 1) != 0;
    constexpr float kTau = 8.f;                              // lazy rescale: p <= 2^8, exact enough in fp16 / fp32 sums
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* lds = reinterpret_cast<half_t*>(smem);

    int bh, qb;
    {
        const int bid = blockIdx.x, nbh = p.Bc * p.H;
        if (p.xcd_map) { const int x = bid & 7, j = bid >> 3, nb8 = nbh >> 3; bh = x + 8 * (j % nb8); qb = j / nb8; }
        else { bh = bid % nbh; qb = bid / nbh; }
    }
    const int b = bh / p.H, h = bh % p.H;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, hh = lane >> 5;
    const int q0 = (qb * WAVES + wave) * 32 * QT;
    const int d8 = p.d >> 3;
    const half_t* kg = p.k + b * p.ksb + h * p.ksh;
    const half_t* vg = p.v + b * p.vsb + h * p.vsh;

    // one-time LDS constants of both buffers: zero K pad columns [d, 16*NK); V ones column 16*NK (ONES)
    if (16 * NK > p.d) {
        const int padc = 16 * NK - p.d;
        for (int idx = threadIdx.x; idx < 2 * kKV * padc; idx += T) {
            const int buf = idx / (kKV * padc), rem = idx % (kKV * padc);
            lds[buf * C::TILE_HALVES + (rem / padc) * KP + p.d + rem % padc] = (half_t)0;
        }
    }
    if (ONES) {
        for (int idx = threadIdx.x; idx < 2 * kKV; idx += T)
            lds[(idx / kKV) * C::TILE_HALVES + kKV * KP + (idx % kKV) * VP + 16 * NK] = (half_t)1;
    }

    // Q^T fragments stay in registers: qf[qt][ks] = Q[q0 + 32 qt + r][16 ks + 8 hh .. +8]
    h8_t qf[QT][NK];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int row = min(q0 + 32 * qt + r, p.L - 1);
        const half_t* qp = p.q + b * p.qsb + h * p.qsh + (long long)row * p.qsl;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int col = 16 * ks + 8 * hh;
            h8_t val = {0, 0, 0, 0, 0, 0, 0, 0};
            if (col < p.d) val = *reinterpret_cast<const h8_t*>(qp + col);
            qf[qt][ks] = val;
        }
    }

    h8_t kst[CH], vst[CH];                                   // staging registers for the next tile
    // per-thread staging geometry, fixed for the whole kernel: chunk c of this thread is row srow[c], 16-byte column scol[c]
    int srow[CH], scol[CH];
    const half_t* kptr[CH];
    const half_t* vptr[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int idx = threadIdx.x + c * T;
        srow[c] = idx / d8;
        scol[c] = idx - srow[c] * d8;
        kptr[c] = kg + (long long)srow[c] * p.kss + scol[c] * 8;
        vptr[c] = vg + (long long)srow[c] * p.vss + scol[c] * 8;
    }
    const long long ktile = (long long)kKV * p.kss, vtile = (long long)kKV * p.vss;
    auto stage_load = [&](int tile) {
        const bool ragged = (tile + 1) * kKV > p.S;              // wave-uniform: only the last tile can run past S
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (srow[c] < kKV) {
                long long ko = (long long)tile * ktile, vo = (long long)tile * vtile;
                if (ragged) {                                    // clamp the row to the last valid key (masked in the softmax)
                    const int back = max(tile * kKV + srow[c] - (p.S - 1), 0);
                    ko -= (long long)back * p.kss;
                    vo -= (long long)back * p.vss;
                }
                hidden_load(kst[c], kptr[c] + ko);
                hidden_load(vst[c], vptr[c] + vo);
            }
        }
    };
    auto stage_write = [&](int buf) {
        half_t* Kb = lds + buf * C::TILE_HALVES;
        half_t* Vb = Kb + kKV * KP;
#pragma unroll
        for (int c = 0; c < CH; ++c)                             // the hidden loads have landed: name every destination
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(kst[c]), "+v"(vst[c]) :: "memory");
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (srow[c] < kKV) {
                *reinterpret_cast<h8_t*>(Kb + srow[c] * KP + scol[c] * 8) = kst[c];
                *reinterpret_cast<h8_t*>(Vb + srow[c] * VP + scol[c] * 8) = vst[c];
            }
        }
    };

    f16x_t o[QT][DM];
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        m_run[qt] = -INFINITY;
        l_run[qt] = 0.f;
#pragma unroll
        for (int dm = 0; dm < DM; ++dm)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[qt][dm][i] = 0.f;
    }

    const int ntiles = (p.S + kKV - 1) / kKV;
    stage_load(0);
    stage_write(0);
    __syncthreads();

    // per-lane constant part of the transposed-read address: row (4 hh + (i >> 2)), column 16 * ((lane >> 4) & 1) + 4 * (i & 3)
    const int tr_off = (4 * hh + ((lane & 15) >> 2)) * VP + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const float c2 = p.scale_log2e;
    const bool dbg = p.stamps != nullptr && blockIdx.x == 0 && wave == 0;
    unsigned long long seg[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tprev = dbg ? __builtin_amdgcn_s_memtime() : 0;

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) stage_load(t + 1);               // in flight during this tile's MFMAs
        SA_STAMP(0)
        const half_t* Kb = lds + buf * C::TILE_HALVES;
        const half_t* Vb = Kb + kKV * KP;

        // ---- HOIST (head dims <= 64, where the registers allow it): all LDS reads of the tile go out first - a read
        // issued right before its MFMA exposes the LDS latency at 1-2 waves per SIMD - K fragments for QK^T, then the
        // V^T fragments, whose latency the QK^T MFMAs and the softmax cover
        constexpr bool HOIST = NK <= 4;
        auto k_frag = [&](int m, int ks) { return *reinterpret_cast<const h8_t*>(Kb + (32 * m + r) * KP + 16 * ks + 8 * hh); };
        auto v_frag = [&](int dm, int tt) {
            const half_t* vp = Vb + tr_off + (16 * tt) * VP + 32 * dm;
            const h4_t lo = tr_read(vp);
            const h4_t hi = tr_read(vp + 8 * VP);
            return h8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        h8_t kf[HOIST ? 2 : 1][HOIST ? NK : 1];
        h8_t vf[HOIST ? DM : 1][HOIST ? 4 : 1];
        if constexpr (HOIST) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) kf[m][ks] = k_frag(m, ks);
#pragma unroll
            for (int dm = 0; dm < DM; ++dm)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) vf[dm][tt] = v_frag(dm, tt);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- S^T = K . Q^T  (2 row tiles of 32 keys); each K fragment feeds the QT query tiles
        f16x_t s[QT][2];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int i = 0; i < 16; ++i) s[qt][m][i] = 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const h8_t kfr = HOIST ? kf[HOIST ? m : 0][HOIST ? ks : 0] : k_frag(m, ks);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) s[qt][m] = mfma_32x32x16(kfr, qf[qt][ks], s[qt][m]);
            }

        if (dbg) asm volatile("" :: "v"(s[0][0][0]), "v"(s[0][1][15]));
        SA_STAMP(1)
        // ---- online softmax, base 2, lazy rescale.  Element i of s[qt][m] is key 32 m + (i & 3) + 8 (i >> 2) + 4 hh.
        const int kv_left = p.S - t * kKV;                   // keys valid in this tile (>= 64 except the last)
        h8_t pf[QT][4];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            if (kv_left < kKV) {                             // wave-uniform: only the ragged last tile masks
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (32 * m + (i & 3) + 8 * (i >> 2) + 4 * hh >= kv_left) s[qt][m][i] = -INFINITY;
            }
            float mx = -INFINITY;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int i = 0; i < 16; i += 2) mx = max3(mx, s[qt][m][i], s[qt][m][i + 1]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c2;     // c2 > 0: max commutes with the scaling
            if (__any(mx > m_run[qt] + kTau)) {              // wave-uniform: some row's max grew by more than 2^tau
                const float m_new = fmaxf(m_run[qt], mx);
                const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
                m_run[qt] = m_new;
                l_run[qt] *= alpha;
#pragma unroll
                for (int dm = 0; dm < DM; ++dm)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[qt][dm][i] *= alpha;
            }
            const float nm = -m_run[qt];
            float psum = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(s[qt][m][i], c2, nm));
                    if (!ONES) psum += e;
                    s[qt][m][i] = e;
                }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int i = 0; i < 16; ++i) pf[qt][2 * m + (i >> 3)][i & 7] = (half_t)s[qt][m][i];
            if (!ONES) l_run[qt] += psum;
        }

        SA_STAMP(2)
        // ---- O^T += V^T . P^T  (4 k-steps of 16 keys, DM row tiles of 32 channels); V^T fragments shared by the QT tiles
#pragma unroll
        for (int dm = 0; dm < DM; ++dm) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const h8_t vfr = HOIST ? vf[HOIST ? dm : 0][HOIST ? tt : 0] : v_frag(dm, tt);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) o[qt][dm] = mfma_32x32x16(vfr, pf[qt][tt], o[qt][dm]);
            }
        }
        if (dbg) asm volatile("" :: "v"(o[0][0][0]), "v"(o[0][DM - 1][15]));
        SA_STAMP(3)
        if (t + 1 < ntiles) stage_write(buf ^ 1);            // buffer buf^1 was last read in iteration t-1
        SA_STAMP(4)
        __syncthreads();
        SA_STAMP(5)
    }
    if (dbg && lane == 0)
        for (int i = 0; i < 6; ++i) p.stamps[i] = seg[i];

    // ---- epilogue: O / l, fp16, out[b, q, h, :]
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l_tot;
        if (ONES) {
            // channel 16*NK of O^T is sum_s p: row tile DM-1, element 8 * ((16 NK - 32 (DM-1)) >> 3) of the hh = 0 lanes
            constexpr int kIdx = 4 * ((16 * NK - 32 * (DM - 1)) >> 3);
            const float mine = o[qt][DM - 1][kIdx];
            l_tot = __shfl(mine, r, 64);                     // lane r (hh = 0) holds it for query row r
        } else {
            l_tot = l_run[qt] + __shfl_xor(l_run[qt], 32, 64);
        }
        const float inv = 1.f / l_tot;
        const int qrow = q0 + 32 * qt + r;
        if (qrow < p.L) {
            half_t* op = p.out + b * p.osb + h * p.osh + (long long)qrow * p.osl;
#pragma unroll
            for (int dm = 0; dm < DM; ++dm)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int dd0 = 32 * dm + 8 * g4 + 4 * hh;
                    if (dd0 < p.d) {
                        const h4_t ov = {(half_t)(o[qt][dm][4 * g4] * inv), (half_t)(o[qt][dm][4 * g4 + 1] * inv),
                                         (half_t)(o[qt][dm][4 * g4 + 2] * inv), (half_t)(o[qt][dm][4 * g4 + 3] * inv)};
                        *reinterpret_cast<h4_t*>(op + dd0) = ov;
                    }
                }
        }
    }
}

template <int NK, int WAVES, int QT>
int launch(const SaParams& p0, hipStream_t st) {
    SaParams p = p0;
    p.nqb = (p.L + 32 * QT * WAVES - 1) / (32 * QT * WAVES);
    p.xcd_map = ((p.Bc * p.H) % 8 == 0) ? 1 : 0;
    const size_t lds = (size_t)2 * SaCfg<NK>::TILE_HALVES * sizeof(half_t);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&self_attn_fwd<NK, WAVES, QT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    DSC_LAUNCH((self_attn_fwd<NK, WAVES, QT>), dim3(p.Bc * p.H * p.nqb), dim3(64 * WAVES), lds, st, p);
    return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
}

int g_sa_variant = 0;            // tuning knob (dsc_debug_set_self_attn_variant): 0 = auto

template <int NK>
int launch_nk(const SaParams& p, hipStream_t st) {
    // enough workgroups to fill 256 CUs: 4 waves (128 query rows) per workgroup when that still gives >= 256 of them
    const long long wg4 = (long long)p.Bc * p.H * ((p.L + 127) / 128);
    if (NK <= 4) {
        const long long wg4x2 = (long long)p.Bc * p.H * ((p.L + 255) / 256);
        (void)wg4x2;    // measured: two query tiles per wave lose to one (220+ VGPRs -> 1 wave/SIMD): tuning variant only
        if (g_sa_variant == 2) return launch<NK, 4, (NK <= 4 ? 2 : 1)>(p, st);
    }
    // 8 waves (256 query rows share each K/V tile: half the L2->LDS traffic per MFMA) once that still leaves >= 2
    // workgroups per CU: +11 % at Bc = 16, nothing at Bc = 2 (tools/mb_sa.py)
    const long long wg8 = (long long)p.Bc * p.H * ((p.L + 255) / 256);
    if (NK <= 5 && ((g_sa_variant == 0 && wg8 >= 512) || g_sa_variant == 3)) return launch<(NK <= 5 ? NK : 3), 8, 1>(p, st);
    if (wg4 >= 256 || g_sa_variant == 1) return launch<NK, 4, 1>(p, st);
    const long long wg2 = (long long)p.Bc * p.H * ((p.L + 63) / 64);
    if (wg2 >= 128 || NK >= 6) return launch<NK, 2, 1>(p, st);   // one wave alone would need 160 staging registers at d = 160
    return launch<NK, (NK >= 6 ? 2 : 1), 1>(p, st);
}

bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
bool strides_ok(const int64_t s[3]) { return s[0] % 8 == 0 && s[1] % 8 == 0 && s[2] % 8 == 0; }

}  // namespace

extern "C" void dsc_debug_set_self_attn_variant(int v) { g_sa_variant = v; }
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
    SaParams p{};
    p.q = static_cast<const half_t*>(q); p.k = static_cast<const half_t*>(k);
    p.v = static_cast<const half_t*>(v); p.out = static_cast<half_t*>(out);
    p.Bc = Bc; p.H = H; p.L = L; p.S = S; p.d = d;
    p.scale_log2e = (scale > 0.f ? scale : 1.0f / sqrtf((float)d)) * 1.4426950408889634f;
    p.qsb = q_strides[0]; p.qsl = q_strides[1]; p.qsh = q_strides[2];
    p.ksb = k_strides[0]; p.kss = k_strides[1]; p.ksh = k_strides[2];
    p.vsb = v_strides[0]; p.vss = v_strides[1]; p.vsh = v_strides[2];
    p.osb = o_strides[0]; p.osl = o_strides[1]; p.osh = o_strides[2];
    p.stamps = g_sa_stamps;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (d <= 48) return launch_nk<3>(p, st);
    if (d <= 64) return launch_nk<4>(p, st);
    if (d <= 80) return launch_nk<5>(p, st);
    if (d <= 96) return launch_nk<6>(p, st);
    if (d <= 128) return launch_nk<8>(p, st);
    return launch_nk<10>(p, st);
}
