// fp16 projection GEMM with fused epilogues for the UNet's token-major linears (dsc_linear_f16).
//
//   out[m, n] = sum_k x[m, k] * w[n, k]  (+ bias[n]) (+ residual[m, n])                      plain
//   out[m, j] = (acc[m, j] + bias[j]) * gelu(acc[m, N/2 + j] + bias[N/2 + j])                 GEGLU (diffusers GEGLU)
//
// Replaces F.linear (+ the separate residual add / GEGLU kernels) for `to_q/to_k/to_v/to_out`, the feed-forward
// linears and the 1x1 `proj_in/proj_out/conv_shortcut` of the UNet blocks that reference
// `source/modules/u_net_condition_modify.py` takes from diffusers.  hipBLASLt serves these
// shapes (M = 128..8192 tokens, K,N = 320..10240) at a ~12.5 us latency floor per call with ~110 calls per UNet step
// (profiles/README.md); this kernel is sized for exactly that regime: few K iterations, latency first.
//
// MI355X mapping:
//   * both operands are K-contiguous (activations [M,K], torch Linear weights [N,K]), the layout MFMA fragments want:
//     D^T[n, m] = W[n, :] . X[m, :]  on v_mfma_f32_32x32x16_f16 (A = W rows, B = X^T), so a lane owns one token row m
//     with 4 consecutive output channels per register group;
//   * tiles (128 tokens x 64 channels x 64 K) go HBM/L2 -> LDS by global_load_lds_dwordx4 (no VGPR staging); the
//     LDS image is lane-linear, so the bank-conflict XOR swizzle (16-byte chunk ^= (row >> 1) & 7: a 128-byte row covers
//     half of the 64 banks, the half being the row parity, so the 16 rows {0-3,12-15,20-27} a ds_read_b128 lane group
//     touches get 16 distinct (parity, chunk) pairs) is applied to the per-lane SOURCE address and again on the
//     ds_read_b128 address (guide rule 21);
//   * a ring of STAGES LDS buffers: STAGES-1 K tiles are in flight by DMA while one is multiplied; a counted
//     `s_waitcnt vmcnt(6*(STAGES-2))` + a RAW s_barrier publish tile t without draining the younger DMAs
//     (__syncthreads() would wait vmcnt(0)); the refill of a buffer is issued right after the barrier that proves every
//     wave finished reading it (one barrier per K tile);
//   * optionally four more waves that do nothing but issue the ring's DMA (NLOAD; an LDS-DMA instruction holds its wave's
//     issue port for 100-200 cycles, 6 per K tile beside 8 MFMAs of 32 cycles): the default for the 64-row tiles;
//   * epilogue through LDS: accumulators (fp32) are transposed via an LDS stage so that bias / residual reads and the
//     fp16 stores are full 128-byte row segments instead of 8-byte column-strided pieces; ONE fp16 rounding.
#include "dsc_common.h"
#include "dsc_hip.h"
#include "gn_partials.h"

extern int g_dsc_tuning_profile;     // c_api.hip

namespace {

constexpr int BN = 64, BK = 64, T = 256;
constexpr int kBHalves = BN * BK;
constexpr int a_halves(int bm) { return bm * BK; }           // one stage: A tile (BM tokens) then B tile (64 NT weight rows)
constexpr int stage_halves(int bm, int nt = 1) { return bm * BK + nt * kBHalves; }   // BM = 128: 24 KiB (NT = 2: 32 KiB); BM = 64: 16 KiB
constexpr int epi_stride(int nt = 1) { return nt * BN + 4; } // fp32 staging row stride: 68 (132) dwords -> b128 writes conflict-free
constexpr int kEpiStride = epi_stride(1);

struct GemmParams {
    const half_t* x; const half_t* w; const half_t* bias; const half_t* res; half_t* out;
    int M, N, K;                 // N = rows of w (2x the output width for GEGLU)
    long long ldx, ldr, ldo;     // row strides (elements) of x, residual, out
    // LayerNorm folding (dsc_linear_f16 header): `x` is the UN-normalised residual stream s, w = W diag(gamma), bias =
    // beta.W + b; the epilogue applies rstd_m (acc - mu_m cvec[n]) with the row statistics of s summed from `ln_in`.
    const float* ln_in;          // [M][ln_nb][2] per-row partial (sum, sum of squares) from the GEMM that produced s
    const float* ln_c;           // [N] cvec[n] = sum_k w[n, k]
    float* ln_out;               // [M][N/64][2] partials of THIS GEMM's fp16 output rows (for the next folded LayerNorm)
    int ln_nb; float ln_inv_c, ln_eps;
    // fused QKV projection with head-major K / V (dsc_linear_qkv_f16): output columns [0, C) go to `out` (the queries, row
    // stride ldo), columns [C, 3C) to kv[which][b][h][l][dd] - each head's keys / values contiguous, so that the flash
    // kernel's 64-key tiles are plain contiguous 1-KiB DMA pieces instead of 80-byte row segments 1920 bytes apart
    half_t* kv; int kv_C, kv_H, kv_d, kv_L, kv_B;
    int xcd_remap, total;        // virtual workgroup order (see the kernel) and the number of real workgroups (tiles x splits)
    // split-K (dsc_linear_splitk_f16): workgroup (tile, split sp) multiplies K tiles [sp * kps, sp * kps + kps) and writes its raw
    // fp32 tile to ws[sp][m][n]; gemm_splitk_reduce adds the splits in order (+ bias + residual): bit-reproducible.  For the
    // few-row, long-K GEMMs of the 16x16 / 8x8 levels (M <= 512, K = 1920 ... 5120), which are bound by how many bytes of
    // WEIGHTS are in flight: 40-160 workgroups walking 30-80 K tiles each with two or three tiles in flight cannot pull 13 MB
    // of cold weights out of HBM quickly, four to eight times as many workgroups can.
    int splits, kps, tiles;
    float* ws;
    // dsc_linear_gn_f16: GroupNorm partial sums of the stored tensor (gn_partials.h): rows are pixels, gn_L of them per image
    // (gn_L % BM == 0: a row tile lies in one image), 64-column tiles only
    float* gn_part; int gn_cpg, gn_G, gn_L;
    FastDiv fd_nb, fd_tiles;     // by the column blocks per row panel and by the tiles per split (linear_impl)
    // a residual of fewer rows than M (a residual stream computed once per image under the shared CFG prefix, added to a result
    // that has a row per CFG branch): row m adds residual row m % res_rows; res_rows % BM == 0, so a row tile lies in one copy
    int res_rows; FastDiv fd_res;
    long long* stamps;           // diagnostics (dsc_debug_set_gemm_stamps): 8 x int64 per workgroup, NULL in normal calls
    int nt_store;                // GEGLU: the hidden tensor (written once, read once by the next GEMM) with non-temporal stores
};

// DMA one [ROWS x 64] K-tile into LDS: piece = 8 rows x 128 B; lane l -> row l/8, LDS chunk l%8 holds global chunk (l%8)^((row>>1)&7)
// `iw` of `NISS` issuing waves takes pieces iw, iw + NISS, ...
template <int ROWS, int NISS>
__device__ __forceinline__ void dma_tile(const half_t* g, long long ld, int row0, int rows_valid, int k0, half_t* lds,
                                         int iw, int lane) {
    static_assert((ROWS / 8) % NISS == 0, "pieces must divide among the issuing waves");
#pragma unroll
    for (int pc = 0; pc < ROWS / 8 / NISS; ++pc) {
        const int piece = pc * NISS + iw;
        const int row = piece * 8 + (lane >> 3);
        const int grow = min(row0 + row, rows_valid - 1);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        const half_t* src = g + (long long)grow * ld + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + piece * 512), 16, 0, 0);
    }
}

__device__ __forceinline__ h8_t lds_frag(const half_t* tile, int row, int kchunk) {
    return *reinterpret_cast<const h8_t*>(tile + row * BK + ((kchunk ^ ((row >> 1) & 7)) << 3));
}

// mean and 1/std of row m of the un-normalised input from the producing GEMM's per-column-block partials (fixed order)
__device__ __forceinline__ void row_stats(const GemmParams& p, long long m, float& mu, float& rs) {
    const float* src = p.ln_in + m * p.ln_nb * 2;
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < p.ln_nb; ++i) { s1 += src[2 * i]; s2 += src[2 * i + 1]; }
    mu = s1 * p.ln_inv_c;
    const float var = fmaxf(s2 * p.ln_inv_c - mu * mu, 0.f);
    rs = rsqrtf(var + p.ln_eps);
}

// GEGLU: the workgroup's 64 weight rows are 32 "hidden" rows n0h.. and the 32 matching "gate" rows N/2 + n0h..
// BM = 128 token rows per workgroup, or 64 for GEMMs with few token rows (the 16x16 level: M = 512 gives 80 workgroups of
// 128 rows on 256 CUs, each walking 20 K tiles alone; 64-row tiles double the workgroups and take 4 instead of 6 DMA pieces and
// 4 instead of 8 MFMAs per wave and K tile)
//
// NLOAD > 0: NLOAD extra waves do nothing but the tiles' LDS-DMA (an LDS-DMA instruction costs the issuing wave 100-200 cycles
// of issue time - 6 per K tile beside 8 MFMAs of 32 cycles in the plain kernel); the four computing waves then only read LDS
// and issue MFMAs between the per-tile barriers, which all waves join.
//
// NT = 2: the workgroup's tile is 128 output columns wide (wave tile 64 x 64: one LDS fragment read per MFMA instead of 1.5,
// and a third less L2 -> LDS traffic per output than two 128 x 64 tiles - the many-workgroup GEMMs are bound by exactly that).
template <bool GEGLU, int STAGES, int BM, int NLOAD, int NT = 1>
__global__ __launch_bounds__(T + 64 * NLOAD, (NLOAD ? 4 : (STAGES <= 3 ? 2 : 1))) void gemm_tn_f16(GemmParams p) {
    static_assert(BM == 128 || (BM == 64 && !GEGLU), "tile heights");
    static_assert(NT == 1 || (NT == 2 && BM == 128 && NLOAD == 0), "tile widths");
    constexpr bool LOADER = NLOAD > 0;
    constexpr int NISS = LOADER ? NLOAD : 4;                 // waves that issue DMA
    constexpr int MT = BM / 64;                              // 32-row fragments per wave
    constexpr int BNT = BN * NT;                             // weight rows per tile (GEGLU: half hidden, half gate)
    constexpr int kAHalves = a_halves(BM), kStage = stage_halves(BM, NT);
    constexpr int kPieces = (BM / 8 + BNT / 8) / NISS;       // DMA instructions per issuing wave and K tile
    constexpr int kES = epi_stride(NT);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* lds = reinterpret_cast<half_t*>(smem);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;                 // wave grid 2 (tokens) x 2 (channels): (BM/2) x 32 NT per wave
    const int nb = p.N / BNT;                                // column blocks (GEGLU: 32 NT output columns per block)
    // XCD-aware order (as in conv3x3.hip): workgroup i runs on XCD i % 8, so consecutive VIRTUAL ids - the column blocks of
    // one activation panel - go to one XCD and that panel is fetched into one L2 instead of eight
    // Each XCD has its own L2 and workgroup i runs on XCD i % 8.  xcd_remap (activation-heavy shapes, M >= 2 N): consecutive
    // VIRTUAL ids - the column blocks of one row panel - go to one XCD, so a panel is fetched into one L2 instead of eight
    // (in the step: M=8192 N=320 K=320 9.7 -> 8.6 us, N=960 18.8 -> 17.5).  Weight-heavy shapes keep the plain order: there it
    // loses (M=512 N=10240 GEGLU 27.0 -> 34.2 us), and so does the mirrored order - one XCD walking all row panels of a few
    // column blocks, an eighth of the weights per L2 - (29.1 -> 33.7).
    int bid = blockIdx.x;
    if (p.xcd_remap) {
        const int per = gridDim.x >> 3;
        bid = (bid & 7) * per + (bid >> 3);
        if (bid >= p.total) return;                          // grid padded to a multiple of 8 (before any barrier)
    }
    long long st0 = 0, st1 = 0, st2 = 0, sc0 = 0, sc1 = 0, sc2 = 0;
    if (p.stamps) { st0 = __builtin_amdgcn_s_memrealtime(); sc0 = __builtin_amdgcn_s_memtime(); }
    int kbeg = 0;                                            // first K tile of this workgroup (split-K)
    if (p.splits > 1) {
        const int sp = fdiv(bid, p.fd_tiles);
        bid -= sp * p.tiles;
        kbeg = sp * p.kps;
    }
    const int bm = fdiv(bid, p.fd_nb), bn = bid - bm * nb;   // consecutive workgroups share the activation panel
    const int m0 = bm * BM;
    const int n0 = GEGLU ? bn * (32 * NT) : bn * BNT;
    const int Nh = p.N / 2;

    const int iw = LOADER ? wave - 4 : wave;                 // index among the issuing waves
    auto issue = [&](int kt, int buf) {
        half_t* a = lds + buf * kStage;
        dma_tile<BM, NISS>(p.x, p.ldx, m0, p.M, (kbeg + kt) * BK, a, iw, lane);
        if (GEGLU) {
            // B tile rows 0..32NT-1 = w[n0 ..), rows 32NT.. = w[Nh + n0 ..): two half-tiles
            half_t* b = a + kAHalves;
#pragma unroll
            for (int pc = 0; pc < BNT / 8 / NISS; ++pc) {
                const int piece = pc * NISS + iw;            // pieces of 8 rows
                const int row = piece * 8 + (lane >> 3);
                const int grow = (row < 32 * NT ? n0 + row : Nh + n0 + row - 32 * NT);
                const int chunk = (lane & 7) ^ ((row >> 1) & 7);
                const half_t* src = p.w + (long long)grow * p.K + (kbeg + kt) * BK + chunk * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(b + piece * 512), 16, 0, 0);
            }
        } else {
            dma_tile<BNT, NISS>(p.w, p.K, n0, p.N, (kbeg + kt) * BK, a + kAHalves, iw, lane);
        }
    };

    f16x_t acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    const int nk = p.splits > 1 ? min(p.K / BK - kbeg, p.kps) : p.K / BK;
    // tile kt has landed once at most min(STAGES-2, nk-1-kt) younger tiles (kPieces DMA instructions each) are outstanding
    auto wait_tile = [&](int kt) {
        const int younger = min(STAGES - 2, nk - 1 - kt);
        if (younger >= 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * kPieces) : "memory");
        else if (younger == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * kPieces) : "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * kPieces) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kPieces) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    if constexpr (LOADER) {
        if (wave >= 4) {                                     // a loader wave: the ring's DMA and nothing else
#pragma unroll
            for (int st = 0; st < STAGES - 1; ++st)
                if (st < nk) issue(st, st);
            for (int kt = 0; kt < nk; ++kt) {
                wait_tile(kt);
                __builtin_amdgcn_s_barrier();                // publishes tile kt; every computing wave has left tile kt-1's stage
                if (kt + STAGES - 1 < nk) issue(kt + STAGES - 1, (kt + STAGES - 1) % STAGES);
            }
            __syncthreads();                                 // the epilogue's two workgroup barriers
            __syncthreads();
            if (p.gn_part) dsc_gn::gn_tile_partials_barriers();
            return;
        }
    } else {
#pragma unroll
        for (int st = 0; st < STAGES - 1; ++st)
            if (st < nk) issue(st, st);
    }
    // folded LayerNorm: thread t < 128 owns row m0 + t's (mean, 1/std); the partial-sum loads ride under the K loop
    float ln_mu = 0.f, ln_rs = 1.f;
    if (p.ln_in && threadIdx.x < BM && m0 + (int)threadIdx.x < p.M) row_stats(p, m0 + threadIdx.x, ln_mu, ln_rs);
    // column constants of this thread's epilogue chunks (its chunk column is the same in every pass): bias and, for a folded
    // LayerNorm, the weight-row sums - fetched here so that their latency rides under the K loop, not in the epilogue
    constexpr int CR = (GEGLU ? 4 : 8) * NT;                 // 8-column output chunks per tile row
    const int ech = threadIdx.x % CR;
    h8_t bpre0 = {0, 0, 0, 0, 0, 0, 0, 0}, bpre1 = {0, 0, 0, 0, 0, 0, 0, 0};
    f4x_t cpre[GEGLU ? 4 : 2];
#pragma unroll
    for (int i = 0; i < (GEGLU ? 4 : 2); ++i) cpre[i] = f4x_t{0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
        bpre0 = *reinterpret_cast<const h8_t*>(p.bias + n0 + ech * 8);
        if (GEGLU) bpre1 = *reinterpret_cast<const h8_t*>(p.bias + Nh + n0 + ech * 8);
    }
    if (p.ln_in) {
        cpre[0] = *reinterpret_cast<const f4x_t*>(p.ln_c + n0 + ech * 8);
        cpre[1] = *reinterpret_cast<const f4x_t*>(p.ln_c + n0 + ech * 8 + 4);
        if (GEGLU) {
            cpre[2] = *reinterpret_cast<const f4x_t*>(p.ln_c + Nh + n0 + ech * 8);
            cpre[3] = *reinterpret_cast<const f4x_t*>(p.ln_c + Nh + n0 + ech * 8 + 4);
        }
    }
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt % STAGES;
        if (!LOADER) wait_tile(kt);
        else asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // raw: publishes tile kt, proves tile kt-1's reads are done
        asm volatile("" ::: "memory");
        if (p.stamps && kt == 0) { st1 = __builtin_amdgcn_s_memrealtime(); sc1 = __builtin_amdgcn_s_memtime(); }
        if (!LOADER && kt + STAGES - 1 < nk) issue(kt + STAGES - 1, (kt + STAGES - 1) % STAGES);
        const half_t* a = lds + buf * kStage;
        const half_t* b = a + kAHalves;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            h8_t wf[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[nt] = lds_frag(b, wn * (32 * NT) + nt * 32 + r, 2 * ks + hh);   // W[n][16 ks + 8 hh ..]
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const h8_t xf = lds_frag(a, wm * (BM / 2) + mt * 32 + r, 2 * ks + hh);   // X[m][...]
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma_32x32x16(wf[nt], xf, acc[mt][nt]);
            }
        }
    }
    // residual rows of this thread's four output chunks: issued before the staging pass so that their latency (HBM /
    // Infinity Cache: the residual stream was written by an earlier kernel) hides under it
    constexpr int NCH = BM * CR / T;                         // output chunks per thread: 4 or 2 (8 / 4 with NT = 2)
    const int mres0 = p.res_rows ? m0 - fdiv(m0, p.fd_res) * p.res_rows : m0;     // (scalar) first residual row of this tile
    h8_t rpre[NCH];
    if (!GEGLU) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int idx = threadIdx.x + c * T, row = idx / CR, ch = idx % CR;
            rpre[c] = h8_t{0, 0, 0, 0, 0, 0, 0, 0};
            if (p.res && m0 + row < p.M) rpre[c] = *reinterpret_cast<const h8_t*>(p.res + (long long)(mres0 + row) * p.ldr + n0 + ch * 8);
        }
    }
    __syncthreads();                                         // all MFMA operand reads done: LDS becomes the epilogue stage
    if (p.stamps) { st2 = __builtin_amdgcn_s_memrealtime(); sc2 = __builtin_amdgcn_s_memtime(); }

    // ---- epilogue stage: stage[m][n] fp32; acc[mt] element i <-> n = wn*32 + (i&3) + 8(i>>2) + 4hh, m = wm*64 + mt*32 + r
    float* stage = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f4x_t v = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
                *reinterpret_cast<f4x_t*>(stage + (wm * (BM / 2) + mt * 32 + r) * kES + wn * (32 * NT) + nt * 32 + 8 * g + 4 * hh) = v;
            }
    float* rowst = stage + BM * kES;                         // [BM][2] (mu, rstd) of the folded LayerNorm, behind the stage
    if (p.ln_in && threadIdx.x < BM) { rowst[2 * threadIdx.x] = ln_mu; rowst[2 * threadIdx.x + 1] = ln_rs; }
    __syncthreads();
    if (GEGLU) {
        // output tile 128 x 32 NT: CR chunks of 8 columns per row, NCH per thread
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int idx = threadIdx.x + c * T, row = idx / CR, ch = idx % CR;
            float mu = 0.f, rs = 1.f;
            if (p.ln_in) { mu = rowst[2 * row]; rs = rowst[2 * row + 1]; }
            if (m0 + row < p.M) {
                const float* sp = stage + row * kES + ch * 8;
                const h8_t bh = bpre0, bg = bpre1;                 // ch == ech: T is a multiple of the chunks per row
                h8_t o;
                float ch_[8], cg_[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) { ch_[j] = cpre[0][j]; ch_[4 + j] = cpre[1][j]; cg_[j] = cpre[GEGLU ? 2 : 0][j]; cg_[4 + j] = cpre[GEGLU ? 3 : 1][j]; }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float ah = sp[j], ag = sp[32 * NT + j];
                    if (p.ln_in) {
                        ah = rs * (ah - mu * ch_[j]);
                        ag = rs * (ag - mu * cg_[j]);
                    }
                    const float hid = ah + (float)bh[j];
                    const float gate = ag + (float)bg[j];
                    // diffusers: proj output is an fp16 tensor; hidden * gelu(gate) with gelu's result in fp16
                    o[j] = (half_t)((float)(half_t)hid * (float)(half_t)gelu_erf((float)(half_t)gate));
                }
                h8_t* dst = reinterpret_cast<h8_t*>(p.out + (long long)(m0 + row) * p.ldo + n0 + ch * 8);
                if (p.nt_store) __builtin_nontemporal_store(o, dst);
                else *dst = o;
            }
        }
    } else {
        // output tile BM x 64: 8 chunks per row -> 1024 (512) chunks, 4 (2) per thread; a row's 8 chunks = one 128-B segment
        const bool scat = p.kv != nullptr && n0 + ech * 8 >= p.kv_C;   // per chunk column (uniform over a 64-column block)
        long long kv_col = 0;
        int kv_b0 = 0, kv_l0 = 0;
        if (scat) {                                               // this thread's chunk column -> (k | v, head, channel)
            int cc = n0 - p.kv_C + ech * 8;
            const int which = cc >= p.kv_C ? 1 : 0;
            cc -= which * p.kv_C;
            const int hd = cc / p.kv_d;
            kv_col = ((long long)(which * p.kv_B) * p.kv_H + hd) * p.kv_L * p.kv_d + (cc - hd * p.kv_d);
            kv_b0 = m0 / p.kv_L;                                  // scalar: the block's first token row
            kv_l0 = m0 - kv_b0 * p.kv_L;
        }
        if (p.ws) {                                               // split-K: the raw fp32 tile of this split, nothing else
            float* wsp = p.ws + (long long)(kbeg / p.kps) * p.M * p.N;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int idx = threadIdx.x + c * T, row = idx / CR, ch = idx % CR;
                if (m0 + row < p.M) {
                    const float* sp = stage + row * kES + ch * 8;
                    float* dst = wsp + (long long)(m0 + row) * p.N + n0 + ch * 8;
                    *reinterpret_cast<f4x_t*>(dst) = *reinterpret_cast<const f4x_t*>(sp);
                    *reinterpret_cast<f4x_t*>(dst + 4) = *reinterpret_cast<const f4x_t*>(sp + 4);
                }
            }
            return;
        }
        float gs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int idx = threadIdx.x + c * T, row = idx / CR, ch = idx % CR;
            const bool live = m0 + row < p.M;                    // a row's 8 chunks of one 64-column block sit in 8 consecutive lanes
            float s1 = 0.f, s2 = 0.f, mu = 0.f, rs = 1.f;
            if (p.ln_in) { mu = rowst[2 * row]; rs = rowst[2 * row + 1]; }
            if (live) {
                const float* sp = stage + row * kES + ch * 8;
                const h8_t bv = bpre0, rv = rpre[c];               // ch == ech
                h8_t o;
                float cv_[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) { cv_[j] = cpre[0][j]; cv_[4 + j] = cpre[1][j]; }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float a = sp[j];
                    if (p.ln_in) a = rs * (a - mu * cv_[j]);
                    o[j] = (half_t)(a + (float)bv[j] + (float)rv[j]);
                    const float f = (float)o[j];                 // statistics of the fp16 row, as the LayerNorm kernel takes them
                    s1 += f; s2 += f * f;
                    gs[j] += f; gq[j] += f * f;                  // ... and per channel over this thread's rows, for a GroupNorm
                }
                if (scat) {
                    int bb = kv_b0, ll = kv_l0 + row;
                    while (ll >= p.kv_L) { ll -= p.kv_L; ++bb; }
                    *reinterpret_cast<h8_t*>(p.kv + kv_col + ((long long)bb * p.kv_H * p.kv_L + ll) * p.kv_d) = o;
                } else {
                    *reinterpret_cast<h8_t*>(p.out + (long long)(m0 + row) * p.ldo + n0 + ch * 8) = o;
                }
            }
            if (p.ln_out) {                                       // wave-uniform
#pragma unroll
                for (int o2 = 1; o2 < 8; o2 <<= 1) { s1 += __shfl_xor(s1, o2, 64); s2 += __shfl_xor(s2, o2, 64); }
                if (live && (ch & 7) == 0) {
                    float* dst = p.ln_out + ((long long)(m0 + row) * (nb * NT) + bn * NT + (ch >> 3)) * 2;
                    dst[0] = s1; dst[1] = s2;
                }
            }
        }
        if constexpr (NT == 1) {
            if (p.gn_part) {                                      // row tile bm = pixel tile (m0 % gn_L) / BM of image m0 / gn_L
                const int gb = m0 / p.gn_L, gpt = (m0 - gb * p.gn_L) / BM;
                float* dst = p.gn_part + ((long long)gb * (p.gn_L / BM) + gpt) * p.gn_G * 4;
                dsc_gn::gn_tile_partials(gs, gq, stage + BM * kES + 2 * BM, n0, p.gn_cpg, p.gn_G, dst);
            }
        }
    }
    if (p.stamps && threadIdx.x == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the output stores have left
        long long* o = p.stamps + (long long)blockIdx.x * 8;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = __builtin_amdgcn_s_memrealtime();
        o[4] = sc1 - sc0; o[5] = sc2 - sc1; o[6] = __builtin_amdgcn_s_memtime() - sc2; o[7] = ((long long)xcc << 32) | hwid;
    }
}

// out = sum over the splits (in split order) + bias + residual, one fp16 rounding (the second launch of dsc_linear_splitk_f16)
__global__ __launch_bounds__(256) void gemm_splitk_reduce(const float* ws, const half_t* bias, const half_t* res, half_t* out,
                                                          long long M, int N, int splits, long long ldr, long long ldo, FastDiv fd_cv) {
    const int cv = N >> 3;
    const int idx = blockIdx.x * 256 + threadIdx.x;           // (M <= 512 rows here: 32-bit, and the division a multiply)
    if (idx >= (int)M * cv) return;
    const long long m = fdiv(idx, fd_cv);
    const int n = (idx - (int)m * cv) * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < splits; ++k) {
        const float* src = ws + ((long long)k * M + m) * N + n;
        const f4x_t a = *reinterpret_cast<const f4x_t*>(src), b = *reinterpret_cast<const f4x_t*>(src + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[j] += a[j]; acc[4 + j] += b[j]; }
    }
    h8_t bv = {0, 0, 0, 0, 0, 0, 0, 0}, rv = {0, 0, 0, 0, 0, 0, 0, 0};
    if (bias) bv = *reinterpret_cast<const h8_t*>(bias + n);
    if (res) rv = *reinterpret_cast<const h8_t*>(res + m * ldr + n);
    h8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)(acc[j] + (float)bv[j] + (float)rv[j]);
    *reinterpret_cast<h8_t*>(out + m * ldo + n) = o;
}

bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

long long* g_gemm_stamps = nullptr;
int g_gemm_stages = 0;           // diagnostics (dsc_debug_set_gemm_stages): 0 = default, 2 / 3 = forced ring depth
int g_gemm_bm = 0;               // ... (stages / 10 of the same call): 0 = default, 64 / 128 = forced tile height
int g_gemm_loaders = 0;          // ... (stages / 10000 % 10): 0 = default (loader waves for the 64-row tiles), 4 = for every tile, 9 = never
int g_gemm_xcd = 0;              // ... (stages / 1000000): 0 = by shape, 1 = plain blockIdx order, 2 = the XCD-aware order everywhere
int g_gemm_wide_min = 512;       // plain GEMMs take 128-column tiles when that leaves at least this many workgroups (linear_impl)
int g_gemm_ntstore = 0;          // ... (stages / 10000000 % 10): 1 = non-temporal stores of the GEGLU output
int g_gemm_nt = 0;               // ... (stages / 100000): 0 = default (128-column tiles for the GEGLU GEMMs), 1 = never, 2 = wherever N allows

}  // namespace

extern "C" void dsc_debug_set_gemm_stamps(void* device_buffer) { g_gemm_stamps = static_cast<long long*>(device_buffer); }
extern "C" void dsc_debug_set_gemm_stages(int stages) {
    if (stages < 0) { g_gemm_wide_min = -stages; return; }      // (negative: the workgroup threshold of the 128-column rule)
    // stages % 10: ring depth (2, 3; else default); stages / 10: tile height (64, 128; else default) - e.g. 640 + 3
    const int bm = (stages / 10) % 1000, st = stages % 10;
    g_gemm_stages = (st == 2 || st == 3) ? st : 0;
    g_gemm_bm = (bm == 64 || bm == 128) ? bm : 0;
    const int ld = stages / 10000 % 10, nt = stages / 100000 % 10;
    g_gemm_xcd = stages / 1000000 % 10 <= 2 ? stages / 1000000 % 10 : 0;
    g_gemm_ntstore = stages / 10000000 % 10 == 1 ? 1 : 0;
    g_gemm_loaders = (ld == 4 || ld == 9) ? ld : 0;
    g_gemm_nt = (nt == 1 || nt == 2) ? nt : 0;
}

extern "C" int dsc_linear_ln_f16(const void* x, const void* w, const void* bias, const void* residual, void* out,
                                 int64_t M, int N, int K, int64_t ldx, int64_t ldr, int64_t ldo, int geglu,
                                 const float* ln_in, int ln_nb, const float* ln_cvec, float ln_eps, float* ln_out,
                                 int dtype, void* stream);

extern "C" int dsc_linear_f16(const void* x, const void* w, const void* bias, const void* residual, void* out,
                              int64_t M, int N, int K, int64_t ldx, int64_t ldr, int64_t ldo, int geglu, int dtype,
                              void* stream) {
    return dsc_linear_ln_f16(x, w, bias, residual, out, M, N, K, ldx, ldr, ldo, geglu, nullptr, 0, nullptr, 0.f, nullptr,
                             dtype, stream);
}

namespace {
struct GnArgs { float* part; int groups; int rows_per_image; };
int linear_impl(const void* x, const void* w, const void* bias, const void* residual, void* out,
                int64_t M, int N, int K, int64_t ldx, int64_t ldr, int64_t ldo, int geglu,
                const float* ln_in, int ln_nb, const float* ln_cvec, float ln_eps, float* ln_out,
                int dtype, void* stream, void* kv_out, int heads, int seq_len, int splits = 1, float* ws = nullptr,
                const GnArgs* gn = nullptr);
int gemm_tile_rows(int64_t M, int N, bool geglu);
}

extern "C" int dsc_linear_ln_f16(const void* x, const void* w, const void* bias, const void* residual, void* out,
                                 int64_t M, int N, int K, int64_t ldx, int64_t ldr, int64_t ldo, int geglu,
                                 const float* ln_in, int ln_nb, const float* ln_cvec, float ln_eps, float* ln_out,
                                 int dtype, void* stream) {
    return linear_impl(x, w, bias, residual, out, M, N, K, ldx, ldr, ldo, geglu, ln_in, ln_nb, ln_cvec, ln_eps, ln_out, dtype,
                       stream, nullptr, 0, 0);
}

extern "C" int dsc_linear_qkv_f16(const void* x, const void* w, const void* bias, void* q_out, void* kv_out,
                                  int64_t M, int C, int K, int64_t ldx, int64_t ldq, int heads, int seq_len,
                                  const float* ln_in, int ln_nb, const float* ln_cvec, float ln_eps, int dtype, void* stream) {
    if (!kv_out || heads <= 0 || seq_len <= 0 || C <= 0) return DSC_ERR_BAD_ARG;
    if (C % 64 != 0 || C % heads != 0 || (C / heads) % 8 != 0 || M % seq_len != 0) return DSC_ERR_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(kv_out) & 15) return DSC_ERR_UNSUPPORTED;
    return linear_impl(x, w, bias, nullptr, q_out, M, 3 * C, K, ldx, 0, ldq, 0, ln_in, ln_nb, ln_cvec, ln_eps, nullptr, dtype,
                       stream, kv_out, heads, seq_len);
}

namespace {
int linear_impl(const void* x, const void* w, const void* bias, const void* residual, void* out,
                int64_t M, int N, int K, int64_t ldx, int64_t ldr, int64_t ldo, int geglu,
                const float* ln_in, int ln_nb, const float* ln_cvec, float ln_eps, float* ln_out,
                int dtype, void* stream, void* kv_out, int heads, int seq_len, int splits, float* ws, const GnArgs* gn) {
    if ((ln_in && (!ln_cvec || ln_nb <= 0)) || (ln_out && geglu)) return DSC_ERR_BAD_ARG;
    if (ln_in && !al16(ln_cvec)) return DSC_ERR_UNSUPPORTED;       // read as float4 pairs
    if (!x || !w || !out || M <= 0 || N <= 0 || K <= 0) return DSC_ERR_BAD_ARG;
    if (dtype != DSC_F16) return DSC_ERR_UNSUPPORTED;
    // ldr: low 32 bits = the residual's row stride; high 32 bits, when set, = its row count R < M (include/dsc_hip.h)
    const int64_t res_rows = residual ? (ldr >> 32) : 0;
    ldr &= 0xffffffffll;
    if (res_rows < 0 || (res_rows > 0 && (M % res_rows != 0 || res_rows % 128 != 0 || splits > 1))) return DSC_ERR_UNSUPPORTED;
    if (K % BK != 0 || N % BN != 0 || ldx % 8 != 0 || ldo % 8 != 0 || (residual && ldr % 8 != 0)) return DSC_ERR_UNSUPPORTED;
    if (geglu && (!bias || residual || (N / 2) % 32 != 0)) return DSC_ERR_UNSUPPORTED;
    if (!al16(x) || !al16(w) || !al16(out) || (bias && !al16(bias)) || (residual && !al16(residual))) return DSC_ERR_UNSUPPORTED;
    if (M > (1ll << 30)) return DSC_ERR_UNSUPPORTED;
    GemmParams p{};
    p.stamps = g_gemm_stamps;
    p.nt_store = g_gemm_ntstore;
    p.x = static_cast<const half_t*>(x); p.w = static_cast<const half_t*>(w);
    p.bias = static_cast<const half_t*>(bias); p.res = static_cast<const half_t*>(residual);
    p.out = static_cast<half_t*>(out);
    p.M = (int)M; p.N = N; p.K = K; p.ldx = ldx; p.ldr = ldr; p.ldo = ldo;
    p.res_rows = (int)res_rows; p.fd_res = make_fastdiv(res_rows > 0 ? res_rows : 1, M);
    p.ln_in = ln_in; p.ln_c = ln_cvec; p.ln_out = ln_out; p.ln_nb = ln_nb; p.ln_inv_c = 1.f / (float)K; p.ln_eps = ln_eps;
    if (kv_out) {
        p.kv = static_cast<half_t*>(kv_out);
        p.kv_C = N / 3; p.kv_H = heads; p.kv_d = N / 3 / heads; p.kv_L = seq_len; p.kv_B = (int)(M / seq_len);
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    static bool attr_set = false;
    if (!attr_set) {
        const void* fns[] = {reinterpret_cast<const void*>(&gemm_tn_f16<true, 3, 128, 0>), reinterpret_cast<const void*>(&gemm_tn_f16<false, 3, 128, 0>),
                             reinterpret_cast<const void*>(&gemm_tn_f16<true, 2, 128, 0>), reinterpret_cast<const void*>(&gemm_tn_f16<false, 2, 128, 0>),
                             reinterpret_cast<const void*>(&gemm_tn_f16<false, 3, 64, 0>), reinterpret_cast<const void*>(&gemm_tn_f16<false, 2, 64, 0>),
                             reinterpret_cast<const void*>(&gemm_tn_f16<true, 3, 128, 4>), reinterpret_cast<const void*>(&gemm_tn_f16<false, 3, 128, 4>),
                             reinterpret_cast<const void*>(&gemm_tn_f16<false, 3, 64, 4>),
                             reinterpret_cast<const void*>(&gemm_tn_f16<true, 2, 128, 0, 2>), reinterpret_cast<const void*>(&gemm_tn_f16<false, 2, 128, 0, 2>)};
        for (const void* f : fns) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    // Tile height: 64 token rows when 128-row tiles would leave the chip half empty (fewer than 192 workgroups - the 16x16
    // level's C->C GEMMs: M = 512, N = 1280 -> 80 workgroups walking 20 K tiles each) and the rows allow it; GEGLU keeps 128.
    int nb = N / BN;
    int bm = gemm_tile_rows(M, N, geglu != 0);
    if (gn) {
        if (geglu || kv_out || splits > 1 || N % gn->groups != 0 || N / gn->groups > 64 || gn->rows_per_image % bm != 0 ||
            M % gn->rows_per_image != 0)
            return DSC_ERR_UNSUPPORTED;
        p.gn_part = gn->part; p.gn_G = gn->groups; p.gn_cpg = N / gn->groups; p.gn_L = gn->rows_per_image;
    }
    // (measured, tools/mb_gemm.py: M=512 N=1280 K=1280 11.8 -> 8.1 us, M=8192 N=320 K=320 9.1 -> 7.5, M=8192 N=320 K=1280 24.4 ->
    // 20.7; at 480+ workgroups of 128 rows the taller tile wins: M=2048 N=1920 K=640 12.3 vs 13.5)
    const int mb = (int)((M + bm - 1) / bm);
    // 3 stages = 72 KiB -> two workgroups per CU.  A deeper ring (6 stages) was measured and changes nothing: a K tile
    // costs ~1000 cycles because a CU ingests only ~24 B/cycle from L2 (24 KiB per tile), not because of DMA latency -
    // the kernel is L2->LDS bandwidth bound at this tile size (43 FLOP per staged byte), which caps it near 25 % of the
    // MFMA peak; the dispatch in ops.linear therefore sends long-K shapes to hipBLASLt's larger macro-tiles.
    p.xcd_remap = g_gemm_xcd == 0 ? (M >= 2ll * N ? 1 : 0) : (g_gemm_xcd == 2 ? 1 : 0);
    p.tiles = mb * nb;
    p.splits = 1;
    if (splits > 1) {                                            // split-K: raw fp32 tiles to ws, no epilogue operands
        const int nkt = K / BK;
        p.kps = (nkt + splits - 1) / splits;
        p.splits = (nkt + p.kps - 1) / p.kps;
        p.ws = ws;
        p.bias = nullptr; p.res = nullptr;
    }
    p.total = p.tiles * p.splits;
    p.fd_nb = make_fastdiv(nb, p.total);
    p.fd_tiles = make_fastdiv(p.tiles, p.total);
    const dim3 grid(p.xcd_remap ? ((p.total + 7) / 8) * 8 : p.total), block(T);
    // Two stages (48 KiB: three workgroups per CU instead of two) for grids of many workgroups per CU: with K = 320 / 640 the
    // K loop is a third of a workgroup's time (prologue DMA chain, LayerNorm / GEGLU epilogue), and a third co-resident
    // workgroup overlaps those parts (tools/mb_gemm.py)
    // measured: >= 300 workgroups -> two stages win (GEGLU N=2560 K=320 33.7 -> 30.9 us, N=5120 K=640 25.4 -> 22.6,
    // M=512 N=10240 K=1280 26.5 -> 22.2); fewer workgroups with a long K loop want the deeper ring (N=640 K=2560 21.3 vs 27.0)
    // ... and only while the K loop is short: K = 2560 / 5120 want the deeper ring whatever the grid (23.9 vs 22.6, 38.9 vs 27.9)
    // In the STEP (cold weights: tools/step_breakdown.py on a kernel trace, gemm_tn per step) - 128-row tiles + 3 stages
    // everywhere 1.65 ms, + this two-stage rule 1.59, 64-row tiles for the small grids with 3 stages 1.53, 64-row tiles WITH two
    // stages 1.68: the short ring loses its prefetch depth exactly where every weight tile comes from HBM, which the
    // back-to-back micro-benchmark (warm weights) ranks the other way round.  So: two stages only for the 128-row many-workgroup grids.
    // (5- and 8-stage rings for the 64-row tiles of the small grids, in the step: M=512 N=1280 K=1280 10.8 -> 11.0 / 11.2 us,
    // M=2048 N=640 K=640 9.0 -> 9.9: the K loop is bound by what one CU ingests from L2, not by DMA latency, cold weights or not)
    const int stages = g_gemm_stages ? g_gemm_stages : ((K <= 1280 && bm == 128 && mb * nb >= 300 && p.splits == 1) ? 2 : 3);
    // 128-column tiles (two stages of 32 KiB, two workgroups per CU) where the grid still gives every CU a workgroup: the
    // GEGLU GEMMs (N/2 = 1280 / 2560 / 5120 -> 1280 / 640 / 320 workgroups)
    const bool wide_ok = bm == 128 && N % 128 == 0 && (!geglu || (N / 2) % 64 == 0) && (g_gemm_stages == 0 || g_gemm_stages == 2) && g_gemm_loaders != 4;
    // ... and, round 4, every plain GEMM whose grid stays full with them (throughput tier: 8 images per generation, coalesced
    // requests): 128 x 128 tiles halve the activation panel's trips through L2 -> LDS.  Measured end to end (tools/ab_benchk.sh,
    // one box, images/s at 8 / 4 / 2 images per generation): never 16.66 / 14.15 / 11.76, grids of >= 512 workgroups 16.75 /
    // 14.62 / 11.78, >= 256: 16.61 / 14.37 / 11.84, >= 64: 16.12 / 14.70 / 11.81 - a +3 % at 4 images, noise elsewhere; batch 1 has
    // no such grid (same bits either way: the accumulation order per element does not depend on the tile).  g_gemm_wide_min
    // workgroups (dsc_debug_set_gemm_stages(-n) sets it).  Not with GroupNorm partial sums (gn_tile_partials is written for
    // 64-column tiles).
    const long long wide_wgs = (long long)mb * (N / 128);
    const bool wide = wide_ok && p.splits == 1 && !gn &&
                      (g_gemm_nt == 2 || (g_gemm_nt == 0 && (geglu ? wide_wgs >= 256 : wide_wgs >= g_gemm_wide_min)));
    if (wide) {
        nb = N / 128;
        p.total = mb * nb;
        p.fd_nb = make_fastdiv(nb, p.total);
        size_t wl = (size_t)2 * stage_halves(128, 2) * sizeof(half_t);
        const size_t we = (size_t)128 * epi_stride(2) * sizeof(float) + (size_t)128 * 2 * sizeof(float);
        if (wl < we) wl = we;
        const dim3 wgrid(p.xcd_remap ? ((mb * nb + 7) / 8) * 8 : mb * nb);
        if (geglu) DSC_LAUNCH((gemm_tn_f16<true, 2, 128, 0, 2>), wgrid, block, wl, st, p);
        else DSC_LAUNCH((gemm_tn_f16<false, 2, 128, 0, 2>), wgrid, block, wl, st, p);
        return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
    }
    // the fp32 epilogue stage (bm x 68 floats + the row statistics) reuses the ring
    size_t lds = (size_t)stages * stage_halves(bm) * sizeof(half_t);
    const size_t epi = (size_t)bm * kEpiStride * sizeof(float) + (size_t)bm * 2 * sizeof(float) +
                       (gn ? (size_t)dsc_gn::kScratchFloats * sizeof(float) : 0);
    if (lds < epi) lds = epi;
    // Four DMA-only loader waves beside the four computing ones: in the step 1-3 % on the 64-row-tile GEMMs (10.7 -> 10.6,
    // 9.6 -> 9.3, 8.9 -> 8.7 us; 5-20 % back to back with warm weights, tools/chk_gemm_loader.py), while the 128-row tiles
    // of the big grids keep the two-stage ring and three workgroups per CU (with loaders and three stages: GEGLU 29.7 -> 31.9 us)
    // - under DSC_TUNE_LATENCY only: eight-wave workgroups cost the other stream's kernels their wave slots
    if (g_gemm_loaders == 4 || (g_gemm_loaders == 0 && bm == 64 && stages == 3 && g_dsc_tuning_profile == DSC_TUNE_LATENCY)) {
        lds = (size_t)3 * stage_halves(bm) * sizeof(half_t);
        if (lds < epi) lds = epi;
        const dim3 block8(T + 256);
        if (bm == 64) DSC_LAUNCH((gemm_tn_f16<false, 3, 64, 4>), grid, block8, lds, st, p);
        else if (geglu) DSC_LAUNCH((gemm_tn_f16<true, 3, 128, 4>), grid, block8, lds, st, p);
        else DSC_LAUNCH((gemm_tn_f16<false, 3, 128, 4>), grid, block8, lds, st, p);
    } else if (bm == 64) {
        if (stages == 2) DSC_LAUNCH((gemm_tn_f16<false, 2, 64, 0>), grid, block, lds, st, p);
        else DSC_LAUNCH((gemm_tn_f16<false, 3, 64, 0>), grid, block, lds, st, p);
    } else if (stages == 2) {
        if (geglu) DSC_LAUNCH((gemm_tn_f16<true, 2, 128, 0>), grid, block, lds, st, p);
        else DSC_LAUNCH((gemm_tn_f16<false, 2, 128, 0>), grid, block, lds, st, p);
    } else {
        if (geglu) DSC_LAUNCH((gemm_tn_f16<true, 3, 128, 0>), grid, block, lds, st, p);
        else DSC_LAUNCH((gemm_tn_f16<false, 3, 128, 0>), grid, block, lds, st, p);
    }
    return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
}
}  // namespace

namespace {
// tile height: 64 token rows when 128-row tiles would leave the chip half empty (see linear_impl); GEGLU keeps 128
int gemm_tile_rows(int64_t M, int N, bool geglu) {
    int bm = 128;
    if (!geglu && (long long)((M + 127) / 128) * (N / BN) <= 320 && g_gemm_bm != 128) bm = 64;
    if (g_gemm_bm == 64 && !geglu) bm = 64;
    return bm;
}
}  // namespace

// partial rows per image of dsc_linear_gn_f16 (row tiles per image), 0 when the shape is not covered
extern "C" int dsc_linear_gn_rows(int64_t M, int N, int K, int rows_per_image, int groups) {
    if (M <= 0 || N <= 0 || K <= 0 || rows_per_image <= 0 || groups <= 0 || K % BK != 0 || N % BN != 0 || N % groups != 0 ||
        N / groups > 64 || N / groups < 2 || M % rows_per_image != 0)     // (>= 2 channels per group: 32 group slots per tile, gn_partials.h)
        return 0;
    const int bm = gemm_tile_rows(M, N, false);
    if (rows_per_image % bm != 0 || rows_per_image / bm > 128) return 0;
    return rows_per_image / bm;
}

extern "C" int dsc_linear_gn_f16(const void* x, const void* w, const void* bias, const void* residual, void* out,
                                 int64_t M, int N, int K, int64_t ldx, int64_t ldr, int64_t ldo, int rows_per_image,
                                 float* gn_part, int groups, int dtype, void* stream) {
    if (!gn_part || (reinterpret_cast<uintptr_t>(gn_part) & 7)) return DSC_ERR_BAD_ARG;
    if (!dsc_linear_gn_rows(M, N, K, rows_per_image, groups)) return DSC_ERR_UNSUPPORTED;
    const GnArgs gn{gn_part, groups, rows_per_image};
    return linear_impl(x, w, bias, residual, out, M, N, K, ldx, ldr, ldo, 0, nullptr, 0, nullptr, 0.f, nullptr, dtype, stream,
                       nullptr, 0, 0, 1, nullptr, &gn);
}

// K tiles per split: as many splits as bring the grid to ~512 workgroups, at least 8 K tiles each
static int splitk_auto(int64_t M, int N, int K) {
    const int nkt = K / BK;
    const long long tiles = ((M + 63) / 64) * (N / BN);
    long long want = tiles > 0 ? 512 / tiles : 1;
    if (want > nkt / 8) want = nkt / 8;
    if (want < 1) want = 1;
    return (int)want;
}

extern "C" size_t dsc_linear_splitk_workspace_bytes(int64_t M, int N, int K, int splits) {
    if (M <= 0 || N <= 0 || K <= 0 || K % BK != 0 || N % BN != 0) return 0;
    if (splits <= 0) splits = splitk_auto(M, N, K);
    const int nkt = K / BK, kps = (nkt + splits - 1) / splits;
    splits = (nkt + kps - 1) / kps;
    return splits > 1 ? (size_t)splits * M * N * sizeof(float) : 0;
}

extern "C" int dsc_linear_splitk_f16(const void* x, const void* w, const void* bias, const void* residual, void* out,
                                     int64_t M, int N, int K, int64_t ldx, int64_t ldr, int64_t ldo, int splits,
                                     void* workspace, size_t workspace_bytes, int dtype, void* stream) {
    if (!x || !w || !out || M <= 0 || N <= 0 || K <= 0) return DSC_ERR_BAD_ARG;
    if (dtype != DSC_F16 || K % BK != 0 || N % BN != 0) return DSC_ERR_UNSUPPORTED;
    if (splits <= 0) splits = splitk_auto(M, N, K);
    const int nkt = K / BK, kps = (nkt + splits - 1) / splits;
    splits = (nkt + kps - 1) / kps;
    if (splits <= 1) return dsc_linear_f16(x, w, bias, residual, out, M, N, K, ldx, ldr, ldo, 0, dtype, stream);
    if (residual && (ldr >> 32) != 0) return DSC_ERR_UNSUPPORTED;      // (a wrapped residual: the unsplit kernel only)
    const size_t need = (size_t)splits * M * N * sizeof(float);
    if (!workspace || workspace_bytes < need || !al16(workspace)) return DSC_ERR_WORKSPACE;
    if ((bias && !al16(bias)) || (residual && (!al16(residual) || ldr % 8 != 0)) || ldo % 8 != 0 || !al16(out)) return DSC_ERR_UNSUPPORTED;
    const int rc = linear_impl(x, w, nullptr, nullptr, out, M, N, K, ldx, 0, ldo, 0, nullptr, 0, nullptr, 0.f, nullptr, dtype, stream,
                               nullptr, 0, 0, splits, static_cast<float*>(workspace));
    if (rc != DSC_OK) return rc;
    const long long n8 = M * (N / 8);
    if (n8 >= (1ll << 30)) return DSC_ERR_UNSUPPORTED;          // the reduce kernel indexes with 32 bits
    DSC_LAUNCH(gemm_splitk_reduce, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
               static_cast<const float*>(workspace), static_cast<const half_t*>(bias), static_cast<const half_t*>(residual),
               static_cast<half_t*>(out), (long long)M, N, splits, (long long)ldr, (long long)ldo, make_fastdiv(N / 8, n8 + 256));
    return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
}
