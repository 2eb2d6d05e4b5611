// 3x3 / stride 1 / pad 1 convolution over channels-last fp16 activations (dsc_conv3x3_nhwc_f16) - the ResnetBlock2D
// conv1/conv2 and the Upsample2D conv of the UNet that reference `source/modules/u_net_condition_modify.py` builds from
// diffusers blocks (49 of the 52 3x3 convolutions of an SD1.5 step; the three stride-2
// Downsample2D convolutions and the 4-channel conv_in / conv_out stay on MIOpen).
//
// Why not MIOpen's NHWC implicit GEMM: at batch 1 (+CFG) the 49 convolutions are 33 % of the UNet step and every one of
// them runs at 270-300 TFLOP/s (profiles/README.md).  A CU ingests only ~24-30 B/cycle from L2
// (MI355X_MICROARCH.md "Indexed rows: gather into LDS"), and a plain implicit GEMM re-fetches every input pixel once per
// filter tap.  This kernel stages the (8 x TW)+halo input patch of one 64-channel slice in LDS ONCE and serves all nine
// taps from it, so per tap only the 8 KiB weight tile crosses L2->LDS: 43 -> 97 FLOP per ingested byte.
//
//   tile       128 output pixels (NSB sub-blocks of 8 x TW pixels; TW = 16, or 8 for 8-wide images) x 64 output channels
//   K loop     input-channel slices of 64 (outer) x 9 taps (inner); optional split over slices (split-K) for the
//              low-resolution levels, partial sums in fp32 to a workspace, reduced in slice order by a second launch -
//              bit-reproducible, unlike MIOpen's atomic split-K (`_GKGS`) solvers
//   LDS        two halo buffers (<= 200 slots x 128 B), a ring of three 8 KiB weight tiles, filled by
//              global_load_lds_dwordx4 with the bank-conflict XOR swizzle applied on the source side; out-of-image halo
//              slots read a 16-byte zero page, so padding costs nothing in the MFMA loop
//   sync       per tap: counted s_waitcnt vmcnt + one raw s_barrier (younger DMAs stay in flight)
//   MFMA       D^T[n, m] = W[n, :] . X[m, :] on v_mfma_f32_32x32x16_f16, waves 2 (pixels) x 2 (channels), 64 x 32 each
//   epilogue   through an fp32 LDS stage: + bias (+ residual), one fp16 rounding, 128-byte row segments
#include "dsc_common.h"
#include "dsc_hip.h"

namespace {

constexpr int BM = 128, BN = 64, BK = 64, T = 256;
constexpr int kBStage = BN * BK;                 // halves per weight tile (8 KiB)
constexpr int kAPiecesMax = 25;                  // 200 halo slots
constexpr int kAHalves = kAPiecesMax * 512;      // one halo buffer: 25.6 KiB
constexpr int kDummy = 512;                      // landing pad for the DMA pieces beyond the halo (keeps vmcnt uniform)
constexpr int lds_halves(int stages) { return 2 * kAHalves + kDummy + stages * kBStage; }   // 3 stages: 75 KiB -> 2 workgroups / CU
constexpr int kEpiStride = BN + 4;

__device__ __attribute__((aligned(16))) half_t g_zero_page[8] = {0, 0, 0, 0, 0, 0, 0, 0};

struct ConvParams {
    const half_t* x; const half_t* w; const half_t* bias; const half_t* res; half_t* out; float* ws;
    int B, H, W, Cin, Cout;
    int up;                       // 1: x is [B, H/2, W/2, Cin] and is read through a nearest-neighbour 2x upsampling
    long long ldx, ldr, ldo;      // pixel strides (elements)
    int nc, splits, cps;          // 64-channel slices, split count, slices per split
    int bpr, bpi, nblk;           // sub-blocks per image row / per image / in total
    int mt, nt;                   // tiles along pixels / output channels
    long long npix;
    long long* stamps;            // diagnostics (dsc_debug_set_conv_stamps): 8 x int64 per workgroup, NULL in normal calls
};

__device__ __forceinline__ void dma16(const half_t* src, half_t* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

// s_waitcnt vmcnt(N) only (expcnt / lgkmcnt fields at their no-wait maxima); gfx9 simm16: vmcnt[3:0] | exp[6:4] | lgkm[11:8] | vmcnt[5:4] << 14
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
    asm volatile("" ::: "memory");
}

// DMA instructions a wave issues after weight tile j+1 (the last thing step j+1-S issued) up to step j-1, for step j at
// tap t: steps j-1 .. j-(S-2), 9 at a tap-0 step, else 2.  At tap 8 the next slice's halo (issued first in this slice's
// tap-0 step, 2 + 7*2 younger instructions) must have landed as well.
template <int S>
constexpr int younger_dmas(int t) {
    int n = 0;
    for (int i = 1; i <= S - 2; ++i) n += ((t - i) % 9 + 9) % 9 == 0 ? 9 : 2;
    if (t == 8 && n > 16) n = 16;
    return n;
}
template <int S>
__device__ __forceinline__ void wait_step(int t) {
    switch (t) {                                                  // t is a compile-time constant after unrolling
        case 0: wait_vm<younger_dmas<S>(0)>(); break;
        case 1: wait_vm<younger_dmas<S>(1)>(); break;
        case 2: wait_vm<younger_dmas<S>(2)>(); break;
        case 3: wait_vm<younger_dmas<S>(3)>(); break;
        case 4: wait_vm<younger_dmas<S>(4)>(); break;
        case 5: wait_vm<younger_dmas<S>(5)>(); break;
        case 6: wait_vm<younger_dmas<S>(6)>(); break;
        case 7: wait_vm<younger_dmas<S>(7)>(); break;
        default: wait_vm<younger_dmas<S>(8)>(); break;
    }
}

// bank-conflict swizzle of a halo slot's 16-byte chunks: ds_read_b128 serves lanes in groups of 16 over 64 banks, a
// slot is 128 B (half the banks, the half = slot parity = halo column parity because the halo pitch is even), so the 16
// lanes of a group - 4..16 consecutive columns of 1..4 halo rows - must get 16 distinct (column parity, chunk) pairs
template <int TW>
__device__ __forceinline__ int halo_swz(int hy, int hx) {
    if constexpr (TW == 16) return (hx >> 1) & 7;
    else return ((hx >> 1) ^ ((hy & 1) << 2)) & 7;
}

struct Frags { h8_t w[4], x0[4], x1[4]; };

template <int TW, int S>
__global__ __launch_bounds__(T, 2) void conv3x3_kernel(ConvParams p) {
    constexpr int NSB = 16 / TW;                 // sub-blocks of 8 x TW pixels per tile
    constexpr int HWD = TW + 2;                  // halo row width (even)
    constexpr int HS = 10 * HWD;                 // halo slots per sub-block
    constexpr int NSLOT = NSB * HS;              // 180 (TW = 16) / 200 (TW = 8)
    constexpr int NPIECE = (NSLOT + 7) / 8;      // DMA pieces (8 slots x 128 B) per halo buffer
    static_assert(NPIECE <= kAPiecesMax && NPIECE <= 28, "halo does not fit");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* lds = reinterpret_cast<half_t*>(smem);
    half_t* dummy = lds + 2 * kAHalves;
    half_t* bring = dummy + kDummy;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware order: workgroup id i runs on XCD i % 8; consecutive virtual ids (same weight slab, neighbouring
    // pixel tiles) are placed on one XCD so that slab is fetched into one L2
    const int total = p.mt * p.nt * p.splits;
    const int per = gridDim.x >> 3;
    const int v = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (v >= total) return;
    long long st0 = 0, st1 = 0, st2 = 0, sc0 = 0, sc1 = 0, sc2 = 0;
    if (p.stamps) { st0 = __builtin_amdgcn_s_memrealtime(); sc0 = __builtin_amdgcn_s_memtime(); }
    const int bm = v % p.mt;
    const int rest = v / p.mt;
    const int bn = rest % p.nt, sp = rest / p.nt;
    const int n0 = bn * BN;
    const int cb = sp * p.cps, ce = min(p.nc, cb + p.cps);
    const int ns = (ce - cb) * 9;
    const long long Kw = 9ll * p.Cin;

    // ---- halo DMA sources: 7 pieces per wave; slot = piece * 8 + lane / 8, LDS chunk lane % 8 <- global chunk ^ swz(slot)
    int aoff[7];                             // element offset of the lane's 16 bytes within slice 0, -1 = zero page
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int piece = i * 4 + wave;
        const int slot = piece * 8 + (lane >> 3);
        int px = -1, sw = 0;
        if (slot < NSLOT) {
            const int sb = slot / HS, rem = slot % HS;
            const int hy = rem / HWD, hx = rem % HWD;
            sw = halo_swz<TW>(hy, hx);
            const int g = bm * NSB + sb;
            if (g < p.nblk) {
                const int b = g / p.bpi, r2 = g % p.bpi;
                const int y = (r2 / p.bpr) * 8 - 1 + hy, x = (r2 % p.bpr) * TW - 1 + hx;
                if (y >= 0 && y < p.H && x >= 0 && x < p.W)
                    px = p.up ? (b * (p.H >> 1) + (y >> 1)) * (p.W >> 1) + (x >> 1) : (b * p.H + y) * p.W + x;
            }
        }
        aoff[i] = px >= 0 ? px * (int)p.ldx + ((lane & 7) ^ sw) * 8 : -1;
    }

    auto issue_a = [&](int i, int c, int ab) {
        const int piece = i * 4 + wave;
        half_t* dst = piece < NPIECE ? lds + ab * kAHalves + piece * 512 : dummy;
        const half_t* src = aoff[i] >= 0 ? p.x + aoff[i] + c * BK : g_zero_page;
        dma16(src, dst);
    };
    // weight tile: row n of the tile at 128 B pitch, chunk ^ ((row >> 1) & 7): conflict-free for 16-lane read groups
    auto issue_b = [&](int c, int t, int stg) {
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            const int piece = pc * 4 + wave;
            const int row = piece * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            dma16(p.w + (long long)(n0 + row) * Kw + (long long)t * p.Cin + c * BK + chunk * 8,
                  bring + stg * kBStage + piece * 512);
        }
    };

    // ---- MFMA operand rows: fragment mt of this wave covers pixels wm*64 + mt*32 + r of the tile; the two fragments
    // share the halo column and the halo row parity, hence the swizzle
    int s0[2];
    int hy0, hx0;
    {
        const int m = wm * 64 + r;
        const int sb = m / (8 * TW), py = (m % (8 * TW)) / TW, pxl = m % TW;
        hy0 = py + 1; hx0 = pxl + 1;
        s0[0] = sb * HS + hy0 * HWD + hx0;
        const int m1 = m + 32;
        s0[1] = (m1 / (8 * TW)) * HS + ((m1 % (8 * TW)) / TW + 1) * HWD + hx0;
    }
    const int wrow = wn * 32 + r;
    const int wsw = (wrow >> 1) & 7;

    auto load_frags = [&](Frags& f, const half_t* a, const half_t* b, int t) {
        const int dy = t / 3 - 1, dx = t % 3 - 1;
        const int off = dy * HWD + dx;
        const int xsw = halo_swz<TW>(hy0 + dy, hx0 + dx);
        const half_t* a0 = a + (s0[0] + off) * BK;
        const half_t* a1 = a + (s0[1] + off) * BK;
        const half_t* bw = b + wrow * BK;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kc = 2 * ks + hh;
            f.w[ks] = *reinterpret_cast<const h8_t*>(bw + ((kc ^ wsw) << 3));
            f.x0[ks] = *reinterpret_cast<const h8_t*>(a0 + ((kc ^ xsw) << 3));
            f.x1[ks] = *reinterpret_cast<const h8_t*>(a1 + ((kc ^ xsw) << 3));
        }
    };

    f16x_t acc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }

    // Every step issues a number of DMA instructions per wave that depends on its tap only (tap 0: the next slice's 7
    // halo pieces + 2 weight pieces, other taps: 2 weight pieces; a 16-byte zero-page read into the landing pad where
    // there is nothing to fetch), so "tile j+1 has landed" is a compile-time vmcnt per tap (younger_dmas).
    auto dummy_dma = [&]() { dma16(g_zero_page, dummy); };

    // ---- prologue: halo of the first slice and the weight tiles of steps 0..S-1 in flight; step 0's fragments in registers
#pragma unroll
    for (int i = 0; i < 7; ++i) issue_a(i, cb, 0);
#pragma unroll
    for (int k = 0; k < S; ++k) {
        if (k < ns) issue_b(cb + k / 9, k % 9, k);
        else { dummy_dma(); dummy_dma(); }
    }
    wait_vm<2 * (S - 1)>();
    __builtin_amdgcn_s_barrier();
    Frags cur, nxt;
    load_frags(cur, lds, bring, 0);
    if (p.stamps) { st1 = __builtin_amdgcn_s_memrealtime(); sc1 = __builtin_amdgcn_s_memtime(); }

    // ---- main loop.  Step j = (slice c, tap t).  The barrier of step j publishes weight tile j+1 (and, at t = 8, the
    // next slice's halo), proves every wave holds tile j in registers (so its ring stage is refilled with tile j+S), and
    // the fragments of step j+1 are read while step j's MFMAs run.
    int j = 0, stg = 0;                      // stg = j % S = ring stage of tile j
    for (int c = cb; c < ce; ++c) {
        const int ab = (c - cb) & 1;
        const bool has_next = c + 1 < ce;
        const half_t* a = lds + ab * kAHalves;
#pragma unroll
        for (int t = 0; t < 9; ++t, ++j) {
            // tile j+1 was issued last in step j+1-S; while it still is a prologue tile (j < S-1) at least the 2(S-2-j)
            // later prologue pieces + the steps so far are younger: 2S-4 is a safe (early) bound there
            if (j >= S - 1) wait_step<S>(t);
            else wait_vm<2 * S - 4>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // step j's fragments are in registers
            __builtin_amdgcn_s_barrier();
            if (t == 0) {
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    if (has_next) issue_a(i, c + 1, ab ^ 1);
                    else dummy_dma();
                }
            }
            if (j + S < ns) issue_b(c + (t + S) / 9, (t + S) % 9, stg);
            else { dummy_dma(); dummy_dma(); }
            if (j + 1 < ns) {
                const int s1 = stg == S - 1 ? 0 : stg + 1;
                if (t < 8) load_frags(nxt, a, bring + s1 * kBStage, t + 1);
                else load_frags(nxt, lds + (ab ^ 1) * kAHalves, bring + s1 * kBStage, 0);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                acc[0] = mfma_32x32x16(cur.w[ks], cur.x0[ks], acc[0]);
                acc[1] = mfma_32x32x16(cur.w[ks], cur.x1[ks], acc[1]);
            }
            cur = nxt;
            stg = stg == S - 1 ? 0 : stg + 1;
        }
    }
    if (p.stamps) { st2 = __builtin_amdgcn_s_memrealtime(); sc2 = __builtin_amdgcn_s_memtime(); }
    __syncthreads();

    // ---- epilogue: stage[m][n] fp32
    float* stage = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f4x_t vv = {acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]};
            *reinterpret_cast<f4x_t*>(stage + (wm * 64 + mt * 32 + r) * kEpiStride + wn * 32 + 8 * g + 4 * hh) = vv;
        }
    __syncthreads();
#pragma unroll
    for (int cidx = 0; cidx < 4; ++cidx) {
        const int idx = threadIdx.x + cidx * T, m = idx >> 3, ch = idx & 7;
        const int sb = m / (8 * TW), py = (m % (8 * TW)) / TW, pxl = m % TW;
        const int g = bm * NSB + sb;
        if (g >= p.nblk) continue;
        const int b = g / p.bpi, r2 = g % p.bpi;
        const long long gp = ((long long)b * p.H + (r2 / p.bpr) * 8 + py) * p.W + (r2 % p.bpr) * TW + pxl;
        const float* sp_ = stage + m * kEpiStride + ch * 8;
        if (p.splits > 1) {
            float* dst = p.ws + ((long long)sp * p.npix + gp) * p.Cout + n0 + ch * 8;
            *reinterpret_cast<f4x_t*>(dst) = *reinterpret_cast<const f4x_t*>(sp_);
            *reinterpret_cast<f4x_t*>(dst + 4) = *reinterpret_cast<const f4x_t*>(sp_ + 4);
        } else {
            h8_t bv = {0, 0, 0, 0, 0, 0, 0, 0}, rv = {0, 0, 0, 0, 0, 0, 0, 0};
            if (p.bias) bv = *reinterpret_cast<const h8_t*>(p.bias + n0 + ch * 8);
            if (p.res) rv = *reinterpret_cast<const h8_t*>(p.res + gp * p.ldr + n0 + ch * 8);
            h8_t o;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) o[jj] = (half_t)(sp_[jj] + (float)bv[jj] + (float)rv[jj]);
            *reinterpret_cast<h8_t*>(p.out + gp * p.ldo + n0 + ch * 8) = o;
        }
    }
    if (p.stamps && threadIdx.x == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        long long* o = p.stamps + (long long)v * 8;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = __builtin_amdgcn_s_memrealtime();
        o[4] = sc1 - sc0; o[5] = sc2 - sc1; o[6] = __builtin_amdgcn_s_memtime() - sc2; o[7] = ((long long)xcc << 32) | hwid;
    }
}

// out = sum over splits (in split order) + bias + residual, one fp16 rounding
__global__ __launch_bounds__(256) void conv3x3_reduce(ConvParams p) {
    const int cv = p.Cout / 8;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.npix * cv) return;
    const long long gp = idx / cv;
    const int n = (int)(idx % cv) * 8;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < p.splits; ++k) {
        const float* src = p.ws + ((long long)k * p.npix + gp) * p.Cout + n;
        const f4x_t a = *reinterpret_cast<const f4x_t*>(src), b = *reinterpret_cast<const f4x_t*>(src + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[j] += a[j]; s[4 + j] += b[j]; }
    }
    h8_t bv = {0, 0, 0, 0, 0, 0, 0, 0}, rv = {0, 0, 0, 0, 0, 0, 0, 0};
    if (p.bias) bv = *reinterpret_cast<const h8_t*>(p.bias + n);
    if (p.res) rv = *reinterpret_cast<const h8_t*>(p.res + gp * p.ldr + n);
    h8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)(s[j] + (float)bv[j] + (float)rv[j]);
    *reinterpret_cast<h8_t*>(p.out + gp * p.ldo + n) = o;
}

long long* g_conv_stamps = nullptr;
int g_conv_ring = 0;

bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// tile width for an image, 0 = unsupported geometry
int tile_width(int H, int W) {
    if (H % 8 != 0) return 0;
    if (W % 16 == 0) return 16;
    if (W % 8 == 0) return 8;
    return 0;
}

int auto_splits(int tiles, int nc) {
    // at most 2 workgroups per CU (512 slots); prefer an even division of the slices
    int smax = 512 / tiles;
    if (smax > nc) smax = nc;
    if (smax <= 1) return 1;
    for (int s = smax; s > 1; --s)
        if (nc % s == 0) return s;
    return 1;
}

int plan(int B, int H, int W, int Cin, int Cout, int splits, ConvParams* p) {
    const int tw = tile_width(H, W);
    if (!tw || Cin % BK != 0 || Cout % BN != 0) return 0;
    p->B = B; p->H = H; p->W = W; p->Cin = Cin; p->Cout = Cout;
    p->nc = Cin / BK;
    p->bpr = W / tw; p->bpi = (H / 8) * p->bpr; p->nblk = B * p->bpi;
    const int nsb = 16 / tw;
    p->mt = (p->nblk + nsb - 1) / nsb; p->nt = Cout / BN;
    p->npix = (long long)B * H * W;
    if (splits <= 0) splits = auto_splits(p->mt * p->nt, p->nc);
    if (splits > p->nc) splits = p->nc;
    p->cps = (p->nc + splits - 1) / splits;
    p->splits = (p->nc + p->cps - 1) / p->cps;
    return tw;
}

}  // namespace

extern "C" void dsc_debug_set_conv_stamps(void* device_buffer) { g_conv_stamps = static_cast<long long*>(device_buffer); }

extern "C" void dsc_debug_set_conv_ring(int stages) { g_conv_ring = stages; }

extern "C" int dsc_conv3x3_supported(int B, int H, int W, int Cin, int Cout) {
    ConvParams p{};
    if (B <= 0 || H <= 0 || W <= 0 || (long long)B * H * W * (long long)(Cin > Cout ? Cin : Cout) >= (1ll << 31)) return 0;
    return plan(B, H, W, Cin, Cout, 0, &p) ? 1 : 0;
}

extern "C" size_t dsc_conv3x3_workspace_bytes(int B, int H, int W, int Cin, int Cout, int splits) {
    ConvParams p{};
    if (B <= 0 || H <= 0 || W <= 0 || !plan(B, H, W, Cin, Cout, splits, &p)) return 0;
    return p.splits > 1 ? (size_t)p.splits * p.npix * Cout * sizeof(float) : 0;
}

extern "C" int dsc_conv3x3_nhwc_f16(const void* x, const void* w, const void* bias, const void* residual, void* out,
                                    int B, int H, int W, int Cin, int Cout, int64_t ldx, int64_t ldr, int64_t ldo,
                                    int upsample2x, int splits, int dtype, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    if (!x || !w || !out || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return DSC_ERR_BAD_ARG;
    if (dtype != DSC_F16) return DSC_ERR_UNSUPPORTED;
    if (ldx < Cin || ldo < Cout || (residual && ldr < Cout)) return DSC_ERR_BAD_ARG;
    if (ldx % 8 != 0 || ldo % 8 != 0 || (residual && ldr % 8 != 0)) return DSC_ERR_UNSUPPORTED;
    if (!al16(x) || !al16(w) || !al16(out) || (bias && !al16(bias)) || (residual && !al16(residual))) return DSC_ERR_UNSUPPORTED;
    if ((long long)B * H * W * (ldx > ldo ? ldx : ldo) >= (1ll << 31)) return DSC_ERR_UNSUPPORTED;
    ConvParams p{};
    const int tw = plan(B, H, W, Cin, Cout, splits, &p);
    if (!tw) return DSC_ERR_UNSUPPORTED;
    p.x = static_cast<const half_t*>(x); p.w = static_cast<const half_t*>(w);
    p.bias = static_cast<const half_t*>(bias); p.res = static_cast<const half_t*>(residual);
    p.out = static_cast<half_t*>(out); p.ws = static_cast<float*>(workspace);
    p.ldx = ldx; p.ldr = ldr; p.ldo = ldo;
    p.stamps = g_conv_stamps;
    p.up = upsample2x ? 1 : 0;
    if (p.splits > 1) {
        const size_t need = (size_t)p.splits * p.npix * Cout * sizeof(float);
        if (!workspace || workspace_bytes < need || !al16(workspace)) return DSC_ERR_WORKSPACE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    static bool attr_set = false;
    if (!attr_set) {
        const void* fns[] = {reinterpret_cast<const void*>(&conv3x3_kernel<16, 3>), reinterpret_cast<const void*>(&conv3x3_kernel<8, 3>),
                             reinterpret_cast<const void*>(&conv3x3_kernel<16, 6>), reinterpret_cast<const void*>(&conv3x3_kernel<8, 6>),
                             reinterpret_cast<const void*>(&conv3x3_kernel<16, 10>), reinterpret_cast<const void*>(&conv3x3_kernel<8, 10>)};
        for (const void* f : fns) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int total = p.mt * p.nt * p.splits;
    const dim3 grid(((total + 7) / 8) * 8), block(T);
    // ring depth: 3 stages (75 KiB, two workgroups per CU).  Deeper rings (6 / 10 stages, one workgroup per CU) were
    // measured and do not shorten a step: the loop is bound by instruction issue, not by L2->LDS latency
    int ring = g_conv_ring;
    if (ring != 3 && ring != 6 && ring != 10) ring = 3;
    const size_t lds = (size_t)lds_halves(ring) * sizeof(half_t);
#define DSC_CONV_LAUNCH(TW_, S_) hipLaunchKernelGGL((conv3x3_kernel<TW_, S_>), grid, block, lds, st, p)
    if (tw == 16) { if (ring == 3) DSC_CONV_LAUNCH(16, 3); else if (ring == 6) DSC_CONV_LAUNCH(16, 6); else DSC_CONV_LAUNCH(16, 10); }
    else { if (ring == 3) DSC_CONV_LAUNCH(8, 3); else if (ring == 6) DSC_CONV_LAUNCH(8, 6); else DSC_CONV_LAUNCH(8, 10); }
#undef DSC_CONV_LAUNCH
    if (hipGetLastError() != hipSuccess) return DSC_ERR_LAUNCH;
    if (p.splits > 1) {
        const long long n = p.npix * (Cout / 8);
        hipLaunchKernelGGL(conv3x3_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
        if (hipGetLastError() != hipSuccess) return DSC_ERR_LAUNCH;
    }
    return DSC_OK;
}
