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
//   tile       128 output pixels (NSB sub-blocks of 8 x TW pixels; TW = 16, or 8 for 8-wide images) x 64 output channels;
//              image sides need not be multiples of the sub-block: the overhang reads zeros and its stores are masked
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
#include "gn_partials.h"
#include <type_traits>

extern int g_dsc_tuning_profile;     // c_api.hip

namespace {

constexpr int BM = 128, BN = 64, BK = 64, T = 256;
constexpr int kBStageBytes = BN * BK * 2;        // one weight tile: 8 KiB
// Ring depth S: 9 taps % S == 0 makes a weight tile's stage its tap % S - compile-time.  S = 3 (75 KiB, two workgroups per
// CU) is what runs; S = 9 (123 KiB, one per CU, eight tiles in flight) is kept as a diagnostic variant
// (dsc_debug_set_conv_ring): it is slower on every shape, i.e. the step is not waiting for DMA latency.
constexpr int kAPiecesMax = 25;                  // 200 halo slots
constexpr int kABytes = kAPiecesMax * 1024;      // one halo buffer: 25 KiB
// LDS map (bytes): weight ring | halo buffer 0 | halo buffer 1 | landing pad for the DMA pieces beyond the halo
constexpr int kRingOff = 0;
constexpr int a_off(int S) { return S * kBStageBytes; }
constexpr int pad_off(int S) { return a_off(S) + 2 * kABytes; }
constexpr int lds_bytes(int S) { return pad_off(S) + 1024; }   // S = 3: 76800 B
constexpr int kEpiStride = BN + 4;
constexpr unsigned kOob = 0x80000000u;           // buffer offset beyond any supported tensor: the load returns zeros

struct ConvParams {
    const half_t* x; const half_t* w; const half_t* bias; const half_t* res; half_t* out; float* ws;
    int B, H, W, Cin, Cout;
    int up;                       // 1: x is [B, H/2, W/2, Cin] and is read through a nearest-neighbour 2x upsampling
    int nchw;                     // 1: out is [B, Cout, H, W] (the UNet's conv_out: 4 channels back to the sampler's layout)
    int sub2;                     // 1 / 2: only the even / odd pixels are kept: out is [B, H/2, W/2, Cout] = the stride-2
                                  // convolution with pad 1 (UNet Downsample2D) / with pad (0,1,0,1) (VAE encoder)
    long long onpix;              // output pixels (npix, or npix / 4 with sub2)
    long long ldx, ldr, ldo;      // pixel strides (elements)
    int nc, splits, cps;          // 64-channel slices, split count, slices per split
    int order;                    // workgroup order within an XCD (see the kernel)
    int bpr, bpi, nblk;           // sub-blocks per image row / per image / in total
    int mt, nt;                   // tiles along pixels / output channels
    FastDiv fd_cv;                // conv3x3_reduce: by the 8-channel vectors per pixel
    FastDiv fd_mt, fd_nt, fd_bpi, fd_bpr;   // (plan(): the workgroup's tile, image and sub-block row without runtime divisions)
    long long npix;
    unsigned x_bytes, w_bytes;    // extents for the buffer descriptors
    long long* stamps;            // diagnostics (dsc_debug_set_conv_stamps): 8 x int64 per workgroup, NULL in normal calls
    // dsc_conv3x3_gn_nhwc_f16: a per-IMAGE bias row add[b][c] (the ResNet block's time-embedding term) and the GroupNorm
    // partial sums of the stored tensor (gn_partials.h) - 16-wide tiles, no split, Cout % 64 == 0 only
    const half_t* add; long long add_ld;
    float* gn_part; int gn_cpg, gn_G;
};

// LDS-DMA of 16 bytes per lane through a buffer descriptor: lanes whose offset lies beyond the extent deposit zeros -
// the convolution's zero padding costs no branch and no memory traffic
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_byte) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(uintptr_t)lds_byte, 16, voff, soff, 0, 0);
}

// s_waitcnt vmcnt(N) only (expcnt / lgkmcnt fields at their no-wait maxima); gfx9 simm16: vmcnt[3:0] | exp[6:4] | lgkm[11:8] | vmcnt[5:4] << 14
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
    asm volatile("" ::: "memory");
}

// DMA instructions a wave issues after weight tile j+1 (the last thing step j+1-S issued) up to step j-1, for step j at
// tap t: steps j-1 .. j-(S-2), 9 for a tap-0 step (7 halo pieces + 2 weight pieces), else 2.  The next slice's halo is
// issued first in the tap-0 step, i.e. before the tile the tap-8 step waits for (S <= 9): no extra condition.
template <int S>
constexpr int younger_dmas(int t) {
    int n = 0;
    for (int i = 1; i <= S - 2; ++i) n += ((t - i) % 9 + 9) % 9 == 0 ? 9 : 2;
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

__device__ __forceinline__ h8_t lds_read(unsigned byte_addr) {
    return *reinterpret_cast<const __attribute__((address_space(3))) h8_t*>((uintptr_t)byte_addr);
}

// NLOAD = 4: four more waves that issue every DMA of the loop and nothing else (one per SIMD beside a computing wave: 2 x 245
// registers fit the 512 of a SIMD, but only with ONE workgroup per CU - the kernel of the <= 256-workgroup grids under the
// latency profile).  A DMA instruction holds its wave's issue port for 60-200 cycles; in the plain kernel the 2.8 of them per
// step sit between the 8 MFMAs of the wave that also has to feed the matrix pipe.
template <int TW, int S, int NLOAD = 0>
__global__ __launch_bounds__(T + 64 * NLOAD, (S == 3 || NLOAD ? 2 : 1)) void conv3x3_kernel(ConvParams p) {
    static_assert(S == 3 || S == 9, "the ring depth must divide the 9 taps");
    static_assert(NLOAD == 0 || NLOAD == 4, "loader waves mirror the four computing waves' DMA shares");
    constexpr bool LOADER = NLOAD > 0;
    constexpr int kAOff = a_off(S), kPadOff = pad_off(S);
    constexpr int NSB = 16 / TW;                 // sub-blocks of 8 x TW pixels per tile
    constexpr int HWD = TW + 2;                  // halo row width (even)
    constexpr int HS = 10 * HWD;                 // halo slots per sub-block
    constexpr int NSLOT = NSB * HS;              // 180 (TW = 16) / 200 (TW = 8)
    constexpr int NPIECE = (NSLOT + 7) / 8;      // DMA pieces (8 slots x 128 B) per halo buffer
    constexpr int NPAR = TW == 16 ? 1 : 2;       // halo-row parities the swizzle distinguishes
    constexpr int MT1 = (32 / TW) * HWD * 128;   // byte distance of the wave's second 32-pixel fragment in the halo
    static_assert(NPIECE <= kAPiecesMax && NPIECE <= 28, "halo does not fit");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63, wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool is_loader = LOADER && wave_all >= 4;
    const int wave = wave_all & 3;                           // DMA share / output quadrant of this wave
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
    int bm, bn, sp;
    if (p.order) {                                           // channel blocks fastest: one XCD shares a pixel tile's halo
        const int rest = fdiv(v, p.fd_nt);
        bn = v - rest * p.nt;
        sp = fdiv(rest, p.fd_mt); bm = rest - sp * p.mt;
    } else {                                                 // pixel tiles fastest: one XCD shares a weight slab
        const int rest = fdiv(v, p.fd_mt);
        bm = v - rest * p.mt;
        sp = fdiv(rest, p.fd_nt); bn = rest - sp * p.nt;
    }
    const int n0 = bn * BN;
    const int cb = sp * p.cps, ce = min(p.nc, cb + p.cps);
    const int ns = (ce - cb) * 9;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.w), 0, p.w_bytes, 0x00020000);

    // ---- weight DMA: piece = 8 rows x 128 B, 2 pieces per wave; LDS chunk lane % 8 <- global chunk ^ ((row >> 1) & 7)
    // (a 128-byte row covers half the banks, the half being the row parity: conflict-free 16-lane read groups)
    unsigned wvoff[2];
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
        const int row = (pc * 4 + wave) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        wvoff[pc] = ((unsigned)(n0 + row) * 9u * (unsigned)p.Cin + chunk * 8) * 2u;
    }
    auto issue_b = [&](unsigned soff, int stage) {
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) dma16(wr, wvoff[pc], soff, kRingOff + stage * kBStageBytes + (pc * 4 + wave) * 1024);
    };
    auto tile_soff = [&](int c, int t) { return ((unsigned)t * (unsigned)p.Cin + (unsigned)c * BK) * 2u; };

    // the weight tiles of steps 0..S-1 go out first: their latency overlaps the halo index arithmetic below
    const bool dma_wave = !LOADER || is_loader;              // this wave issues DMA
    if (dma_wave) {
#pragma unroll
        for (int k = 0; k < S; ++k) issue_b(tile_soff(cb, k), k);      // ns >= 9 >= S
    }

    // ---- sub-block origins (wave-uniform: at most two per tile, so the runtime divisions run once, not per lane)
    int ob[NSB], oy[NSB], ox[NSB];
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) {
        const int g = bm * NSB + sb;
        const int b = fdiv(g, p.fd_bpi), r2 = g - b * p.bpi, byy = fdiv(r2, p.fd_bpr);
        ob[sb] = g < p.nblk ? b : -1;            // -1: the tile's last sub-block does not exist
        oy[sb] = byy * 8; ox[sb] = (r2 - byy * p.bpr) * TW;
    }
    // ---- halo DMA: 7 pieces per wave; slot = piece * 8 + lane / 8, LDS chunk lane % 8 <- global chunk ^ swz(slot);
    // slots outside the image (zero padding), beyond the halo or of a missing sub-block read out of bounds -> zeros
    unsigned aoff[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int slot = (i * 4 + wave) * 8 + (lane >> 3);
        unsigned off = kOob;
        if (slot < NSLOT) {
            const int sb = slot / HS, rem = slot % HS;
            const int hy = rem / HWD, hx = rem % HWD;
            const int b = NSB == 1 ? ob[0] : (sb ? ob[NSB - 1] : ob[0]);
            const int y = (NSB == 1 ? oy[0] : (sb ? oy[NSB - 1] : oy[0])) - 1 + hy;
            const int x = (NSB == 1 ? ox[0] : (sb ? ox[NSB - 1] : ox[0])) - 1 + hx;
            if (b >= 0 && y >= 0 && y < p.H && x >= 0 && x < p.W) {
                const int px = p.up ? (b * (p.H >> 1) + (y >> 1)) * (p.W >> 1) + (x >> 1) : (b * p.H + y) * p.W + x;
                off = ((unsigned)px * (unsigned)p.ldx + (((lane & 7) ^ halo_swz<TW>(hy, hx)) << 3)) * 2u;
            }
        }
        aoff[i] = off;
    }
    auto issue_a = [&](int i, int c, int ab) {
        const int piece = i * 4 + wave;
        const unsigned dst = piece < NPIECE ? kAOff + ab * kABytes + piece * 1024 : kPadOff;
        dma16(xr, aoff[i], (unsigned)c * (BK * 2), dst);
    };
    // ---- MFMA operand addresses (bytes).  Fragment 0 of this wave covers pixels wm*64 + r, fragment 1 the 32 pixels
    // after them: the same halo column, 2 (TW = 16) or 4 (TW = 8) halo rows below -> a constant byte distance.
    // xaddr[par][dx][ks]: top-left tap (dy = dx = -1) slot + dx, chunk (2 ks + hh) ^ swizzle; a tap adds dy * pitch.
    unsigned xaddr[NPAR][3][4], waddr[4], waddr_hi[4];
    {
        const int m = wm * 64 + r;
        const int sb = m / (8 * TW), py = (m % (8 * TW)) / TW, pxl = m % TW;
        const int slot00 = sb * HS + py * HWD + pxl;             // halo slot of tap (-1, -1)
#pragma unroll
        for (int par = 0; par < NPAR; ++par)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int sw = halo_swz<TW>(py + par, pxl + dx);   // par = parity offset of the tap row (dy index & 1)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) xaddr[par][dx][ks] = (unsigned)(kAOff + (slot00 + dx) * 128 + (((2 * ks + hh) ^ sw) << 4));
            }
        const int wrow = wn * 32 + r;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            waddr[ks] = (unsigned)(kRingOff + wrow * 128 + (((2 * ks + hh) ^ ((wrow >> 1) & 7)) << 4));
            waddr_hi[ks] = waddr[ks] + 4 * kBStageBytes;        // stages 4..8 of the deep ring
        }
    }
    // fragments of step (halo buffer ab, tap t, ring stage t % 3)
    auto load_frags = [&](Frags& f, int ab, int t) {
        const int dy = t / 3, dx = t % 3;
        const unsigned abase = ab * kABytes + dy * HWD * 128;          // kAOff is folded into xaddr (16-bit ds_read immediates)
        const int stg = t % S;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f.w[ks] = stg < 4 ? lds_read(waddr[ks] + stg * kBStageBytes) : lds_read(waddr_hi[ks] + (stg - 4) * kBStageBytes);
            f.x0[ks] = lds_read(xaddr[NPAR == 1 ? 0 : (dy & 1)][dx][ks] + abase);
            f.x1[ks] = lds_read(xaddr[NPAR == 1 ? 0 : (dy & 1)][dx][ks] + abase + MT1);
        }
    };

    f16x_t acc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }

    // ---- prologue: halo of the first slice (issued after the three weight tiles: everything must land); step 0's
    // fragments in registers
    if (dma_wave) {
#pragma unroll
        for (int i = 0; i < 7; ++i) issue_a(i, cb, 0);
    }
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if constexpr (LOADER) {
        if (is_loader) {
            // the loop's DMA with the plain kernel's cadence: per step a counted wait, the step's barrier, the step's issues
            for (int c = cb; c < ce; ++c) {
                const int jb = (c - cb) * 9;
                const int cn = c + 1 < ce ? c + 1 : c;
                const int P = (c - cb) & 1;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    wait_step<S>(t);
                    __builtin_amdgcn_s_barrier();
                    if (t == 0) {
#pragma unroll
                        for (int i = 0; i < 7; ++i) issue_a(i, cn, P ^ 1);
                    }
                    const bool more = jb + t + S < ns;
                    const int c3 = t + S >= 9 ? c + 1 : c, t3 = (t + S) % 9;
                    issue_b(more ? tile_soff(c3, t3) : tile_soff(c, t), t % S);
                }
            }
            __syncthreads();                                 // the epilogue's two workgroup barriers
            __syncthreads();
            if (p.gn_part) dsc_gn::gn_tile_partials_barriers();
            return;
        }
    }
    Frags f[2];
    load_frags(f[0], 0, 0);
    if (p.stamps) { st1 = __builtin_amdgcn_s_memrealtime(); sc1 = __builtin_amdgcn_s_memtime(); }

    // ---- main loop.  Step j = (slice c, tap t).  The barrier of step j publishes weight tile j+1 (and, at t = 8, the
    // next slice's halo), proves every wave holds tile j in registers (so its ring stage is refilled with tile j+S), and
    // the fragments of step j+1 are read while step j's MFMAs run.  A step has no branch: past the last tile / slice the
    // DMAs re-fetch a valid tile into a stage nobody reads again, and the last fragment read is discarded.
    // Every step issues 2 weight pieces (+ the next slice's 7 halo pieces at tap 0) per wave: compile-time vmcnt.
    // The slice body is instantiated for both parities of the slice index: 9 steps flip which register set is "current".
    auto slice = [&](auto parity, int c) {
        constexpr int P = decltype(parity)::value;               // halo buffer of this slice; f[P] holds tap 0's fragments
        const int jb = (c - cb) * 9;
        const int cn = c + 1 < ce ? c + 1 : c;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            Frags& cur = f[(P + t) & 1];
            Frags& nxt = f[(P + t + 1) & 1];
            if (!LOADER) wait_step<S>(t);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // step j's fragments are in registers
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);                       // keep step j+1's MFMAs out of step j (they would wait on their reads)
            if (!LOADER) {
                if (t == 0) {
#pragma unroll
                    for (int i = 0; i < 7; ++i) issue_a(i, cn, P ^ 1);
                }
                const bool more = jb + t + S < ns;
                const int c3 = t + S >= 9 ? c + 1 : c, t3 = (t + S) % 9;
                issue_b(more ? tile_soff(c3, t3) : tile_soff(c, t), t % S);
            }
            if (t < 8) load_frags(nxt, P, t + 1);
            else load_frags(nxt, P ^ 1, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                acc[0] = mfma_32x32x16(cur.w[ks], cur.x0[ks], acc[0]);
                acc[1] = mfma_32x32x16(cur.w[ks], cur.x1[ks], acc[1]);
            }
            // issue order within the step (the loop is instruction-issue bound: one wave per SIMD, and a 32-cycle MFMA
            // hides ~24 cycles of other issue): the next fragments' 12 LDS reads ride two per gap in the first six MFMA
            // gaps, the last two MFMAs cover their latency (a ds_read_b128 costs ~16 issue cycles: three per gap in four
            // gaps - round 2 - made those gaps 56 cycles; in the step <16, 3, 0> 35.4 -> 34.1 us, <16, 9, 4> 30.1 -> 29.8;
            // 2,2,2,2,1,1,1,1 and 2,1,2,1,... are no better: 34.6-34.7); the DMAs (1 KiB each, ~60 cycles of issue) one per gap
            if (!LOADER) __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);    // VMEM read (first DMA piece)
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (g < 6) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // DS read
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);               // MFMA
                if (!LOADER && g < (t == 0 ? 8 : 1)) __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int c = cb; c < ce; ++c) {
        if ((c - cb) & 1) slice(std::integral_constant<int, 1>{}, c);
        else slice(std::integral_constant<int, 0>{}, c);
    }
    if (p.stamps) { st2 = __builtin_amdgcn_s_memrealtime(); sc2 = __builtin_amdgcn_s_memtime(); }
    // residual pixels of this thread's four output chunks: issued before the staging pass so that their latency hides
    // under it (the skip tensor was written by an earlier kernel: Infinity Cache / HBM)
    h8_t rpre[4];
    h8_t bpre = {0, 0, 0, 0, 0, 0, 0, 0};                           // bias of this thread's chunk column (the same in all four passes)
    if (p.bias && p.splits == 1 && !p.nchw && n0 + (int)(threadIdx.x & 7) * 8 + 8 <= p.Cout)
        bpre = *reinterpret_cast<const h8_t*>(p.bias + n0 + (threadIdx.x & 7) * 8);
    h8_t apre = {0, 0, 0, 0, 0, 0, 0, 0};                          // per-image bias row (gn entry: TW = 16, one image per tile)
    if (p.add && ob[0] >= 0) apre = *reinterpret_cast<const h8_t*>(p.add + (long long)ob[0] * p.add_ld + n0 + (threadIdx.x & 7) * 8);
    float gs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cidx = 0; cidx < 4; ++cidx) {
        rpre[cidx] = h8_t{0, 0, 0, 0, 0, 0, 0, 0};
        if (p.res && p.splits == 1 && !p.nchw) {
            const int idx = threadIdx.x + cidx * T, m = idx >> 3, ch = idx & 7;
            const int sb = m / (8 * TW), py = (m % (8 * TW)) / TW, pxl = m % TW;
            const int b = NSB == 1 ? ob[0] : (sb ? ob[NSB - 1] : ob[0]);
            const int yy = (NSB == 1 ? oy[0] : (sb ? oy[NSB - 1] : oy[0])) + py, xx = (NSB == 1 ? ox[0] : (sb ? ox[NSB - 1] : ox[0])) + pxl;
            long long gp = ((long long)b * p.H + yy) * p.W + xx;
            bool live = b >= 0 && yy < p.H && xx < p.W && n0 + ch * 8 + 8 <= p.Cout;
            if (p.sub2) { live = live && (p.sub2 == 1 ? !((yy | xx) & 1) : (yy & xx & 1) != 0); gp = ((long long)b * (p.H >> 1) + (yy >> 1)) * (p.W >> 1) + (xx >> 1); }
            if (live) rpre[cidx] = *reinterpret_cast<const h8_t*>(p.res + gp * p.ldr + n0 + ch * 8);
        }
    }
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
        const int sb = m / (8 * TW), py = (m % (8 * TW)) / TW, pxl = m % TW;      // sb = cidx / 2 for TW = 8: uniform
        const int b = NSB == 1 ? ob[0] : (sb ? ob[NSB - 1] : ob[0]);
        if (b < 0) continue;
        const int yy = (NSB == 1 ? oy[0] : (sb ? oy[NSB - 1] : oy[0])) + py, xx = (NSB == 1 ? ox[0] : (sb ? ox[NSB - 1] : ox[0])) + pxl;
        if (yy >= p.H || xx >= p.W) continue;                        // overhang of a ragged image side
        long long gp = ((long long)b * p.H + yy) * p.W + xx;
        if (p.sub2) {                                                // stride 2 = the even (odd) pixels of the stride-1 result
            if (p.sub2 == 1 ? ((yy | xx) & 1) != 0 : (yy & xx & 1) == 0) continue;
            gp = ((long long)b * (p.H >> 1) + (yy >> 1)) * (p.W >> 1) + (xx >> 1);
        }
        const float* sp_ = stage + m * kEpiStride + ch * 8;
        if (p.splits > 1) {
            float* dst = p.ws + ((long long)sp * p.onpix + gp) * p.Cout + n0 + ch * 8;
            *reinterpret_cast<f4x_t*>(dst) = *reinterpret_cast<const f4x_t*>(sp_);
            *reinterpret_cast<f4x_t*>(dst + 4) = *reinterpret_cast<const f4x_t*>(sp_ + 4);
        } else {
            const int c0 = n0 + ch * 8;
            if (c0 >= p.Cout) continue;                              // channel padding of a ragged last tile
            if (c0 + 8 <= p.Cout && !p.nchw) {
                const h8_t bv = bpre, rv = rpre[cidx];
                h8_t o;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    o[jj] = (half_t)(sp_[jj] + (float)bv[jj] + (float)apre[jj] + (float)rv[jj]);
                    const float f = (float)o[jj];                    // statistics of the fp16 tensor the GroupNorm will read
                    gs[jj] += f; gq[jj] += f * f;
                }
                *reinterpret_cast<h8_t*>(p.out + gp * p.ldo + c0) = o;
            } else {
                // few output channels (conv_out) and / or channel-major output: element-wise
                const long long hw = p.onpix / p.B;
                const long long pin = gp - (long long)b * hw;        // pixel index inside the image
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int c = c0 + jj;
                    if (c < p.Cout) {
                        float v = sp_[jj] + (p.bias ? (float)p.bias[c] : 0.f);
                        if (p.res) v += (float)p.res[gp * p.ldr + c];
                        if (p.nchw) p.out[((long long)b * p.Cout + c) * hw + pin] = (half_t)v;
                        else p.out[gp * p.ldo + c] = (half_t)v;
                    }
                }
            }
        }
    }
    if (p.gn_part) {
        // this tile's (group, part) sums of what it stored: tile bm = sub-block bm (TW = 16) = pixel tile bm % bpi of image ob[0]
        float* dst = ob[0] >= 0 ? p.gn_part + ((long long)ob[0] * p.bpi + (bm - ob[0] * p.bpi)) * p.gn_G * 4 : nullptr;
        dsc_gn::gn_tile_partials(gs, gq, reinterpret_cast<float*>(smem) + BM * kEpiStride, n0, p.gn_cpg, p.gn_G, dst);
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
    // (32-bit indices - the entry point bounds pixels x channels below 2^30 - and the division by the vectors per pixel as a
    // multiply: as a 64-bit `idx / cv` it was a ~150-instruction software division at the head of a 5 us kernel)
    const int cv = p.Cout >> 3;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int)p.onpix * cv) return;
    const int gpi = fdiv(idx, p.fd_cv);
    const long long gp = gpi;
    const int n = (idx - gpi * cv) * 8;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < p.splits; ++k) {
        const float* src = p.ws + ((long long)k * p.onpix + gp) * p.Cout + n;
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
int g_conv_ring = 0;          // diagnostics (dsc_debug_set_conv_ring): 0 = from the grid size, 3 or 9 = forced

bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// tile width for an image, 0 = unsupported geometry
// Image sides need not be multiples of the tile (12 x 12: the lowest level of a 768 x 768 generation): sub-blocks hang over the
// bottom / right edge, their halo reads beyond the image deposit zeros like the padding does, their stores are masked.
int tile_width(int H, int W) {
    if (H < 1 || W < 1) return 0;
    if (W % 16 == 0) return 16;
    if (W % 8 == 0) return 8;
    return ((W + 15) / 16 * 16 - W <= (W + 7) / 8 * 8 - W) ? 16 : 8;      // the less overhang
}

// Split count from a cost model fitted to tools/mb_conv3.py on MI355X (us):
//   loop      9 * nc / S steps x 0.26 us (x1.15 once two workgroups share a CU (> 256), x wgs/512 beyond 512); g_conv_small_step
//             for grids of <= 256 workgroups
//   fixed     5 (launch, prologue, epilogue)
//   split     4 (reduce launch) + the fp32 partials written and read back: S * M * N * 8 bytes at ~3 TB/s
// e.g. 640->640 @32x32: S=1 26.5 (S=2 30.0); 1280->1280 @16x16: S=5 27 (S=1 46); 1280->1280 @8x8: S=10 17 (S=20 22)
int g_conv_loaders = 1;            // dsc_debug_set_conv_ring(400 / 401 / 402): nine-stage kernels without loader waves / by rule / always
int g_conv_order = -1;             // dsc_debug_set_conv_ring(300 / 301): pixel tiles / channel blocks fastest within an XCD (-1: by shape)
// us per step the model charges a grid of <= 256 workgroups (dsc_debug_set_conv_ring(200 + hundredths)).  Chosen on images/s,
// not on the kernels' own times (tools/ab_bench.sh, one box): 0.26 / 0.22 / 0.18 / 0.14 / 0.10 -> 11.32 / 11.40 / 11.59 / 11.67 /
// 11.66 images/s with two generations in flight, 8.11-8.19 one at a time whatever the value: few splits mean fewer workgroups and
// fewer partial sums through the L2 - work, which counts when two streams share the chip - and with the nine-stage ring and the
// loader waves the one-workgroup-per-CU launches are no slower for the stream that owns them
double g_conv_small_step = 0.14;
int auto_splits(int tiles, int nc, long long npix, int cout) {
    int best = 1;
    double best_t = 1e30;
    for (int s = 1; s <= nc; ++s) {
        if (nc % s != 0) continue;
        const long long wgs = (long long)tiles * s;
        if (s > 1 && wgs > 1024) break;
        const double load = wgs <= 256 ? g_conv_small_step / 0.26 : (wgs <= 512 ? 1.15 : 1.15 * (double)wgs / 512.0);
        double t = 9.0 * nc / s * 0.26 * load + 5.0;
        if (s > 1) t += 4.0 + (double)s * (double)npix * cout * 8.0 / 3.0e6;
        if (t < best_t) { best_t = t; best = s; }
    }
    return best;
}

int plan(int B, int H, int W, int Cin, int Cout, int splits, ConvParams* p) {
    const int tw = tile_width(H, W);
    if (!tw || Cin % BK != 0 || Cout <= 0) return 0;
    p->B = B; p->H = H; p->W = W; p->Cin = Cin; p->Cout = Cout;
    p->nc = Cin / BK;
    p->bpr = (W + tw - 1) / tw; p->bpi = ((H + 7) / 8) * p->bpr; p->nblk = B * p->bpi;
    const int nsb = 16 / tw;
    p->mt = (p->nblk + nsb - 1) / nsb; p->nt = (Cout + BN - 1) / BN;    // a ragged last tile reads zero weight rows (buffer bounds)
    p->npix = (long long)B * H * W;
    if (Cout % BN != 0) splits = 1;                                      // the partial-sum layout assumes whole tiles
    if (splits <= 0) splits = auto_splits(p->mt * p->nt, p->nc, p->npix, Cout);
    if (splits > p->nc) splits = p->nc;
    p->cps = (p->nc + splits - 1) / splits;
    p->splits = (p->nc + p->cps - 1) / p->cps;
    const long long total = (long long)p->mt * p->nt * p->splits;
    p->fd_mt = make_fastdiv(p->mt, total); p->fd_nt = make_fastdiv(p->nt, total);
    p->fd_bpi = make_fastdiv(p->bpi, (long long)p->mt * nsb + nsb); p->fd_bpr = make_fastdiv(p->bpr, p->bpi);
    return tw;
}

}  // namespace

extern "C" void dsc_debug_set_conv_stamps(void* device_buffer) { g_conv_stamps = static_cast<long long*>(device_buffer); }

extern "C" void dsc_debug_set_conv_ring(int stages) {
    if (stages >= 400) g_conv_loaders = stages - 400;
    else if (stages >= 300) g_conv_order = stages - 300;
    else if (stages >= 200) g_conv_small_step = (stages - 200) / 100.0;
    else g_conv_ring = stages;
}

extern "C" int dsc_conv3x3_supported(int B, int H, int W, int Cin, int Cout) {
    ConvParams p{};
    if (B <= 0 || H <= 0 || W <= 0 || (long long)B * H * W * (long long)(Cin > Cout ? Cin : Cout) >= (1ll << 30) ||
        9ll * Cin * Cout >= (1ll << 30)) return 0;
    return plan(B, H, W, Cin, Cout, 0, &p) ? 1 : 0;
}

extern "C" size_t dsc_conv3x3_workspace_bytes(int B, int H, int W, int Cin, int Cout, int splits) {
    ConvParams p{};
    if (B <= 0 || H <= 0 || W <= 0 || !plan(B, H, W, Cin, Cout, splits, &p)) return 0;
    return p.splits > 1 ? (size_t)p.splits * p.npix * Cout * sizeof(float) : 0;
}

namespace {
struct GnArgs { const void* add; int64_t add_ld; float* part; int groups; };
int conv_impl(const void* x, const void* w, const void* bias, const void* residual, void* out,
              int B, int H, int W, int Cin, int Cout, int64_t ldx, int64_t ldr, int64_t ldo,
              int resample, int out_nchw, int splits, int dtype, void* workspace, size_t workspace_bytes, void* stream,
              const GnArgs* gn);
}

extern "C" int dsc_conv3x3_nhwc_f16(const void* x, const void* w, const void* bias, const void* residual, void* out,
                                    int B, int H, int W, int Cin, int Cout, int64_t ldx, int64_t ldr, int64_t ldo,
                                    int resample, int out_nchw, int splits, int dtype, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    return conv_impl(x, w, bias, residual, out, B, H, W, Cin, Cout, ldx, ldr, ldo, resample, out_nchw, splits, dtype, workspace,
                     workspace_bytes, stream, nullptr);
}

// partial rows per image (pixel tiles of 8 x 16) when the GroupNorm-statistics form covers the shape, else 0: 16-wide tiles, no
// split chosen by the cost model (a split convolution's output is rounded by the reduce launch), whole 64-channel tiles,
// groups of at most 64 channels, at most 128 partial rows per image
extern "C" int dsc_conv3x3_gn_rows(int B, int H, int W, int Cin, int Cout, int groups, int resample) {
    ConvParams p{};
    const int Hc = H, Wc = W;
    // 2 <= channels per group <= 64: gn_tile_partials writes at most 32 group slots per 64-channel tile (gn_partials.h)
    if (B <= 0 || H <= 0 || W <= 0 || groups <= 0 || Cout % BN != 0 || Cout % groups != 0 || Cout / groups > 64 || Cout / groups < 2) return 0;
    if (resample == DSC_CONV_STRIDE2 || resample == DSC_CONV_STRIDE2_PAD_BR) return 0;   // (the kept pixels are a quarter of a tile's)
    if (plan(B, Hc, Wc, Cin, Cout, 0, &p) != 16 || p.splits != 1 || p.bpi > 128) return 0;
    return p.bpi;
}

extern "C" int dsc_conv3x3_gn_nhwc_f16(const void* x, const void* w, const void* bias, const void* add, int64_t add_ld,
                                       const void* residual, void* out, int B, int H, int W, int Cin, int Cout, int64_t ldx,
                                       int64_t ldr, int64_t ldo, int resample, float* gn_part, int groups, int dtype, void* stream) {
    if (!gn_part || groups <= 0) return DSC_ERR_BAD_ARG;
    if (!dsc_conv3x3_gn_rows(B, H, W, Cin, Cout, groups, resample)) return DSC_ERR_UNSUPPORTED;
    if (add && (add_ld < Cout || add_ld % 8 != 0 || (reinterpret_cast<uintptr_t>(add) & 15))) return DSC_ERR_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(gn_part) & 7) return DSC_ERR_UNSUPPORTED;
    const GnArgs gn{add, add_ld, gn_part, groups};
    return conv_impl(x, w, bias, residual, out, B, H, W, Cin, Cout, ldx, ldr, ldo, resample, 0, 1, dtype, nullptr, 0, stream, &gn);
}

namespace {
int conv_impl(const void* x, const void* w, const void* bias, const void* residual, void* out,
              int B, int H, int W, int Cin, int Cout, int64_t ldx, int64_t ldr, int64_t ldo,
              int resample, int out_nchw, int splits, int dtype, void* workspace, size_t workspace_bytes, void* stream,
              const GnArgs* gn) {
    if (!x || !w || !out || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return DSC_ERR_BAD_ARG;
    if (dtype != DSC_F16) return DSC_ERR_UNSUPPORTED;
    if (ldx < Cin || (!out_nchw && ldo < Cout) || (residual && ldr < Cout)) return DSC_ERR_BAD_ARG;
    if (ldx % 8 != 0 || (Cout % 8 == 0 && !out_nchw && ldo % 8 != 0) || (residual && Cout % 8 == 0 && ldr % 8 != 0)) return DSC_ERR_UNSUPPORTED;
    if (!al16(x) || !al16(w) || !al16(out) || (bias && !al16(bias)) || (residual && !al16(residual))) return DSC_ERR_UNSUPPORTED;
    // 32-bit byte offsets in the buffer-addressed DMAs (and kOob must lie beyond every extent)
    if ((long long)B * H * W * (ldx > ldo ? ldx : ldo) >= (1ll << 30) || 9ll * Cin * Cout >= (1ll << 30)) return DSC_ERR_UNSUPPORTED;
    ConvParams p{};
    const int tw = plan(B, H, W, Cin, Cout, splits, &p);
    if (!tw) return DSC_ERR_UNSUPPORTED;
    p.x = static_cast<const half_t*>(x); p.w = static_cast<const half_t*>(w);
    p.bias = static_cast<const half_t*>(bias); p.res = static_cast<const half_t*>(residual);
    p.out = static_cast<half_t*>(out); p.ws = static_cast<float*>(workspace);
    p.ldx = ldx; p.ldr = ldr; p.ldo = ldo;
    p.stamps = g_conv_stamps;
    if (gn) {
        p.add = static_cast<const half_t*>(gn->add); p.add_ld = gn->add_ld;
        p.gn_part = gn->part; p.gn_G = gn->groups; p.gn_cpg = Cout / gn->groups;
    }
    // activation-heavy shapes (the 64x64 level): a pixel tile's halo is fetched into one L2 for all of its channel blocks
    // (640->320 @64x64 66.8 -> 61.5 us in the step); weight-heavy ones keep sharing the weight slab
    p.order = g_conv_order >= 0 ? g_conv_order : (p.npix >= 2ll * Cout ? 1 : 0);
    if (resample < 0 || resample > 3 || ((resample == DSC_CONV_STRIDE2 || resample == DSC_CONV_STRIDE2_PAD_BR) && out_nchw)) return DSC_ERR_UNSUPPORTED;
    if ((resample == DSC_CONV_STRIDE2 || resample == DSC_CONV_STRIDE2_PAD_BR) && ((H | W) & 1)) return DSC_ERR_UNSUPPORTED;
    p.up = resample == DSC_CONV_UPSAMPLE2X ? 1 : 0;
    p.sub2 = resample == DSC_CONV_STRIDE2 ? 1 : (resample == DSC_CONV_STRIDE2_PAD_BR ? 2 : 0);
    p.onpix = p.sub2 ? p.npix / 4 : p.npix;
    p.nchw = out_nchw ? 1 : 0;
    {
        const long long in_pix = p.up ? (long long)B * (H / 2) * (W / 2) : (long long)B * H * W;
        p.x_bytes = (unsigned)(((in_pix - 1) * ldx + Cin) * 2);
        p.w_bytes = (unsigned)(9ll * Cin * Cout * 2);
    }
    if (p.splits > 1) {
        const size_t need = (size_t)p.splits * p.onpix * Cout * sizeof(float);
        if (!workspace || workspace_bytes < need || !al16(workspace)) return DSC_ERR_WORKSPACE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    static bool attr_set = false;
    if (!attr_set) {
        const void* fns[] = {reinterpret_cast<const void*>(&conv3x3_kernel<16, 3>), reinterpret_cast<const void*>(&conv3x3_kernel<8, 3>),
                             reinterpret_cast<const void*>(&conv3x3_kernel<16, 9>), reinterpret_cast<const void*>(&conv3x3_kernel<8, 9>),
                             reinterpret_cast<const void*>(&conv3x3_kernel<16, 9, 4>), reinterpret_cast<const void*>(&conv3x3_kernel<8, 9, 4>)};
        for (const void* f : fns) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int total = p.mt * p.nt * p.splits;
    const dim3 grid(((total + 7) / 8) * 8), block(T);
    // Ring depth.  The 9-stage ring needs the whole LDS (one workgroup per CU), so every grid of more than 256 workgroups loses
    // its second co-resident workgroup to it (320->320 @64x64 23.6 vs 31.4 us back to back, 29.7 vs 38.9 in the step).  A grid
    // that has at most one workgroup per CU anyway gains: in the step its weight tiles come from HBM, and two tiles in flight
    // (three stages) at ~550 cycles per step are less than that latency - 160 workgroups 32.9 -> 24.3 us, 200 (8-wide tiles)
    // 15.4 -> 14.2, 64 22.6 -> 20.2 (tools/ab_step.sh, DSC_CONV_RING=3 / 9; the warm micro-benchmark shows no difference).
    int ring = g_conv_ring;
    // ... and only while this stream owns the chip: with a second generation in flight the whole-LDS workgroups keep the other
    // stream's kernels off their CUs (dsc_set_tuning_profile)
    if (ring != 3 && ring != 9) ring = (total <= 256 && g_dsc_tuning_profile == DSC_TUNE_LATENCY) ? 9 : 3;
    // the nine-stage ring has the CU to itself anyway.  In the step: 16-wide tiles 29.2 -> 25.0 us (160 workgroups), 21.2 ->
    // 19.5 (64); the 8-wide kernel (8x8 level, two halo parities, 12 spilled registers at the 256 cap) 15.0 -> 16.0: not used
    const bool loaders = ring == 9 && (g_conv_loaders == 2 || (g_conv_loaders == 1 && tw == 16));
    const dim3 block8(T + 256);
    if (tw == 16) {
        if (loaders) DSC_LAUNCH((conv3x3_kernel<16, 9, 4>), grid, block8, (size_t)lds_bytes(9), st, p);
        else if (ring == 9) DSC_LAUNCH((conv3x3_kernel<16, 9>), grid, block, (size_t)lds_bytes(9), st, p);
        else DSC_LAUNCH((conv3x3_kernel<16, 3>), grid, block, (size_t)lds_bytes(3), st, p);
    } else {
        if (loaders) DSC_LAUNCH((conv3x3_kernel<8, 9, 4>), grid, block8, (size_t)lds_bytes(9), st, p);
        else if (ring == 9) DSC_LAUNCH((conv3x3_kernel<8, 9>), grid, block, (size_t)lds_bytes(9), st, p);
        else DSC_LAUNCH((conv3x3_kernel<8, 3>), grid, block, (size_t)lds_bytes(3), st, p);
    }
    if (hipGetLastError() != hipSuccess) return DSC_ERR_LAUNCH;
    if (p.splits > 1) {
        const long long n = p.onpix * (Cout / 8);
        p.fd_cv = make_fastdiv(Cout / 8, n + 256);
        DSC_LAUNCH(conv3x3_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
        if (hipGetLastError() != hipSuccess) return DSC_ERR_LAUNCH;
    }
    return DSC_OK;
}
}  // namespace
