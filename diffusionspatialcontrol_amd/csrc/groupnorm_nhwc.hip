// GroupNorm (+SiLU, + fused per-(b,c) add) over channels-last fp16 activations (dsc_groupnorm_silu_nhwc).
//
// In NHWC a row (one pixel) holds all C channels contiguously, so a group is cpg = C/groups adjacent halves of
// EVERY row: the statistics are column sums.  A workgroup owns a chunk of rows of one image and ALL groups:
// thread t reads the fixed 16-byte channel vector c8 = t % (C/8) of rows slice, slice+k, ... (k = row slices per
// workgroup), so every load/store is a coalesced 16-byte access and the per-channel constants stay in registers.
//   launch 1 gn_nhwc_stats    : workgroup = (image, row chunk, channel slab of whole groups): per-thread channel sums -> LDS
//                               [slices][slab] -> per-channel -> per-group (sum, sumsq) in fp64 -> partials[b][chunk][g].
//                               At most 32 row chunks per image; the slabs supply the rest of the parallelism.
//   (gn_nhwc_finalize         : one wave per (image, group) adds the chunk partials -> (mean, rstd)[b][g] - only in the
//                               diagnostic three-launch form (mode 4) now: before the slabs a large tensor needed ~256 row
//                               chunks per image to fill the chip, too many for every apply workgroup to re-add)
//   launch 2 gn_nhwc_apply    : adds the <= 32 partial rows of its image itself (<= 16 KB of L2 reads), folds
//                               mean/rstd/gamma/beta(/add) into one scale+shift per channel, streams its rows:
//                               y = silu(x*sc + sh)
// HBM-bound: algorithmic bytes = 2 * B*hw*C*2 (read + write); the second read hits L2 / Infinity Cache.
#include "dsc_common.h"
#include "dsc_hip.h"

namespace {

constexpr int kMaxT = 512;

struct GnN {
    const half_t* x; half_t* y; const half_t* gamma; const half_t* beta; const half_t* add;
    double* partials;            // [B][nchunk][G][2]
    float* stats;                // [B][G][2] (mean, rstd), behind the partials in the workspace
    int B, HW, C, G, cpg, cv, k, nchunk, rows;       // statistics pass: nchunk row chunks of `rows` rows per image
    int anchunk, arows;                               // apply pass: its own (finer) chunking
    int inline_stats;                                 // 1: the apply workgroups reduce the chunk partials themselves (no finalize launch)
    int scv, nslab, sk;                               // statistics pass: channel slabs of scv vectors (whole groups), sk row slices per workgroup
    long long add_stride;
    float eps; int silu;
    // concatenation mode (dsc_groupnorm_silu_nhwc_cat): the input is the channel concatenation [x | x2] (C1 channels from
    // x, C - C1 from x2) that was never materialised; the pass that reads the sources also writes it to `cat`
    const half_t* x2; half_t* cat; int C1;
    // statistics from the PRODUCER of x (dsc_groupnorm_apply_nhwc; gn_partials.h): fpart[b][pt][g][which][2] fp32, fPT pixel
    // tiles per image - the apply pass is then the whole GroupNorm
    const float* fpart; int fPT;
    // index arithmetic without runtime divisions (dsc_common.h FastDiv; gn_fastdivs() below fills them before every launch)
    FastDiv fd_G, fd_anchunk, fd_cv, fd_cpg, fd_vpp, fd_nslab, fd_nchunk, fd_scv;
};

// element (b, pix, channel c .. c+7) of the input; in concatenation mode read from the source that holds c and copied to cat
__device__ __forceinline__ h8_t gn_load(const GnN& p, int b, int pix, int c) {
    const long long bp = (long long)b * p.HW + pix;
    if (!p.x2) return *reinterpret_cast<const h8_t*>(p.x + bp * p.C + c);
    const h8_t v = c < p.C1 ? *reinterpret_cast<const h8_t*>(p.x + bp * p.C1 + c)
                            : *reinterpret_cast<const h8_t*>(p.x2 + bp * (p.C - p.C1) + (c - p.C1));
    *reinterpret_cast<h8_t*>(p.cat + bp * p.C + c) = v;
    return v;
}

__global__ __launch_bounds__(kMaxT) void gn_nhwc_stats(GnN p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // workgroup = (image, row chunk, channel slab); a slab is scv 16-byte vectors = whole groups, so the workgroup owns its
    // groups' partial sums over its rows.  Slabs are the second axis of parallelism: few row chunks (<= 32 partial rows for
    // the apply pass to add up itself, no finalize launch) and still a chip-filling number of workgroups.
    const int bc = fdiv(blockIdx.x, p.fd_nslab), slab = blockIdx.x - bc * p.nslab;
    const int b = fdiv(bc, p.fd_nchunk), chunk = bc - b * p.nchunk;
    const int v0 = slab * p.scv;                              // first vector of the slab
    const int sw = min(p.scv, p.cv - v0);                     // this slab's width in vectors (the last one may be narrower)
    const int SC = p.scv * 8;                                 // LDS row pitch in channels
    float* ssum = reinterpret_cast<float*>(smem);            // [sk][SC]
    float* ssq = ssum + p.sk * SC;                           // [sk][SC]
    const int slice = fdiv(threadIdx.x, p.fd_scv), lv = threadIdx.x - slice * p.scv;
    const bool live = lv < sw;
    const int c8 = v0 + lv;
    const int r0 = chunk * p.rows, r1 = min(r0 + p.rows, p.HW);
    float s[8], q[8], ad[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; ad[j] = 0.f; }
    if (live) {
        if (p.add) {
            const h8_t a = *reinterpret_cast<const h8_t*>(p.add + (long long)b * p.add_stride + c8 * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) ad[j] = (float)a[j];
        }
        // this thread's channel vector lives in one source for all rows: (base, row stride) fixed up front
        const half_t* base = p.x + (long long)b * p.HW * p.C + c8 * 8;
        int rs = p.C;
        half_t* cdst = nullptr;
        if (p.x2) {
            const int c2 = p.C - p.C1;
            if (c8 * 8 < p.C1) { base = p.x + (long long)b * p.HW * p.C1 + c8 * 8; rs = p.C1; }
            else { base = p.x2 + (long long)b * p.HW * c2 + (c8 * 8 - p.C1); rs = c2; }
            cdst = p.cat + (long long)b * p.HW * p.C + c8 * 8;
        }
        // four rows in flight per thread (the loads of a runtime-bounded loop are otherwise issued one latency apart)
        for (int row = r0 + slice; row < r1; row += 4 * p.sk) {
            h8_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = h8_t{0, 0, 0, 0, 0, 0, 0, 0};
                if (row + u * p.sk < r1) v[u] = *reinterpret_cast<const h8_t*>(base + (long long)(row + u * p.sk) * rs);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (row + u * p.sk < r1) {
                    if (cdst) *reinterpret_cast<h8_t*>(cdst + (long long)(row + u * p.sk) * p.C) = v[u];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float f = (float)v[u][j] + ad[j]; s[j] += f; q[j] += f * f; }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { ssum[slice * SC + lv * 8 + j] = s[j]; ssq[slice * SC + lv * 8 + j] = q[j]; }
    __syncthreads();
    // per-channel totals over the sk slices (threads stride the channels, fixed order, fp64) into tot[SC][2] behind the
    // slice sums, then per-group totals over the cpg channels: two short dependent chains (sk, cpg) instead of one of
    // sk * cpg fp64 adds
    double* tot = reinterpret_cast<double*>(ssq + p.sk * SC);
    const int nch = sw * 8;                                   // channels of this slab
    if (p.sk > 1) {                                           // one slice: the group sum reads the slice values directly
        for (int c = threadIdx.x; c < nch; c += blockDim.x) {
            double c1 = 0.0, c2 = 0.0;
            for (int sl = 0; sl < p.sk; ++sl) { c1 += (double)ssum[sl * SC + c]; c2 += (double)ssq[sl * SC + c]; }
            tot[2 * c] = c1; tot[2 * c + 1] = c2;
        }
        __syncthreads();
    }
    const int ng = nch / p.cpg;                               // groups of this slab
    if (threadIdx.x < ng) {
        const int gl = threadIdx.x;
        double a1 = 0.0, a2 = 0.0;
        if (p.sk > 1) for (int c = gl * p.cpg; c < (gl + 1) * p.cpg; ++c) { a1 += tot[2 * c]; a2 += tot[2 * c + 1]; }
        else for (int c = gl * p.cpg; c < (gl + 1) * p.cpg; ++c) { a1 += (double)ssum[c]; a2 += (double)ssq[c]; }
        const int g = v0 * 8 / p.cpg + gl;
        double* dst = p.partials + (((long long)b * p.nchunk + chunk) * p.G + g) * 2;
        dst[0] = a1; dst[1] = a2;
    }
}

// one wave per (image, group): lane i adds chunk partials i, i+64, ... (fixed order), then a fixed-order butterfly
__global__ __launch_bounds__(64) void gn_nhwc_finalize(GnN p) {
    const int b = fdiv(blockIdx.x, p.fd_G), g = blockIdx.x - b * p.G;
    const double* src = p.partials + ((long long)b * p.nchunk * p.G + g) * 2;
    double a1 = 0.0, a2 = 0.0;
    for (int i = threadIdx.x; i < p.nchunk; i += 64) { a1 += src[(long long)i * p.G * 2]; a2 += src[(long long)i * p.G * 2 + 1]; }
    a1 = wave_sum_f64(a1); a2 = wave_sum_f64(a2);
    if (threadIdx.x == 0) {
        const double n = (double)p.HW * p.cpg;
        const double m = a1 / n;
        double var = a2 / n - m * m;
        var = var > 0.0 ? var : 0.0;
        p.stats[((long long)b * p.G + g) * 2] = (float)m;
        p.stats[((long long)b * p.G + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
}

__global__ __launch_bounds__(kMaxT) void gn_nhwc_apply(GnN p) {
    __shared__ float mean_s[64], rstd_s[64];
    __shared__ double red[8 * 64 * 2];
    const int b = fdiv(blockIdx.x, p.fd_anchunk), chunk = blockIdx.x - b * p.anchunk;
    // Every load this workgroup's first rows need is issued up front, in the order the values are needed (a wave's loads
    // return in order): the statistics partials, the channel constants, then the first rows - so the partial sums are being
    // added while the rows are still in flight, and nothing is fetched behind a barrier.
    constexpr int kPre = 4, kPL = 5;                          // rows / partial rows per thread in flight
    const int slice = fdiv(threadIdx.x, p.fd_cv), c8 = threadIdx.x - slice * p.cv;
    const int r0 = chunk * p.arows, r1 = min(r0 + p.arows, p.HW);
    const long long off = (long long)b * p.HW * p.C + c8 * 8;
    // few statistics chunks (<= 32 per image): every apply workgroup adds them itself, in a fixed order - 16 KB of
    // L2 reads instead of a third launch.  thread (g, part): chunks part, part + P, ...; then the P parts in order
    const int P = min(8, fdiv((int)blockDim.x, p.fd_G));
    const int part = fdiv(threadIdx.x, p.fd_G), g = threadIdx.x - part * p.G;
    const double* src = p.partials + ((long long)b * p.nchunk * p.G + g) * 2;
    double pl1[kPL], pl2[kPL];
    // producer partials: group g's sums over pixel tile i are slot (g, 0) plus, when the group straddles a 64-channel tile
    // boundary, slot (g, 1) (gn_partials.h); thread (g, part) takes tiles part, part + P, ... in order
    const bool straddles = p.fpart && (g * p.cpg) / 64 != ((g + 1) * p.cpg - 1) / 64;
    const float* fsrc = p.fpart ? p.fpart + ((long long)b * p.fPT * p.G + g) * 4 : nullptr;
    f4x_t fp[kPL];
#pragma unroll
    for (int u = 0; u < kPL; ++u) {
        const int i = part + u * P;
        pl1[u] = 0.0; pl2[u] = 0.0;
        fp[u] = f4x_t{0.f, 0.f, 0.f, 0.f};
        if (p.fpart) {
            if (part < P && i < p.fPT) fp[u] = *reinterpret_cast<const f4x_t*>(fsrc + (long long)i * p.G * 4);
        } else if (p.inline_stats && part < P && i < p.nchunk) { pl1[u] = src[(long long)i * p.G * 2]; pl2[u] = src[(long long)i * p.G * 2 + 1]; }
    }
    const h8_t ga = *reinterpret_cast<const h8_t*>(p.gamma + c8 * 8);
    const h8_t be = *reinterpret_cast<const h8_t*>(p.beta + c8 * 8);
    h8_t ad = {0, 0, 0, 0, 0, 0, 0, 0};
    if (p.add) ad = *reinterpret_cast<const h8_t*>(p.add + (long long)b * p.add_stride + c8 * 8);
    h8_t pre[kPre];
#pragma unroll
    for (int i = 0; i < kPre; ++i) {
        const int row = r0 + slice + i * p.k;
        pre[i] = h8_t{0, 0, 0, 0, 0, 0, 0, 0};
        if (row < r1) pre[i] = *reinterpret_cast<const h8_t*>(p.x + off + (long long)row * p.C);
    }
    if (p.inline_stats) {
        if (part < P) {
            double a1 = 0.0, a2 = 0.0;
            if (p.fpart) {
#pragma unroll
                for (int u = 0; u < kPL; ++u) {                                   // (missing tiles add zero)
                    a1 += (double)fp[u][0]; a2 += (double)fp[u][1];
                    if (straddles) { a1 += (double)fp[u][2]; a2 += (double)fp[u][3]; }
                }
                for (int i = part + kPL * P; i < p.fPT; i += P) {
                    const f4x_t v = *reinterpret_cast<const f4x_t*>(fsrc + (long long)i * p.G * 4);
                    a1 += (double)v[0]; a2 += (double)v[1];
                    if (straddles) { a1 += (double)v[2]; a2 += (double)v[3]; }
                }
            } else {
#pragma unroll
                for (int u = 0; u < kPL; ++u) { a1 += pl1[u]; a2 += pl2[u]; }         // (missing chunks add zero)
                for (int i = part + kPL * P; i < p.nchunk; i += P) { a1 += src[(long long)i * p.G * 2]; a2 += src[(long long)i * p.G * 2 + 1]; }
            }
            red[(part * 64 + g) * 2] = a1; red[(part * 64 + g) * 2 + 1] = a2;
        }
        __syncthreads();
        if (threadIdx.x < p.G) {
            double t1 = 0.0, t2 = 0.0;
            for (int q = 0; q < P; ++q) { t1 += red[(q * 64 + threadIdx.x) * 2]; t2 += red[(q * 64 + threadIdx.x) * 2 + 1]; }
            const double n = (double)p.HW * p.cpg;
            const double m = t1 / n;
            double var = t2 / n - m * m;
            var = var > 0.0 ? var : 0.0;
            mean_s[threadIdx.x] = (float)m;
            rstd_s[threadIdx.x] = (float)(1.0 / sqrt(var + (double)p.eps));
        }
    } else if (threadIdx.x < p.G) {
        mean_s[threadIdx.x] = p.stats[((long long)b * p.G + threadIdx.x) * 2];
        rstd_s[threadIdx.x] = p.stats[((long long)b * p.G + threadIdx.x) * 2 + 1];
    }
    __syncthreads();
    float sc[8], sh[8];
    {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int gg = fdiv(c8 * 8 + j, p.fd_cpg);
            sc[j] = (float)ga[j] * rstd_s[gg];
            sh[j] = (float)be[j] + ((float)ad[j] - mean_s[gg]) * sc[j];
        }
    }
    auto emit = [&](int row, const h8_t& v) {
        h8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = (float)v[j] * sc[j] + sh[j];
            if (p.silu) f = silu_f(f);
            o[j] = (half_t)f;
        }
        *reinterpret_cast<h8_t*>(p.y + off + (long long)row * p.C) = o;
    };
#pragma unroll
    for (int i = 0; i < kPre; ++i) {
        const int row = r0 + slice + i * p.k;
        if (row < r1) emit(row, pre[i]);
    }
    for (int row = r0 + slice + kPre * p.k; row < r1; row += p.k)
        emit(row, *reinterpret_cast<const h8_t*>(p.x + off + (long long)row * p.C));
}

// Small images (the 8x8 / 16x16 UNet levels): ONE launch, one workgroup per (image, group).  The group's slab
// (hw pixels x cpg channels, cpg % 8 == 0, at most kSmallVec 16-byte vectors per thread) is read once into registers,
// reduced (two-pass variance on the register copy), normalised and written - instead of three launches of ~3 us each.
constexpr int kSmallVec = 8;
__global__ __launch_bounds__(256) void gn_nhwc_small(GnN p) {
    __shared__ float red[8];
    const int b = fdiv(blockIdx.x, p.fd_G), g = blockIdx.x - b * p.G;
    const int vpp = p.cpg >> 3;                               // vectors per pixel of this group
    const int nvec = p.HW * vpp;
    const long long base = (long long)b * p.HW * p.C + g * p.cpg;
    float v[kSmallVec][8];
    h8_t gav[kSmallVec], bev[kSmallVec];
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < kSmallVec; ++i) {
        const int idx = threadIdx.x + 256 * i;
        if (idx < nvec) {
            const int pix = fdiv(idx, p.fd_vpp), j8 = idx - pix * vpp;
            const h8_t x = gn_load(p, b, pix, g * p.cpg + j8 * 8);
            h8_t ad = {0, 0, 0, 0, 0, 0, 0, 0};
            if (p.add) ad = *reinterpret_cast<const h8_t*>(p.add + (long long)b * p.add_stride + g * p.cpg + j8 * 8);
            // the channel constants ride along now: fetched after the second reduction they were a memory round trip of
            // their own at the end of a kernel that is nothing but a latency chain
            gav[i] = *reinterpret_cast<const h8_t*>(p.gamma + g * p.cpg + j8 * 8);
            bev[i] = *reinterpret_cast<const h8_t*>(p.beta + g * p.cpg + j8 * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[i][j] = (float)x[j] + (float)ad[j]; s1 += v[i][j]; }
        } else {
            gav[i] = h8_t{0, 0, 0, 0, 0, 0, 0, 0}; bev[i] = gav[i];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    s1 = wave_sum_f32(s1);
    if (lane == 0) red[wave] = s1;
    __syncthreads();
    const float n = (float)p.HW * p.cpg;
    const float mean = (red[0] + red[1] + red[2] + red[3]) / n;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < kSmallVec; ++i) {
        const int idx = threadIdx.x + 256 * i;
        if (idx < nvec) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float dlt = v[i][j] - mean; s2 += dlt * dlt; }
        }
    }
    s2 = wave_sum_f32(s2);
    if (lane == 0) red[4 + wave] = s2;
    __syncthreads();
    const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / n + p.eps);
#pragma unroll
    for (int i = 0; i < kSmallVec; ++i) {
        const int idx = threadIdx.x + 256 * i;
        if (idx < nvec) {
            const int pix = fdiv(idx, p.fd_vpp), j8 = idx - pix * vpp;
            const h8_t ga = gav[i], be = bev[i];
            h8_t o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f = (v[i][j] - mean) * rstd * (float)ga[j] + (float)be[j];
                if (p.silu) f = silu_f(f);
                o[j] = (half_t)f;
            }
            *reinterpret_cast<h8_t*>(p.y + base + (long long)pix * p.C + j8 * 8) = o;
        }
    }
}

// Medium images and group widths that are not multiples of 8 channels (the 32x32 / 16x16 UNet levels: 20, 30, 60
// channels per group): ONE launch, one workgroup per (image, bundle), a bundle = the smallest run of groups whose
// channels fill whole 16-byte vectors (lcm(cpg, 8) channels: 1, 2 or 4 groups).  Thread t owns vector t % nvec of the
// pixels t / nvec, t / nvec + npl, ... (coalesced rows), keeps them PACKED in registers (NV x 4 VGPRs), and the
// reductions run per channel -> fixed-order LDS tree over the pixel lanes -> per group: bit-reproducible.
// Two-pass variance on the register copy, like gn_nhwc_small.
struct GnBundle { int gb, nvec, npl; };          // groups per bundle, vectors per pixel of a bundle, pixel lanes

template <int T, int NV>
__global__ __launch_bounds__(T) void gn_nhwc_bundle(GnN p, GnBundle q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* S = reinterpret_cast<float*>(smem);                // [npl][nvec * 8] channel partials
    __shared__ float gstat[8];                                // per group of the bundle: mean, then rstd
    const int nb = p.G / q.gb;
    const int b = blockIdx.x / nb, bun = blockIdx.x % nb;
    const int cw = q.nvec * 8;                                // channels of a bundle
    const int t = threadIdx.x;
    const bool active = t < q.nvec * q.npl;
    const int v = t % q.nvec, lp = t / q.nvec;
    const long long base = (long long)b * p.HW * p.C + (long long)bun * cw + v * 8;
    h8_t x[NV];
    float ad[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) ad[j] = 0.f;
    if (active && p.add) {
        const h8_t a = *reinterpret_cast<const h8_t*>(p.add + (long long)b * p.add_stride + bun * cw + v * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) ad[j] = (float)a[j];
    }
    float cs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[j] = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int pix = lp + i * q.npl;
        x[i] = h8_t{0, 0, 0, 0, 0, 0, 0, 0};
        if (active && pix < p.HW) {
            x[i] = *reinterpret_cast<const h8_t*>(p.x + base + (long long)pix * p.C);
#pragma unroll
            for (int j = 0; j < 8; ++j) cs[j] += (float)x[i][j] + ad[j];
        }
    }
    // fixed-order tree over the pixel lanes: S[lp][v*8 + j]
    auto tree = [&](const float* mine) {
        if (active) {
#pragma unroll
            for (int j = 0; j < 8; ++j) S[t * 8 + j] = mine[j];
        }
        __syncthreads();
        int n = q.npl;
        while (n > 1) {
            const int half = (n + 1) >> 1;
            if (active && lp + half < n) {
#pragma unroll
                for (int j = 0; j < 8; ++j) S[t * 8 + j] += S[(t + half * q.nvec) * 8 + j];
            }
            __syncthreads();
            n = half;
        }
    };
    tree(cs);
    const float cnt = (float)p.HW * p.cpg;
    if (t < q.gb) {
        float a = 0.f;
        for (int c = t * p.cpg; c < (t + 1) * p.cpg; ++c) a += S[c];      // row 0 of S = per-channel totals
        gstat[t] = a / cnt;
    }
    __syncthreads();
    float mj[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) mj[j] = gstat[(v * 8 + j) / p.cpg] - ad[j];   // (x + ad) - mean = x - mj
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[j] = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int pix = lp + i * q.npl;
        if (active && pix < p.HW) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = (float)x[i][j] - mj[j]; cs[j] += d * d; }
        }
    }
    tree(cs);
    if (t < q.gb) {
        float a = 0.f;
        for (int c = t * p.cpg; c < (t + 1) * p.cpg; ++c) a += S[c];
        gstat[4 + t] = rsqrtf(a / cnt + p.eps);
    }
    __syncthreads();
    if (!active) return;
    float sc[8], sh[8];
    {
        const h8_t ga = *reinterpret_cast<const h8_t*>(p.gamma + bun * cw + v * 8);
        const h8_t be = *reinterpret_cast<const h8_t*>(p.beta + bun * cw + v * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sc[j] = gstat[4 + (v * 8 + j) / p.cpg] * (float)ga[j];
            sh[j] = (float)be[j] - mj[j] * sc[j];
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int pix = lp + i * q.npl;
        if (pix < p.HW) {
            h8_t o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f = (float)x[i][j] * sc[j] + sh[j];
                if (p.silu) f = silu_f(f);
                o[j] = (half_t)f;
            }
            *reinterpret_cast<h8_t*>(p.y + base + (long long)pix * p.C) = o;
        }
    }
}

int g_gn_slabs = 1;  // diagnostics (dsc_debug_set_gn_mode(10 / 11)): 0 = one channel slab per statistics workgroup

// bundle geometry for T threads and NV vectors per thread; false when the image does not fit
bool bundle_plan(const GnN& p, int T, int NV, GnBundle* q) {
    int gb = 1;
    while ((gb * p.cpg) % 8 != 0) ++gb;                       // 1, 2, 4 or 8 groups
    if (gb > 4 || p.G % gb != 0) return false;
    q->gb = gb;
    q->nvec = gb * p.cpg / 8;
    if (q->nvec > T) return false;
    q->npl = T / q->nvec;
    return (long long)q->npl * NV >= p.HW;
}

bool plan(GnN& p) {
    p.cv = p.C / 8;
    if (p.C % 8 != 0 || p.cv > kMaxT || p.G > 64 || p.C % p.G != 0) return false;
    p.cpg = p.C / p.G;
    p.k = 256 / p.cv;
    if (p.k < 1) p.k = 1;
    while (p.cv * p.k < p.G) ++p.k;                          // the partial re-add needs >= G threads
    int target = 512 / p.B;                                  // ~2 workgroups per CU
    if (target < 1) target = 1;
    int min_rows = 2 * p.k;                                  // at least two rows per thread
    int nchunk = (p.HW + min_rows - 1) / min_rows;
    if (nchunk > target) nchunk = target;
    if (nchunk < 1) nchunk = 1;
    p.arows = (p.HW + nchunk - 1) / nchunk;
    p.anchunk = (p.HW + p.arows - 1) / p.arows;
    // Statistics pass: at most 32 row chunks per image (every apply workgroup adds the <= 32 partial rows itself: <= 16 KB
    // of L2 reads, no finalize launch) times as many channel slabs (whole groups, whole 16-byte vectors) as fill the chip.
    // Before the slabs a large tensor needed the fine row chunking to occupy the chip, hence ~256 partial rows per image and
    // a finalize launch in between (640ch @64x64: 18.1 us with three launches against 23.9 with two and no slabs).
    {
        const int sch = p.anchunk < 32 ? p.anchunk : 32;          // (16 row chunks: the same GroupNorm time per step, 8: +5 %)
        p.rows = (p.HW + sch - 1) / sch;
        p.nchunk = (p.HW + p.rows - 1) / p.rows;
        p.inline_stats = 1;
        int unit = 1;                                        // vectors per smallest slab: lcm(cpg, 8) channels
        while ((unit * 8) % p.cpg != 0) ++unit;
        const int units = p.cv / unit;
        int want = 256 / (p.B * p.nchunk);                   // slabs that bring the grid to ~256 workgroups
        if (want < 1 || g_gn_slabs == 0) want = 1;
        if (want > units) want = units;
        const int per = (units + want - 1) / want;
        p.scv = per * unit;
        p.nslab = (p.cv + p.scv - 1) / p.scv;
        p.sk = 256 / p.scv;
        if (p.sk < 1) p.sk = 1;
        while (p.scv * p.sk < p.scv * 8 / p.cpg) ++p.sk;     // >= as many threads as groups in a slab
    }
    return true;
}

// every divisor the kernels' index arithmetic uses, as a multiply (indices stay far below 2^32 / divisor: thread ids, grid sizes)
void gn_fastdivs(GnN& p) {
    const long long big = 1ll << 20;
    p.fd_G = make_fastdiv(p.G, big); p.fd_anchunk = make_fastdiv(p.anchunk > 0 ? p.anchunk : 1, big);
    p.fd_cv = make_fastdiv(p.cv > 0 ? p.cv : 1, big); p.fd_cpg = make_fastdiv(p.cpg > 0 ? p.cpg : 1, big);
    p.fd_vpp = make_fastdiv(p.cpg >= 8 ? p.cpg >> 3 : 1, big); p.fd_nslab = make_fastdiv(p.nslab > 0 ? p.nslab : 1, big);
    p.fd_nchunk = make_fastdiv(p.nchunk > 0 ? p.nchunk : 1, big); p.fd_scv = make_fastdiv(p.scv > 0 ? p.scv : 1, big);
}

int g_gn_mode = 0;   // diagnostics (dsc_debug_set_gn_mode): 0 auto, 2 never a single-launch kernel, 3 also the 1024-thread bundle kernel, 4 = 2 + separate finalize launch

bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

__global__ __launch_bounds__(256) void add_bias_kernel(const half_t* a, const half_t* b, const half_t* bias, half_t* out,
                                                       long long n8, int cv) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const h8_t va = *reinterpret_cast<const h8_t*>(a + i * 8);
        const h8_t vb = *reinterpret_cast<const h8_t*>(b + i * 8);
        h8_t bi = {0, 0, 0, 0, 0, 0, 0, 0};
        if (bias) bi = *reinterpret_cast<const h8_t*>(bias + (i % cv) * 8);
        h8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)va[j] + (float)vb[j] + (float)bi[j]);
        *reinterpret_cast<h8_t*>(out + i * 8) = o;
    }
}

}  // namespace

extern "C" int dsc_add_bias_residual(const void* a, const void* b, const void* bias, void* out, int64_t rows, int C,
                                     int dtype, void* stream) {
    if (!a || !b || !out || rows <= 0 || C <= 0) return DSC_ERR_BAD_ARG;
    if (dtype != DSC_F16 || C % 8 != 0 || !al16(a) || !al16(b) || !al16(out) || (bias && !al16(bias))) return DSC_ERR_UNSUPPORTED;
    const long long n8 = rows * (C / 8);
    long long g = (n8 + 255) / 256;
    if (g > 2048) g = 2048;
    DSC_LAUNCH(add_bias_kernel, dim3((int)g), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const half_t*>(a), static_cast<const half_t*>(b), static_cast<const half_t*>(bias),
                       static_cast<half_t*>(out), n8, C / 8);
    return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
}

extern "C" void dsc_debug_set_gn_mode(int mode) {
    if (mode == 10 || mode == 11) g_gn_slabs = mode - 10;      // 10: one channel slab per statistics workgroup, 11: slabs (default)
    else g_gn_mode = mode;
}

extern "C" size_t dsc_groupnorm_nhwc_workspace_bytes(int B, int C, int hw, int groups) {
    GnN p{};
    p.B = B; p.C = C; p.HW = hw; p.G = groups;
    if (B <= 0 || C <= 0 || hw <= 0 || groups <= 0 || !plan(p)) return 0;
    return (size_t)B * p.anchunk * groups * 2 * sizeof(double) + (size_t)B * groups * 2 * sizeof(float);
}

namespace {
int run_groupnorm(const void* x, const void* x2, int C1, void* cat, void* y, const void* gamma, const void* beta, const void* add,
                  int64_t add_row_stride, int B, int C, int hw, int groups, float eps, int apply_silu, int dtype,
                  void* workspace, size_t workspace_bytes, void* stream);
}

extern "C" int dsc_groupnorm_silu_nhwc(const void* x, void* y, const void* gamma, const void* beta, const void* add,
                                       int64_t add_row_stride, int B, int C, int hw, int groups, float eps, int apply_silu, int dtype,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    return run_groupnorm(x, nullptr, 0, nullptr, y, gamma, beta, add, add_row_stride, B, C, hw, groups, eps, apply_silu, dtype,
                         workspace, workspace_bytes, stream);
}

extern "C" int dsc_groupnorm_silu_nhwc_cat(const void* x1, const void* x2, int C1, void* cat, void* y, const void* gamma,
                                           const void* beta, const void* add, int64_t add_row_stride, int B, int C, int hw,
                                           int groups, float eps, int apply_silu, int dtype, void* workspace,
                                           size_t workspace_bytes, void* stream) {
    if (!x2 || !cat || C1 <= 0 || C1 >= C) return DSC_ERR_BAD_ARG;
    if (C1 % 8 != 0 || !al16(x2) || !al16(cat) || cat == x1 || cat == x2 || cat == y) return DSC_ERR_UNSUPPORTED;
    return run_groupnorm(x1, x2, C1, cat, y, gamma, beta, add, add_row_stride, B, C, hw, groups, eps, apply_silu, dtype,
                         workspace, workspace_bytes, stream);
}

namespace {
int run_groupnorm(const void* x, const void* x2, int C1, void* cat, void* y, const void* gamma, const void* beta, const void* add,
                  int64_t add_row_stride, int B, int C, int hw, int groups, float eps, int apply_silu, int dtype,
                  void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !y || !gamma || !beta || B <= 0 || C <= 0 || hw <= 0 || groups <= 0) return DSC_ERR_BAD_ARG;
    if (dtype != DSC_F16) return DSC_ERR_UNSUPPORTED;
    GnN p{};
    p.B = B; p.C = C; p.HW = hw; p.G = groups;
    p.x2 = static_cast<const half_t*>(x2); p.cat = static_cast<half_t*>(cat); p.C1 = C1;
    if (!plan(p)) return DSC_ERR_UNSUPPORTED;
    if (!al16(x) || !al16(y) || !al16(gamma) || !al16(beta) || (add && (!al16(add) || add_row_stride % 8 != 0 || add_row_stride < C)))
        return DSC_ERR_UNSUPPORTED;
    const size_t need = (size_t)B * p.anchunk * groups * 2 * sizeof(double) + (size_t)B * groups * 2 * sizeof(float);
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 7)) return DSC_ERR_WORKSPACE;
    p.x = static_cast<const half_t*>(x); p.y = static_cast<half_t*>(y);
    p.gamma = static_cast<const half_t*>(gamma); p.beta = static_cast<const half_t*>(beta);
    p.add = static_cast<const half_t*>(add);
    p.add_stride = add_row_stride;
    p.partials = static_cast<double*>(workspace);
    p.stats = reinterpret_cast<float*>(p.partials + (size_t)B * p.anchunk * groups * 2);
    p.eps = eps; p.silu = apply_silu;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (p.cpg % 8 == 0 && (long long)p.HW * (p.cpg / 8) <= 256 * kSmallVec && g_gn_mode != 2) {      // small image: single launch
        gn_fastdivs(p);
        DSC_LAUNCH(gn_nhwc_small, dim3(B * groups), dim3(256), 0, st, p);
        return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
    }
    {
        // one workgroup streams a whole (image, bundle) slab: it wins only while the slab is small (measured,
        // tools/mb_gn.py: 1280 vectors 9.8 us vs 11.9 us for three launches; 3840 vectors 17.9 vs 12.6)
        GnBundle q{};
        if (g_gn_mode < 2 && !p.x2 && bundle_plan(p, 256, 16, &q) && (long long)p.HW * q.nvec <= 1536) {
            DSC_LAUNCH((gn_nhwc_bundle<256, 16>), dim3(B * (groups / q.gb)), dim3(256), (size_t)256 * 8 * sizeof(float), st, p, q);
            return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
        }
        if (g_gn_mode == 3 && !p.x2 && bundle_plan(p, 1024, 16, &q)) {       // diagnostics only: slower than three launches
            DSC_LAUNCH((gn_nhwc_bundle<1024, 16>), dim3(B * (groups / q.gb)), dim3(1024), (size_t)1024 * 8 * sizeof(float), st, p, q);
            return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
        }
    }
    const dim3 block(p.cv * p.k);
    if (g_gn_mode == 4) {                                        // diagnostics: the three-launch form (fine row chunks, no slabs)
        p.rows = p.arows; p.nchunk = p.anchunk; p.inline_stats = 0;
        p.scv = p.cv; p.nslab = 1; p.sk = p.k;
    }
    const size_t stats_lds = (size_t)2 * p.sk * p.scv * 8 * sizeof(float) + (size_t)2 * p.scv * 8 * sizeof(double);
    if (stats_lds > 64 * 1024) {
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_nhwc_stats), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
    }
    gn_fastdivs(p);
    DSC_LAUNCH(gn_nhwc_stats, dim3(B * p.nchunk * p.nslab), dim3(p.scv * p.sk), stats_lds, st, p);
    if (!p.inline_stats) DSC_LAUNCH(gn_nhwc_finalize, dim3(B * groups), dim3(64), 0, st, p);
    if (p.x2) { p.x = p.cat; p.x2 = nullptr; }                  // the statistics pass wrote the concatenation: apply streams it
    DSC_LAUNCH(gn_nhwc_apply, dim3(B * p.anchunk), block, 0, st, p);
    return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
}
}  // namespace

// The GroupNorm whose statistics came out of the producing kernel's epilogue (gn_partials.h): ONE launch
extern "C" int dsc_groupnorm_apply_nhwc(const void* x, void* y, const void* gamma, const void* beta, const float* gn_part,
                                        int part_rows, int B, int C, int hw, int groups, float eps, int apply_silu, int dtype,
                                        void* stream) {
    if (!x || !y || !gamma || !beta || !gn_part || part_rows <= 0 || B <= 0 || C <= 0 || hw <= 0 || groups <= 0) return DSC_ERR_BAD_ARG;
    if (dtype != DSC_F16) return DSC_ERR_UNSUPPORTED;
    GnN p{};
    p.B = B; p.C = C; p.HW = hw; p.G = groups;
    if (!plan(p) || p.cpg > 64 || C % 64 != 0) return DSC_ERR_UNSUPPORTED;
    if (!al16(x) || !al16(y) || !al16(gamma) || !al16(beta) || !al16(gn_part)) return DSC_ERR_UNSUPPORTED;
    p.x = static_cast<const half_t*>(x); p.y = static_cast<half_t*>(y);
    p.gamma = static_cast<const half_t*>(gamma); p.beta = static_cast<const half_t*>(beta);
    p.eps = eps; p.silu = apply_silu;
    p.fpart = gn_part; p.fPT = part_rows; p.inline_stats = 1;
    gn_fastdivs(p);
    DSC_LAUNCH(gn_nhwc_apply, dim3(B * p.anchunk), dim3(p.cv * p.k), 0, static_cast<hipStream_t>(stream), p);
    return hipGetLastError() == hipSuccess ? DSC_OK : DSC_ERR_LAUNCH;
}
