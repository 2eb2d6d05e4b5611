// probe: what a PERSISTENT many-row GEMM gets out of a CU (throughput tier: 8 images per generation, M = 4096 ... 65536).
// out[m][n] = sum_k x[m][k] w[n][k] (+ bias), fp16 in / fp32 accumulate / fp16 out, or the GEGLU form hid * gelu(gate).
// Workgroups stay resident and walk their tiles in an XCD-aware order; DMA-only loader waves keep one ring of K tiles full ACROSS
// output tiles (no pipeline drain between them); the epilogue runs from registers (v_permlane32_swap -> 16-byte stores), so the
// ring is never reused as a stage.  Template knobs: WM (wave rows: tile = 64 WM rows x 128 weight rows), STAGES, NLOAD.
//
//     hipcc -O3 --offload-arch=gfx950 -o gemm_big gemm_big.hip && ./gemm_big
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
typedef _Float16 half_t;
typedef half_t h8_t __attribute__((ext_vector_type(8)));
typedef half_t h4_t __attribute__((ext_vector_type(4)));
typedef float f16x_t __attribute__((ext_vector_type(16)));
typedef unsigned u2_t __attribute__((ext_vector_type(2)));
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
constexpr int BK = 64;

struct P {
    const half_t* x; const half_t* w; const half_t* bias; half_t* out;
    int M, N, K;                // GEGLU: N = 2 x output columns
    int rm, cn;                 // row panels, column blocks
    int tiles;
    long long* prof;            // [workgroup][wave][4] cycle sums (NULL: off)
};

__device__ __forceinline__ h8_t lds_frag(const half_t* tile, int row, int kchunk) {
    return *reinterpret_cast<const h8_t*>(tile + row * BK + ((kchunk ^ ((row >> 1) & 7)) << 3));
}
__device__ __forceinline__ f16x_t mfma(h8_t a, h8_t b, f16x_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = poly * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);   // erfc(|z|)
    return x > 0.f ? x * (1.f - 0.5f * e) : x * (0.5f * e);
}

// tile index of workgroup `wg` (runs on XCD wg % 8) at its iteration j: row panel rp -> XCD rp % 8; an XCD walks its panels
// column block by column block, its workgroups side by side on consecutive entries
__device__ __forceinline__ bool tile_of(const P& p, int wg, int j, int nwg, int& rp, int& cb) {
    const int xcd = wg & 7, slot = wg >> 3, per = nwg >> 3;
    const int idx = j * per + slot;
    const int rpl = idx / p.cn;
    cb = idx - rpl * p.cn;
    rp = rpl * 8 + xcd;
    return rp < p.rm;
}

template <int WM, int STAGES, int NLOAD, bool GEGLU>
__global__ __launch_bounds__(128 * WM + 64 * NLOAD, 1) void gemm_big(P p) {
    constexpr int BM = 64 * WM, NC = 2 * WM;                 // tile rows, computing waves
    constexpr int kA = BM * BK, kStage = (BM + 128) * BK;    // halves
    constexpr int kPieces = (BM + 128) / 8 / NLOAD;
    static_assert((BM + 128) / 8 % NLOAD == 0, "pieces per loader wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* lds = reinterpret_cast<half_t*>(smem);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int nk = p.K / BK;
    const int nwg = gridDim.x, wg = blockIdx.x;
    // this workgroup's tile count
    int nt_mine = 0;
    { int rp, cb; while (tile_of(p, wg, nt_mine, nwg, rp, cb)) ++nt_mine; }
    const int Q = nt_mine * nk;
    const int Nh = p.N / 2;

    if (wave >= NC) {                                        // ---- loader waves
        const int iw = wave - NC;
        int jq = 0, ktq = 0, rp = 0, cb = 0;                 // tile / K tile of the next issue
        tile_of(p, wg, 0, nwg, rp, cb);
        auto issue = [&](int buf) {
            half_t* a = lds + buf * kStage;
            const int m0 = rp * BM;
#pragma unroll
            for (int pc = 0; pc < kPieces; ++pc) {
                const int piece = pc * NLOAD + iw;
                const int row = piece * 8 + (lane >> 3);
                const int chunk = (lane & 7) ^ ((row >> 1) & 7);
                const half_t* src;
                if (row < BM) {
                    src = p.x + (long long)min(m0 + row, p.M - 1) * p.K + ktq * BK + chunk * 8;
                } else {
                    const int j = row - BM;                  // weight row of the tile: wave column j / 64, fragment (j % 64) / 32
                    int grow;
                    if (GEGLU) grow = ((j & 63) < 32 ? 0 : Nh) + cb * 64 + (j >> 6) * 32 + (j & 31);
                    else grow = cb * 128 + j;
                    src = p.w + (long long)grow * p.K + ktq * BK + chunk * 8;
                }
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(a + piece * 512), 16, 0, 0);
            }
            if (++ktq == nk) { ktq = 0; ++jq; tile_of(p, wg, jq, nwg, rp, cb); }
        };
        for (int q = 0; q < STAGES - 1; ++q)
            if (q < Q) issue(q);
        long long tw = 0, tb = 0, ti = 0;
        for (int q = 0; q < Q; ++q) {
            const long long c0 = __builtin_amdgcn_s_memtime();
            const int younger = min(STAGES - 2, Q - 1 - q);
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * kPieces) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kPieces) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const long long c1 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            const long long c2 = __builtin_amdgcn_s_memtime();
            if (q + STAGES - 1 < Q) issue((q + STAGES - 1) % STAGES);
            const long long c3 = __builtin_amdgcn_s_memtime();
            tw += c1 - c0; tb += c2 - c1; ti += c3 - c2;
        }
        if (p.prof && lane == 0) { long long* d = p.prof + ((long long)wg * 16 + wave) * 4; d[0] = tw; d[1] = tb; d[2] = ti; d[3] = Q; }
        return;
    }
    // ---- computing waves: wave grid WM x 2, wave tile 64 x 64
    const int wm = wave >> 1, wn = wave & 1;
    f16x_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    int j = 0, kt = 0, rp = 0, cb = 0;
    tile_of(p, wg, 0, nwg, rp, cb);
    long long tb = 0, tc = 0, te = 0;
    for (int q = 0; q < Q; ++q) {
        const long long c0 = __builtin_amdgcn_s_memtime();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const long long c1 = __builtin_amdgcn_s_memtime();
        tb += c1 - c0;
        const half_t* a = lds + (q % STAGES) * kStage;
        const half_t* b = a + kA;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            h8_t wf[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) wf[nt] = lds_frag(b, wn * 64 + nt * 32 + r, 2 * ks + hh);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const h8_t xf = lds_frag(a, wm * 64 + mt * 32 + r, 2 * ks + hh);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma(wf[nt], xf, acc[mt][nt]);
            }
        }
        asm volatile("s_nop 0" ::: "memory");
        const long long c2 = __builtin_amdgcn_s_memtime();
        tc += c2 - c1;
        if (++kt == nk) {
            // epilogue from registers: lane (r, hh) holds row m = ... + r, columns (i & 3) + 8 (i >> 2) + 4 hh of each fragment
            const int m0 = rp * BM + wm * 64;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int m = m0 + mt * 32 + r;
                const bool ok = m < p.M;
                if (GEGLU) {
                    half_t* row = p.out + (long long)min(m, p.M - 1) * Nh + cb * 64 + wn * 32;
                    const half_t* bh = p.bias + cb * 64 + wn * 32;
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        h4_t ev, od;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int c0 = 16 * gp + 4 * hh + e, c1 = c0 + 8;
                            const float h0 = acc[mt][0][8 * gp + e] + (float)bh[c0], g0 = acc[mt][1][8 * gp + e] + (float)bh[Nh + c0];
                            const float h1 = acc[mt][0][8 * gp + 4 + e] + (float)bh[c1], g1 = acc[mt][1][8 * gp + 4 + e] + (float)bh[Nh + c1];
                            ev[e] = (half_t)((float)(half_t)h0 * (float)(half_t)gelu_erf((float)(half_t)g0));
                            od[e] = (half_t)((float)(half_t)h1 * (float)(half_t)gelu_erf((float)(half_t)g1));
                        }
                        const u2_t e2 = __builtin_bit_cast(u2_t, ev), o2 = __builtin_bit_cast(u2_t, od);
                        const auto s0 = __builtin_amdgcn_permlane32_swap(e2[0], o2[0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(e2[1], o2[1], false, false);
                        const u4_t w4 = {s0[0], s1[0], s0[1], s1[1]};
                        if (ok) *reinterpret_cast<u4_t*>(row + 16 * gp + 8 * hh) = w4;
                    }
                } else {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const int n0 = cb * 128 + wn * 64 + nt * 32;
                        half_t* row = p.out + (long long)min(m, p.M - 1) * p.N + n0;
                        const half_t* bh = p.bias + n0;
#pragma unroll
                        for (int gp = 0; gp < 2; ++gp) {
                            h4_t ev, od;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                ev[e] = (half_t)(acc[mt][nt][8 * gp + e] + (float)bh[16 * gp + 4 * hh + e]);
                                od[e] = (half_t)(acc[mt][nt][8 * gp + 4 + e] + (float)bh[16 * gp + 8 + 4 * hh + e]);
                            }
                            const u2_t e2 = __builtin_bit_cast(u2_t, ev), o2 = __builtin_bit_cast(u2_t, od);
                            const auto s0 = __builtin_amdgcn_permlane32_swap(e2[0], o2[0], false, false);
                            const auto s1 = __builtin_amdgcn_permlane32_swap(e2[1], o2[1], false, false);
                            const u4_t w4 = {s0[0], s1[0], s0[1], s1[1]};
                            if (ok) *reinterpret_cast<u4_t*>(row + 16 * gp + 8 * hh) = w4;
                        }
                    }
                }
            }
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[a2][b2][i] = 0.f;
            kt = 0; ++j;
            tile_of(p, wg, j, nwg, rp, cb);
            te += __builtin_amdgcn_s_memtime() - c2;
        }
    }
    if (p.prof && lane == 0) { long long* d = p.prof + ((long long)wg * 16 + wave) * 4; d[0] = tb; d[1] = tc; d[2] = te; d[3] = Q; }
}

static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

template <int WM, int STAGES, int NLOAD, bool GEGLU>
void run(const char* name, int M, int N, int K, int wgs_per_cu) {
    constexpr int BM = 64 * WM;
    const int Nout = GEGLU ? N / 2 : N;
    std::vector<half_t> hx((size_t)M * K), hw((size_t)N * K), hb(N);
    srand(1);
    for (auto& v : hx) v = (half_t)frand();
    const float sc = 1.f / sqrtf((float)K);
    for (auto& v : hw) v = (half_t)(frand() * sc * 1.7f);
    for (auto& v : hb) v = (half_t)(frand() * 0.1f);
    half_t *x, *w, *b, *o;
    (void)hipMalloc(&x, hx.size() * 2); (void)hipMalloc(&w, hw.size() * 2); (void)hipMalloc(&b, hb.size() * 2);
    (void)hipMalloc(&o, (size_t)M * Nout * 2);
    (void)hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(b, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemset(o, 0, (size_t)M * Nout * 2);
    P p{x, w, b, o, M, N, K, (M + BM - 1) / BM, GEGLU ? N / 2 / 64 : N / 128, 0, nullptr};
    p.tiles = p.rm * p.cn;
    const int lds = STAGES * (BM + 128) * BK * 2;
    const int grid = 256 * wgs_per_cu;
    auto kern = gemm_big<WM, STAGES, NLOAD, GEGLU>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(128 * WM + 64 * NLOAD), lds, 0, p);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed: %s\n", name, hipGetErrorString(hipGetLastError())); exit(1); }
    (void)hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(128 * WM + 64 * NLOAD), lds, 0, p);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
    // check a sample
    std::vector<half_t> ho((size_t)M * Nout);
    (void)hipMemcpy(ho.data(), o, ho.size() * 2, hipMemcpyDeviceToHost);
    double worst = 0; int bad = 0;
    for (int s = 0; s < 4000; ++s) {
        const int m = (s < 64) ? (M - 1 - s) : rand() % M, n = (s & 1) ? rand() % Nout : (Nout - 1 - (s % 64));
        auto dot = [&](int wn) { float a = 0.f; for (int k = 0; k < K; ++k) a += (float)hx[(size_t)m * K + k] * (float)hw[(size_t)wn * K + k]; return a + (float)hb[wn]; };
        float ref;
        if (GEGLU) {
            const float h = (float)(half_t)dot(n), g = (float)(half_t)dot(N / 2 + n);
            ref = h * (float)(half_t)(0.5f * g * (1.f + erff(g * 0.70710678f)));
        } else ref = dot(n);
        const double d = fabs((double)(float)ho[(size_t)m * Nout + n] - ref);
        if (d > worst) worst = d;
        if (d > 2e-2 + 1e-2 * fabs(ref)) ++bad;
    }
    printf("%-34s M=%6d N=%5d K=%5d  %8.1f us  %7.1f TFLOP/s  (%d tiles, %d wg/CU, lds %d KB)  max err %.4f%s\n", name, M, N, K, us, tf, p.tiles,
           wgs_per_cu, lds / 1024, worst, bad ? "  MISMATCH" : "");
    {
        long long* prof; const int grid2 = grid;
        (void)hipMalloc(&prof, (size_t)grid2 * 16 * 4 * 8); (void)hipMemset(prof, 0, (size_t)grid2 * 16 * 4 * 8);
        P p2 = p; p2.prof = prof;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(128 * WM + 64 * NLOAD), lds, 0, p2);
        (void)hipDeviceSynchronize();
        std::vector<long long> h((size_t)grid2 * 16 * 4);
        (void)hipMemcpy(h.data(), prof, h.size() * 8, hipMemcpyDeviceToHost);
        double c[3] = {0, 0, 0}, l[3] = {0, 0, 0}; double qc = 0, ql = 0;
        for (int g = 0; g < grid2; ++g)
            for (int wv = 0; wv < 2 * WM + NLOAD; ++wv) {
                const long long* d = &h[((size_t)g * 16 + wv) * 4];
                if (wv < 2 * WM) { for (int i = 0; i < 3; ++i) c[i] += d[i]; qc += d[3]; }
                else { for (int i = 0; i < 3; ++i) l[i] += d[i]; ql += d[3]; }
            }
        printf("      per K tile (memtime ticks): computing waves barrier %.0f  loop %.0f  epilogue %.0f | loaders wait %.0f  barrier %.0f  issue %.0f\n",
               c[0] / qc, c[1] / qc, c[2] / qc, l[0] / ql, l[1] / ql, l[2] / ql);
        (void)hipFree(prof);
    }
    (void)hipFree(x); (void)hipFree(w); (void)hipFree(b); (void)hipFree(o);
}

int main() {
    struct S { int M, N, K; bool geglu; };
    const S shapes[] = {{65536, 2560, 320, true}, {16384, 5120, 640, true},
                        {65536, 640, 320, false}, {65536, 1280, 1280, false}, {16384, 640, 2560, false}};
    for (const S& s : shapes) {
        if (s.geglu) {
            run<4, 3, 8, true>("256x128 3 stages 8+8 waves", s.M, s.N, s.K, 1);
            run<4, 3, 4, true>("256x128 3 stages 8+4 waves", s.M, s.N, s.K, 1);
            run<4, 2, 8, true>("256x128 2 stages 8+8 waves", s.M, s.N, s.K, 1);
            run<2, 2, 4, true>("128x128 2 stages 4+4 waves x2", s.M, s.N, s.K, 2);
            run<2, 4, 4, true>("128x128 4 stages 4+4 waves", s.M, s.N, s.K, 1);
        } else {
            run<4, 3, 8, false>("256x128 3 stages 8+8 waves", s.M, s.N, s.K, 1);
            run<4, 3, 4, false>("256x128 3 stages 8+4 waves", s.M, s.N, s.K, 1);
            run<4, 2, 8, false>("256x128 2 stages 8+8 waves", s.M, s.N, s.K, 1);
            run<2, 2, 4, false>("128x128 2 stages 4+4 waves x2", s.M, s.N, s.K, 2);
            run<2, 4, 4, false>("128x128 4 stages 4+4 waves", s.M, s.N, s.K, 1);
        }
    }
    return 0;
}
