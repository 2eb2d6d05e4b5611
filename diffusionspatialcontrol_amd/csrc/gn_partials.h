// GroupNorm statistics from a PRODUCER's epilogue (conv3x3.hip, gemm.hip): the kernel that writes a tensor also emits, per
// (image, pixel tile, group), the (sum, sum of squares) of the fp16 values it stored, so the GroupNorm that follows is one
// launch (gn_nhwc_apply_part: it adds the <= 128 partial rows of its image itself) instead of statistics + apply - and x is read
// once less.  No workgroup waits for another: the partials cross the kernel boundary.
//
// Layout  part[b][pt][g][which][2] fp32: pt = pixel tile of image b (PT per image), g = group, which = 0 for the part of the
// group that lies in the 64-channel tile where the group BEGINS, 1 for the part in the next tile (a group of cpg <= 64 channels
// touches at most two 64-channel tiles; which = 1 exists only for groups that straddle a tile boundary - the consumer knows
// which ones from cpg).  Fixed summation orders everywhere: bit-reproducible.
#pragma once
#include "dsc_common.h"

namespace dsc_gn {

constexpr int kScratchFloats = 256 * 16 + 64 * 2;            // per-thread channel sums, then per-channel totals

// Called by all 256 computing threads of a workgroup whose epilogue thread t owns the 8-channel chunk t % 8 of a 64-channel
// tile for a fixed set of pixels: s[j] / q[j] = sum / sum of squares over those pixels of channel n0 + 8 (t % 8) + j.
// `extra_waves_sync`: nothing - waves that do not compute must call gn_tile_partials_barriers() instead.  Two barriers.
__device__ __forceinline__ void gn_tile_partials(const float (&s)[8], const float (&q)[8], float* scratch, int n0, int cpg, int G,
                                                 float* dst /* &part[b][pt][0][0][0], or nullptr: tile has no pixels */) {
    float* mine = scratch + threadIdx.x * 16;
    *reinterpret_cast<f4x_t*>(mine) = f4x_t{s[0], s[1], s[2], s[3]};
    *reinterpret_cast<f4x_t*>(mine + 4) = f4x_t{s[4], s[5], s[6], s[7]};
    *reinterpret_cast<f4x_t*>(mine + 8) = f4x_t{q[0], q[1], q[2], q[3]};
    *reinterpret_cast<f4x_t*>(mine + 12) = f4x_t{q[4], q[5], q[6], q[7]};
    __syncthreads();
    float* chtot = scratch + 256 * 16;
    if (threadIdx.x < 64) {                                   // channel n0 + t: the 32 threads that own its chunk, in thread order
        const int chunk = threadIdx.x >> 3, j = threadIdx.x & 7;
        float S = 0.f, Q = 0.f;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            const float* src = scratch + (chunk + 8 * k) * 16;
            S += src[j];
            Q += src[8 + j];
        }
        chtot[2 * threadIdx.x] = S;
        chtot[2 * threadIdx.x + 1] = Q;
    }
    __syncthreads();
    if (dst && threadIdx.x < 32) {                            // group slot: groups that touch channels [n0, n0 + 64)
        const int g = n0 / cpg + (int)threadIdx.x;
        if (g < G && g * cpg < n0 + 64) {
            const int c0 = max(g * cpg, n0) - n0, c1 = min((g + 1) * cpg, n0 + 64) - n0;
            float S = 0.f, Q = 0.f;
            for (int c = c0; c < c1; ++c) { S += chtot[2 * c]; Q += chtot[2 * c + 1]; }
            const int which = g * cpg < n0 ? 1 : 0;
            float* o = dst + (g * 2 + which) * 2;
            o[0] = S; o[1] = Q;
        }
    }
}
// the two barriers of gn_tile_partials for waves of the workgroup that take no part in it (DMA-only loader waves)
__device__ __forceinline__ void gn_tile_partials_barriers() {
    __syncthreads();
    __syncthreads();
}

}  // namespace dsc_gn
