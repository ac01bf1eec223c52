// TIMING PROBE (round 6, verdict item 1a): what would a forward kernel cost that computes the inverse plane transform of all 24
// channels INSIDE the pointwise kernel that consumes it (PadInverse -> SELU -> conv_concat, reference nets/hnosegxs.py:263-275), so
// that the transformed branch u is written once (the backward needs it) but never read back?
//
// This file is NOT a product path: hno_debug_invpw_fwd_probe issues the memory traffic, the LDS traffic and the matrix / vector
// instruction counts such a kernel needs, with simplified tables -- its outputs are not the transform's.  It exists so that the
// decision "build it or not" rests on a measurement (LESSONS 66: time the structure before the refactor).  Structure:
//   * a workgroup (8 waves) owns a CONTIGUOUS range of 32-voxel tiles of one sample (~67 tiles = ~34 rows of 65 voxels);
//   * phase 1: the axis-H step of those rows for all 24 channels on 16x16x4 MFMAs (operands from the intermediate of the fused middle,
//     L2 resident), results F[row][k2, re | im][channel] into LDS (105 KB);
//   * phase 2, per tile: the axis-W step as a 32x32x2 product with M = channels (A operand = F rows from LDS, B operand = twiddles of
//     the lanes' columns; a tile that straddles two rows runs the product twice with complementary lane masks), scale + SELU on the
//     accumulators, which ARE the B operand of the pointwise product (k-slot <-> channel as in pwconv_fwd_chain_kernel); u and the
//     block output are stored; the second input t streams through a per-wave LDS-DMA ring as in pwconv_fwd_fast_kernel.
#include "hno_common.h"

namespace hno {

typedef float ip_f32x16 __attribute__((ext_vector_type(16)));

struct IpArgs {
    const float *Y, *t, *W, *bias, *TH, *TW;
    float *u, *xi;
    int B, zplanes;
    unsigned V, tiles_per_b;
    float scale;
    int flags;      // ablations: 1 = no phase 1, 2 = no axis-W products, 4 = u not stored
};

#define IP_ROWS 36
#define IP_FCH 24          // channel pitch of F
#define IP_NW 8

__global__ __launch_bounds__(64 * IP_NW) void invpw_fwd_probe_kernel(IpArgs a) {
    extern __shared__ float ip_lds[];
    float *F = ip_lds;                                   // [IP_ROWS][32 j][24 ch]
    float *TWl = F + IP_ROWS * 32 * IP_FCH;              // [32 j][66]
    float *rings = TWl + 32 * 66;                        // IP_NW x 12 x 64 floats
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, c = lane & 31, q = lane >> 4, l15 = lane & 15;
    const unsigned V = a.V;
    const unsigned ntiles = a.tiles_per_b * a.B;
    const unsigned t0 = (unsigned)(((unsigned long long)blockIdx.x * ntiles) / gridDim.x);
    const unsigned t1 = (unsigned)(((unsigned long long)(blockIdx.x + 1) * ntiles) / gridDim.x);
    const unsigned b = t0 / a.tiles_per_b;
    const unsigned v_lo = (t0 - b * a.tiles_per_b) * 32;
    const unsigned row_lo = v_lo / 65;                   // first row (of the sample's 65 x 65 rows) this workgroup touches
    // ---- twiddles of the axis-W step into LDS
    for (int i = threadIdx.x; i < 32 * 65; i += 64 * IP_NW) TWl[(i / 65) * 66 + (i % 65)] = a.TW[i];
    // ---- phase 1: F[row][j][ch] for rows row_lo .. row_lo + IP_ROWS, all channels
    for (int job = wave; job < ((a.flags & 1) ? 0 : 3 * 24); job += IP_NW) {
        const int mt = job / 24, ch = job - mt * 24;
        const unsigned row = row_lo + mt * 16 + l15;     // this lane's A-operand row
        const unsigned n0 = (row / 65) % 65, n1 = row % 65;
        const int pl = (b * 24 + ch) * 65 + n0;          // (simplified: the tile's plane; a real kernel takes it per row)
        float thc[4], ths[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            thc[ks] = a.TH[(n1 * 16 + 4 * ks + q) * 2];
            ths[ks] = a.TH[(n1 * 16 + 4 * ks + q) * 2 + 1];
        }
        f32x4 cRe = {0.f, 0.f, 0.f, 0.f}, cIm = cRe, sRe = cRe, sIm = cRe;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k1 = 4 * ks + q;
            const bool ok = k1 <= 14;
            const int rp = ok ? 14 + k1 : 14, rm = ok ? 14 - k1 : 14;
            const size_t op = ((size_t)(rp * 4 + (l15 >> 2)) * a.zplanes + pl) * 8 + (l15 & 3);
            const size_t om = ((size_t)(rm * 4 + (l15 >> 2)) * a.zplanes + pl) * 8 + (l15 & 3);
            const float erp = a.Y[op], erm = a.Y[om], eip = a.Y[op + 4], eim = a.Y[om + 4];
            const float sR = ok ? erp + erm : 0.f, dR = ok ? erp - erm : 0.f, sI = ok ? eip + eim : 0.f, dI = ok ? eip - eim : 0.f;
            // (A and B roles as in dht_inv_items_kernel: A = folded spectrum rows over k1, B = the axis-H twiddles of 16 rows)
            cRe = mfma16(sR, thc[ks], cRe);
            sIm = mfma16(dI, ths[ks], sIm);
            cIm = mfma16(sI, thc[ks], cIm);
            sRe = mfma16(dR, ths[ks], sRe);
        }
        // D: lane (q, l15), register r: k2 = 4 q + r, row mt * 16 + l15
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = mt * 16 + l15;
            if (rr < IP_ROWS) {
                F[(rr * 32 + 4 * q + r) * IP_FCH + ch] = cRe[r] - sIm[r];
                F[(rr * 32 + 16 + 4 * q + r) * IP_FCH + ch] = cIm[r] + sRe[r];
            }
        }
    }
    __syncthreads();
    // ---- phase 2
    float wu[12], wt[12], b1[12];
#pragma unroll
    for (int ks = 0; ks < 12; ++ks) {
        const int chn = (ks & 3) + 8 * (ks >> 2) + 4 * h;
        wu[ks] = c < 24 ? a.W[c * 48 + chn] : 0.f;                  // u part: accumulator-register order
        wt[ks] = c < 24 ? a.W[c * 48 + 24 + 2 * ks + h] : 0.f;      // t part: row pairs
        b1[ks] = a.bias ? a.bias[chn] : 0.f;
    }
    const float ap = HNO_SELU_SCALE, aq = HNO_SELU_SCALE * HNO_SELU_ALPHA;
    float *ring = rings + wave * (12 * 64);
    const unsigned ring_b = (unsigned)(size_t)ring;
    const unsigned hoffV = h ? V : 0u, hoff4V = h ? 4u * V : 0u;
    const float *t_b = a.t + (size_t)b * 24 * V;
    auto issue = [&](unsigned t) {
        const unsigned v = (t - b * a.tiles_per_b) * 32 + c;
        const unsigned boff = (hoffV + (v < V ? v : 0u)) * 4u;
#pragma unroll
        for (int ks = 0; ks < 12; ++ks) dma_row_pair(t_b + (size_t)(2 * ks) * V, boff, __builtin_amdgcn_readfirstlane(ring_b + ks * 256));
    };
    unsigned t = t0 + wave;
    if (t < t1) issue(t);
    for (; t < t1; t += IP_NW) {
        const unsigned v0 = (t - b * a.tiles_per_b) * 32, v = v0 + c;
        const unsigned ra = v0 / 65;                                   // wave-uniform: first row of the tile
        const unsigned split = (ra + 1) * 65 - v0;                     // lanes c >= split belong to row ra + 1
        const bool second = split < 32;
        const bool in_b = (unsigned)c >= split;
        const unsigned n2 = in_b ? c - split : v - ra * 65;
        const float *Fa = F + ((ra - row_lo) * 32 + h) * IP_FCH + (c < 24 ? c : 23);
        const float *Tl = TWl + h * 66 + n2;
        float fa[16], fb[16], tw[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            fa[ks] = Fa[(2 * ks) * IP_FCH];
            tw[ks] = Tl[(2 * ks) * 66];
        }
        if (second) {
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) fb[ks] = Fa[(32 + 2 * ks) * IP_FCH];
        }
        ip_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (a.flags & 2) {
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) acc[ks] = fa[ks] + tw[ks];
        } else {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ks], in_b ? 0.f : tw[ks], acc, 0, 0, 0);
        }
        if (second && !(a.flags & 2)) {
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[ks], in_b ? tw[ks] : 0.f, acc, 0, 0, 0);
        }
        // u = selu(scale * inverse): registers r < 12 hold channels (r & 3) + 8 (r >> 2) + 4 h
        float uv[12];
#pragma unroll
        for (int r = 0; r < 12; ++r) uv[r] = acc[r] * a.scale;
        selu_like_regs<12>(uv, ap, aq);
        float *u_l = a.u + (size_t)b * 24 * V + (hoff4V + v);
        const bool vin = v < V;
#pragma unroll
        for (int r = 0; r < 12; ++r)
            if (vin && !(a.flags & 4)) u_l[(size_t)((r & 3) + 8 * (r >> 2)) * V] = uv[r];
        // the second input's rows have landed
        dma_wait<0>();
        const float *sl = ring + lane;
        float tv[12];
#pragma unroll
        for (int ks = 0; ks < 12; ++ks) tv[ks] = sl[ks * 64];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (t + IP_NW < t1) issue(t + IP_NW);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 12; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wu[ks], uv[ks], acc, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 12; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[ks], tv[ks], acc, 0, 0, 0);
        float xv[12];
#pragma unroll
        for (int r = 0; r < 12; ++r) xv[r] = acc[r] + b1[r];
        selu_like_regs<12>(xv, ap, aq);
        float *x_l = a.xi + (size_t)b * 24 * V + (hoff4V + v);
#pragma unroll
        for (int r = 0; r < 12; ++r)
            if (vin) x_l[(size_t)((r & 3) + 8 * (r >> 2)) * V] = xv[r];
    }
}

}  // namespace hno

using namespace hno;

// Timing probe only (see the head of this file): NOT the transform's results.  Y: the fused middle's workspace (zplanes = B * 24 * 65
// planes), t (B, 24, V) with channel stride V, W (24, 48), bias (24), TH (65 x 16 x 2), TW (32 x 65) any finite tables; u, xi (B, 24, V).
extern "C" int hno_debug_invpw_fwd_probe(const float *Y, const float *t, const float *W, const float *bias, const float *TH, const float *TW,
                                         float *u, float *xi, int B, long long V, float scale, int grid, void *stream) {
    const int flags = grid >> 16;
    grid &= 0xffff;
    HNO_REQUIRE(Y && t && W && TH && TW && u && xi && B > 0 && V > 0 && V % 32 == 0, "hno_debug_invpw_fwd_probe: bad argument");
    IpArgs a;
    a.Y = Y; a.t = t; a.W = W; a.bias = bias; a.TH = TH; a.TW = TW; a.u = u; a.xi = xi;
    a.B = B; a.zplanes = B * 24 * 65; a.V = (unsigned)V; a.tiles_per_b = (unsigned)(V / 32); a.scale = scale; a.flags = flags;
    const size_t lds = sizeof(float) * ((size_t)IP_ROWS * 32 * IP_FCH + 32 * 66 + IP_NW * 12 * 64);
    static int attr = -1;
    if (attr != current_device()) {
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)invpw_fwd_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = current_device();
    }
    if (grid <= 0) grid = 256;
    hipLaunchKernelGGL(invpw_fwd_probe_kernel, dim3(grid), dim3(64 * IP_NW), lds, (hipStream_t)stream, a);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}
