// Plans, kernel arguments and small device helpers shared by the transform kernels (hno_dht.hip: generic / specialised kernels and
// the launch logic; hno_dht_items.hip: the item kernels for general plane sizes).
#pragma once
#include "hno_common.h"

namespace hno {

struct Axis {
    int N, m;      // size, kept modes (clamped: 2m <= N)
    int J, Js;     // cos fold range 0..J, sin fold range 1..Js
    int KT, KP;    // k tiles of 16 covering 0..m, KP = 16*KT
    int KcP, KsP;  // forward reduction lengths padded to 4
    int KmP;       // inverse reduction length (0..m) padded to 4
    int NT;        // inverse output tiles of 16 covering 0..J
    int cosF, sinF, cosI, sinI;  // offsets (floats) into the table buffer
};

struct DhtPlan {
    Axis ax[3];
    float *tables;     // device
    int table_floats;
    int K1S;           // 2*m1 + 1 signed k1 values
    int hperm, NP1;    // axis-H tables in accumulator order for the DMA forward plane kernel (odd N1, one k1 tile): offset, row-pair tiles
    int dmatab, dmatab_stride;   // kDmaTabCopies copies of [axis-W cos | axis-W sin | hperm] in lane order (one 256-byte row per register)
    int itab, itab_stride;       // inverse item kernel: copies of [axis-H: tile X][cos | sin][ks] then [axis-W: tile nt2][cos | sin][r], lane order
    int CP;            // K1S * KP2 columns per part (re / im) of the intermediate
    int MP1;           // N1 rounded up to 16
    // forward plane kernel LDS layout (floats)
    int lda2, TP, ldt, f_tabW, f_tabH, f_xs, f_T, f_lds_floats;
    // inverse plane kernel LDS layout
    int ldE, ldF, ldo, i_tabH, i_tabW, i_Es, i_Ed, i_FR, i_FI, i_O, i_lds_floats;
};

// Every wave of the DMA plane kernel loads the same ~8 KB of table rows at its start; with one copy, 2 048 waves queue on the same L2
// lines (5 us before the first MFMA, measured with in-kernel stamps).  Workgroup b reads copy b % kDmaTabCopies.
static constexpr int kDmaTabCopies = 32;

struct DhtArgs {
    DhtPlan p;
    int BC;
    float scale;
    int act;  // forward: activation whose derivative multiplies the input; inverse: epilogue act
    int dbg;  // ablation switches (timing only)
    long long *stamps;  // phase stamps (debug flag 64), else NULL
    // spectrum convention of the D kernels: 0 = Hartley block (real, [low|high] on all three axes);
    // 1 / 2 = Fourier half spectrum (B, 2, C, 2m0, 2m1, m2) with re / im planes, k2 in [0, m2):
    //   1: unit weights (rfftn forward, and the backward of rfftn)
    //   2: weights (1, 2, 2, ...) along k2 (irfftn forward, and the backward of irfftn)
    int mode;
    int C;    // channels per batch element (Fourier layout only)
    // axis keeps ALL N = 2m + 1 frequencies k = -m..m (block size 2m + 1: k >= 0 at position k, k < 0 at
    // 2m + 1 + k).  Used for un-truncated transforms of odd sizes (hno_dht3_full) and for a degenerate
    // first axis (N0 = 1, m0 = 0: the 2-D transforms).  full1 / full2 are Hartley-layout only.
    int full0, full1, full2;
    // channel-padded activations (ops.chan_stride): floats between consecutive (b, c) volumes of the spatial tensors (input of the
    // forward, output / residual of the inverse); 0 = contiguous (N0 N1 N2)
    unsigned ldbc;
    // layout of the intermediate (the plane kernels' output / input).  0: [plane][part][k1 position][k2] -- what the axis-D kernels read
    // and write.  1 (round 5, planes-only launches around the fused spectral middle): [k1 position][k2 / 4][plane][part][4] -- the
    // 16-byte pieces a middle workgroup (sample, k1, four k2 columns) needs of all planes and channels lie in ONE contiguous run
    // instead of one 128-byte line each (its D steps: 20.4 -> 16 us forward, 27.8 -> 23.3 us backward at the benchmark size)
    int zl;
    unsigned zplanes;   // BC * N0
};

// float offset of element (part, k1 position `row` in [0, K1S), column k2) of intermediate plane `plane`
__device__ __forceinline__ size_t ymid(const DhtArgs &a, int plane, int part, int row, int k2) {
    if (a.zl) return ((size_t)(row * (a.p.ax[2].KP >> 2) + (k2 >> 2)) * a.zplanes + (size_t)plane) * 8 + part * 4 + (k2 & 3);
    return (size_t)plane * (2 * a.p.CP) + (size_t)(part * a.p.K1S + row) * a.p.ax[2].KP + k2;
}

// float offset of plane (bc, n0) = `plane` of a spatial tensor
__device__ __forceinline__ size_t plane_base(const DhtArgs &a, int plane, size_t plane_elems) {
    if (a.ldbc == 0) return (size_t)plane * plane_elems;
    const int N0 = a.p.ax[0].N, bc = plane / N0;
    return (size_t)bc * a.ldbc + (size_t)(plane - bc * N0) * plane_elems;
}
// inverse kernels: the workgroup that writes the last plane of a (b, c) volume zeroes the volume's padding
__device__ __forceinline__ void zero_volume_padding(const DhtArgs &a, float *out, int plane, size_t plane_elems, int tid) {
    if (a.ldbc == 0) return;
    const int N0 = a.p.ax[0].N, bc = plane / N0;
    if (plane - bc * N0 != N0 - 1) return;
    const unsigned vol = (unsigned)((size_t)N0 * plane_elems), npad = a.ldbc - vol;
    if ((unsigned)tid < npad) out[(size_t)bc * a.ldbc + vol + tid] = 0.f;
}

constexpr int round_up_c(int a, int b) { return (a + b - 1) / b * b; }

// signed frequency -> index in the [low | high] block, or -1 if not kept
__device__ __forceinline__ int kept_pos(int k, int m, int full = 0) {
    return (k >= 0) ? (k < m + full ? k : -1) : (k >= -m ? k + 2 * m + full : -1);
}

// ---- LDS-DMA pieces (dht_fwd_plane_dma_kernel explains the scheme)
__device__ __forceinline__ void dma_piece16(const void *base, unsigned lane_byte_off, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane_byte_off), "s"(base), "s"(lds_dst) : "memory");
}
// four consecutive KiB: the instruction offset advances the global and the LDS address alike
__device__ __forceinline__ void dma_piece16x4(const void *base, unsigned lane_byte_off, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072"
                 : : "v"(lane_byte_off), "s"(base), "s"(lds_dst) : "memory");
}

// 16-byte store that goes through to memory (agent scope): no dirty line stays behind in the XCD's L2
__device__ __forceinline__ void store16_wt(float *ptr, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(ptr), "v"(v) : "memory");
}

// item plane kernels for general plane sizes (hno_dht_items.hip).  1: launched; 0: no kernel for this geometry; < 0: error
// elem_bytes 2: the planes (forward: x; inverse: out) are bf16 in memory (round 6); the inverse's addend stays fp32
int fwd_items_launch(const void *x, float *workspace, const DhtArgs &a, int BC, long long ldbc, hipStream_t s, int elem_bytes = 4);
int inv_items_launch(const void *workspace, const float *addend, void *out, const DhtArgs &a, int BC, long long ldbc, hipStream_t s, int elem_bytes = 4);
}  // namespace hno
