// bf16 matrix-core path of the V-Net-DS convolutions (reference nets/architectures.py:26-252 run under
// torch.autocast(bfloat16), experiments/train_test.py:79,154-168): convolutions and their gradients take bf16
// operands and accumulate in fp32 on v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16, GroupNorm statistics and
// the normalisation arithmetic stay fp32, parameters and their gradients stay fp32.
//
// Layout: activations are CHANNELS-LAST bf16, x[b][d][h][w][c] ("NDHWC"), C a multiple of 8.  The K index of every
// convolution GEMM is (tap, input channel), so with channels last one lane's MFMA operand -- 8 consecutive k -- is
// one 16-byte load, and one lane's 16 accumulator rows store as channel-contiguous runs.  The weight gradient sums
// over voxels instead: its operands are read from the same channels-last tiles in LDS with the transposing
// ds_read_b64_tr_b16, so no second (channels-first) copy of any tensor exists.
//
//   hno_cb_pack_weights   fp32 parameter (any of the four operator roles) -> bf16 [q = tap * Cin/8 + c8][CoutP][8]
//   hno_cb_conv           gather-GEMM: conv / strided conv / ConvTranspose / their input gradients, kernel 1, 2 or 3,
//                         two concatenated inputs, fused bias, bf16 store, fused GroupNorm partial statistics,
//                         split-K over taps for the deep (small-grid, wide-channel) levels
//   hno_cb_wgrad          weight gradient, LDS halo tile + transposing reads, slab partials + ordered reduce
//   hno_cb_gn_*           GroupNorm(1, C) + ELU/SELU forward (two-branch residual sum fused) and backward
//   hno_cb_pack_input / hno_cb_unpack   fp32 NCDHW <-> bf16 NDHWC at the two ends of the network
#include "hno_common.h"
#include <stdlib.h>

namespace hno {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16b __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;   // storage type (raw bits)

__device__ __forceinline__ bf16_t f2bf(float f) {
    return __builtin_bit_cast(bf16_t, (__bf16)f);   // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
}
__device__ __forceinline__ float bf2f(bf16_t v) {
    return __builtin_bit_cast(float, (unsigned)v << 16);
}

// ------------------------------------------------------------------------------------------------ weights
// src: fp32 [C0][C1][T] (T = ks^3 taps).  dst: bf16 [nq2][CoP][8], q = tap * nC8 + c8, element j = GEMM input channel
// 8 c8 + j; nq2 = nq rounded up to even (a K-step is two chunks), CoP = Cout rounded up to 32; pads are zero.
__global__ __launch_bounds__(256) void cb_pack_weights_kernel(const float *__restrict__ w, bf16_t *__restrict__ dst, int C0, int C1,
                                                             int T, int out_is_axis0, int Ci, int Co, int CoP, int nq2) {
    const int nC8 = Ci / 8;
    const long long n = (long long)nq2 * CoP * 8;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256) {
        const int j = (int)(idx & 7);
        const int o = (int)((idx >> 3) % CoP);
        const int q = (int)((idx >> 3) / CoP);
        const int t = q / nC8, i = (q % nC8) * 8 + j;
        float v = 0.f;
        if (t < T && o < Co) {
            const int c0 = out_is_axis0 ? o : i, c1 = out_is_axis0 ? i : o;
            v = w[((size_t)c0 * C1 + c1) * T + t];
        }
        dst[idx] = f2bf(v);
    }
}

// ------------------------------------------------------------------------------------------------ gather GEMM
struct CbArgs {
    const bf16_t *xa, *xb;     // channels-last inputs; the GEMM input channels are [xa's Ca | xb's Cb] (fused concat)
    int Ca, Cb;
    const bf16_t *w;           // packed weights
    const float *bias;         // fp32 per output channel or null
    bf16_t *y;                 // channels-last output (null when split-K writes partials)
    bf16_t *y2;                // round 5: second output tensor for channels >= csplit (the two inputs of a decoder convolution get their
    int csplit;                // gradients as two contiguous tensors: no channel-split copies behind the input-gradient GEMM); 0: one output
    float *part;               // split-K: fp32 partials [z][b][v][Cout]
    float *stats;              // per-block (sum, sum of squares) of the ROUNDED outputs: [b][gridDim.x * gridDim.y][2], or null
    int B, Cout, CoP;
    int Di, Hi, Wi, Do, Ho, Wo;
    int ks, stride, pad, frac;
    int ntaps, nq2;            // ks^3, padded chunk count
    int ksplit;                // gridDim.z slices of the chunk range
    // stride-2 fractional gather (ConvTranspose forward, input gradient of a strided conv) by PARITY CLASS: an output voxel
    // (od, oh, ow) only meets the taps with t = (o + pad) mod 2 per axis -- 1 or 2 of 3 per axis, 27 / 8 taps on average.  With the
    // voxels of a workgroup taken from ONE class (od % 2, oh % 2, ow % 2) the valid taps are workgroup-uniform and the others are
    // not visited at all (round 3 multiplied zeros for them: 8x the flops; decode_layers.0.0 228 us).  cls_blk0[c]: first
    // blockIdx.x of class c (class c = 4 pz + 2 py + px; c = 8: end).
    int cls;
    int cls_blk0[9];
};

// One wave = MT tiles of 32 consecutive output voxels (flattened d,h,w of one sample) x NT tiles of 32 output channels.
// A operand (activations): lane (r = l & 31, h = l >> 5) holds voxel r's 8 channels of chunk q = 2 s + h: ONE 16-byte
// load from the channels-last tensor at the tap's input voxel (zero when the tap falls into the padding).
// B operand (weights): lane (r, h) holds output channel r's 8 k of chunk q: one 16-byte load from the packed array.
template <int MT, int NT>
__global__ __launch_bounds__(256) void cb_gather_kernel(CbArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.z / a.ksplit, kz = blockIdx.z % a.ksplit;
    const int Vo = a.Do * a.Ho * a.Wo;
    const int n0 = blockIdx.y * 32 * NT;
    const int Cin = a.Ca + a.Cb, nC8 = Cin >> 3, nCa8 = a.Ca >> 3;
    const size_t Vi = (size_t)a.Di * a.Hi * a.Wi;
    const bf16_t *xa = a.xa + (size_t)b * Vi * a.Ca;
    const bf16_t *xb = a.xb ? a.xb + (size_t)b * Vi * a.Cb : nullptr;
    // parity-class mode: this workgroup's class, the class's sub-grid and its valid taps (all workgroup-uniform)
    int cls = 0, blk_local = blockIdx.x, pz = 0, py = 0, px = 0, Dc = a.Do, Hc = a.Ho, Wc = a.Wo;
    unsigned long long tap_list = 0;      // 5 bits per valid tap
    int ntl = a.ntaps;
    if (a.cls) {
        while (cls < 7 && (int)blockIdx.x >= a.cls_blk0[cls + 1]) ++cls;
        blk_local = blockIdx.x - a.cls_blk0[cls];
        pz = cls >> 2; py = (cls >> 1) & 1; px = cls & 1;
        Dc = (a.Do + 1 - pz) >> 1; Hc = (a.Ho + 1 - py) >> 1; Wc = (a.Wo + 1 - px) >> 1;
        ntl = 0;
        for (int td = (pz + a.pad) & 1; td < a.ks; td += 2)
            for (int th = (py + a.pad) & 1; th < a.ks; th += 2)
                for (int tw = (px + a.pad) & 1; tw < a.ks; tw += 2) {
                    tap_list |= (unsigned long long)((td * a.ks + th) * a.ks + tw) << (5 * ntl);
                    ++ntl;
                }
    }
    const int Vc = a.cls ? Dc * Hc * Wc : Vo;
    const int v0 = (blk_local * 4 + wave) * 32 * MT;
    const bool wave_active = v0 < Vc;

    int od[MT], oh[MT], ow[MT], lin[MT];      // lin: flattened output voxel of this lane's voxel (-1: none)
    bool vok[MT];
    const int HWc = Hc * Wc;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int v = v0 + 32 * m + r;
        vok[m] = v < Vc;
        const int vv = vok[m] ? v : 0;
        od[m] = vv / HWc;
        const int rem = vv - od[m] * HWc;
        oh[m] = rem / Wc;
        ow[m] = rem - oh[m] * Wc;
        if (a.cls) { od[m] = 2 * od[m] + pz; oh[m] = 2 * oh[m] + py; ow[m] = 2 * ow[m] + px; }
        lin[m] = vok[m] ? (od[m] * a.Ho + oh[m]) * a.Wo + ow[m] : -1;
    }
    f32x16b acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

    // chunk range of this K slice (whole K-steps).  Virtual chunk qv = tap index x nC8 + c8 over the taps this workgroup visits
    // (all of them, or the class's valid ones); the packed weights are addressed by the real q = tap x nC8 + c8.
    const int nqv = ntl * nC8;
    const int nsteps = (nqv + 1) >> 1;
    const int s_lo = (int)((long long)nsteps * kz / a.ksplit), s_hi = (int)((long long)nsteps * (kz + 1) / a.ksplit);
    if (wave_active && s_lo < s_hi) {
        int qv = 2 * s_lo + h;
        int tapi = qv / nC8, c8 = qv - tapi * nC8;
        int tap = 0, q = 0;
        long long off[MT];    // element offset of the tap's input voxel (channel 0) in a tensor with 1 channel; < 0: padding
        auto set_tap = [&](int ti) {
            const bool live = ti < ntl;
            const int t = a.cls ? (int)((tap_list >> (5 * (live ? ti : 0))) & 31) : (live ? ti : 0);
            tap = t;
            const int ks2 = a.ks * a.ks;
            const int td = t / ks2, th = (t / a.ks) % a.ks, tw = t % a.ks;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                int zi, yi, xi;
                bool ok = vok[m] && live;
                if (!a.frac) {
                    zi = a.stride * od[m] - a.pad + td;
                    yi = a.stride * oh[m] - a.pad + th;
                    xi = a.stride * ow[m] - a.pad + tw;
                    ok = ok && zi >= 0 && zi < a.Di && yi >= 0 && yi < a.Hi && xi >= 0 && xi < a.Wi;
                } else {
                    const int nz = od[m] + a.pad - td, ny = oh[m] + a.pad - th, nx = ow[m] + a.pad - tw;
                    ok = ok && nz >= 0 && ny >= 0 && nx >= 0;
                    if (a.stride == 2) {
                        ok = ok && !((nz | ny | nx) & 1);
                        zi = nz >> 1; yi = ny >> 1; xi = nx >> 1;
                    } else {
                        zi = nz; yi = ny; xi = nx;
                    }
                    ok = ok && zi < a.Di && yi < a.Hi && xi < a.Wi;
                }
                off[m] = ok ? ((long long)zi * a.Hi + yi) * a.Wi + xi : -1;
            }
        };
        set_tap(tapi);
        q = tap * nC8 + c8;
        const uint4 zero4 = make_uint4(0, 0, 0, 0);
        auto load_a = [&](int m) -> uint4 {
            if (off[m] < 0) return zero4;
            const bf16_t *p = c8 < nCa8 ? xa + (size_t)off[m] * a.Ca + c8 * 8 : xb + (size_t)off[m] * a.Cb + (c8 - nCa8) * 8;
            return *reinterpret_cast<const uint4 *>(p);
        };
        auto load_b = [&](int n) -> uint4 {      // (a chunk past the last tap multiplies zero activations: any finite weights do)
            return *reinterpret_cast<const uint4 *>(a.w + ((size_t)q * a.CoP + n0 + 32 * n + r) * 8);
        };
        uint4 av[MT], bv[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) av[m] = load_a(m);
#pragma unroll
        for (int n = 0; n < NT; ++n) bv[n] = load_b(n);
        for (int s = s_lo; s < s_hi; ++s) {
            uint4 an[MT], bn[NT];
            const bool more = s + 1 < s_hi;
            if (more) {     // next K-step's operands are in flight during this step's MFMAs
                c8 += 2;
                if (c8 >= nC8) {
                    c8 -= nC8;
                    ++tapi;
                    if (c8 >= nC8) { c8 -= nC8; ++tapi; }      // nC8 == 1
                    set_tap(tapi);
                }
                q = tap * nC8 + c8;
#pragma unroll
                for (int m = 0; m < MT; ++m) an[m] = load_a(m);
#pragma unroll
                for (int n = 0; n < NT; ++n) bn[n] = load_b(n);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bv[n]), __builtin_bit_cast(bf16x8, av[m]),
                                                                        acc[m][n], 0, 0, 0);      // weights as the A operand: see the epilogue
            if (more) {
#pragma unroll
                for (int m = 0; m < MT; ++m) av[m] = an[m];
#pragma unroll
                for (int n = 0; n < NT; ++n) bv[n] = bn[n];
            }
        }
    }
    // epilogue.  C/D (weights = A, activations = B): column (lane & 31) = this lane's voxel, rows (i & 3) + 8 (i >> 2) + 4 h = output
    // channel within the 32-channel tile: registers 4 g .. 4 g + 3 are 4 CONSECUTIVE channels 8 g + 4 h .. + 3 -- one 8-byte store each
    // (16-byte for the fp32 split-K partials).  Until round 6 the operands were the other way round (column = channel, rows = voxels):
    // sixteen 2-byte stores per tile and lane, the bound of the 1x1x1 / strided layers of the shallow levels (24 MB outputs at 1 TB/s)
    float ssum = 0.f, ssq = 0.f;
    if (wave_active) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int v = lin[m];
            if (v < 0) continue;
            const size_t vb = (size_t)b * Vo + v;
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = n0 + 32 * n + 8 * g + 4 * h;
                    if (ch >= a.Cout) continue;          // (Cout is a multiple of 8 and ch of 4: a group is all in or all out)
                    const bool wb = a.bias && kz == 0;
                    float val[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[e] = acc[m][n][4 * g + e] + (wb ? a.bias[ch + e] : 0.f);
                    if (a.part) {
                        *reinterpret_cast<float4 *>(a.part + (((size_t)kz * a.B + b) * Vo + v) * a.Cout + ch) = make_float4(val[0], val[1], val[2], val[3]);
                        continue;
                    }
                    bf16_t o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = f2bf(val[e]);
                        const float f = bf2f(o[e]);
                        ssum += f;
                        ssq = fmaf(f, f, ssq);
                    }
                    const uint2 packed = make_uint2((unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16));
                    if (a.csplit && ch >= a.csplit) *reinterpret_cast<uint2 *>(a.y2 + vb * (a.Cout - a.csplit) + (ch - a.csplit)) = packed;
                    else *reinterpret_cast<uint2 *>(a.y + vb * (a.csplit ? a.csplit : a.Cout) + ch) = packed;
                }
        }
    }
    if (a.stats && !a.part) {
        __shared__ float red[8];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ssum += __shfl_xor(ssum, o);
            ssq += __shfl_xor(ssq, o);
        }
        if (lane == 0) { red[wave * 2] = ssum; red[wave * 2 + 1] = ssq; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int nblk = gridDim.x * gridDim.y;
            float *dst = a.stats + ((size_t)b * nblk + blockIdx.y * gridDim.x + blockIdx.x) * 2;
            dst[0] = red[0] + red[2] + red[4] + red[6];
            dst[1] = red[1] + red[3] + red[5] + red[7];
        }
    }
}

// ------------------------------------------------------------------------------------------------ halo-tile GEMM
// Stride-1 3x3x3 convolutions (and their input gradients: the same operator with the taps mirrored) on grids wide enough to
// fill tiles: the shallow levels, where 85 % of the V-Net's flops are.  The gather kernel above re-reads every input
// element once per tap through the texture path (27 x 16-byte gathers per element; measured 170 TFLOP/s, TA-bound).  Here a
// workgroup owns (b, od, a band of TH output rows): the 3 planes x (TH + 2) rows x (W + 2) columns its taps reach are
// staged ONCE per 24-channel chunk in LDS as [position][24 ch] (48-byte pitch: consecutive positions sit in consecutive
// 16-byte bank slots 3 apart, so a ds_read_b128 of 32 positions is conflict-free), and tap (td, th, tw) of output position
// p = hh * S + ww (S = W + 2) is image position  p + (td * (TH + 2) + th) * S + tw : a flat shift, no index arithmetic
// per element.  Outputs at the two pad columns of each row are computed and dropped.  Weights stream from L2 in the packed
// [chunk][CoP][8] layout (the same addresses in every workgroup), one K-step ahead.  LDS per workgroup <= 52 KB so that
// three workgroups share a CU: one stages while the others multiply.
struct ChArgs {
    const bf16_t *xa, *xb;
    int Ca, Cb;
    const bf16_t *w;
    const float *bias;
    bf16_t *y, *y2;           // y2 / csplit: as CbArgs
    int csplit;
    float *stats;
    int B, Cout, CoP, D, H, W;
    int TH, S, nbands, npos;
    int flip;                 // 1: input gradient (taps mirrored)
    int ksplit;               // gridDim.z slices of the 24-channel chunks; > 1: fp32 partial sums to `part` [z][b][v][Cout]
    float *part;
    int dbg;                  // ablation switches (hno_set_debug): 512 skip the staging, 1024 skip the K loop (results WRONG)
};

template <int MT, int NT, bool WIDE>
__global__ __launch_bounds__(256) void cb_halo_kernel(ChArgs a) {
    extern __shared__ uint4 img[];                        // [npos][3] image, then the K-step table int2[96]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int band = blockIdx.x % a.nbands;
    const int od = (blockIdx.x / a.nbands) % a.D;
    const int b = blockIdx.x / (a.nbands * a.D);
    const int oh0 = band * a.TH;
    const int n0 = blockIdx.y * 32 * NT;
    const int Cin = a.Ca + a.Cb, nC8 = Cin >> 3, nchunk = Cin / 24;
    const int rows = a.TH + 2, S = a.S;
    const int p0 = wave * 32 * MT;                         // this wave's first output position
    // K-step table, the same for every wave: entry 2 s + h = (image offset in uint4 of the tap's shift and 8-channel part,
    // weight offset in elements of chunk (tap, part); -1: padding step).  Keeps the K loop free of index arithmetic (the
    // divisions by 3 / 9 per operand made the loop VALU-bound at 5x its MFMA time).
    int2 *tab = reinterpret_cast<int2 *>(img + (size_t)a.npos * 3);
    if (threadIdx.x < 96) {
        const int ql = threadIdx.x;
        int2 e = make_int2(0, -1);
        if (ql < 81) {
            const int tap = ql / 3, j = ql - tap * 3;
            int td = tap / 9, th = (tap / 3) % 3, tw = tap % 3;
            if (a.flip) { td = 2 - td; th = 2 - th; tw = 2 - tw; }
            e.x = ((td * rows + th) * S + tw) * 3 + j;
            e.y = (tap * nC8 + j) * a.CoP * 8;
        }
        tab[ql] = e;
    }
    f32x16b acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
    const size_t plane = (size_t)a.H * a.W;
    const int row_items = 3 * S;                           // uint4 per image row
    const int nrows_img = (a.npos + S - 1) / S;            // image rows incl. the zero slack behind the third plane
    const int kz = blockIdx.z;
    const int cc_lo = nchunk * kz / a.ksplit, cc_hi = nchunk * (kz + 1) / a.ksplit;
    for (int cc = cc_lo; cc < cc_hi; ++cc) {
        const bool from_a = cc * 24 < a.Ca;
        const bf16_t *src = from_a ? a.xa + (size_t)b * a.D * plane * a.Ca + cc * 24 : a.xb + (size_t)b * a.D * plane * a.Cb + (cc * 24 - a.Ca);
        const int Cs = from_a ? a.Ca : a.Cb;
        __syncthreads();                                   // the previous chunk's reads are done
        if (!(a.dbg & 512))
            // one wave per image row (row validity is wave-uniform).  On the widest grid the loads of two rows (8 per lane) are
            // issued back to back from clamped addresses and zeroed by a select afterwards: a load inside an `if` followed by
            // its LDS store is one exposed global round trip per element (20 of that layer's 120 us)
            if constexpr (WIDE) {
                // wide rows (level 0): two rows of this wave per pass, 8 loads per lane in flight
                for (int row0 = wave; row0 < nrows_img; row0 += 8) {
                    uint4 v[2][4];
                    bool okk[2][4];
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) {
                        const int row = row0 + 4 * rb;
                        const int pl = row / rows, rr = row - pl * rows;
                        const int id = od - 1 + pl, ih = oh0 - 1 + rr;
                        const bool rok = row < nrows_img && pl < 3 && id >= 0 && id < a.D && ih >= 0 && ih < a.H;
                        const bf16_t *rsrc = src + ((size_t)(rok ? id : 0) * plane + (size_t)(rok ? ih : 0) * a.W) * Cs;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int it = lane + 64 * k;
                            const int cx = it / 3, part = it - cx * 3;
                            const int iw = cx - 1;
                            okk[rb][k] = rok && it < row_items && iw >= 0 && iw < a.W;
                            v[rb][k] = *reinterpret_cast<const uint4 *>(okk[rb][k] ? rsrc + (size_t)iw * Cs + part * 8 : src);
                        }
                    }
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) {
                        const int row = row0 + 4 * rb;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int it = lane + 64 * k;
                            const int dst = row * row_items + it;
                            if (row < nrows_img && it < row_items && dst < a.npos * 3) img[dst] = okk[rb][k] ? v[rb][k] : make_uint4(0, 0, 0, 0);
                        }
                    }
                }
            } else {
                // narrow rows (levels 1, 2; many small workgroups): one row per pass (measured faster there than the batched form)
                for (int row = wave; row < nrows_img; row += 4) {
                    const int pl = row / rows, rr = row - pl * rows;
                    const int id = od - 1 + pl, ih = oh0 - 1 + rr;
                    const bool rok = pl < 3 && id >= 0 && id < a.D && ih >= 0 && ih < a.H;
                    const bf16_t *rsrc = src + ((size_t)(rok ? id : 0) * plane + (size_t)(rok ? ih : 0) * a.W) * Cs;
                    for (int it = lane; it < row_items; it += 64) {
                        const int cx = it / 3, part = it - cx * 3;
                        const int iw = cx - 1;
                        uint4 v = make_uint4(0, 0, 0, 0);
                        if (rok && iw >= 0 && iw < a.W) v = *reinterpret_cast<const uint4 *>(rsrc + (size_t)iw * Cs + part * 8);
                        const int dst = row * row_items + it;
                        if (dst < a.npos * 3) img[dst] = v;
                    }
                }
            }
        __syncthreads();
        if (a.dbg & 1024) continue;
        // 81 chunks of 8 k (27 taps x 3) -> 41 K-steps of 16 (padded to 48); lane half h takes chunk 2 s + h.  The weight
        // operand streams from L2: a ring of PD K-steps of weight fragments is kept in flight.
        constexpr int PD = 8;
        const bf16_t *wl = a.w + ((size_t)cc * 3 * a.CoP + n0 + r) * 8;
        const int lane_a = (p0 + r) * 3;
        auto load_b = [&](int s, uint4 *bv) {
            const int off = tab[2 * s + h].y;
#pragma unroll
            for (int n = 0; n < NT; ++n) bv[n] = off >= 0 ? *reinterpret_cast<const uint4 *>(wl + off + n * 256) : make_uint4(0, 0, 0, 0);
        };
        auto load_a = [&](int s, uint4 *av) {
            const int off = tab[2 * s + h].x + lane_a;
#pragma unroll
            for (int m = 0; m < MT; ++m) av[m] = img[off + 96 * m];
        };
        uint4 ring[PD][NT];
#pragma unroll
        for (int u = 0; u < PD; ++u) load_b(u, ring[u]);
        uint4 av[MT];
        load_a(0, av);
        for (int s0 = 0; s0 < 48; s0 += PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                const int s = s0 + u;
                uint4 an[MT];
                load_a(s + 1 < 41 ? s + 1 : 40, an);
                if (s < 41) {
                    // A operand = weights (rows = output channels), B operand = activations (columns = positions)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring[u][n]),
                                                                                __builtin_bit_cast(bf16x8, av[m]), acc[m][n], 0, 0, 0);
                }
                if (s + PD < 41) load_b(s + PD, ring[u]);
#pragma unroll
                for (int m = 0; m < MT; ++m) av[m] = an[m];
            }
        }
    }
    // epilogue.  C/D: column (lane & 31) = position p0 + 32 m + r, rows (i & 3) + 8 (i >> 2) + 4 h = output channel within the
    // 32-channel tile: registers 4 g .. 4 g + 3 are 4 CONSECUTIVE channels 8 g + 4 h .. + 3 of this lane's position: one
    // 8-byte store each (the transposed orientation needed 2-byte stores: 35 us of a 220 us layer)
    float ssum = 0.f, ssq = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int p = p0 + 32 * m + r;
        const int hh = p / S, ww = p - hh * S;
        const int oh = oh0 + hh;
        const bool pok = hh < a.TH && oh < a.H && ww < a.W;
        const size_t ypos = ((size_t)b * a.D + od) * plane + (size_t)(pok ? oh : 0) * a.W + (pok ? ww : 0);
        bf16_t *yrow = a.y + ypos * (a.csplit ? a.csplit : a.Cout);
        bf16_t *yrow2 = a.csplit ? a.y2 + ypos * (a.Cout - a.csplit) - a.csplit : nullptr;      // (indexed by the GEMM's channel ch >= csplit)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = n0 + 32 * n + 8 * g + 4 * h;
                if (!pok || ch >= a.Cout) continue;          // Cout is a multiple of 8 and ch of 4: a group is all in or all out ... (ch + 3 < Cout when ch < Cout since Cout % 4 == 0)
                if (a.part) {      // split-K: raw fp32 partial sums (bias from slice 0), finished by cb_splitk_finish_kernel
                    float4 o;
                    const bool wb = a.bias && kz == 0;
                    o.x = acc[m][n][4 * g + 0] + (wb ? a.bias[ch + 0] : 0.f);
                    o.y = acc[m][n][4 * g + 1] + (wb ? a.bias[ch + 1] : 0.f);
                    o.z = acc[m][n][4 * g + 2] + (wb ? a.bias[ch + 2] : 0.f);
                    o.w = acc[m][n][4 * g + 3] + (wb ? a.bias[ch + 3] : 0.f);
                    const size_t v = ((size_t)b * a.D + od) * plane + (size_t)oh * a.W + ww;
                    *reinterpret_cast<float4 *>(a.part + ((size_t)kz * a.B * a.D * plane + v) * a.Cout + ch) = o;
                    continue;
                }
                bf16_t o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = f2bf(acc[m][n][4 * g + e] + (a.bias ? a.bias[ch + e] : 0.f));
                    const float f = bf2f(o[e]);
                    ssum += f;
                    ssq = fmaf(f, f, ssq);
                }
                *reinterpret_cast<uint2 *>(((a.csplit && ch >= a.csplit) ? yrow2 : yrow) + ch) = make_uint2((unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16));
            }
    }
    if (a.stats && !a.part) {
        __shared__ float red[8];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ssum += __shfl_xor(ssum, o);
            ssq += __shfl_xor(ssq, o);
        }
        if (lane == 0) { red[wave * 2] = ssum; red[wave * 2 + 1] = ssq; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int per_b = a.D * a.nbands;
            const int nblk = per_b * gridDim.y;
            float *dst = a.stats + ((size_t)b * nblk + blockIdx.y * per_b + (blockIdx.x % per_b)) * 2;
            dst[0] = red[0] + red[2] + red[4] + red[6];
            dst[1] = red[1] + red[3] + red[5] + red[7];
        }
    }
}

// split-K finish: y = bf16(sum_z part[z] ) (bias was added by slice 0), + GroupNorm partial statistics
__global__ __launch_bounds__(256) void cb_splitk_finish_kernel(const float *__restrict__ part, bf16_t *__restrict__ y, float *__restrict__ stats,
                                                              int ksplit, long long per_sample, int B, bf16_t *__restrict__ y2 = nullptr,
                                                              int csplit = 0, int Cout = 0) {
    const int b = blockIdx.y;
    const size_t slice = (size_t)B * per_sample;
    float ssum = 0.f, ssq = 0.f;
    // four consecutive elements per thread (channel counts are multiples of 8, so they share a voxel and a side of csplit), the slices four
    // at a time with independent sums: a thread that added <= 32 slices of ONE element, one load after the other, took 10-12 us for the
    // 10 MB of partials of a deepest-level layer
    for (long long i4 = (long long)blockIdx.x * 256 + threadIdx.x; i4 < (per_sample >> 2); i4 += (long long)gridDim.x * 256) {
        const long long i = i4 << 2;
        const float4 *src = reinterpret_cast<const float4 *>(part + (size_t)b * per_sample + i);
        const size_t step = slice >> 2;
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
        int z = 0;
        for (; z + 3 < ksplit; z += 4) {
            const float4 v0 = src[(size_t)z * step], v1 = src[(size_t)(z + 1) * step], v2 = src[(size_t)(z + 2) * step], v3 = src[(size_t)(z + 3) * step];
            a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
            a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
            a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
            a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
        }
        for (; z < ksplit; ++z) {
            const float4 v0 = src[(size_t)z * step];
            a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        }
        const float sv[4] = {(a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w)};
        bf16_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o[k] = f2bf(sv[k]);
            const float f = bf2f(o[k]);
            ssum += f;
            ssq = fmaf(f, f, ssq);
        }
        const uint2 packed = make_uint2((unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16));
        if (csplit) {      // two output tensors (CbArgs.csplit)
            const long long v = i / Cout;
            const int ch = (int)(i - v * Cout);
            const size_t vb = (size_t)b * (per_sample / Cout) + v;
            if (ch >= csplit) *reinterpret_cast<uint2 *>(y2 + vb * (Cout - csplit) + (ch - csplit)) = packed;
            else *reinterpret_cast<uint2 *>(y + vb * csplit + ch) = packed;
        } else
            *reinterpret_cast<uint2 *>(y + (size_t)b * per_sample + i) = packed;
    }
    if (stats) {
        __shared__ float red[8];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ssum += __shfl_xor(ssum, o);
            ssq += __shfl_xor(ssq, o);
        }
        if (lane == 0) { red[wave * 2] = ssum; red[wave * 2 + 1] = ssq; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float *dst = stats + ((size_t)b * gridDim.x + blockIdx.x) * 2;
            dst[0] = red[0] + red[2] + red[4] + red[6];
            dst[1] = red[1] + red[3] + red[5] + red[7];
        }
    }
}

// (sum, sumsq) partials -> (mean, rstd) of one sample; fp64 accumulation, fixed order (256 threads; `sh` = 512 doubles of LDS).
// Every thread returns the same pair.
__device__ __forceinline__ void cb_gn_finalize_block(const float *__restrict__ part, int nblk, double count, float eps, double *sh,
                                                     float &mean_out, float &rstd_out) {
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) {
        s += (double)part[(size_t)i * 2];
        q += (double)part[(size_t)i * 2 + 1];
    }
    sh[threadIdx.x] = s;
    sh[256 + threadIdx.x] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sh[threadIdx.x] += sh[threadIdx.x + o];
            sh[256 + threadIdx.x] += sh[256 + threadIdx.x + o];
        }
        __syncthreads();
    }
    const double mean = sh[0] / count;
    double var = sh[256] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    mean_out = (float)mean;
    rstd_out = (float)(1.0 / sqrt(var + (double)eps));
    __syncthreads();      // sh may be reused
}
__global__ __launch_bounds__(256) void cb_gn_finalize_kernel(const float *__restrict__ part, int nblk, double count, float eps,
                                                            float *__restrict__ mr) {
    const int b = blockIdx.x;
    __shared__ double sh[512];
    float mean, rstd;
    cb_gn_finalize_block(part + (size_t)b * nblk * 2, nblk, count, eps, sh, mean, rstd);
    if (threadIdx.x == 0) {
        mr[2 * b] = mean;
        mr[2 * b + 1] = rstd;
    }
}

// act'(u) from the PRE-activation u (one exponential; the forward's polynomial expm1 and the output form are not needed here)
__device__ __forceinline__ float act_grad_from_pre(float u, int act) {
    const float e = __builtin_amdgcn_exp2f(fminf(u, 0.f) * 1.4426950408889634f);
    if (act == HNO_ACT_SELU) return u > 0.f ? HNO_SELU_SCALE : (HNO_SELU_SCALE * HNO_SELU_ALPHA) * e;
    if (act == HNO_ACT_ELU) return u > 0.f ? 1.f : e;
    return 1.f;
}

// ------------------------------------------------------------------------------------------------ GroupNorm(1, C) + act
// z = act(gamma1 (y1 - mean1) rstd1 + beta1) [+ act(gamma2 (y2 - mean2) rstd2 + beta2)], all tensors channels-last bf16,
// arithmetic fp32.  8 channels (16 bytes) per thread and iteration; C % 8 == 0.
__global__ __launch_bounds__(256) void cb_gn_apply_kernel(const bf16_t *__restrict__ y1, const float *__restrict__ mr1, const float *__restrict__ g1,
                                                         const float *__restrict__ b1, const bf16_t *__restrict__ y2,
                                                         const float *__restrict__ mr2, const float *__restrict__ g2,
                                                         const float *__restrict__ b2, bf16_t *__restrict__ z, int C, long long per_sample,
                                                         int act, int nstat1, int nstat2, float eps, int B) {
    const int b = blockIdx.y;
    // nstat > 0: the producing convolution left its (sum, sum of squares) partials behind the (B, 2) slot of mr (hno_cb_conv with
    // nstat_out): every workgroup reduces them itself in the finalize kernel's order (all agree bit for bit; one dependent ~5 us
    // launch per layer less), workgroup x = 0 of each sample stores (mean, rstd) for the backward kernels
    __shared__ double fin[512];
    float m1, r1, m2 = 0.f, r2 = 0.f;
    if (nstat1 > 0) {
        cb_gn_finalize_block(mr1 + 2 * B + (size_t)b * nstat1 * 2, nstat1, (double)per_sample, eps, fin, m1, r1);
        if (blockIdx.x == 0 && threadIdx.x == 0) { const_cast<float *>(mr1)[2 * b] = m1; const_cast<float *>(mr1)[2 * b + 1] = r1; }
    } else {
        m1 = mr1[2 * b];
        r1 = mr1[2 * b + 1];
    }
    if (y2) {
        if (nstat2 > 0) {
            cb_gn_finalize_block(mr2 + 2 * B + (size_t)b * nstat2 * 2, nstat2, (double)per_sample, eps, fin, m2, r2);
            if (blockIdx.x == 0 && threadIdx.x == 0) { const_cast<float *>(mr2)[2 * b] = m2; const_cast<float *>(mr2)[2 * b + 1] = r2; }
        } else {
            m2 = mr2[2 * b];
            r2 = mr2[2 * b + 1];
        }
    }
    const long long n8 = per_sample >> 3;
    const int C8 = C >> 3;
    // gamma / beta from LDS (round 6): a thread's channel group changes from item to item, and 16 (32 with two branches) 4-byte global
    // loads per 16-byte item were most of this kernel's memory instructions (24 us for the 49 MB of a shallow-level layer)
    extern __shared__ float gb[];     // [4][C]: gamma1, beta1, gamma2, beta2
    for (int c = threadIdx.x; c < C; c += 256) {
        gb[c] = g1[c];
        gb[C + c] = b1[c];
        if (y2) { gb[2 * C + c] = g2[c]; gb[3 * C + c] = b2[c]; }
    }
    __syncthreads();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const int c0 = (int)((unsigned)i % (unsigned)C8) * 8;      // i < 2^32 (host check): a 64-bit modulo per item costs more than the item
        const size_t e = (size_t)b * per_sample + (size_t)i * 8;
        const uint4 v1 = *reinterpret_cast<const uint4 *>(y1 + e);
        uint4 v2 = make_uint4(0, 0, 0, 0);
        if (y2) v2 = *reinterpret_cast<const uint4 *>(y2 + e);
        const unsigned w1[4] = {v1.x, v1.y, v1.z, v1.w}, w2[4] = {v2.x, v2.y, v2.z, v2.w};
        float G1[8], B1[8], G2[8], B2[8];
        *reinterpret_cast<float4 *>(G1) = *reinterpret_cast<const float4 *>(gb + c0);
        *reinterpret_cast<float4 *>(G1 + 4) = *reinterpret_cast<const float4 *>(gb + c0 + 4);
        *reinterpret_cast<float4 *>(B1) = *reinterpret_cast<const float4 *>(gb + C + c0);
        *reinterpret_cast<float4 *>(B1 + 4) = *reinterpret_cast<const float4 *>(gb + C + c0 + 4);
        if (y2) {
            *reinterpret_cast<float4 *>(G2) = *reinterpret_cast<const float4 *>(gb + 2 * C + c0);
            *reinterpret_cast<float4 *>(G2 + 4) = *reinterpret_cast<const float4 *>(gb + 2 * C + c0 + 4);
            *reinterpret_cast<float4 *>(B2) = *reinterpret_cast<const float4 *>(gb + 3 * C + c0);
            *reinterpret_cast<float4 *>(B2 + 4) = *reinterpret_cast<const float4 *>(gb + 3 * C + c0 + 4);
        }
        unsigned out[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float res[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = 2 * k + u;
                const float a1 = bf2f((bf16_t)(u ? w1[k] >> 16 : w1[k] & 0xffff));
                float t = act_apply(fmaf((a1 - m1) * r1, G1[j], B1[j]), act);
                if (y2) {
                    const float a2 = bf2f((bf16_t)(u ? w2[k] >> 16 : w2[k] & 0xffff));
                    t += act_apply(fmaf((a2 - m2) * r2, G2[j], B2[j]), act);
                }
                res[u] = t;
            }
            out[k] = (unsigned)f2bf(res[0]) | ((unsigned)f2bf(res[1]) << 16);
        }
        *reinterpret_cast<uint4 *>(z + e) = make_uint4(out[0], out[1], out[2], out[3]);
    }
}

// backward, pass 1: with u = gamma xhat + beta, t = dz act'(u):  per (block, channel) partial sums of t and t xhat.
// Each block owns a contiguous run of voxels of ONE sample; thread t of 256 handles channel group (t % C8) of voxel
// (t / C8) + k * (256 / C8) ... the partials go through LDS to one slab row per block: [b][blk][2][C].
// kpart[b][blk] = (sum_c gamma_c S1_c, sum_c gamma_c S2_c) of this workgroup's rows: what pass 2 needs of pass 1 is only their sum over
// the workgroups (k1, k2), so pass 2 adds these up itself and the per-channel sums (pass 1b) move into ONE of its workgroups -- no launch
// between the two streaming passes (round 4b; the finalize launch was 39 x 7.2 us of a V-Net-DS step)
__global__ __launch_bounds__(256) void cb_gn_bwd_reduce_kernel(const bf16_t *__restrict__ dz, const bf16_t *__restrict__ y,
                                                              const float *__restrict__ mr, const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, float *__restrict__ slab, int C, long long V,
                                                              int act, float *__restrict__ kpart) {
    extern __shared__ float lds[];    // [256][24]
    const int b = blockIdx.y, nblk = gridDim.x;
    const float mean = mr[2 * b], rstd = mr[2 * b + 1];
    const int C8 = C >> 3;
    const long long per_blk = (V + nblk - 1) / nblk;
    const long long vlo = (long long)blockIdx.x * per_blk, vhi = vlo + per_blk < V ? vlo + per_blk : V;
    const long long items = (vhi > vlo ? vhi - vlo : 0) * C8;
    // threads 0 .. S-1 with S = the largest multiple of C8 <= 256 stride the items by S, so a thread keeps ONE channel
    // group (it % C8 is constant) and its 16 sums stay in registers until the end
    const int S = 256 - 256 % C8;
    float s1[8], s2[8], s3[8];      // sums of t, t xhat and xhat (the last one gives the convolution's bias gradient, see finalize)
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = s3[j] = 0.f;
    const int cg = (int)(threadIdx.x % C8);
    float gm[8], bt[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { gm[j] = gamma[cg * 8 + j]; bt[j] = beta[cg * 8 + j]; }
    if ((int)threadIdx.x < S)
        for (long long it = threadIdx.x; it < items; it += S) {
            const size_t e = ((size_t)b * V + vlo) * C + (size_t)it * 8;
            const uint4 gv = *reinterpret_cast<const uint4 *>(dz + e), yv = *reinterpret_cast<const uint4 *>(y + e);
            const unsigned gw[4] = {gv.x, gv.y, gv.z, gv.w}, yw[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int j = 2 * k + u;
                    const float g = bf2f((bf16_t)(u ? gw[k] >> 16 : gw[k] & 0xffff));
                    const float xh = (bf2f((bf16_t)(u ? yw[k] >> 16 : yw[k] & 0xffff)) - mean) * rstd;
                    const float t = g * act_grad_from_pre(fmaf(xh, gm[j], bt[j]), act);
                    s1[j] += t;
                    s2[j] = fmaf(t, xh, s2[j]);
                    s3[j] += xh;
                }
        }
    // per-thread sums -> LDS [thread][24]; then channel c = 8 cg + j sums its S / C8 contributing threads in a fixed order
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        lds[threadIdx.x * 24 + j] = s1[j];
        lds[threadIdx.x * 24 + 8 + j] = s2[j];
        lds[threadIdx.x * 24 + 16 + j] = s3[j];
    }
    __syncthreads();
    float *dst = slab + ((size_t)b * nblk + blockIdx.x) * 3 * C;
    for (int i = threadIdx.x; i < 3 * C; i += 256) {
        const int which = i / C, c = i - which * C, g8 = c >> 3, j = c & 7;
        float acc = 0.f;
        for (int t = g8; t < S; t += C8) acc += lds[t * 24 + which * 8 + j];
        dst[i] = acc;
    }
    if (kpart) {
        float q1 = 0.f, q2 = 0.f;
        for (int i = threadIdx.x; i < 2 * C; i += 256) {       // (re-summed from LDS in the same order: bit-identical to dst[i])
            const int which = i / C, c = i - which * C, g8 = c >> 3, j = c & 7;
            float acc = 0.f;
            for (int t = g8; t < S; t += C8) acc += lds[t * 24 + which * 8 + j];
            if (which == 0) q1 = fmaf(gamma[c], acc, q1);
            else q2 = fmaf(gamma[c], acc, q2);
        }
        __syncthreads();
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            q1 += __shfl_xor(q1, o);
            q2 += __shfl_xor(q2, o);
        }
        if ((threadIdx.x & 63) == 0) { lds[(threadIdx.x >> 6) * 2] = q1; lds[(threadIdx.x >> 6) * 2 + 1] = q2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float *kp = kpart + ((size_t)b * nblk + blockIdx.x) * 2;
            kp[0] = (lds[0] + lds[2]) + (lds[4] + lds[6]);
            kp[1] = (lds[1] + lds[3]) + (lds[5] + lds[7]);
        }
    }
}

// pass 1b: slabs -> dgamma[c], dbeta[c] (summed over samples and blocks, fixed order) and gS[b][which][c] = gamma_c S_which[b][c].
// A workgroup step owns a GROUP of CH = 256 / SL channels; SL (a power of two, 32 ... 256) threads per channel each sum every SL-th
// slab row, then a tree over the SL slices.  SL follows the number of slab rows (cb_gn_slices): with 1024 rows (the shallow levels) a
// channel gets all 256 threads -- four rows each -- where the first form (8 channels x 32 slices whatever the row count, the slices then
// added one after the other by 8 threads) walked 32 dependent rows per thread in C / 8 = 3 ... 6 workgroups: 11-12 us, the long pole of
// the apply kernel it is fused into (round 6 trace).  (A single-workgroup version of this step -- 256 dependent loads per thread --
// cost 0.2-0.5 ms per call, 10 ms of the 18 ms V-Net step, in round 2.)
__host__ __device__ __forceinline__ int cb_gn_slices(int nblk) {
    int sl = 32;
    while (sl < 256 && sl < nblk) sl <<= 1;
    return sl;
}
// sh: 3 * 256 floats.  Returns with (t1, t2, t3) = (S1, S2, S3) of sample b, channel c, valid in the threads with slice == 0.
__device__ __forceinline__ void cb_gn_bwd_group_sums(const float *__restrict__ slab, int b, int nblk, int C, int c, int SL, int CH, float *sh, float &t1,
                                                     float &t2, float &t3) {
    const int slice = threadIdx.x / CH;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < C)
        for (int k = slice; k < nblk; k += SL) {
            const float *p = slab + ((size_t)b * nblk + k) * 3 * C;
            s1 += p[c];
            s2 += p[C + c];
            s3 += p[2 * C + c];
        }
    sh[threadIdx.x] = s1;
    sh[256 + threadIdx.x] = s2;
    sh[512 + threadIdx.x] = s3;
    __syncthreads();
    for (int o = SL >> 1; o > 0; o >>= 1) {
        if (slice < o) {
            sh[threadIdx.x] += sh[threadIdx.x + o * CH];
            sh[256 + threadIdx.x] += sh[256 + threadIdx.x + o * CH];
            sh[512 + threadIdx.x] += sh[512 + threadIdx.x + o * CH];
        }
        __syncthreads();
    }
    t1 = sh[threadIdx.x];
    t2 = sh[256 + threadIdx.x];
    t3 = sh[512 + threadIdx.x];
    __syncthreads();          // sh is rewritten by the next sample
}
__device__ __forceinline__ void cb_gn_bwd_finalize_group(const float *__restrict__ slab, const float *__restrict__ gamma, int B, int nblk,
                                                         int C, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                         float *__restrict__ gS, int accumulate, int group, float *sh) {
    const int SL = cb_gn_slices(nblk), CH = 256 / SL;
    const int j = threadIdx.x % CH, slice = threadIdx.x / CH;
    const int c = group * CH + j;
    float dg = 0.f, db = 0.f;
    for (int b = 0; b < B; ++b) {
        float t1, t2, t3;
        cb_gn_bwd_group_sums(slab, b, nblk, C, c, SL, CH, sh, t1, t2, t3);
        if (slice == 0 && c < C) {
            db += t1;
            dg += t2;
            gS[((size_t)b * 4) * C + c] = gamma[c] * t1;
            gS[((size_t)b * 4 + 1) * C + c] = gamma[c] * t2;
            gS[((size_t)b * 4 + 2) * C + c] = t3;          // sum_v xhat[c][v]
        }
    }
    if (slice == 0 && c < C) {
        if (accumulate) { dgamma[c] += dg; dbeta[c] += db; }
        else { dgamma[c] = dg; dbeta[c] = db; }
    }
}

__global__ __launch_bounds__(256) void cb_gn_bwd_finalize_kernel(const float *__restrict__ slab, const float *__restrict__ gamma, int B, int nblk,
                                                                int C, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                                float *__restrict__ gS, int accumulate) {
    __shared__ float sh[768];
    cb_gn_bwd_finalize_group(slab, gamma, B, nblk, C, dgamma, dbeta, gS, accumulate, blockIdx.x, sh);
}

// pass 2: dy = rstd (gamma t - k1 - xhat k2), bf16 channels-last
__global__ __launch_bounds__(256) void cb_gn_bwd_apply_kernel(const bf16_t *__restrict__ dz, const bf16_t *__restrict__ y, const float *__restrict__ mr,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta,
                                                             const float *__restrict__ gS, bf16_t *__restrict__ dy, int C, long long per_sample,
                                                             int act, float *__restrict__ dcolsum, int B, const float *__restrict__ kpart,
                                                             int nblk, const float *__restrict__ slab, float *__restrict__ dgamma,
                                                             float *__restrict__ dbeta, float *__restrict__ gS_w, int accumulate) {
    const int b = blockIdx.y;
    // pass 1b (kpart form): workgroup g of sample 0 owns channel group g (cb_gn_slices: 1 ... 8 channels): their sums over samples and
    // slab rows -> dgamma, dbeta, gS and -- per channel independent -- the convolution's bias gradient.  (All groups in ONE workgroup made that
    // workgroup the kernel's long pole on the deep levels: 48 groups x two barriers and a global round trip each = 0.7 ms per step.)
    {
        const int SL = cb_gn_slices(nblk), CH = 256 / SL, ngroups = (C + CH - 1) / CH;
        if (kpart && blockIdx.y == 0 && (int)blockIdx.x < ngroups) {
            __shared__ float sh[768];
            __shared__ float kq[2][256];
            const float inv_n0 = 1.0f / (float)per_sample, Vf0 = (float)(per_sample / C);
            const int j = threadIdx.x % CH, slice = threadIdx.x / CH;
            for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
                const int c = grp * CH + j;
                float dg = 0.f, db = 0.f, colsum = 0.f;
                for (int bb = 0; bb < B; ++bb) {
                    float t1, t2, t3;
                    cb_gn_bwd_group_sums(slab, bb, nblk, C, c, SL, CH, sh, t1, t2, t3);
                    float q1 = 0.f, q2 = 0.f;
                    if (dcolsum) {      // k1, k2 of sample bb: all threads add the kpart rows, fixed order
                        float a1 = 0.f, a2 = 0.f;
                        for (int k = threadIdx.x; k < nblk; k += 256) {
                            a1 += kpart[((size_t)bb * nblk + k) * 2];
                            a2 += kpart[((size_t)bb * nblk + k) * 2 + 1];
                        }
                        kq[0][threadIdx.x] = a1;
                        kq[1][threadIdx.x] = a2;
                        __syncthreads();
                        for (int o = 128; o > 0; o >>= 1) {
                            if ((int)threadIdx.x < o) {
                                kq[0][threadIdx.x] += kq[0][threadIdx.x + o];
                                kq[1][threadIdx.x] += kq[1][threadIdx.x + o];
                            }
                            __syncthreads();
                        }
                        q1 = kq[0][0];
                        q2 = kq[1][0];
                        __syncthreads();
                    }
                    if (slice == 0 && c < C) {
                        db += t1;
                        dg += t2;
                        const float g1 = gamma[c] * t1;
                        gS_w[((size_t)bb * 4) * C + c] = g1;
                        gS_w[((size_t)bb * 4 + 1) * C + c] = gamma[c] * t2;
                        gS_w[((size_t)bb * 4 + 2) * C + c] = t3;
                        colsum += mr[2 * bb + 1] * (g1 - Vf0 * (q1 * inv_n0) - (q2 * inv_n0) * t3);
                    }
                }
                if (slice == 0 && c < C) {
                    if (accumulate) { dgamma[c] += dg; dbeta[c] += db; }
                    else { dgamma[c] = dg; dbeta[c] = db; }
                    if (dcolsum) dcolsum[c] = colsum;
                }
            }
            __syncthreads();
        }
    }
    // k1 = sum_c gS[b][0][c] / N, k2 = sum_c gS[b][1][c] / N: every workgroup re-reduces the <= 2 x 2048 values itself (fixed
    // order, so all workgroups agree bit for bit) instead of waiting for one more tiny launch
    __shared__ float kred[2][256];
    {
        float a1 = 0.f, a2 = 0.f;
        if (kpart) {
            for (int k = threadIdx.x; k < nblk; k += 256) {
                a1 += kpart[((size_t)b * nblk + k) * 2];
                a2 += kpart[((size_t)b * nblk + k) * 2 + 1];
            }
        } else {
            for (int c = threadIdx.x; c < C; c += 256) {
                a1 += gS[((size_t)b * 4) * C + c];
                a2 += gS[((size_t)b * 4 + 1) * C + c];
            }
        }
        kred[0][threadIdx.x] = a1;
        kred[1][threadIdx.x] = a2;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) {
                kred[0][threadIdx.x] += kred[0][threadIdx.x + o];
                kred[1][threadIdx.x] += kred[1][threadIdx.x + o];
            }
            __syncthreads();
        }
    }
    const float inv_n = 1.0f / (float)per_sample;
    const float mean = mr[2 * b], rstd = mr[2 * b + 1], k1 = kred[0][0] * inv_n, k2 = kred[1][0] * inv_n;
    // Column sums of dy = the bias gradient of the convolution that produced y, without another pass over dy:
    //   sum_v dy[c][v] = rstd (gamma_c S1_c - V k1 - k2 S3_c),  S1 = sum_v t, S3 = sum_v xhat  (both already reduced per channel).
    // Workgroup (0, 0) does it for every sample (k1, k2 of the other samples re-reduced the same way), fixed order.
    if (dcolsum && !kpart && blockIdx.x == 0 && blockIdx.y == 0) {
        const float Vf = (float)(per_sample / C);
        for (int c = threadIdx.x; c < C; c += 256) dcolsum[c] = 0.f;
        for (int bb = 0; bb < B; ++bb) {
            float q1 = k1, q2 = k2;
            if (bb != b) {                       // uniform branch (b == 0 here)
                __syncthreads();
                float a1 = 0.f, a2 = 0.f;
                for (int c = threadIdx.x; c < C; c += 256) {
                    a1 += gS[((size_t)bb * 4) * C + c];
                    a2 += gS[((size_t)bb * 4 + 1) * C + c];
                }
                kred[0][threadIdx.x] = a1;
                kred[1][threadIdx.x] = a2;
                __syncthreads();
                for (int o = 128; o > 0; o >>= 1) {
                    if ((int)threadIdx.x < o) {
                        kred[0][threadIdx.x] += kred[0][threadIdx.x + o];
                        kred[1][threadIdx.x] += kred[1][threadIdx.x + o];
                    }
                    __syncthreads();
                }
                q1 = kred[0][0] * inv_n;
                q2 = kred[1][0] * inv_n;
            }
            const float rs = mr[2 * bb + 1];
            for (int c = threadIdx.x; c < C; c += 256)
                dcolsum[c] += rs * (gS[((size_t)bb * 4) * C + c] - Vf * q1 - q2 * gS[((size_t)bb * 4 + 2) * C + c]);
        }
    }
    const long long n8 = per_sample >> 3;
    const int C8 = C >> 3;
    extern __shared__ float gb[];     // [2][C]: gamma, beta (see cb_gn_apply_kernel)
    for (int c = threadIdx.x; c < C; c += 256) {
        gb[c] = gamma[c];
        gb[C + c] = beta[c];
    }
    __syncthreads();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const int c0 = (int)((unsigned)i % (unsigned)C8) * 8;      // i < 2^32 (host check): a 64-bit modulo per item costs more than the item
        const size_t e = (size_t)b * per_sample + (size_t)i * 8;
        const uint4 gv = *reinterpret_cast<const uint4 *>(dz + e), yv = *reinterpret_cast<const uint4 *>(y + e);
        const unsigned gw[4] = {gv.x, gv.y, gv.z, gv.w}, yw[4] = {yv.x, yv.y, yv.z, yv.w};
        float G[8], Bt[8];
        *reinterpret_cast<float4 *>(G) = *reinterpret_cast<const float4 *>(gb + c0);
        *reinterpret_cast<float4 *>(G + 4) = *reinterpret_cast<const float4 *>(gb + c0 + 4);
        *reinterpret_cast<float4 *>(Bt) = *reinterpret_cast<const float4 *>(gb + C + c0);
        *reinterpret_cast<float4 *>(Bt + 4) = *reinterpret_cast<const float4 *>(gb + C + c0 + 4);
        unsigned out[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float res[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = 2 * k + u;
                const float g = bf2f((bf16_t)(u ? gw[k] >> 16 : gw[k] & 0xffff));
                const float xh = (bf2f((bf16_t)(u ? yw[k] >> 16 : yw[k] & 0xffff)) - mean) * rstd;
                const float t = g * act_grad_from_pre(fmaf(xh, G[j], Bt[j]), act);
                res[u] = rstd * (G[j] * t - k1 - xh * k2);
            }
            out[k] = (unsigned)f2bf(res[0]) | ((unsigned)f2bf(res[1]) << 16);
        }
        *reinterpret_cast<uint4 *>(dy + e) = make_uint4(out[0], out[1], out[2], out[3]);
    }
}

// ------------------------------------------------------------------------------------------------ layout / precision
// fp32 NCDHW (C channels) -> bf16 NDHWC with CP >= C channels (pad channels zero).  One thread = one voxel.
__global__ __launch_bounds__(256) void cb_pack_input_kernel(const float *__restrict__ x, bf16_t *__restrict__ y, int C, int CP, long long V) {
    const int b = blockIdx.y;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long long)gridDim.x * 256) {
        for (int c0 = 0; c0 < CP; c0 += 8) {
            unsigned out[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + 2 * k;
                const float f0 = c < C ? x[((size_t)b * C + c) * V + v] : 0.f;
                const float f1 = c + 1 < C ? x[((size_t)b * C + c + 1) * V + v] : 0.f;
                out[k] = (unsigned)f2bf(f0) | ((unsigned)f2bf(f1) << 16);
            }
            *reinterpret_cast<uint4 *>(y + ((size_t)b * V + v) * CP + c0) = make_uint4(out[0], out[1], out[2], out[3]);
        }
    }
}

// bf16 NDHWC (CP channels, the first C used) -> fp32 NCDHW
__global__ __launch_bounds__(256) void cb_unpack_kernel(const bf16_t *__restrict__ x, float *__restrict__ y, int C, int CP, long long V) {
    const int b = blockIdx.y;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long long)gridDim.x * 256) {
        for (int c0 = 0; c0 < C; c0 += 8) {
            const uint4 q = *reinterpret_cast<const uint4 *>(x + ((size_t)b * V + v) * CP + c0);
            const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + 2 * k;
                if (c < C) y[((size_t)b * C + c) * V + v] = bf2f((bf16_t)(w[k] & 0xffff));
                if (c + 1 < C) y[((size_t)b * C + c + 1) * V + v] = bf2f((bf16_t)(w[k] >> 16));
            }
        }
    }
}

// per-channel sum over samples and voxels of a channels-last bf16 tensor (bias gradients): slab per block + ordered reduce
__global__ __launch_bounds__(256) void cb_colsum_kernel(const bf16_t *__restrict__ g, float *__restrict__ slab, int C, long long rows) {
    extern __shared__ float lds[];   // [256][8]
    const int C8 = C >> 3;
    const long long per_blk = (rows + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per_blk, hi = lo + per_blk < rows ? lo + per_blk : rows;
    const long long items = (hi > lo ? hi - lo : 0) * C8;
    const int S = 256 - 256 % C8;       // see cb_gn_bwd_reduce_kernel
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    const int cg = (int)(threadIdx.x % C8);
    if ((int)threadIdx.x < S) {
        for (long long it = threadIdx.x; it < items; it += S) {
            const uint4 gv = *reinterpret_cast<const uint4 *>(g + (size_t)lo * C + (size_t)it * 8);
            const unsigned gw[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s[2 * k] += bf2f((bf16_t)(gw[k] & 0xffff));
                s[2 * k + 1] += bf2f((bf16_t)(gw[k] >> 16));
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) lds[threadIdx.x * 8 + j] = s[j];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc = 0.f;
        for (int t = c >> 3; t < S; t += C8) acc += lds[t * 8 + (c & 7)];
        slab[(size_t)blockIdx.x * C + c] = acc;
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[t][o][i] = sum_{b, v} G[b][v][o] * X[b][in(v, t)][i].  The reduction index is the voxel, which channels-last tensors
// do not have contiguous: the tiles are staged in LDS as they lie in memory ([voxel][channel] rows) and the MFMA
// operands are read with ds_read_b64_tr_b16 (per 16-lane group: 4 voxel rows x 16 channels, delivered channel-major).
//
// Work decomposition.  A block owns (b, od, a band of TH output rows) per iteration and a (CO x CI) channel block
// (CO, CI <= 48).  The G band is staged on a padded row pitch S (columns Wo .. S-1 are zero), the X halo -- the rows
// and planes the band's taps reach -- on the same pitch, so that for output position p = hh * Sg + ww the tap
// (td, th, tw) reads X-image element  base(tap) + stride * p : one flat shift per tap, no per-element index
// arithmetic, and padding positions multiply zeros of G.  Each of the 4 waves accumulates ~ntaps / 4 taps in registers
// across all the block's bands; the block then writes one slab [tap][CO][CI] and an ordered reduce sums the slabs.
struct CwArgs {
    const bf16_t *g, *xa, *xb;   // g: (B, Vo, Cg) output gradient; x = [xa | xb] (B, Vi, Ca / Cb) layer input
    int Ca, Cb, Cg;
    float *slab;                 // [nblk][ntaps][CO][CI] per (co block, ci block): see host
    int B, Di, Hi, Wi, Do, Ho, Wo;
    int ks, stride, pad, ntaps;
    int TH, Sg, Sx, xrows, xplanes;   // band height; pitches; staged X rows per plane = stride * (TH - 1) + ks, planes = ks
    int xpos;                         // positions of the X image incl. slack (every one of them is written by the staging pass)
    int co0, ci0, CO, CI;        // channel block (multiples of 8; CO, CI <= 48); with gridDim.y > 1: block y = (y / nci) , (y % nci)
    int nci;                     // ci blocks per co block when the channel blocks ride on gridDim.y (all of size CO x CI)
    int nbands;                  // bands per (b, od) = ceil(Ho / TH)
    int swap;                    // 1: transposed-conv weight gradient (roles of g and x swapped by the host; informational)
    int dbg;                     // ablation switches: 512 skip the staging, 1024 skip the K loop (results WRONG)
    int slide, nseg;             // slide = 1: items are runs of output planes, the input planes slide through a 3-slot ring
};

// 16 bytes of zeros in global memory: the source of every LDS-DMA piece that lies outside its tensor
__device__ __attribute__((aligned(16))) unsigned int g_cb_zero16[4] = {0u, 0u, 0u, 0u};

// ds_read_b64_tr_b16 through the compiler builtin (it then places the lgkmcnt waits itself).  Per 16-lane group: lane 4q + p
// supplies the address of row q, columns 4p .. 4p+3 of a 4 x 16 block; lane i receives column i of the 4 rows.
__device__ __forceinline__ s16x4 lds_tr_read(const bf16_t *p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p));
}

// TA = ceil(CO / 16), TB = ceil(CI / 16) 16-channel tiles; TPW = taps per wave (ceil(ntaps / 4))
template <int TA, int TB, int TPW>
__global__ __launch_bounds__(256) void cb_wgrad_kernel(CwArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int PG = TA * 16, PX = TB * 16;                 // channel pitch (elements) of the G / X images; pad channels zero
    const int gpos = a.TH * a.Sg;                          // G image positions (rounded up to 32 below)
    const int gpos32 = (gpos + 31) & ~31;
    const int xpos = a.xpos;                               // X image positions incl. slack on both ends (host: wg_plan)
    bf16_t *gi = reinterpret_cast<bf16_t *>(smem);                         // 2 x [gpos32][PG] (double buffer)
    bf16_t *xi = gi + (size_t)2 * gpos32 * PG;                             // [xpos][PX], position 0 = slack
    const int xorg = a.Sx + 32;                                            // image position of (plane 0, row 0, col -pad.. see below)
    f32x4 acc[TPW][TA][TB];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
            for (int j = 0; j < TB; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int co0 = a.co0 + (gridDim.y > 1 ? (int)(blockIdx.y / a.nci) * a.CO : 0);
    const int ci0 = a.ci0 + (gridDim.y > 1 ? (int)(blockIdx.y % a.nci) * a.CI : 0);
    const int grp = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
    // the whole X image once: the slack positions around the plane slots are read (against zeros of G) and never staged
    for (int i = threadIdx.x; i < xpos * (PX >> 3); i += 256) reinterpret_cast<uint4 *>(xi)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    // ---- staging by LDS-DMA (round 4).  Round 3 staged through registers between two barriers: stage, sync, multiply, sync -- the
    // two phases ADD (48 -> 24 layer at 81 x 97 x 65: staging alone 98 us, K loop alone 105 us, both 184 us; tools/bench_cb_conv.py).
    // Now the images of step n + 1 are requested (global_load_lds_dwordx4: 64 lanes x 16 bytes -> 1 KiB of contiguous LDS, no
    // registers) before the K loop of step n runs, into the other G buffer and the free plane slot(s).  An image is a flat array of
    // 16-byte pieces [position][channel block / 8]; piece f of a plane slot is (row f / (Sx LPP), column, channel piece); pieces outside
    // the tensor (spatial padding, pad channels) come from a 16-byte block of zeros in global memory (a DMA cannot mask).
    const unsigned gi_b = (unsigned)(size_t)gi, xi_b = (unsigned)(size_t)xi;    // LDS byte addresses
    const bf16_t *zsrc = reinterpret_cast<const bf16_t *>(g_cb_zero16);
    auto dma16 = [&](const bf16_t *src, unsigned dst_wave) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(dst_wave) : "memory");
    };
    // one wave per image row (row validity and the row's base address are wave-uniform: scalar arithmetic), a lane per 16-byte piece:
    // per DMA instruction a lane spends ~10 VALU operations on its address (the first version decoded a flat piece index with two
    // runtime divisions per piece: the address arithmetic of a step cost as much as its K loop)
    auto stage_plane = [&](int b, int id, int ih0, int slot) {
        constexpr int LPP = TB * 2;
        const int c8n = (a.CI + 7) >> 3;
        const bool pok = id >= 0 && id < a.Di;
        const int npr = a.Sx * LPP;                                   // pieces per image row
        const unsigned dst0 = xi_b + (unsigned)(xorg + slot * a.xrows * a.Sx) * (PX * 2);
        for (int rr = wave; rr < a.xrows; rr += 4) {
            const int ih = ih0 + rr;
            const bool rok = pok && ih >= 0 && ih < a.Hi;
            const size_t vrow = (((size_t)b * a.Di + (rok ? id : 0)) * a.Hi + (rok ? ih : 0)) * a.Wi;
            const bf16_t *ra = a.xa + vrow * a.Ca + ci0, *rb = a.xb ? a.xb + vrow * a.Cb + (ci0 - a.Ca) : a.xa;
            const unsigned drow = dst0 + (unsigned)(rr * a.Sx) * (PX * 2);
            for (int f0 = 0; f0 < npr; f0 += 64) {
                const int f = f0 + lane;
                const int cc = f / LPP, piece = f - cc * LPP;
                const int iw = cc - a.pad, ch = ci0 + piece * 8;
                const bool ok = rok && iw >= 0 && iw < a.Wi && piece < c8n;
                const bf16_t *src = !ok ? zsrc : (ch < a.Ca ? ra + (size_t)iw * a.Ca + piece * 8 : rb + (size_t)iw * a.Cb + piece * 8);
                if (f < npr) dma16(src, __builtin_amdgcn_readfirstlane(drow + (unsigned)f0 * 16u));
            }
        }
    };
    auto stage_g = [&](int b, int od, int oh0, int buf) {
        constexpr int LPG = TA * 2;
        const int c8n = (a.CO + 7) >> 3;
        const int npr = a.Sg * LPG;
        const int rows_img = (gpos32 + a.Sg - 1) / a.Sg;              // image rows incl. the (partial) rounding row: all zero beyond TH
        const unsigned dst0 = gi_b + (unsigned)buf * (unsigned)(gpos32 * PG * 2);
        for (int hh = wave; hh < rows_img; hh += 4) {
            const int oh = oh0 + hh;
            const bool rok = hh < a.TH && oh < a.Ho;
            const size_t vrow = (((size_t)b * a.Do + od) * a.Ho + (rok ? oh : 0)) * a.Wo;
            const bf16_t *rg = a.g + vrow * a.Cg + co0;
            const unsigned drow = dst0 + (unsigned)(hh * a.Sg) * (PG * 2);
            const int nrow = (gpos32 - hh * a.Sg) * LPG < npr ? (gpos32 - hh * a.Sg) * LPG : npr;     // the last row stops at gpos32
            for (int f0 = 0; f0 < nrow; f0 += 64) {
                const int f = f0 + lane;
                const int ww = f / LPG, piece = f - ww * LPG;
                const bool ok = rok && ww < a.Wo && piece < c8n;
                const bf16_t *src = ok ? rg + (size_t)ww * a.Cg + piece * 8 : zsrc;
                if (f < nrow) dma16(src, __builtin_amdgcn_readfirstlane(drow + (unsigned)f0 * 16u));
            }
        }
    };
    // work items.  slide = 1 (3x3x3, stride 1): an item is (b, band, a run of consecutive output planes); the input planes of a run
    // live in a ring of FOUR plane slots (plane id in slot (id + 1) & 3) and every step requests ONE new plane -- the plane of the
    // NEXT step.  slide = 0: an item is one (b, od, band) with its ks input planes in one of two slot sets.
    const long long nwork = a.slide ? (long long)a.B * a.nbands * a.nseg : (long long)a.B * a.Do * a.nbands;
    // a step = (work item wk, output plane od); item -> (b, band, od_lo, od_hi).  Plain integers: a struct with a bool went through scratch
    int c_b = 0, c_band = 0, c_od = 0, c_lo = 0, c_hi = 0;
    long long c_wk = blockIdx.x;
    auto decode = [&](long long wk, int &sb, int &sband, int &slo, int &shi) -> bool {
        if (wk >= nwork) return false;
        if (a.slide) {
            const int seg = (int)(wk % a.nseg);
            sband = (int)((wk / a.nseg) % a.nbands);
            sb = (int)(wk / ((long long)a.nseg * a.nbands));
            slo = (int)((long long)a.Do * seg / a.nseg);
            shi = (int)((long long)a.Do * (seg + 1) / a.nseg);
        } else {
            sband = (int)(wk % a.nbands);
            slo = (int)((wk / a.nbands) % a.Do);
            shi = slo + 1;
            sb = (int)(wk / ((long long)a.nbands * a.Do));
        }
        return true;        // (nseg <= Do on the host: no empty runs)
    };
    // request the images of a step (G into buffer `par`; X: the planes the step needs that are not in the ring yet)
    auto request = [&](int sb, int sband, int sod, int slo, int par) {
        const int oh0 = sband * a.TH, ih0 = a.stride * oh0 - a.pad;
        stage_g(sb, sod, oh0, par);
        if (a.slide) {
            if (sod == slo) { stage_plane(sb, sod - 1, ih0, sod & 3); stage_plane(sb, sod, ih0, (sod + 1) & 3); }
            stage_plane(sb, sod + 1, ih0, (sod + 2) & 3);
        } else {
            for (int pl = 0; pl < a.xplanes; ++pl) stage_plane(sb, a.stride * sod - a.pad + pl, ih0, par * a.xplanes + pl);
        }
    };
    bool c_valid = decode(c_wk, c_b, c_band, c_lo, c_hi);
    c_od = c_lo;
    int par = 0;
    if (c_valid && !(a.dbg & 512)) request(c_b, c_band, c_od, c_lo, par);
    while (c_valid) {
        int n_b = c_b, n_band = c_band, n_od = c_od + 1, n_lo = c_lo, n_hi = c_hi;
        long long n_wk = c_wk;
        bool n_valid = true;
        if (n_od >= n_hi) {
            n_wk = c_wk + gridDim.x;
            n_valid = decode(n_wk, n_b, n_band, n_lo, n_hi);
            n_od = n_lo;
        }
        // a new run needs three fresh planes: they do not fit beside the three the current step reads -> requested after its K loop
        const bool early = n_valid && !(a.slide && n_od == n_lo);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's requests for `cur` have landed ...
        __syncthreads();                                       // ... everybody's have, and everybody is done with the previous step
        if (early && !(a.dbg & 512)) request(n_b, n_band, n_od, n_lo, par ^ 1);
        const bf16_t *gcur = gi + (size_t)par * gpos32 * PG;
        const int od = c_od;
        {
        const int slot_rot = 0;
        (void)slot_rot;
        if (!(a.dbg & 1024)) {
        // ---- K loop over the band's positions, 32 per step (v_mfma_f32_16x16x32_bf16: lane group grp holds k = 8 grp .. 8 grp + 7).
        // Round 4: software-pipelined and branch-free.  The first version read a tap's six B fragments, waited for them and multiplied,
        // tap after tap, behind a wave-uniform `tap < ntaps` branch per tap (ISA: s_cbranch_execz + s_waitcnt lgkmcnt(0) in front of
        // every group of 9 MFMAs: ~100 cycles of exposed LDS latency per 144 MFMA cycles with one wave per SIMD -- the weight gradient
        // ran at half the forward rate).  Now the fragments of step (p0, t + 1) are requested BEFORE the MFMAs of step (p0, t); a wave
        // whose last tap does not exist (27 taps over 4 waves) multiplies a clamped tap and drops the result at the slab store.
        int boff[TPW];        // wave-uniform LDS element offset of tap t's X window (position 0 of the band)
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            int tap = wave + 4 * t;
            tap = tap < a.ntaps ? tap : a.ntaps - 1;
            const int ks2 = a.ks * a.ks;
            const int td = tap / ks2, th = (tap / a.ks) % a.ks, tw = tap % a.ks;
            const int slot = a.slide ? ((od + td) & 3) : par * a.xplanes + td;      // input plane od - 1 + td sits in slot (od + td) & 3
            boff[t] = (xorg + (slot * a.xrows + th) * a.Sx + tw) * PX;
        }
        const int a_lane = (8 * grp + lq) * PG + 4 * lp;                 // + (p0 + 4 u) PG + 16 i
        const int b_lane = a.stride * (8 * grp + lq) * PX + 4 * lp;      // + boff[t] + stride (p0 + 4 u) PX + 16 j
        auto load_a = [&](int p0, s16x4 (&af)[TA][2]) {
#pragma unroll
            for (int i = 0; i < TA; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u) af[i][u] = lds_tr_read(gcur + (a_lane + (p0 + 4 * u) * PG + 16 * i));
        };
        auto load_b = [&](int p0, int off, s16x4 (&bf)[TB][2]) {
#pragma unroll
            for (int j = 0; j < TB; ++j)
#pragma unroll
                for (int u = 0; u < 2; ++u) bf[j][u] = lds_tr_read(xi + (b_lane + off + a.stride * (p0 + 4 * u) * PX + 16 * j));
        };
        s16x4 af[TA][2], bcur[TB][2];
        load_a(0, af);
        load_b(0, boff[0], bcur);
        for (int p0 = 0; p0 < gpos32; p0 += 32) {
            const int pn = p0 + 32 < gpos32 ? p0 + 32 : p0;      // (the last step re-requests its own fragments: harmless)
            s16x4 an[TA][2];
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                s16x4 bn[TB][2];
                if (t + 1 < TPW) {
                    load_b(p0, boff[t + 1], bn);
                } else {
                    load_a(pn, an);
                    load_b(pn, boff[0], bn);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TA; ++i)
#pragma unroll
                    for (int j = 0; j < TB; ++j) {
                        typedef short s16x8 __attribute__((ext_vector_type(8)));
                        const s16x8 av = {af[i][0][0], af[i][0][1], af[i][0][2], af[i][0][3], af[i][1][0], af[i][1][1], af[i][1][2], af[i][1][3]};
                        const s16x8 bv = {bcur[j][0][0], bcur[j][0][1], bcur[j][0][2], bcur[j][0][3], bcur[j][1][0], bcur[j][1][1], bcur[j][1][2], bcur[j][1][3]};
                        acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bv), __builtin_bit_cast(bf16x8, av),
                                                                               acc[t][i][j], 0, 0, 0);      // X rows, G columns: see the slab store
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < TB; ++j)
#pragma unroll
                    for (int u = 0; u < 2; ++u) bcur[j][u] = bn[j][u];
            }
#pragma unroll
            for (int i = 0; i < TA; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u) af[i][u] = an[i][u];
        }
        }
        }
        if (n_valid && !early) {
            __syncthreads();                                   // everybody has left the K loop: the ring may be refilled
            if (!(a.dbg & 512)) request(n_b, n_band, n_od, n_lo, par ^ 1);
        }
        c_valid = n_valid; c_wk = n_wk; c_b = n_b; c_band = n_band; c_od = n_od; c_lo = n_lo; c_hi = n_hi;
        par ^= 1;
    }
    // ---- slab: [tap][CO][CI] for this block.  C/D of 16x16x32 (X = A operand, G = B operand): column (lane & 15) = output channel, rows
    // 4 (lane >> 4) + e = FOUR CONSECUTIVE input channels: one 16-byte store per tile and lane (the other orientation wrote the 250 KB
    // slab of a workgroup with 4-byte stores, four instructions per tile)
    float *dst = a.slab + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * a.ntaps * a.CO * a.CI;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tap = wave + 4 * t;
        if (tap >= a.ntaps) continue;
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                const int o = 16 * i + li, c = 16 * j + 4 * grp;
                if (o < a.CO && c < a.CI)       // (CI is a multiple of 8 and c of 4: the four channels are all in or all out)
                    *reinterpret_cast<float4 *>(dst + ((size_t)tap * a.CO + o) * a.CI + c) = make_float4(acc[t][i][j][0], acc[t][i][j][1], acc[t][i][j][2], acc[t][i][j][3]);
            }
    }
}

// slabs [nblk][ntaps][CO][CI] -> dW fp32 in the parameter's own layout: element (o, i, tap) of the GEMM goes to
// dW[(c0 * C1 + c1) * T + tap] with (c0, c1) = (o, i) or (i, o).  A workgroup = 64 consecutive elements x 4 slab slices
// (slice q sums slabs q, q + 4, ...; four independent loads in flight per thread), the 4 slices are added in order.
// (One thread walking all 256 slabs of its element -- 256 dependent-latency loads -- made this 3.5 ms of the V-Net step.)
__global__ __launch_bounds__(256) void cb_wgrad_reduce_kernel(const float *__restrict__ slab, int nblk, int ntaps, int CO, int CI, int co0, int ci0,
                                                             float *__restrict__ dW, int C1, int T, int out_is_axis0, int nci) {
    __shared__ float sh[4][64];
    const int n = ntaps * CO * CI;
    slab += (size_t)blockIdx.y * nblk * n;
    if (gridDim.y > 1) {
        co0 += (int)(blockIdx.y / nci) * CO;
        ci0 += (int)(blockIdx.y % nci) * CI;
    }
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    for (int base = blockIdx.x * 64; base < n; base += gridDim.x * 64) {
        const int idx = base + e;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (idx < n) {
            int k = q;
            for (; k + 12 < nblk; k += 16) {
                s0 += slab[(size_t)k * n + idx];
                s1 += slab[(size_t)(k + 4) * n + idx];
                s2 += slab[(size_t)(k + 8) * n + idx];
                s3 += slab[(size_t)(k + 12) * n + idx];
            }
            for (; k < nblk; k += 4) s0 += slab[(size_t)k * n + idx];
        }
        sh[q][e] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (q == 0 && idx < n) {
            const float s = (sh[0][e] + sh[1][e]) + (sh[2][e] + sh[3][e]);
            const int c = idx % CI, o = (idx / CI) % CO, tap = idx / (CI * CO);
            const int go = co0 + o, gc = ci0 + c;
            const int c0 = out_is_axis0 ? go : gc, c1 = out_is_axis0 ? gc : go;
            dW[((size_t)c0 * C1 + c1) * T + tap] = s;
        }
        __syncthreads();
    }
}

// The same reduction for layers with MANY channel blocks and FEW slabs per block (the deep levels: 384 x 384 = 64 blocks x 4 slabs).
// There the kernel above is bound by its stores: consecutive threads hold consecutive input channels of one tap, which lie T floats
// apart in the parameter -- 4 M scattered 4-byte stores for a 16 MB gradient (27-45 us per launch in the cfg4 step, round 6).  Here a
// workgroup owns one output channel of one channel block: its T x CI sums are formed with coalesced reads (CI consecutive floats per
// tap and slab), transposed through LDS and written as ONE contiguous run of CI * T floats of the parameter.
__global__ __launch_bounds__(256) void cb_wgrad_reduce_rows_kernel(const float *__restrict__ slab, int nblk, int ntaps, int CO, int CI, int co0, int ci0,
                                                                  float *__restrict__ dW, int C1, int T, int nci) {
    __shared__ float sh[27 * 48];
    const int n = ntaps * CO * CI;
    const int o = blockIdx.x;
    slab += (size_t)blockIdx.y * nblk * n;
    if (gridDim.y > 1) {
        co0 += (int)(blockIdx.y / nci) * CO;
        ci0 += (int)(blockIdx.y % nci) * CI;
    }
    const int ne = ntaps * CI;
    for (int e = threadIdx.x; e < ne; e += 256) {
        const int tap = e / CI, c = e - tap * CI;
        const float *src = slab + ((size_t)tap * CO + o) * CI + c;
        float s0 = 0.f, s1 = 0.f;
        int k = 0;
        for (; k + 1 < nblk; k += 2) {
            s0 += src[(size_t)k * n];
            s1 += src[(size_t)(k + 1) * n];
        }
        if (k < nblk) s0 += src[(size_t)k * n];
        sh[c * ntaps + tap] = s0 + s1;
    }
    __syncthreads();
    float *dst = dW + ((size_t)(co0 + o) * C1 + ci0) * T;
    for (int j = threadIdx.x; j < ne; j += 256) dst[j] = sh[j];
}

// per-channel slabs [nblk][C] -> out[C]: one workgroup per 8 channels, 32 slab slices per channel added in order
__global__ __launch_bounds__(256) void cb_colsum_reduce_kernel(const float *__restrict__ slab, int nblk, int C, float *__restrict__ out) {
    __shared__ float sh[32][8];
    const int j = threadIdx.x & 7, slice = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + j;
    float s = 0.f;
    for (int k = slice; k < nblk; k += 32) s += slab[(size_t)k * C + c];
    sh[slice][j] = s;
    __syncthreads();
    if (slice == 0) {
        float t = 0.f;
        for (int q = 0; q < 32; ++q) t += sh[q][j];
        out[c] = t;
    }
}

static int gsz(long long n, int per = 256, int cap = 4096) {
    long long g = (n + per - 1) / per;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace hno

using namespace hno;

// ================================================================================================ C ABI
extern "C" size_t hno_cb_packed_weight_bytes(int Cin, int Cout, int ks) {
    const int nq = ks * ks * ks * (Cin / 8), nq2 = (nq + 1) & ~1;
    const int CoP = (Cout + 31) / 32 * 32;
    return (size_t)nq2 * CoP * 8 * sizeof(bf16_t);
}

// role: 0 conv forward (W[Cout][Cin][T]), 1 conv input gradient (same tensor, channel roles swapped),
//       2 ConvTranspose forward (Wt[Cin][Cout][T]), 3 ConvTranspose input gradient.  Cin / Cout are the LAYER's.
// both GEMM operands of a layer in one launch: dst_fwd = role 0 (conv) / 2 (transposed), dst_bwd = role 1 / 3
__global__ __launch_bounds__(256) void cb_pack_weights2_kernel(const float *__restrict__ w, bf16_t *__restrict__ d0, bf16_t *__restrict__ d1, int C0, int C1,
                                                              int T, int oa0, int Ci0, int Co0, int CoP0, int nq20, int oa1, int Ci1, int Co1, int CoP1,
                                                              int nq21) {
    const long long n0 = (long long)nq20 * CoP0 * 8, n1 = (long long)nq21 * CoP1 * 8;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < n0 + n1; idx += (long long)gridDim.x * 256) {
        const bool second = idx >= n0;
        const long long k = second ? idx - n0 : idx;
        const int Ci = second ? Ci1 : Ci0, Co = second ? Co1 : Co0, CoP = second ? CoP1 : CoP0, oa = second ? oa1 : oa0;
        const int nC8 = Ci / 8;
        const int j = (int)(k & 7), o = (int)((k >> 3) % CoP), q = (int)((k >> 3) / CoP);
        const int t = q / nC8, i = (q % nC8) * 8 + j;
        float v = 0.f;
        if (t < T && o < Co) {
            const int c0 = oa ? o : i, c1 = oa ? i : o;
            v = w[((size_t)c0 * C1 + c1) * T + t];
        }
        (second ? d1 : d0)[k] = f2bf(v);
    }
}

extern "C" int hno_cb_pack_weights_both(const float *W, void *dst_fwd, void *dst_bwd, int transposed, int Cin, int Cout, int ks, void *stream) {
    HNO_REQUIRE(W && dst_fwd && dst_bwd && Cin > 0 && Cout > 0 && ks >= 1 && ks <= 3, "hno_cb_pack_weights_both: bad argument");
    if ((Cin % 8) || (Cout % 8)) return fail(HNO_ELIMIT, "hno_cb_pack_weights_both: %d -> %d channels (multiples of 8)", Cin, Cout);
    const int T = ks * ks * ks;
    const int C0 = transposed ? Cin : Cout, C1 = transposed ? Cout : Cin;
    // forward: GEMM in = Cin, out = Cout; backward (input gradient): in = Cout, out = Cin
    const int oa0 = transposed ? 0 : 1, oa1 = transposed ? 1 : 0;
    const int nq0 = T * (Cin / 8), nq20 = (nq0 + 1) & ~1, CoP0 = (Cout + 31) / 32 * 32;
    const int nq1 = T * (Cout / 8), nq21 = (nq1 + 1) & ~1, CoP1 = (Cin + 31) / 32 * 32;
    hipLaunchKernelGGL(cb_pack_weights2_kernel, dim3(gsz((long long)nq20 * CoP0 * 8 + (long long)nq21 * CoP1 * 8)), dim3(256), 0, (hipStream_t)stream, W,
                       (bf16_t *)dst_fwd, (bf16_t *)dst_bwd, C0, C1, T, oa0, Cin, Cout, CoP0, nq20, oa1, Cout, Cin, CoP1, nq21);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

// every layer of a model in ONE launch: `table` is (nrows, 16) int64 on the device:
//   [W, dst_fwd, dst_bwd, C0, C1, T, oa0, Ci0, Co0, CoP0, nq20, oa1, Ci1, Co1, CoP1, nq21]  (see hno_cb_pack_weights_both)
// Tiled form of cb_pack_weights_multi_kernel: a workgroup takes the weights of 8 values of the parameter's first channel index and 32
// of its second one, all taps -- 8 contiguous runs of 32 T floats -- through LDS, and writes both packed images in whole 16-byte rows
// (8 consecutive GEMM input channels of one output channel): 128- or 512-byte contiguous pieces.  The element-wise kernel below reads the
// parameter with a stride of T floats (one 4-byte gather per element: 184 us for the 22.5 M weights of V-Net-DS).  Padding (output channels
// Co .. CoP, the odd q row) is NOT written here: the buffers are zeroed once when they are allocated.
#define CB_PT_C0 8
#define CB_PT_C1 32
__global__ __launch_bounds__(256) void cb_pack_weights_tiled_kernel(const long long *__restrict__ table, int nrows) {
    // + 1: in the output-major branch the eight lanes of a 16-byte chunk read the eight rows at the same column -- a row stride of
    // 32 x 27 floats put all of them in one bank (8-way conflict on every read)
    __shared__ float wt[CB_PT_C0][CB_PT_C1 * 27 + 1];
    int row_id = 0;
    long long first = 0;
    int g1n = 1;
    for (; row_id < nrows; ++row_id) {
        const long long *r = table + (size_t)row_id * 16;
        g1n = (int)((r[4] + CB_PT_C1 - 1) / CB_PT_C1);
        const long long nb = (r[3] / CB_PT_C0) * g1n;
        if ((long long)blockIdx.x < first + nb) break;
        first += nb;
    }
    if (row_id >= nrows) return;
    const long long *row = table + (size_t)row_id * 16;
    const float *w = reinterpret_cast<const float *>(row[0]);
    const int C1 = (int)row[4], T = (int)row[5];
    const int local = (int)((long long)blockIdx.x - first);
    const int g0 = local / g1n, g1 = local - g0 * g1n;
    const int c0_0 = g0 * CB_PT_C0, c1_0 = g1 * CB_PT_C1;
    const int c1n = C1 - c1_0 < CB_PT_C1 ? C1 - c1_0 : CB_PT_C1;       // a multiple of 8
    for (int c0l = 0; c0l < CB_PT_C0; ++c0l) {
        const float *src = w + ((size_t)(c0_0 + c0l) * C1 + c1_0) * T;
        for (int e = threadIdx.x; e < c1n * T; e += 256) wt[c0l][e] = src[e];
    }
    __syncthreads();
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        bf16_t *d = reinterpret_cast<bf16_t *>(row[1 + which]);
        const int oa = (int)row[6 + 5 * which], Ci = (int)row[7 + 5 * which], CoP = (int)row[9 + 5 * which];
        const int nC8 = Ci / 8;
        if (oa) {            // first index = output channel o, second = input channel i
            const int ng = c1n / 8, nchunk = T * ng * CB_PT_C0;
            for (int ch = threadIdx.x; ch < nchunk; ch += 256) {
                const int ol = ch & 7, rest = ch >> 3, gi = rest % ng, t = rest / ng;
                const int q = t * nC8 + c1_0 / 8 + gi, o = c0_0 + ol;
                unsigned out[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    out[k] = (unsigned)f2bf(wt[ol][(gi * 8 + 2 * k) * T + t]) | ((unsigned)f2bf(wt[ol][(gi * 8 + 2 * k + 1) * T + t]) << 16);
                *reinterpret_cast<uint4 *>(d + ((size_t)q * CoP + o) * 8) = make_uint4(out[0], out[1], out[2], out[3]);
            }
        } else {             // first index = input channel i (one group of 8), second = output channel o
            const int nchunk = T * c1n;
            for (int ch = threadIdx.x; ch < nchunk; ch += 256) {
                const int ol = ch % c1n, t = ch / c1n;
                const int q = t * nC8 + c0_0 / 8, o = c1_0 + ol;
                unsigned out[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) out[k] = (unsigned)f2bf(wt[2 * k][ol * T + t]) | ((unsigned)f2bf(wt[2 * k + 1][ol * T + t]) << 16);
                *reinterpret_cast<uint4 *>(d + ((size_t)q * CoP + o) * 8) = make_uint4(out[0], out[1], out[2], out[3]);
            }
        }
    }
}

__global__ __launch_bounds__(256) void cb_pack_weights_multi_kernel(const long long *__restrict__ table, int nrows) {
    // workgroup -> (row, chunk of 2048 elements): walk the rows' chunk counts (a few dozen rows)
    int row_id = 0;
    long long first = 0;
    for (; row_id < nrows; ++row_id) {
        const long long *r = table + (size_t)row_id * 16;
        const long long n = r[10] * r[9] * 8 + r[15] * r[14] * 8;
        const long long nb = (n + 2047) / 2048;
        if ((long long)blockIdx.x < first + nb) break;
        first += nb;
    }
    if (row_id >= nrows) return;
    const long long *row = table + (size_t)row_id * 16;
    const float *w = reinterpret_cast<const float *>(row[0]);
    bf16_t *d0 = reinterpret_cast<bf16_t *>(row[1]), *d1 = reinterpret_cast<bf16_t *>(row[2]);
    const int C1 = (int)row[4], T = (int)row[5];
    const int oa0 = (int)row[6], Ci0 = (int)row[7], Co0 = (int)row[8], CoP0 = (int)row[9], nq20 = (int)row[10];
    const int oa1 = (int)row[11], Ci1 = (int)row[12], Co1 = (int)row[13], CoP1 = (int)row[14], nq21 = (int)row[15];
    // 32-bit element arithmetic (a layer's packed weights are far below 2^31 elements: checked by the host when the row is built);
    // the 64-bit divisions of the first version were most of this kernel's time
    const unsigned n0 = (unsigned)nq20 * CoP0 * 8u, n1 = (unsigned)nq21 * CoP1 * 8u;
    const unsigned lo = (unsigned)((long long)blockIdx.x - first) * 2048u;
    for (unsigned idx = lo + threadIdx.x; idx < lo + 2048u && idx < n0 + n1; idx += 256u) {
        const bool second = idx >= n0;
        const unsigned k = second ? idx - n0 : idx;
        const int Ci = second ? Ci1 : Ci0, Co = second ? Co1 : Co0, CoP = second ? CoP1 : CoP0, oa = second ? oa1 : oa0;
        const unsigned nC8 = (unsigned)Ci / 8u;
        const unsigned k8 = k >> 3, qq = k8 / (unsigned)CoP;
        const int j = (int)(k & 7u), o = (int)(k8 - qq * (unsigned)CoP), q = (int)qq;
        const int t = (int)(qq / nC8), i = (int)(qq - (qq / nC8) * nC8) * 8 + j;
        float v = 0.f;
        if (t < T && o < Co) {
            const int c0 = oa ? o : i, c1 = oa ? i : o;
            v = w[((size_t)c0 * C1 + c1) * T + t];
        }
        (second ? d1 : d0)[k] = f2bf(v);
    }
}

// fills row `r` (16 int64, HOST memory) of the table of hno_cb_pack_weights_multi for one layer
extern "C" int hno_cb_pack_table_row(long long *row, const float *W, void *dst_fwd, void *dst_bwd, int transposed, int Cin, int Cout, int ks) {
    HNO_REQUIRE(row && W && dst_fwd && dst_bwd && Cin > 0 && Cout > 0 && ks >= 1 && ks <= 3, "hno_cb_pack_table_row: bad argument");
    if ((Cin % 8) || (Cout % 8)) return fail(HNO_ELIMIT, "hno_cb_pack_table_row: %d -> %d channels (multiples of 8)", Cin, Cout);
    const int T = ks * ks * ks;
    if ((long long)T * Cin * ((Cout + 31) / 32 * 32) + (long long)T * Cout * ((Cin + 31) / 32 * 32) >= (1ll << 30))
        return fail(HNO_ELIMIT, "hno_cb_pack_table_row: %d -> %d channels exceed the pack kernel's 32-bit element range", Cin, Cout);
    row[0] = (long long)(size_t)W; row[1] = (long long)(size_t)dst_fwd; row[2] = (long long)(size_t)dst_bwd;
    row[3] = transposed ? Cin : Cout; row[4] = transposed ? Cout : Cin; row[5] = T;
    const int nq0 = T * (Cin / 8), nq1 = T * (Cout / 8);
    row[6] = transposed ? 0 : 1; row[7] = Cin; row[8] = Cout; row[9] = (Cout + 31) / 32 * 32; row[10] = (nq0 + 1) & ~1;
    row[11] = transposed ? 1 : 0; row[12] = Cout; row[13] = Cin; row[14] = (Cin + 31) / 32 * 32; row[15] = (nq1 + 1) & ~1;
    return HNO_OK;
}

// workgroups hno_cb_pack_weights_multi needs for this row (sum them over the table for its `total_chunks`)
extern "C" long long hno_cb_pack_row_chunks(const long long *row) {
    return row ? (row[3] / CB_PT_C0) * ((row[4] + CB_PT_C1 - 1) / CB_PT_C1) : 0;
}

extern "C" int hno_cb_pack_weights_multi(const void *table_dev, int nrows, long long total_chunks, void *stream) {
    HNO_REQUIRE(table_dev && nrows > 0 && total_chunks > 0 && total_chunks < (1ll << 31), "hno_cb_pack_weights_multi: bad argument");
    hipLaunchKernelGGL(cb_pack_weights_tiled_kernel, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)stream, (const long long *)table_dev, nrows);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

extern "C" int hno_cb_pack_weights(const float *W, void *dst, int role, int Cin, int Cout, int ks, void *stream) {
    HNO_REQUIRE(W && dst && role >= 0 && role <= 3 && Cin > 0 && Cout > 0 && ks >= 1 && ks <= 3, "hno_cb_pack_weights: bad argument");
    const int gi = (role == 0 || role == 2) ? Cin : Cout, go = (role == 0 || role == 2) ? Cout : Cin;   // GEMM in / out channels
    if (gi % 8) return fail(HNO_ELIMIT, "hno_cb_pack_weights: %d GEMM input channels (must be a multiple of 8)", gi);
    const int C0 = role <= 1 ? Cout : Cin, C1 = role <= 1 ? Cin : Cout;
    const int out_is_axis0 = (role == 0 || role == 3);
    const int T = ks * ks * ks;
    const int nq = T * (gi / 8), nq2 = (nq + 1) & ~1, CoP = (go + 31) / 32 * 32;
    hipLaunchKernelGGL(cb_pack_weights_kernel, dim3(gsz((long long)nq2 * CoP * 8)), dim3(256), 0, (hipStream_t)stream, W, (bf16_t *)dst, C0, C1, T,
                       out_is_axis0, gi, go, CoP, nq2);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

static int cb_pick_ksplit(int B, int Cout, long long Vo, int nsteps) {
    const long long waves = (long long)B * ((Vo + 63) / 64) * ((Cout + 95) / 96);
    static const int ks_max = getenv("HNO_CB_KSPLIT_MAX") ? atoi(getenv("HNO_CB_KSPLIT_MAX")) : 32;     // A/B aid (1 ... 32)
    // split K until the launch has one wave per SIMD (1 024).  Rounds 2-4 split to two (2 048): the mid levels of V-Net-DS then wrote and
    // re-read twice the partial slabs for no gain in occupancy -- cfg4 bf16 step 7.78 -> 7.59 ms (384 ... 1 024: 7.57-7.59; 256: 7.66;
    // 128: 7.93; capping the split depth instead hurts the deepest levels: max 16 / 8 / 4 -> 7.82 / 7.93 / 8.36)
    static const int wave_target = getenv("HNO_CB_KSPLIT_WAVES") ? atoi(getenv("HNO_CB_KSPLIT_WAVES")) : 1024;
    int ks = 1;
    while (waves * ks < wave_target && ks * 2 <= nsteps && ks < 32 && ks * 2 <= ks_max) ks *= 2;
    return ks;
}

extern "C" size_t hno_cb_conv_workspace_bytes(int B, int Cin, int Cout, int Do, int Ho, int Wo, int ks) {
    const long long Vo = (long long)Do * Ho * Wo;
    const int nq = ks * ks * ks * (Cin / 8), nsteps = ((nq + 1) & ~1) / 2;
    const int kz = cb_pick_ksplit(B, Cout, Vo, nsteps);
    size_t part = kz > 1 ? (size_t)kz * B * Vo * Cout * sizeof(float) : 0;
    if (ks == 3 && Cin % 24 == 0 && Cin >= 48) {             // the halo-tile kernel may slice the channel chunks 2 or 4 ways
        const size_t halo_part = (size_t)8 * B * Vo * Cout * sizeof(float);
        if (halo_part <= ((size_t)96 << 20) && halo_part > part) part = halo_part;
    }
    // statistics partials: at most one pair per 64-voxel x 32-channel block, or per finish block
    size_t stats = (size_t)B * (((Vo + 63) / 64) * ((Cout + 31) / 32) + (size_t)Do * Ho * ((Cout + 31) / 32) + 4096) * 2 * sizeof(float);
    return part + stats + 256;
}

// y = conv(x) (+ bias), channels-last bf16 in and out.  mode 0: correlation gather in = stride * out - pad + tap;
// mode 1: fractional gather in = (out + pad - tap) / stride (ConvTranspose forward, input gradient of a conv).
// wpacked: from hno_cb_pack_weights with the matching role.  Cin = Ca + Cb GEMM input channels, Cout GEMM output channels.
// mean_rstd (B, 2) non-null: GroupNorm(1, Cout) statistics of the rounded output are produced in the same pass.
// nstat_out non-null ("lazy statistics"): mean_rstd must hold hno_cb_conv_stats_floats() floats; the (sum, sum of squares) partials
// are left behind its (B, 2) slot, their count per sample is returned, and hno_cb_gn_apply (nstat > 0) finishes them -- no finalize launch.
extern "C" size_t hno_cb_conv_stats_floats(int B, int Cout, int Do, int Ho, int Wo) {
    const long long Vo = (long long)Do * Ho * Wo;
    return (size_t)2 * B + (size_t)B * (((Vo + 63) / 64) * ((Cout + 31) / 32) + (size_t)Do * Ho * ((Cout + 31) / 32) + 4096) * 2;
}
static int cb_conv_impl(const void *xa, int Ca, const void *xb, int Cb, const void *wpacked, const float *bias, void *y,
                        float *mean_rstd, float eps, void *workspace, size_t workspace_bytes, int mode, int B, int Cout,
                        int Di, int Hi, int Wi, int Do, int Ho, int Wo, int ks, int stride, int pad, int *nstat_out, void *stream,
                        void *y2, int c_split) {
    HNO_REQUIRE(c_split == 0 || (y2 && c_split > 0 && c_split < Cout && c_split % 8 == 0), "hno_cb_conv_split: bad channel split");
    const bool lazy = nstat_out != nullptr && mean_rstd != nullptr;
    if (nstat_out) *nstat_out = 0;
    HNO_REQUIRE(xa && wpacked && y && workspace && B > 0 && Cout > 0 && Ca > 0 && Cb >= 0, "hno_cb_conv: bad argument");
    HNO_REQUIRE((mode == 0 || mode == 1) && (stride == 1 || stride == 2) && ks >= 1 && ks <= 3, "hno_cb_conv: bad mode / stride / kernel");
    HNO_REQUIRE(Cb == 0 || xb, "hno_cb_conv: second input missing");
    if ((Ca % 8) || (Cb % 8) || (Cout % 8)) return fail(HNO_ELIMIT, "hno_cb_conv: channel counts %d + %d -> %d must be multiples of 8", Ca, Cb, Cout);
    const long long Vo = (long long)Do * Ho * Wo;
    if ((long long)(Ca + Cb) * Di * Hi * Wi >= (1ll << 40) || Vo >= (1ll << 31)) return fail(HNO_ELIMIT, "hno_cb_conv: grid too large");
    hipStream_t s = (hipStream_t)stream;
    CbArgs a = {};
    a.xa = (const bf16_t *)xa; a.xb = (const bf16_t *)xb; a.Ca = Ca; a.Cb = Cb; a.w = (const bf16_t *)wpacked; a.bias = bias;
    a.y = (bf16_t *)y; a.y2 = (bf16_t *)y2; a.csplit = c_split; a.B = B; a.Cout = Cout; a.CoP = (Cout + 31) / 32 * 32;
    a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Do = Do; a.Ho = Ho; a.Wo = Wo; a.ks = ks; a.stride = stride; a.pad = pad; a.frac = mode;
    a.ntaps = ks * ks * ks;
    const int nq = a.ntaps * ((Ca + Cb) / 8);
    a.nq2 = (nq + 1) & ~1;
    // stride-2 fractional gather: by parity class (CbArgs.cls); a class visits 1, 2, 4 or 8 of the 27 taps (ks = 3) -- the split over K
    // is chosen for the average class
    a.cls = (mode == 1 && stride == 2 && !(debug_flags() & 2048)) ? 1 : 0;
    const int nsteps = a.cls ? (((a.ntaps + 7) / 8) * ((Ca + Cb) / 8) + 1) / 2 : a.nq2 / 2;
    const int kz = cb_pick_ksplit(B, Cout, Vo, nsteps);
    a.ksplit = kz;
    const size_t part_bytes = kz > 1 ? (size_t)kz * B * Vo * Cout * sizeof(float) : 0;
    HNO_REQUIRE(workspace_bytes >= hno_cb_conv_workspace_bytes(B, Ca + Cb, Cout, Do, Ho, Wo, ks), "hno_cb_conv: workspace too small");
    a.part = kz > 1 ? (float *)workspace : nullptr;
    float *stats = lazy ? mean_rstd + 2 * B : (float *)((char *)workspace + ((part_bytes + 255) & ~(size_t)255));
    a.stats = mean_rstd ? stats : nullptr;
    int nblk_stats;
    const double flops = 2.0 * B * Vo * (double)Cout * a.ntaps * (Ca + Cb);
    // halo-tile kernel: 3x3x3, stride 1, same grid in and out, channel chunks of 24, and a grid wide enough to fill the tiles.
    // Tile shape per layer: MT position tiles per wave (band of 128 MT positions), NT channel tiles, and a split of the channel
    // chunks over gridDim.z -- the largest tiles that still give every CU ~3 workgroups (a 21 x 25 x 17 level with 96 -> 96
    // channels is 42 workgroups at <2, 3>: 72 TFLOP/s; at <1, 1> x 2 chunk slices 630).
    if (ks == 3 && stride == 1 && pad == 1 && Di == Do && Hi == Ho && Wi == Wo && Ca % 24 == 0 && Cb % 24 == 0 && !(debug_flags() & 256)) {
        const int S = Wo + 2;
        const int nchunk = (Ca + Cb) / 24;
        const int ntmax = Cout <= 32 ? 1 : (Cout <= 64 ? 2 : 3);
        int bMT = 0, bNT = 0, bKS = 0, bTH = 0;
        double best = 0.0;
        static const int halo_ksmax = getenv("HNO_HALO_KSMAX") ? atoi(getenv("HNO_HALO_KSMAX")) : 8;          // A/B aids (4 and 512 until round 6: cfg4 step 6.48 -> 6.43 ms)
        static const double halo_fill = getenv("HNO_HALO_FILL") ? atof(getenv("HNO_HALO_FILL")) : 1024.0;
        for (int MT = 2; MT >= 1; --MT)
            for (int NT = ntmax; NT >= 1; --NT)
                for (int ksp = 1; ksp <= halo_ksmax && ksp <= nchunk; ksp *= 2) {
                    if (nchunk % ksp) continue;
                    int TH = (4 * MT * 32) / S;
                    if (TH > Ho) TH = Ho;
                    if (TH < 1) continue;
                    const double util = (double)TH * S / (4 * MT * 32);
                    if (util < 0.6) continue;
                    const long long blocks = (long long)B * Do * ((Ho + TH - 1) / TH) * ((Cout + 32 * NT - 1) / (32 * NT)) * ksp;
                    const double fill = blocks >= halo_fill ? 1.0 : (double)blocks / halo_fill;
                    // shape factors fitted to a sweep of all shapes over the level-1 / level-2 layers (tools/bench_cb_conv.py with
                    // HNO_HALO_SHAPE): small bands and many workgroups win (co-resident workgroups run in lockstep, so parallelism
                    // has to come from the grid), fewer channel tiles re-stage the image, slices add the fp32 round trip
                    const double shape = (MT == 2 ? 0.95 : 1.0) * (1.0 - 0.1 * (ntmax - NT)) * (ksp == 1 ? 1.0 : (ksp == 2 ? 0.8 : (ksp == 4 ? 0.65 : 0.55)));
                    const double score = fill * util * shape;
                    if (score > best) { best = score; bMT = MT; bNT = NT; bKS = ksp; bTH = TH; }
                }
        if (const char *force = getenv("HNO_HALO_SHAPE")) {      // tuning aid: "MT,NT,KS"
            int fm = 0, fn = 0, fk = 0;
            if (sscanf(force, "%d,%d,%d", &fm, &fn, &fk) == 3 && fm >= 1 && fm <= 2 && fn >= 1 && fn <= ntmax && fk >= 1 && nchunk % fk == 0) {
                int TH = (4 * fm * 32) / S;
                if (TH > Ho) TH = Ho;
                if (TH >= 1) { bMT = fm; bNT = fn; bKS = fk; bTH = TH; }
            }
        }
        const size_t part_need = bKS > 1 ? (size_t)bKS * B * Vo * Cout * sizeof(float) : 0;
        const size_t stat_need = (size_t)B * ((size_t)Do * ((Ho + (bTH ? bTH : 1) - 1) / (bTH ? bTH : 1)) * ((Cout + 32 * (bNT ? bNT : 1) - 1) / (32 * (bNT ? bNT : 1))) + 1024) * 2 * sizeof(float);
        if (bMT && ((part_need + 255) & ~(size_t)255) + stat_need <= workspace_bytes) {
            ChArgs h = {};
            h.xa = a.xa; h.xb = a.xb; h.Ca = Ca; h.Cb = Cb; h.w = a.w; h.bias = bias; h.y = a.y; h.y2 = a.y2; h.csplit = a.csplit;
            h.B = B; h.Cout = Cout; h.CoP = a.CoP; h.D = Do; h.H = Ho; h.W = Wo; h.TH = bTH; h.S = S; h.flip = mode; h.dbg = debug_flags();
            h.nbands = (Ho + bTH - 1) / bTH;
            h.npos = (2 * (bTH + 2) + 2) * S + 2 + 4 * bMT * 32 + 8;
            if (h.npos < 3 * (bTH + 2) * S) h.npos = 3 * (bTH + 2) * S;
            h.ksplit = bKS;
            h.part = bKS > 1 ? (float *)workspace : nullptr;
            float *hstats = lazy ? mean_rstd + 2 * B : (float *)((char *)workspace + ((part_need + 255) & ~(size_t)255));
            h.stats = mean_rstd ? hstats : nullptr;
            const size_t lds = (size_t)h.npos * 48 + 96 * sizeof(int2);
            const dim3 g((unsigned)(B * Do * h.nbands), (Cout + 32 * bNT - 1) / (32 * bNT), bKS);
            {
                ProfScope _ps(KID_CB_CONV, s, flops);
                const bool wide = 3 * S > 128;      // staging variant (template: the batched form's registers stay out of the narrow kernels)
#define HNO_HALO_CASE(MTv, NTv, WIDEv)                                                                                                       \
    if (bMT == MTv && bNT == NTv && wide == WIDEv) {                                                                                         \
        static int attr_set = -1;                                                                                                        \
        if (lds > 48 * 1024 && attr_set != current_device()) {                                                                                             \
            HNO_CHECK_HIP(hipFuncSetAttribute((const void *)cb_halo_kernel<MTv, NTv, WIDEv>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024)); \
            attr_set = current_device();                                                                                                                 \
        }                                                                                                                                    \
        hipLaunchKernelGGL((cb_halo_kernel<MTv, NTv, WIDEv>), g, dim3(256), lds, s, h);                                                      \
    }
                HNO_HALO_CASE(2, 1, true) HNO_HALO_CASE(2, 2, true) HNO_HALO_CASE(2, 3, true) HNO_HALO_CASE(1, 1, true) HNO_HALO_CASE(1, 2, true) HNO_HALO_CASE(1, 3, true)
                HNO_HALO_CASE(2, 1, false) HNO_HALO_CASE(2, 2, false) HNO_HALO_CASE(2, 3, false) HNO_HALO_CASE(1, 1, false) HNO_HALO_CASE(1, 2, false) HNO_HALO_CASE(1, 3, false)
#undef HNO_HALO_CASE
                HNO_CHECK_LAUNCH();
            }
            int nstat = (int)(Do * h.nbands * g.y);
            if (bKS > 1) {
                const long long per_sample = Vo * Cout;
                const int gx = gsz(per_sample / 4, 256, 1024);      // four elements per thread
                hipLaunchKernelGGL(cb_splitk_finish_kernel, dim3(gx, B), dim3(256), 0, s, (const float *)h.part, (bf16_t *)y, mean_rstd ? hstats : nullptr,
                                   bKS, per_sample, B, a.y2, a.csplit, Cout);
                HNO_CHECK_LAUNCH();
                nstat = gx;
            }
            if (lazy) {
                *nstat_out = nstat;
            } else if (mean_rstd) {
                hipLaunchKernelGGL(cb_gn_finalize_kernel, dim3(B), dim3(256), 0, s, (const float *)hstats, nstat, (double)Vo * Cout, eps, mean_rstd);
                HNO_CHECK_LAUNCH();
            }
            return HNO_OK;
        }
    }
    {
        ProfScope _ps(KID_CB_CONV, s, flops);
        // workgroups along x: 128 MT voxels each; in class mode per parity class (a workgroup never mixes classes)
        auto grid_x = [&](int MT) -> unsigned {
            const int per = 4 * 32 * MT;
            if (!a.cls) return (unsigned)((Vo + per - 1) / per);
            int n = 0;
            for (int c = 0; c < 8; ++c) {
                a.cls_blk0[c] = n;
                const long long vc = (long long)((Do + 1 - (c >> 2)) >> 1) * ((Ho + 1 - ((c >> 1) & 1)) >> 1) * ((Wo + 1 - (c & 1)) >> 1);
                n += (int)((vc + per - 1) / per);
            }
            a.cls_blk0[8] = n;
            return (unsigned)n;
        };
        // position tiles per wave (MT): the largest that still gives the chip `mt_wgs` workgroups -- 4 x 128 voxels per workgroup left the
        // strided 24 -> 24 convolution between the two shallow levels with 130 workgroups (58 us for 25 MB, round 6 trace)
        static const int mt_wgs = getenv("HNO_CB_MT_WGS") ? atoi(getenv("HNO_CB_MT_WGS")) : 1536;     // A/B aid (0: the fixed shapes of rounds 2-5; cfg4 step at 0 / 768 / 1536 / 3072: 6.56 / 6.52 / 6.48 / 6.55 ms)
        const int ny = Cout <= 64 ? 1 : (Cout + 95) / 96;
        auto pick_mt = [&](int mt_max) {
            int mt = mt_max;
            while (mt > 1 && (long long)grid_x(mt) * ny * B * kz < mt_wgs) mt >>= 1;
            return mt;
        };
        if (Cout <= 32) {
            const int mt = pick_mt(4);
            const dim3 g(grid_x(mt), 1, B * kz);
            if (mt == 4) hipLaunchKernelGGL((cb_gather_kernel<4, 1>), g, dim3(256), 0, s, a);
            else if (mt == 2) hipLaunchKernelGGL((cb_gather_kernel<2, 1>), g, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((cb_gather_kernel<1, 1>), g, dim3(256), 0, s, a);
            nblk_stats = g.x * g.y;
        } else if (Cout <= 64) {
            const int mt = pick_mt(2);
            const dim3 g(grid_x(mt), 1, B * kz);
            if (mt == 2) hipLaunchKernelGGL((cb_gather_kernel<2, 2>), g, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((cb_gather_kernel<1, 2>), g, dim3(256), 0, s, a);
            nblk_stats = g.x * g.y;
        } else {
            const int mt = pick_mt(2);
            const dim3 g(grid_x(mt), (Cout + 95) / 96, B * kz);
            if (mt == 2) hipLaunchKernelGGL((cb_gather_kernel<2, 3>), g, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((cb_gather_kernel<1, 3>), g, dim3(256), 0, s, a);
            nblk_stats = g.x * g.y;
        }
        HNO_CHECK_LAUNCH();
    }
    if (kz > 1) {
        const long long per_sample = Vo * Cout;
        const int gx = gsz(per_sample / 4, 256, 1024);      // four elements per thread
        hipLaunchKernelGGL(cb_splitk_finish_kernel, dim3(gx, B), dim3(256), 0, s, (const float *)a.part, (bf16_t *)y, mean_rstd ? stats : nullptr, kz,
                           per_sample, B, a.y2, a.csplit, Cout);
        HNO_CHECK_LAUNCH();
        nblk_stats = gx;
    }
    if (lazy) {
        *nstat_out = nblk_stats;
    } else if (mean_rstd) {
        hipLaunchKernelGGL(cb_gn_finalize_kernel, dim3(B), dim3(256), 0, s, (const float *)stats, nblk_stats, (double)Vo * Cout, eps, mean_rstd);
        HNO_CHECK_LAUNCH();
    }
    return HNO_OK;
}

extern "C" int hno_cb_conv(const void *xa, int Ca, const void *xb, int Cb, const void *wpacked, const float *bias, void *y,
                           float *mean_rstd, float eps, void *workspace, size_t workspace_bytes, int mode, int B, int Cout,
                           int Di, int Hi, int Wi, int Do, int Ho, int Wo, int ks, int stride, int pad, int *nstat_out, void *stream) {
    return cb_conv_impl(xa, Ca, xb, Cb, wpacked, bias, y, mean_rstd, eps, workspace, workspace_bytes, mode, B, Cout, Di, Hi, Wi, Do, Ho, Wo, ks,
                        stride, pad, nstat_out, stream, nullptr, 0);
}

// The same GEMM with its output channels written to TWO tensors: channels [0, c_split) to y (B, Do, Ho, Wo, c_split), the rest to y2
// (B, Do, Ho, Wo, Cout - c_split).  The input gradient of a two-input convolution (the decoder's fused concat) is the gradient of both
// inputs: written apart, the consumers need no channel-split copies (16 ATen copies per V-Net-DS step in round 4).
extern "C" int hno_cb_conv_split(const void *xa, int Ca, const void *xb, int Cb, const void *wpacked, const float *bias, void *y, void *y2,
                                 int c_split, void *workspace, size_t workspace_bytes, int mode, int B, int Cout, int Di, int Hi, int Wi,
                                 int Do, int Ho, int Wo, int ks, int stride, int pad, void *stream) {
    HNO_REQUIRE(y2 && c_split > 0, "hno_cb_conv_split: bad argument");
    return cb_conv_impl(xa, Ca, xb, Cb, wpacked, bias, y, nullptr, 1e-5f, workspace, workspace_bytes, mode, B, Cout, Di, Hi, Wi, Do, Ho, Wo, ks,
                        stride, pad, nullptr, stream, y2, c_split);
}

extern "C" int hno_cb_gn_apply(const void *y1, const float *mr1, const float *gamma1, const float *beta1, const void *y2,
                               const float *mr2, const float *gamma2, const float *beta2, void *z, int B, int C, long long V, int act,
                               int nstat1, int nstat2, float eps, void *stream) {
    HNO_REQUIRE(y1 && mr1 && gamma1 && beta1 && z && B > 0 && C > 0 && V > 0, "hno_cb_gn_apply: bad argument");
    HNO_REQUIRE(!y2 || (mr2 && gamma2 && beta2), "hno_cb_gn_apply: second branch incomplete");
    if (C % 8 || C > 2048) return fail(HNO_ELIMIT, "hno_cb_gn_apply: C = %d must be a multiple of 8 (<= 2048)", C);
    hipStream_t s = (hipStream_t)stream;
    const long long per_sample = V * C;
    if (per_sample >= (1ll << 34)) return fail(HNO_ELIMIT, "hno_cb_gn_apply: %lld elements per sample exceed the kernel's 32-bit item index", per_sample);
    ProfScope _ps(KID_CB_GN, s, (double)B * per_sample * (y2 ? 6.0 : 4.0));
    // (lazy statistics: every workgroup re-reads the partials, so fewer, longer-running workgroups)
    const int cap = (nstat1 > 512 || nstat2 > 512) ? 1024 : 2048;
    hipLaunchKernelGGL(cb_gn_apply_kernel, dim3(gsz(per_sample / 8, 256, cap), B), dim3(256), 4 * (size_t)C * sizeof(float), s, (const bf16_t *)y1, mr1, gamma1, beta1,
                       (const bf16_t *)y2, mr2, gamma2, beta2, (bf16_t *)z, C, per_sample, act, nstat1, nstat2, eps, B);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

#define CB_GN_BWD_BLOCKS 1024     // upper bound; small tensors use fewer (>= 64 voxels per workgroup)
extern "C" size_t hno_cb_gn_bwd_workspace_bytes(int B, int C) {
    return ((size_t)B * CB_GN_BWD_BLOCKS * 3 * C + 4 * (size_t)B * C + 2 * (size_t)(B > 8 ? B : 8) + 2 * (size_t)B * CB_GN_BWD_BLOCKS) * sizeof(float) + 256;
}

// backward of z = act(gamma (y - mean) rstd + beta) w.r.t. y, gamma, beta.  accumulate != 0: dgamma / dbeta += .
extern "C" int hno_cb_gn_bwd(const void *dz, const void *y, const float *mr, const float *gamma, const float *beta, void *dy,
                             float *dgamma, float *dbeta, float *dy_colsum, void *workspace, int B, int C, long long V, int act,
                             int accumulate, void *stream) {
    HNO_REQUIRE(dz && y && mr && gamma && beta && dy && dgamma && dbeta && workspace && B > 0 && C > 0 && V > 0, "hno_cb_gn_bwd: bad argument");
    if (C % 8 || C > 2048) return fail(HNO_ELIMIT, "hno_cb_gn_bwd: C = %d must be a multiple of 8 (<= 2048)", C);
    if (V * C >= (1ll << 34)) return fail(HNO_ELIMIT, "hno_cb_gn_bwd: %lld elements per sample exceed the kernels' 32-bit item index", V * C);
    hipStream_t s = (hipStream_t)stream;
    float *slab = (float *)workspace;
    float *gS = slab + (size_t)B * CB_GN_BWD_BLOCKS * 3 * C;
    // a streaming pass: enough workgroups to cover the latency (256 of them left one workgroup per CU: 0.9 TB/s)
    // (>= 64 voxels per workgroup left the wide deep levels with 4-21 workgroups whose threads walked 11 dependent load pairs: 12-17 us
    // for 0.3-1 MB.  Now ~2 items (8 channels of a voxel) per thread: 512 / (C / 8) voxels per workgroup, between 8 and 64)
    static const int per_wg_items = getenv("HNO_GN_BWD_ITEMS") ? atoi(getenv("HNO_GN_BWD_ITEMS")) : 512;      // A/B aid
    int vox = per_wg_items / (C / 8);
    vox = vox < 8 ? 8 : (vox > 64 ? 64 : vox);
    int nblk = (int)((V + vox - 1) / vox);
    if (nblk > CB_GN_BWD_BLOCKS) nblk = CB_GN_BWD_BLOCKS;
    if (nblk < 1) nblk = 1;
    {
        ProfScope _ps(KID_CB_GN, s, (double)B * V * C * 4.0);
        static const bool apart = getenv("HNO_GN_FINALIZE_LAUNCH") && atoi(getenv("HNO_GN_FINALIZE_LAUNCH")) == 1;      // A/B: pass 1b as its own launch
        float *kpart = apart ? nullptr : gS + 4 * (size_t)B * C + 2 * (size_t)(B > 8 ? B : 8);
        hipLaunchKernelGGL(cb_gn_bwd_reduce_kernel, dim3(nblk, B), dim3(256), 256 * 24 * sizeof(float), s, (const bf16_t *)dz, (const bf16_t *)y, mr, gamma,
                           beta, slab, C, V, act, kpart);
        HNO_CHECK_LAUNCH();
        if (apart) {
            hipLaunchKernelGGL(cb_gn_bwd_finalize_kernel, dim3((C + 256 / cb_gn_slices(nblk) - 1) / (256 / cb_gn_slices(nblk))), dim3(256), 0, s, (const float *)slab, gamma, B, nblk, C, dgamma, dbeta,
                               gS, accumulate);
            HNO_CHECK_LAUNCH();
        }
        ProfScope _ps2(KID_CB_GN, s, (double)B * V * C * 6.0);
        hipLaunchKernelGGL(cb_gn_bwd_apply_kernel, dim3(gsz(V * C / 8, 256, 2048), B), dim3(256), 2 * (size_t)C * sizeof(float), s, (const bf16_t *)dz, (const bf16_t *)y, mr, gamma,
                           beta, (const float *)gS, (bf16_t *)dy, C, V * C, act, dy_colsum, B, (const float *)kpart, nblk, (const float *)slab, dgamma,
                           dbeta, gS, accumulate);
        HNO_CHECK_LAUNCH();
    }
    return HNO_OK;
}

extern "C" int hno_cb_pack_input(const float *x, void *y, int B, int C, int CP, long long V, void *stream) {
    HNO_REQUIRE(x && y && B > 0 && C > 0 && CP >= C && CP % 8 == 0 && V > 0, "hno_cb_pack_input: bad argument");
    hipLaunchKernelGGL(cb_pack_input_kernel, dim3(gsz(V, 256, 4096), B), dim3(256), 0, (hipStream_t)stream, x, (bf16_t *)y, C, CP, V);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

extern "C" int hno_cb_unpack(const void *x, float *y, int B, int C, int CP, long long V, void *stream) {
    HNO_REQUIRE(x && y && B > 0 && C > 0 && CP >= C && CP % 8 == 0 && V > 0, "hno_cb_unpack: bad argument");
    hipLaunchKernelGGL(cb_unpack_kernel, dim3(gsz(V, 256, 4096), B), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)x, y, C, CP, V);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

#define CB_COLSUM_BLOCKS 256
extern "C" size_t hno_cb_colsum_workspace_bytes(int C) { return (size_t)CB_COLSUM_BLOCKS * C * sizeof(float); }

extern "C" int hno_cb_colsum(const void *g, float *out, void *workspace, int C, long long rows, void *stream) {
    HNO_REQUIRE(g && out && workspace && C > 0 && C % 8 == 0 && rows > 0, "hno_cb_colsum: bad argument");
    hipStream_t s = (hipStream_t)stream;
    int nblk = CB_COLSUM_BLOCKS;
    if (rows < nblk) nblk = (int)rows;
    hipLaunchKernelGGL(cb_colsum_kernel, dim3(nblk), dim3(256), 256 * 8 * sizeof(float), s, (const bf16_t *)g, (float *)workspace, C, rows);
    HNO_CHECK_LAUNCH();
    hipLaunchKernelGGL(cb_colsum_reduce_kernel, dim3(C / 8), dim3(256), 0, s, (const float *)workspace, nblk, C, out);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

// ---- weight gradient --------------------------------------------------------------------------------------------
namespace {
struct WgPlan {
    int TH, Sg, Sx, xrows, gpos32, xpos, nblk, slide, nseg;
    size_t lds;
};
// workgroups per channel block (= slabs the reduce kernel reads back) of a launch with ny channel blocks on gridDim.y; shared by
// wg_plan and hno_cb_wgrad_workspace_bytes so that the workspace bound follows the tuning aids.
// cfg4 step: 256 -> 8.34 ms, 384 -> 8.37, 512 -> 8.39, 128 -> 8.89
long long wg_want(int ny) {
    static const int want1 = getenv("HNO_WGRAD_WANT1") ? atoi(getenv("HNO_WGRAD_WANT1")) : 256, wantn = getenv("HNO_WGRAD_WANTN") ? atoi(getenv("HNO_WGRAD_WANTN")) : 256;   // tuning aids
    long long want = ny <= 1 ? want1 : wantn / ny;
    return want < 1 ? 1 : want;
}
// channel blocking of the weight-gradient GEMM (rows CP, columns CQ): blocks of <= 48 x 48; equal-sized blocks ride on gridDim.y
struct WgBlocks {
    int CO, CI, nco, nci;
    bool uniform;
};
WgBlocks wg_blocks(int CP, int CQ) {
    WgBlocks b;
    b.CO = CP < 48 ? CP : 48;
    b.CI = CQ < 48 ? CQ : 48;
    b.uniform = CP % b.CO == 0 && CQ % b.CI == 0;
    b.nco = (CP + b.CO - 1) / b.CO;
    b.nci = (CQ + b.CI - 1) / b.CI;
    return b;
}
// band height: as many output rows as fit in LDS (G band + X halo at the block's channel pitches), at least 1
WgPlan wg_plan(int B, int Do, int Ho, int Wo, int ks, int stride, int pad, int PG, int PX, int ny) {
    // two workgroups per CU (one stages while the other computes) when there is enough work: cap the LDS at half
    // (measured: the one-channel-block layers of the two shallow levels are faster with one big band per CU, 228 vs 390 us)
    const size_t cap = (ny == 1 || (debug_flags() & 128)) ? 150 * 1024 : 76 * 1024;
    WgPlan p = {};
    // one pitch for both images: output position p = hh * S + ww reads X-image element base(tap) + stride * p, which is
    // (row stride * hh + th, column stride * ww + tw) when the X rows have the SAME pitch S >= stride * (Wo - 1) + ks
    p.Sg = stride * (Wo - 1) + ks;
    p.Sx = p.Sg;
    // plane slots of the X image: a ring of 4 when the input planes slide (3x3x3, stride 1: three in use + the next one on its way),
    // two sets of ks otherwise (LDS-DMA double buffering, cb_wgrad_kernel); the G band is double-buffered too
    const int nslot = (ks == 3 && stride == 1) ? 4 : 2 * ks;
    int best = 1;
    for (int th = 1; th <= Ho; ++th) {
        const int xrows = stride * (th - 1) + ks;
        const int gpos32 = (th * p.Sg + 31) & ~31;
        const int xpos = nslot * xrows * p.Sx + 2 * p.Sx + 64 + stride * 32;
        const size_t bytes = ((size_t)2 * gpos32 * PG + (size_t)xpos * PX) * 2;
        if (bytes <= cap || th == 1) best = th; else break;
    }
    p.TH = best;
    p.xrows = stride * (best - 1) + ks;
    p.gpos32 = (best * p.Sg + 31) & ~31;
    p.xpos = nslot * p.xrows * p.Sx + 2 * p.Sx + 64 + stride * 32;
    p.lds = ((size_t)2 * p.gpos32 * PG + (size_t)p.xpos * PX) * 2;
    // workgroups per channel block: ~one per CU over all channel blocks.  Every workgroup writes a slab of ntaps x 48 x 48 floats
    // that the reduce kernel reads back: with 512 / ny (>= 8) workgroups the slabs of a 384 x 384 layer were 127 MB, more traffic
    // than everything else in that layer
    const long long want = wg_want(ny);
    const int nbands = (Ho + best - 1) / best;
    p.slide = (ks == 3 && stride == 1) ? 1 : 0;
    p.nseg = 1;
    long long nwork = (long long)B * Do * nbands;
    if (p.slide) {       // runs of output planes: as long as the grid allows (each run re-stages 2 planes at its start)
        long long nseg = (want + (long long)B * nbands - 1) / ((long long)B * nbands);
        if (nseg < 1) nseg = 1;
        if (nseg > Do) nseg = Do;
        p.nseg = (int)nseg;
        nwork = (long long)B * nbands * nseg;
    }
    p.nblk = (int)(nwork < want ? nwork : want);
    return p;
}
}  // namespace

extern "C" size_t hno_cb_wgrad_workspace_bytes(int Cin, int Cout, int ks) {
    // slabs [channel block][workgroup][tap][<= 48][<= 48] of ONE launch (the launches of a non-uniform blocking reuse the workspace):
    // the true upper bound of what hno_cb_wgrad writes for these channel counts (symmetric in Cin / Cout, so it holds for the
    // transposed form too); hno_cb_wgrad checks its plan against the size it is handed
    if (Cin < 1 || Cout < 1 || ks < 1) return 0;
    const WgBlocks b = wg_blocks(Cout, Cin);
    const int ny = b.uniform ? b.nco * b.nci : 1;
    return (size_t)wg_want(ny) * ny * ks * ks * ks * b.CO * b.CI * sizeof(float);
}

template <int TA, int TB, int TPW>
static int wg_launch1(const CwArgs &a, dim3 grid, size_t lds, hipStream_t s) {
    static int attr_set = -1;       // once per instantiation (not a stream operation; kept out of graph captures after warm-up)
    if (attr_set != current_device()) {
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)cb_wgrad_kernel<TA, TB, TPW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = current_device();
    }
    hipLaunchKernelGGL((cb_wgrad_kernel<TA, TB, TPW>), grid, dim3(256), lds, s, a);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}
template <int TA, int TB>
static int wg_launch(const CwArgs &a, dim3 grid, size_t lds, hipStream_t s) {
    const int tpw = (a.ntaps + 3) / 4;
    if (tpw <= 1) return wg_launch1<TA, TB, 1>(a, grid, lds, s);
    if (tpw <= 2) return wg_launch1<TA, TB, 2>(a, grid, lds, s);
    return wg_launch1<TA, TB, 7>(a, grid, lds, s);
}

// dW (fp32, the parameter's layout) of a convolution (transposed = 0: W[Cout][Cin][T]; g on the OUTPUT grid (Dg..), x = [xa | xb]
// on the input grid) or of a ConvTranspose (transposed = 1: Wt[Cin][Cout][T]; the transposed convolution's input x plays the
// role of the "output gradient" operand and g of the gathered operand -- the host swaps them).
extern "C" int hno_cb_wgrad(const void *g, int Cg, const void *xa, int Ca, const void *xb, int Cb, float *dW, void *workspace,
                            size_t workspace_bytes, int transposed, int B, int Dx, int Hx, int Wx, int Dg, int Hg, int Wg, int ks,
                            int stride, int pad, void *stream) {
    HNO_REQUIRE(g && xa && dW && workspace && B > 0 && Cg > 0 && Ca > 0 && Cb >= 0, "hno_cb_wgrad: bad argument");
    HNO_REQUIRE((stride == 1 || stride == 2) && ks >= 1 && ks <= 3, "hno_cb_wgrad: bad stride / kernel");
    if ((Cg % 8) || (Ca % 8) || (Cb % 8)) return fail(HNO_ELIMIT, "hno_cb_wgrad: channel counts must be multiples of 8");
    hipStream_t s = (hipStream_t)stream;
    // GEMM: rows = channels of the "positions" operand P (un-gathered, on the small grid), columns = channels of the gathered
    // operand Q (on the grid the taps reach into).  conv: P = g (Cout), Q = x (Cin).  transposed conv: P = x (Cin), Q = g (Cout).
    CwArgs a = {};
    int CP, CQ;
    if (!transposed) {
        a.g = (const bf16_t *)g; a.Cg = Cg; a.xa = (const bf16_t *)xa; a.xb = (const bf16_t *)xb; a.Ca = Ca; a.Cb = Cb;
        a.Do = Dg; a.Ho = Hg; a.Wo = Wg; a.Di = Dx; a.Hi = Hx; a.Wi = Wx;
        CP = Cg; CQ = Ca + Cb;
    } else {
        HNO_REQUIRE(Cb == 0, "hno_cb_wgrad: transposed convolutions take one input");
        a.g = (const bf16_t *)xa; a.Cg = Ca; a.xa = (const bf16_t *)g; a.xb = nullptr; a.Ca = Cg; a.Cb = 0;
        a.Do = Dx; a.Ho = Hx; a.Wo = Wx; a.Di = Dg; a.Hi = Hg; a.Wi = Wg;
        CP = Ca; CQ = Cg;
    }
    a.B = B; a.ks = ks; a.stride = stride; a.pad = pad; a.ntaps = ks * ks * ks; a.swap = transposed; a.dbg = debug_flags();
    const int T = a.ntaps;
    // parameter layout: conv W[Cout = P][Cin = Q][T]: out_is_axis0 = 1 with (o, i) = (P, Q); transposed Wt[Cin = P][Cout = Q][T] likewise
    const int C1 = CQ;
    // channel blocks of <= 48 x 48.  When every block has the same size (channel counts <= 48 or multiples of 48: all V-Net
    // layers) they ride on gridDim.y of ONE launch; otherwise one launch per block.
    const WgBlocks blk = wg_blocks(CP, CQ);
    const int CO = blk.CO, CI = blk.CI, nco = blk.nco, nci = blk.nci;
    const bool uniform = blk.uniform;
    for (int bi = 0; bi < (uniform ? 1 : nco * nci); ++bi) {
        a.co0 = uniform ? 0 : (bi / nci) * CO;
        a.ci0 = uniform ? 0 : (bi % nci) * CI;
        a.CO = uniform ? CO : (CP - a.co0 < CO ? CP - a.co0 : CO);
        a.CI = uniform ? CI : (CQ - a.ci0 < CI ? CQ - a.ci0 : CI);
        a.nci = nci;
        const int ny = uniform ? nco * nci : 1;
        const int TA = (a.CO + 15) / 16, TB = (a.CI + 15) / 16;
        const WgPlan p = wg_plan(B, a.Do, a.Ho, a.Wo, ks, stride, pad, TA * 16, TB * 16, ny);
        a.TH = p.TH; a.Sg = p.Sg; a.Sx = p.Sx; a.xrows = p.xrows; a.xplanes = ks; a.xpos = p.xpos; a.slide = p.slide; a.nseg = p.nseg;
        a.nbands = (a.Ho + p.TH - 1) / p.TH;
        a.slab = (float *)workspace;
        // every workgroup (blockIdx.y * gridDim.x + blockIdx.x) writes one slab of T x CO x CI floats
        HNO_REQUIRE((size_t)p.nblk * ny * T * a.CO * a.CI * sizeof(float) <= workspace_bytes,
                    "hno_cb_wgrad: workspace too small (size it with hno_cb_wgrad_workspace_bytes(Cin, Cout, ks))");
        int rc;
        {
            ProfScope _ps(KID_CB_WGRAD, s, 2.0 * B * (double)a.Do * a.Ho * a.Wo * T * a.CO * a.CI * ny);
            const dim3 grid(p.nblk, ny);
            if (TA == 1 && TB == 1) rc = wg_launch<1, 1>(a, grid, p.lds, s);
            else if (TA == 1 && TB == 2) rc = wg_launch<1, 2>(a, grid, p.lds, s);
            else if (TA == 1 && TB == 3) rc = wg_launch<1, 3>(a, grid, p.lds, s);
            else if (TA == 2 && TB == 1) rc = wg_launch<2, 1>(a, grid, p.lds, s);
            else if (TA == 2 && TB == 2) rc = wg_launch<2, 2>(a, grid, p.lds, s);
            else if (TA == 2 && TB == 3) rc = wg_launch<2, 3>(a, grid, p.lds, s);
            else if (TA == 3 && TB == 1) rc = wg_launch<3, 1>(a, grid, p.lds, s);
            else if (TA == 3 && TB == 2) rc = wg_launch<3, 2>(a, grid, p.lds, s);
            else rc = wg_launch<3, 3>(a, grid, p.lds, s);
        }
        if (rc != HNO_OK) return rc;
        static const int rows_max = getenv("HNO_WGRAD_REDUCE_ROWS") ? atoi(getenv("HNO_WGRAD_REDUCE_ROWS")) : 8;      // A/B aid: 0 = never (16 slabs: 18.4 against 14.0 us)
        if (p.nblk <= rows_max && ny * a.CO >= 256 && T <= 27 && a.CI <= 48)
            hipLaunchKernelGGL(cb_wgrad_reduce_rows_kernel, dim3(a.CO, ny), dim3(256), 0, s, (const float *)a.slab, p.nblk, T, a.CO, a.CI, a.co0, a.ci0,
                               dW, C1, T, nci);
        else
            hipLaunchKernelGGL(cb_wgrad_reduce_kernel, dim3(gsz((long long)T * a.CO * a.CI, 64, 1024), ny), dim3(256), 0, s, (const float *)a.slab, p.nblk,
                               T, a.CO, a.CI, a.co0, a.ci0, dW, C1, T, 1, nci);
        HNO_CHECK_LAUNCH();
    }
    return HNO_OK;
}
