// 1x1x1 ("pointwise") convolution with fused concat / bias / activation, forward and backward,
// and the shared-weight spectral channel mix built on it.
//
// Reference call sites: torch.cat + ConvNormAct(k=1) in HNOXSBlock.forward
// (nets/hnosegxs.py:254-255, 274-275), conv1 (:153), conv_out (:178), _OpNormAct.forward
// (nets/nets_utils.py:127-133); NeuralOperatorBlock x n_XS (nets/hnosegxs.py:307-329) with the
// einsum of HartleyOperator._call3d_notransform (nets/hartley_operator.py:287-292).
// Data stay NCDHW (voxel index fastest).
//
// HBM-bound streaming op (8 flop/B at 48->24).  The channel contraction runs on
// v_mfma_f32_32x32x2_f32 so that every global access is a 128-byte segment (32 consecutive
// voxels of one channel) and needs no LDS: the weight matrix lives in VGPRs as the A operand
// (rows = output channels), the activations stream through as the B operand (cols = voxels).
//   A operand: lane l holds A[row = l & 31][k = l >> 5]
//   B operand: lane l holds B[k = l >> 5][col = l & 31]
//   C/D      : lane l, reg r holds C[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31]
// Addressing: every access is  (wave-uniform 64-bit base: batch, channel pair)  +  (32-bit per-lane
// offset: lane-half channel + voxel, computed once per tile), which keeps the address math on
// the scalar unit and the VGPR count low enough for several waves per SIMD.
#include "hno_common.h"

namespace hno {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---- bf16 matrix-core variants (torch.autocast(bfloat16): nn.Conv3d takes bf16 operands, accumulates in fp32 and returns bf16;
// reference experiments/train_test.py:154-160).  The kernels keep their fp32 load / store structure: a lane still holds
// channel 2 ks + h of its voxel for ks = 0 .. NK-1; six consecutive ks (+ two zero slots) are ONE fragment of
// v_mfma_f32_32x32x16_bf16 (k-slot 8 h + j  <->  channel 2 (6 s + j) + h), so 12 fp32 MFMAs of 64 cycles become 2 of 32.
typedef __bf16 pw_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ pw_bf16x8 pack6(const float *v) {
    pw_bf16x8 f;
#pragma unroll
    for (int j = 0; j < 6; ++j) f[j] = (__bf16)v[j];
    f[6] = (__bf16)0.f;
    f[7] = (__bf16)0.f;
    return f;
}
__device__ __forceinline__ f32x16 mfma_bf(pw_bf16x8 a, pw_bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float bf16_round(float x) { return (float)(__bf16)x; }
// bf16 activations IN MEMORY (round 6: the tensors torch.autocast makes bf16 -- the outputs of nn.Conv3d and what follows them
// elementwise, i.e. the FNOSeg / HNOSeg block inputs and outputs, nets/architectures.py:521-546 under train_test.py:154-160)
typedef unsigned short pw_u16;
__device__ __forceinline__ float bf16_bits_to_f32(pw_u16 h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__device__ __forceinline__ pw_u16 f32_to_bf16_bits(float x) { return __builtin_bit_cast(pw_u16, (__bf16)x); }
// one LDS-DMA instruction = FOUR bf16 rows x 32 voxels (64 bytes each): lane l fetches dword (l & 15) of row (l >> 4)
__device__ __forceinline__ void dma_row_quad16(const pw_u16 *base, unsigned lane_byte_off, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_byte_off), "s"(base), "s"(lds_dst) : "memory");
}

struct PwArgs {
    const float *xa, *xb, *W, *bias;
    float *y;
    int Ca, Cb, Cin, Cout, B;
    unsigned V;
    int act;
    int k_begin, k_count;  // input-channel chunk handled by this launch
    int accumulate;        // start from the values already in y (pre-activation)
    int finalize;          // add bias and apply the activation
    int o_begin;           // first output channel of this launch (32 per launch)
    int residual;          // use W + I (frequency-domain residual of the HNO-XS mix)
    int dbg;
};

// x[channel i0 + h][v] of the virtual concat [xa ; xb] for one batch element.  i0 and nvalid
// (1 or 2 channels of the pair exist) are wave-uniform; lanes of a missing channel read a valid
// address (their weight is zero).
__device__ __forceinline__ float load_pair(const float *xa_b, const float *xb_b, int Ca, int i0, int nvalid, unsigned V,
                                           unsigned vcl, unsigned hoffV, int h) {
    const unsigned off = (nvalid == 2 ? hoffV : 0u) + vcl;
    if (i0 + nvalid <= Ca) return (xa_b + (size_t)i0 * V)[off];
    if (i0 >= Ca) return (xb_b + (size_t)(i0 - Ca) * V)[off];
    const float *p = h ? xb_b : xa_b + (size_t)i0 * V;  // pair straddles the concat boundary
    return p[vcl];
}

// one wave = one tile of 32 voxels; 4 waves per block
template <int KS_MAX>
__global__ __launch_bounds__(256, 2) void pwconv_fwd_kernel(PwArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, c = lane & 31;
    const int nks = (a.k_count + 1) / 2;
    const unsigned V = a.V;
    float w[KS_MAX];  // weights of this chunk: A[row = o][k = i]
#pragma unroll
    for (int ks = 0; ks < KS_MAX; ++ks) {
        const int i = a.k_begin + 2 * ks + h, o = a.o_begin + c;
        const bool ok = ks < nks && 2 * ks + h < a.k_count && o < a.Cout;
        w[ks] = ok ? a.W[(size_t)o * a.Cin + i] : 0.f;
        if (a.residual && ok && o == i) w[ks] += 1.f;
    }
    const unsigned tiles_per_b = (V + 31) / 32;
    const unsigned ntiles = tiles_per_b * a.B;
    const unsigned hoffV = h ? V : 0u, hoff4V = h ? 4u * V : 0u;
    for (unsigned t = blockIdx.x * 4 + wave; t < ntiles; t += gridDim.x * 4) {
        const unsigned b = t / tiles_per_b;
        const unsigned v = (t - b * tiles_per_b) * 32 + c;
        const bool vin = v < V;
        const unsigned vcl = vin ? v : 0u;
        const float *xa_b = a.xa + (size_t)b * a.Ca * V;
        const float *xb_b = a.xb ? a.xb + (size_t)b * a.Cb * V : a.xa;
        float *y_b = a.y + (size_t)b * a.Cout * V;
        float xv[KS_MAX];
#pragma unroll
        for (int ks = 0; ks < KS_MAX; ++ks) {
            xv[ks] = 0.f;
            if (ks < nks) {
                const int nvalid = (2 * ks + 1 < a.k_count) ? 2 : 1;
                xv[ks] = load_pair(xa_b, xb_b, a.Ca, a.k_begin + 2 * ks, nvalid, V, vcl, hoffV, h);
            }
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[r] = 0.f;
            if (a.accumulate) {
                const int orow = a.o_begin + (r & 3) + 8 * (r >> 2);
                if (orow + 4 * h < a.Cout) acc[r] = (y_b + (size_t)orow * V)[hoff4V + vcl];
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS_MAX; ++ks)
            if (ks < nks) acc = mfma32(w[ks], xv[ks], acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int orow = a.o_begin + (r & 3) + 8 * (r >> 2);  // uniform; this lane's row is orow + 4h
            if (orow < a.Cout) {
                float val = acc[r];
                if (a.finalize) {
                    if (a.bias) val += a.bias[orow + 4 * h < a.Cout ? orow + 4 * h : orow];
                    val = act_apply(val, a.act);
                }
                if (vin && orow + 4 * h < a.Cout) (y_b + (size_t)orow * V)[hoff4V + v] = val;
            }
        }
    }
}

// ------------------------------------------------------------------------------ backward
// One wave = 32 voxels.  g = gy * act'(y) is loaded once in B-operand layout
// (2 channels x 32 voxels per register), used for
//   dgrad: gx[i][v] = sum_o W[o][i] g[o][v]        (32x32x2 MFMA, A = W^T in VGPRs)
//   wgrad: dW[o][i] += sum_v g[o][v] x[i][v]       (16x16x4 MFMA from a wave-private LDS tile)
//   dbias: per-lane partial sums, reduced once at the end.
// Limits of this kernel: Cout <= 32, Cin <= 64 (the C wrapper checks).
struct PwBwdArgs {
    const float *gy, *y, *xa, *xb, *W;
    float *gxa, *gxb, *dW, *dbias;
    float *partials;  // [gridDim.x][Cout*Cin + Cout] per-block slabs
    int Ca, Cb, Cin, Cout, B;
    unsigned V;
    int act;
    int residual;
    int dbg;
    int accum;   // gxa / gxb += instead of = (gradient accumulation fused into the store)
    // generic kernel only: element strides between batch items and between weight rows, so that channel
    // sub-ranges of larger tensors can be processed in place (chunked launches for wide layers)
    long long s_gy, s_xa, s_xb, s_gxa, s_gxb;
    int ldw;
    int xa_act;  // gxa *= act'(xa): xa is itself the OUTPUT of that activation (fuses PadInverse's SELU backward)
    // fused conv branch (fast kernel, BR = 1): xa = act(s + Wbr xb + bbr) came from a second 1x1x1 conv of xb.  With
    // p = gxa * act'(xa) (the xa_act product): gxb += Wbr^T p, dWbr = sum p xb^T, dbbr = sum p -- one pass instead of two
    const float *Wbr;
};

#define PWB_LD 34  // 32 voxels + 2: row stride == 2 (mod 4) -> conflict-free 16x16x4 operand reads
#define PWB_XP 68  // floats between DMA'd row pairs of the fast backward kernel's x ring (see there)

template <int KSO_MAX, int ICH>  // KSO_MAX >= ceil(Cout/2), ICH = number of 32-wide input-channel chunks
__global__ __launch_bounds__(256, 2) void pwconv_bwd_kernel(PwBwdArgs a) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, c = lane & 31;
    const int nkso = (a.Cout + 1) / 2;
    const unsigned V = a.V;
    constexpr int rowsG = 32, rowsX = ICH * 32;
    float *G = lds + (size_t)wave * (rowsG + rowsX) * PWB_LD;  // [o][v]
    float *X = G + rowsG * PWB_LD;                              // [i][v]
    // A operand of dgrad: Wt[row = i][k = o] = W[o][i]
    float wt[ICH][KSO_MAX];
#pragma unroll
    for (int ic = 0; ic < ICH; ++ic)
#pragma unroll
        for (int ks = 0; ks < KSO_MAX; ++ks) {
            const int i = ic * 32 + c, o = 2 * ks + h;
            const bool ok = i < a.Cin && o < a.Cout;
            wt[ic][ks] = ok ? a.W[(size_t)o * a.ldw + i] : 0.f;
            if (a.residual && ok && i == o) wt[ic][ks] += 1.f;
        }
    // zero the wave's LDS tile once: rows beyond Cout / Cin must stay finite and zero
    for (int i = lane; i < (rowsG + rowsX) * PWB_LD; i += 64) G[i] = 0.f;
    float db[KSO_MAX];
#pragma unroll
    for (int ks = 0; ks < KSO_MAX; ++ks) db[ks] = 0.f;
    // wgrad accumulators: M tiles (o) x N tiles (i) of 16x16
    constexpr int MT = 2, NTI = ICH * 2;
    f32x4 dw[MT][NTI];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTI; ++n) dw[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const unsigned tiles_per_b = (V + 31) / 32;
    const unsigned ntiles = tiles_per_b * a.B;
    const unsigned ngroups = (ntiles + 3) / 4;
    const unsigned hoffV = h ? V : 0u, hoff4V = h ? 4u * V : 0u;
    for (unsigned grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const unsigned t = grp * 4 + wave;
        const bool live = t < ntiles;
        const unsigned b = live ? t / tiles_per_b : 0u;
        const unsigned v = live ? (t - b * tiles_per_b) * 32 + c : 0u;
        const bool vin = live && v < V;
        const unsigned vcl = vin ? v : 0u;
        const float *gy_b = a.gy + (size_t)b * a.s_gy, *y_b = a.y + (size_t)b * a.s_gy;
        const float *xa_b = a.xa + (size_t)b * a.s_xa;
        const float *xb_b = a.xb ? a.xb + (size_t)b * a.s_xb : a.xa;
        // ---- g = gy * act'(y), B-operand layout; also staged to LDS as G[o][v]
        float g[KSO_MAX];
#pragma unroll
        for (int ks = 0; ks < KSO_MAX; ++ks) {
            g[ks] = 0.f;
            if (ks < nkso) {
                const int o0 = 2 * ks;
                const bool two = o0 + 1 < a.Cout;
                const unsigned off = (two ? hoffV : 0u) + vcl;
                float gv = (gy_b + (size_t)o0 * V)[off];
                if (a.act != HNO_ACT_NONE) gv *= act_grad_from_out((y_b + (size_t)o0 * V)[off], a.act);
                g[ks] = (vin && o0 + h < a.Cout) ? gv : 0.f;
                G[(o0 + h) * PWB_LD + c] = g[ks];
                db[ks] += g[ks];
            }
        }
        // ---- x tile -> LDS X[i][v]  (partial unroll: bounds the loads in flight / VGPRs)
#pragma unroll 8
        for (int j = 0; j < ICH * 16; ++j) {
            const int i0 = 2 * j;
            if (i0 < a.Cin) {
                const int nvalid = i0 + 1 < a.Cin ? 2 : 1;
                const float xv = load_pair(xa_b, xb_b, a.Ca, i0, nvalid, V, vcl, hoffV, h);
                if (i0 + h < a.Cin) X[(i0 + h) * PWB_LD + c] = vin ? xv : 0.f;
            }
        }
        // ---- dgrad
#pragma unroll
        for (int ic = 0; ic < ICH; ++ic) {
            if (ic * 32 >= a.Cin) continue;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KSO_MAX; ++ks)
                if (ks < nkso) acc = mfma32(wt[ic][ks], g[ks], acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int irow = ic * 32 + (r & 3) + 8 * (r >> 2);  // uniform; lane's channel is irow + 4h
                if (irow < a.Cin) {
                    const int i = irow + 4 * h;
                    const bool ok = vin && i < a.Cin;
                    if (irow + 4 < a.Ca || irow + 4 >= a.Cin) {        // both lane halves in xa (or half 1 absent)
                        if (irow < a.Ca) {
                            if (a.gxa && ok) { float *q_ = a.gxa + (size_t)b * a.s_gxa + (size_t)irow * V + hoff4V + v; *q_ = (acc[r] * (a.xa_act != HNO_ACT_NONE ? act_grad_from_out(X[i * PWB_LD + c], a.xa_act) : 1.f)) + (a.accum ? *q_ : 0.f); }
                        } else if (a.gxb && ok) {
                            { float *q_ = a.gxb + (size_t)b * a.s_gxb + (size_t)(irow - a.Ca) * V + hoff4V + v; *q_ = acc[r] + (a.accum ? *q_ : 0.f); }
                        }
                    } else if (irow >= a.Ca) {                          // both halves in xb
                        if (a.gxb && ok) { float *q_ = a.gxb + (size_t)b * a.s_gxb + (size_t)(irow - a.Ca) * V + hoff4V + v; *q_ = acc[r] + (a.accum ? *q_ : 0.f); }
                    } else if (ok) {                                    // halves straddle the concat boundary
                        if (i < a.Ca) {
                            if (a.gxa) { float *q_ = a.gxa + (size_t)b * a.s_gxa + (size_t)i * V + v; *q_ = (acc[r] * (a.xa_act != HNO_ACT_NONE ? act_grad_from_out(X[i * PWB_LD + c], a.xa_act) : 1.f)) + (a.accum ? *q_ : 0.f); }
                        } else if (a.gxb) {
                            { float *q_ = a.gxb + (size_t)b * a.s_gxb + (size_t)(i - a.Ca) * V + v; *q_ = acc[r] + (a.accum ? *q_ : 0.f); }
                        }
                    }
                }
            }
        }
        __syncthreads();  // LDS tile written (wave-private, but keep the ordering explicit)
        // ---- wgrad: dw[m][n] += G[m-tile rows][v] * X[n-tile rows][v]^T  (K = 32 voxels = 8 steps)
        // k-step outermost: 2 + NTI operand registers live at a time
        {
            const float *ga = G + (lane & 15) * PWB_LD + (lane >> 4);
            const float *xb = X + (lane & 15) * PWB_LD + (lane >> 4);
#pragma unroll 2
            for (int ks = 0; ks < 8; ++ks) {
                float av[MT], bv[NTI];
#pragma unroll
                for (int m = 0; m < MT; ++m) av[m] = ga[m * 16 * PWB_LD + ks * 4];
#pragma unroll
                for (int n = 0; n < NTI; ++n) bv[n] = xb[n * 16 * PWB_LD + ks * 4];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NTI; ++n) dw[m][n] = mfma16(av[m], bv[n], dw[m][n]);
            }
        }
        __syncthreads();  // before the next iteration overwrites the tile
    }
    // ---- flush: reduce the 4 waves through LDS, one slab per block (no atomics)
    {
        const int n = a.Cout * a.Cin + a.Cout;
        __syncthreads();
        float *mine = lds + (size_t)wave * n;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nn = 0; nn < NTI; ++nn)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = m * 16 + (lane >> 4) * 4 + r, i = nn * 16 + (lane & 15);
                    if (o < a.Cout && i < a.Cin) mine[o * a.Cin + i] = dw[m][nn][r];
                }
#pragma unroll
        for (int ks = 0; ks < KSO_MAX; ++ks) {
            float s = db[ks];
            for (int off = 16; off >= 1; off >>= 1) s += __shfl_xor(s, off);  // within each 32-lane half
            const int o = 2 * ks + h;
            if (c == 0 && ks < nkso && o < a.Cout) mine[a.Cout * a.Cin + o] = s;
        }
        block_sum_to_slab(lds, n, a.partials + (size_t)blockIdx.x * n, threadIdx.x);
    }
}

// ------------------------------------------------------------------------------ fast paths
// Exact-size specialisations (no runtime channel guards, hence no branchy code and ~100 VGPRs):
// Ca % 8 == 0, Cb % 8 == 0, CIN = 2*NKI, COUT = 2*NKO (<= 32), whole problem in one launch.
// Tuning notes (tools/pwbench.hip, 48->24 at 2 x 65^3): this streaming shape runs best with FEW waves
// per CU (one 512-thread block per CU, grid = number of CUs, grid-stride over tiles): 37 us against
// 42 us at 16 waves/CU and 46-50 us with one tile per wave; 8/16-byte-per-lane loads, explicit
// software prefetch, XCD-contiguous tile order and non-temporal stores made no difference or hurt.
// waves per workgroup of the 24 + 24 -> 24 forward kernel (one workgroup per CU).  Round 3, with channel-padded (128-byte aligned) rows:
// 8 waves = two per SIMD, so one wave's bias + SELU epilogue (~1 200 VALU cycles per tile) runs under the other's MFMA chain
// (they do not overlap within a wave, DESIGN lesson 34): 39.0 -> 36.4 us isolated, 2.686 -> 2.667 ms per step; 12 waves: the same.
// With the misaligned rows of rounds 1-2 four waves were best.  HNO_PWF_WAVES=4 / 12 select the others (A/B).
// HNO_PW_STAGGER=n (A/B, round 6): odd wave slots of a SIMD start n x 1024 cycles late (stagger_start)
static int pw_stagger() {
    static const int v = getenv("HNO_PW_STAGGER") ? atoi(getenv("HNO_PW_STAGGER")) : 0;
    return v < 0 ? 0 : (v > 63 ? 63 : v);
}
static int pwf_waves() {
    static const int v = getenv("HNO_PWF_WAVES") ? atoi(getenv("HNO_PWF_WAVES")) : 8;
    return (v == 4 || v == 12) ? v : 8;      // only the built instantiations
}
#define PWF_FAST_WAVES 8
#define PWF_DMA_WAVES 4      // fast forward kernel: 4 waves (one per SIMD), each with a private two-slot LDS ring

template <int CA, int CB, int COUT, bool BF16 = false, int NWV = PWF_DMA_WAVES>   // channel counts of xa / xb (both even) are compile-time: all address selects fold
__global__ __launch_bounds__(64 * NWV) void pwconv_fwd_fast_kernel(PwArgs a) {
    stagger_start((a.dbg >> 24) & 63);
    constexpr int NW = NWV;
    extern __shared__ float pwf_ring[];      // NW x 2 slots x NKI x 64 floats
    constexpr int NKI = (CA + CB) / 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, c = lane & 31;
    constexpr int CIN = 2 * NKI;
    constexpr int OT = (COUT + 31) / 32;          // output tiles of 32 rows (COUT <= 64)
    constexpr int CL = COUT - 32 * (OT - 1);      // rows of the last tile
    constexpr int NRL = CL > 24 ? 16 : (CL > 16 ? 12 : (CL > 8 ? 8 : 4));    // accumulator registers holding rows < COUT (register r: rows (r & 3) + 8 (r >> 2) + 4 h)
    const unsigned V = a.V;
    float w[OT][NKI];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int ks = 0; ks < NKI; ++ks) {
            const int i = 2 * ks + h, o = ot * 32 + c;
            w[ot][ks] = o < COUT ? a.W[(size_t)o * CIN + i] : 0.f;
            if (a.residual && o == i) w[ot][ks] += 1.f;
        }
    static_assert(!BF16 || NKI % 6 == 0, "bf16 fragments take six channel pairs");
    pw_bf16x8 wf[OT][BF16 ? NKI / 6 : 1];
    if constexpr (BF16) {
#pragma unroll
        for (int ot = 0; ot < OT; ++ot)
#pragma unroll
            for (int s6 = 0; s6 < NKI / 6; ++s6) wf[ot][s6] = pack6(&w[ot][6 * s6]);
    }
    float bias_r[OT][16];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            bias_r[ot][r] = (a.bias && o < COUT) ? a.bias[o] : 0.f;
        }
    // branch-free activation: act(x) = (x > 0 or linear) ? ap * x : aq * (e^x - 1)
    const float ap = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE : 1.f;
    const float aq = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE * HNO_SELU_ALPHA : 1.f;
    const bool lin = a.act == HNO_ACT_NONE;
    const unsigned tiles_per_b = (V + 31) / 32;
    const unsigned ntiles = tiles_per_b * a.B;
    const unsigned hoffV = h ? V : 0u, hoff4V = h ? 4u * V : 0u;
    // The input rows arrive by LDS-DMA, one tile ahead of the products: slot layout [k-step][h][32 voxels] is exactly the MFMA B operand
    // order (lane (c, h) reads its own 4 bytes), so the B operands are conflict-free ds_read_b32.  Loads never wait behind the MFMA chain
    // and the stores of tile t overlap the loads of tile t + 1 (DESIGN.md lesson 27).
    float *ring = pwf_ring + wave * (2 * NKI * 64);
    const unsigned ring_b = (unsigned)(size_t)ring;      // LDS byte address
    const unsigned stride = gridDim.x * NW;
    auto issue = [&](unsigned t, int slot) {
        const unsigned b = t / tiles_per_b;
        const unsigned v = (t - b * tiles_per_b) * 32 + c;
        const unsigned boff = (hoffV + (v < V ? v : 0u)) * 4u;     // V * 8 < 2^32 (host check)
        const float *xa_b = a.xa + (size_t)b * CA * V;
        const float *xb_b = CB > 0 ? a.xb + (size_t)b * CB * V : a.xa;
#pragma unroll
        for (int ks = 0; ks < NKI; ++ks) {
            const int i0 = 2 * ks;
            const float *base = i0 < CA ? xa_b + (size_t)i0 * V : xb_b + (size_t)(i0 - CA) * V;
            dma_row_pair(base, boff, __builtin_amdgcn_readfirstlane(ring_b + (slot * NKI + ks) * 256));
        }
    };
    unsigned t = blockIdx.x * NW + wave;
    if (t < ntiles) issue(t, 0);
    for (int slot = 0; t < ntiles; t += stride, slot ^= 1) {
        static_assert(NKI <= 63, "the DMA of one tile must fit the vmcnt counter");
        if (t + stride < ntiles) {
            issue(t + stride, slot ^ 1);
            dma_wait<NKI>();      // only the next tile's DMA stays in flight (counting the previous tile's stores too would be wrong on the first pass)
        } else {
            dma_wait<0>();
        }
        const unsigned b = t / tiles_per_b;
        const unsigned v0 = (t - b * tiles_per_b) * 32, v = v0 + c;
        const bool vin = v < V, full = v0 + 32 <= V;      // `full` is wave-uniform
        const float *sl = ring + slot * NKI * 64 + lane;
        float xv[NKI];
#pragma unroll
        for (int ks = 0; ks < NKI; ++ks) xv[ks] = sl[ks * 64];
        float *y_b = a.y + (size_t)b * COUT * V;
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if (a.dbg & 1) {
#pragma unroll
                for (int ks = 0; ks < NKI; ++ks) acc[ks & 15] += xv[ks];
            } else if constexpr (BF16) {
#pragma unroll
                for (int s6 = 0; s6 < NKI / 6; ++s6) acc = mfma_bf(wf[ot][s6], pack6(&xv[6 * s6]), acc);
            } else {
#pragma unroll
                for (int ks = 0; ks < NKI; ++ks) acc = mfma32(w[ot][ks], xv[ks], acc);
            }
            if (!(a.dbg & 2)) {
                // epilogue without divergent control flow: the activation is computed for every lane and selected (a branch per output
                // register costs more than the ~12 instructions it skips), and whole tiles (all but the last of a batch) store unpredicated
                float val[16];
#pragma unroll
                for (int r = 0; r < (ot == OT - 1 ? NRL : 16); ++r) {
                    float x = acc[r] + bias_r[ot][r];
                    if constexpr (BF16) x = bf16_round(x);      // the convolution's output dtype under autocast
                    val[r] = x;
                }
                if (!lin) {
                    if (ot == OT - 1) selu_like_regs<NRL>(val, ap, aq);
                    else selu_like_regs<16>(val, ap, aq);
                }
                float *y_l = y_b + (hoff4V + v);
                if (full) {
#pragma unroll
                    for (int r = 0; r < (ot == OT - 1 ? NRL : 16); ++r) {
                        const int orow = ot * 32 + (r & 3) + 8 * (r >> 2);
                        const float o = BF16 ? bf16_round(val[r]) : val[r];
                        if (orow + 4 < COUT) y_l[(size_t)orow * V] = o;
                        else if (h == 0) y_l[(size_t)orow * V] = o;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < (ot == OT - 1 ? NRL : 16); ++r) {
                        const int orow = ot * 32 + (r & 3) + 8 * (r >> 2);
                        const float o = BF16 ? bf16_round(val[r]) : val[r];
                        if (vin && (orow + 4 < COUT || h == 0)) y_l[(size_t)orow * V] = o;
                    }
                }
            } else {
                float sacc = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[r];
                if (sacc == 12345.678f) y_b[0] = sacc;
            }
        }
    }
}

// ---- forward of the FNOSeg / HNOSeg block tail in one pass (nets/architectures.py:521-546) -------------------
//   y   = act(s + Wbr x + bbr)            s: operator output (inverse transform), x: block input, Wbr: conv_branch
//   out = act(Wc [y ; x] + bc)            conv_concat over the virtual concat
// The first product's accumulator rows are the second product's B operand: k-slot (r, h) of the y part carries channel
// (r & 3) + 8 (r >> 2) + 4 h, and the y columns of Wc are loaded in that order, so y never leaves registers between the
// two products.  Reads s, x; writes y (saved for the backward) and out: 4 streams instead of 8 for three separate kernels.
struct PwBranchArgs {
    const float *s, *x, *Wbr, *bbr, *W, *bias;
    float *y, *out;
    int B;
    unsigned V;
    int act;
};

// IO16: x and out are bf16 IN MEMORY (what autocast makes of a block's input and output; s and y stay fp32 as in the reference); needs
// BF16 arithmetic and V % 32 == 0 (channel-padded rows).  x arrives four rows per DMA instruction (dma_row_quad16).
template <int CA, int CB, int COUT, bool BF16 = false, bool IO16 = false>
__global__ __launch_bounds__(64 * PWF_DMA_WAVES) void pwconv_fwd_branch_kernel(PwBranchArgs a) {
    static_assert(CA % 8 == 0 && CA <= 32 && CB % 8 == 0 && COUT <= 32 && COUT % 8 == 0, "one 32-row tile per product");
    static_assert(!IO16 || BF16, "bf16 tensors only under bf16 arithmetic");
    constexpr int NW = PWF_DMA_WAVES;
    extern __shared__ float pwf_ring[];      // NW x 2 slots x (NKX + RA) x 64 floats: see pwconv_fwd_fast_kernel
    constexpr int RA = CA / 2, RO = COUT / 2, NKX = CB / 2, CIN = CA + CB;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, c = lane & 31;
    const unsigned V = a.V;
    float wbr[NKX], wcx[NKX], wcy[RA], bb[RA], bc[RO];
#pragma unroll
    for (int ks = 0; ks < NKX; ++ks) {
        wbr[ks] = c < CA ? a.Wbr[(size_t)c * CB + 2 * ks + h] : 0.f;
        wcx[ks] = c < COUT ? a.W[(size_t)c * CIN + CA + 2 * ks + h] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < RA; ++r) {
        const int ch = (r & 3) + 8 * (r >> 2) + 4 * h;
        wcy[r] = c < COUT ? a.W[(size_t)c * CIN + ch] : 0.f;
        bb[r] = a.bbr ? a.bbr[ch] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < RO; ++r) bc[r] = a.bias ? a.bias[(r & 3) + 8 * (r >> 2) + 4 * h] : 0.f;
    static_assert(!BF16 || (NKX % 6 == 0 && RA % 6 == 0), "bf16 fragments take six k-slots");
    pw_bf16x8 fbr[BF16 ? NKX / 6 : 1], fcx[BF16 ? NKX / 6 : 1], fcy[BF16 ? RA / 6 : 1];
    if constexpr (BF16) {
#pragma unroll
        for (int s6 = 0; s6 < NKX / 6; ++s6) { fbr[s6] = pack6(&wbr[6 * s6]); fcx[s6] = pack6(&wcx[6 * s6]); }
#pragma unroll
        for (int s6 = 0; s6 < RA / 6; ++s6) fcy[s6] = pack6(&wcy[6 * s6]);
    }
    const float ap = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE : 1.f;
    const float aq = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE * HNO_SELU_ALPHA : 1.f;
    const bool lin = a.act == HNO_ACT_NONE;
    const unsigned tiles_per_b = (V + 31) / 32;
    const unsigned ntiles = tiles_per_b * a.B;
    const unsigned hoffV = h ? V : 0u, hoff4V = h ? 4u * V : 0u;
    // both inputs arrive by LDS-DMA one tile ahead (pwconv_fwd_fast_kernel): x as B-operand row pairs (2 ks, 2 ks + 1), s in the
    // accumulator layout (rows r' and r' + 4 per instruction); a lane reads back its own 4 bytes
    constexpr int NXD = IO16 ? NKX / 2 : NKX;         // DMA instructions of the x rows: bf16 rows go four to an instruction
    constexpr int NDMA = NXD + RA;
    static_assert(NDMA <= 63, "the DMA of one tile must fit the vmcnt counter");
    static_assert(!IO16 || NKX % 2 == 0, "four bf16 rows per DMA instruction");
    float *ring = pwf_ring + wave * (2 * NDMA * 64);
    const unsigned ring_b = (unsigned)(size_t)ring;
    const unsigned stride = gridDim.x * NW;
    auto issue = [&](unsigned t, int slot) {
        const unsigned b = t / tiles_per_b;
        const unsigned v = (t - b * tiles_per_b) * 32 + c;
        const unsigned vc = v < V ? v : 0u;
        const float *s_b = a.s + (size_t)b * CA * V;
        if constexpr (IO16) {
            const pw_u16 *x_b = reinterpret_cast<const pw_u16 *>(a.x) + (size_t)b * CB * V;
            const unsigned v0 = (t - b * tiles_per_b) * 32;
            const unsigned qoff = ((unsigned)(lane >> 4) * V + v0) * 2u + (unsigned)(lane & 15) * 4u;
#pragma unroll
            for (int j = 0; j < NXD; ++j)
                dma_row_quad16(x_b + (size_t)(4 * j) * V, qoff, __builtin_amdgcn_readfirstlane(ring_b + (slot * NDMA + j) * 256));
        } else {
            const float *x_b = a.x + (size_t)b * CB * V;
#pragma unroll
            for (int ks = 0; ks < NKX; ++ks)
                dma_row_pair(x_b + (size_t)(2 * ks) * V, (hoffV + vc) * 4u, __builtin_amdgcn_readfirstlane(ring_b + (slot * NDMA + ks) * 256));
        }
#pragma unroll
        for (int r = 0; r < RA; ++r)
            dma_row_pair(s_b + (size_t)((r & 3) + 8 * (r >> 2)) * V, (hoff4V + vc) * 4u,
                         __builtin_amdgcn_readfirstlane(ring_b + (slot * NDMA + NXD + r) * 256));
    };
    unsigned t = blockIdx.x * NW + wave;
    if (t < ntiles) issue(t, 0);
    for (int slot = 0; t < ntiles; t += stride, slot ^= 1) {
        if (t + stride < ntiles) {
            issue(t + stride, slot ^ 1);
            dma_wait<NDMA>();
        } else {
            dma_wait<0>();
        }
        const unsigned b = t / tiles_per_b;
        const unsigned v0 = (t - b * tiles_per_b) * 32, v = v0 + c;
        const bool vin = v < V, full = v0 + 32 <= V;
        const float *sl = ring + slot * NDMA * 64 + lane;
        float xv[NKX], sv[RA];
        if constexpr (IO16) {
            // channel 2 ks + h of voxel c: DMA instruction ks >> 1, row 2 (ks & 1) + h of its four, halfword c
            const pw_u16 *sx = reinterpret_cast<const pw_u16 *>(ring + slot * NDMA * 64) + h * 32 + c;
#pragma unroll
            for (int ks = 0; ks < NKX; ++ks) xv[ks] = bf16_bits_to_f32(sx[(ks >> 1) * 128 + (ks & 1) * 64]);
        } else {
#pragma unroll
            for (int ks = 0; ks < NKX; ++ks) xv[ks] = sl[ks * 64];
        }
#pragma unroll
        for (int r = 0; r < RA; ++r) sv[r] = sl[(NXD + r) * 64];
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        pw_bf16x8 xf[BF16 ? NKX / 6 : 1];
        if constexpr (BF16) {
#pragma unroll
            for (int s6 = 0; s6 < NKX / 6; ++s6) { xf[s6] = pack6(&xv[6 * s6]); acc = mfma_bf(fbr[s6], xf[s6], acc); }
        } else {
#pragma unroll
            for (int ks = 0; ks < NKX; ++ks) acc = mfma32(wbr[ks], xv[ks], acc);
        }
        float *y_l = a.y + (size_t)b * CA * V + (hoff4V + v);
        float yv[RA];
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            // autocast: the branch convolution returns bf16 (bias included), the sum with the fp32 operator output and the
            // activation are fp32 (nets/architectures.py:521-539)
            yv[r] = BF16 ? bf16_round(acc[r] + bb[r]) + sv[r] : acc[r] + sv[r] + bb[r];
        }
        if (!lin) {             // no branch per register (see pwconv_fwd_fast_kernel)
            selu_like_regs<RA>(yv, ap, aq);
        }
        if (full) {
#pragma unroll
            for (int r = 0; r < RA; ++r) y_l[(size_t)((r & 3) + 8 * (r >> 2)) * V] = yv[r];
        } else {
#pragma unroll
            for (int r = 0; r < RA; ++r)
                if (vin) y_l[(size_t)((r & 3) + 8 * (r >> 2)) * V] = yv[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if constexpr (BF16) {
#pragma unroll
            for (int s6 = 0; s6 < RA / 6; ++s6) acc = mfma_bf(fcy[s6], pack6(&yv[6 * s6]), acc);
#pragma unroll
            for (int s6 = 0; s6 < NKX / 6; ++s6) acc = mfma_bf(fcx[s6], xf[s6], acc);
        } else {
#pragma unroll
            for (int r = 0; r < RA; ++r) acc = mfma32(wcy[r], yv[r], acc);
#pragma unroll
            for (int ks = 0; ks < NKX; ++ks) acc = mfma32(wcx[ks], xv[ks], acc);
        }
        float *o_l = a.out + (size_t)b * COUT * V + (hoff4V + v);
        float ov[RO];
#pragma unroll
        for (int r = 0; r < RO; ++r) {
            ov[r] = acc[r] + bc[r];
            if constexpr (BF16) ov[r] = bf16_round(ov[r]);
        }
        if (!lin) {
            selu_like_regs<RO>(ov, ap, aq);
        }
        if constexpr (IO16) {
            pw_u16 *o16 = reinterpret_cast<pw_u16 *>(a.out) + (size_t)b * COUT * V + (hoff4V + v);
#pragma unroll
            for (int r = 0; r < RO; ++r)
                if (full || vin) o16[(size_t)((r & 3) + 8 * (r >> 2)) * V] = f32_to_bf16_bits(ov[r]);
        } else {
#pragma unroll
            for (int r = 0; r < RO; ++r) {
                const float val = BF16 ? bf16_round(ov[r]) : ov[r];
                if (full) o_l[(size_t)((r & 3) + 8 * (r >> 2)) * V] = val;
                else if (vin) o_l[(size_t)((r & 3) + 8 * (r >> 2)) * V] = val;
            }
        }
    }
}

// ---- two chained pointwise layers in one pass (round 4): the concat convolution that ends HNO-XS block i and the mapping convolution
// that opens block i + 1 (decoder blocks, nets/hnosegxs.py:253-279 with the U-Net concatenation of :161-162):
//   xi = act(Wc [u ; t] + bc)         u: the block's transformed branch (PadInverse output), t: the block's input
//   xn = act(Wm [xi ; k] + bm)        k: the U-Net skip tensor of block i + 1
// xi is written (the backward of both layers needs it) but not read back: the first product's accumulator registers are the B
// operand of the second product's xi part (k-slot r of lane half h = channel (r & 3) + 8 (r >> 2) + 4 h, matched by the order in which
// the xi columns of Wm are loaded).  Reads u, t, k; writes xi, xn: 5 streams of 24 channels instead of 6 for the two layers apart
// (pwconv_fwd_fast_kernel twice: 2 x (48 in + 24 out)).  Same DMA ring as pwconv_fwd_fast_kernel, three tensors per tile.
struct PwChainArgs {
    const float *u, *t, *k, *Wc, *bc, *Wm, *bm;
    float *xi, *xn;
    int B;
    unsigned V;
    int act, act2;
    int dbg;
};

// C2 / HASK: the second layer has C2 output channels and (HASK) a second input k of C channels.  <24, 24, true>: the next block's
// mapping_conv; <24, 4, false>: the model's conv_out (24 -> out_channels, no bias, no activation: a.act2) behind the LAST block.
template <int C, int NWV, int C2 = C, bool HASK = true>
__global__ __launch_bounds__(64 * NWV) void pwconv_fwd_chain_kernel(PwChainArgs a) {
    stagger_start((a.dbg >> 24) & 63);
    static_assert(C % 8 == 0 && C <= 32 && C2 <= 32 && (C2 % 8 == 0 || C2 == 4), "one 32-row tile per product");
    constexpr int NW = NWV, NK = C / 2, CIN = 2 * C, NDMA = (HASK ? 3 : 2) * NK, CIN2 = HASK ? 2 * C : C;
    constexpr int NR2 = C2 >= 8 ? C2 / 2 : C2;        // accumulator registers of the second product that hold rows < C2 (C2 = 4: rows 0..3, h = 0 lanes)
    extern __shared__ float pwf_ring[];      // NW x 2 slots x NDMA x 64 floats
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, c = lane & 31;
    const unsigned V = a.V;
    // A operands (row c = output channel): first product over [u ; t] in B-operand row pairs (channels 2 ks + h); second product: the xi
    // part in accumulator-register order, the k part in row pairs
    float wcu[NK], wct[NK], wmx[NK], wmk[NK], b1[NK], b2[NK];
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        wcu[ks] = c < C ? a.Wc[(size_t)c * CIN + 2 * ks + h] : 0.f;
        wct[ks] = c < C ? a.Wc[(size_t)c * CIN + C + 2 * ks + h] : 0.f;
        wmk[ks] = (HASK && c < C2) ? a.Wm[(size_t)c * CIN2 + C + 2 * ks + h] : 0.f;
        const int ch = (ks & 3) + 8 * (ks >> 2) + 4 * h;
        wmx[ks] = c < C2 ? a.Wm[(size_t)c * CIN2 + ch] : 0.f;
        b1[ks] = a.bc ? a.bc[ch] : 0.f;
        b2[ks] = (a.bm && ch < C2) ? a.bm[ch] : 0.f;
    }
    const float ap = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE : 1.f;
    const float aq = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE * HNO_SELU_ALPHA : 1.f;
    const bool lin = a.act == HNO_ACT_NONE;
    const float ap2 = a.act2 == HNO_ACT_SELU ? HNO_SELU_SCALE : 1.f, aq2 = a.act2 == HNO_ACT_SELU ? HNO_SELU_SCALE * HNO_SELU_ALPHA : 1.f;
    const bool lin2 = a.act2 == HNO_ACT_NONE;
    const unsigned tiles_per_b = (V + 31) / 32;
    const unsigned ntiles = tiles_per_b * a.B;
    const unsigned hoffV = h ? V : 0u, hoff4V = h ? 4u * V : 0u;
    static_assert(NDMA <= 63, "the DMA of one tile must fit the vmcnt counter");
    float *ring = pwf_ring + wave * (2 * NDMA * 64);
    const unsigned ring_b = (unsigned)(size_t)ring;
    const unsigned stride = gridDim.x * NW;
    auto issue = [&](unsigned t, int slot) {
        const unsigned b = t / tiles_per_b;
        const unsigned v = (t - b * tiles_per_b) * 32 + c;
        const unsigned boff = (hoffV + (v < V ? v : 0u)) * 4u;
        const float *u_b = a.u + (size_t)b * C * V, *t_b = a.t + (size_t)b * C * V, *k_b = HASK ? a.k + (size_t)b * C * V : a.u;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) dma_row_pair(u_b + (size_t)(2 * ks) * V, boff, __builtin_amdgcn_readfirstlane(ring_b + (slot * NDMA + ks) * 256));
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) dma_row_pair(t_b + (size_t)(2 * ks) * V, boff, __builtin_amdgcn_readfirstlane(ring_b + (slot * NDMA + NK + ks) * 256));
        if constexpr (HASK) {
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) dma_row_pair(k_b + (size_t)(2 * ks) * V, boff, __builtin_amdgcn_readfirstlane(ring_b + (slot * NDMA + 2 * NK + ks) * 256));
        }
    };
    unsigned t = blockIdx.x * NW + wave;
    if (t < ntiles) issue(t, 0);
    for (int slot = 0; t < ntiles; t += stride, slot ^= 1) {
        if (t + stride < ntiles) {
            issue(t + stride, slot ^ 1);
            dma_wait<NDMA>();
        } else {
            dma_wait<0>();
        }
        const unsigned b = t / tiles_per_b;
        const unsigned v0 = (t - b * tiles_per_b) * 32, v = v0 + c;
        const bool vin = v < V, full = v0 + 32 <= V;      // `full` is wave-uniform
        const float *sl = ring + slot * NDMA * 64 + lane;
        float uv[NK], tv[NK], kv[NK];
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            uv[ks] = sl[ks * 64];
            tv[ks] = sl[(NK + ks) * 64];
            kv[ks] = HASK ? sl[(2 * NK + ks) * 64] : 0.f;
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) acc = mfma32(wcu[ks], uv[ks], acc);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) acc = mfma32(wct[ks], tv[ks], acc);
        float xi[NK];
#pragma unroll
        for (int r = 0; r < NK; ++r) xi[r] = acc[r] + b1[r];
        if (!lin) {             // no branch per register (see pwconv_fwd_fast_kernel)
            selu_like_regs<NK>(xi, ap, aq);
        }
        float *xi_l = a.xi + (size_t)b * C * V + (hoff4V + v);
        if (!a.xi) {            // inference: nobody reads the first layer's output (the backward would)
        } else if (full) {
#pragma unroll
            for (int r = 0; r < NK; ++r) xi_l[(size_t)((r & 3) + 8 * (r >> 2)) * V] = xi[r];
        } else {
#pragma unroll
            for (int r = 0; r < NK; ++r)
                if (vin) xi_l[(size_t)((r & 3) + 8 * (r >> 2)) * V] = xi[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int r = 0; r < NK; ++r) acc = mfma32(wmx[r], xi[r], acc);
        if constexpr (HASK) {
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) acc = mfma32(wmk[ks], kv[ks], acc);
        }
        float xn[NR2];
#pragma unroll
        for (int r = 0; r < NR2; ++r) xn[r] = acc[r] + b2[r];
        if (!lin2) {
            selu_like_regs<NR2>(xn, ap2, aq2);
        }
        float *xn_l = a.xn + (size_t)b * C2 * V + (hoff4V + v);
        const bool hok = C2 >= 8 || h == 0;           // C2 = 4: the rows live on the h = 0 lanes only
#pragma unroll
        for (int r = 0; r < NR2; ++r)
            if (vin && hok) xn_l[(size_t)((r & 3) + 8 * (r >> 2)) * V] = xn[r];
    }
}

#define PWB_FAST_WAVES 4   // 256-thread blocks, two per CU (512 slabs): measured best of {4, 8, 12} waves x {256, 512, 1024} blocks
// IO16 (with BR and BF16: the FNOSeg / HNOSeg block tail under autocast, round 6): gy, y (the block output and its gradient) and xb (the block
// input) are bf16 IN MEMORY; xa (the operator + branch sum, fp32 in the reference too) and both input gradients stay fp32.  The raw
// halfwords sit zero-extended in the prefetch registers and are widened where they are consumed (a conversion at the load would wait for it).
template <int COUT, int CA, int CB, int NW = PWB_FAST_WAVES, int BR = 0, bool BF16 = false, bool IO16 = false>   // compile-time channel counts: every address select folds
__global__ __launch_bounds__(64 * NW, (NW == 4 && COUT <= 48) ? 2 : 1) void pwconv_bwd_fast_kernel(PwBwdArgs a) {
    static_assert(!IO16 || (BR == 1 && BF16), "bf16 tensors: the fused-branch block tail under bf16 arithmetic only");   // 2 blocks per CU: a 256-register budget, all in VGPRs (no AGPR copies); COUT = 144: one block, 512 registers
    stagger_start((a.dbg >> 24) & 63);
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, c = lane & 31;
    constexpr int CIN = CA + CB, NKO = COUT / 2, NKI = CIN / 2, ICH = (CIN + 31) / 32;
    constexpr int MT = (COUT + 15) / 16, NTI = (CIN + 15) / 16;
    // BR: the xa rows all sit in the first 32-row chunk; P holds p = gxa * act'(xa) as [o][v] for the branch weight gradient
    static_assert(!BR || (CA % 8 == 0 && CA <= 32 && CB % 8 == 0 && CB > 0), "fused branch needs CA <= 32");
    constexpr int RA = CA / 2;                          // accumulator registers of chunk 0 that hold xa rows
    constexpr int MTP = BR ? (CA + 15) / 16 : 0, NTB = BR ? (CB + 15) / 16 : 0;
    constexpr int rowsG = MT * 16, rowsX = BR ? (CA + NTB * 16 > NTI * 16 ? CA + NTB * 16 : NTI * 16) : NTI * 16, rowsP = MTP * 16;
    const unsigned V = a.V;
    // DMA (all but the fused-branch variant): the x rows of the NEXT tile arrive by LDS-DMA into a two-slot ring while this tile is
    // multiplied -- no registers, no ds_write pass, and the loads never queue behind the MFMA chains.  A DMA instruction fills 256
    // contiguous bytes (row pair 2j, 2j + 1 x 32 voxels); pairs are PWB_XP = 68 floats apart, which makes the 16x16x4 operand reads of
    // the weight gradient (16 rows x 4 voxels per instruction) hit 64 different banks: bank = 4 (row >> 1) + 32 (row & 1) + voxel.
    constexpr bool DMA = BR == 0;
    constexpr int XS = DMA ? (rowsX / 2) * PWB_XP : rowsX * PWB_LD;   // floats per x slot
    // WG16 (with IO16, round 6): the weight gradients' operands are bf16 tiles read eight voxels at a time for v_mfma_f32_32x32x16_bf16
    // (6 matrix instructions and 10 ds_read_b128 per tile instead of 80 fp32 products and 72 ds_read_b32); fp32 accumulation.  What
    // torch.autocast does to the convolutions' weight gradients (bf16 operands), with a wider accumulator.  The fp32 tile shrinks to
    // the xa rows (their activation derivative is read back in accumulator order).
    constexpr bool WG16 = IO16;
    constexpr int WP = 40;                                      // halfwords between tile rows: 80 B -> conflict-free ds_read_b128
    constexpr int WAVE_FLOATS = WG16 ? CA * PWB_LD + ((32 + 64 + 32) * WP) / 2 : rowsG * PWB_LD + (DMA ? 2 : 1) * XS + rowsP * PWB_LD;
    float *G = lds + (size_t)wave * WAVE_FLOATS;                // [o][v]
    float *X = WG16 ? G : G + rowsG * PWB_LD;                   // [i][v] (DMA: two slots of [i / 2][i & 1][v]; WG16: the xa rows only)
    float *P = X + (DMA ? 2 : 1) * XS;                          // [o][v] of the branch (BR only)
    pw_u16 *G16 = reinterpret_cast<pw_u16 *>(G + CA * PWB_LD), *X16 = G16 + 32 * WP, *P16 = X16 + 64 * WP;      // (WG16) [row][voxel] bf16
    float wbr[BR ? RA : 1];   // A operand of the branch dgrad: Wbr^T[row = i][k-slot ks -> channel (ks & 3) + 8 (ks >> 2) + 4 h]
    if (BR) {
#pragma unroll
        for (int ks = 0; ks < RA; ++ks) {
            const int o = (ks & 3) + 8 * (ks >> 2) + 4 * h;
            wbr[ks] = c < CB ? a.Wbr[(size_t)o * CB + c] : 0.f;
        }
    }
    f32x4 dwb[BR ? MTP : 1][BR ? NTB : 1];
    float dbb[BR ? RA : 1];
    if (BR) {
#pragma unroll
        for (int m = 0; m < MTP; ++m)
#pragma unroll
            for (int n = 0; n < NTB; ++n) dwb[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < RA; ++ks) dbb[ks] = 0.f;
    }
    float wt[ICH][NKO];  // A operand of dgrad: Wt[row = i][k = o]
#pragma unroll
    for (int ic = 0; ic < ICH; ++ic)
#pragma unroll
        for (int ks = 0; ks < NKO; ++ks) {
            const int i = ic * 32 + c, o = 2 * ks + h;
            wt[ic][ks] = i < CIN ? a.W[(size_t)o * CIN + i] : 0.f;
            if (a.residual && i == o) wt[ic][ks] += 1.f;
        }
    // bf16 (autocast): the input-gradient products run on v_mfma_f32_32x32x16_bf16 (six k-slots per fragment, see pack6); the
    // weight gradient below keeps its fp32 tiles (more accurate than the reference's bf16 weight gradient, same cost as before)
    static_assert(!BF16 || (NKO % 6 == 0 && (!BR || RA % 6 == 0)), "bf16 fragments take six k-slots");
    pw_bf16x8 wtf[ICH][BF16 ? NKO / 6 : 1], wbrf[(BF16 && BR) ? RA / 6 : 1];
    if constexpr (BF16) {
#pragma unroll
        for (int ic = 0; ic < ICH; ++ic)
#pragma unroll
            for (int s6 = 0; s6 < NKO / 6; ++s6) wtf[ic][s6] = pack6(&wt[ic][6 * s6]);
        if constexpr (BR != 0) {
#pragma unroll
            for (int s6 = 0; s6 < RA / 6; ++s6) wbrf[s6] = pack6(&wbr[6 * s6]);
        }
    }
    for (int i = lane; i < WAVE_FLOATS; i += 64) G[i] = 0.f;
    float db[NKO];
#pragma unroll
    for (int ks = 0; ks < NKO; ++ks) db[ks] = 0.f;
    f32x4 dw[MT][NTI];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTI; ++n) dw[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x16 dw16[WG16 ? 3 : 1];      // (WG16) dW columns 0..31, dW columns 32..63, dWbr: rows o = (r & 3) + 8 (r >> 2) + 4 h, column lane & 31
#pragma unroll
    for (int t_ = 0; t_ < (WG16 ? 3 : 1); ++t_)
#pragma unroll
        for (int r = 0; r < 16; ++r) dw16[t_][r] = 0.f;
    static_assert(!WG16 || (COUT <= 32 && CIN <= 64 && CA <= 32 && CA + 32 <= 64), "one 32-row A tile, two 32-column B tiles");
    // branch-free activation derivatives from the saved outputs: act'(y) = (y > 0 or linear) ? dp : y + dq
    const float dp = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE : 1.f, dq = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE * HNO_SELU_ALPHA : 1.f;
    const bool lin = a.act == HNO_ACT_NONE;
    const float xp = a.xa_act == HNO_ACT_SELU ? HNO_SELU_SCALE : 1.f, xq = a.xa_act == HNO_ACT_SELU ? HNO_SELU_SCALE * HNO_SELU_ALPHA : 1.f;
    const bool xact = a.xa_act != HNO_ACT_NONE;
    const unsigned tiles_per_b = (V + 31) / 32;
    const unsigned ntiles = tiles_per_b * a.B;
    const unsigned ngroups = (ntiles + NW - 1) / NW;
    const unsigned hoffV = h ? V : 0u, hoff4V = h ? 4u * V : 0u;
    const float *ga = G + (lane & 15) * PWB_LD + (lane >> 4);
    const float *xbp = DMA ? X + ((lane & 15) >> 1) * PWB_XP + (lane & 1) * 32 + (lane >> 4) : X + (lane & 15) * PWB_LD + (lane >> 4);
    constexpr int XT = DMA ? 8 * PWB_XP : 16 * PWB_LD;          // 16 rows further
    const float *pa = P + (lane & 15) * PWB_LD + (lane >> 4);
    const unsigned x_lds = (unsigned)(size_t)X;                 // LDS byte address of the ring
    // software pipeline: raw loads of the NEXT tile (gy, y, x) are issued before this tile's
    // LDS staging and MFMA work; activation gradient and masking happen when they are consumed
    float pg[NKO], py[NKO] = {}, px[DMA ? 1 : NKI];
    // gradient already stored in gxb (accumulate bit 1 << 1: the U-Net skip consumer arrived first): prefetched with the tile --
    // read inside the store loop instead, every row is a dependent load -> add -> store round trip (the compiler cannot move a
    // load above the previous row's store): 130 us instead of 95 per call
    constexpr int NQ = (!BR && CB > 0) ? CB / 2 : 1;
    float pq[NQ] = {};
    const bool accb = !BR && CB > 0 && CA % 8 == 0 && (a.accum & 2) && a.gxb;      // (straddling rows take the read-modify-write path)
    auto fetch = [&](unsigned grp, int slot) {
        const unsigned t = grp * NW + wave;
        const bool live = t < ntiles;
        const unsigned b = live ? t / tiles_per_b : 0u;
        const unsigned v = live ? (t - b * tiles_per_b) * 32 + c : 0u;
        const unsigned off = hoffV + ((live && v < V) ? v : 0u);
        const float *gy_b = a.gy + (size_t)b * COUT * V, *y_b = a.y + (size_t)b * COUT * V;
        const float *xa_b = a.xa + (size_t)b * CA * V;
        const float *xb_b = CB > 0 ? a.xb + (size_t)b * CB * V : a.xa;
        if constexpr (DMA) {
            // issued BEFORE the register loads below: vector-memory loads complete in order, so the wait hipcc places in front of the
            // first use of pg / py (younger loads) also retires these; out-of-range lanes repeat voxel 0 (finite, and g is 0 there)
#pragma unroll
            for (int j = 0; j < NKI; ++j) {
                const int i0 = 2 * j;
                const float *base = i0 < CA ? xa_b + (size_t)i0 * V : xb_b + (size_t)(i0 - CA) * V;
                dma_row_pair_nt(base, off * 4u, __builtin_amdgcn_readfirstlane(x_lds + (slot * XS + j * PWB_XP) * 4));
            }
        }
        if constexpr (IO16) {
            const pw_u16 *gy16 = reinterpret_cast<const pw_u16 *>(a.gy) + (size_t)b * COUT * V, *y16 = reinterpret_cast<const pw_u16 *>(a.y) + (size_t)b * COUT * V;
            const pw_u16 *xb16 = reinterpret_cast<const pw_u16 *>(a.xb) + (size_t)b * CB * V;
#pragma unroll
            for (int ks = 0; ks < NKO; ++ks) pg[ks] = __builtin_bit_cast(float, (unsigned)(gy16 + (size_t)(2 * ks) * V)[off]);
            if (!lin) {
#pragma unroll
                for (int ks = 0; ks < NKO; ++ks) py[ks] = __builtin_bit_cast(float, (unsigned)(y16 + (size_t)(2 * ks) * V)[off]);
            }
#pragma unroll
            for (int j = 0; j < NKI; ++j) {
                const int i0 = 2 * j;
                // (saved activations, read once: streaming hint -- as dma_row_pair_nt in the DMA variants)
                if (i0 < CA) px[j] = __builtin_nontemporal_load(xa_b + (size_t)i0 * V + off);
                else px[j] = __builtin_bit_cast(float, (unsigned)__builtin_nontemporal_load(xb16 + (size_t)(i0 - CA) * V + off));
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < NKO; ++ks) pg[ks] = (gy_b + (size_t)(2 * ks) * V)[off];
        if (!lin) {
#pragma unroll
            for (int ks = 0; ks < NKO; ++ks) py[ks] = (y_b + (size_t)(2 * ks) * V)[off];
        }
        if constexpr (!DMA) {
#pragma unroll
            for (int j = 0; j < NKI; ++j) {
                const int i0 = 2 * j;
                const float *base = i0 < CA ? xa_b + (size_t)i0 * V : xb_b + (size_t)(i0 - CA) * V;
                px[j] = __builtin_nontemporal_load(base + off);
            }
        }
        }
        if (accb) {
            const float *q_b = a.gxb + (size_t)b * CB * V + (h ? 4u * V : 0u) + ((live && v < V) ? v : 0u);
#pragma unroll
            for (int j = 0; j < NQ; ++j) pq[j] = q_b[(size_t)((j & 3) + 8 * (j >> 2)) * V];   // rows (j & 3) + 8 (j >> 2) + 4 h
        }
    };
    unsigned bid = blockIdx.x;
    if ((a.dbg & 32) && (gridDim.x & 7) == 0) bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    unsigned g_begin = bid, g_end = ngroups, g_step = gridDim.x;
    if (a.dbg & 16) {
        const unsigned per = (ngroups + gridDim.x - 1) / gridDim.x;
        g_begin = bid * per; g_end = g_begin + per < ngroups ? g_begin + per : ngroups; g_step = 1;
    }
    if (g_begin < g_end) fetch(g_begin, 0);
    int slot = 0;
    for (unsigned grp = g_begin; grp < g_end; grp += g_step, slot ^= 1) {
        const float *Xc = X + (DMA ? slot * XS : 0);            // this tile's x rows
        const unsigned tu = grp * NW + wave;                    // wave-uniform copies of the tile coordinates
        const unsigned bu = tu < ntiles ? tu / tiles_per_b : 0u;
        const bool br_fast = BR != 0 && tu < ntiles && (tu - bu * tiles_per_b) * 32 + 32 <= V && a.gxa && a.gxb && !a.accum && !(a.dbg & 4);
        const bool fast_store = BR == 0 && tu < ntiles && (tu - bu * tiles_per_b) * 32 + 32 <= V && a.gxa && (CB == 0 || a.gxb) &&
                                !(a.accum & 1) && !(CA % 8 != 0 && (a.accum & 2)) && !(a.dbg & 4);
        const unsigned t = grp * NW + wave;
        const bool live = t < ntiles;
        const unsigned b = live ? t / tiles_per_b : 0u;
        const unsigned v = live ? (t - b * tiles_per_b) * 32 + c : 0u;
        const bool vin = live && v < V;
        float g[NKO];
        auto widen = [](float raw) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, raw) << 16); };   // zero-extended bf16 bits -> fp32
#pragma unroll
        for (int ks = 0; ks < NKO; ++ks) {
            const float pgv = IO16 ? widen(pg[ks]) : pg[ks], pyv = IO16 ? widen(py[ks]) : py[ks];
            const float gv = pgv * ((lin || pyv > 0.f) ? dp : pyv + dq);
            g[ks] = vin ? gv : 0.f;
            if constexpr (WG16) G16[(2 * ks + h) * WP + c] = f32_to_bf16_bits(g[ks]);
            else G[(2 * ks + h) * PWB_LD + c] = g[ks];
            db[ks] += g[ks];
        }
        if constexpr (WG16) {
#pragma unroll
            for (int j = 0; j < NKI; ++j) {
                if (2 * j < CA) {      // xa row (fp32): kept for its activation derivative, rounded for the weight gradient
                    const float xv_ = vin ? px[j] : 0.f;
                    X[(2 * j + h) * PWB_LD + c] = xv_;
                    X16[(2 * j + h) * WP + c] = f32_to_bf16_bits(xv_);
                } else {               // xb row: the prefetch register holds the bf16 bits as they lie in memory
                    X16[(2 * j + h) * WP + c] = vin ? (pw_u16)__builtin_bit_cast(unsigned, px[j]) : (pw_u16)0;
                }
            }
        } else if constexpr (!DMA) {
#pragma unroll
            for (int j = 0; j < NKI; ++j) X[(2 * j + h) * PWB_LD + c] = vin ? ((IO16 && 2 * j >= CA) ? widen(px[j]) : px[j]) : 0.f;
        }
        float qc[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) qc[j] = pq[j];
        if (grp + g_step < g_end) fetch(grp + g_step, slot ^ 1);
        f32x16 acc2;   // BR: Wbr^T p, rows = xb channels
#pragma unroll
        for (int ic = 0; ic < ICH; ++ic) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if constexpr (BF16) {
#pragma unroll
                for (int s6 = 0; s6 < NKO / 6; ++s6) acc = mfma_bf(wtf[ic][s6], pack6(&g[6 * s6]), acc);
            } else if (!(a.dbg & 1)) {
#pragma unroll
                for (int ks = 0; ks < NKO; ++ks) acc = mfma32(wt[ic][ks], g[ks], acc);
            } else {
#pragma unroll
                for (int ks = 0; ks < NKO; ++ks) acc[ks & 15] += g[ks];
            }
            auto store_row = [&](int r) {
                const int irow = ic * 32 + (r & 3) + 8 * (r >> 2);
                if (irow < CIN) {  // compile-time; rows irow, irow+4 are on one side since CA % 8 == 0
                    float *base = irow < CA ? (a.gxa ? a.gxa + ((size_t)b * CA + irow) * V : nullptr)
                                            : (a.gxb ? a.gxb + ((size_t)b * CB + (irow - CA)) * V : nullptr);
                    float gv = acc[r];
                    if (irow < CA && xact) {
                        const float xo = X[(irow + 4 * h) * PWB_LD + c];
                        gv *= xo > 0.f ? xp : xo + xq;
                    }
                    if (BR && irow < CA) {   // p: B operand of the branch dgrad (k-slot r), A operand of its weight gradient
                        acc[r] = gv;
                        if constexpr (WG16) P16[(irow + 4 * h) * WP + c] = f32_to_bf16_bits(vin ? gv : 0.f);
                        else P[(irow + 4 * h) * PWB_LD + c] = gv;
                        dbb[r < RA ? r : 0] += gv;
                    }
                    if (BR && irow >= CA) gv += acc2[(r + 16 * ic - RA) & 15];
                    if (base && vin && (irow + 4 < CIN || h == 0) && !((a.dbg & 4) && acc[r] != 12345.678f)) {
                        if (a.accum & (irow < CA ? 1 : 2)) gv += base[hoff4V + v];
                        base[hoff4V + v] = gv;
                    }
                }
            };
            if constexpr (BR == 0) {
              if (fast_store) {
                // whole tile, both gradients wanted, nothing to accumulate from memory: no per-row control flow.  The activation
                // derivative's x values are read from LDS in one batch (a dependent ds_read + wait per row otherwise).
                float xo[16];
                if (xact) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int irow = ic * 32 + (r & 3) + 8 * (r >> 2);
                        if (irow < CA) xo[r] = DMA ? Xc[((irow >> 1) + 2 * h) * PWB_XP + (irow & 1) * 32 + c] : X[(irow + 4 * h) * PWB_LD + c];
                    }
                }
                float *ga_l = a.gxa + (size_t)b * CA * V + (hoff4V + v);
                float *gb_l = CB > 0 ? a.gxb + (size_t)b * CB * V + (hoff4V + v) : nullptr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int irow = ic * 32 + (r & 3) + 8 * (r >> 2);
                    if (irow < CIN) {
                        // register r holds row irow (h = 0 lanes) and irow + 4 (h = 1 lanes): both xa rows, both xb rows, or -- when CA is
                        // not a multiple of 8 (12 + 12 channels) -- an xa row and an xb row
                        const bool a0 = irow < CA, a1 = irow + 4 < CA;         // fold after unrolling
                        float gv = acc[r];
                        if (a0 && xact) {
                            const float f = xo[r] > 0.f ? xp : xo[r] + xq;
                            gv *= (a1 || h == 0) ? f : 1.f;
                        }
                        if (irow >= CA && CB > 0 && accb) {
                            const int i = irow - CA;
                            gv += qc[((i >> 3) << 2) + (i & 3) < NQ ? ((i >> 3) << 2) + (i & 3) : 0];
                        }
                        float *dst;
                        if (a0 == a1) dst = a0 ? ga_l + (size_t)irow * V : gb_l + (size_t)(irow - CA) * V;
                        else dst = h ? a.gxb + ((size_t)b * CB + (irow + 4 - CA)) * V + v : a.gxa + ((size_t)b * CA + irow) * V + v;
                        if (irow + 4 < CIN) *dst = gv;
                        else if (h == 0) *dst = gv;
                    }
                }
              } else
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int irow = ic * 32 + (r & 3) + 8 * (r >> 2);
                    if (irow < CIN) {  // compile-time
                        const bool a0 = irow < CA, a1 = irow + 4 < CA;          // see the whole-tile path above
                        const bool on_a = (a0 == a1) ? a0 : h == 0;           // this lane's row is an xa row
                        const int lrow = irow + 4 * h;                          // this lane's row of [xa ; xb]
                        float *base = on_a ? (a.gxa ? a.gxa + ((size_t)b * CA + lrow) * V : nullptr)
                                           : (a.gxb ? a.gxb + ((size_t)b * CB + (lrow - CA)) * V : nullptr);
                        if (base && vin && (irow + 4 < CIN || h == 0) && !((a.dbg & 4) && acc[r] != 12345.678f)) {
                            float gv = acc[r];
                            if (a0 && xact && on_a) {
                                const float xo = DMA ? Xc[((irow >> 1) + 2 * h) * PWB_XP + (irow & 1) * 32 + c] : X[(irow + 4 * h) * PWB_LD + c];
                                gv *= xo > 0.f ? xp : xo + xq;
                            }
                            if (irow >= CA && CB > 0) {
                                const int i = irow - CA;                       // compile-time: xb row (of the h = 0 half)
                                if (accb) gv += qc[((i >> 3) << 2) + (i & 3) < NQ ? ((i >> 3) << 2) + (i & 3) : 0];
                                else if (CA % 8 != 0 && (a.accum & 2)) gv += base[v];
                            } else if (a.accum & (on_a ? 1 : 2)) gv += base[v];
                            base[v] = gv;
                        }
                    }
                }
            } else if (br_fast) {
                // whole tile, both gradients stored, nothing accumulated from memory: the rows of store_row without per-row control flow,
                // the activation derivative's x values read from LDS in one batch
                float *ga_l = a.gxa + (size_t)b * CA * V + (hoff4V + v);
                float *gb_l = a.gxb + (size_t)b * CB * V + (hoff4V + v);
                if (ic == 0) {
                    float xo[RA];
                    if (xact) {
#pragma unroll
                        for (int r = 0; r < RA; ++r) xo[r] = X[((r & 3) + 8 * (r >> 2) + 4 * h) * PWB_LD + c];
                    }
#pragma unroll
                    for (int r = 0; r < RA; ++r) {
                        const int irow = (r & 3) + 8 * (r >> 2);
                        float gv = acc[r];
                        if (xact) gv *= xo[r] > 0.f ? xp : xo[r] + xq;
                        acc[r] = gv;
                        if constexpr (WG16) P16[(irow + 4 * h) * WP + c] = f32_to_bf16_bits(gv);
                        else P[(irow + 4 * h) * PWB_LD + c] = gv;
                        dbb[r] += gv;
                        ga_l[(size_t)irow * V] = gv;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
                    if constexpr (BF16) {
                        float pr[RA];
#pragma unroll
                        for (int ks = 0; ks < RA; ++ks) pr[ks] = acc[ks];
#pragma unroll
                        for (int s6 = 0; s6 < RA / 6; ++s6) acc2 = mfma_bf(wbrf[s6], pack6(&pr[6 * s6]), acc2);
                    } else {
#pragma unroll
                        for (int ks = 0; ks < RA; ++ks) acc2 = mfma32(wbr[ks], acc[ks], acc2);
                    }
                }
#pragma unroll
                for (int r = (ic == 0 ? RA : 0); r < 16; ++r) {
                    const int irow = ic * 32 + (r & 3) + 8 * (r >> 2);
                    if (irow < CIN) gb_l[(size_t)(irow - CA) * V] = acc[r] + acc2[(r + 16 * ic - RA) & 15];
                }
            } else if (ic == 0) {
#pragma unroll
                for (int r = 0; r < RA; ++r) store_row(r);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
                if constexpr (BF16) {
                    float pr[RA];
#pragma unroll
                    for (int ks = 0; ks < RA; ++ks) pr[ks] = acc[ks];
#pragma unroll
                    for (int s6 = 0; s6 < RA / 6; ++s6) acc2 = mfma_bf(wbrf[s6], pack6(&pr[6 * s6]), acc2);
                } else {
#pragma unroll
                    for (int ks = 0; ks < RA; ++ks) acc2 = mfma32(wbr[ks], acc[ks], acc2);
                }
#pragma unroll
                for (int r = RA; r < 16; ++r) store_row(r);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) store_row(r);
            }
        }
        // the LDS tile is wave-private: LDS ops of one wave execute in order, so only the compiler
        // has to be kept from reordering the reads above the writes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if constexpr (WG16) {
            // dW[o][i] += sum_v g[o][v] x[i][v]: A = rows of G16 (o = lane & 31), B = rows of X16 (i = lane & 31 (+ 32)), eight voxels
            // 8 h + j (+ 16 per k-step) per lane; dWbr[o][i] += sum_v p[o][v] xb[i][v] with the xb rows starting at row CA of X16
            const pw_u16 *ga16 = G16 + c * WP + 8 * h, *pa16 = P16 + c * WP + 8 * h;
            const pw_u16 *x0 = X16 + c * WP + 8 * h, *x1 = X16 + (32 + c) * WP + 8 * h, *xbr = X16 + (CA + c) * WP + 8 * h;
#pragma unroll
            for (int ks = 0; ks < ((a.dbg & 2) ? 0 : 2); ++ks) {
                const pw_bf16x8 av = *reinterpret_cast<const pw_bf16x8 *>(ga16 + 16 * ks), pv = *reinterpret_cast<const pw_bf16x8 *>(pa16 + 16 * ks);
                const pw_bf16x8 b0 = *reinterpret_cast<const pw_bf16x8 *>(x0 + 16 * ks), b1 = *reinterpret_cast<const pw_bf16x8 *>(x1 + 16 * ks);
                const pw_bf16x8 bb = *reinterpret_cast<const pw_bf16x8 *>(xbr + 16 * ks);
                dw16[0] = mfma_bf(av, b0, dw16[0]);
                dw16[1] = mfma_bf(av, b1, dw16[1]);
                dw16[2] = mfma_bf(pv, bb, dw16[2]);
            }
        } else
#pragma unroll 2
        for (int ks = 0; ks < ((a.dbg & 2) ? 0 : 8); ++ks) {
            float av[MT], bv[NTI];
#pragma unroll
            for (int m = 0; m < MT; ++m) av[m] = ga[m * 16 * PWB_LD + ks * 4];
#pragma unroll
            for (int n = 0; n < NTI; ++n) bv[n] = xbp[(DMA ? slot * XS : 0) + n * XT + ks * 4];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NTI; ++n) dw[m][n] = mfma16(av[m], bv[n], dw[m][n]);
            if (BR) {   // dWbr[o][i] += p[o][v] xb[i][v]: the xb rows start at row CA of X
                float pv[BR ? MTP : 1], xv2[BR ? NTB : 1];
#pragma unroll
                for (int m = 0; m < MTP; ++m) pv[m] = pa[m * 16 * PWB_LD + ks * 4];
#pragma unroll
                for (int n = 0; n < NTB; ++n) xv2[n] = xbp[(CA + n * 16) * PWB_LD + ks * 4];
#pragma unroll
                for (int m = 0; m < MTP; ++m)
#pragma unroll
                    for (int n = 0; n < NTB; ++n) dwb[m][n] = mfma16(pv[m], xv2[n], dwb[m][n]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    {
        constexpr int n = COUT * CIN + COUT + (BR ? CA * CB + CA : 0);
        __syncthreads();
        float *mine = lds + (size_t)wave * n;
        if constexpr (WG16) {
            float *mb = mine + COUT * CIN + COUT;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = (r & 3) + 8 * (r >> 2) + 4 * h;
                if (o < COUT) {
                    mine[o * CIN + c] = dw16[0][r];
                    if (32 + c < CIN) mine[o * CIN + 32 + c] = dw16[1][r];
                }
                if (o < CA && c < CB) mb[o * CB + c] = dw16[2][r];
            }
#pragma unroll
            for (int ks = 0; ks < RA; ++ks) {
                float sb = dbb[ks];
                for (int off2 = 16; off2 >= 1; off2 >>= 1) sb += __shfl_xor(sb, off2);
                if (c == 0) mb[CA * CB + (ks & 3) + 8 * (ks >> 2) + 4 * h] = sb;
            }
        } else if (BR) {   // slab tail: dWbr [o][i], dbbr [o]
            float *mb = mine + COUT * CIN + COUT;
#pragma unroll
            for (int m = 0; m < MTP; ++m)
#pragma unroll
                for (int nn = 0; nn < NTB; ++nn)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int o = m * 16 + (lane >> 4) * 4 + r, i = nn * 16 + (lane & 15);
                        if (o < CA && i < CB) mb[o * CB + i] = dwb[m][nn][r];
                    }
#pragma unroll
            for (int ks = 0; ks < RA; ++ks) {
                float sb = dbb[ks];
                for (int off2 = 16; off2 >= 1; off2 >>= 1) sb += __shfl_xor(sb, off2);
                if (c == 0) mb[CA * CB + (ks & 3) + 8 * (ks >> 2) + 4 * h] = sb;
            }
        }
        if constexpr (!WG16) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nn = 0; nn < NTI; ++nn)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = m * 16 + (lane >> 4) * 4 + r, i = nn * 16 + (lane & 15);
                    if (o < COUT && i < CIN) mine[o * CIN + i] = dw[m][nn][r];
                }
        }
#pragma unroll
        for (int ks = 0; ks < NKO; ++ks) {
            float s = db[ks];
            for (int off2 = 16; off2 >= 1; off2 >>= 1) s += __shfl_xor(s, off2);
            if (c == 0) mine[COUT * CIN + 2 * ks + h] = s;
        }
        block_sum_to_slab(lds, n, a.partials + (size_t)blockIdx.x * n, threadIdx.x, NW);
    }
}

// ---- backward of the two chained pointwise layers (pwconv_fwd_chain_kernel) in one pass:
//   g1 = gn * act'(xn)                                 gradient of the mapping convolution's pre-activation
//   [gxi ; gk] = Wm^T g1                               gk: gradient of the U-Net skip tensor (stored)
//   p  = gxi * act'(xi)                                gradient of the concat convolution's pre-activation -- stays in registers
//   [gu ; gt] = Wc^T p,  gu *= xa_act'(u)              (the activation of PadInverse's output is applied to u by its producer)
//   dWm += g1 [xi ; k]^T, dbm += sum g1;   dWc += p [u ; t]^T, dbc += sum p
// The two layers apart (pwconv_bwd_fast_kernel twice) move 12 streams of 24 channels -- gn, xn, xi, k -> gxi, gk and gxi, xi, u, t ->
// gu, gt --, this kernel 9: the gradient between the layers never reaches memory and xi is read once.  Structure as the fast kernel:
// the four x tensors of the NEXT tile arrive by LDS-DMA into the other slot of a two-slot ring (row pairs, PWB_XP apart) while this
// tile is multiplied, gn / xn are register-prefetched right behind them (the compiler's wait for those retires the DMAs: in-order
// return), dgrads on 32x32x2 with the weights as A operands, weight gradients on 16x16x4 over wave-private LDS tiles (g1 then p in ONE
// [channel][voxel] tile), one slab of partial sums per workgroup.
struct PwChainBwdArgs {
    const float *gn, *xn, *xi, *k, *u, *t;    // (B, C, V)
    const float *Wm, *Wc;                      // (C, 2C)
    float *gu, *gt, *gk;                       // (B, C, V)
    float *partials;                           // per workgroup: [dWc C x 2C | dbc C | dWm C2 x CIN2 | dbm C2]
    int B;
    unsigned V;
    int act, act2, xa_act, dbg;
};

// C2 / HASK as pwconv_fwd_chain_kernel's: <24, NW, 24, true> mapping_conv behind conv_concat; <24, NW, 4, false> the model's conv_out
// (no second input, no bias; its activation a.act2 is none) behind the last block's conv_concat.
// SLOTS: 2 = the next tile's x rows land in the other ring slot while this tile is multiplied (30.5 KB of LDS per wave: four waves per
// CU, one per SIMD -- nothing overlaps a wave's VALU / scalar work and waits with MFMAs: 38 % MFMA-busy, 128 us); 1 = one slot, the next
// tile requested when this one's LDS reads are done (17.4 KB: eight waves per CU, two workgroups of four -- the partner wave on the
// SIMD computes while this one waits for its rows).
template <int C, int NW, int C2 = C, bool HASK = true, int SLOTS = 2>
__global__ __launch_bounds__(64 * NW, SLOTS == 1 ? 2 : 1) void pwconv_bwd_chain_kernel(PwChainBwdArgs a) {
    stagger_start((a.dbg >> 24) & 63);
    static_assert(C == 24, "accumulator-row bookkeeping below is written for 24 channels");
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, c = lane & 31;
    constexpr int CIN = 2 * C, NK = C / 2, NT = HASK ? 4 : 3, NPAIR = NT * NK;   // row pairs per slot: xi, (k,) u, t
    constexpr int CIN2 = HASK ? 2 * C : C, NK2 = C2 / 2, UP = (HASK ? 2 : 1) * NK;   // second layer: inputs, k-slots of g1; first pair of u
    constexpr int MT2 = (C2 + 15) / 16, NTI2 = (CIN2 + 15) / 16;      // weight-gradient tiles of the second layer
    constexpr int XS = NPAIR * PWB_XP;                                // floats per slot
    constexpr int WAVE_FLOATS = 32 * PWB_LD + SLOTS * XS;
    float *G = lds + (size_t)wave * WAVE_FLOATS;                      // [o][v]: g1, later p
    float *X = G + 32 * PWB_LD;                                       // two slots of [pair][row & 1][v]
    const unsigned V = a.V;
    // A operands of the two input-gradient products.  First: Wm^T, rows i = 32 ic + c, k-slots = output channels 2 ks + h (g1 sits in
    // registers as row pairs).  Second: Wc^T, rows i, k-slot r of lane half h = channel (r & 3) + 8 (r >> 2) + 4 h (p sits in the first
    // product's accumulator registers).
    float wm[2][NK2], wc[2][NK];
#pragma unroll
    for (int ic = 0; ic < 2; ++ic) {
        const int i = ic * 32 + c;
#pragma unroll
        for (int ks = 0; ks < NK2; ++ks) wm[ic][ks] = i < CIN2 ? a.Wm[(size_t)(2 * ks + h) * CIN2 + i] : 0.f;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) wc[ic][ks] = i < CIN ? a.Wc[(size_t)((ks & 3) + 8 * (ks >> 2) + 4 * h) * CIN + i] : 0.f;
    }
    for (int i = lane; i < WAVE_FLOATS; i += 64) G[i] = 0.f;
    float dbm[NK2], dbc[NK];
#pragma unroll
    for (int ks = 0; ks < NK2; ++ks) dbm[ks] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) dbc[ks] = 0.f;
    f32x4 dwm[MT2][NTI2], dwc[2][3];
#pragma unroll
    for (int m = 0; m < MT2; ++m)
#pragma unroll
        for (int n = 0; n < NTI2; ++n) dwm[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) dwc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float dp2 = a.act2 == HNO_ACT_SELU ? HNO_SELU_SCALE : 1.f, dq2 = a.act2 == HNO_ACT_SELU ? HNO_SELU_SCALE * HNO_SELU_ALPHA : 1.f;
    const bool lin2 = a.act2 == HNO_ACT_NONE;
    const float dp = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE : 1.f, dq = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE * HNO_SELU_ALPHA : 1.f;
    const bool lin = a.act == HNO_ACT_NONE;
    const float xp = a.xa_act == HNO_ACT_SELU ? HNO_SELU_SCALE : 1.f, xq = a.xa_act == HNO_ACT_SELU ? HNO_SELU_SCALE * HNO_SELU_ALPHA : 1.f;
    const bool xact = a.xa_act != HNO_ACT_NONE;
    const unsigned tiles_per_b = (V + 31) / 32;
    const unsigned ntiles = tiles_per_b * a.B;
    const unsigned ngroups = (ntiles + NW - 1) / NW;
    const unsigned hoffV = h ? V : 0u, hoff4V = h ? 4u * V : 0u;
    const float *ga = G + (lane & 15) * PWB_LD + (lane >> 4);
    const float *xbp = X + ((lane & 15) >> 1) * PWB_XP + (lane & 1) * 32 + (lane >> 4);      // + slot XS + tensor-pair offset + n 8 XP + 4 ks
    const unsigned x_lds = (unsigned)(size_t)X;
    float pg[NK2], py[NK2] = {};
    auto fetch = [&](unsigned grp, int slot) {
        const unsigned t = grp * NW + wave;
        const bool live = t < ntiles;
        const unsigned b = live ? t / tiles_per_b : 0u;
        const unsigned v = live ? (t - b * tiles_per_b) * 32 + c : 0u;
        const unsigned off = hoffV + ((live && v < V) ? v : 0u);
        const size_t bo = (size_t)b * C * V;
        const float *srcs[4] = {a.xi + bo, HASK ? a.k + bo : a.u + bo, HASK ? a.u + bo : a.t + bo, a.t + bo};
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int j = 0; j < NK; ++j)
                dma_row_pair_nt(srcs[q] + (size_t)(2 * j) * V, off * 4u, __builtin_amdgcn_readfirstlane(x_lds + (slot * XS + (q * NK + j) * PWB_XP) * 4));
        // register loads issued BEHIND the DMAs: the wait hipcc places in front of their first use retires the DMAs too (in-order return)
        const size_t bo2 = (size_t)b * C2 * V;
        const float *gn_b = a.gn + bo2, *xn_b = a.xn + bo2;
#pragma unroll
        for (int ks = 0; ks < NK2; ++ks) pg[ks] = (gn_b + (size_t)(2 * ks) * V)[off];
        if (!lin2) {
#pragma unroll
            for (int ks = 0; ks < NK2; ++ks) py[ks] = (xn_b + (size_t)(2 * ks) * V)[off];
        }
    };
    if (blockIdx.x < ngroups) fetch(blockIdx.x, 0);
    int slot = 0;
    for (unsigned grp = blockIdx.x; grp < ngroups; grp += gridDim.x, slot ^= (SLOTS - 1)) {
        const float *Xc = X + slot * XS;
        const unsigned t = grp * NW + wave;
        const bool live = t < ntiles;
        const unsigned b = live ? t / tiles_per_b : 0u;
        const unsigned v = live ? (t - b * tiles_per_b) * 32 + c : 0u;
        const bool vin = live && v < V;
        // ---- g1 = gn act'(xn): B operand of the first product (registers), A operand of dWm (LDS tile)
        float g[NK2];
#pragma unroll
        for (int ks = 0; ks < NK2; ++ks) {
            const float gv = pg[ks] * ((lin2 || py[ks] > 0.f) ? dp2 : py[ks] + dq2);
            g[ks] = vin ? gv : 0.f;
            G[(2 * ks + h) * PWB_LD + c] = g[ks];
            dbm[ks] += g[ks];
        }
        if (SLOTS == 2 && grp + gridDim.x < ngroups) fetch(grp + gridDim.x, slot ^ 1);
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NK2; ++ks) acc0 = mfma32(wm[0][ks], g[ks], acc0);
        if constexpr (HASK) {
#pragma unroll
            for (int ks = 0; ks < NK2; ++ks) acc1 = mfma32(wm[1][ks], g[ks], acc1);
        }
        // rows of [gxi ; gk]: acc0 registers 0..11 = gxi rows (r & 3) + 8 (r >> 2) + 4 h; registers 12..15 = gk rows 0..7; acc1 registers
        // 0..7 = gk rows 8..23
        float xo[NK];
#pragma unroll
        for (int r = 0; r < NK; ++r) {
            const int irow = (r & 3) + 8 * (r >> 2);
            xo[r] = Xc[((irow >> 1) + 2 * h) * PWB_XP + (irow & 1) * 32 + c];                    // xi: pairs 0 .. NK - 1
        }
        float *gk_l = HASK ? a.gk + (size_t)b * C * V + (hoff4V + v) : nullptr;
        if (HASK && vin) {
#pragma unroll
            for (int r = 12; r < 16; ++r) gk_l[(size_t)((r & 3) + 8 * (r >> 2) - 24) * V] = acc0[r];
#pragma unroll
            for (int r = 0; r < 8; ++r) gk_l[(size_t)(8 + (r & 3) + 8 * (r >> 2)) * V] = acc1[r];
        }
        // the wave's LDS reads of g1 (weight gradient of the mapping layer) before the tile is overwritten with p
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll 2
        for (int ks = 0; ks < 8; ++ks) {
            float av[MT2], bv[NTI2];
#pragma unroll
            for (int m = 0; m < MT2; ++m) av[m] = ga[m * 16 * PWB_LD + ks * 4];
#pragma unroll
            for (int n = 0; n < NTI2; ++n) bv[n] = xbp[slot * XS + n * 8 * PWB_XP + ks * 4];       // [xi (; k)]: pairs 0 ..; columns >= CIN2 dropped below
#pragma unroll
            for (int m = 0; m < MT2; ++m)
#pragma unroll
                for (int n = 0; n < NTI2; ++n) dwm[m][n] = mfma16(av[m], bv[n], dwm[m][n]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- p = gxi act'(xi): stays in acc0[0..11] (B operand of the second product), copy to the LDS tile for dWc
        float pr[NK];
#pragma unroll
        for (int r = 0; r < NK; ++r) {
            const int irow = (r & 3) + 8 * (r >> 2);
            const float pv = vin ? acc0[r] * ((lin || xo[r] > 0.f) ? dp : xo[r] + dq) : 0.f;
            pr[r] = pv;
            G[(irow + 4 * h) * PWB_LD + c] = pv;
            dbc[r] += pv;
        }
        f32x16 acc2, acc3;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = acc3[r] = 0.f;
#pragma unroll
        for (int r = 0; r < NK; ++r) acc2 = mfma32(wc[0][r], pr[r], acc2);
#pragma unroll
        for (int r = 0; r < NK; ++r) acc3 = mfma32(wc[1][r], pr[r], acc3);
        float uo[NK];
        if (xact) {
#pragma unroll
            for (int r = 0; r < NK; ++r) {
                const int irow = (r & 3) + 8 * (r >> 2);
                uo[r] = Xc[(UP + (irow >> 1) + 2 * h) * PWB_XP + (irow & 1) * 32 + c];          // u: pairs UP .. UP + NK - 1
            }
        }
        float *gu_l = a.gu + (size_t)b * C * V + (hoff4V + v), *gt_l = a.gt + (size_t)b * C * V + (hoff4V + v);
        if (vin) {
#pragma unroll
            for (int r = 0; r < NK; ++r) {
                float gv = acc2[r];
                if (xact) gv *= uo[r] > 0.f ? xp : uo[r] + xq;
                gu_l[(size_t)((r & 3) + 8 * (r >> 2)) * V] = gv;
            }
#pragma unroll
            for (int r = 12; r < 16; ++r) gt_l[(size_t)((r & 3) + 8 * (r >> 2) - 24) * V] = acc2[r];
#pragma unroll
            for (int r = 0; r < 8; ++r) gt_l[(size_t)(8 + (r & 3) + 8 * (r >> 2)) * V] = acc3[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll 2
        for (int ks = 0; ks < 8; ++ks) {
            float av[2], bv[3];
#pragma unroll
            for (int m = 0; m < 2; ++m) av[m] = ga[m * 16 * PWB_LD + ks * 4];
#pragma unroll
            for (int n = 0; n < 3; ++n) bv[n] = xbp[slot * XS + (UP + n * 8) * PWB_XP + ks * 4];   // [u ; t]: pairs UP .. UP + 2 NK - 1
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) dwc[m][n] = mfma16(av[m], bv[n], dwc[m][n]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (SLOTS == 1 && grp + gridDim.x < ngroups) fetch(grp + gridDim.x, 0);      // (this tile's LDS reads are done)
    }
    {
        // slab [dWc | dbc | dWm | dbm]: the order of the four parameters in the model (conv_concat of block i, mapping_conv of block i + 1 /
        // conv_out), so that under a data-parallel replica the reduced slab IS their run of the flat gradient buffer
        constexpr int nc = C * CIN + C, n = nc + C2 * CIN2 + C2;
        __syncthreads();
        float *mine = lds + (size_t)wave * n;
#pragma unroll
        for (int m = 0; m < MT2; ++m)
#pragma unroll
            for (int nn = 0; nn < NTI2; ++nn)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = m * 16 + (lane >> 4) * 4 + r, i = nn * 16 + (lane & 15);
                    if (o < C2 && i < CIN2) mine[nc + o * CIN2 + i] = dwm[m][nn][r];
                }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nn = 0; nn < 3; ++nn)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = m * 16 + (lane >> 4) * 4 + r, i = nn * 16 + (lane & 15);
                    if (o < C) mine[o * CIN + i] = dwc[m][nn][r];
                }
#pragma unroll
        for (int ks = 0; ks < NK2; ++ks) {
            float s1 = dbm[ks];
            for (int off2 = 16; off2 >= 1; off2 >>= 1) s1 += __shfl_xor(s1, off2);
            if (c == 0) mine[nc + C2 * CIN2 + 2 * ks + h] = s1;                         // g1 rows are pairs 2 ks + h
        }
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            float s2 = dbc[ks];
            for (int off2 = 16; off2 >= 1; off2 >>= 1) s2 += __shfl_xor(s2, off2);
            if (c == 0) mine[C * CIN + (ks & 3) + 8 * (ks >> 2) + 4 * h] = s2;         // p rows are accumulator rows
        }
        block_sum_to_slab(lds, n, a.partials + (size_t)blockIdx.x * n, threadIdx.x, NW);
    }
}

// dynamic LDS of pwconv_bwd_fast_kernel<COUT, CA, CB, NW, BR>: per wave G (rows of 16) + x (ring of two DMA slots, or one padded tile with
// the fused branch) + P; never less than the slab staging of its epilogue
template <int COUT, int CA, int CB, int BR>
static size_t pwb_fast_lds_bytes(int nw) {
    constexpr int CIN = CA + CB, MT = (COUT + 15) / 16, NTI = (CIN + 15) / 16;
    constexpr int MTP = BR ? (CA + 15) / 16 : 0, NTB = BR ? (CB + 15) / 16 : 0;
    constexpr int rowsG = MT * 16, rowsX = BR ? (CA + NTB * 16 > NTI * 16 ? CA + NTB * 16 : NTI * 16) : NTI * 16, rowsP = MTP * 16;
    constexpr int xs = BR ? rowsX * PWB_LD : 2 * (rowsX / 2) * PWB_XP;
    constexpr int slab = COUT * CIN + COUT + (BR ? CA * CB + CA : 0);
    constexpr int per_wave = rowsG * PWB_LD + xs + rowsP * PWB_LD;
    return sizeof(float) * (size_t)nw * (per_wave > slab ? per_wave : slab);
}

static int grid_for(long long work_items, int per_block) {
    long long g = (work_items + per_block - 1) / per_block;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

int pwconv_fwd_launch(const float *xa, int Ca, const float *xb, int Cb, const float *W, const float *bias,
                      float *y, int B, int Cout, long long V, int act, int residual, void *stream) {
    const int bf16 = (act >> 12) & 1;      // HNO_ACT_BF16: bf16 matrix-core arithmetic (autocast); built for the 24 / 48-channel shapes
    act &= 0xfff;
    HNO_REQUIRE(xa && W && y && Ca > 0 && Cb >= 0 && B > 0 && Cout > 0 && V > 0, "hno_pwconv_fwd: bad argument");
    HNO_REQUIRE(Cb == 0 || xb, "hno_pwconv_fwd: xb is NULL but Cb > 0");
    if (V * 8 >= (1ll << 32) || ((V + 31) / 32) * B >= (1ll << 31))
        return fail(HNO_ELIMIT, "hno_pwconv_fwd: V=%lld voxels per channel exceeds the 32-bit offset range", V);
    // 144 <- 12 in one launch (five 32-row output tiles per wave; see pwconv_bwd_launch): the three thirds cost three latency-bound launches
    static const bool wide144 = !(getenv("HNO_PW_WIDE144") && atoi(getenv("HNO_PW_WIDE144")) == 0);
    const bool one_launch_144 = wide144 && Cout == 144 && !bf16;
    if (B == 1 && Ca == 12 && Cb == 0 && Cout > 48 && Cout % 48 == 0 && !residual && !one_launch_144) {
        // HartleyMHASeg's fused q / k / v projection (12 -> 144 on the kept spectrum): three launches of the 12 -> 48 fast kernel on
        // output-channel thirds instead of five of the generic one (one sample: the thirds are contiguous blocks of y)
        for (int o0 = 0; o0 < Cout; o0 += 48) {
            const int rc = pwconv_fwd_launch(xa, Ca, nullptr, 0, W + (size_t)o0 * Ca, bias ? bias + o0 : nullptr, y + (size_t)o0 * V, 1, 48, V,
                                             act | (bf16 ? HNO_ACT_BF16 : 0), 0, stream);
            if (rc) return rc;
        }
        return HNO_OK;
    }
    PwArgs a;
    a.xa = xa; a.xb = xb; a.W = W; a.bias = bias; a.y = y;
    a.Ca = Ca; a.Cb = Cb; a.Cin = Ca + Cb; a.Cout = Cout; a.B = B; a.V = (unsigned)V; a.act = act;
    a.residual = residual;
    a.dbg = debug_flags() | (pw_stagger() << 24);
    const long long ntiles = ((V + 31) / 32) * B;
    int grid = grid_for(ntiles, 4);
    if (Ca % 4 == 0 && Cb % 4 == 0) {  // exact-size fast paths (the HNOSeg-XS shapes; 12 channels: HartleyMHASeg)
        a.o_begin = 0; a.k_begin = 0; a.k_count = a.Cin; a.accumulate = 0; a.finalize = 1;
        hipStream_t fs = (hipStream_t)stream;
        bool done = true;
        ProfScope ps(KID_PWCONV_FWD, fs, 4.0 * B * (double)V * (a.Cin + Cout));
        int fgrid = (int)((ntiles + PWF_DMA_WAVES - 1) / PWF_DMA_WAVES);
        static const int fcap_env = getenv("HNO_PWF_GRID_CAP") ? atoi(getenv("HNO_PWF_GRID_CAP")) : 0;      // A/B aid
        // 24-channel shapes (HNOSeg-XS): one block per CU measured best; the 12-channel shapes of HartleyMHASeg do half the products per
        // tile and want two (model step 10.63 -> 10.55 ms; 1 024: 10.57)
        const int fcap = fcap_env > 0 ? fcap_env : (Ca == 12 ? 512 : 256);
        if (fgrid > fcap) fgrid = fcap;   // one block per CU
        if (debug_grid()) fgrid = debug_grid();
        const dim3 fb(64 * PWF_DMA_WAVES);
        const size_t fl = (size_t)PWF_DMA_WAVES * 2 * (a.Cin / 2) * 256;      // the waves' DMA rings
        if (bf16 && Ca == 24 && Cb == 0 && Cout == 24) hipLaunchKernelGGL((pwconv_fwd_fast_kernel<24, 0, 24, true>), dim3(fgrid), fb, fl, fs, a);
        else if (bf16 && Ca == 24 && Cb == 24 && Cout == 24) hipLaunchKernelGGL((pwconv_fwd_fast_kernel<24, 24, 24, true>), dim3(fgrid), fb, fl, fs, a);
        else if (bf16 && Ca == 24 && Cb == 0 && Cout == 4) hipLaunchKernelGGL((pwconv_fwd_fast_kernel<24, 0, 4, true>), dim3(fgrid), fb, fl, fs, a);
        else if (Ca == 24 && Cb == 0 && Cout == 24) hipLaunchKernelGGL((pwconv_fwd_fast_kernel<24, 0, 24>), dim3(fgrid), fb, fl, fs, a);
        else if (Ca == 24 && Cb == 24 && Cout == 24 && (pwf_waves() == 8 || pwf_waves() == 12)) {   // 2 (default) / 3 waves per SIMD
            const int nw = pwf_waves();
            int g8 = (int)((ntiles + nw - 1) / nw);
            if (g8 > 256) g8 = 256;
            static int attr8 = -1;
            if (attr8 != current_device()) {
                (void)hipFuncSetAttribute((const void *)pwconv_fwd_fast_kernel<24, 24, 24, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void *)pwconv_fwd_fast_kernel<24, 24, 24, false, 12>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attr8 = current_device();
            }
            if (nw == 8) hipLaunchKernelGGL((pwconv_fwd_fast_kernel<24, 24, 24, false, 8>), dim3(g8), dim3(512), (size_t)8 * 2 * 24 * 256, fs, a);
            else hipLaunchKernelGGL((pwconv_fwd_fast_kernel<24, 24, 24, false, 12>), dim3(g8), dim3(768), (size_t)12 * 2 * 24 * 256, fs, a);
        }
        else if (Ca == 24 && Cb == 24 && Cout == 24) hipLaunchKernelGGL((pwconv_fwd_fast_kernel<24, 24, 24>), dim3(fgrid), fb, fl, fs, a);
        else if (Ca == 24 && Cb == 0 && Cout == 4) hipLaunchKernelGGL((pwconv_fwd_fast_kernel<24, 0, 4>), dim3(fgrid), fb, fl, fs, a);
        else if (Ca == 48 && Cb == 0 && Cout == 48) hipLaunchKernelGGL((pwconv_fwd_fast_kernel<48, 0, 48>), dim3(fgrid), fb, fl, fs, a);   // composed complex mix
        else if (Ca == 12 && Cb == 0 && Cout == 12) hipLaunchKernelGGL((pwconv_fwd_fast_kernel<12, 0, 12>), dim3(fgrid), fb, fl, fs, a);
        else if (Ca == 12 && Cb == 12 && Cout == 12) hipLaunchKernelGGL((pwconv_fwd_fast_kernel<12, 12, 12>), dim3(fgrid), fb, fl, fs, a);
        else if (Ca == 12 && Cb == 0 && Cout == 4) hipLaunchKernelGGL((pwconv_fwd_fast_kernel<12, 0, 4>), dim3(fgrid), fb, fl, fs, a);
        else if (Ca == 12 && Cb == 0 && Cout == 48) hipLaunchKernelGGL((pwconv_fwd_fast_kernel<12, 0, 48>), dim3(fgrid), fb, fl, fs, a);     // a third of the attention's q / k / v projection
        else if (Ca == 12 && Cb == 0 && Cout == 144 && B == 1 && one_launch_144) hipLaunchKernelGGL((pwconv_fwd_fast_kernel<12, 0, 144>), dim3(fgrid), fb, fl, fs, a);   // the stacked projection
        else if (Ca == 48 && Cb == 0 && Cout == 12) hipLaunchKernelGGL((pwconv_fwd_fast_kernel<48, 0, 12>), dim3(fgrid), fb, fl, fs, a);     // attention output projection
        else done = false;
        if (done) {
            HNO_CHECK_LAUNCH();
            return HNO_OK;
        }
    }
    const int KCH = 64;  // input channels per launch (32 k-steps of 2)
    for (int o0 = 0; o0 < Cout; o0 += 32) {
        for (int k0 = 0; k0 < a.Cin; k0 += KCH) {
            a.o_begin = o0;
            a.k_begin = k0;
            a.k_count = a.Cin - k0 < KCH ? a.Cin - k0 : KCH;
            a.accumulate = k0 > 0;
            a.finalize = k0 + KCH >= a.Cin;
            ProfScope ps(KID_PWCONV_FWD, (hipStream_t)stream, 4.0 * B * (double)V * (a.k_count + (a.accumulate ? 2 : 1) * (Cout - o0 < 32 ? Cout - o0 : 32)));
            if (a.k_count <= 24)
                hipLaunchKernelGGL(pwconv_fwd_kernel<12>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
            else if (a.k_count <= 48)
                hipLaunchKernelGGL(pwconv_fwd_kernel<24>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
            else
                hipLaunchKernelGGL(pwconv_fwd_kernel<32>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
            HNO_CHECK_LAUNCH();
        }
    }
    return HNO_OK;
}

int pwconv_bwd_launch(const float *gy, const float *y, const float *xa, int Ca, const float *xb, int Cb,
                      const float *W, float *gxa, float *gxb, float *dW, float *dbias, void *workspace,
                      int B, int Cout, long long V, int act, int residual, void *stream, int xa_act = HNO_ACT_NONE,
                      int accumulate_gx = 0, const float *Wbr = nullptr) {   // also declared in hno_specmix.hip
    const int bf16 = (act >> 12) & 1;      // HNO_ACT_BF16 (see pwconv_fwd_launch)
    const int io16 = (act >> 13) & 1;      // HNO_ACT_IO16: gy, y, xb are bf16 tensors (fused-branch block tail only)
    act &= 0xfff;
    HNO_REQUIRE(!io16 || Wbr, "hno_pwconv_bwd: bf16 tensors are built for the fused-branch block tail only (hno_pwconv_bwd_branch)");
    HNO_REQUIRE(workspace, "hno_pwconv_bwd: workspace of hno_pwconv_bwd_workspace_bytes() is required");
    HNO_REQUIRE(gy && xa && W && dW && Ca > 0 && Cb >= 0 && B > 0 && Cout > 0 && V > 0, "hno_pwconv_bwd: bad argument");
    HNO_REQUIRE(act == HNO_ACT_NONE || y, "hno_pwconv_bwd: saved output y needed for the activation gradient");
    HNO_REQUIRE(Cb == 0 || xb, "hno_pwconv_bwd: xb is NULL but Cb > 0");
    const int Cin = Ca + Cb;
    if (V * 8 >= (1ll << 32) || ((V + 31) / 32) * B >= (1ll << 31))
        return fail(HNO_ELIMIT, "hno_pwconv_bwd: V=%lld voxels per channel exceeds the 32-bit offset range", V);
    // 144 <- 12 (the stacked q / k / v projection of HartleyMHASeg): one launch of the 144-row instantiation (round 4c) instead of three
    // 48-row launches + three slab reductions that serialise on the accumulated input gradient (3 x 15.5 us -> one launch; at 490 tiles the
    // launch has fewer workgroups than the chip has CUs, so the 512-register single-block form costs no occupancy).  HNO_PW_WIDE144=0: thirds
    static const bool wide144 = !(getenv("HNO_PW_WIDE144") && atoi(getenv("HNO_PW_WIDE144")) == 0);
    const bool one_launch_144 = wide144 && Cout == 144 && !bf16 && ((V + 31) / 32 + PWB_FAST_WAVES - 1) / PWB_FAST_WAVES <= 512;
    if (B == 1 && Ca == 12 && Cb == 0 && Cout > 48 && Cout % 48 == 0 && !residual && !Wbr && !one_launch_144) {
        // see pwconv_fwd_launch: output-channel thirds on the 48 <- 12 fast kernel; the input gradient accumulates over the thirds
        // each third keeps its own slab region (a deferred slab reduction reads it after the next third has run): the workspace of
        // (12, Cout) is 1024 slabs of Cout (12 + 1) floats = Cout / 48 regions of 1024 slabs of 48 (12 + 1)
        for (int o0 = 0; o0 < Cout; o0 += 48) {
            float *ws_k = (float *)workspace + (size_t)(o0 / 48) * 1024 * (48 * Ca + 48);
            const int rc = pwconv_bwd_launch(gy + (size_t)o0 * V, y ? y + (size_t)o0 * V : nullptr, xa, Ca, nullptr, 0, W + (size_t)o0 * Ca, gxa,
                                             nullptr, dW + (size_t)o0 * Ca, dbias ? dbias + o0 : nullptr, ws_k, 1, 48, V,
                                             act | (bf16 ? HNO_ACT_BF16 : 0), 0, stream, xa_act, o0 > 0 ? (accumulate_gx | 1) : accumulate_gx, nullptr);
            if (rc) return rc;
        }
        return HNO_OK;
    }
    PwBwdArgs a;
    a.gy = gy; a.y = y ? y : gy; a.xa = xa; a.xb = xb; a.W = W;
    a.gxa = gxa; a.gxb = gxb; a.dW = dW; a.dbias = dbias; a.partials = (float *)workspace;
    a.Ca = Ca; a.Cb = Cb; a.Cin = Cin; a.Cout = Cout; a.B = B; a.V = (unsigned)V; a.act = act;
    a.residual = residual;
    a.dbg = debug_flags() | (pw_stagger() << 24);
    a.xa_act = xa_act;
    a.accum = accumulate_gx;
    a.Wbr = Wbr;
    a.s_gy = (long long)Cout * V; a.s_xa = (long long)Ca * V; a.s_xb = (long long)Cb * V;
    a.s_gxa = (long long)Ca * V; a.s_gxb = (long long)Cb * V; a.ldw = Cin;
    const long long ntiles = ((V + 31) / 32) * B;
    int grid = grid_for(ntiles, 4);
    if (grid > 1024) grid = 1024;  // fewer blocks -> fewer dW atomics
    if (residual && (Cout > 32 || Cin > 64 || Cb > 0))
        return fail(HNO_ELIMIT, "hno_specmix_shared_bwd: the residual (W + I) form supports C <= 32 channels");
    hipStream_t s = (hipStream_t)stream;
    const int nslab = Cout * Cin + Cout;
    if (Ca % 4 == 0 && Cb % 4 == 0) {  // exact-size fast paths
        bool done = true;
        int NW = PWB_FAST_WAVES;
        const bool s2424 = Ca == 24 && Cb == 24 && Cout == 24;
        if (Wbr) {
            // fused conv branch: dW is ONE flat buffer [dW (Cout x Cin) | dbias (Cout) | dWbr (Ca x Cb) | dbbr (Ca)]
            if (!s2424) return fail(HNO_ELIMIT, "hno_pwconv_bwd_branch: only the 24 + 24 -> 24 block shape is built (got %d + %d -> %d)", Ca, Cb, Cout);
            HNO_REQUIRE(xa_act != HNO_ACT_NONE && gxa && gxb && !residual, "hno_pwconv_bwd_branch: needs the activation of xa and both input gradients");
            long long fgb = (ntiles + PWB_FAST_WAVES - 1) / PWB_FAST_WAVES;
            if (fgb > 512) fgb = 512;
            HNO_REQUIRE(!io16 || (bf16 && V % 32 == 0 && act != HNO_ACT_NONE), "hno_pwconv_bwd_branch: bf16 tensors need bf16 arithmetic, an activation and V %% 32 == 0");
            auto kern = io16 ? pwconv_bwd_fast_kernel<24, 24, 24, PWB_FAST_WAVES, 1, true, true>
                             : bf16 ? pwconv_bwd_fast_kernel<24, 24, 24, PWB_FAST_WAVES, 1, true> : pwconv_bwd_fast_kernel<24, 24, 24, PWB_FAST_WAVES, 1>;
            static int battr = -1;
            if (battr != current_device()) {
                (void)hipFuncSetAttribute((const void *)pwconv_bwd_fast_kernel<24, 24, 24, PWB_FAST_WAVES, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void *)pwconv_bwd_fast_kernel<24, 24, 24, PWB_FAST_WAVES, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void *)pwconv_bwd_fast_kernel<24, 24, 24, PWB_FAST_WAVES, 1, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                battr = current_device();
            }
            const int nb = Cout * Cin + Cout + Ca * Cb + Ca;
            {
                // algorithmic bytes: gy, y (out) and xb are 2-byte tensors with bf16 activations in memory
                ProfScope ps(KID_PWCONV_BWD, s, B * (double)V * (io16 ? 2.0 * (2 * Cout + Cb) + 4.0 * (2 * Ca + Cb)
                                                                      : 4.0 * ((act != HNO_ACT_NONE ? 2 : 1) * Cout + Cin + Ca + Cb)));
                hipLaunchKernelGGL(kern, dim3((int)fgb), dim3(64 * PWB_FAST_WAVES), (pwb_fast_lds_bytes<24, 24, 24, 1>(PWB_FAST_WAVES)), s, a);
            }
            HNO_CHECK_LAUNCH();
            return reduce_partials_launch(a.partials, (int)fgb, nb, dW, nb, nullptr, s);
        }
        if (s2424 && (a.dbg & 64)) NW = 8;
        long long fg = (ntiles + NW - 1) / NW;
        static const int bcap_env = getenv("HNO_PWB_GRID_CAP") ? atoi(getenv("HNO_PWB_GRID_CAP")) : 0;      // A/B aid
        if (fg > (bcap_env > 0 ? bcap_env : 512)) fg = bcap_env > 0 ? bcap_env : 512;   // two blocks per CU
        if (debug_grid()) fg = debug_grid();
        static int attr_done = -1;
        if (attr_done != current_device()) {
            (void)hipFuncSetAttribute((const void *)pwconv_bwd_fast_kernel<24, 24, 24>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)pwconv_bwd_fast_kernel<24, 24, 24, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)pwconv_bwd_fast_kernel<24, 24, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)pwconv_bwd_fast_kernel<4, 24, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)pwconv_bwd_fast_kernel<48, 48, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)pwconv_bwd_fast_kernel<144, 12, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)pwconv_bwd_fast_kernel<24, 24, 0, PWB_FAST_WAVES, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)pwconv_bwd_fast_kernel<24, 24, 24, PWB_FAST_WAVES, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_done = current_device();
        }
        {
            ProfScope ps(KID_PWCONV_BWD, s, 4.0 * B * (double)V * ((act != HNO_ACT_NONE ? 2 : 1) * Cout + Cin + (gxa ? Ca : 0) + (gxb ? Cb : 0)));
            const dim3 g((int)fg), blk(64 * NW);
            if (bf16 && NW == PWB_FAST_WAVES && Ca == 24 && Cb == 0 && Cout == 24)
                hipLaunchKernelGGL((pwconv_bwd_fast_kernel<24, 24, 0, PWB_FAST_WAVES, 0, true>), g, blk, (pwb_fast_lds_bytes<24, 24, 0, 0>(NW)), s, a);
            else if (bf16 && NW == PWB_FAST_WAVES && s2424)
                hipLaunchKernelGGL((pwconv_bwd_fast_kernel<24, 24, 24, PWB_FAST_WAVES, 0, true>), g, blk, (pwb_fast_lds_bytes<24, 24, 24, 0>(NW)), s, a);
            else if (Ca == 24 && Cb == 0 && Cout == 24) hipLaunchKernelGGL((pwconv_bwd_fast_kernel<24, 24, 0>), g, blk, (pwb_fast_lds_bytes<24, 24, 0, 0>(NW)), s, a);
            else if (s2424 && NW == 8) hipLaunchKernelGGL((pwconv_bwd_fast_kernel<24, 24, 24, 8>), g, blk, (pwb_fast_lds_bytes<24, 24, 24, 0>(NW)), s, a);
            else if (s2424) hipLaunchKernelGGL((pwconv_bwd_fast_kernel<24, 24, 24>), g, blk, (pwb_fast_lds_bytes<24, 24, 24, 0>(NW)), s, a);
            else if (Ca == 24 && Cb == 0 && Cout == 4) hipLaunchKernelGGL((pwconv_bwd_fast_kernel<4, 24, 0>), g, blk, (pwb_fast_lds_bytes<4, 24, 0, 0>(NW)), s, a);
            else if (Ca == 48 && Cb == 0 && Cout == 48) hipLaunchKernelGGL((pwconv_bwd_fast_kernel<48, 48, 0>), g, blk, (pwb_fast_lds_bytes<48, 48, 0, 0>(NW)), s, a);   // composed complex mix
            else if (Ca == 12 && Cb == 0 && Cout == 12) hipLaunchKernelGGL((pwconv_bwd_fast_kernel<12, 12, 0>), g, blk, (pwb_fast_lds_bytes<12, 12, 0, 0>(NW)), s, a);   // HartleyMHASeg
            else if (Ca == 12 && Cb == 12 && Cout == 12) hipLaunchKernelGGL((pwconv_bwd_fast_kernel<12, 12, 12>), g, blk, (pwb_fast_lds_bytes<12, 12, 12, 0>(NW)), s, a);
            else if (Ca == 12 && Cb == 0 && Cout == 4) hipLaunchKernelGGL((pwconv_bwd_fast_kernel<4, 12, 0>), g, blk, (pwb_fast_lds_bytes<4, 12, 0, 0>(NW)), s, a);
            else if (Ca == 12 && Cb == 0 && Cout == 48) hipLaunchKernelGGL((pwconv_bwd_fast_kernel<48, 12, 0>), g, blk, (pwb_fast_lds_bytes<48, 12, 0, 0>(NW)), s, a);
            else if (Ca == 12 && Cb == 0 && Cout == 144 && B == 1 && one_launch_144) hipLaunchKernelGGL((pwconv_bwd_fast_kernel<144, 12, 0>), g, blk, (pwb_fast_lds_bytes<144, 12, 0, 0>(NW)), s, a);
            else if (Ca == 48 && Cb == 0 && Cout == 12) hipLaunchKernelGGL((pwconv_bwd_fast_kernel<12, 48, 0>), g, blk, (pwb_fast_lds_bytes<12, 48, 0, 0>(NW)), s, a);
            else done = false;
        }
        if (done) {
            HNO_CHECK_LAUNCH();
            return reduce_partials_launch(a.partials, (int)fg, nslab, dW, Cout * Cin, dbias, s);
        }
    }
    if (Wbr) return fail(HNO_ELIMIT, "hno_pwconv_bwd_branch: only the 24 + 24 -> 24 block shape is built (got %d + %d -> %d)", Ca, Cb, Cout);
    // generic path: any channel counts, processed as (<= 32 output) x (<= 64 input) channel blocks of the
    // concatenated input; input gradients accumulate over the output blocks, each weight block is reduced
    // from its own slabs into the strided sub-block of dW
    struct Seg { const float *x; float *gx; int C; int off; };
    const Seg segs[2] = {{xa, gxa, Ca, 0}, {xb, gxb, Cb, Ca}};
    bool first_block = true;
    for (int sgi = 0; sgi < 2; ++sgi) {
        const Seg &sg = segs[sgi];
        for (int i0 = 0; i0 < sg.C; i0 += 64) {
            const int ci = sg.C - i0 < 64 ? sg.C - i0 : 64;
            for (int o0 = 0; o0 < Cout; o0 += 32) {
                const int co = Cout - o0 < 32 ? Cout - o0 : 32;
                PwBwdArgs c = a;
                c.gy = gy + (size_t)o0 * V;
                c.y = (y ? y : gy) + (size_t)o0 * V;
                c.s_gy = (long long)Cout * V;
                c.xa = sg.x + (size_t)i0 * V;
                c.s_xa = (long long)sg.C * V;
                c.xb = nullptr; c.s_xb = 0;
                c.Ca = ci; c.Cb = 0; c.Cin = ci; c.Cout = co;
                c.W = W + (size_t)o0 * Cin + sg.off + i0;
                c.ldw = Cin;
                c.gxa = sg.gx ? sg.gx + (size_t)i0 * V : nullptr;
                c.s_gxa = (long long)sg.C * V;
                c.gxb = nullptr; c.s_gxb = 0;
                c.accum = ((accumulate_gx & (sgi == 0 ? 1 : 2)) || o0 > 0) ? 1 : 0;
                c.xa_act = (sgi == 0) ? xa_act : HNO_ACT_NONE;
                c.dbias = (dbias && first_block) ? dbias : nullptr;   // kernel only needs non-null to compute it
                const int ich = ci <= 32 ? 1 : 2;
                const size_t lds = sizeof(float) * 4 * (32 + ich * 32) * PWB_LD;
                {
                    ProfScope ps(KID_PWCONV_BWD, s, 4.0 * B * (double)V * ((act != HNO_ACT_NONE ? 2 : 1) * co + ci + (c.gxa ? ci : 0)));
                    if (co <= 8) {
                        if (ich == 1) hipLaunchKernelGGL((pwconv_bwd_kernel<4, 1>), dim3(grid), dim3(256), lds, s, c);
                        else hipLaunchKernelGGL((pwconv_bwd_kernel<4, 2>), dim3(grid), dim3(256), lds, s, c);
                    } else if (co <= 24) {
                        if (ich == 1) hipLaunchKernelGGL((pwconv_bwd_kernel<12, 1>), dim3(grid), dim3(256), lds, s, c);
                        else hipLaunchKernelGGL((pwconv_bwd_kernel<12, 2>), dim3(grid), dim3(256), lds, s, c);
                    } else {
                        if (ich == 1) hipLaunchKernelGGL((pwconv_bwd_kernel<16, 1>), dim3(grid), dim3(256), lds, s, c);
                        else hipLaunchKernelGGL((pwconv_bwd_kernel<16, 2>), dim3(grid), dim3(256), lds, s, c);
                    }
                }
                HNO_CHECK_LAUNCH();
                // the bias gradient of an output block is the same in every input block: take it from the first
                const bool want_bias = dbias && sgi == 0 && i0 == 0;
                int rc = reduce_partials_launch(a.partials, grid, co * ci + co, dW + (size_t)o0 * Cin + sg.off + i0, co * ci,
                                                want_bias ? dbias + o0 : nullptr, s, ci, Cin, false);   // the slabs are reused by the next block
                if (rc) return rc;
                first_block = false;
            }
        }
    }
    return HNO_OK;
}

// ---- complex shared mix of the Fourier operator as ONE real pointwise conv -----------------------------------
// [Yr; Yi] = [[Wr, -Wi], [Wi, Wr]] [Xr; Xi] on the [re | im] channel layout of the kept half spectrum
// (nets/fourier_operator.py:164-172 'oi,bi...->bo...' with complex weights).
__global__ void cmix_compose_kernel(const float *__restrict__ wr, const float *__restrict__ wi, float *__restrict__ w2, int Co, int Ci) {
    const int n = 4 * Co * Ci;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const int o = e / (2 * Ci), i = e - o * 2 * Ci;
        const int oo = o < Co ? o : o - Co, ii = i < Ci ? i : i - Ci;
        const float r = wr[oo * Ci + ii], im = wi[oo * Ci + ii];
        w2[e] = (o < Co) == (i < Ci) ? r : (o < Co ? -im : im);
    }
}

// the same for up to 64 blocks in one launch: blockIdx.y = block of the model (FNOSeg: 24 launches of ~4 us per forward pass -> 1)
struct CmixBatch {
    const float *wr[64], *wi[64];
};
__global__ void cmix_compose_multi_kernel(CmixBatch b, float *__restrict__ w2_all, int Co, int Ci) {
    const int n = 4 * Co * Ci;
    const float *wr = b.wr[blockIdx.y], *wi = b.wi[blockIdx.y];
    float *w2 = w2_all + (size_t)blockIdx.y * n;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const int o = e / (2 * Ci), i = e - o * 2 * Ci;
        const int oo = o < Co ? o : o - Co, ii = i < Ci ? i : i - Ci;
        const float r = wr[oo * Ci + ii], im = wi[oo * Ci + ii];
        w2[e] = (o < Co) == (i < Ci) ? r : (o < Co ? -im : im);
    }
}

// dWr = dW2[re, re] + dW2[im, im];  dWi = dW2[im, re] - dW2[re, im]
__global__ void cmix_split_kernel(const float *__restrict__ dw2, float *__restrict__ dwr, float *__restrict__ dwi, int Co, int Ci) {
    const int n = Co * Ci;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const int o = e / Ci, i = e - o * Ci;
        const float rr = dw2[(size_t)o * 2 * Ci + i], ri = dw2[(size_t)o * 2 * Ci + Ci + i];
        const float ir = dw2[(size_t)(o + Co) * 2 * Ci + i], ii = dw2[(size_t)(o + Co) * 2 * Ci + Ci + i];
        dwr[e] = rr + ii;
        dwi[e] = ir - ri;
    }
}

}  // namespace hno

using namespace hno;

extern "C" int hno_pwconv_fwd_branch(const float *s_in, const float *x, const float *Wbr, const float *bbr, const float *W,
                                     const float *bias, float *y, float *out, int B, int Ca, int Cb, int Cout, long long V,
                                     int act, void *stream) {
    const int bf16 = (act >> 12) & 1, io16 = (act >> 13) & 1;      // HNO_ACT_BF16, HNO_ACT_IO16 (x and out are bf16 tensors)
    act &= 0xfff;
    HNO_REQUIRE(s_in && x && Wbr && W && y && out && B > 0 && V > 0, "hno_pwconv_fwd_branch: bad argument");
    HNO_REQUIRE(!io16 || (bf16 && V % 32 == 0 && ((size_t)x & 63) == 0 && ((size_t)out & 63) == 0),
                "hno_pwconv_fwd_branch: bf16 tensors need bf16 arithmetic, V %% 32 == 0 and 64-byte aligned rows");
    if (!(Ca == 24 && Cb == 24 && Cout == 24))
        return fail(HNO_ELIMIT, "hno_pwconv_fwd_branch: only the 24 + 24 -> 24 block shape is built (got %d + %d -> %d)", Ca, Cb, Cout);
    if (V * 8 >= (1ll << 32) || ((V + 31) / 32) * B >= (1ll << 31))
        return fail(HNO_ELIMIT, "hno_pwconv_fwd_branch: V=%lld voxels per channel exceeds the 32-bit offset range", V);
    PwBranchArgs a;
    a.s = s_in; a.x = x; a.Wbr = Wbr; a.bbr = bbr; a.W = W; a.bias = bias; a.y = y; a.out = out;
    a.B = B; a.V = (unsigned)V; a.act = act;
    const long long ntiles = ((V + 31) / 32) * B;
    long long grid = (ntiles + PWF_DMA_WAVES - 1) / PWF_DMA_WAVES;
    if (grid > 512) grid = 512;   // two 4-wave blocks per CU (each wave one tile ahead through its LDS ring; 256: 62 us, 512: 57)
    if (debug_grid()) grid = debug_grid();
    hipStream_t fs = (hipStream_t)stream;
    ProfScope ps(KID_PWCONV_FWD, fs, B * (double)V * (4.0 * 2 * Ca + (io16 ? 2.0 : 4.0) * (Cb + Cout)));
    const size_t fl = (size_t)PWF_DMA_WAVES * 2 * (12 + 12) * 256;
    if (io16) hipLaunchKernelGGL((pwconv_fwd_branch_kernel<24, 24, 24, true, true>), dim3((int)grid), dim3(64 * PWF_DMA_WAVES), fl, fs, a);
    else if (bf16) hipLaunchKernelGGL((pwconv_fwd_branch_kernel<24, 24, 24, true>), dim3((int)grid), dim3(64 * PWF_DMA_WAVES), fl, fs, a);
    else hipLaunchKernelGGL((pwconv_fwd_branch_kernel<24, 24, 24>), dim3((int)grid), dim3(64 * PWF_DMA_WAVES), fl, fs, a);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

// xi = act(Wc [u ; t] + bc), xn = act(Wm [xi ; k] + bm): conv_concat of one HNO-XS block and mapping_conv of the next in one pass
// (pwconv_fwd_chain_kernel).  All tensors (B, 24, V) fp32 (V = the channel stride of channel-padded activations); Wc, Wm (24, 48).
extern "C" int hno_pwconv_fwd_chain_supported(int C, int C2, int has_k) { return C == 24 && ((C2 == 24 && has_k) || (C2 == 4 && !has_k)); }
// C2 = 24 with k (mapping_conv of the next block, activation act2 = act) or C2 = 4 without k (conv_out behind the last block: Wm (4, 24),
// bm NULL, act2 none); xn (B, C2, V)
extern "C" int hno_pwconv_fwd_chain(const float *u, const float *t, const float *k, const float *Wc, const float *bc, const float *Wm,
                                    const float *bm, float *xi, float *xn, int B, int C, int C2, long long V, int act, int act2, void *stream) {
    HNO_REQUIRE(u && t && Wc && Wm && xn && B > 0 && V > 0, "hno_pwconv_fwd_chain: bad argument");   // xi NULL: not stored (inference)
    if (!hno_pwconv_fwd_chain_supported(C, C2, k != nullptr))
        return fail(HNO_ELIMIT, "hno_pwconv_fwd_chain: 24 -> 24 (with a second input) and 24 -> 4 (without) are built (got %d -> %d)", C, C2);
    if (V * 8 >= (1ll << 32) || ((V + 31) / 32) * B >= (1ll << 31))
        return fail(HNO_ELIMIT, "hno_pwconv_fwd_chain: V=%lld voxels per channel exceeds the 32-bit offset range", V);
    PwChainArgs a;
    a.u = u; a.t = t; a.k = k; a.Wc = Wc; a.bc = bc; a.Wm = Wm; a.bm = bm; a.xi = xi; a.xn = xn;
    a.B = B; a.V = (unsigned)V; a.act = act & 0xfff; a.act2 = act2 & 0xfff; a.dbg = debug_flags() | (pw_stagger() << 24);
    const long long ntiles = ((V + 31) / 32) * B;
    // two 4-wave workgroups per CU (63.3 us at 2 x 24 x 65^3) or one of 8 waves (67.7 us): HNO_PWCHAIN_WAVES=8 selects the latter (A/B)
    static const int nw_env = getenv("HNO_PWCHAIN_WAVES") ? atoi(getenv("HNO_PWCHAIN_WAVES")) : 4;
    const int nw = (nw_env == 8 && C2 == 24) ? 8 : 4;
    long long grid = (ntiles + nw - 1) / nw;
    const long long cap = nw == 4 ? 512 : 256;
    if (grid > cap) grid = cap;
    if (debug_grid()) grid = debug_grid();
    hipStream_t fs = (hipStream_t)stream;
    ProfScope ps(KID_PWCONV_FWD, fs, 4.0 * B * (double)V * ((k ? 3 : 2) * C + (xi ? C : 0) + C2));
    const size_t fl = (size_t)nw * 2 * (k ? 36 : 24) * 256;
    static int attr = -1;
    if (attr != current_device()) {
        (void)hipFuncSetAttribute((const void *)pwconv_fwd_chain_kernel<24, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)pwconv_fwd_chain_kernel<24, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)pwconv_fwd_chain_kernel<24, 4, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = current_device();
    }
    if (C2 == 4) hipLaunchKernelGGL((pwconv_fwd_chain_kernel<24, 4, 4, false>), dim3((int)grid), dim3(256), fl, fs, a);
    else if (nw == 8) hipLaunchKernelGGL((pwconv_fwd_chain_kernel<24, 8>), dim3((int)grid), dim3(512), fl, fs, a);
    else hipLaunchKernelGGL((pwconv_fwd_chain_kernel<24, 4>), dim3((int)grid), dim3(256), fl, fs, a);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

// backward of hno_pwconv_fwd_chain in one pass (pwconv_bwd_chain_kernel): gn = gradient of xn; -> gu, gt, gk and
// grads = [dWc (C, 2C) | dbc (C) | dWm (C2, CIN2) | dbm (C2)] (one flat buffer: the parameters' order in the model).  xa_act: activation whose output u is (its derivative is
// applied to gu), as hno_pwconv_bwd's.  workspace: hno_pwconv_bwd_chain_workspace_bytes(C).  bit 8 of xa_act: defer the slab reduction.
extern "C" size_t hno_pwconv_bwd_chain_workspace_bytes(int C) { return sizeof(float) * 512 * 2 * ((size_t)C * 2 * C + C); }
// grads: C2 = 24: [dWc (24, 48) | dbc (24) | dWm (24, 48) | dbm (24)];  C2 = 4 (k, gk NULL): [dWc | dbc | dWm (4, 24)] -- dbm is not written
// (conv_out has no bias): the buffer ends behind dWm
extern "C" int hno_pwconv_bwd_chain(const float *gn, const float *xn, const float *xi, const float *k, const float *u, const float *t,
                                    const float *Wm, const float *Wc, float *gu, float *gt, float *gk, float *grads, void *workspace,
                                    int B, int C, int C2, long long V, int act, int act2, int xa_act, void *stream) {
    HNO_REQUIRE(gn && xi && u && t && Wm && Wc && gu && gt && grads && workspace && B > 0 && V > 0, "hno_pwconv_bwd_chain: bad argument");
    HNO_REQUIRE((k != nullptr) == (gk != nullptr), "hno_pwconv_bwd_chain: k and gk go together");
    HNO_REQUIRE((act2 & 0xfff) == HNO_ACT_NONE || xn, "hno_pwconv_bwd_chain: xn needed for the second layer's activation gradient");
    if (!hno_pwconv_fwd_chain_supported(C, C2, k != nullptr))
        return fail(HNO_ELIMIT, "hno_pwconv_bwd_chain: 24 -> 24 (with a second input) and 24 -> 4 (without) are built (got %d -> %d)", C, C2);
    if (V * 8 >= (1ll << 32) || ((V + 31) / 32) * B >= (1ll << 31))
        return fail(HNO_ELIMIT, "hno_pwconv_bwd_chain: V=%lld voxels per channel exceeds the 32-bit offset range", V);
    const int defer_bit = (xa_act >> 8) & 1;
    PwChainBwdArgs a;
    a.gn = gn; a.xn = xn ? xn : gn; a.xi = xi; a.k = k; a.u = u; a.t = t; a.Wm = Wm; a.Wc = Wc; a.gu = gu; a.gt = gt; a.gk = gk;
    a.partials = (float *)workspace; a.B = B; a.V = (unsigned)V; a.act = act & 0xfff; a.act2 = act2 & 0xfff; a.xa_act = xa_act & 0xff;
    a.dbg = debug_flags() | (pw_stagger() << 24);
    constexpr int NW = 4;
    static const int slots_env = getenv("HNO_PWCHAIN_SLOTS") ? atoi(getenv("HNO_PWCHAIN_SLOTS")) : 1;     // A/B: 2 = double-buffered ring
    const int slots = slots_env == 2 ? 2 : 1;
    const long long ntiles = ((V + 31) / 32) * B, ngroups = (ntiles + NW - 1) / NW;
    const int cap = slots == 1 ? 512 : 256;                   // two 4-wave workgroups per CU (17 KB of LDS per wave) / one (30 KB)
    int grid = (int)(ngroups < cap ? ngroups : cap);
    hipStream_t s = (hipStream_t)stream;
    const int nt = k ? 4 : 3, cin2 = k ? 2 * C : C;
    const size_t per_wave = (size_t)(32 * PWB_LD + slots * nt * 12 * PWB_XP) * sizeof(float);
    const int n = C2 * cin2 + C2 + C * 2 * C + C;
    size_t ldsb = NW * per_wave;
    if (ldsb < (size_t)NW * n * sizeof(float)) ldsb = (size_t)NW * n * sizeof(float);
    static int attr = -1;
    if (attr != current_device()) {
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)pwconv_bwd_chain_kernel<24, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)pwconv_bwd_chain_kernel<24, NW, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)pwconv_bwd_chain_kernel<24, NW, 24, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)pwconv_bwd_chain_kernel<24, NW, 4, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = current_device();
    }
    {
        ProfScope ps(KID_PWCONV_BWD, s, 4.0 * B * (double)V * ((k ? 7 : 5) * C + 2 * C2));
        if (C2 == 4 && slots == 1) hipLaunchKernelGGL((pwconv_bwd_chain_kernel<24, NW, 4, false, 1>), dim3(grid), dim3(64 * NW), ldsb, s, a);
        else if (C2 == 4) hipLaunchKernelGGL((pwconv_bwd_chain_kernel<24, NW, 4, false>), dim3(grid), dim3(64 * NW), ldsb, s, a);
        else if (slots == 1) hipLaunchKernelGGL((pwconv_bwd_chain_kernel<24, NW, 24, true, 1>), dim3(grid), dim3(64 * NW), ldsb, s, a);
        else hipLaunchKernelGGL((pwconv_bwd_chain_kernel<24, NW>), dim3(grid), dim3(64 * NW), ldsb, s, a);
        HNO_CHECK_LAUNCH();
    }
    const int prev = hno_set_defer_reduce(0);
    struct Restore { int v; ~Restore() { hno_set_defer_reduce(v); } } restore{prev};
    hno_set_defer_reduce(defer_bit ? 1 : prev);
    return reduce_partials_launch(a.partials, grid, n, grads, k ? n : n - C2, nullptr, s);
}

extern "C" size_t hno_pwconv_bwd_branch_workspace_bytes(int Ca, int Cb, int Cout) {
    return sizeof(float) * 512 * ((size_t)Cout * (Ca + Cb) + Cout + (size_t)Ca * Cb + Ca);
}

extern "C" int hno_pwconv_bwd_branch(const float *gy, const float *y, const float *xa, int Ca, const float *xb, int Cb,
                                     const float *W, const float *Wbr, float *p_out, float *gxb, float *dflat, void *workspace,
                                     int B, int Cout, long long V, int act, int xa_act, void *stream) {
    HNO_REQUIRE(Wbr && dflat, "hno_pwconv_bwd_branch: null pointer");
    // bit 8 of xa_act: record the slab reduction for hno_flush_reduces
    const int prev = hno_set_defer_reduce(0);
    struct Restore { int v; ~Restore() { hno_set_defer_reduce(v); } } restore{prev};
    hno_set_defer_reduce(((xa_act >> 8) & 1) ? 1 : prev);
    return pwconv_bwd_launch(gy, y, xa, Ca, xb, Cb, W, p_out, gxb, dflat, nullptr, workspace, B, Cout, V, act, 0, stream, xa_act & 0xff, 0, Wbr);
}

// n <= 64 weight pairs of equal shape -> w2_all (n, 2Co, 2Ci) in one launch
extern "C" int hno_cmix_compose_multi(const float *const *w_real, const float *const *w_imag, float *w2_all, int n, int Co, int Ci, void *stream) {
    HNO_REQUIRE(w_real && w_imag && w2_all && n > 0 && n <= 64 && Co > 0 && Ci > 0, "hno_cmix_compose_multi: bad argument (1 <= n <= 64)");
    CmixBatch b = {};
    for (int i = 0; i < n; ++i) {
        HNO_REQUIRE(w_real[i] && w_imag[i], "hno_cmix_compose_multi: weight %d is NULL", i);
        b.wr[i] = w_real[i];
        b.wi[i] = w_imag[i];
    }
    hipLaunchKernelGGL(cmix_compose_multi_kernel, dim3(ceil_div(4 * Co * Ci, 256), n), dim3(256), 0, (hipStream_t)stream, b, w2_all, Co, Ci);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

extern "C" int hno_cmix_compose(const float *w_real, const float *w_imag, float *w2, int Co, int Ci, void *stream) {
    HNO_REQUIRE(w_real && w_imag && w2 && Co > 0 && Ci > 0, "hno_cmix_compose: bad argument");
    hipLaunchKernelGGL(cmix_compose_kernel, dim3(ceil_div(4 * Co * Ci, 256)), dim3(256), 0, (hipStream_t)stream, w_real, w_imag, w2, Co, Ci);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

// defer != 0: when deferred slab reductions are on (hno_set_defer_reduce) the split is recorded and runs in hno_flush_reduces, behind the
// reduction that writes dw2 (which must then have been called with its own defer bit: hno_spec_mid_fourier_bwd, bit 8 of w_fwd)
extern "C" int hno_cmix_split_grad_ex(const float *dw2, float *dw_real, float *dw_imag, int Co, int Ci, int defer, void *stream) {
    HNO_REQUIRE(dw2 && dw_real && dw_imag && Co > 0 && Ci > 0, "hno_cmix_split_grad: bad argument");
    if (defer) {
        const int prev = hno_set_defer_reduce(1);
        const bool rec = cmix_split_defer(dw2, dw_real, dw_imag, Co, Ci, true);
        hno_set_defer_reduce(prev);
        if (rec) return HNO_OK;
    }
    return hno_cmix_split_grad(dw2, dw_real, dw_imag, Co, Ci, stream);
}

extern "C" int hno_cmix_split_grad(const float *dw2, float *dw_real, float *dw_imag, int Co, int Ci, void *stream) {
    HNO_REQUIRE(dw2 && dw_real && dw_imag && Co > 0 && Ci > 0, "hno_cmix_split_grad: bad argument");
    hipLaunchKernelGGL(cmix_split_kernel, dim3(ceil_div(Co * Ci, 256)), dim3(256), 0, (hipStream_t)stream, dw2, dw_real, dw_imag, Co, Ci);
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

extern "C" int hno_pwconv_fwd(const float *xa, int Ca, const float *xb, int Cb, const float *W, const float *bias,
                              float *y, int B, int Cout, long long V, int act, void *stream) {
    return pwconv_fwd_launch(xa, Ca, xb, Cb, W, bias, y, B, Cout, V, act, 0, stream);
}

extern "C" size_t hno_pwconv_bwd_workspace_bytes(int Cin, int Cout) {
    return sizeof(float) * 1024 * ((size_t)Cout * Cin + Cout);  // one slab per block, <= 1024 blocks
}

extern "C" int hno_pwconv_bwd(const float *gy, const float *y, const float *xa, int Ca, const float *xb, int Cb,
                              const float *W, float *gxa, float *gxb, float *dW, float *dbias, void *workspace,
                              int B, int Cout, long long V, int act, int xa_act, int accumulate_gx, void *stream) {
    // bit 8 of accumulate_gx: record this call's slab reduction for hno_flush_reduces (per-call form of hno_set_defer_reduce)
    const int prev = hno_set_defer_reduce(0);
    struct Restore { int v; ~Restore() { hno_set_defer_reduce(v); } } restore{prev};
    hno_set_defer_reduce(((accumulate_gx >> 8) & 1) ? 1 : prev);
    return pwconv_bwd_launch(gy, y, xa, Ca, xb, Cb, W, gxa, gxb, dW, dbias, workspace, B, Cout, V, act, 0, stream, xa_act,
                             accumulate_gx & 0xff);
}
