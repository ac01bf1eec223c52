// Shared device/host helpers for libhno (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/hno.h"

namespace hno {

// ------------------------------------------------------------------ error plumbing
void set_error(const std::string &msg);
int fail(int code, const char *fmt, ...);

#define HNO_CHECK_HIP(expr)                                                                    \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return ::hno::fail(HNO_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                               __FILE__, __LINE__);                                            \
    } while (0)

#define HNO_CHECK_LAUNCH()                                                                      \
    do {                                                                                        \
        hipError_t _e = hipGetLastError();                                                      \
        if (_e != hipSuccess)                                                                   \
            return ::hno::fail(HNO_EHIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), \
                               __FILE__, __LINE__);                                             \
    } while (0)

#define HNO_REQUIRE(cond, ...)                               \
    do {                                                     \
        if (!(cond)) return ::hno::fail(HNO_EINVAL, __VA_ARGS__); \
    } while (0)

// ------------------------------------------------------------------ per-kernel HIP-event profiler
// Off by default (zero cost beyond one branch).  hno_profile_begin() arms it; every kernel launch
// wrapped in a ProfScope is then bracketed by two hipEventRecord calls on ITS OWN stream.
enum KernelId {
    KID_DHT_FWD_PLANE = 0, KID_DHT_FWD_D, KID_DHT_INV_D, KID_DHT_INV_PLANE, KID_PWCONV_FWD, KID_PWCONV_BWD,
    KID_CONV_K2S2_FWD, KID_CONV_K2S2_BWD, KID_UPSOFTMAX_FWD, KID_UPSOFTMAX_BWD, KID_LOSS_STATS, KID_LOSS_FINALIZE,
    KID_LOSS_BWD, KID_LABELS, KID_SPECMIX_FWD, KID_SPECMIX_BWD, KID_REDUCE_PARTIALS, KID_UPSOFTMAX_BWD_D, KID_BMM, KID_PERMODE_FWD, KID_PERMODE_BWD, KID_CONV3D_GEMM,
    KID_CONV3D_WGRAD, KID_GROUPNORM, KID_RESAMPLE, KID_CB_CONV, KID_CB_WGRAD, KID_CB_GN, KID_HMHA, KID_SPEC_MID_FWD, KID_SPEC_MID_BWD, KID_COUNT
};
struct ProfScope {
    int slot;
    hipStream_t stream;
    ProfScope(int kernel_id, hipStream_t s, double algorithmic_bytes = 0.0);
    ~ProfScope();
};

// Weight-gradient partial sums: every block writes ONE slab of `n` floats (already reduced over
// its waves through LDS); reduce_partials_launch sums the slabs in a fixed order and WRITES the
// result to dst.  No float atomics -> bitwise reproducible gradients, and none of the
// same-address atomic contention that made the first version of these kernels 10x slower.
int reduce_partials_launch(const float *partials, int nblocks, int n, float *dst0, int n0, float *dst1,
                           hipStream_t stream, int cols = 0, int ldd = 0, bool allow_defer = true);   // allow_defer = false: callers that reuse the slab workspace for several rounds   // cols/ldd: dst0 is a sub-block with row stride ldd
bool defer_reduce_enabled();
// layout of the plane kernels' intermediate around the fused spectral middle (hno_dht.hip: DhtArgs.zl; HNO_MID_ZLAYOUT=0 keeps the old one)
bool mid_zlayout();
// records the real / imaginary split of a Fourier block's weight gradient for hno_flush_reduces (hno_core.hip); false: launch it now
bool cmix_split_defer(const float *dw2, float *dwr, float *dwi, int Co, int Ci, bool allow_defer);

// wave-private accumulator fragments -> one slab per block.  `scratch` is >= nwaves * n floats of LDS.
// frag(idx) semantic: each wave calls store(idx, value) for the elements it owns; all 4 waves own
// the same index set.
__device__ __forceinline__ void block_sum_to_slab(float *scratch, int n, float *slab, int tid, int nwaves = 4) {
    __syncthreads();
    for (int i = tid; i < n; i += 64 * nwaves) {
        float s = 0.f;
        for (int w = 0; w < nwaves; ++w) s += scratch[w * n + i];
        slab[i] = s;
    }
}

// in-kernel phase stamps (debug flag 64): thread 0 of block 0 stores clock64() at phase boundaries into
// the buffer returned by debug_stamp_buffer() (64 entries, allocated on first use)
long long *debug_stamp_buffer();
#define HNO_STAMP(buf, idx)                                                                         \
    do {                                                                                            \
        if ((buf) && blockIdx.x == 0 && threadIdx.x == 0 && (idx) < 64) (buf)[(idx)] = clock64();   \
    } while (0)

// Zero `n` doubles by a KERNEL.  Not hipMemsetAsync: as a memset NODE of a captured graph the clear of the loss statistics was lost
// in the first replay that followed an eager kernel launch of another library (that replay's loss came out NaN, every later replay
// was right; tools/dbg/graph_first_replay.py, DESIGN lesson 36).
int clear_doubles(double *p, int n, hipStream_t s);

// debug/ablation switches (hno_set_debug): timing-only builds of a kernel phase, results are WRONG
int debug_flags();
// bits 8..23 of the debug flags: a forced grid size for the launchers that have one (tuning sweeps); switches of other kernels live
// above bit 23 (the attention A/B switches sat at bits 14 / 15 and silently forced 64 / 128-workgroup grids elsewhere)
inline int debug_grid() { return (debug_flags() >> 8) & 0xFFFF; }
#define HNO_DBG_HM_ROUND2 (1 << 24)   // attention: the round-2 kernels instead of the shared-tile ones (A/B)
#define HNO_DBG_HM_PAIR (1 << 25)     // attention: the pair layout of the streamed tile even when T % 4 == 0 (A/B)
// the calling thread's current HIP device (-2 on error): kernel attributes are set once per device, not once per process
int current_device();

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
// Grid of a persistent (grid-stride) kernel: no more workgroups than are resident at once (occupancy from the
// runtime, cached per kernel and LDS size -- a larger grid runs a second, mostly idle wave of workgroups), and
// then the smallest grid that needs the same number of iterations, so every workgroup does equal work.
int persistent_grid(const void *kernel, int block_threads, size_t dynamic_lds, int work_items);
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

// ------------------------------------------------------------------ activations
#define HNO_SELU_ALPHA 1.6732632423543772848170429916717f
#define HNO_SELU_SCALE 1.0507009873554804934193349852946f

// e^x - 1 for x <= 0, branch-free and accurate in the RELATIVE sense (the spectral coefficients that go through
// SELU are tiny: an absolute-error-only form such as exp(x) - 1 everywhere costs 7e-4 on the HNOSeg-XS outputs):
// degree-4 near-minimax polynomial of expm1(x)/x on [-0.25, 0] (relative error 1e-7 in fp32 Horner form),
// hardware exp2 below (relative error < 3e-7).  ~10 VALU instructions; libm expm1f costs ~10x more.
__device__ __forceinline__ float neg_expm1(float x) {
    float p = 0.007513605989515781f;
    p = fmaf(p, x, 0.04149065539240837f);
    p = fmaf(p, x, 0.16665108501911163f);
    p = fmaf(p, x, 0.4999995231628418f);
    p = fmaf(p, x, 1.0f);
    const float e = __builtin_amdgcn_exp2f(x * 1.4426950408889634f) - 1.f;
    return x > -0.25f ? p * x : e;
}
// Two values at once on the packed fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: one instruction per PAIR; measured 9.5
// cycles per packed instruction and wave against 6.5 per scalar one, tools/dbg/valu_rate.hip).  Same arithmetic, operation for
// operation, as neg_expm1 above: results are bit-identical to the scalar form.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 neg_expm1_pk(f32x2 x) {
    f32x2 p = {0.007513605989515781f, 0.007513605989515781f};
    p = __builtin_elementwise_fma(p, x, f32x2{0.04149065539240837f, 0.04149065539240837f});
    p = __builtin_elementwise_fma(p, x, f32x2{0.16665108501911163f, 0.16665108501911163f});
    p = __builtin_elementwise_fma(p, x, f32x2{0.4999995231628418f, 0.4999995231628418f});
    p = __builtin_elementwise_fma(p, x, f32x2{1.0f, 1.0f});
    const f32x2 u = x * f32x2{1.4426950408889634f, 1.4426950408889634f};
    f32x2 e = {__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])};
    e = e - f32x2{1.f, 1.f};
    const f32x2 px = p * x;
    return f32x2{x[0] > -0.25f ? px[0] : e[0], x[1] > -0.25f ? px[1] : e[1]};
}
// act(x) for SELU / ELU-like activations given as (ap, aq): x > 0 ? ap x : aq expm1(x)
__device__ __forceinline__ f32x2 selu_like_pk(f32x2 x, float ap, float aq) {
    const f32x2 pos = x * f32x2{ap, ap}, neg = neg_expm1_pk(x) * f32x2{aq, aq};
    return f32x2{x[0] > 0.f ? pos[0] : neg[0], x[1] > 0.f ? pos[1] : neg[1]};
}
// the activation over N accumulator registers (N even), pairs on the packed pipe; the empty asm keeps the compiler from
// sinking the computation into a branch per register
template <int N>
__device__ __forceinline__ void selu_like_regs(float *v, float ap, float aq) {
    static_assert(N % 2 == 0, "register pairs");
#pragma unroll
    for (int r = 0; r < N; r += 2) {
        f32x2 y = selu_like_pk(f32x2{v[r], v[r + 1]}, ap, aq);
        asm volatile("" : "+v"(y));
        v[r] = y[0];
        v[r + 1] = y[1];
    }
}
// byte offsets are formed in 32 bits and added to a wave-uniform base: the loads and stores then take the (SGPR base + VGPR offset)
// address form -- 64-bit per-lane address arithmetic was a third of the head kernel's instructions
__device__ __forceinline__ float ld_off(const float *base, unsigned byte_off) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off);
}

__device__ __forceinline__ float act_apply(float x, int act) {
    if (act == HNO_ACT_SELU) return x > 0.f ? HNO_SELU_SCALE * x : (HNO_SELU_SCALE * HNO_SELU_ALPHA) * neg_expm1(x);
    if (act == HNO_ACT_ELU) return x > 0.f ? x : neg_expm1(x);
    if (act == HNO_ACT_SIGMOID) return 1.f / (1.f + __expf(-x));   // elementwise kernels only (hno_act_fwd / hno_act_bwd)
    return x;
}
// derivative expressed through the saved OUTPUT y = act(x)
__device__ __forceinline__ float act_grad_from_out(float y, int act) {
    if (act == HNO_ACT_SELU) return y > 0.f ? HNO_SELU_SCALE : y + HNO_SELU_SCALE * HNO_SELU_ALPHA;
    if (act == HNO_ACT_ELU) return y > 0.f ? 1.f : y + 1.f;
    if (act == HNO_ACT_SIGMOID) return y * (1.f - y);
    return 1.f;
}

// ------------------------------------------------------------------ LDS-DMA (global -> LDS without a VGPR destination)
// One LDS-DMA instruction: 64 lanes x 4 bytes from (wave-uniform base + per-lane byte offset) to 256 contiguous LDS bytes at `lds_dst`
// (wave-uniform).  No VGPR destination: the load is in flight while the wave multiplies the previous tile; it is retired by a counted
// s_waitcnt vmcnt(N) (hipcc does not count asm memory operations -- see dma_wait).  M0 (the destination base) is saved and restored:
// it is compiler-reserved.
__device__ __forceinline__ void dma_row_pair(const float *base, unsigned lane_byte_off, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_byte_off), "s"(base), "s"(lds_dst) : "memory");
}
// The same with the streaming hint `nt` (round 6, LESSONS 99): for rows that were written long ago and are read exactly once -- the saved
// activations a backward kernel streams.  On the forward kernels' inputs (written by the previous launch) the hint costs time; on the
// backward pointwise kernels' x rows it is worth 1.1 % of the headline step (six same-box pairs of two builds).
__device__ __forceinline__ void dma_row_pair_nt(const float *base, unsigned lane_byte_off, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_byte_off), "s"(base), "s"(lds_dst) : "memory");
}
// At most N vector-memory operations stay in flight (DMA, loads and stores share one in-order counter).  N must not exceed the number
// of operations CERTAINLY issued after the ones waited for: a larger N would let them stay in flight.
template <int N>
__device__ __forceinline__ void dma_wait() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Stagger (MI355X_MICROARCH.md, "Two waves per SIMD", item 9): waves that run the same program tile after tile tend to reach their
// matrix, vector and memory phases together; the wave in an ODD wave slot of its SIMD sleeps `units` x 1024 cycles before its first
// tile, so that one partner computes while the other waits.  No effect on results.
__device__ __forceinline__ void stagger_start(int units) {
    if (units <= 0) return;
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(hw));
    if (hw & 1u)
        for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(16);
}

// ------------------------------------------------------------------ MFMA tile engine
// v_mfma_f32_16x16x4_f32: D(16x16) += A(16x4) * B(4x16), exact fp32 FMA chain.
//   A operand: lane l holds A[row = l & 15][k = l >> 4]
//   B operand: lane l holds B[k = l >> 4][col = l & 15]
//   C/D      : lane l, reg r holds C[row = (l >> 4) * 4 + r][col = l & 15]
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// acc += A_tile(16 x 4*ksteps) * B_tile(4*ksteps x 16) with arbitrary element strides.
// a points at A[row0][k0], b points at B[k0][col0].  Operand reads are batched four k-steps at a
// time so the LDS latency overlaps the MFMA chain.
__device__ __forceinline__ f32x4 tile_mma(const float *a, int sa_r, int sa_k, const float *b, int sb_k, int sb_c,
                                          int ksteps, f32x4 acc, int lane) {
    const float *ap = a + (lane & 15) * sa_r + (lane >> 4) * sa_k;
    const float *bp = b + (lane >> 4) * sb_k + (lane & 15) * sb_c;
    const int da = 4 * sa_k, db = 4 * sb_k;
    int ks = 0;
    for (; ks + 4 <= ksteps; ks += 4) {
        const float a0 = ap[0], a1 = ap[da], a2 = ap[2 * da], a3 = ap[3 * da];
        const float b0 = bp[0], b1 = bp[db], b2 = bp[2 * db], b3 = bp[3 * db];
        acc = mfma16(a0, b0, acc);
        acc = mfma16(a1, b1, acc);
        acc = mfma16(a2, b2, acc);
        acc = mfma16(a3, b3, acc);
        ap += 4 * da;
        bp += 4 * db;
    }
    for (; ks < ksteps; ++ks) {
        acc = mfma16(*ap, *bp, acc);
        ap += da;
        bp += db;
    }
    return acc;
}

// Two independent products sharing nothing: interleaved so the two MFMA dependency chains
// (40-cycle latency, 32-cycle issue) fill each other's gaps.
__device__ __forceinline__ void tile_mma2(const float *a1, const float *b1, int ks1, f32x4 &acc1, const float *a2,
                                          const float *b2, int ks2, f32x4 &acc2, int sa_r, int sa_k, int sb_k, int sb_c,
                                          int lane) {
    const float *ap1 = a1 + (lane & 15) * sa_r + (lane >> 4) * sa_k, *ap2 = a2 + (lane & 15) * sa_r + (lane >> 4) * sa_k;
    const float *bp1 = b1 + (lane >> 4) * sb_k + (lane & 15) * sb_c, *bp2 = b2 + (lane >> 4) * sb_k + (lane & 15) * sb_c;
    const int da = 4 * sa_k, db = 4 * sb_k;
    const int kmin = ks1 < ks2 ? ks1 : ks2;
    int ks = 0;
    for (; ks + 2 <= kmin; ks += 2) {
        const float x0 = ap1[0], x1 = ap1[da], y0 = bp1[0], y1 = bp1[db];
        const float u0 = ap2[0], u1 = ap2[da], v0 = bp2[0], v1 = bp2[db];
        acc1 = mfma16(x0, y0, acc1);
        acc2 = mfma16(u0, v0, acc2);
        acc1 = mfma16(x1, y1, acc1);
        acc2 = mfma16(u1, v1, acc2);
        ap1 += 2 * da; bp1 += 2 * db; ap2 += 2 * da; bp2 += 2 * db;
    }
    for (int k = ks; k < ks1; ++k) {
        acc1 = mfma16(*ap1, *bp1, acc1);
        ap1 += da; bp1 += db;
    }
    for (int k = ks; k < ks2; ++k) {
        acc2 = mfma16(*ap2, *bp2, acc2);
        ap2 += da; bp2 += db;
    }
}

// Folded pair of products used by the Hartley kernels: with s(i) = S[i * sk],
//   accC += sum_c  (s(c) + s(N - c)) * Bc[c][.]      c = 0 .. 4*ksc-1   (cos part)
//   accS += sum_kk (s(j) - s(N - j)) * Bs[kk][.]     j = Js - kk         (sin part)
// The fold x[n] +- x[N-n] is applied while the operand is read, so no separate LDS pass is needed.
// Positions outside 0..N multiply zero table rows; they are clamped so every read stays in range.
// `ac` / `as_` point at the element (row0, position 0) of the cos / sin source rows.
__device__ __forceinline__ void tile_mma_fold2(const float *ac, const float *as_, int sa_r, int sk, int N, int Js,
                                               const float *bc, int ksc, const float *bs, int kss, f32x4 &accC,
                                               f32x4 &accS, int lane) {
    const int q = lane >> 4;
    const float *pc = ac + (lane & 15) * sa_r, *ps = as_ + (lane & 15) * sa_r;
    const float *bpc = bc + q * 16 + (lane & 15), *bps = bs + q * 16 + (lane & 15);
    const int kmax = ksc > kss ? ksc : kss;
    for (int ks = 0; ks < kmax; ks += 2) {
        float ca[2], cb[2], sa[2], sb[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kk = (ks + u) * 4 + q;
            int c = kk < N ? kk : N;
            int j = Js - kk;
            j = j > 0 ? j : 0;
            const bool oc = ks + u < ksc, os = ks + u < kss;
            ca[u] = oc ? pc[c * sk] + pc[(N - c) * sk] : 0.f;
            cb[u] = oc ? bpc[(ks + u) * 64] : 0.f;
            sa[u] = os ? ps[j * sk] - ps[(N - j) * sk] : 0.f;
            sb[u] = os ? bps[(ks + u) * 64] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (ks + u < ksc) accC = mfma16(ca[u], cb[u], accC);
            if (ks + u < kss) accS = mfma16(sa[u], sb[u], accS);
        }
    }
}

}  // namespace hno
