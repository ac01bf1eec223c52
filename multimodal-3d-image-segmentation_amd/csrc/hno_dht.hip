// Mode-truncated 3-D discrete Hartley transform for gfx950 (MI355X).
//
// Reference semantics: nets/dht.py:16-36 (H = Re F - Im F, 1/N on the forward only),
// nets/hnosegxs.py:378-410 (TransformCrop) and :454-494 (PadInverse).
//
// Formulation (see DESIGN.md "DHT kernels"): the kept block is 2m of N outputs per axis, so
// each axis is a *pruned direct DFT* written as a small real GEMM against cos/sin tables and
// run on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32 FMA chains):
//   * real-input / real-output Hermitian symmetry halves the last-axis work (k2 in 0..m2),
//   * input folding  x[n] +- x[N-n]  halves every reduction length (cos part / sin part),
//   * output pairing (+k, -k share the same cos and sin sums) halves the row count.
// Forward  = [plane kernel: axis W then axis H, one (b*c, n0) plane per workgroup, staged in
//             LDS] -> small intermediate in L2/MALL -> [D kernel: axis D + Re-/+Im + crop].
// Inverse  = the transposed chain: [D kernel] -> intermediate -> [plane kernel: H then W,
//             fused scale + residual add + SELU on store].
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "hno_common.h"
#include "hno_dht_plan.h"

namespace hno {

// (plans, kernel arguments, LDS-DMA helpers: hno_dht_plan.h)
static int pad_2mod4(int v) {
    while ((v & 3) != 2) ++v;
    return v;
}

static void fill_axis(Axis &a, int N, int m, int &cursor) {
    a.N = N;
    a.m = m;
    a.J = N / 2;
    a.Js = (N - 1) / 2;
    a.KT = ceil_div(m + 1, 16);
    a.KP = a.KT * 16;
    a.KcP = round_up(a.J + 1, 4);
    a.KsP = round_up(a.Js, 4);
    a.KmP = round_up(m + 1, 4);
    a.NT = ceil_div(a.J + 1, 16);
    a.cosF = cursor;
    cursor += a.KT * a.KcP * 16;
    a.sinF = cursor;
    cursor += a.KT * a.KsP * 16;
    a.cosI = cursor;
    cursor += a.NT * a.KmP * 16;
    a.sinI = cursor;
    cursor += a.NT * a.KmP * 16;
}

// plane_axis: the table is used by the plane kernels, which fold their operands on the fly as
// s(c) + s(N-c); positions that are their own mirror (c = N/2 forward, frequency 0 inverse) would be
// counted twice, so their cos rows are halved.  The D kernels fold explicitly and use plain tables.
static void build_axis_tables(const Axis &a, std::vector<float> &t, bool plane_axis, bool inv_fold) {
    const double th = 2.0 * M_PI / a.N;
    // forward: B[kk][k]; cos row kk <-> folded position c = kk (0..J);
    //          sin row kk <-> folded position J+1+kk <-> j = Js - kk (1..Js)
    for (int kt = 0; kt < a.KT; ++kt)
        for (int kk = 0; kk < a.KcP; ++kk)
            for (int c = 0; c < 16; ++c) {
                int k = kt * 16 + c;
                double v = (kk <= a.J && k <= a.m) ? cos(th * (double)((long long)k * kk % a.N)) : 0.0;
                if (plane_axis && 2 * kk == a.N) v *= 0.5;
                t[a.cosF + (kt * a.KcP + kk) * 16 + c] = (float)v;
            }
    for (int kt = 0; kt < a.KT; ++kt)
        for (int kk = 0; kk < a.KsP; ++kk)
            for (int c = 0; c < 16; ++c) {
                int k = kt * 16 + c;
                int j = a.Js - kk;
                double v = (kk < a.Js && k <= a.m) ? sin(th * (double)((long long)k * j % a.N)) : 0.0;
                t[a.sinF + (kt * a.KsP + kk) * 16 + c] = (float)v;
            }
    // inverse: B[kk][n], kk = frequency 0..m, n = output position 0..J
    for (int nt = 0; nt < a.NT; ++nt)
        for (int kk = 0; kk < a.KmP; ++kk)
            for (int c = 0; c < 16; ++c) {
                int n = nt * 16 + c;
                bool ok = (kk <= a.m && n <= a.J);
                double ang = th * (double)((long long)kk * n % a.N);
                t[a.cosI + (nt * a.KmP + kk) * 16 + c] = ok ? (float)(cos(ang) * ((inv_fold && kk == 0) ? 0.5 : 1.0)) : 0.f;
                t[a.sinI + (nt * a.KmP + kk) * 16 + c] = ok ? (float)sin(ang) : 0.f;
            }
}

typedef std::tuple<int, int, int, int, int, int, int> PlanKey;
static std::map<PlanKey, DhtPlan> g_plans;
static std::mutex g_plan_mutex;

static int get_plan(int N0, int N1, int N2, int m0, int m1, int m2, const DhtPlan **out) {
    int dev = 0;
    HNO_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    PlanKey key(dev, N0, N1, N2, m0, m1, m2);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) {
        *out = &it->second;
        return HNO_OK;
    }
    DhtPlan p;
    int cursor = 0;
    fill_axis(p.ax[0], N0, m0, cursor);
    fill_axis(p.ax[1], N1, m1, cursor);
    fill_axis(p.ax[2], N2, m2, cursor);
    // dht_fwd_plane_dma_kernel keeps the axis-W result in its MFMA accumulators and feeds them to the axis-H product as they
    // lie: tile X (0..NP1-1) holds plane row n1 = 1 + 16 X + i in accumulator row i = 4 q + r (lane group q, register r), its
    // partner tile the mirror row N1 - n1.  K-step (X, r) of the axis-H product therefore sums over n1(X, 4 q + r), q = 0..3:
    // table [X][r][cos | sin][q][k1]
    p.NP1 = ceil_div(p.ax[1].Js, 16);
    p.hperm = cursor;
    if ((N1 & 1) && p.ax[1].KT == 1) cursor += p.NP1 * 4 * 2 * 4 * 16;
    const bool dma_tab = (N1 & 1) && p.ax[1].KT == 1 && p.ax[2].KT == 1;
    p.dmatab = cursor;
    p.dmatab_stride = round_up((p.ax[2].KcP / 4 + p.ax[2].KsP / 4 + p.NP1 * 8) * 64, 64);
    if (dma_tab) cursor += kDmaTabCopies * p.dmatab_stride;
    p.itab = cursor;
    const int i_nt2 = ceil_div(p.ax[2].J, 16);
    p.itab_stride = (p.NP1 * 2 * (p.ax[1].KmP / 4) + i_nt2 * 8) * 64;
    if (dma_tab) cursor += kDmaTabCopies * p.itab_stride;
    p.table_floats = cursor;
    std::vector<float> host(cursor, 0.f);
    build_axis_tables(p.ax[0], host, false, false);
    build_axis_tables(p.ax[1], host, false, false);
    build_axis_tables(p.ax[2], host, false, false);
    if ((N1 & 1) && p.ax[1].KT == 1) {
        const double th1 = 2.0 * M_PI / N1;
        for (int X = 0; X < p.NP1; ++X)
            for (int r = 0; r < 4; ++r)
                for (int q = 0; q < 4; ++q)
                    for (int k1 = 0; k1 < 16; ++k1) {
                        const int n1 = 1 + 16 * X + 4 * q + r;
                        const bool ok = n1 <= p.ax[1].Js && k1 <= p.ax[1].m;
                        const double ang = th1 * (double)((long long)k1 * n1 % N1);
                        float *t = host.data() + p.hperm + ((X * 4 + r) * 2) * 64 + q * 16 + k1;
                        t[0] = ok ? (float)cos(ang) : 0.f;
                        t[64] = ok ? (float)sin(ang) : 0.f;
                    }
    }
    if (dma_tab) {
        // row ks of the axis-W tables in lane order: lane (q, l15) <-> table element (4 ks + q, l15) = cosF[ks * 64 + lane]
        const Axis &w = p.ax[2];
        std::vector<float> blk;
        blk.insert(blk.end(), host.begin() + w.cosF, host.begin() + w.cosF + (w.KcP / 4) * 64);
        // the plane kernels fold x[c] + x[N2 - c]: the column N2 / 2 of an even N2 is its own mirror and would count twice
        if (!(N2 & 1))
            for (int c = 0; c < 16; ++c) blk[(N2 / 2) * 16 + c] *= 0.5f;
        blk.insert(blk.end(), host.begin() + w.sinF, host.begin() + w.sinF + (w.KsP / 4) * 64);
        blk.insert(blk.end(), host.begin() + p.hperm, host.begin() + p.hperm + p.NP1 * 8 * 64);
        for (int c = 0; c < kDmaTabCopies; ++c) std::copy(blk.begin(), blk.end(), host.begin() + p.dmatab + c * p.dmatab_stride);
        // inverse item kernel (dht_inv_item_kernel).  Axis H: output column n1 = 1 + 16 X + l15 against frequency kk = 4 ks + q (kk = 0
        // is its own mirror in the +-k1 fold: cosine halved).  Axis W: the A operand is the axis-H accumulator as it lies, so k-step r
        // of lane group q is frequency k2 = 4 q + r; output column n2 = 1 + 16 nt2 + l15.
        const Axis &h = p.ax[1];
        const double thH = 2.0 * M_PI / h.N, thW = 2.0 * M_PI / w.N;
        std::vector<float> ib;
        for (int X = 0; X < p.NP1; ++X)
            for (int cs = 0; cs < 2; ++cs)
                for (int ks = 0; ks < h.KmP / 4; ++ks)
                    for (int ln = 0; ln < 64; ++ln) {
                        const int n = 1 + 16 * X + (ln & 15), kk = 4 * ks + (ln >> 4);
                        const bool ok = n <= h.J && kk <= h.m;
                        const double ang = thH * (double)((long long)kk * n % h.N);
                        ib.push_back(!ok ? 0.f : (float)(cs ? sin(ang) : cos(ang) * (kk == 0 ? 0.5 : 1.0)));
                    }
        for (int nt = 0; nt < i_nt2; ++nt)
            for (int cs = 0; cs < 2; ++cs)
                for (int r = 0; r < 4; ++r)
                    for (int ln = 0; ln < 64; ++ln) {
                        const int n = 1 + 16 * nt + (ln & 15), kk = 4 * (ln >> 4) + r;
                        const bool ok = n <= w.J && kk <= w.m;
                        const double ang = thW * (double)((long long)kk * n % w.N);
                        ib.push_back(!ok ? 0.f : (float)(cs ? sin(ang) : cos(ang)));
                    }
        for (int c = 0; c < kDmaTabCopies; ++c) std::copy(ib.begin(), ib.end(), host.begin() + p.itab + c * p.itab_stride);
    }
    // table creation is the one place that allocates: do it outside graph capture (warm-up)
    HNO_CHECK_HIP(hipMalloc((void **)&p.tables, sizeof(float) * cursor));
    HNO_CHECK_HIP(hipMemcpy(p.tables, host.data(), sizeof(float) * cursor, hipMemcpyHostToDevice));
    const Axis &a1 = p.ax[1], &a2 = p.ax[2];
    p.K1S = 2 * m1 + 1;
    p.CP = p.K1S * a2.KP;
    p.MP1 = round_up(N1, 16);
    // forward plane LDS
    int need = a2.KcP > (a2.J + 1 + a2.KsP) ? a2.KcP : (a2.J + 1 + a2.KsP);
    if (need < N2 + 4) need = N2 + 4;   // columns N2.. are zeros read by the folded operand fetch
    p.lda2 = pad_2mod4(need);
    p.TP = a1.KcP > (a1.J + 1 + a1.KsP) ? a1.KcP : (a1.J + 1 + a1.KsP);
    if (p.TP < N1 + 4) p.TP = N1 + 4;
    p.ldt = 2 * a2.KP + 16;
    int c = 0;
    p.f_tabW = c;
    c += a2.KT * (a2.KcP + a2.KsP) * 16;
    p.f_tabH = c;
    c += a1.KT * (a1.KcP + a1.KsP) * 16;
    p.f_xs = c;
    c += p.MP1 * p.lda2;
    p.f_T = c;
    c += p.TP * p.ldt;
    p.f_lds_floats = c;
    // inverse plane LDS
    p.ldE = pad_2mod4(a1.KmP);
    p.ldF = pad_2mod4(a2.KmP);
    p.ldo = N2;
    while ((p.ldo & 7) != 4) ++p.ldo;   // 4 * ldo == 16 (mod 32): C-layout stores spread over banks
    c = 0;
    p.i_tabH = c;
    c += 2 * a1.NT * a1.KmP * 16;
    p.i_tabW = c;
    c += 2 * a2.NT * a2.KmP * 16;
    p.i_Es = c;
    c += 2 * a2.KP * p.ldE;
    p.i_Ed = c;
    c += 2 * a2.KP * p.ldE;
    p.i_FR = c;
    c += p.MP1 * p.ldF;
    p.i_FI = c;
    c += p.MP1 * p.ldF;
    p.i_O = c;
    c += N1 * p.ldo;
    p.i_lds_floats = c;
    auto res = g_plans.emplace(key, p);
    *out = &res.first->second;
    return HNO_OK;
}

// ------------------------------------------------------------------------------ kernels
// ---- forward, axes W and H, one (bc, n0) plane per workgroup iteration -------------------
// MAXE > 0: the whole plane (<= 256 * MAXE elements) is fetched into registers with fully
// coalesced, independent loads, and the NEXT plane's fetch is issued before the current plane's
// MFMA phases so HBM latency hides behind compute.  MAXE == 0: generic row-by-row path.
// The folds x[n] +- x[N-n] (axis W) and T[n1] +- T[N1-n1] (axis H) are cheap in-place LDS passes
// (a wave per row, no integer division), so the MFMA loops read plain pre-folded operands.
template <int MAXE, int NTH = 256>
__global__ __launch_bounds__(NTH) void dht_fwd_plane_kernel(const float *__restrict__ x, const float *__restrict__ xact,
                                                            float *__restrict__ Y, DhtArgs a) {
    extern __shared__ float lds[];
    const DhtPlan &p = a.p;
    const Axis &a1 = p.ax[1], &a2 = p.ax[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N1 = a1.N, N2 = a2.N;
    float *tabW = lds + p.f_tabW, *tabH = lds + p.f_tabH, *xs = lds + p.f_xs, *T = lds + p.f_T;
    const float *cosW = tabW, *sinW = tabW + a2.KT * a2.KcP * 16;
    const float *cosH = tabH, *sinH = tabH + a1.KT * a1.KcP * 16;
    // tables: contiguous in the plan buffer as cosF | sinF per axis
    for (int i = tid; i < a2.KT * (a2.KcP + a2.KsP) * 16; i += NTH) tabW[i] = p.tables[a2.cosF + i];
    for (int i = tid; i < a1.KT * (a1.KcP + a1.KsP) * 16; i += NTH) tabH[i] = p.tables[a1.cosF + i];
    // zero the padding of xs once (rows >= N1, columns >= N2): never overwritten below
    for (int i = tid; i < (p.MP1 - N1) * p.lda2; i += NTH) xs[N1 * p.lda2 + i] = 0.f;
    const int padc = p.lda2 - N2;
    for (int i = tid; i < N1 * padc; i += NTH) xs[(i / padc) * p.lda2 + N2 + (i % padc)] = 0.f;
    // T rows >= N1 are reduction padding for axis H: keep them zero (stage W never stores there)
    for (int i = tid; i < (p.TP - N1) * p.ldt; i += NTH) T[N1 * p.ldt + i] = 0.f;

    const int planes = a.BC * p.ax[0].N;
    const size_t plane_elems = (size_t)N1 * N2;
    const int MT1 = p.MP1 / 16;
    constexpr int NE = MAXE > 0 ? MAXE : 1;
    float rx[NE], ru[NE];
    // (row, col) of element tid + NTH * j, advanced incrementally
    const int r0 = tid / N2, c0 = tid - r0 * N2, dr = NTH / N2, dc = NTH - dr * N2;
    auto fetch = [&](int plane) {
        const float *xp = x + plane_base(a, plane, plane_elems);
        const float *up = xact ? xact + plane_base(a, plane, plane_elems) : nullptr;
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const unsigned e = tid + (unsigned)NTH * j;
            const bool in = e < plane_elems;
            rx[j] = in ? xp[e] : 0.f;
            ru[j] = (in && up) ? up[e] : 0.f;
        }
    };
    if (MAXE > 0 && blockIdx.x < planes) fetch(blockIdx.x);
    for (int plane = blockIdx.x; plane < planes; plane += gridDim.x) {
        __syncthreads();  // previous iteration finished reading xs / T
        if (MAXE > 0) {
            // ---- registers -> LDS rows, activation gradient applied here
            int r = r0, c = c0;
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                if (tid + (unsigned)NTH * j < plane_elems) xs[r * p.lda2 + c] = xact ? rx[j] * act_grad_from_out(ru[j], a.act) : rx[j];
                r += dr;
                c += dc;
                if (c >= N2) {
                    c -= N2;
                    ++r;
                }
            }
            const int next = plane + gridDim.x;
            if (next < planes) fetch(next);  // in flight during this plane's compute
        } else {
            const float *xp = x + plane_base(a, plane, plane_elems);
            const float *up = xact ? xact + plane_base(a, plane, plane_elems) : nullptr;
            for (int r = wave; r < N1; r += (NTH / 64))
                for (int c = lane; c < N2; c += 64) {
                    float v = xp[(size_t)r * N2 + c];
                    if (up) v *= act_grad_from_out(up[(size_t)r * N2 + c], a.act);
                    xs[r * p.lda2 + c] = v;
                }
        }
        __syncthreads();
        // ---- fold along W in place (division-free: a wave per row, lanes over columns)
        for (int r = wave; r < N1; r += (NTH / 64)) {
            float *row = xs + r * p.lda2;
            for (int c = 1 + lane; c <= a2.Js; c += 64) {
                const float va = row[c], vb = row[N2 - c];
                row[c] = va + vb;
                row[N2 - c] = va - vb;
            }
        }
        __syncthreads();
        // ---- axis W: T[n1][k2]: cos part -> columns [0,KP2), sin part -> [KP2, 2KP2)
        const int ntaskW = (a.dbg & 1) ? 0 : MT1 * a2.KT;
        for (int t = wave; t < ntaskW; t += (NTH / 64)) {
            const int kt = t % a2.KT, mt = t / a2.KT;
            f32x4 accC = {0.f, 0.f, 0.f, 0.f}, accS = accC;
            const float *rows = xs + mt * 16 * p.lda2;
            tile_mma2(rows, cosW + kt * a2.KcP * 16, a2.KcP / 4, accC, rows + a2.J + 1, sinW + kt * a2.KsP * 16, a2.KsP / 4,
                      accS, p.lda2, 1, 16, 1, lane);
            const int col = kt * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = mt * 16 + (lane >> 4) * 4 + r;
                if (row < N1) {
                    T[row * p.ldt + col] = accC[r];
                    T[row * p.ldt + a2.KP + col] = accS[r];
                }
            }
        }
        __syncthreads();
        // ---- fold T along n1 in place (a wave per row pair, lanes over the 2*KP2 columns)
        for (int c = 1 + wave; c <= a1.Js; c += (NTH / 64)) {
            float *ra_ = T + c * p.ldt, *rb_ = T + (N1 - c) * p.ldt;
            for (int col = lane; col < 2 * a2.KP; col += 64) {
                const float va = ra_[col], vb = rb_[col];
                ra_[col] = va + vb;
                rb_[col] = va - vb;
            }
        }
        __syncthreads();
        // ---- axis H: rows = T columns (Ac | As), reduce over n1, outputs k1 in 0..m1 (+/-)
        float *Yp = Y + (size_t)plane * (2 * p.CP);
        const int ntaskH = (a.dbg & 2) ? 0 : a2.KT * a1.KT * 2;
        for (int t = wave; t < ntaskH; t += (NTH / 64)) {
            const int part = t & 1, kt1 = (t >> 1) % a1.KT, kt2 = (t >> 1) / a1.KT;
            const float *Tc = T + kt2 * 16;             // Ac columns of this k2 tile
            const float *Ts = T + a2.KP + kt2 * 16;     // As columns
            const float *bc = cosH + kt1 * a1.KcP * 16, *bs = sinH + kt1 * a1.KsP * 16;
            f32x4 accP = {0.f, 0.f, 0.f, 0.f}, accQ = {0.f, 0.f, 0.f, 0.f};
            if (part == 0)    // BR(+-k1) = P_Ac -+ Q_As
                tile_mma2(Tc, bc, a1.KcP / 4, accP, Ts + (a1.J + 1) * p.ldt, bs, a1.KsP / 4, accQ, 1, p.ldt, 16, 1, lane);
            else              // BI(+k1) = -(Q_Ac + P_As), BI(-k1) = Q_Ac - P_As
                tile_mma2(Ts, bc, a1.KcP / 4, accP, Tc + (a1.J + 1) * p.ldt, bs, a1.KsP / 4, accQ, 1, p.ldt, 16, 1, lane);
            const int k1 = kt1 * 16 + (lane & 15);
            if (k1 <= a1.m && !((a.dbg & 4) && accP[0] != 12345.f)) {
                f32x4 vp, vm;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (part == 0) {
                        vp[r] = accP[r] - accQ[r];
                        vm[r] = accP[r] + accQ[r];
                    } else {
                        vp[r] = -(accQ[r] + accP[r]);
                        vm[r] = accQ[r] - accP[r];
                    }
                }
                const int k2 = kt2 * 16 + (lane >> 4) * 4;
                float *dstp = Y + ymid(a, plane, part, a1.m + k1, k2);
                *reinterpret_cast<f32x4 *>(dstp) = vp;
                if (k1 >= 1) {
                    float *dstm = Y + ymid(a, plane, part, a1.m - k1, k2);
                    *reinterpret_cast<f32x4 *>(dstm) = vm;
                }
            }
        }
    }
}

// ---- forward plane kernel, specialised ------------------------------------------------------
// Compile-time reduction lengths (KC/KS k-steps of 4 for the cos/sin parts of axis W = 2 and
// axis H = 1), one k tile per axis (m + 1 <= 16).  Differences to the generic kernel:
//   * the cos/sin tables live in VGPRs (34 registers for N = 65), not in LDS;
//   * the folds x[c] +- x[N-c] happen while the MFMA operands are read, through two pointers
//     (forward / mirrored) and compile-time LDS offsets: no fold passes, two barriers fewer;
//   * every task issues all of its operand reads up front, then runs its two MFMA chains.
template <int KC2, int KS2, int KC1, int KS1, bool HAS_ACT>
__global__ __launch_bounds__(256, 3) void dht_fwd_plane_spec_kernel(const float *__restrict__ x, const float *__restrict__ xact,
                                                                 float *__restrict__ Y, DhtArgs a) {
    extern __shared__ float lds[];
    constexpr int NE = 20;
    const DhtPlan &p = a.p;
    const Axis &a1 = p.ax[1], &a2 = p.ax[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l15 = lane & 15;
    const int N1 = a1.N, N2 = a2.N;
    const int lda = p.lda2, ldt = p.ldt;
    // Work roles instead of wave numbers: wave w of every workgroup sits on SIMD w, and the workgroups that
    // share a CU differ by multiples of 256 in blockIdx, so rotating the roles by blockIdx / 256 spreads the
    // heavier roles (0 and 1 also run the axis-H chains) over the four matrix pipes of the CU.
    const int role = (a.dbg & 128) ? wave : ((wave + (int)(blockIdx.x >> 8)) & 3);
    // one or two leftover rows (N1 = 16 t + 1: 65, 33, 97) are not worth a 16-row MFMA tile: VALU dot products
    const int MTF = (N1 % 16 >= 1 && N1 % 16 <= 2 && !(a.dbg & 256)) ? N1 / 16 : p.MP1 / 16;
    float *xs = lds + 16;                       // 16 zero floats in front: sin-part reads at j = -1..-3
    float *T = xs + p.MP1 * lda;
    // B operands (tables) in registers: lane holds B[k = ks*4 + q][col = l15]
    // axis-W tables in registers (every wave uses them once per plane); the axis-H tables, used by two waves
    // only, stay in LDS and are read together with the A operands (keeps the kernel at 3 workgroups per CU
    // without spilling: a spill reload waits on vmcnt(0), i.e. on the prefetched plane)
    float bwc[KC2], bws[KS2];
#pragma unroll
    for (int ks = 0; ks < KC2; ++ks) bwc[ks] = p.tables[a2.cosF + (ks * 4 + q) * 16 + l15];
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) bws[ks] = p.tables[a2.sinF + (ks * 4 + q) * 16 + l15];
#pragma unroll
    for (int ks = 0; ks < KC2; ++ks) asm volatile("" : "+v"(bwc[ks]));   // keep in registers (no re-load inside the loop)
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) asm volatile("" : "+v"(bws[ks]));
    float *tabH = T + p.TP * ldt;   // [KC1 + KS1][64]
    for (int i = tid; i < (KC1 + KS1) * 64; i += 256) tabH[i] = i < KC1 * 64 ? p.tables[a1.cosF + i] : p.tables[a1.sinF + (i - KC1 * 64)];
    const float *thc = tabH + lane, *ths = tabH + KC1 * 64 + lane;
    // zero: guard, xs padding (rows >= N1, columns >= N2), T rows >= N1
    if (tid < 16) lds[tid] = 0.f;
    for (int i = tid; i < (p.MP1 - N1) * lda; i += 256) xs[N1 * lda + i] = 0.f;
    const int padc = lda - N2;
    for (int i = tid; i < N1 * padc; i += 256) xs[(i / padc) * lda + N2 + (i % padc)] = 0.f;
    for (int i = tid; i < (p.TP - N1) * ldt; i += 256) T[N1 * ldt + i] = 0.f;

    const int planes = a.BC * p.ax[0].N;
    const size_t plane_elems = (size_t)N1 * N2;
    float rx[NE], ru[HAS_ACT ? NE : 1];
    const int r0 = tid / N2, c0 = tid - r0 * N2, dr = 256 / N2, dc = 256 - dr * N2;
    auto fetch = [&](int plane) {
        const float *xp = x + plane_base(a, plane, plane_elems);
        const float *up = xact ? xact + plane_base(a, plane, plane_elems) : nullptr;
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const unsigned e = tid + 256u * j;
            const bool in = e < plane_elems;
            rx[j] = in ? xp[e] : 0.f;
            if (HAS_ACT) ru[j] = in ? up[e] : 0.f;
        }
    };
    HNO_STAMP(a.stamps, 0);
    if (a.stamps && blockIdx.x == 0 && tid == 0) { a.stamps[60] = wall_clock64(); a.stamps[62] = clock64(); }
    if (a.stamps && blockIdx.x == gridDim.x - 1 && tid == 0) a.stamps[58] = wall_clock64();
    if (a.stamps && blockIdx.x == 600 && tid == 0) a.stamps[56] = wall_clock64();
    if (blockIdx.x < planes) fetch(blockIdx.x);
    int it = 0;
    for (int plane = blockIdx.x; plane < planes; plane += gridDim.x, ++it) {
        HNO_STAMP(a.stamps, 1 + it * 6);
        __syncthreads();  // previous iteration finished reading xs / T
        HNO_STAMP(a.stamps, 2 + it * 6);
        {
            int r = r0, c = c0;
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                if (tid + 256u * j < plane_elems) xs[r * lda + c] = HAS_ACT ? rx[j] * act_grad_from_out(ru[HAS_ACT ? j : 0], a.act) : rx[j];
                r += dr;
                c += dc;
                if (c >= N2) {
                    c -= N2;
                    ++r;
                }
            }
        }
        if (plane + gridDim.x < planes) fetch(plane + gridDim.x);  // in flight during this plane's compute
        HNO_STAMP(a.stamps, 3 + it * 6);
        __syncthreads();
        HNO_STAMP(a.stamps, 4 + it * 6);
        // ---- axis W.  cos: (x[c] + x[N-c]) c = 4ks+q ; sin: (x[j] - x[N-j]) j = Js - (4ks+q)
        if (!(a.dbg & 1)) {
            for (int mt = role; mt < MTF; mt += 4) {
                const float *row = xs + (mt * 16 + l15) * lda;
                const float *pf = row + q, *pb = row + N2 - q;
                const float *sf = row + a2.Js - q, *sb = row + N2 - a2.Js + q;
                // one chain at a time: all raw LDS reads of the chain are issued, then folded, then the MFMAs run
                // (left alone, the scheduler sinks each read next to its use and every MFMA waits for an LDS
                // round trip; both chains at once need more registers than 3 workgroups per CU leave)
                f32x4 accC = {0.f, 0.f, 0.f, 0.f}, accS = accC;
                {
                    float ac[KC2], cb[KC2];
#pragma unroll
                    for (int ks = 0; ks < KC2; ++ks) {
                        ac[ks] = pf[4 * ks];
                        cb[ks] = pb[-4 * ks];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = 0; ks < KC2; ++ks) ac[ks] += cb[ks];
#pragma unroll
                    for (int ks = 0; ks < KC2; ++ks) accC = mfma16(ac[ks], bwc[ks], accC);
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    float as_[KS2], sb_[KS2];
#pragma unroll
                    for (int ks = 0; ks < KS2; ++ks) {
                        as_[ks] = sf[-4 * ks];
                        sb_[ks] = sb[4 * ks];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = 0; ks < KS2; ++ks) as_[ks] -= sb_[ks];
#pragma unroll
                    for (int ks = 0; ks < KS2; ++ks) accS = mfma16(as_[ks], bws[ks], accS);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = mt * 16 + q * 4 + r;
                    if (rr < N1) {
                        T[rr * ldt + l15] = accC[r];
                        T[rr * ldt + 16 + l15] = accS[r];
                    }
                }
            }
            if (role == 3) {
                for (int rr = MTF * 16; rr < N1; ++rr) {   // same operands as the MFMA path, k split over the 4 lane groups
                    const float *row = xs + rr * lda;
                    const float *pf = row + q, *pb = row + N2 - q;
                    const float *sf = row + a2.Js - q, *sb = row + N2 - a2.Js + q;
                    float pc = 0.f, ps = 0.f;
#pragma unroll
                    for (int ks = 0; ks < KC2; ++ks) pc += (pf[4 * ks] + pb[-4 * ks]) * bwc[ks];
#pragma unroll
                    for (int ks = 0; ks < KS2; ++ks) ps += (sf[-4 * ks] - sb[4 * ks]) * bws[ks];
                    pc += __shfl_xor(pc, 16);
                    ps += __shfl_xor(ps, 16);
                    pc += __shfl_xor(pc, 32);
                    ps += __shfl_xor(ps, 32);
                    if (q == 0) {
                        T[rr * ldt + l15] = pc;
                        T[rr * ldt + 16 + l15] = ps;
                    }
                }
            }
        }
        HNO_STAMP(a.stamps, 5 + it * 6);
        __syncthreads();
        HNO_STAMP(a.stamps, 6 + it * 6);
        // ---- axis H (fold along n1 while reading).  wave 0: BR = P_Ac -+ Q_As ; wave 1: BI from Q_Ac, P_As
        float *Yp = Y + (size_t)plane * (2 * p.CP);
        if (role < 2 && !(a.dbg & 2)) {
            const int part = role;
            const float *cs = T + (part == 0 ? 0 : 16) + l15;   // source columns of the cos sum
            const float *ss = T + (part == 0 ? 16 : 0) + l15;   // source columns of the sin sum
            const float *pf = cs + q * ldt, *pb = cs + (N1 - q) * ldt;
            const float *sf = ss + (a1.Js - q) * ldt, *sb = ss + (N1 - a1.Js + q) * ldt;
            f32x4 accP = {0.f, 0.f, 0.f, 0.f}, accQ = accP;
            {
                float ac[KC1], cb[KC1], bhc[KC1];
#pragma unroll
                for (int ks = 0; ks < KC1; ++ks) {
                    ac[ks] = pf[4 * ks * ldt];
                    cb[ks] = pb[-4 * ks * ldt];
                    bhc[ks] = thc[64 * ks];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KC1; ++ks) ac[ks] += cb[ks];
#pragma unroll
                for (int ks = 0; ks < KC1; ++ks) accP = mfma16(ac[ks], bhc[ks], accP);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                float as_[KS1], sb_[KS1], bhs[KS1];
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) {
                    as_[ks] = sf[-4 * ks * ldt];
                    sb_[ks] = sb[4 * ks * ldt];
                    bhs[ks] = ths[64 * ks];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) as_[ks] -= sb_[ks];
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) accQ = mfma16(as_[ks], bhs[ks], accQ);
            }
            const int k1 = l15;
            if (k1 <= a1.m && !((a.dbg & 4) && accP[0] != 12345.f)) {
                f32x4 vp, vm;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (part == 0) {
                        vp[r] = accP[r] - accQ[r];
                        vm[r] = accP[r] + accQ[r];
                    } else {
                        vp[r] = -(accQ[r] + accP[r]);
                        vm[r] = accQ[r] - accP[r];
                    }
                }
                const int k2 = q * 4;
                *reinterpret_cast<f32x4 *>(Y + ymid(a, plane, part, a1.m + k1, k2)) = vp;
                if (k1 >= 1) *reinterpret_cast<f32x4 *>(Y + ymid(a, plane, part, a1.m - k1, k2)) = vm;
            }
        }
    }
    if (a.stamps && blockIdx.x == 0 && tid == 0) { a.stamps[61] = wall_clock64(); a.stamps[63] = clock64(); }
    if (a.stamps && blockIdx.x == gridDim.x - 1 && tid == 0) a.stamps[59] = wall_clock64();
    if (a.stamps && blockIdx.x == 600 && tid == 0) a.stamps[57] = wall_clock64();
}

// Recombination + crop store of one (kt0, k1s, kt2) tile of the forward D transform.  The lane holds, for
// k0 = kt0*16 + (lane&15) and k2 = kt2*16 + q*4 + r:
//   X(+k0) = (PR + QI) + i (PI - QR),  X(-k0) = (PR - QI) + i (PI + QR)
__device__ __forceinline__ void fwd_d_store(const DhtArgs &a, float *__restrict__ out, int bc, int k1s, int kt2, int kt0, int lane,
                                            const f32x4 &PR, const f32x4 &PI, const f32x4 &QR, const f32x4 &QI) {
    const DhtPlan &p = a.p;
    const int m0 = p.ax[0].m, m1 = p.ax[1].m, m2 = p.ax[2].m;
    const int q = lane >> 4;
    const int s0 = 2 * m0 + a.full0, s1 = 2 * m1 + a.full1, s2 = 2 * m2 + a.full2;
    float *ob = out + (size_t)bc * s0 * s1 * s2;
    (void)ob;
    const int k0 = kt0 * 16 + (lane & 15);
    const int k1 = k1s - m1;
    if (k0 <= m0 && a.mode != 0) {
        // Fourier half spectrum: S[b][re|im][c][o0][o1][k2], k2 in [0, m2)
        const int b = bc / a.C, c = bc - b * a.C;
        const size_t msz = (size_t)s0 * (2 * m1) * m2;
        float *sr = out + ((size_t)(b * 2 + 0) * a.C + c) * msz, *si = out + ((size_t)(b * 2 + 1) * a.C + c) * msz;
        const int o1 = kept_pos(k1, m1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k2 = kt2 * 16 + q * 4 + r;
            if (k2 >= m2 || o1 < 0) continue;
            const float w = a.scale * ((a.mode == 2 && k2 > 0) ? 2.f : 1.f);
#pragma unroll
            for (int sgn = 0; sgn < 2; ++sgn) {
                if (sgn == 1 && k0 == 0) continue;
                const int o0 = kept_pos(sgn ? -k0 : k0, m0, a.full0);
                if (o0 < 0) continue;
                const size_t idx = ((size_t)o0 * (2 * m1) + o1) * m2 + k2;
                sr[idx] = w * (sgn ? PR[r] - QI[r] : PR[r] + QI[r]);
                si[idx] = w * (sgn ? PI[r] + QR[r] : PI[r] - QR[r]);
            }
        }
    } else if (k0 <= m0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k2 = kt2 * 16 + q * 4 + r;
            if (k2 > m2) continue;
#pragma unroll
            for (int sgn = 0; sgn < 2; ++sgn) {
                if (sgn == 1 && k0 == 0) continue;
                const int kk0 = sgn ? -k0 : k0;
                const float xr = sgn ? PR[r] - QI[r] : PR[r] + QI[r];
                const float xi = sgn ? PI[r] + QR[r] : PI[r] - QR[r];
                // H[k] = Re X[k] - Im X[k]
                int o0 = kept_pos(kk0, m0, a.full0), o1 = kept_pos(k1, m1, a.full1);
                if (k2 < m2 + a.full2 && o0 >= 0 && o1 >= 0)
                    ob[((size_t)o0 * s1 + o1) * s2 + k2] = a.scale * (xr - xi);
                // H[-k] = Re X[k] + Im X[k]
                o0 = kept_pos(-kk0, m0, a.full0);
                o1 = kept_pos(-k1, m1, a.full1);
                if (k2 >= 1 && o0 >= 0 && o1 >= 0)
                    ob[((size_t)o0 * s1 + o1) * s2 + (s2 - k2)] = a.scale * (xr + xi);
            }
        }
    }
}

// ---- forward plane kernel, one WAVE per plane ------------------------------------------------
// The workgroup-per-plane kernels above spend most of their time in barriers and in the dependency chain of
// one plane (load -> LDS -> GEMM -> barrier -> GEMM -> barrier, ~7 000 cycles, 3 planes in flight per CU).
// Here every wave owns a plane: no workgroup barrier at all, 8 planes in flight per CU (two waves per SIMD keep
// each matrix pipe fed while the other wave waits on LDS or HBM).
//   * the plane is kept FLAT in the wave's LDS slice (row stride N2, odd), so staging it is a ds_write_b128 per
//     four elements at an immediate offset; a 16-row MFMA tile takes every second row of a 32-row group
//     (rows 32 g + 2 i + parity), which makes the operand reads bank-conflict free for any odd stride;
//   * the axis-W result T[n1][0..31] overwrites the first 32 columns of row n1 in place (a tile's reads precede
//     its writes, LDS operations of one wave execute in order), so the slice is 17 KB and 8 waves fit a CU;
//   * the next plane is prefetched into registers with 16-byte loads during the GEMMs (2 waves per SIMD leave
//     256 VGPRs each); two tiles (axis W) / both parts (axis H) run as four interleaved MFMA chains, which hides
//     the 40-cycle dependent-accumulator latency that a single wave would otherwise expose.
// Requires N1 % 32 == 1 (one leftover row, done on the VALU), odd N2, one k tile per axis, Js == KsP.
struct __attribute__((packed, aligned(4))) f4u_t { float x, y, z, w; };

template <int KC2, int KS2, int KC1, int KS1, int NV, int NWV>   // NV = N1 * N2 / 256 full 16-byte-per-lane iterations; NWV waves per workgroup
__global__ __launch_bounds__(64 * NWV, 1) void dht_fwd_plane_wave_kernel(const float *__restrict__ x, float *__restrict__ Y, DhtArgs a) {
    extern __shared__ float lds[];
    typedef float f4v __attribute__((ext_vector_type(4)));
    const DhtPlan &p = a.p;
    const Axis &a1 = p.ax[1], &a2 = p.ax[2];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, l15 = lane & 15;
    const int N1 = a1.N, N2 = a2.N;
    const int pe = N1 * N2;
    const int LW = 16 + ((pe + 32 + 15) & ~15);   // guard | plane | 32 zeros (row N1 of T, mirror of column 0)
    float *xs = lds + (size_t)wave * LW + 16;
    // one 8-wave workgroup per CU; workgroup b owns the contiguous planes [b P / G, (b + 1) P / G), so every CU
    // gets the same number of planes (+-1) and its waves share them round-robin
    const int planes_all = a.BC * p.ax[0].N;
    const int p_begin = (int)((long long)planes_all * blockIdx.x / gridDim.x);
    const int planes = (int)((long long)planes_all * (blockIdx.x + 1) / gridDim.x);
    constexpr int nwaves = NWV;
    constexpr int NT = 4;   // scalar tail iterations: pe - 256 NV < 256 elements
    f4v rv[NV];
    float rt[NT];
    auto fetch = [&](int plane) {
        const float *xp = x + plane_base(a, plane, (size_t)pe);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const f4u_t v = *reinterpret_cast<const f4u_t *>(xp + 4 * lane + 256 * j);
            rv[j] = f4v{v.x, v.y, v.z, v.w};
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int e = 256 * NV + lane + 64 * j;
            rt[j] = e < pe ? xp[e] : 0.f;
        }
    };
    int plane = p_begin + wave;
    if (plane < planes) fetch(plane);   // in flight while the tables load
    float bwc[KC2], bws[KS2], bhc[KC1], bhs[KS1];
#pragma unroll
    for (int ks = 0; ks < KC2; ++ks) bwc[ks] = p.tables[a2.cosF + (ks * 4 + q) * 16 + l15];
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) bws[ks] = p.tables[a2.sinF + (ks * 4 + q) * 16 + l15];
#pragma unroll
    for (int ks = 0; ks < KC1; ++ks) bhc[ks] = p.tables[a1.cosF + (ks * 4 + q) * 16 + l15];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) bhs[ks] = p.tables[a1.sinF + (ks * 4 + q) * 16 + l15];
#pragma unroll
    for (int ks = 0; ks < KC2; ++ks) asm volatile("" : "+v"(bwc[ks]));   // stay in registers (no re-load in the loop)
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) asm volatile("" : "+v"(bws[ks]));
#pragma unroll
    for (int ks = 0; ks < KC1; ++ks) asm volatile("" : "+v"(bhc[ks]));
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) asm volatile("" : "+v"(bhs[ks]));
    if (lane < 16) xs[lane - 16] = 0.f;
    if (lane < 32) xs[pe + lane] = 0.f;
    const float cmask = q == 0 ? 0.f : 1.f;   // the c = 0 term of the cos fold has no mirror element
    HNO_STAMP(a.stamps, 0);
    if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) { a.stamps[60] = wall_clock64(); a.stamps[62] = clock64(); }
    if (a.stamps && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) a.stamps[58] = wall_clock64();
    if (a.stamps && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) a.stamps[56] = wall_clock64();
    int it = 0;
    for (; plane < planes; plane += nwaves, ++it) {
        HNO_STAMP(a.stamps, 1 + it * 6);
        // ---- stage the plane (flat image); the previous plane's axis-H reads are complete (in-order LDS)
#pragma unroll
        for (int j = 0; j < NV; ++j) *reinterpret_cast<f4v *>(xs + 4 * lane + 256 * j) = rv[j];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int e = 256 * NV + lane + 64 * j;
            if (e < pe) xs[e] = rt[j];
        }
        HNO_STAMP(a.stamps, 2 + it * 6);
        if (plane + nwaves < planes) fetch(plane + nwaves);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        HNO_STAMP(a.stamps, 3 + it * 6);
        // ---- axis W: tiles of 16 rows (rows 32 g + 2 i + parity), the two parities of a group together
#pragma unroll 1
        for (int g = 0; g < N1 / 32; ++g) {
            const float *row0 = xs + (32 * g + 2 * l15) * N2, *row1 = row0 + N2;
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, s0 = c0, c1 = c0, s1 = c0;
            // software pipeline: the LDS reads of k-steps [s + G, s + 2G) are in flight while the MFMAs of [s, s + G) run
            constexpr int G = 3;
            {
                const float *pf0 = row0 + q, *pb0 = row0 + N2 - q, *pf1 = row1 + q, *pb1 = row1 + N2 - q;
                float a0[KC2], b0[KC2], a1_[KC2], b1[KC2];
                auto ld = [&](int ks) {
                    a0[ks] = pf0[4 * ks];
                    b0[ks] = pb0[-4 * ks];
                    a1_[ks] = pf1[4 * ks];
                    b1[ks] = pb1[-4 * ks];
                };
#pragma unroll
                for (int ks = 0; ks < G && ks < KC2; ++ks) ld(ks);
#pragma unroll
                for (int s0_ = 0; s0_ < KC2; s0_ += G) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = s0_ + G; ks < s0_ + 2 * G && ks < KC2; ++ks) ld(ks);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = s0_; ks < s0_ + G && ks < KC2; ++ks) {
                        const float m = ks == 0 ? cmask : 1.f;
                        c0 = mfma16(a0[ks] + m * b0[ks], bwc[ks], c0);
                        c1 = mfma16(a1_[ks] + m * b1[ks], bwc[ks], c1);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const float *sf0 = row0 + a2.Js - q, *sb0 = row0 + N2 - a2.Js + q, *sf1 = row1 + a2.Js - q, *sb1 = row1 + N2 - a2.Js + q;
                float a0[KS2], b0[KS2], a1_[KS2], b1[KS2];
                auto ld = [&](int ks) {
                    a0[ks] = sf0[-4 * ks];
                    b0[ks] = sb0[4 * ks];
                    a1_[ks] = sf1[-4 * ks];
                    b1[ks] = sb1[4 * ks];
                };
#pragma unroll
                for (int ks = 0; ks < G && ks < KS2; ++ks) ld(ks);
#pragma unroll
                for (int s0_ = 0; s0_ < KS2; s0_ += G) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = s0_ + G; ks < s0_ + 2 * G && ks < KS2; ++ks) ld(ks);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = s0_; ks < s0_ + G && ks < KS2; ++ks) {
                        s0 = mfma16(a0[ks] - b0[ks], bws[ks], s0);
                        s1 = mfma16(a1_[ks] - b1[ks], bws[ks], s1);
                    }
                }
            }
            // T rows of the two tiles overwrite columns 0..31 of the same plane rows
            float *t0 = xs + (32 * g + 2 * (q * 4)) * N2 + l15, *t1 = t0 + N2;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                t0[2 * r * N2] = c0[r];
                t0[2 * r * N2 + 16] = s0[r];
                t1[2 * r * N2] = c1[r];
                t1[2 * r * N2 + 16] = s1[r];
            }
        }
        HNO_STAMP(a.stamps, 4 + it * 6);
        {   // leftover row N1 - 1: same operands on the VALU, k split over the 4 lane groups
            const float *row = xs + (N1 - 1) * N2;
            const float *pf = row + q, *pb = row + N2 - q;
            const float *sf = row + a2.Js - q, *sb = row + N2 - a2.Js + q;
            float pc = (pf[0] + cmask * pb[0]) * bwc[0], ps = 0.f;
#pragma unroll
            for (int ks = 1; ks < KC2; ++ks) pc += (pf[4 * ks] + pb[-4 * ks]) * bwc[ks];
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) ps += (sf[-4 * ks] - sb[4 * ks]) * bws[ks];
            pc += __shfl_xor(pc, 16);
            ps += __shfl_xor(ps, 16);
            pc += __shfl_xor(pc, 32);
            ps += __shfl_xor(ps, 32);
            __builtin_amdgcn_sched_barrier(0);   // every lane has read the row before lanes 0..15 overwrite it
            if (q == 0) {
                xs[(N1 - 1) * N2 + l15] = pc;
                xs[(N1 - 1) * N2 + 16 + l15] = ps;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        HNO_STAMP(a.stamps, 5 + it * 6);
        // ---- axis H (fold along n1 while reading), both parts at once: BR = P_Ac -+ Q_As ; BI from Q_Ac, P_As
        {
            float *Yp = Y + (size_t)plane * (2 * p.CP);
            const float *cA = xs + l15, *cB = xs + 16 + l15;   // Ac and As columns
            f32x4 pA = {0.f, 0.f, 0.f, 0.f}, pB = pA, qA = pA, qB = pA;   // P/Q sums of the Ac / As columns
            constexpr int G = 3;
            {
                const float *fA = cA + q * N2, *bA = cA + (N1 - q) * N2, *fB = cB + q * N2, *bB = cB + (N1 - q) * N2;
                float a0[KC1], b0[KC1], a1_[KC1], b1[KC1];
                auto ld = [&](int ks) {
                    a0[ks] = fA[4 * ks * N2];
                    b0[ks] = bA[-4 * ks * N2];   // ks = 0, q = 0 reads the zero row N1
                    a1_[ks] = fB[4 * ks * N2];
                    b1[ks] = bB[-4 * ks * N2];
                };
#pragma unroll
                for (int ks = 0; ks < G && ks < KC1; ++ks) ld(ks);
#pragma unroll
                for (int s0_ = 0; s0_ < KC1; s0_ += G) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = s0_ + G; ks < s0_ + 2 * G && ks < KC1; ++ks) ld(ks);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = s0_; ks < s0_ + G && ks < KC1; ++ks) {
                        pA = mfma16(a0[ks] + b0[ks], bhc[ks], pA);
                        pB = mfma16(a1_[ks] + b1[ks], bhc[ks], pB);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const float *fA = cA + (a1.Js - q) * N2, *bA = cA + (N1 - a1.Js + q) * N2;
                const float *fB = cB + (a1.Js - q) * N2, *bB = cB + (N1 - a1.Js + q) * N2;
                float a0[KS1], b0[KS1], a1_[KS1], b1[KS1];
                auto ld = [&](int ks) {
                    a0[ks] = fA[-4 * ks * N2];
                    b0[ks] = bA[4 * ks * N2];
                    a1_[ks] = fB[-4 * ks * N2];
                    b1[ks] = bB[4 * ks * N2];
                };
#pragma unroll
                for (int ks = 0; ks < G && ks < KS1; ++ks) ld(ks);
#pragma unroll
                for (int s0_ = 0; s0_ < KS1; s0_ += G) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = s0_ + G; ks < s0_ + 2 * G && ks < KS1; ++ks) ld(ks);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = s0_; ks < s0_ + G && ks < KS1; ++ks) {
                        qA = mfma16(a0[ks] - b0[ks], bhs[ks], qA);
                        qB = mfma16(a1_[ks] - b1[ks], bhs[ks], qB);
                    }
                }
            }
            const int k1 = l15;
            if (k1 <= a1.m) {
                // part 0: cos sum of Ac (pA), sin sum of As (qB);  part 1: cos sum of As (pB), sin sum of Ac (qA)
                f32x4 vp0, vm0, vp1, vm1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    vp0[r] = pA[r] - qB[r];
                    vm0[r] = pA[r] + qB[r];
                    vp1[r] = -(qA[r] + pB[r]);
                    vm1[r] = qA[r] - pB[r];
                }
                const int k2 = q * 4;
                *reinterpret_cast<f32x4 *>(Y + ymid(a, plane, 0, a1.m + k1, k2)) = vp0;
                *reinterpret_cast<f32x4 *>(Y + ymid(a, plane, 1, a1.m + k1, k2)) = vp1;
                if (k1 >= 1) {
                    *reinterpret_cast<f32x4 *>(Y + ymid(a, plane, 0, a1.m - k1, k2)) = vm0;
                    *reinterpret_cast<f32x4 *>(Y + ymid(a, plane, 1, a1.m - k1, k2)) = vm1;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        HNO_STAMP(a.stamps, 6 + it * 6);
    }
    if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) { a.stamps[61] = wall_clock64(); a.stamps[63] = clock64(); }
    if (a.stamps && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) a.stamps[59] = wall_clock64();
    if (a.stamps && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) a.stamps[57] = wall_clock64();
}

// ---- forward plane kernel, one wave per plane, planes streamed into LDS by DMA, axis-W result kept in registers ------------
// What bounded the kernel above (profiles/r02_d_pmc_sq_per_kernel.json, in-kernel wall-clock stamps): all 2 048 waves fetch their first
// plane at once (35 MB in flight, ~6 us with no arithmetic running), the prefetch of the next plane costs 64 VGPRs of 16-byte loads
// at 4-byte alignment (split requests), the axis-W result goes through LDS (16 stores + 68 conflicting reads + a fence per plane), and
// every MFMA waits for the VALU fold that writes its A operand into the register the previous MFMA is still reading.
// Here:
//   * a plane reaches LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per instruction, no VGPRs).  Every source range starts at the
//     16-byte boundary below the data (which then sits 0..3 floats into its LDS image), so all requests are aligned;
//   * the unit of work is an ITEM = the rows of one row-pair tile: item X of a plane holds rows n1 = 1 + 16 X + i (i = 0..15) and their
//     mirror rows N1 - n1 (item 0 also row 0, which is its own mirror and runs on the VALU).  For N1 = 65 that is rows [0, 17) + [49, 65)
//     and rows [17, 49): contiguous chunks of <= 10 KiB, so 8 waves x 2 item slots fill the 160 KiB of a CU, two waves share a SIMD and
//     every wave has its next item in flight behind the one it multiplies (counted vmcnt) -- 76 KB in flight per CU all the time;
//   * the C layout of v_mfma_f32_16x16x4_f32 (column k2 on the lane, rows 4 q + r in register r) IS the A-operand layout of the axis-H
//     product when its K index is the tile row: lane group q supplies row 4 q + r at k-step r, and the tables are stored in that order
//     (DhtPlan::hperm).  The fold T[n1] +- T[N1 - n1] is an add / subtract of the two tiles' registers: the axis-W result never touches
//     LDS, the axis-H sums run across the items of a plane in registers;
//   * folds are formed for a whole burst into distinct registers before its MFMAs issue back to back;
//   * nothing is multiplied by bytes outside a row (the c = 0 mirror term is selected away, not masked by a product): slots need no
//     zero guards and a DMA may over-read into the neighbouring rows (the tail pieces of the last plane are clamped into the tensor).
// Requires N1 = 32 NP + 1 (NP = 1, 2), odd N2 with Js2 == 4 KS2 (65 x 65 and 33 x 33 planes), one k tile per axis, x 4-byte aligned.
// ABL: timing ablations (results wrong): 1 = a VALU op in place of every MFMA, 2 = no LDS operand reads
template <int KC2, int KS2, int NP, int NWV, int ABL = 0, int ZL = 0>   // ZL: DhtArgs.zl at compile time (0 rows, 1 z-layout / plain stores, 2 z-layout / write-through)
__global__ __launch_bounds__(64 * NWV, 2) void dht_fwd_plane_dma_kernel(const float *__restrict__ xal, float *__restrict__ Y, DhtArgs a,
                                                                        unsigned shift0, unsigned max_off, int pl_base, int pl_rem,
                                                                        unsigned ldbc) {
    extern __shared__ float lds[];
    auto MM = [](float av, float bv, f32x4 c) -> f32x4 {
        if (ABL == 1) {
            c[0] = fmaf(av, bv, c[0]);
            return c;
        }
        return mfma16(av, bv, c);
    };
    const DhtPlan &p = a.p;
    const Axis &a1 = p.ax[1], &a2 = p.ax[2];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, l15 = lane & 15;
    const int N2 = a2.N;
    constexpr int N1 = 32 * NP + 1;
    const unsigned pe = (unsigned)(N1 * N2);
    // float index of plane (bc, n0) from the aligned base: consecutive (b, c) volumes are ldbc floats apart (N0 pe, or padded up
    // to a multiple of 32: channel-padded activations)
    const int N0p = p.ax[0].N;
    auto plane_f0 = [&](int plane) -> unsigned {
        const int bc = plane / N0p;
        return shift0 + (unsigned)bc * ldbc + (unsigned)(plane - bc * N0p) * pe;
    };
    // chunks of an item (rows [r0, r1) of the plane) and their KiB pieces; item slot = SLOTP pieces
    constexpr int SLOTP = NP == 2 ? 10 : 5;
    constexpr int SLOTF = SLOTP * 256;
    float *ring = lds + (size_t)wave * 2 * SLOTF;
    const unsigned ring_b = (unsigned)(size_t)ring;
    // one workgroup per CU owns a contiguous range of planes: P / G each (pl_base), the first P % G (pl_rem) workgroups one more (the
    // host divides: a 64-bit division here costs every wave ~2 000 cycles before its first load); its waves take them round-robin
    const int bid = blockIdx.x;
    const int p_begin = bid * pl_base + (bid < pl_rem ? bid : pl_rem);
    const int p_end = p_begin + pl_base + (bid < pl_rem ? 1 : 0);
    HNO_STAMP(a.stamps, 20);
    if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[60] = wall_clock64();
    if (a.stamps && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) a.stamps[58] = wall_clock64();
    if (a.stamps && blockIdx.x == gridDim.x / 2 && threadIdx.x == 64 * (NWV - 1)) a.stamps[56] = wall_clock64();
    // rows [r0, r0 + nr) of `plane` -> LDS at byte address dst, NPC pieces
    auto issue_chunk = [&](int plane, int r0, int NPC, unsigned dst) {
        const unsigned f0 = plane_f0(plane) + (unsigned)(r0 * N2);   // float index of the chunk from the aligned base
        const unsigned boff = (f0 & ~3u) * 4u + (unsigned)lane * 16u;
        const unsigned first = __builtin_amdgcn_readfirstlane(boff);
        if (first + (unsigned)NPC * 1024u <= max_off) {   // wave-uniform: every piece inside the tensor
            int j = 0;
            for (; j + 4 <= NPC; j += 4) dma_piece16x4(xal, boff + 1024u * j, __builtin_amdgcn_readfirstlane(dst + 1024u * j));
            for (; j < NPC; ++j) dma_piece16(xal, boff + 1024u * j, __builtin_amdgcn_readfirstlane(dst + 1024u * j));
        } else {
            for (int j = 0; j < NPC; ++j) {
                unsigned off = boff + 1024u * j;
                off = off < max_off ? off : max_off;                 // the last plane's tail pieces stay inside the tensor
                dma_piece16(xal, off, __builtin_amdgcn_readfirstlane(dst + 1024u * j));
            }
        }
    };
    auto issue_item = [&](int plane, int X) {
        const unsigned dst = ring_b + (unsigned)(X & 1) * (SLOTF * 4);
        if (NP == 1) issue_chunk(plane, 0, 5, dst);
        else if (X == 0) {
            issue_chunk(plane, 0, 5, dst);
            issue_chunk(plane, 49, 5, dst + 5 * 1024);
        } else
            issue_chunk(plane, 17, 9, dst);
    };
    // Start-up: the table rows (lane order, copy b % kDmaTabCopies: see get_plan) are requested first, the first two items right behind
    // them.  The table loads are asm loads hipcc does not count: an ordinary load would make it drain the whole vector-memory queue
    // (vmcnt(0), both items) at the tables' first use; here the wait leaves the NPRO younger DMA pieces in flight.
    int plane = p_begin + wave;
    float bwc[KC2], bws[KS2], bhc[NP][4], bhs[NP][4];
    {
        const float *tb = p.tables + p.dmatab + (size_t)(blockIdx.x % kDmaTabCopies) * p.dmatab_stride;
        const unsigned lo = (unsigned)lane * 4u;
#define HNO_TLOAD(dst, row) asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(dst) : "v"(lo + ((row) >> 4) * 4096u), "s"(tb), "n"(((row) & 15) * 256))
#pragma unroll
        for (int ks = 0; ks < KC2; ++ks) HNO_TLOAD(bwc[ks], ks);
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) HNO_TLOAD(bws[ks], KC2 + ks);
#pragma unroll
        for (int X = 0; X < NP; ++X)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                HNO_TLOAD(bhc[X][r], KC2 + KS2 + (X * 4 + r) * 2);
                HNO_TLOAD(bhs[X][r], KC2 + KS2 + (X * 4 + r) * 2 + 1);
            }
#undef HNO_TLOAD
    }
    HNO_STAMP(a.stamps, 21);
    int npro = 0;   // DMA pieces issued behind the table loads
    if (plane < p_end) {
        issue_item(plane, 0);
        npro = NP == 2 ? 10 : 5;
        if (NP == 2) {
            issue_item(plane, 1);
            npro += 9;
        } else if (plane + NWV < p_end) {
            issue_item(plane + NWV, 1);   // NP == 1: slot parity alternates between planes
            npro += 5;
        }
    }
    if (npro == 19) dma_wait<19>();
    else if (npro == 10) dma_wait<10>();
    else if (npro == 5) dma_wait<5>();
    else dma_wait<0>();
#pragma unroll
    for (int ks = 0; ks < KC2; ++ks) asm volatile("" : "+v"(bwc[ks]));
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) asm volatile("" : "+v"(bws[ks]));
#pragma unroll
    for (int X = 0; X < NP; ++X)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            asm volatile("" : "+v"(bhc[X][r]));
            asm volatile("" : "+v"(bhs[X][r]));
        }
    __builtin_amdgcn_sched_barrier(0);
    HNO_STAMP(a.stamps, 22);
    const bool q0 = q == 0;
    // z-layout: float offsets of this lane's two 32-byte pieces (rows m +- k1, k2 tile q) of plane 0; a plane adds 8 floats
    const size_t zoff_p = ((size_t)((a1.m + (l15 <= a1.m ? l15 : 0)) * 4 + q) * a.zplanes) * 8;
    const size_t zoff_m = ((size_t)((a1.m - (l15 <= a1.m ? l15 : 0)) * 4 + q) * a.zplanes) * 8;
    // plane rows of item (plane, X) in its slot: row r of the plane at rowsP + r N2 (rows of the tile) / rowsM + r N2 (mirror rows)
    auto item_rows = [&](int pl, int X, int slot, const float *&rowsP, const float *&rowsM) {
        const unsigned g0 = plane_f0(pl);
        const float *sl = ring + slot * SLOTF;
        if (NP == 1) rowsP = rowsM = sl + (g0 & 3u);
        else if (X == 0) {
            rowsP = sl + (g0 & 3u);
            rowsM = sl + 5 * 256 + ((g0 + 49u * N2) & 3u) - 49 * N2;
        } else
            rowsP = rowsM = sl + ((g0 + 17u * N2) & 3u) - 17 * N2;
    };
    // cosine-part operands of the two tiles of an item (column c = 4 ks + q and its mirror N2 - c)
    auto read_cos = [&](int pl, int X, int slot, float (&ra0)[KC2], float (&rb0)[KC2], float (&ra1)[KC2], float (&rb1)[KC2]) {
        if (ABL == 2) {
#pragma unroll
            for (int ks = 0; ks < KC2; ++ks) ra0[ks] = rb0[ks] = ra1[ks] = rb1[ks] = bwc[ks];
            return;
        }
        const float *rowsP, *rowsM;
        item_rows(pl, X, slot, rowsP, rowsM);
        const int rp = 1 + 16 * X + l15;
        const float *row0 = rowsP + rp * N2, *row1 = rowsM + (N1 - rp) * N2;
        const float *pf0 = row0 + q, *pb0 = row0 + N2 - q - 4 * (KC2 - 1), *pf1 = row1 + q, *pb1 = row1 + N2 - q - 4 * (KC2 - 1);
#pragma unroll
        for (int ks = 0; ks < KC2; ++ks) {
            ra0[ks] = pf0[4 * ks];
            rb0[ks] = pb0[4 * (KC2 - 1 - ks)];
            ra1[ks] = pf1[4 * ks];
            rb1[ks] = pb1[4 * (KC2 - 1 - ks)];
        }
    };
    // The loop is software-pipelined over items: the cosine operands of item t + 1 are read from LDS (its DMA has landed: counted wait)
    // before the sine / axis-H MFMAs of item t are issued, so an item starts on operands that are already in registers.
    float a0[KC2], b0[KC2], a1_[KC2], b1[KC2];
    if (plane < p_end) {
        if (NP == 2) dma_wait<9>();
        else if (plane + NWV < p_end) dma_wait<5>();
        else dma_wait<0>();
        if (!(a.dbg & 1)) read_cos(plane, 0, 0, a0, b0, a1_, b1);
    }
    int it = 0;
    for (; plane < p_end; plane += NWV, ++it) {
        const bool more = plane + NWV < p_end;
        f32x4 pA = {0.f, 0.f, 0.f, 0.f}, pB = pA, qA = pA, qB = pA;   // axis-H sums of the plane
#pragma unroll
        for (int X = 0; X < NP; ++X) {
            HNO_STAMP(a.stamps, 24 + (it * NP + X) * 6);
            const int slot = NP == 2 ? X : (it & 1);
            const bool refill = NP == 2 ? more : (plane + 2 * NWV < p_end);
            const int rf_plane = NP == 2 ? plane + NWV : plane + 2 * NWV;
            if (a.dbg & 1) {   // timing aid: DMA stream and waits only
                if (refill) issue_item(rf_plane, slot);
                if (refill) {
                    if (NP == 2 && X == 0) dma_wait<10>();
                    else if (NP == 2) dma_wait<9>();
                    else dma_wait<5>();
                } else
                    dma_wait<0>();
                continue;
            }
            const float *rowsP, *rowsM;
            item_rows(plane, X, slot, rowsP, rowsM);
            const int rp = 1 + 16 * X + l15;
            const float *row0 = rowsP + rp * N2, *row1 = rowsM + (N1 - rp) * N2;
            // ---- the item's remaining operand reads (sine part, row 0) go out first: they land behind the cosine part's MFMAs
            float sa0[KS2], sb0[KS2], sa1[KS2], sb1[KS2];
            if (ABL == 2) {
#pragma unroll
                for (int ks = 0; ks < KS2; ++ks) sa0[ks] = sb0[ks] = sa1[ks] = sb1[ks] = bws[ks];
            } else {
                const float *sf0 = row0 + a2.Js - q - 4 * (KS2 - 1), *sb0_ = row0 + N2 - a2.Js + q;
                const float *sf1 = row1 + a2.Js - q - 4 * (KS2 - 1), *sb1_ = row1 + N2 - a2.Js + q;
#pragma unroll
                for (int ks = 0; ks < KS2; ++ks) {
                    sa0[ks] = sf0[4 * (KS2 - 1 - ks)];
                    sb0[ks] = sb0_[4 * ks];
                    sa1[ks] = sf1[4 * (KS2 - 1 - ks)];
                    sb1[ks] = sb1_[4 * ks];
                }
            }
            float r0a[KC2], r0b[KC2], r0c[KS2], r0d[KS2];   // row 0 (item 0 only)
            if (X == 0 && ABL == 2) {
#pragma unroll
                for (int ks = 0; ks < KC2; ++ks) r0a[ks] = r0b[ks] = bwc[ks];
#pragma unroll
                for (int ks = 0; ks < KS2; ++ks) r0c[ks] = r0d[ks] = bws[ks];
            } else if (X == 0) {
                const float *rz = rowsP;
                const float *pf = rz + q, *pb = rz + N2 - q - 4 * (KC2 - 1), *sf = rz + a2.Js - q - 4 * (KS2 - 1), *sb = rz + N2 - a2.Js + q;
#pragma unroll
                for (int ks = 0; ks < KC2; ++ks) {
                    r0a[ks] = pf[4 * ks];
                    r0b[ks] = pb[4 * (KC2 - 1 - ks)];
                }
#pragma unroll
                for (int ks = 0; ks < KS2; ++ks) {
                    r0c[ks] = sf[4 * (KS2 - 1 - ks)];
                    r0d[ks] = sb[4 * ks];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // c = 0 has no mirror element: what was read there (the next row's first element, or bytes outside the chunk) is dropped
            // by a select, so no garbage is ever multiplied
            b0[0] = q0 ? 0.f : b0[0];
            b1[0] = q0 ? 0.f : b1[0];
#pragma unroll
            for (int ks = 0; ks < KC2; ++ks) {
                a0[ks] += b0[ks];
                a1_[ks] += b1[ks];
            }
            // ---- cosine part.  Its first MFMAs cover the latency of the reads just issued; as soon as those have returned the slot is
            //      refilled (the earlier a slot is free, the more bytes the wave keeps in flight: with the refill behind the whole
            //      cosine part the kernel was latency-bound at 4.2 TB/s).  The VALU work of the next bursts (sine folds, row 0) rides
            //      between the remaining MFMAs: an MFMA holds the matrix pipe for 32 cycles but the issue port only for 8
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, s0 = c0, c1 = c0, s1 = c0;
            float pc = 0.f, ps = 0.f;   // row 0: this lane group's share of the sums over the folded columns
            constexpr int KSPLIT = KC2 / 2;
#pragma unroll
            for (int ks = 0; ks < KSPLIT; ++ks) {
                c0 = MM(a0[ks], bwc[ks], c0);
                c1 = MM(a1_[ks], bwc[ks], c1);
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every LDS read of the item has returned
            HNO_STAMP(a.stamps, 25 + (it * NP + X) * 6);
            if (refill) issue_item(rf_plane, slot);
            HNO_STAMP(a.stamps, 26 + (it * NP + X) * 6);
#pragma unroll
            for (int ks = KSPLIT; ks < KC2; ++ks) {
                __builtin_amdgcn_sched_barrier(0);
                c0 = MM(a0[ks], bwc[ks], c0);
                c1 = MM(a1_[ks], bwc[ks], c1);
#pragma unroll
                for (int j = 2 * (ks - KSPLIT); j < 2 * (ks - KSPLIT) + 2; ++j) {
                    if (j < KS2) {
                        sa0[j] -= sb0[j];
                        sa1[j] -= sb1[j];
                    }
                    if (X == 0) {
                        if (j < KC2) pc = fmaf(r0a[j] + ((j == 0 && q0) ? 0.f : r0b[j]), bwc[j], pc);
                        if (j < KS2) ps = fmaf(r0c[j] - r0d[j], bws[j], ps);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- sine part, with the axis-H products of the finished cosine part between its MFMAs (four independent chains).
            //      Axis H, this item's 16 row pairs: P = cos sums of the folded rows, Q = sin sums of the row differences
            float fa[4], fb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                fa[r] = c0[r] + c1[r];
                fb[r] = c0[r] - c1[r];
            }
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                __builtin_amdgcn_sched_barrier(0);
                s0 = MM(sa0[ks], bws[ks], s0);
                s1 = MM(sa1[ks], bws[ks], s1);
                if ((ks & 1) == 0 && ks / 2 < 4) pA = MM(fa[ks / 2], bhc[X][ks / 2], pA);
                if ((ks & 1) == 1 && ks / 2 < 4) qA = MM(fb[ks / 2], bhs[X][ks / 2], qA);
            }
            if (KS2 < 8) {
#pragma unroll
                for (int r = KS2 / 2; r < 4; ++r) {
                    pA = MM(fa[r], bhc[X][r], pA);
                    qA = MM(fb[r], bhs[X][r], qA);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- the next item has landed when at most the pieces of the refill just issued are outstanding (vector memory completes
            //      in order; Y stores in between only make the wait a little longer than necessary): fetch its cosine operands now
            {
                const bool have_next = X + 1 < NP || more;
                if (have_next) {
                    if (refill) {
                        if (NP == 2 && X == 0) dma_wait<10>();
                        else if (NP == 2) dma_wait<9>();
                        else dma_wait<5>();
                    } else
                        dma_wait<0>();
                    if (X + 1 < NP) read_cos(plane, X + 1, X + 1, a0, b0, a1_, b1);
                    else read_cos(plane + NWV, 0, NP == 2 ? 0 : ((it + 1) & 1), a0, b0, a1_, b1);
                }
            }
            HNO_STAMP(a.stamps, 27 + (it * NP + X) * 6);
            // row 0: the sum over the four lane groups is the K sum of an MFMA against ones
            if (X == 0) {
                pA = MM(pc, 1.f, pA);
                pB = MM(ps, 1.f, pB);
            }
            float fc[4], fd[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                fc[r] = s0[r] + s1[r];
                fd[r] = s0[r] - s1[r];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pB = MM(fc[r], bhc[X][r], pB);
                qB = MM(fd[r], bhs[X][r], qB);
            }
            HNO_STAMP(a.stamps, 28 + (it * NP + X) * 6);
        }
        if (!(a.dbg & 1)) {
            float *Yp = Y + (size_t)plane * (2 * p.CP);
            const int k1 = l15;
            if (k1 <= a1.m) {
                // part 0: cos sum of Ac (pA), sin sum of As (qB);  part 1: cos sum of As (pB), sin sum of Ac (qA)
                f32x4 vp0, vm0, vp1, vm1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    vp0[r] = pA[r] - qB[r];
                    vm0[r] = pA[r] + qB[r];
                    vp1[r] = -(qA[r] + pB[r]);
                    vm1[r] = qA[r] - pB[r];
                }
                const int k2 = q * 4;
                if constexpr (ZL == 1) {
                    // z-layout, plain stores: the XCD's L2 merges the 32-byte pieces of neighbouring planes into lines (write-through
                    // pieces, HNO_MID_ZLAYOUT=2, cost this kernel 4.4 us: 19.3 against 23.7 us per launch in the step)
                    float *d0 = Y + zoff_p + (size_t)plane * 8, *d1 = Y + zoff_m + (size_t)plane * 8;
                    *reinterpret_cast<f32x4 *>(d0) = vp0;
                    *reinterpret_cast<f32x4 *>(d0 + 4) = vp1;
                    if (k1 >= 1) {
                        *reinterpret_cast<f32x4 *>(d1) = vm0;
                        *reinterpret_cast<f32x4 *>(d1 + 4) = vm1;
                    }
                } else if constexpr (ZL == 2) {
                    float *d0 = Y + zoff_p + (size_t)plane * 8, *d1 = Y + zoff_m + (size_t)plane * 8;
                    store16_wt(d0, vp0);
                    store16_wt(d0 + 4, vp1);
                    if (k1 >= 1) {
                        store16_wt(d1, vm0);
                        store16_wt(d1 + 4, vm1);
                    }
                } else {
                    // write-through stores: the 11.6 MB of Y would otherwise sit dirty in the L2s until the end-of-kernel write-back
                    store16_wt(Y + ymid(a, plane, 0, a1.m + k1, k2), vp0);
                    store16_wt(Y + ymid(a, plane, 1, a1.m + k1, k2), vp1);
                    if (k1 >= 1) {
                        store16_wt(Y + ymid(a, plane, 0, a1.m - k1, k2), vm0);
                        store16_wt(Y + ymid(a, plane, 1, a1.m - k1, k2), vm1);
                    }
                }
            }
        }
    }
    HNO_STAMP(a.stamps, 23);
    if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[61] = wall_clock64();
    if (a.stamps && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) a.stamps[59] = wall_clock64();
    if (a.stamps && blockIdx.x == gridDim.x / 2 && threadIdx.x == 64 * (NWV - 1)) a.stamps[57] = wall_clock64();
}

// ---- forward, axis D + (Re -/+ Im) + crop: one wave per (bc, column tile) ----------------
__global__ __launch_bounds__(64) void dht_fwd_d_kernel(const float *__restrict__ Y, float *__restrict__ out, DhtArgs a) {
    extern __shared__ float lds[];
    const DhtPlan &p = a.p;
    const Axis &a0 = p.ax[0], &a2 = p.ax[2];
    const int lane = threadIdx.x;
    const int ntab = a0.KT * (a0.KcP + a0.KsP) * 16;
    for (int i = lane; i < ntab; i += 64) lds[i] = p.tables[a0.cosF + i];
    __syncthreads();
    const float *cosD = lds, *sinD = lds + a0.KT * a0.KcP * 16;
    const int ct = blockIdx.x;  // column tile: (k1s, kt2)
    const int bc = blockIdx.y;
    const int k1s = ct / a2.KT, kt2 = ct % a2.KT;
    const int N0 = a0.N;
    const size_t pstride = (size_t)2 * p.CP;  // floats per n0 plane
    const float *Yb = Y + (size_t)bc * N0 * pstride;
    const int colR = ct * 16 + (lane & 15), colI = p.CP + colR;
    const int q = lane >> 4;
    for (int kt0 = 0; kt0 < a0.KT; ++kt0) {
        f32x4 PR = {0.f, 0.f, 0.f, 0.f}, PI = PR, QR = PR, QI = PR;
        const float *bc_ = cosD + kt0 * a0.KcP * 16, *bs_ = sinD + kt0 * a0.KsP * 16;
        // operands are gathered four k-steps at a time (16 independent loads in flight)
        for (int ks0 = 0; ks0 < a0.KcP / 4; ks0 += 4) {
            float vr[4], vi[4], wr[4], wi[4], bb[4];
            bool inn[4], prd[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ks = ks0 + u;
                const int c = ks * 4 + q;
                inn[u] = ks < a0.KcP / 4 && c <= a0.J;
                prd[u] = inn[u] && c >= 1 && c <= a0.Js;
                const int c1 = inn[u] ? c : 0, c2 = prd[u] ? N0 - c : 0;
                vr[u] = Yb[c1 * pstride + colR];
                vi[u] = Yb[c1 * pstride + colI];
                wr[u] = Yb[c2 * pstride + colR];
                wi[u] = Yb[c2 * pstride + colI];
                bb[u] = ks < a0.KcP / 4 ? bc_[(ks * 4 + q) * 16 + (lane & 15)] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float ar = inn[u] ? vr[u] + (prd[u] ? wr[u] : 0.f) : 0.f;
                const float ai = inn[u] ? vi[u] + (prd[u] ? wi[u] : 0.f) : 0.f;
                PR = mfma16(ar, bb[u], PR);
                PI = mfma16(ai, bb[u], PI);
            }
        }
        for (int ks0 = 0; ks0 < a0.KsP / 4; ks0 += 4) {
            float vr[4], vi[4], wr[4], wi[4], bb[4];
            bool inn[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ks = ks0 + u;
                const int kk = ks * 4 + q;
                inn[u] = ks < a0.KsP / 4 && kk < a0.Js;
                const int j = inn[u] ? a0.Js - kk : 0;
                const int j2 = inn[u] ? N0 - j : 0;
                vr[u] = Yb[j * pstride + colR];
                vi[u] = Yb[j * pstride + colI];
                wr[u] = Yb[j2 * pstride + colR];
                wi[u] = Yb[j2 * pstride + colI];
                bb[u] = ks < a0.KsP / 4 ? bs_[(ks * 4 + q) * 16 + (lane & 15)] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                QR = mfma16(inn[u] ? vr[u] - wr[u] : 0.f, bb[u], QR);
                QI = mfma16(inn[u] ? vi[u] - wi[u] : 0.f, bb[u], QI);
            }
        }
        fwd_d_store(a, out, bc, k1s, kt2, kt0, lane, PR, PI, QR, QI);
    }
}

// Fast form for m0 <= 15 (one k0 tile) and N0 <= 4*KCS - 1: every gather load of the wave (and its
// table operands, straight from global memory) is issued before the first use, so the kernel is one
// memory round trip instead of one per group of four k-steps, and needs neither LDS nor a barrier.
template <int KCS, int KSS>
__global__ __launch_bounds__(64) void dht_fwd_d_fast_kernel(const float *__restrict__ Y, float *__restrict__ out, DhtArgs a) {
    const DhtPlan &p = a.p;
    const Axis &a0 = p.ax[0], &a2 = p.ax[2];
    const int lane = threadIdx.x;
    const bool st = a.stamps && blockIdx.x == 3 && blockIdx.y == 5 && lane == 0;
    if (st) { a.stamps[0] = clock64(); a.stamps[10] = wall_clock64(); }
    if (a.stamps && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) a.stamps[11] = wall_clock64();
    if (a.stamps && blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1 && lane == 0) a.stamps[12] = wall_clock64();
    const int ct = blockIdx.x, bc = blockIdx.y;
    const int k1s = ct / a2.KT, kt2 = ct % a2.KT;
    const int N0 = a0.N;
    const size_t pstride = (size_t)2 * p.CP;
    const float *Yb = Y + (size_t)bc * N0 * pstride;
    const int colR = ct * 16 + (lane & 15), colI = p.CP + colR;
    const int q = lane >> 4;
    const float *__restrict__ tc = p.tables + a0.cosF, *__restrict__ ts = p.tables + a0.sinF;
    const int nkc = a0.KcP / 4, nks = a0.KsP / 4;
    // every row of the column tile is loaded ONCE: rows c and N0 - c give the even combination (cosine part) and the odd one (sine
    // part).  The sine table is stored by kk = Js - j; reading it at kk = Js - c puts its rows in the cosine part's k order, so both
    // products take their A operands from the same registers (the first version loaded every row twice: 85 -> 52 loads per lane).
    (void)KSS;
    float vr[KCS], vi[KCS], wr[KCS], wi[KCS], bcv[KCS], bsv[KCS];
    bool prd[KCS];
#pragma unroll
    for (int ks = 0; ks < KCS; ++ks) {
        const int c = ks * 4 + q;
        const bool inn = ks < nkc && c <= a0.J;
        prd[ks] = inn && c >= 1 && c <= a0.Js;
        const int c1 = inn ? c : 0, c2 = prd[ks] ? N0 - c : 0;
        vr[ks] = Yb[c1 * pstride + colR];
        vi[ks] = Yb[c1 * pstride + colI];
        wr[ks] = Yb[c2 * pstride + colR];
        wi[ks] = Yb[c2 * pstride + colI];
        bcv[ks] = ks < nkc ? tc[(ks * 4 + q) * 16 + (lane & 15)] : 0.f;   // zero rows beyond J mask the operand
        const int kk = prd[ks] ? a0.Js - c : 0;
        const float sv = ts[kk * 16 + (lane & 15)];
        bsv[ks] = (prd[ks] && kk < 4 * nks) ? sv : 0.f;
    }
    if (st) a.stamps[1] = clock64();
    f32x4 PR = {0.f, 0.f, 0.f, 0.f}, PI = PR, QR = PR, QI = PR;
#pragma unroll
    for (int ks = 0; ks < KCS; ++ks) {
        PR = mfma16(vr[ks] + (prd[ks] ? wr[ks] : 0.f), bcv[ks], PR);
        PI = mfma16(vi[ks] + (prd[ks] ? wi[ks] : 0.f), bcv[ks], PI);
    }
#pragma unroll
    for (int ks = 0; ks < KCS; ++ks) {
        QR = mfma16(prd[ks] ? vr[ks] - wr[ks] : 0.f, bsv[ks], QR);
        QI = mfma16(prd[ks] ? vi[ks] - wi[ks] : 0.f, bsv[ks], QI);
    }
    if (st) { asm volatile("s_nop 0" :: "v"(PR[0]), "v"(QI[0])); a.stamps[2] = clock64(); }
    fwd_d_store(a, out, bc, k1s, kt2, 0, lane, PR, PI, QR, QI);
    if (st) { a.stamps[3] = clock64(); a.stamps[13] = wall_clock64(); }
    if (a.stamps && blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1 && lane == 0) a.stamps[14] = wall_clock64();
}

// ---- inverse, axis D: spectrum block -> E[bc][n0][part][k1s][k2] --------------------------
__device__ __forceinline__ float zk_load(const float *__restrict__ zb, int k0, int k1, int k2, int m0, int m1, int m2,
                                         const DhtArgs &a) {
    const int o0 = kept_pos(k0, m0, a.full0), o1 = kept_pos(k1, m1, a.full1), o2 = kept_pos(k2, m2, a.full2);
    const bool ok = (o0 >= 0) & (o1 >= 0) & (o2 >= 0);
    const size_t idx = ok ? ((size_t)o0 * (2 * m1 + a.full1) + o1) * (2 * m2 + a.full2) + o2 : 0;
    const float v = zb[idx];
    return ok ? v : 0.f;
}

// NTM > 0: fast form for NT <= NTM output tiles -- the table operands of every (tile, k-step) are loaded
// straight from global memory into registers together with the spectrum gathers, so the wave makes
// one memory round trip and needs neither LDS nor a barrier.  NTM == 0: tables staged in LDS.
template <int NTM>
__global__ __launch_bounds__(64) void dht_inv_d_kernel(const float *__restrict__ z, float *__restrict__ E, DhtArgs a) {
    extern __shared__ float lds[];
    const DhtPlan &p = a.p;
    const Axis &a0 = p.ax[0], &a1 = p.ax[1], &a2 = p.ax[2];
    const int lane = threadIdx.x;
    constexpr int KSM = 8;
    constexpr int NTR = NTM > 0 ? NTM : 1;
    float tcv[NTR][KSM], tsv[NTR][KSM];
    const float *cosD = lds, *sinD = lds + a0.NT * a0.KmP * 16;
    if (NTM > 0) {
        const float *__restrict__ tc = p.tables + a0.cosI, *__restrict__ ts = p.tables + a0.sinI;
#pragma unroll
        for (int nt = 0; nt < NTR; ++nt)
#pragma unroll
            for (int ks = 0; ks < KSM; ++ks) {
                const bool ok = nt < a0.NT && ks < a0.KmP / 4;
                const int idx = ok ? (nt * a0.KmP + ks * 4 + (lane >> 4)) * 16 + (lane & 15) : 0;
                tcv[nt][ks] = tc[idx];
                tsv[nt][ks] = ts[idx];
            }
    } else {
        const int ntab = 2 * a0.NT * a0.KmP * 16;
        for (int i = lane; i < ntab; i += 64) lds[i] = p.tables[a0.cosI + i];
        __syncthreads();
    }
    const int ct = blockIdx.x, bc = blockIdx.y;
    const int k1s = ct / a2.KT, kt2 = ct % a2.KT;
    const int m0 = a0.m, m1 = a1.m, m2 = a2.m, N0 = a0.N;
    const float *zb = z + (size_t)bc * (2 * m0 + a.full0) * (2 * m1 + a.full1) * (2 * m2 + a.full2);
    const int q = lane >> 4;
    const int k1 = k1s - m1;
    const int k2 = kt2 * 16 + (lane & 15);  // A-operand row of this lane
    // A operands for all k steps (k0 = ks*4 + q): Gs (cos part) and Gd (sin part), re & im
    const int KS = a0.KmP / 4;
    const size_t pstride = (size_t)2 * p.CP;
    float *Eb = E + (size_t)bc * N0 * pstride;
    // operands do not depend on the output tile: gather them once (up to 8 k-steps = m0 <= 31)
    float gsr[KSM], gsi[KSM], gdr[KSM], gdi[KSM];
    // Fourier layout: re / im planes of this (b, c)
    const int fb = a.mode ? bc / a.C : 0, fc = a.mode ? bc - fb * a.C : 0;
    const size_t msz = (size_t)(2 * m0 + a.full0) * (2 * m1) * m2;
    const float *sr = z + ((size_t)(fb * 2 + 0) * a.C + fc) * msz, *si = z + ((size_t)(fb * 2 + 1) * a.C + fc) * msz;
#pragma unroll
    for (int ks = 0; ks < KSM; ++ks) {
        gsr[ks] = gsi[ks] = gdr[ks] = gdi[ks] = 0.f;
        const int k0 = ks * 4 + q;
        if (a.mode != 0) {
            // G'(k) = w(k2) * S[k] on kept (k0, k1), k2 in [0, m2); no conjugate partner (the c2r weights
            // already account for the omitted half)
            const int o1 = kept_pos(k1, m1);
            if (ks < KS && k0 <= m0 && k2 < m2 && o1 >= 0) {
                const float w = (a.mode == 2 && k2 > 0) ? 2.f : 1.f;
                const int op = kept_pos(k0, m0, a.full0), om = k0 >= 1 ? kept_pos(-k0, m0, a.full0) : -1;
                const size_t ip = ((size_t)(op >= 0 ? op : 0) * (2 * m1) + o1) * m2 + k2;
                const size_t im = ((size_t)(om >= 0 ? om : 0) * (2 * m1) + o1) * m2 + k2;
                const float pr = op >= 0 ? w * sr[ip] : 0.f, pi = op >= 0 ? w * si[ip] : 0.f;
                const float mr = om >= 0 ? w * sr[im] : 0.f, mi = om >= 0 ? w * si[im] : 0.f;
                gsr[ks] = pr + mr;
                gsi[ks] = pi + mi;
                gdr[ks] = k0 >= 1 ? pr - mr : 0.f;
                gdi[ks] = k0 >= 1 ? pi - mi : 0.f;
            }
        } else if (ks < KS && k0 <= m0 && k2 <= m2) {
            const float va = zk_load(zb, k0, k1, k2, m0, m1, m2, a);
            const float vb = k2 >= 1 ? zk_load(zb, -k0, -k1, -k2, m0, m1, m2, a) : 0.f;
            // G'(+k0) = (va + vb) + i (vb - va)
            gsr[ks] = va + vb;
            gsi[ks] = vb - va;
            if (k0 >= 1) {
                const float vc = zk_load(zb, -k0, k1, k2, m0, m1, m2, a);
                const float vd = k2 >= 1 ? zk_load(zb, k0, -k1, -k2, m0, m1, m2, a) : 0.f;
                // G'(-k0) = (vc + vd) + i (vd - vc)
                gdr[ks] = gsr[ks] - (vc + vd);
                gdi[ks] = gsi[ks] - (vd - vc);
                gsr[ks] += vc + vd;
                gsi[ks] += vd - vc;
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < (NTM > 0 ? NTM : 64); ++nt) {
        if (nt >= a0.NT) break;
        f32x4 UR = {0.f, 0.f, 0.f, 0.f}, UI = UR, VR = UR, VI = UR;
#pragma unroll
        for (int ks = 0; ks < KSM; ++ks) {
            if (ks < KS) {
                const float bcv = NTM > 0 ? tcv[nt < NTR ? nt : 0][ks] : cosD[(nt * a0.KmP + ks * 4 + q) * 16 + (lane & 15)];
                const float bsv = NTM > 0 ? tsv[nt < NTR ? nt : 0][ks] : sinD[(nt * a0.KmP + ks * 4 + q) * 16 + (lane & 15)];
                UR = mfma16(gsr[ks], bcv, UR);
                UI = mfma16(gsi[ks], bcv, UI);
                VR = mfma16(gdr[ks], bsv, VR);
                VI = mfma16(gdi[ks], bsv, VI);
            }
        }
        // lane holds rows k2' = kt2*16 + q*4 + r, column n0 = nt*16 + (lane&15)
        const int n0 = nt * 16 + (lane & 15);
        if (n0 <= a0.J) {
            f32x4 er, ei, fr, fi;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                er[r] = UR[r] - VI[r];  // E[n0]    = U + iV
                ei[r] = UI[r] + VR[r];
                fr[r] = UR[r] + VI[r];  // E[N0-n0] = U - iV
                fi[r] = UI[r] - VR[r];
            }
            const size_t col = (size_t)k1s * a2.KP + kt2 * 16 + q * 4;
            *reinterpret_cast<f32x4 *>(Eb + (size_t)n0 * pstride + col) = er;
            *reinterpret_cast<f32x4 *>(Eb + (size_t)n0 * pstride + p.CP + col) = ei;
            if (n0 >= 1 && n0 <= a0.Js) {
                *reinterpret_cast<f32x4 *>(Eb + (size_t)(N0 - n0) * pstride + col) = fr;
                *reinterpret_cast<f32x4 *>(Eb + (size_t)(N0 - n0) * pstride + p.CP + col) = fi;
            }
        }
    }
}

// ---- inverse, axes H and W, one (bc, n0) plane per workgroup iteration -------------------
// The intermediate plane is copied raw into LDS (prefetched one plane ahead); the +-k1 fold is
// done while the axis-H operands are read.  The axis-W result goes to an LDS image of the output
// plane, and a final flat pass applies scale / residual / activation with fully coalesced
// loads and stores.
template <int MAXE, int NTH = 256>
__global__ __launch_bounds__(NTH) void dht_inv_plane_kernel(const float *__restrict__ E, const float *__restrict__ addend,
                                                            float *__restrict__ out, DhtArgs a) {
    extern __shared__ float lds[];
    const DhtPlan &p = a.p;
    const Axis &a1 = p.ax[1], &a2 = p.ax[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N1 = a1.N, N2 = a2.N;
    float *tabH = lds + p.i_tabH, *tabW = lds + p.i_tabW;
    float *Es = lds + p.i_Es, *Ed = lds + p.i_Ed, *FR = lds + p.i_FR, *FI = lds + p.i_FI, *O = lds + p.i_O;
    const float *cosH = tabH, *sinH = tabH + a1.NT * a1.KmP * 16;
    const float *cosW = tabW, *sinW = tabW + a2.NT * a2.KmP * 16;
    for (int i = tid; i < 2 * a1.NT * a1.KmP * 16; i += NTH) tabH[i] = p.tables[a1.cosI + i];
    for (int i = tid; i < 2 * a2.NT * a2.KmP * 16; i += NTH) tabW[i] = p.tables[a2.cosI + i];
    for (int i = tid; i < (p.MP1 - N1) * p.ldF; i += NTH) {
        FR[N1 * p.ldF + i] = 0.f;
        FI[N1 * p.ldF + i] = 0.f;
    }
    const int planes = a.BC * p.ax[0].N;
    const size_t plane_elems = (size_t)N1 * N2;
    const int MT1 = p.MP1 / 16;
    const int m1 = a1.m;
    const int ne = 2 * p.CP;                  // floats of one intermediate plane
    const int rows = 2 * a2.KP;               // (part, k2)
    const int nitem = rows * a1.KmP;          // folded operand entries Es/Ed[row][k1]
    constexpr int NI = 4;                     // entries per thread held in registers (nitem <= 1024)
    constexpr int NE = MAXE > 0 ? MAXE : 1;
    float ea[NI], eb[NI], ra[NE];
    const int r0 = tid / N2, c0 = tid - r0 * N2, dr = NTH / N2, dc = NTH - dr * N2;
    // item i -> (k1 = i / rows, row = i % rows); rows is a multiple of 32
    auto fetch_e = [&](int plane) {
        const float *Ep = E + (size_t)plane * ne;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int i = tid + NTH * j;
            const int k1 = i / rows, row = i - k1 * rows;
            const int part = row / a2.KP, k2 = row - part * a2.KP;
            const bool ok = i < nitem && k1 <= m1;
            const int kc = ok ? k1 : 0;
            ea[j] = ok ? E[ymid(a, plane, part, m1 + kc, k2)] : 0.f;
            eb[j] = (ok && k1 >= 1) ? E[ymid(a, plane, part, m1 - kc, k2)] : 0.f;
        }
    };
    if (blockIdx.x < planes) fetch_e(blockIdx.x);
    for (int plane = blockIdx.x; plane < planes; plane += gridDim.x) {
        __syncthreads();
        // ---- Es[row][k1] = E[+k1] + E[-k1], Ed = E[+k1] - E[-k1] (k1 = 0: E[0], 0) from the prefetched pairs
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int i = tid + NTH * j;
            if (i < nitem) {
                const int k1 = i / rows, row = i - k1 * rows;
                Es[row * p.ldE + k1] = ea[j] + eb[j];
                Ed[row * p.ldE + k1] = (k1 >= 1 && k1 <= m1) ? ea[j] - eb[j] : 0.f;
            }
        }
        for (int i = tid + NTH * NI; i < nitem; i += NTH) {  // only for very large mode counts
            const float *Ep = E + (size_t)plane * ne;
            const int k1 = i / rows, row = i - k1 * rows;
            const int part = row / a2.KP, k2 = row - part * a2.KP;
            float sv = 0.f, dv = 0.f;
            if (k1 <= m1) {
                const float va = E[ymid(a, plane, part, m1 + k1, k2)];
                const float vb = k1 >= 1 ? E[ymid(a, plane, part, m1 - k1, k2)] : 0.f;
                sv = va + vb;
                dv = k1 >= 1 ? va - vb : 0.f;
            }
            Es[row * p.ldE + k1] = sv;
            Ed[row * p.ldE + k1] = dv;
        }
        if (plane + gridDim.x < planes) fetch_e(plane + gridDim.x);
        // residual of THIS plane: issued now, consumed in the epilogue after both MFMA stages
        const float *ad = addend ? addend + plane_base(a, plane, plane_elems) : nullptr;
        if (MAXE > 0 && ad) {
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const unsigned e = tid + (unsigned)NTH * j;
                ra[j] = e < plane_elems ? ad[e] : 0.f;
            }
        }
        __syncthreads();
        // ---- axis H: F[n1][k2] = sum_k1 E[k1][k2] e^{+i th k1 n1}
        const int ntaskH = (a.dbg & 1) ? 0 : a1.NT * a2.KT * 2;
        for (int t = wave; t < ntaskH; t += (NTH / 64)) {
            const int part = t & 1, kt2 = (t >> 1) % a2.KT, nt1 = (t >> 1) / a2.KT;
            const int rR = kt2 * 16, rI = a2.KP + kt2 * 16;
            const float *bc = cosH + nt1 * a1.KmP * 16, *bs = sinH + nt1 * a1.KmP * 16;
            f32x4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = acc1;
            // FR = cos * Es_R - sin * Ed_I ; FI = cos * Es_I + sin * Ed_R
            tile_mma2(Es + (part ? rI : rR) * p.ldE, bc, a1.KmP / 4, acc1, Ed + (part ? rR : rI) * p.ldE, bs, a1.KmP / 4, acc2,
                      p.ldE, 1, 16, 1, lane);
            float *F = part ? FI : FR;
            const int n1 = nt1 * 16 + (lane & 15);
            if (n1 <= a1.J) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k2 = kt2 * 16 + (lane >> 4) * 4 + r;
                    if (k2 < a2.KmP) {
                        const float v = part ? acc1[r] + acc2[r] : acc1[r] - acc2[r];
                        const float w = part ? acc1[r] - acc2[r] : acc1[r] + acc2[r];
                        F[n1 * p.ldF + k2] = v;
                        if (n1 >= 1 && n1 <= a1.Js) F[(N1 - n1) * p.ldF + k2] = w;
                    }
                }
            }
        }
        __syncthreads();
        // ---- axis W: O[n1][n2] = sum_k2 FR cos - FI sin ; mirror n2 -> N2 - n2 gets +
        const int ntaskW = MT1 * a2.NT;
        for (int t = wave; t < ntaskW; t += (NTH / 64)) {
            const int nt2 = t % a2.NT, mt = t / a2.NT;
            f32x4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = acc1;
            if (!(a.dbg & 2))
                tile_mma2(FR + mt * 16 * p.ldF, cosW + nt2 * a2.KmP * 16, a2.KmP / 4, acc1, FI + mt * 16 * p.ldF,
                          sinW + nt2 * a2.KmP * 16, a2.KmP / 4, acc2, p.ldF, 1, 16, 1, lane);
            const int n2 = nt2 * 16 + (lane & 15);
            if (n2 <= a2.J) {
                const bool mirror = n2 >= 1 && n2 <= a2.Js;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n1 = mt * 16 + (lane >> 4) * 4 + r;
                    if (n1 < N1) {
                        O[n1 * p.ldo + n2] = acc1[r] - acc2[r];
                        if (mirror) O[n1 * p.ldo + (N2 - n2)] = acc1[r] + acc2[r];
                    }
                }
            }
        }
        __syncthreads();
        // ---- epilogue: out = act(scale * O + residual), flat and fully coalesced
        float *op = out + plane_base(a, plane, plane_elems);
            zero_volume_padding(a, out, plane, plane_elems, tid);
        if (!(a.dbg & 4)) {
            if (MAXE > 0) {
                int r = r0, c = c0;
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    const unsigned e = tid + (unsigned)NTH * j;
                    if (e < plane_elems) {
                        float v = a.scale * O[r * p.ldo + c];
                        if (ad) v += ra[j];
                        op[e] = act_apply(v, a.act);
                    }
                    r += dr;
                    c += dc;
                    if (c >= N2) {
                        c -= N2;
                        ++r;
                    }
                }
            } else {
                for (int r = wave; r < N1; r += (NTH / 64))
                    for (int c = lane; c < N2; c += 64) {
                        float v = a.scale * O[r * p.ldo + c];
                        if (ad) v += ad[(size_t)r * N2 + c];
                        op[(size_t)r * N2 + c] = act_apply(v, a.act);
                    }
            }
        }
    }
}

// ---- inverse plane kernel, specialised -------------------------------------------------------
// Compile-time k-step counts (KM1, KM2) and output tile counts (NT1, NT2), one k tile per axis.
// Tables in VGPRs, +-k1 fold fused into the operand reads of axis H (raw intermediate plane in LDS),
// all operand reads of a task issued before its MFMA chains, LDS-staged coalesced epilogue.
// NFULL = plane_elems / 256: epilogue elements j < NFULL exist for every thread (no guards, no exec juggling),
// j == NFULL is the ragged tail, j > NFULL does not exist.
template <int KM1, int KM2, int NT1, int NT2, bool HAS_ADD, int NFULL>
__global__ __launch_bounds__(256, 3) void dht_inv_plane_spec_kernel(const float *__restrict__ E,
                                                                    const float *__restrict__ addend,
                                                                    float *__restrict__ out, DhtArgs a) {
    extern __shared__ float lds[];
    constexpr int NE = NFULL + 1, NI = 4, NV4 = NFULL / 4;
    const DhtPlan &p = a.p;
    const Axis &a1 = p.ax[1], &a2 = p.ax[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l15 = lane & 15;
    const int N1 = a1.N, N2 = a2.N, m1 = a1.m;
    const int ldF = p.ldF;
    const int ldo = N2;   // flat image of the output plane: the epilogue reads O[tid + 256 j] with immediate offsets
                          // (C-layout stores then see 2-way bank conflicts on a few banks; they are 16 per task)
    const int ne = 2 * p.CP;
    float *Er = lds + 16;                      // 16 zero floats in front (fold reads at k1s = -1)
    float *FR = Er + ne + 16, *FI = FR + p.MP1 * ldF, *O = FI + p.MP1 * ldF;
    // tables in registers: B[k = ks*4 + q][n = 1 + nt*16 + l15].  Output position 0 (cos = 1, sin = 0) is
    // peeled off and computed as a plain sum, so positions 1..J fill whole 16-wide tiles (J = 32:
    // 2 tiles instead of 3).  Frequency 0 is its own mirror in the +-k1 fold of axis H: cos row halved.
    // axis-H tables: LDS image in operand order [nt][cos|sin][ks][lane] (used once per plane by each wave; keeping
    // them out of the register file avoids spills -- a spill reload waits on vmcnt(0), i.e. on the prefetches)
    float bwc[NT2][KM2], bws[NT2][KM2];
    float *tabH = O + 256 * NE;
    for (int i2 = tid; i2 < NT1 * 2 * KM1 * 64; i2 += 256) {
        const int ln = i2 & 63, ks = (i2 >> 6) % KM1, cs = ((i2 >> 6) / KM1) & 1, nt = (i2 >> 6) / (2 * KM1);
        const int n = 1 + nt * 16 + (ln & 15), kk = ks * 4 + (ln >> 4);
        const bool ok = n <= a1.J;
        const int idx = ((n >> 4) * a1.KmP + kk) * 16 + (n & 15);
        tabH[i2] = !ok ? 0.f : (cs ? p.tables[a1.sinI + idx] : p.tables[a1.cosI + idx] * (kk == 0 ? 0.5f : 1.f));
    }
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
        for (int ks = 0; ks < KM2; ++ks) {
            const int n = 1 + nt * 16 + l15, kk = ks * 4 + q;
            const bool ok = n <= a2.J;
            const int idx = ((n >> 4) * a2.KmP + kk) * 16 + (n & 15);
            bwc[nt][ks] = ok ? p.tables[a2.cosI + idx] : 0.f;
            bws[nt][ks] = ok ? p.tables[a2.sinI + idx] : 0.f;
        }
    // the tables must STAY in registers: left visible as loads from read-only memory, the compiler re-issues the
    // global loads inside the plane loop (rematerialisation), and every MFMA chain then waits on vmcnt
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
        for (int ks = 0; ks < KM2; ++ks) {
            asm volatile("" : "+v"(bwc[nt][ks]));
            asm volatile("" : "+v"(bws[nt][ks]));
        }
    if (tid < 16) {
        lds[tid] = 0.f;
        Er[ne + tid] = 0.f;
    }
    for (int i = tid; i < (p.MP1 - N1) * ldF; i += 256) {
        FR[N1 * ldF + i] = 0.f;
        FI[N1 * ldF + i] = 0.f;
    }
    const int planes = a.BC * p.ax[0].N;
    const size_t plane_elems = (size_t)N1 * N2;
    const int MT1 = p.MP1 / 16;
    float re[NI], ra[HAS_ADD ? NE : 1];
    auto fetch_e = [&](int plane) {
        const float *Ep = E + (size_t)plane * ne;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int e = tid + 256 * j;
            if (a.zl) {      // the LDS image keeps the [part][k1 position][16] order (this kernel needs one k2 tile: KP == 16)
                const int es = e < ne ? e : 0, part = es / p.CP, rem = es - part * p.CP;
                re[j] = e < ne ? E[ymid(a, plane, part, rem >> 4, rem & 15)] : 0.f;
            } else
                re[j] = e < ne ? Ep[e] : 0.f;
        }
    };
    HNO_STAMP(a.stamps, 0);
    if (a.stamps && blockIdx.x == 0 && tid == 0) { a.stamps[60] = wall_clock64(); a.stamps[62] = clock64(); }
    if (blockIdx.x < planes) fetch_e(blockIdx.x);
    int it = 0;
    for (int plane = blockIdx.x; plane < planes; plane += gridDim.x, ++it) {
        HNO_STAMP(a.stamps, 1 + it * 6);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int e = tid + 256 * j;
            if (e < ne) Er[e] = re[j];
        }
        if (plane + gridDim.x < planes) fetch_e(plane + gridDim.x);
        if (HAS_ADD) {  // residual of THIS plane: consumed in the epilogue, after both MFMA stages
            const float *ad = addend + plane_base(a, plane, plane_elems);
#pragma unroll
            // elements [0, 1024 NV4) as 16-byte vectors (4 tid + 1024 jv + 0..3), the rest as scalars (tid + 256 j)
            for (int jv = 0; jv < NV4; ++jv) {
                const f4u_t v = *reinterpret_cast<const f4u_t *>(ad + 4 * tid + 1024 * jv);
                ra[HAS_ADD ? 4 * jv : 0] = v.x;
                ra[HAS_ADD ? 4 * jv + 1 : 0] = v.y;
                ra[HAS_ADD ? 4 * jv + 2 : 0] = v.z;
                ra[HAS_ADD ? 4 * jv + 3 : 0] = v.w;
            }
#pragma unroll
            for (int j = 4 * NV4; j < NE; ++j) {
                const unsigned e = tid + 256u * j;
                ra[HAS_ADD ? j : 0] = (j < NFULL || e < plane_elems) ? ad[e] : 0.f;
            }
        }
        HNO_STAMP(a.stamps, 2 + it * 6);
        __syncthreads();
        HNO_STAMP(a.stamps, 3 + it * 6);
        // ---- axis H: F[n1][k2] = sum_k1 E[k1][k2] e^{+i th k1 n1}; E[+k1] +- E[-k1] formed while reading
        if (!(a.dbg & 1)) {
            for (int t = wave; t < NT1 * 2; t += 4) {
                const int part = t & 1, nt1 = t >> 1;
                // FR = cos * Es_R - sin * Ed_I ; FI = cos * Es_I + sin * Ed_R
                const float *es = Er + (part ? p.CP : 0) + l15, *ed = Er + (part ? 0 : p.CP) + l15;
                const float *sf = es + (m1 + q) * 16, *sb = es + (m1 - q) * 16;
                const float *df = ed + (m1 + q) * 16, *db = ed + (m1 - q) * 16;
                float as_[KM1], ad_[KM1], asb[KM1], adb[KM1], tc[KM1], ts[KM1];
                const float *th = tabH + nt1 * 2 * KM1 * 64 + lane;
#pragma unroll
                for (int ks = 0; ks < KM1; ++ks) {
                    as_[ks] = sf[64 * ks];
                    asb[ks] = sb[-64 * ks];
                    ad_[ks] = df[64 * ks];
                    adb[ks] = db[-64 * ks];
                    tc[ks] = th[64 * ks];
                    ts[ks] = th[64 * (KM1 + ks)];
                }
                __builtin_amdgcn_sched_barrier(0);   // all raw reads in flight before the first fold
#pragma unroll
                for (int ks = 0; ks < KM1; ++ks) {
                    as_[ks] += asb[ks];
                    ad_[ks] -= adb[ks];
                }
                __builtin_amdgcn_sched_barrier(0);   // all operand reads before the MFMA chains
                f32x4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = acc1;
#pragma unroll
                for (int ks = 0; ks < KM1; ++ks) {
                    acc1 = mfma16(as_[ks], tc[ks], acc1);
                    acc2 = mfma16(ad_[ks], ts[ks], acc2);
                }
                float *F = part ? FI : FR;
                const int n1 = 1 + nt1 * 16 + l15;
                if (n1 <= a1.J) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k2 = q * 4 + r;
                        const float v = part ? acc1[r] + acc2[r] : acc1[r] - acc2[r];
                        const float w = part ? acc1[r] - acc2[r] : acc1[r] + acc2[r];
                        F[n1 * ldF + k2] = v;
                        if (n1 <= a1.Js) F[(N1 - n1) * ldF + k2] = w;
                    }
                }
            }
            // output row n1 = 0: every e^{i th k1 0} = 1, so F[0][k2] is the plain sum over all 2*m1+1 k1
            if (wave == 3) {   // lanes 0..31: (part, k2) with k1s = 0, 4, 8, .. and 2, 6, ..; lanes 32..63: the odd k1s
                const int o5 = lane & 31, g = lane >> 5;
                const float *src = Er + (o5 >> 4) * p.CP + l15;
                float s0 = 0.f, s1 = 0.f;
                for (int k1s = g; k1s < p.K1S; k1s += 4) {
                    s0 += src[k1s * 16];
                    s1 += k1s + 2 < p.K1S ? src[(k1s + 2) * 16] : 0.f;
                }
                s0 += s1;
                s0 += __shfl_xor(s0, 32);
                if (g == 0) ((o5 >> 4) ? FI : FR)[l15] = s0;
            }
        }
        HNO_STAMP(a.stamps, 4 + it * 6);
        __syncthreads();
        // ---- axis W: O[n1][n2] = sum_k2 FR cos - FI sin ; mirror n2 -> N2 - n2 gets +
        // tasks = (16-row tile mt, 16-column tile nt2); a wave takes two tasks per round and issues the operand
        // reads of both before the first MFMA.  One or two leftover rows (N1 = 16 t + 1) go to the VALU instead
        // of a whole MFMA tile.
        const int MTF = (N1 % 16 >= 1 && N1 % 16 <= 2) ? N1 / 16 : MT1;
        for (int t0 = wave; t0 < MTF * NT2; t0 += 8) {
            const int t1 = t0 + 4;
            const bool two = t1 < MTF * NT2;
            const int nt2a = t0 % NT2, mta = t0 / NT2, nt2b = two ? t1 % NT2 : nt2a, mtb = two ? t1 / NT2 : mta;
            const float *fra = FR + (mta * 16 + l15) * ldF + q, *fia = FI + (mta * 16 + l15) * ldF + q;
            const float *frb = FR + (mtb * 16 + l15) * ldF + q, *fib = FI + (mtb * 16 + l15) * ldF + q;
            float ar[KM2], ai[KM2], br[KM2], bi[KM2];
#pragma unroll
            for (int ks = 0; ks < KM2; ++ks) {
                ar[ks] = fra[4 * ks];
                ai[ks] = fia[4 * ks];
                br[ks] = frb[4 * ks];
                bi[ks] = fib[4 * ks];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (it == 1) HNO_STAMP(a.stamps, 40);
            f32x4 a1_ = {0.f, 0.f, 0.f, 0.f}, a2_ = a1_, b1_ = a1_, b2_ = a1_;
            if (!(a.dbg & 2)) {
#pragma unroll
                for (int n = 0; n < NT2; ++n) {
                    if (n == nt2a) {
#pragma unroll
                        for (int ks = 0; ks < KM2; ++ks) {
                            a1_ = mfma16(ar[ks], bwc[n][ks], a1_);
                            a2_ = mfma16(ai[ks], bws[n][ks], a2_);
                        }
                    }
                    if (two && n == nt2b) {
#pragma unroll
                        for (int ks = 0; ks < KM2; ++ks) {
                            b1_ = mfma16(br[ks], bwc[n][ks], b1_);
                            b2_ = mfma16(bi[ks], bws[n][ks], b2_);
                        }
                    }
                }
            }
            if (it == 1) { asm volatile("s_nop 0" :: "v"(a1_[0]), "v"(a2_[0]), "v"(b1_[0]), "v"(b2_[0])); HNO_STAMP(a.stamps, 41); }
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                if (h2 == 1 && !two) break;
                const int nt2 = h2 ? nt2b : nt2a, mt = h2 ? mtb : mta;
                const f32x4 acc1 = h2 ? b1_ : a1_, acc2 = h2 ? b2_ : a2_;
                const int n2 = 1 + nt2 * 16 + l15;
                if (nt2 * 16 + 16 <= a2.Js && mt * 16 + 16 <= N1) {   // whole tile inside the plane (wave-uniform): no lane masks
                    float *o = O + (mt * 16 + q * 4) * ldo + n2, *om = O + (mt * 16 + q * 4) * ldo + (N2 - n2);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        o[r * ldo] = acc1[r] - acc2[r];
                        om[r * ldo] = acc1[r] + acc2[r];
                    }
                } else if (n2 <= a2.J) {
                    const bool mirror = n2 <= a2.Js;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n1 = mt * 16 + q * 4 + r;
                        if (n1 < N1) {
                            O[n1 * ldo + n2] = acc1[r] - acc2[r];
                            if (mirror) O[n1 * ldo + (N2 - n2)] = acc1[r] + acc2[r];
                        }
                    }
                }
            }
        }
        if (it == 1) HNO_STAMP(a.stamps, 42);
        if (wave == 2 && !(a.dbg & 2)) {
            for (int n1 = MTF * 16; n1 < N1; ++n1) {   // leftover rows: the k2 sum is split over the 4 lane groups
                const float *fr = FR + n1 * ldF + q, *fi = FI + n1 * ldF + q;
#pragma unroll
                for (int n = 0; n < NT2; ++n) {
                    float pc = 0.f, ps = 0.f;
#pragma unroll
                    for (int ks = 0; ks < KM2; ++ks) {
                        pc += fr[4 * ks] * bwc[n][ks];
                        ps += fi[4 * ks] * bws[n][ks];
                    }
                    pc += __shfl_xor(pc, 16);
                    ps += __shfl_xor(ps, 16);
                    pc += __shfl_xor(pc, 32);
                    ps += __shfl_xor(ps, 32);
                    const int n2 = 1 + n * 16 + l15;
                    if (q == 0 && n2 <= a2.J) {
                        O[n1 * ldo + n2] = pc - ps;
                        if (n2 <= a2.Js) O[n1 * ldo + (N2 - n2)] = pc + ps;
                    }
                }
            }
        }
        // output column n2 = 0: cos = 1, sin = 0 -> plain sum of FR over k2 (one lane per row)
        if (wave == 3) {
            for (int n1 = lane; n1 < N1; n1 += 64) {
                float sum = 0.f;
#pragma unroll
                for (int k2 = 0; k2 < 4 * KM2; ++k2) sum += FR[n1 * ldF + k2];
                O[n1 * ldo] = sum;
            }
        }
        HNO_STAMP(a.stamps, 5 + it * 6);
        __syncthreads();
        HNO_STAMP(a.stamps, 6 + it * 6);
        // ---- epilogue: out = act(scale * O + residual), flat and fully coalesced
        if (!(a.dbg & 4)) {
            const float ap = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE : 1.f;
            const float aq = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE * HNO_SELU_ALPHA : 1.f;
            const bool lin = a.act == HNO_ACT_NONE;
            float *op = out + plane_base(a, plane, plane_elems);
            zero_volume_padding(a, out, plane, plane_elems, tid);
            float ov[NE];   // all LDS reads first (one wait), then the arithmetic and the stores
            typedef float f4v __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int jv = 0; jv < NV4; ++jv) {
                const f4v t = *reinterpret_cast<const f4v *>(O + 4 * tid + 1024 * jv);
                ov[4 * jv] = t.x;
                ov[4 * jv + 1] = t.y;
                ov[4 * jv + 2] = t.z;
                ov[4 * jv + 3] = t.w;
            }
#pragma unroll
            for (int j = 4 * NV4; j < NE; ++j) ov[j] = O[tid + 256 * j];   // the O region is padded to 256 * NE floats
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                float v = a.scale * ov[j];
                if (HAS_ADD) v += ra[HAS_ADD ? j : 0];
                ov[j] = (v > 0.f || lin) ? ap * v : aq * neg_expm1(v);
            }
            // 16-byte stores issue 4x fewer vector-memory instructions (the address path, not HBM, was the limit)
#pragma unroll
            for (int jv = 0; jv < NV4; ++jv) {
                f4u_t t;
                t.x = ov[4 * jv];
                t.y = ov[4 * jv + 1];
                t.z = ov[4 * jv + 2];
                t.w = ov[4 * jv + 3];
                *reinterpret_cast<f4u_t *>(op + 4 * tid + 1024 * jv) = t;
            }
#pragma unroll
            for (int j = 4 * NV4; j < NE; ++j)
                if (j < NFULL || tid + 256u * j < plane_elems) op[tid + 256 * j] = ov[j];
        }
    }
    if (a.stamps && blockIdx.x == 0 && tid == 0) { a.stamps[61] = wall_clock64(); a.stamps[63] = clock64(); }
}

// ---- inverse plane kernel, one wave per ITEM (half plane), no workgroup barrier, both GEMMs chained in registers ---------------
// What bounded the kernel above (profiles/r02_d_pmc_sq_per_kernel.json): three workgroup barriers per plane, the axis-H result through
// LDS, 3 300 instructions per wave of which ~1 000 are the SELU epilogue, and fp32 MFMAs that do not overlap VALU work of the same wave.
// Here an item = the output rows of one row-pair tile: rows n1 = 1 + 16 X + i and their mirrors N1 - n1 (item 0 also row 0), i.e. rows
// [0, 17) + [49, 65) and rows [17, 49) of a 65-row plane -- items are independent, so 6 240 of them spread evenly over all SIMDs:
//   * axis H reads its operands (the +-k1 rows of the 3.7 KB intermediate plane) straight from global memory / L2, 16 dwords per lane;
//   * its accumulators (column n1 on the lane, rows k2 = 4 q + r in register r) ARE the A operands of the axis-W product when the
//     tables are stored with k-step r <-> k2 = 4 q + r (DhtPlan::itab): no LDS between the two GEMMs;
//   * the axis-W result goes to a wave-private LDS image of the item's rows, laid out at the 16-byte phase of the global destination,
//     so the epilogue (scale, residual, activation) reads ds_read_b128 and writes aligned 16-byte rows; only the first and last
//     group of a chunk are written element-wise;
//   * twelve waves per CU (three per SIMD): while one wave's MFMAs hold the matrix pipe the others run their epilogues.
// Requires N1 = 32 NP + 1 (NP = 1, 2), N2 = 16 NT2 * 2 + 1, one k tile per axis, out (and addend) 4-byte aligned with equal 16-byte phase.
template <int NP, int N2c, int KM1, int NT2, bool HAS_ADD, int NWV, bool ZL>   // ZL: layout of the intermediate (DhtArgs.zl) at compile time:
// as a run-time branch per operand load it grew the kernel from 5.6 to 9.0 KB and put a wait behind every load of the row-layout path
__global__ __launch_bounds__(64 * NWV, NWV / 4) void dht_inv_item_kernel(const float *__restrict__ E, const float *__restrict__ add_al,
                                                              float *__restrict__ out_al, DhtArgs a, unsigned shift0, int it_base,
                                                              int it_rem, unsigned ldbc) {
    extern __shared__ float lds[];
    HNO_STAMP(a.stamps, 20);
    if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[60] = wall_clock64();
    if (a.stamps && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) a.stamps[58] = wall_clock64();
    if (a.stamps && blockIdx.x == gridDim.x / 2 && threadIdx.x == 64 * (NWV - 1)) a.stamps[56] = wall_clock64();
    constexpr int N1 = 32 * NP + 1, N2 = N2c;
    constexpr int OBUF = NP == 2 ? 2176 : round_up_c(N1 * N2 + 4, 64);   // floats per wave: the item's rows (+ phase slack)
    constexpr int WSTRIDE = OBUF + 64;                                    // + scratch for row 0
    const DhtPlan &p = a.p;
    const Axis &a1 = p.ax[1];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, l15 = lane & 15;
    const int m1 = a1.m;
    float *obuf = lds + (size_t)wave * WSTRIDE, *scr = obuf + OBUF;
    const int bid = blockIdx.x;
    const int t_begin = bid * it_base + (bid < it_rem ? bid : it_rem);
    const int t_end = t_begin + it_base + (bid < it_rem ? 1 : 0);
    // the items of a wave all have the same X (NWV is even): X = parity of its first item
    int t = t_begin + wave;
    const int X = NP == 2 ? (t & 1) : 0;
    // ---- tables (lane order, copy b % kDmaTabCopies): axis H of tile X, axis W of both column tiles
    float thc[KM1], ths[KM1], bwc[NT2][4], bws[NT2][4];
    {
        const float *tb = p.tables + p.itab + (size_t)(bid % kDmaTabCopies) * p.itab_stride + lane;
#pragma unroll
        for (int ks = 0; ks < KM1; ++ks) {
            thc[ks] = tb[((X * 2 + 0) * KM1 + ks) * 64];
            ths[ks] = tb[((X * 2 + 1) * KM1 + ks) * 64];
        }
        const float *tw = tb + NP * 2 * KM1 * 64;
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bwc[nt][r] = tw[((nt * 2 + 0) * 4 + r) * 64];
                bws[nt][r] = tw[((nt * 2 + 1) * 4 + r) * 64];
            }
    }
#pragma unroll
    for (int ks = 0; ks < KM1; ++ks) {
        asm volatile("" : "+v"(thc[ks]));
        asm volatile("" : "+v"(ths[ks]));
    }
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            asm volatile("" : "+v"(bwc[nt][r]));
            asm volatile("" : "+v"(bws[nt][r]));
        }
    const unsigned pe = N1 * N2;
    const int ne = 2 * p.CP;                                   // floats per intermediate plane
    const float ap = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE : 1.f;
    const float aq = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE * HNO_SELU_ALPHA : 1.f;
    const bool lin = a.act == HNO_ACT_NONE;
    const bool q0 = q == 0;
    // chunk geometry of this wave's items: rows [R0, R0 + NR) at LDS float offset LO (a second chunk for X = 0 of a 65-row plane)
    // X = 0: rows [0, 17) (+ [49, 65));  X = 1: rows [17, 49);  NP = 1: rows [0, 33)
    // axis-H operands of an item: rows k1s = m1 +- (4 ks + q) of the re / im parts (out-of-range rows are selected away: the last
    // k-step may reach |k1| = m1 + 1, whose table entries are zero but whose bytes are arbitrary).  Four waves per SIMD cover the
    // L2 round trip; prefetching the next item's operands was measured and changed nothing at three waves per SIMD.
    float erp[KM1], erm[KM1], eip[KM1], eim[KM1];
    auto load_e = [&](int tt) {
        const int pl = NP == 2 ? (tt >> 1) : tt;
        const float *Ep = E + (size_t)pl * ne;
#pragma unroll
        for (int ks = 0; ks < KM1; ++ks) {
            const int k1 = 4 * ks + q;
            const bool ok = k1 <= m1;
            const int rp = ok ? m1 + k1 : m1, rm = ok ? m1 - k1 : m1;
            if constexpr (ZL) {
                const size_t op = ((size_t)(rp * 4 + (l15 >> 2)) * a.zplanes + pl) * 8 + (l15 & 3);
                const size_t om = ((size_t)(rm * 4 + (l15 >> 2)) * a.zplanes + pl) * 8 + (l15 & 3);
                erp[ks] = E[op];
                erm[ks] = E[om];
                eip[ks] = E[op + 4];
                eim[ks] = E[om + 4];
            } else {
                erp[ks] = Ep[rp * 16 + l15];
                erm[ks] = Ep[rm * 16 + l15];
                eip[ks] = Ep[p.CP + rp * 16 + l15];
                eim[ks] = Ep[p.CP + rm * 16 + l15];
            }
        }
    };
    HNO_STAMP(a.stamps, 21);
    // (Staggering the four waves of a SIMD by n x 1 024 cycles -- one computes while the others store -- was measured: 34.7 us at
    // n = 0, 35.3 / 36.1 / 38.2 at n = 1 / 3 / 8 without residual; with residual 38.5 -> 36.9 at n = 3.  Not kept.)
    int it = 0;
    // the next item's operands are requested as soon as this item's are folded (round 5: -0.7 us per launch of the residual variant on
    // boxes where the operand round trip is long)
    constexpr bool pf = true;
    if (pf && t < t_end) load_e(t);
    for (; t < t_end; t += NWV, ++it) {
        HNO_STAMP(a.stamps, 24 + it * 6);
        const int plane = NP == 2 ? (t >> 1) : t;
        // ---- chunk geometry of the item (rows [r0, r0 + nr) of the plane, LDS float offset): X = 0: rows [0, 17) (+ [49, 65) at 1112);
        //      X = 1: rows [17, 49);  NP = 1: rows [0, 33)
        // float index of the plane from the aligned base: consecutive (b, c) volumes are ldbc floats apart (channel-padded
        // activations: N0 pe rounded up to a multiple of 32; the wave that writes a volume's last rows zeroes the padding)
        const int N0p = p.ax[0].N, bcv = plane / N0p, n0v = plane - bcv * N0p;
        const unsigned f0 = shift0 + (unsigned)bcv * ldbc + (unsigned)n0v * pe;
        if (n0v == N0p - 1 && X == 0 && ldbc > (unsigned)N0p * pe) {
            const unsigned npad = ldbc - (unsigned)N0p * pe;
            if ((unsigned)lane < npad) out_al[shift0 + (unsigned)bcv * ldbc + (unsigned)N0p * pe + lane] = 0.f;
        }
        constexpr int NG0 = NP == 2 ? 5 : (N1 * N2 + 3 + 255) / 256, NG1 = NP == 2 ? 5 : 0, NGB = 9;
        const int c0_r0 = NP == 2 ? (X == 0 ? 0 : 17) : 0;
        const int c1_r0 = 49, c1_lo = 1112;
        const unsigned fA = f0 + (unsigned)(c0_r0 * N2), fC = f0 + 49u * N2;
        const unsigned shA = fA & 3u, shC = fC & 3u;
        if (!pf) load_e(t);
        // ---- folds of the +-k1 rows
        float sR[KM1], dR[KM1], sI[KM1], dI[KM1];
#pragma unroll
        for (int ks = 0; ks < KM1; ++ks) {
            const bool ok = 4 * ks + q <= m1;
            sR[ks] = ok ? erp[ks] + erm[ks] : 0.f;
            dR[ks] = ok ? erp[ks] - erm[ks] : 0.f;
            sI[ks] = ok ? eip[ks] + eim[ks] : 0.f;
            dI[ks] = ok ? eip[ks] - eim[ks] : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (pf && t + NWV < t_end) load_e(t + NWV);
        __builtin_amdgcn_sched_barrier(0);
        if (a.stamps) asm volatile("s_nop 0" ::"v"(sR[0]), "v"(dI[KM1 - 1]));
        if (HAS_ADD) {
            // The residual rows go by LDS-DMA straight into the item's image -- the same aligned 16-byte groups the epilogue reads --
            // at the TOP of the item, ahead of the operand loads: one HBM round trip per item, overlapped with the operands' L2 round
            // trip and covered by the other waves of the SIMD.  The GEMM results are then added into the image (read - add - write;
            // ds_add_f32 was measured at ~0.4 lanes per cycle and CU: 93 us per launch) and the epilogue is the one without a
            // residual.  Loading the residual in the epilogue (two groups ahead) paid the HBM latency five times per item: 37 us
            // per launch inside a training step against 23.5 us without residual (in isolation the 53 MB residual sat in the MALL).
            // Lanes beyond the chunk are masked: they would overwrite the neighbouring chunk's image.
            const unsigned ob = (unsigned)(size_t)obuf;
            const int nch = (NP == 2 && X == 0) ? 2 : 1;
#pragma unroll 1
            for (int ch = 0; ch < nch; ++ch) {
                const unsigned fch = ch ? fC : fA;
                const unsigned n = NP == 1 ? (unsigned)(N1 * N2) : (X == 0 ? (ch ? 16u : 17u) : 32u) * N2;
                const int ng = NP == 1 ? NG0 : (X == 0 ? NG0 : NGB);
                const unsigned lim = (fch & 3u) + n, lo = ch ? (unsigned)c1_lo : 0u;
                const unsigned goff = ((fch & ~3u) + 4u * lane) * 4u;
#pragma unroll 1
                for (int j = 0; j < ng; ++j)
                    if (4u * (64u * j + lane) < lim)
                        dma_piece16(add_al, goff + 1024u * j, __builtin_amdgcn_readfirstlane(ob + (lo + 256u * j) * 4u));
            }
        }
        HNO_STAMP(a.stamps, 25 + it * 6);
        // ---- axis H, tile X: F[n1][k2] = sum_k1 E[k1][k2] e^{+i th k1 n1}: cR, cI = cosine sums of re / im, sR_, sI_ = sine sums
        f32x4 cRe = {0.f, 0.f, 0.f, 0.f}, cIm = cRe, sRe = cRe, sIm = cRe;
        if (!(a.dbg & 8))   // timing aid: no MFMA
#pragma unroll
        for (int ks = 0; ks < KM1; ++ks) {
            cRe = mfma16(sR[ks], thc[ks], cRe);
            sIm = mfma16(dI[ks], ths[ks], sIm);
            cIm = mfma16(sI[ks], thc[ks], cIm);
            sRe = mfma16(dR[ks], ths[ks], sRe);
        }
        // row n1: FR = cRe - sIm, FI = cIm + sRe;  mirror row N1 - n1: FR = cRe + sIm, FI = cIm - sRe
        float FRp[4], FIp[4], FRm[4], FIm[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            FRp[r] = cRe[r] - sIm[r];
            FRm[r] = cRe[r] + sIm[r];
            FIp[r] = cIm[r] + sRe[r];
            FIm[r] = cIm[r] - sRe[r];
        }
        if (a.stamps) asm volatile("s_nop 0" ::"v"(FRp[0]), "v"(FIm[3]));
        HNO_STAMP(a.stamps, 26 + it * 6);
        // ---- LDS image of the item's rows: plane row n1 of chunk 0 at img0 + n1 N2, of chunk 1 at img1 + n1 N2
        float *img0 = obuf + shA - c0_r0 * N2, *img1 = obuf + c1_lo + shC - c1_r0 * N2;
        const int n1p = 1 + 16 * X + 4 * q, n1m = N1 - n1p;     // rows of accumulator register 0 (plus tile ascending, mirror descending)
        float *rowP = img0 + n1p * N2, *rowM = ((NP == 2 && X == 0) ? img1 : img0) + n1m * N2;
        // image element <- v (no residual) or += scale v on top of the residual the DMA put there
        auto put = [&](float *dst, float v) {
            if (HAS_ADD) *dst = fmaf(v, a.scale, *dst);
            else *dst = v;
        };
        // ---- axis W: O[n1][n2] = sum_k2 FR cos - FI sin, mirror column N2 - n2 gets +; the A operands are the F registers
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) {
            f32x4 pc = {0.f, 0.f, 0.f, 0.f}, ps = pc, mc = pc, ms = pc;
            if (!(a.dbg & 8))
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pc = mfma16(FRp[r], bwc[nt][r], pc);
                ps = mfma16(FIp[r], bws[nt][r], ps);
                mc = mfma16(FRm[r], bwc[nt][r], mc);
                ms = mfma16(FIm[r], bws[nt][r], ms);
            }
            const int n2 = 1 + 16 * nt + l15;
            if (HAS_ADD) {
                if (nt == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the residual is in the image (long since: in-order
                                                                                // return behind the operand loads)
#pragma unroll
                for (int r = 0; r < 4; ++r) {   // four at a time: sixteen residual values in flight spilled registers
                    const float v0 = rowP[r * N2 + n2], v1 = rowP[r * N2 + N2 - n2], v2 = rowM[-r * N2 + n2], v3 = rowM[-r * N2 + N2 - n2];
                    rowP[r * N2 + n2] = fmaf(pc[r] - ps[r], a.scale, v0);
                    rowP[r * N2 + N2 - n2] = fmaf(pc[r] + ps[r], a.scale, v1);
                    rowM[-r * N2 + n2] = fmaf(mc[r] - ms[r], a.scale, v2);
                    rowM[-r * N2 + N2 - n2] = fmaf(mc[r] + ms[r], a.scale, v3);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    rowP[r * N2 + n2] = pc[r] - ps[r];
                    rowP[r * N2 + N2 - n2] = pc[r] + ps[r];
                    rowM[-r * N2 + n2] = mc[r] - ms[r];
                    rowM[-r * N2 + N2 - n2] = mc[r] + ms[r];
                }
            }
        }
        HNO_STAMP(a.stamps, 27 + it * 6);
        // ---- output column n2 = 0: cos = 1, sin = 0 -> the plain sum of FR over k2 = over (q, r): row i = l15 of each tile
        {
            float zp = (FRp[0] + FRp[1]) + (FRp[2] + FRp[3]), zm = (FRm[0] + FRm[1]) + (FRm[2] + FRm[3]);
            zp += __shfl_xor(zp, 16);
            zm += __shfl_xor(zm, 16);
            zp += __shfl_xor(zp, 32);
            zm += __shfl_xor(zm, 32);
            // ... but the lane layout of F is (column n1 index = l15, k2 group q): the sum over q is the row's value
            if (q0) {
                put(img0 + (1 + 16 * X + l15) * N2, zp);
                put(((NP == 2 && X == 0) ? img1 : img0) + (N1 - 1 - 16 * X - l15) * N2, zm);
            }
        }
        // ---- output row 0 (item 0): F[0][k2] = plain sum over all k1 (the k1 = 0 row was doubled by the fold)
        if (X == 0) {
            float f0r = q0 ? 0.5f * sR[0] : sR[0], f0i = q0 ? 0.5f * sI[0] : sI[0];
#pragma unroll
            for (int ks = 1; ks < KM1; ++ks) {
                f0r += sR[ks];
                f0i += sI[ks];
            }
            f0r += __shfl_xor(f0r, 16);
            f0i += __shfl_xor(f0i, 16);
            f0r += __shfl_xor(f0r, 32);
            f0i += __shfl_xor(f0i, 32);
            // every lane group holds F0[k2 = l15]; the axis-W tables want k2 = 4 q + r: through the wave's scratch
            if (q0) {
                scr[l15] = f0r;
                scr[16 + l15] = f0i;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const f32x4 gr = *reinterpret_cast<const f32x4 *>(scr + 4 * q), gi = *reinterpret_cast<const f32x4 *>(scr + 16 + 4 * q);
            float z0 = (gr[0] + gr[1]) + (gr[2] + gr[3]);
            z0 += __shfl_xor(z0, 16);
            z0 += __shfl_xor(z0, 32);
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) {
                float c = 0.f, s_ = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    c = fmaf(gr[r], bwc[nt][r], c);
                    s_ = fmaf(gi[r], bws[nt][r], s_);
                }
                c += __shfl_xor(c, 16);
                s_ += __shfl_xor(s_, 16);
                c += __shfl_xor(c, 32);
                s_ += __shfl_xor(s_, 32);
                const int n2 = 1 + 16 * nt + l15;
                if (q0) {
                    put(img0 + n2, c - s_);
                    put(img0 + N2 - n2, c + s_);
                }
            }
            if (lane == 0) put(img0, z0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        HNO_STAMP(a.stamps, 28 + it * 6);
        // ---- epilogue: out = act(scale * O + residual) in aligned 16-byte groups; group g of a chunk covers chunk elements
        //      4 g - sh .. 4 g - sh + 3 (element e at LDS float lo + sh + e and at global float (f & ~3) + sh + e)
        // A LOOP, not an unrolled sequence: every kernel starts with a cold instruction cache and a wave runs this code once or twice
        // per launch, so straight-line code is paid for in instruction fetches (~0.4 us per KB, DESIGN lesson 1); unrolled, the
        // epilogue of the two item kinds was 20 KB.
        const int nchunk = (a.dbg & 4) ? 0 : ((NP == 2 && X == 0) ? 2 : 1);
#pragma unroll 1
        for (int ch = 0; ch < nchunk; ++ch) {
            const unsigned fch = ch ? fC : fA;
            const unsigned n = NP == 1 ? (unsigned)(N1 * N2) : (X == 0 ? (ch ? 16u : 17u) : 32u) * N2;
            const int ng = NP == 1 ? NG0 : (X == 0 ? NG0 : NGB);
            const float *limg = obuf + (ch ? c1_lo : 0);
            const unsigned sh = fch & 3u;
            float *gbase = out_al + (fch & ~3u) + 4 * lane;
            // (with a residual the image already holds scale * O + residual: see the DMA above)
            const f32x2 sc = {HAS_ADD ? 1.f : a.scale, HAS_ADD ? 1.f : a.scale};
            // the image read of group j + 1 is issued before group j is worked on (reads past the last group stay inside the wave's own
            // LDS area or the next wave's, values unused)
            f32x4 onext = *reinterpret_cast<const f32x4 *>(limg + 4 * lane);
#pragma unroll 1
            for (int j = 0; j < ng; ++j) {
                const int e0 = (int)(4 * (64u * j + lane)) - (int)sh;
                const f32x4 o = onext;
                if (j + 1 < ng) onext = *reinterpret_cast<const f32x4 *>(limg + 256 * (j + 1) + 4 * lane);
                // scale and activation on the packed fp32 pipe, two elements per instruction
                f32x2 x0 = f32x2{o[0], o[1]} * sc;
                f32x2 x1 = f32x2{o[2], o[3]} * sc;
                if (!lin) {   // wave-uniform
                    x0 = selu_like_pk(x0, ap, aq);
                    x1 = selu_like_pk(x1, ap, aq);
                }
                const f32x4 v = {x0[0], x0[1], x1[0], x1[1]};
                if (a.dbg & 2) {   // timing aid: no stores
                    asm volatile("" ::"v"(v));
                } else if (e0 >= 0 && e0 + 3 < (int)n) {
                    *reinterpret_cast<f32x4 *>(gbase + 256 * j) = v;          // wholly inside the chunk (write-through `sc1` stores, which
                                                                              // pay in the forward kernel, cost 5 us here: 40.5 vs 35.2 us)
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (e0 + c >= 0 && e0 + c < (int)n) gbase[256 * j + c] = v[c];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        HNO_STAMP(a.stamps, 29 + it * 6);
    }
    HNO_STAMP(a.stamps, 23);
    if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[61] = wall_clock64();
    if (a.stamps && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) a.stamps[59] = wall_clock64();
    if (a.stamps && blockIdx.x == gridDim.x / 2 && threadIdx.x == 64 * (NWV - 1)) a.stamps[57] = wall_clock64();
}

// full: bit i set = axis i keeps all N_i = 2 m_i + 1 frequencies (see DhtArgs).  A degenerate first axis
// (N0 = 1, m0 = 0: 2-D data) is always "full".
static int check_sizes(int BC, int N0, int N1, int N2, int m0, int m1, int m2, int &full, int mode) {
    HNO_REQUIRE(BC > 0 && N0 > 0 && N1 > 0 && N2 > 0, "dht3: non-positive size");
    if (m0 > 31) return fail(HNO_ELIMIT, "dht3: m0 = %d modes along the first axis (max 31)", m0);
    if (N0 == 1 && m0 == 0) full |= 1;
    HNO_REQUIRE((m0 > 0 || (full & 1)) && m1 > 0 && m2 > 0, "dht3: modes must be positive");
    const int N[3] = {N0, N1, N2}, m[3] = {m0, m1, m2};
    for (int i = 0; i < 3; ++i) {
        if ((full >> i) & 1)
            HNO_REQUIRE(2 * m[i] + 1 == N[i], "dht3: a full axis needs N = 2 m + 1");
        else
            HNO_REQUIRE(2 * m[i] <= N[i], "dht3: modes must be clamped to N // 2 by the caller");
    }
    HNO_REQUIRE(mode == 0 || !(full & 6), "dht3: full axes 1 / 2 exist in the Hartley layout only");
    if (BC > 65535) return fail(HNO_ELIMIT, "dht3: B*C = %d exceeds 65535", BC);
    return HNO_OK;
}

static const size_t kMaxLds = 160 * 1024;

// Which plane-kernel family the last forward / inverse launch of this thread took: 0 none yet, 1 generic (workgroup per plane), 2 the
// round-2 specialised kernels (spec / wave), 3 the LDS-DMA forward / half-plane item inverse kernels.  Test aid (tests/test_hip_ops.py
// pins the families of the headline shapes: a refactoring that drops an instantiation falls back to a slower family with every parity
// test still green -- round 4 lost 9 % of the headline that way for an hour).
static thread_local int g_last_plane_family[2] = {0, 0};
extern "C" int hno_debug_last_plane_family(int inverse) { return g_last_plane_family[inverse ? 1 : 0]; }


// Waves per plane of the generic plane kernels on planes above 5 120 elements (121 x 121 at the published inference size).  Those
// kernels are bound by instruction issue and latency, not by LDS capacity (two planes per CU either way): measured single-image
// inference at 240 x 240 x 155, GPU forward 4.26 ms with 4 waves per plane, 3.14 with 8, 2.82 with 16 (default).
// HNO_GENERIC_WAVES=4 / 8 select the others (A/B).
static int generic_waves() {
    static const int v = getenv("HNO_GENERIC_WAVES") ? atoi(getenv("HNO_GENERIC_WAVES")) : 16;
    return v;
}
// planes of up to 5 120 elements outside the 65 / 61 / 33 specialisations: eight waves per plane above 2 800 elements (measured on the
// HNOSeg-XS step: 57 x 57 planes (112^3 inputs) 3.10 -> 2.90 ms with 8 waves, 3.35 with 16; 49 x 49 (96^3) 2.55 / 2.54 / 2.95);
// HNO_GENERIC_WAVES_MID = 4 / 8 / 16 forces one form (A/B)
static int generic_waves_mid(int pe = 0) {
    static const int v = getenv("HNO_GENERIC_WAVES_MID") ? atoi(getenv("HNO_GENERIC_WAVES_MID")) : 0;
    return v ? v : (pe > 2800 ? 8 : 4);
}
static bool generic_waves8() { return generic_waves() == 8; }
static bool generic_waves16() { return generic_waves() != 8 && generic_waves() != 4; }

// HNO_INV_PLANE: "spec" = 1 (round-2 kernel: a workgroup per plane, axis-H result through LDS), default 0 = item kernel.  A/B aid.
static int inv_plane_variant() {
    static const int v = [] {
        const char *e = getenv("HNO_INV_PLANE");
        return (e && !strcmp(e, "spec")) ? 1 : 0;
    }();
    return v;
}

// HNO_FWD_PLANE: "wave" = 1 (round-2 kernel: register prefetch, axis-W result through LDS), "dma8" = 2 (DMA kernel, 8 waves x 1 slot),
// default 0 = DMA kernel with 4 waves x 2 slots.  A/B aid; read once.
static int fwd_plane_variant() {
    static const int v = [] {
        const char *e = getenv("HNO_FWD_PLANE");
        if (!e) return 0;
        if (!strcmp(e, "wave")) return 1;
        if (!strcmp(e, "dma8")) return 2;
        return 0;
    }();
    return v;
}

// HNO_MID_ZLAYOUT=0: the planes-only launches and the fused middle keep the [plane][part][k1][k2] intermediate (A/B aid; read once)
// HNO_ITEMS=0: plane sizes other than 65 / 33 run the specialised / generic workgroup-per-plane kernels as before round 5 (A/B)
static bool items_enabled() {
    static const bool v = !(getenv("HNO_ITEMS") && atoi(getenv("HNO_ITEMS")) == 0);
    return v;
}

bool mid_zlayout() {
    static const bool v = !(getenv("HNO_MID_ZLAYOUT") && atoi(getenv("HNO_MID_ZLAYOUT")) == 0);
    return v;
}
static int mid_zlayout_store() {   // 1: plain stores in the forward plane kernel; 2 (A/B): write-through stores as in the old layout
    static const int v = (getenv("HNO_MID_ZLAYOUT") && atoi(getenv("HNO_MID_ZLAYOUT")) == 2) ? 2 : 1;
    return v;
}

}  // namespace hno

using namespace hno;

extern "C" size_t hno_dht3_workspace_bytes(int BC, int N0, int N1, int N2, int m0, int m1, int m2) {
    if (BC <= 0 || N0 <= 0 || m1 <= 0 || m2 < 0) return 0;
    const int KP2 = ceil_div(m2 + 1, 16) * 16;
    return (size_t)BC * N0 * 2 * (2 * m1 + 1) * KP2 * sizeof(float);
}

// planes_only: stop after the plane kernel (the workspace then holds the axis-W / axis-H transform of every plane: the operand of
// dht_fwd_d_kernel or of the fused spectral middle, hno_specmid.hip)
static int dht_forward_launch(const float *x, const float *x_act_out, int act_grad, float *out, void *workspace,
                              int BC, int N0, int N1, int N2, int m0, int m1, int m2, float scale, void *stream,
                              int mode, int C, int full = 0, bool planes_only = false, long long ldbc = 0, int elem_bytes = 4) {
    HNO_REQUIRE(x && (out || planes_only) && workspace, "hno_dht3_crop: null pointer");
    const long long vol = (long long)N0 * N1 * N2;
    if (ldbc == 0) ldbc = vol;
    HNO_REQUIRE(ldbc >= vol && ldbc < vol + 64, "hno_dht3_crop: volume stride %lld for %lld voxels", ldbc, vol);
    int rc = check_sizes(BC, N0, N1, N2, m0, m1, m2, full, mode);
    if (rc) return rc;
    const DhtPlan *plan;
    rc = get_plan(N0, N1, N2, m0, m1, m2, &plan);
    if (rc) return rc;
    DhtArgs a;
    a.p = *plan;
    a.BC = BC;
    a.scale = scale;
    a.act = x_act_out ? act_grad : HNO_ACT_NONE;
    a.ldbc = ldbc != vol ? (unsigned)ldbc : 0u;
    a.dbg = debug_flags();
    a.stamps = (a.dbg & 64) ? debug_stamp_buffer() : nullptr;
    a.mode = mode;
    a.C = C > 0 ? C : 1;
    a.full0 = full & 1;
    a.full1 = (full >> 1) & 1;
    a.full2 = (full >> 2) & 1;
    a.zl = planes_only && mid_zlayout() ? mid_zlayout_store() : 0;
    a.zplanes = (unsigned)(BC * N0);
    const size_t lds = sizeof(float) * plan->f_lds_floats;
    if (lds > kMaxLds) return fail(HNO_ELIMIT, "hno_dht3_crop: plane %dx%d needs %zu B of LDS (> 160 KiB)", N1, N2, lds);
    hipStream_t s = (hipStream_t)stream;
    static int attr_done = -1;
    if (attr_done != current_device()) {
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)dht_fwd_plane_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)dht_fwd_plane_kernel<20>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)dht_fwd_plane_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)dht_fwd_plane_kernel<32, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)dht_fwd_plane_kernel<16, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)dht_fwd_plane_kernel<10, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)dht_fwd_plane_kernel<5, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
        attr_done = current_device();
    }
    const int planes = BC * N0;
    if (elem_bytes == 2) {      // bf16 planes (hno_dht3_planes_b16): the item kernel or nothing -- no fallback, no conversion pass
        HNO_REQUIRE(planes_only && !x_act_out, "hno_dht3_planes_b16: plane transforms only");
        ProfScope _ps(KID_DHT_FWD_PLANE, s, 2.0 * BC * (double)N0 * N1 * N2);
        const int rc16 = items_enabled() ? fwd_items_launch(x, (float *)workspace, a, BC, ldbc, s, 2) : 0;
        if (rc16 < 0) return rc16;
        if (rc16 == 0) return fail(HNO_ELIMIT, "hno_dht3_planes_b16: bf16 planes are built for 65 x 65 planes (got %d x %d)", N1, N2);
        g_last_plane_family[0] = 4;
        HNO_CHECK_LAUNCH();
        return HNO_OK;
    }
    // resident workgroups: LDS-limited; a grid of exactly that size lets every workgroup prefetch
    int per_cu = (int)(kMaxLds / lds);
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    const int grid = planes < 256 * per_cu ? planes : 256 * per_cu;
    const int pe = N1 * N2;
    {
        ProfScope _ps(KID_DHT_FWD_PLANE, s, 4.0 * BC * (double)N0 * N1 * N2 * (x_act_out ? 2 : 1));
        const Axis &b1 = plan->ax[1], &b2 = plan->ax[2];
        const bool spec_ok = pe <= 256 * 20 && b1.KT == 1 && b2.KT == 1 && N1 >= 16 && N2 >= 16 && !(a.dbg & 16);
        const size_t lds_spec = sizeof(float) * (16 + plan->MP1 * plan->lda2 + plan->TP * plan->ldt + (b1.KcP + b1.KsP) * 16);
        int g_spec = (int)(kMaxLds / lds_spec);
        if (g_spec > 8) g_spec = 8;
        g_spec = planes < 256 * g_spec ? planes : 256 * g_spec;
        bool launched = false;
        g_last_plane_family[0] = 1;
        // one wave per plane (no workgroup barriers): N1 = 32 g + 1, odd N2, no activation-gradient input
#define HNO_WAVE(KC2, KS2, KC1, KS1, NEV)                                                                                  \
    if (!launched && spec_ok && !x_act_out && !(a.dbg & 512) && N1 % 32 == 1 && (N2 & 1) && b2.KcP == 4 * KC2 &&           \
        b2.KsP == 4 * KS2 && b1.KcP == 4 * KC1 && b1.KsP == 4 * KS1 && b1.Js == b1.KsP && b2.Js == b2.KsP &&              \
        pe / 256 == NEV) {                                                                                                 \
        constexpr int NWV = 8;   /* one 8-wave workgroup per CU (4 waves per CU measured 10 % slower) */                   \
        auto kern = dht_fwd_plane_wave_kernel<KC2, KS2, KC1, KS1, NEV, NWV>;                                               \
        const size_t lds_w = sizeof(float) * NWV * (16 + ((pe + 32 + 15) & ~15));                                          \
        static int attr = -1;                                                                                          \
        if (attr != current_device()) {                                                                                                       \
            HNO_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds)); \
            attr = current_device();                                                                                                   \
        }                                                                                                                  \
        const int gw = persistent_grid((const void *)kern, 64 * NWV, lds_w, (planes + NWV - 1) / NWV);                     \
        hipLaunchKernelGGL(kern, dim3(gw), dim3(64 * NWV), lds_w, s, x, (float *)workspace, a);                            \
        g_last_plane_family[0] = 2; launched = true;                                                                       \
    }
        // planes by LDS-DMA, axis-W result in registers (see dht_fwd_plane_dma_kernel); HNO_FWD_PLANE=wave selects the older kernel
#define HNO_DMA(KC2, KS2, NP)                                                                                              \
    if (!launched && spec_ok && !x_act_out && !(a.dbg & 512) && fwd_plane_variant() != 1 && N1 == 32 * NP + 1 && (N2 & 1) && \
        b2.KcP == 4 * KC2 && b2.KsP == 4 * KS2 && b2.Js == b2.KsP && plan->NP1 == NP && b1.KT == 1 &&                      \
        (NP == 2 ? N2 == 65 : N2 <= 37) && (double)BC * ldbc < 1.0e9 && ((size_t)x & 3) == 0) {                            \
        const unsigned shift0 = (unsigned)(((size_t)x >> 2) & 3);                                                          \
        const float *xal = x - shift0;                                                                                     \
        const unsigned max_off = (unsigned)((((size_t)shift0 + (size_t)(BC - 1) * ldbc + (size_t)vol) * 4 - 1) & ~(size_t)15); \
        constexpr int NWV = 8;                                                                                             \
        auto kern = (a.dbg & 2) ? dht_fwd_plane_dma_kernel<KC2, KS2, NP, NWV, 1>                                          \
                                : (a.dbg & 4) ? dht_fwd_plane_dma_kernel<KC2, KS2, NP, NWV, 2>                            \
                                : a.zl == 1 ? dht_fwd_plane_dma_kernel<KC2, KS2, NP, NWV, 0, 1>                           \
                                : a.zl == 2 ? dht_fwd_plane_dma_kernel<KC2, KS2, NP, NWV, 0, 2> : dht_fwd_plane_dma_kernel<KC2, KS2, NP, NWV, 0>; \
        const size_t lds_w = (size_t)NWV * 2 * (NP == 2 ? 10 : 5) * 1024;                                                  \
        static int attr = -1;                                                                                          \
        if (attr != current_device() || (a.dbg & 6)) {                                                                                        \
            HNO_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds)); \
            /* (every layout variant of this size: the selection above changes from launch to launch) */                  \
            HNO_CHECK_HIP(hipFuncSetAttribute((const void *)dht_fwd_plane_dma_kernel<KC2, KS2, NP, NWV, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds)); \
            HNO_CHECK_HIP(hipFuncSetAttribute((const void *)dht_fwd_plane_dma_kernel<KC2, KS2, NP, NWV, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds)); \
            HNO_CHECK_HIP(hipFuncSetAttribute((const void *)dht_fwd_plane_dma_kernel<KC2, KS2, NP, NWV, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds)); \
            attr = current_device();                                                                                                   \
        }                                                                                                                  \
        static const int gforce = getenv("HNO_FWD_GRID") ? atoi(getenv("HNO_FWD_GRID")) : 0;                               \
        int gw = persistent_grid((const void *)kern, 64 * NWV, lds_w, (planes + NWV - 1) / NWV);                           \
        if (gforce > 0) gw = gforce < planes ? gforce : planes;                                                            \
        hipLaunchKernelGGL(kern, dim3(gw), dim3(64 * NWV), lds_w, s, xal, (float *)workspace, a, shift0, max_off,          \
                           planes / gw, planes % gw, (unsigned)ldbc);                                                      \
        g_last_plane_family[0] = 3; launched = true;                                                                       \
    }
        // plane sizes with an item kernel of hno_dht_items.hip (65 x 65 too: 17.8 against 18.9 us); HNO_ITEMS=0 keeps the older kernels (A/B)
        if (!launched && !x_act_out && !(a.dbg & (512 | 16)) && items_enabled() && fwd_plane_variant() != 1) {
            const int rc_items = fwd_items_launch(x, (float *)workspace, a, BC, ldbc, s);
            if (rc_items < 0) return rc_items;
            if (rc_items > 0) {
                g_last_plane_family[0] = 4;
                launched = true;
            }
        }
        HNO_DMA(9, 8, 2)           // 65 x 65 planes
        HNO_DMA(5, 4, 1)           // 33 x 33 planes
#undef HNO_DMA
        HNO_WAVE(9, 8, 9, 8, 16)   // 65 x 65 planes
        HNO_WAVE(5, 4, 5, 4, 4)    // 33 x 33 planes
#undef HNO_WAVE
#define HNO_SPEC(KC2, KS2, KC1, KS1)                                                                                      \
    if (!launched && spec_ok && b2.KcP == 4 * KC2 && b2.KsP == 4 * KS2 && b1.KcP == 4 * KC1 && b1.KsP == 4 * KS1) {      \
        if (x_act_out) {                                                                                                  \
            auto kern = dht_fwd_plane_spec_kernel<KC2, KS2, KC1, KS1, true>;                                              \
            hipLaunchKernelGGL(kern, dim3(persistent_grid((const void *)kern, 256, lds_spec, planes)), dim3(256), lds_spec, \
                               s, x, x_act_out, (float *)workspace, a);                                                   \
        } else {                                                                                                          \
            auto kern = dht_fwd_plane_spec_kernel<KC2, KS2, KC1, KS1, false>;                                             \
            hipLaunchKernelGGL(kern, dim3(persistent_grid((const void *)kern, 256, lds_spec, planes)), dim3(256), lds_spec, \
                               s, x, x_act_out, (float *)workspace, a);                                                   \
        }                                                                                                                 \
        g_last_plane_family[0] = 2; launched = true;                                                                      \
    }
        HNO_SPEC(9, 8, 9, 8)   // 65 x 65 planes (128^3 inputs)
        HNO_SPEC(5, 4, 5, 4)   // 33 x 33 planes (64^3 inputs)
        HNO_SPEC(8, 8, 8, 8)   // 61 x 61 planes (120 x 120 inputs)
        HNO_SPEC(8, 7, 8, 7)   // 57 x 57 planes (112^3 inputs)
        HNO_SPEC(7, 6, 7, 6)   // 49 x 49 planes (96^3 inputs)
        HNO_SPEC(6, 5, 6, 5)   // 41 x 41 planes (80^3 inputs)
#undef HNO_SPEC
        if (launched) {
        } else if (pe <= 256 * 20) {
            if (generic_waves_mid(pe) == 8) hipLaunchKernelGGL((dht_fwd_plane_kernel<10, 512>), dim3(grid), dim3(512), lds, s, x, x_act_out, (float *)workspace, a);
            else if (generic_waves_mid(pe) == 16) hipLaunchKernelGGL((dht_fwd_plane_kernel<5, 1024>), dim3(grid), dim3(1024), lds, s, x, x_act_out, (float *)workspace, a);
            else hipLaunchKernelGGL(dht_fwd_plane_kernel<20>, dim3(grid), dim3(256), lds, s, x, x_act_out, (float *)workspace, a);
        }
        else if (pe <= 256 * 64) {
            // large planes: sixteen waves per plane (generic_waves)
            if (generic_waves16()) hipLaunchKernelGGL((dht_fwd_plane_kernel<16, 1024>), dim3(grid), dim3(1024), lds, s, x, x_act_out, (float *)workspace, a);
            else if (generic_waves8()) hipLaunchKernelGGL((dht_fwd_plane_kernel<32, 512>), dim3(grid), dim3(512), lds, s, x, x_act_out, (float *)workspace, a);
            else hipLaunchKernelGGL(dht_fwd_plane_kernel<64>, dim3(grid), dim3(256), lds, s, x, x_act_out, (float *)workspace, a);
        } else
            hipLaunchKernelGGL(dht_fwd_plane_kernel<0>, dim3(grid), dim3(256), lds, s, x, x_act_out, (float *)workspace, a);
    }
    if (planes_only) {
        HNO_CHECK_LAUNCH();
        return HNO_OK;
    }
    const Axis &a0 = plan->ax[0];
    const size_t ldsd = sizeof(float) * a0.KT * (a0.KcP + a0.KsP) * 16;
    {
        ProfScope _ps(KID_DHT_FWD_D, s, 4.0 * BC * 8.0 * m0 * m1 * m2);
        const dim3 gd(plan->K1S * plan->ax[2].KT, BC);
        if (a0.KT == 1 && a0.KcP <= 36 && a0.KsP <= 32 && !(a.dbg & 32))
            hipLaunchKernelGGL((dht_fwd_d_fast_kernel<9, 8>), gd, dim3(64), 0, s, (const float *)workspace, out, a);
        else
            hipLaunchKernelGGL(dht_fwd_d_kernel, gd, dim3(64), ldsd, s, (const float *)workspace, out, a);
    }
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

// HNO_ITEM_SMALL: item count up to which the inverse item kernel runs with 8 instead of 16 waves per workgroup (0: never).  A/B aid.
static int item_small_limit() {
    static const int v = getenv("HNO_ITEM_SMALL") ? atoi(getenv("HNO_ITEM_SMALL")) : 2048;
    return v;
}

template <int NP, int N2c, int KM1, int NT2, int NWV, bool ZL>
static int inv_item_launch_z(const void *workspace, const float *add_al, float *out_al, const DhtArgs &a, unsigned shift0, int items,
                             long long ldbc, hipStream_t s, size_t lds_w, int gw);

template <int NP, int N2c, int KM1, int NT2, int NWV>
static int inv_item_launch(const void *workspace, const float *addend, float *out, const DhtArgs &a, unsigned shift0, int items,
                           long long ldbc, hipStream_t s) {
    float *out_al = out - shift0;
    const float *add_al = addend ? addend - shift0 : nullptr;
    const size_t lds_w = sizeof(float) * NWV * ((NP == 2 ? 2176 : round_up_c((32 * NP + 1) * N2c + 4, 64)) + 64);
    const int gw = items < 256 * NWV ? (items + NWV - 1) / NWV : 256;
    if (a.zl) return inv_item_launch_z<NP, N2c, KM1, NT2, NWV, true>(workspace, add_al, out_al, a, shift0, items, ldbc, s, lds_w, gw);
    return inv_item_launch_z<NP, N2c, KM1, NT2, NWV, false>(workspace, add_al, out_al, a, shift0, items, ldbc, s, lds_w, gw);
}

template <int NP, int N2c, int KM1, int NT2, int NWV, bool ZL>
static int inv_item_launch_z(const void *workspace, const float *add_al, float *out_al, const DhtArgs &a, unsigned shift0, int items,
                             long long ldbc, hipStream_t s, size_t lds_w, int gw) {
    const float *addend = add_al;
    if (addend) {
        auto kern = dht_inv_item_kernel<NP, N2c, KM1, NT2, true, NWV, ZL>;
        static int attr = -1;
        if (attr != current_device()) {
            HNO_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
            attr = current_device();
        }
        hipLaunchKernelGGL(kern, dim3(gw), dim3(64 * NWV), lds_w, s, (const float *)workspace, add_al, out_al, a, shift0, items / gw, items % gw,
                           (unsigned)ldbc);
    } else {
        auto kern = dht_inv_item_kernel<NP, N2c, KM1, NT2, false, NWV, ZL>;
        static int attr = -1;
        if (attr != current_device()) {
            HNO_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
            attr = current_device();
        }
        hipLaunchKernelGGL(kern, dim3(gw), dim3(64 * NWV), lds_w, s, (const float *)workspace, add_al, out_al, a, shift0, items / gw, items % gw,
                           (unsigned)ldbc);
    }
    return HNO_OK;
}

// planes_only: the workspace already holds the axis-D step's output (written by the fused spectral middle): plane kernel only
static int dht_inverse_launch(const float *z, const float *addend, int act, float *out, void *workspace,
                              int BC, int N0, int N1, int N2, int m0, int m1, int m2, float scale, void *stream,
                              int mode, int C, int full = 0, bool planes_only = false, long long ldbc = 0, int elem_bytes = 4) {
    HNO_REQUIRE((z || planes_only) && out && workspace, "hno_pad_idht3: null pointer");
    const long long vol = (long long)N0 * N1 * N2;
    if (ldbc == 0) ldbc = vol;
    HNO_REQUIRE(ldbc >= vol && ldbc < vol + 64, "hno_pad_idht3: volume stride %lld for %lld voxels", ldbc, vol);
    int rc = check_sizes(BC, N0, N1, N2, m0, m1, m2, full, mode);
    if (rc) return rc;
    const DhtPlan *plan;
    rc = get_plan(N0, N1, N2, m0, m1, m2, &plan);
    if (rc) return rc;
    DhtArgs a;
    a.p = *plan;
    a.BC = BC;
    a.scale = scale;
    a.act = act;
    a.ldbc = ldbc != vol ? (unsigned)ldbc : 0u;
    a.dbg = debug_flags();
    a.stamps = (a.dbg & 64) ? debug_stamp_buffer() : nullptr;
    a.mode = mode;
    a.C = C > 0 ? C : 1;
    a.full0 = full & 1;
    a.full1 = (full >> 1) & 1;
    a.full2 = (full >> 2) & 1;
    a.zl = planes_only && mid_zlayout() ? 1 : 0;
    a.zplanes = (unsigned)(BC * N0);
    const size_t lds = sizeof(float) * plan->i_lds_floats;
    if (lds > kMaxLds) return fail(HNO_ELIMIT, "hno_pad_idht3: plane %dx%d needs %zu B of LDS (> 160 KiB)", N1, N2, lds);
    hipStream_t s = (hipStream_t)stream;
    if (a.dbg & 8) a.act = HNO_ACT_NONE;
    static int attr_done = -1;
    if (attr_done != current_device()) {
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)dht_inv_plane_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)dht_inv_plane_kernel<20>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)dht_inv_plane_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)dht_inv_plane_kernel<32, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)dht_inv_plane_kernel<16, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)dht_inv_plane_kernel<10, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)dht_inv_plane_kernel<5, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
        attr_done = current_device();
    }
    const Axis &a0 = plan->ax[0];
    const size_t ldsd = sizeof(float) * 2 * a0.NT * a0.KmP * 16;
    if (!planes_only) {
        ProfScope _ps(KID_DHT_INV_D, s, 4.0 * BC * 8.0 * m0 * m1 * m2);
        const dim3 gd(plan->K1S * plan->ax[2].KT, BC);
        if (a0.NT <= 3 && !(a.dbg & 32))
            hipLaunchKernelGGL(dht_inv_d_kernel<3>, gd, dim3(64), 0, s, z, (float *)workspace, a);
        else
            hipLaunchKernelGGL(dht_inv_d_kernel<0>, gd, dim3(64), ldsd, s, z, (float *)workspace, a);
    }
    HNO_CHECK_LAUNCH();
    const int planes = BC * N0;
    if (elem_bytes == 2) {      // bf16 output planes (hno_idht3_planes_b16): the item kernel or nothing
        HNO_REQUIRE(planes_only, "hno_idht3_planes_b16: plane transforms only");
        ProfScope _ps(KID_DHT_INV_PLANE, s, BC * (double)N0 * N1 * N2 * (addend ? 6.0 : 2.0));
        const int rc16 = items_enabled() ? inv_items_launch(workspace, addend, (void *)out, a, BC, ldbc, s, 2) : 0;
        if (rc16 < 0) return rc16;
        if (rc16 == 0) return fail(HNO_ELIMIT, "hno_idht3_planes_b16: bf16 planes are built for 65 x 65 planes with output and addend at the same 4-element phase (got %d x %d)", N1, N2);
        g_last_plane_family[1] = 4;
        HNO_CHECK_LAUNCH();
        return HNO_OK;
    }
    int per_cu = (int)(kMaxLds / lds);
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    const int grid = planes < 256 * per_cu ? planes : 256 * per_cu;
    const int pe = N1 * N2;
    {
        ProfScope _ps(KID_DHT_INV_PLANE, s, 4.0 * BC * (double)N0 * N1 * N2 * (addend ? 2 : 1));
        const Axis &b1 = plan->ax[1], &b2 = plan->ax[2];
        const bool spec_ok = pe <= 256 * 20 && b1.KT == 1 && b2.KT == 1 && 2 * plan->CP <= 1024 && N1 >= 16 && N2 >= 16 &&
                             !(a.dbg & 16);
        const size_t lds_spec = sizeof(float) * (16 + 2 * plan->CP + 16 + 2 * plan->MP1 * plan->ldF + 256 * (pe / 256 + 1) + 2 * ((b1.J + 15) / 16) * b1.KmP * 16);
        int g_spec = (int)(kMaxLds / lds_spec);
        if (g_spec > 8) g_spec = 8;
        g_spec = planes < 256 * g_spec ? planes : 256 * g_spec;
        bool launched = false;
        g_last_plane_family[1] = 1;
        // one wave per half-plane item, GEMMs chained in registers (dht_inv_item_kernel); HNO_INV_PLANE=spec selects the older kernel
#define HNO_ITEM(NP, N2c, KM1, NT2)                                                                                        \
    if (!launched && spec_ok && !(a.dbg & 512) && inv_plane_variant() != 1 && N1 == 32 * NP + 1 && N2 == N2c && b1.KT == 1 && \
        b2.KT == 1 && b1.KmP == 4 * KM1 && b2.KmP <= 16 && plan->NP1 == NP && (b2.J + 15) / 16 == NT2 &&                   \
        (double)BC * ldbc < 1.0e9 && ((size_t)out & 3) == 0 && (!addend || ((size_t)addend & 15) == ((size_t)out & 15))) { \
        const unsigned shift0 = (unsigned)(((size_t)out >> 2) & 3);                                                        \
        const int items = planes * NP;                                                                                     \
        /* few items (one sample of a 12-channel model: 1 560): eight waves per workgroup spread them over twice the CUs */ \
        const int rc_item = (items <= item_small_limit())                                                                  \
            ? inv_item_launch<NP, N2c, KM1, NT2, 8>(workspace, addend, out, a, shift0, items, ldbc, s)                      \
            : inv_item_launch<NP, N2c, KM1, NT2, 16>(workspace, addend, out, a, shift0, items, ldbc, s);                    \
        if (rc_item) return rc_item;                                                                                       \
        g_last_plane_family[1] = 3; launched = true;                                                                       \
    }
        // (65 x 65 with a residual stays with dht_inv_item_kernel: 30.1 against 30.6 us; without one the item kernel runs 24.2 against 25.4)
        if (!launched && !(a.dbg & (512 | 16)) && items_enabled() && inv_plane_variant() != 1 && !(addend && N1 == 65 && N2 == 65)) {
            const int rc_items = inv_items_launch(workspace, addend, out, a, BC, ldbc, s);
            if (rc_items < 0) return rc_items;
            if (rc_items > 0) {
                g_last_plane_family[1] = 4;
                launched = true;
            }
        }
        HNO_ITEM(2, 65, 4, 2)      // 65 x 65 planes, modes (., 14, 14)
        HNO_ITEM(1, 33, 4, 1)      // 33 x 33 planes
#undef HNO_ITEM
#define HNO_SPEC(KM1, KM2, NT1, NT2, NFULL)                                                                               \
    if (!launched && spec_ok && b1.KmP == 4 * KM1 && b2.KmP == 4 * KM2 && (b1.J + 15) / 16 == NT1 &&                      \
        (b2.J + 15) / 16 == NT2 && pe / 256 == NFULL) {                                                                   \
        if (addend) {                                                                                                     \
            auto kern = dht_inv_plane_spec_kernel<KM1, KM2, NT1, NT2, true, NFULL>;                                       \
            hipLaunchKernelGGL(kern, dim3(persistent_grid((const void *)kern, 256, lds_spec, planes)), dim3(256), lds_spec, \
                               s, (const float *)workspace, addend, out, a);                                              \
        } else {                                                                                                          \
            auto kern = dht_inv_plane_spec_kernel<KM1, KM2, NT1, NT2, false, NFULL>;                                      \
            hipLaunchKernelGGL(kern, dim3(persistent_grid((const void *)kern, 256, lds_spec, planes)), dim3(256), lds_spec, \
                               s, (const float *)workspace, addend, out, a);                                              \
        }                                                                                                                 \
        g_last_plane_family[1] = 2; launched = true;                                                                      \
    }
        HNO_SPEC(4, 4, 2, 2, 16)   // 65 x 65 planes, modes (., 14, 14): positions 1..32
        HNO_SPEC(4, 4, 2, 2, 14)   // 61 x 61 planes: positions 1..30
        HNO_SPEC(4, 4, 1, 1, 4)    // 33 x 33 planes: positions 1..16
        HNO_SPEC(4, 4, 2, 2, 12)   // 57 x 57 planes: positions 1..28
        HNO_SPEC(4, 4, 2, 2, 9)    // 49 x 49 planes: positions 1..24
        HNO_SPEC(4, 4, 2, 2, 6)    // 41 x 41 planes: positions 1..20
#undef HNO_SPEC
        if (launched) {
        } else if (pe <= 256 * 20) {
            if (generic_waves_mid(pe) == 8) hipLaunchKernelGGL((dht_inv_plane_kernel<10, 512>), dim3(grid), dim3(512), lds, s, (const float *)workspace, addend, out, a);
            else if (generic_waves_mid(pe) == 16) hipLaunchKernelGGL((dht_inv_plane_kernel<5, 1024>), dim3(grid), dim3(1024), lds, s, (const float *)workspace, addend, out, a);
            else hipLaunchKernelGGL(dht_inv_plane_kernel<20>, dim3(grid), dim3(256), lds, s, (const float *)workspace, addend, out, a);
        }
        else if (pe <= 256 * 64) {
            if (generic_waves16()) hipLaunchKernelGGL((dht_inv_plane_kernel<16, 1024>), dim3(grid), dim3(1024), lds, s, (const float *)workspace, addend, out, a);
            else if (generic_waves8()) hipLaunchKernelGGL((dht_inv_plane_kernel<32, 512>), dim3(grid), dim3(512), lds, s, (const float *)workspace, addend, out, a);
            else hipLaunchKernelGGL(dht_inv_plane_kernel<64>, dim3(grid), dim3(256), lds, s, (const float *)workspace, addend, out, a);
        }
        else
            hipLaunchKernelGGL(dht_inv_plane_kernel<0>, dim3(grid), dim3(256), lds, s, (const float *)workspace, addend, out, a);
    }
    HNO_CHECK_LAUNCH();
    return HNO_OK;
}

extern "C" int hno_dht3_crop(const float *x, const float *x_act_out, int act_grad, float *out, void *workspace,
                             int BC, int N0, int N1, int N2, int m0, int m1, int m2, float scale, void *stream) {
    return dht_forward_launch(x, x_act_out, act_grad, out, workspace, BC, N0, N1, N2, m0, m1, m2, scale, stream, 0, 1);
}

// The two plane transforms alone, around the fused spectral middle (hno_spec_mid_fwd / _bwd, hno_specmid.hip)
extern "C" int hno_dht3_planes(const float *x, void *workspace, int BC, int N0, int N1, int N2, int m0, int m1, int m2, long long ldbc,
                               void *stream) {
    return dht_forward_launch(x, nullptr, HNO_ACT_NONE, nullptr, workspace, BC, N0, N1, N2, m0, m1, m2, 1.f, stream, 0, 1, 0, true, ldbc);
}

extern "C" int hno_idht3_planes(const void *workspace, const float *addend, int act, float *out, int BC, int N0, int N1, int N2, int m0,
                                int m1, int m2, float scale, long long ldbc, void *stream) {
    return dht_inverse_launch(nullptr, addend, act, out, (void *)workspace, BC, N0, N1, N2, m0, m1, m2, scale, stream, 0, 1, 0, true, ldbc);
}

// The same pair for activations that are bf16 IN MEMORY (round 6; include/hno.h: HNO_ACT_IO16): x / out are (BC, ldbc) arrays of 2-byte
// elements, the addend of the inverse stays fp32; workspace and arithmetic are the fp32 kernels'.  HNO_ELIMIT for plane sizes without a
// bf16 item kernel (built: 65 x 65).
extern "C" int hno_dht3_planes_b16(const void *x_bf16, void *workspace, int BC, int N0, int N1, int N2, int m0, int m1, int m2, long long ldbc,
                                   void *stream) {
    return dht_forward_launch((const float *)x_bf16, nullptr, HNO_ACT_NONE, nullptr, workspace, BC, N0, N1, N2, m0, m1, m2, 1.f, stream, 0, 1, 0, true,
                              ldbc, 2);
}

extern "C" int hno_idht3_planes_b16(const void *workspace, const float *addend, int act, void *out_bf16, int BC, int N0, int N1, int N2, int m0,
                                    int m1, int m2, float scale, long long ldbc, void *stream) {
    return dht_inverse_launch(nullptr, addend, act, (float *)out_bf16, (void *)workspace, BC, N0, N1, N2, m0, m1, m2, scale, stream, 0, 1, 0, true,
                              ldbc, 2);
}

// 1 when the transforms take a padded volume stride for this geometry: every plane kernel does (the 65 x 65 / 33 x 33 kernels through
// their stride argument, the generic / specialised workgroup-per-plane kernels through DhtArgs.ldbc); 0 only for sizes the transform
// itself refuses.  The host side asks before it hands out channel-padded activations.
extern "C" int hno_dht3_ld_supported(int N0, int N1, int N2, int m0, int m1, int m2) {
    if (N0 < 1 || N1 < 1 || N2 < 1 || m0 < 0 || m1 < 1 || m2 < 1) return 0;
    const DhtPlan *plan;
    if (get_plan(N0, N1, N2, m0, m1, m2, &plan)) {
        set_error("");
        return 0;
    }
    return sizeof(float) * plan->f_lds_floats <= kMaxLds && sizeof(float) * plan->i_lds_floats <= kMaxLds ? 1 : 0;
}

extern "C" int hno_dht3_crop_ld(const float *x, const float *x_act_out, int act_grad, float *out, void *workspace, int BC, int N0, int N1,
                                int N2, int m0, int m1, int m2, float scale, long long ldbc, void *stream) {
    return dht_forward_launch(x, x_act_out, act_grad, out, workspace, BC, N0, N1, N2, m0, m1, m2, scale, stream, 0, 1, 0, false, ldbc);
}

extern "C" int hno_pad_idht3_ld(const float *z, const float *addend, int act, float *out, void *workspace, int BC, int N0, int N1, int N2,
                                int m0, int m1, int m2, float scale, long long ldbc, void *stream) {
    return dht_inverse_launch(z, addend, act, out, workspace, BC, N0, N1, N2, m0, m1, m2, scale, stream, 0, 1, 0, false, ldbc);
}

extern "C" int hno_dht3_full(const float *x, float *out, void *workspace, int BC, int N0, int N1, int N2, float scale,
                             void *stream) {
    // every frequency of every axis: m = N / 2 keeps all of an even axis (the Nyquist term sits at -m), an
    // odd axis additionally keeps +m
    const int full = (N0 & 1) | ((N1 & 1) << 1) | ((N2 & 1) << 2);
    return dht_forward_launch(x, nullptr, HNO_ACT_NONE, out, workspace, BC, N0, N1, N2, N0 / 2, N1 / 2, N2 / 2, scale,
                              stream, 0, 1, full);
}

extern "C" int hno_pad_idht3(const float *z, const float *addend, int act, float *out, void *workspace,
                             int BC, int N0, int N1, int N2, int m0, int m1, int m2, float scale, void *stream) {
    return dht_inverse_launch(z, addend, act, out, workspace, BC, N0, N1, N2, m0, m1, m2, scale, stream, 0, 1);
}

extern "C" int hno_rfft3_crop(const float *x, const float *x_act_out, int act_grad, float *spec, void *workspace,
                              int B, int C, int N0, int N1, int N2, int m0, int m1, int m2, float scale, int k2_weights,
                              void *stream) {
    HNO_REQUIRE(B > 0 && C > 0, "hno_rfft3_crop: bad batch / channel count");
    return dht_forward_launch(x, x_act_out, act_grad, spec, workspace, B * C, N0, N1, N2, m0, m1, m2, scale, stream,
                              k2_weights ? 2 : 1, C);
}

extern "C" int hno_irfft3_pad(const float *spec, const float *addend, int act, float *out, void *workspace,
                              int B, int C, int N0, int N1, int N2, int m0, int m1, int m2, float scale, int k2_weights,
                              void *stream) {
    HNO_REQUIRE(B > 0 && C > 0, "hno_irfft3_pad: bad batch / channel count");
    return dht_inverse_launch(spec, addend, act, out, workspace, B * C, N0, N1, N2, m0, m1, m2, scale, stream,
                              k2_weights ? 2 : 1, C);
}

// the Fourier pair on channel-padded activations (ldbc: floats between consecutive (b, c) volumes of x, resp. out / addend)
extern "C" int hno_rfft3_crop_ld(const float *x, const float *x_act_out, int act_grad, float *spec, void *workspace, int B, int C, int N0,
                                 int N1, int N2, int m0, int m1, int m2, float scale, int k2_weights, long long ldbc, void *stream) {
    HNO_REQUIRE(B > 0 && C > 0, "hno_rfft3_crop: bad batch / channel count");
    return dht_forward_launch(x, x_act_out, act_grad, spec, workspace, B * C, N0, N1, N2, m0, m1, m2, scale, stream, k2_weights ? 2 : 1, C,
                              0, false, ldbc);
}

extern "C" int hno_irfft3_pad_ld(const float *spec, const float *addend, int act, float *out, void *workspace, int B, int C, int N0, int N1,
                                 int N2, int m0, int m1, int m2, float scale, int k2_weights, long long ldbc, void *stream) {
    HNO_REQUIRE(B > 0 && C > 0, "hno_irfft3_pad: bad batch / channel count");
    return dht_inverse_launch(spec, addend, act, out, workspace, B * C, N0, N1, N2, m0, m1, m2, scale, stream, k2_weights ? 2 : 1, C, 0,
                              false, ldbc);
}
