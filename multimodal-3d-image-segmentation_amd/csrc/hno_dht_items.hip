// Item plane kernels for general plane sizes (round 5): the scheme of dht_fwd_plane_dma_kernel / dht_inv_item_kernel (hno_dht.hip, built
// for the 65 x 65 and 33 x 33 planes of the benchmark) with the plane size as a template parameter, for the working grids of other
// image sizes -- 240 x 240 x 155 images (the reference's published inference size, README.md:10, experiments/train_test.py:383-426) give
// 121 x 78 planes, which ran the round-1 workgroup-per-plane kernels (84 / 104 us per launch).
//
// Reference semantics as in hno_dht.hip: nets/dht.py:16-36, nets/hnosegxs.py:378-410 (TransformCrop), :454-494 (PadInverse).
//
// An ITEM is one tile of 16 row pairs of a plane with odd N1: rows n1 = 1 + 16 X + i (i = 0..15) and their mirrors N1 - n1; item 0
// also holds row 0.  Js1 = (N1 - 1) / 2 need not be a multiple of 16: the last item is partial -- its surplus tile rows are read from
// (forward) or computed for (inverse) real rows of the plane, multiplied by zero table entries and never stored.  N2 may be even: the
// column N2 / 2 is its own mirror (forward: its cosine table row is halved; inverse: the mirror store is masked).
#include <type_traits>

#include "hno_common.h"
#include "hno_dht_plan.h"

namespace hno {

// ES: bytes per element of the FORWARD kernel's input planes (4: fp32, 2: bf16 -- round 6); only the DMA piece counts depend on it
template <int N1, int N2, int ES = 4>
struct ItemGeo {
    static_assert(N1 & 1, "odd N1");
    static constexpr int Js1 = (N1 - 1) / 2;
    static constexpr int NP = (Js1 + 15) / 16;                     // items per plane
    static constexpr int J2 = N2 / 2, Js2 = (N2 - 1) / 2;
    static constexpr int KC2 = (J2 + 1 + 3) / 4, KS2 = (Js2 + 3) / 4;   // k-steps of the axis-W cosine / sine parts (Axis::KcP / KsP over 4)
    static constexpr int NT2 = (J2 + 15) / 16;                     // inverse: output column tiles n2 = 1 + 16 nt + i
    static constexpr int hi(int X) { return 16 + 16 * X < Js1 ? 16 + 16 * X : Js1; }   // last row pair of item X
    static constexpr bool merged(int X) { return hi(X) == Js1; }   // the mirror rows follow the tile rows directly: one chunk
    static constexpr int p_r0(int X) { return X == 0 ? 0 : 1 + 16 * X; }
    static constexpr int p_rows(int X) { return merged(X) ? N1 - 16 * X - p_r0(X) : hi(X) + 1 - p_r0(X); }
    static constexpr int m_r0(int X) { return N1 - hi(X); }
    static constexpr int m_rows(int X) { return merged(X) ? 0 : hi(X) - 16 * X; }
    // KiB pieces of a chunk: its rows plus the elements between the 16-byte boundary below it and its first element
    static constexpr int pieces_of(int rows) { return rows == 0 ? 0 : (rows * N2 * ES + (16 - ES) + 1023) / 1024; }
    static constexpr int p_pieces(int X) { return pieces_of(p_rows(X)); }
    static constexpr int m_pieces(int X) { return pieces_of(m_rows(X)); }
    static constexpr int pieces(int X) { return p_pieces(X) + m_pieces(X); }
    static constexpr int slot_pieces() {
        int m = 0;
        for (int X = 0; X < NP; ++X) m = pieces(X) > m ? pieces(X) : m;
        return m;
    }
    // inverse: LDS image of an item's rows (floats): chunk 0 at 0, chunk 1 (the mirror rows of an item that is not the last) at C1LO
    static constexpr int C1LO = round_up_c(17 * N2 + 3, 4);
    static constexpr int OBUF = round_up_c((C1LO + 16 * N2 + 3) > (32 * N2 + 3) ? (C1LO + 16 * N2 + 3) : (32 * N2 + 3), 64);
    static constexpr int WSTRIDE = OBUF + 64;                      // + scratch for row 0
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ---- forward: one wave per plane, items streamed into a two-slot LDS ring by DMA, axis-W result kept in registers -----------------
// (dht_fwd_plane_dma_kernel documents the scheme; differences: chunk geometry from ItemGeo, any number of items per plane, the slot of
// an item is the parity of its running number, operands whose table entries are zero padding are selected away where they may lie
// outside the wave's ring.)
// T: float, or unsigned short = bf16 planes (the activations torch.autocast keeps in bf16: round 6).  The planes land in LDS as they lie
// in memory; every operand read widens its halfword (bits << 16), all arithmetic is the fp32 kernel's.  shift0 / ldbc / offsets of the
// ring are in ELEMENTS, max_off in bytes.
template <int N1, int N2, int NWV, int ZL, typename T = float>
__global__ __launch_bounds__(64 * NWV, 1) void dht_fwd_items_kernel(const T *__restrict__ xal, float *__restrict__ Y, DhtArgs a,
                                                                    unsigned shift0, unsigned max_off, int pl_base, int pl_rem,
                                                                    unsigned ldbc) {
    constexpr int ES = (int)sizeof(T), EPV = 16 / ES, EPK = 1024 / ES;     // bytes per element, elements per 16 bytes / per KiB piece
    using G = ItemGeo<N1, N2, ES>;
    auto ldv = [](const T *q_) -> float {
        if constexpr (ES == 4) return *q_;
        else return __builtin_bit_cast(float, (unsigned)*q_ << 16);
    };
    constexpr int NP = G::NP, KC2 = G::KC2, KS2 = G::KS2, Js2 = G::Js2;
    static_assert(NP >= 2, "at least two items per plane");
    constexpr int SLOTF = G::slot_pieces() * EPK;          // elements per ring slot
    extern __shared__ float lds[];
    const DhtPlan &p = a.p;
    const Axis &a1 = p.ax[1];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, l15 = lane & 15;
    constexpr unsigned pe = (unsigned)(N1 * N2);
    const int N0p = p.ax[0].N;
    auto plane_f0 = [&](int plane) -> unsigned {
        const int bc = plane / N0p;
        return shift0 + (unsigned)bc * ldbc + (unsigned)(plane - bc * N0p) * pe;
    };
    T *ring = reinterpret_cast<T *>(lds + 4) + (size_t)wave * 2 * SLOTF;      // (+ 16 bytes: an operand two elements below a chunk stays inside LDS)
    const unsigned ring_b = (unsigned)(size_t)ring;
    const int bid = blockIdx.x;
    const int p_begin = bid * pl_base + (bid < pl_rem ? bid : pl_rem);
    const int p_end = p_begin + pl_base + (bid < pl_rem ? 1 : 0);
    auto issue_chunk = [&](int plane, int r0, int NPC, unsigned dst) {
        const unsigned f0 = plane_f0(plane) + (unsigned)(r0 * N2);
        const unsigned boff = (f0 & ~(unsigned)(EPV - 1)) * (unsigned)ES + (unsigned)lane * 16u;
        const unsigned first = __builtin_amdgcn_readfirstlane(boff);
        if (first + (unsigned)NPC * 1024u <= max_off) {
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
    auto issue_item = [&](int plane, auto Xc, int slot) {
        constexpr int X = decltype(Xc)::value;
        const unsigned dst = ring_b + (unsigned)slot * (SLOTF * ES);
        issue_chunk(plane, G::p_r0(X), G::p_pieces(X), dst);
        if constexpr (G::m_rows(X) > 0) issue_chunk(plane, G::m_r0(X), G::m_pieces(X), dst + G::p_pieces(X) * 1024);
    };
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
    constexpr int PRO = G::pieces(0) + G::pieces(1);
    static_assert(PRO <= 63, "prologue pieces exceed the vmcnt range");
    if (plane < p_end) {
        issue_item(plane, std::integral_constant<int, 0>{}, 0);
        issue_item(plane, std::integral_constant<int, 1>{}, 1);
        dma_wait<PRO>();
    } else
        dma_wait<0>();
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
    const bool q0 = q == 0;
    const size_t zoff_p = ((size_t)((a1.m + (l15 <= a1.m ? l15 : 0)) * 4 + q) * a.zplanes) * 8;
    const size_t zoff_m = ((size_t)((a1.m - (l15 <= a1.m ? l15 : 0)) * 4 + q) * a.zplanes) * 8;
    // row r of the plane in the slot of item (pl, X): rowsP + r N2 (tile rows), rowsM + r N2 (mirror rows)
    auto item_rows = [&](int pl, auto Xc, int slot, const T *&rowsP, const T *&rowsM) {
        constexpr int X = decltype(Xc)::value;
        const unsigned g0 = plane_f0(pl);
        const T *sl = ring + slot * SLOTF;
        rowsP = sl + ((g0 + (unsigned)(G::p_r0(X) * N2)) & (unsigned)(EPV - 1)) - G::p_r0(X) * N2;
        if constexpr (G::merged(X)) rowsM = rowsP;
        else rowsM = sl + G::p_pieces(X) * EPK + ((g0 + (unsigned)(G::m_r0(X) * N2)) & (unsigned)(EPV - 1)) - G::m_r0(X) * N2;
    };
    auto read_cos = [&](int pl, auto Xc, int slot, float (&ra0)[KC2], float (&rb0)[KC2], float (&ra1)[KC2], float (&rb1)[KC2]) {
        constexpr int X = decltype(Xc)::value;
        const T *rowsP, *rowsM;
        item_rows(pl, Xc, slot, rowsP, rowsM);
        int rp = 1 + 16 * X + l15;
        if constexpr (16 + 16 * X > G::Js1) rp = rp < G::Js1 ? rp : G::Js1;   // partial item: surplus tile rows read the last pair again
        const T *row0 = rowsP + rp * N2, *row1 = rowsM + (N1 - rp) * N2;
        const T *pf0 = row0 + q, *pb0 = row0 + N2 - q - 4 * (KC2 - 1), *pf1 = row1 + q, *pb1 = row1 + N2 - q - 4 * (KC2 - 1);
#pragma unroll
        for (int ks = 0; ks < KC2; ++ks) {
            ra0[ks] = ldv(pf0 + 4 * ks);
            rb0[ks] = ldv(pb0 + 4 * (KC2 - 1 - ks));
            ra1[ks] = ldv(pf1 + 4 * ks);
            rb1[ks] = ldv(pb1 + 4 * (KC2 - 1 - ks));
        }
    };
    float a0[KC2], b0[KC2], a1_[KC2], b1[KC2];
    if (plane < p_end) {
        dma_wait<G::pieces(1)>();
        read_cos(plane, std::integral_constant<int, 0>{}, 0, a0, b0, a1_, b1);
    }
    int it = 0;
    for (; plane < p_end; plane += NWV, ++it) {
        const bool more = plane + NWV < p_end;
        f32x4 pA = {0.f, 0.f, 0.f, 0.f}, pB = pA, qA = pA, qB = pA;   // axis-H sums of the plane
        static_for<0, NP>([&](auto Xc) {
            constexpr int X = decltype(Xc)::value;
            // slot of item (it, X) = parity of its running number it NP + X
            const int slot = (NP & 1) ? ((it + X) & 1) : (X & 1);
            constexpr int X2 = (X + 2) % NP, D2 = (X + 2) / NP;         // the item two ahead refills this slot
            const int rf_plane = plane + D2 * NWV;
            const bool refill = rf_plane < p_end;
            const T *rowsP, *rowsM;
            item_rows(plane, Xc, slot, rowsP, rowsM);
            int rp = 1 + 16 * X + l15;
            if constexpr (16 + 16 * X > G::Js1) rp = rp < G::Js1 ? rp : G::Js1;
            const T *row0 = rowsP + rp * N2, *row1 = rowsM + (N1 - rp) * N2;
            float sa0[KS2], sb0[KS2], sa1[KS2], sb1[KS2];
            {
                const T *sf0 = row0 + Js2 - q - 4 * (KS2 - 1), *sb0_ = row0 + N2 - Js2 + q;
                const T *sf1 = row1 + Js2 - q - 4 * (KS2 - 1), *sb1_ = row1 + N2 - Js2 + q;
#pragma unroll
                for (int ks = 0; ks < KS2; ++ks) {
                    sa0[ks] = ldv(sf0 + 4 * (KS2 - 1 - ks));
                    sb0[ks] = ldv(sb0_ + 4 * ks);
                    sa1[ks] = ldv(sf1 + 4 * (KS2 - 1 - ks));
                    sb1[ks] = ldv(sb1_ + 4 * ks);
                }
            }
            float r0a[KC2], r0b[KC2], r0c[KS2], r0d[KS2];   // row 0 (item 0 only)
            if constexpr (X == 0) {
                const T *rz = rowsP;
                const T *pf = rz + q, *pb = rz + N2 - q - 4 * (KC2 - 1), *sf = rz + Js2 - q - 4 * (KS2 - 1), *sb = rz + N2 - Js2 + q;
#pragma unroll
                for (int ks = 0; ks < KC2; ++ks) {
                    r0a[ks] = ldv(pf + 4 * ks);
                    r0b[ks] = ldv(pb + 4 * (KC2 - 1 - ks));
                }
#pragma unroll
                for (int ks = 0; ks < KS2; ++ks) {
                    r0c[ks] = ldv(sf + 4 * (KS2 - 1 - ks));
                    r0d[ks] = ldv(sb + 4 * ks);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // c = 0 has no mirror element: what was read there is dropped by a select, so no garbage is ever multiplied
            b0[0] = q0 ? 0.f : b0[0];
            b1[0] = q0 ? 0.f : b1[0];
#pragma unroll
            for (int ks = 0; ks < KC2; ++ks) {
                a0[ks] += b0[ks];
                a1_[ks] += b1[ks];
            }
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, s0 = c0, c1 = c0, s1 = c0;
            float pc = 0.f, ps = 0.f;   // row 0: this lane group's share of the sums over the folded columns
            constexpr int KSPLIT = KC2 / 2;
#pragma unroll
            for (int ks = 0; ks < KSPLIT; ++ks) {
                c0 = mfma16(a0[ks], bwc[ks], c0);
                c1 = mfma16(a1_[ks], bwc[ks], c1);
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every LDS read of the item has returned: the slot is free
            if (refill) issue_item(rf_plane, std::integral_constant<int, X2>{}, slot);
#pragma unroll
            for (int ks = KSPLIT; ks < KC2; ++ks) {
                __builtin_amdgcn_sched_barrier(0);
                c0 = mfma16(a0[ks], bwc[ks], c0);
                c1 = mfma16(a1_[ks], bwc[ks], c1);
#pragma unroll
                for (int j = 2 * (ks - KSPLIT); j < 2 * (ks - KSPLIT) + 2; ++j) {
                    if (j < KS2) {
                        sa0[j] -= sb0[j];
                        sa1[j] -= sb1[j];
                        if (4 * j + 3 >= Js2) {   // sine positions >= Js2 are table padding (their operands may lie outside the chunk)
                            sa0[j] = 4 * j + q < Js2 ? sa0[j] : 0.f;
                            sa1[j] = 4 * j + q < Js2 ? sa1[j] : 0.f;
                        }
                    }
                    if constexpr (X == 0) {
                        if (j < KC2) pc = fmaf(r0a[j] + ((j == 0 && q0) ? 0.f : r0b[j]), bwc[j], pc);
                        if (j < KS2) {
                            float d = r0c[j] - r0d[j];
                            if (4 * j + 3 >= Js2) d = 4 * j + q < Js2 ? d : 0.f;
                            ps = fmaf(d, bws[j], ps);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- sine part, with the axis-H products of the finished cosine part between its MFMAs
            float fa[4], fb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                fa[r] = c0[r] + c1[r];
                fb[r] = c0[r] - c1[r];
            }
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                __builtin_amdgcn_sched_barrier(0);
                s0 = mfma16(sa0[ks], bws[ks], s0);
                s1 = mfma16(sa1[ks], bws[ks], s1);
                if ((ks & 1) == 0 && ks / 2 < 4) pA = mfma16(fa[ks / 2], bhc[X][ks / 2], pA);
                if ((ks & 1) == 1 && ks / 2 < 4) qA = mfma16(fb[ks / 2], bhs[X][ks / 2], qA);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (2 * r >= KS2) pA = mfma16(fa[r], bhc[X][r], pA);
                if (2 * r + 1 >= KS2) qA = mfma16(fb[r], bhs[X][r], qA);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- the next item has landed when at most the pieces of the refill just issued are outstanding: fetch its cosine operands
            {
                const bool have_next = X + 1 < NP || more;
                if (have_next) {
                    if (refill) dma_wait<G::pieces(X2)>();
                    else dma_wait<0>();
                    if constexpr (X + 1 < NP) read_cos(plane, std::integral_constant<int, X + 1>{}, (NP & 1) ? ((it + X + 1) & 1) : ((X + 1) & 1), a0, b0, a1_, b1);
                    else read_cos(plane + NWV, std::integral_constant<int, 0>{}, (NP & 1) ? ((it + 1) & 1) : 0, a0, b0, a1_, b1);
                }
            }
            if constexpr (X == 0) {   // row 0: the sum over the four lane groups is the K sum of an MFMA against ones
                pA = mfma16(pc, 1.f, pA);
                pB = mfma16(ps, 1.f, pB);
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
                pB = mfma16(fc[r], bhc[X][r], pB);
                qB = mfma16(fd[r], bhs[X][r], qB);
            }
        });
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
                float *d0 = Y + zoff_p + (size_t)plane * 8, *d1 = Y + zoff_m + (size_t)plane * 8;
                *reinterpret_cast<f32x4 *>(d0) = vp0;
                *reinterpret_cast<f32x4 *>(d0 + 4) = vp1;
                if (k1 >= 1) {
                    *reinterpret_cast<f32x4 *>(d1) = vm0;
                    *reinterpret_cast<f32x4 *>(d1 + 4) = vm1;
                }
            } else {
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

// ---- inverse: one wave per item, both GEMMs chained in registers, the item's rows through a wave-private LDS image ---------------
// (dht_inv_item_kernel documents the scheme; differences: chunk geometry from the item number at run time -- a wave's items all have
// the same X because NWV is a multiple of the items per plane --, partial tiles masked on store, even N2.)
// TO: float, or unsigned short = bf16 OUTPUT planes (round 6: the gradient of a block input that torch.autocast keeps in bf16).  The LDS
// image, the residual (add_al, fp32) and all arithmetic are the fp32 kernel's; the epilogue rounds (nearest even) and stores four
// elements = 8 bytes per lane.  out_al is 4-element aligned (16 bytes for fp32, 8 for bf16); shift0 / ldbc are in elements.
template <int N1, int N2, bool HAS_ADD, int NWV, bool ZL, typename TO = float>
__global__ __launch_bounds__(64 * NWV, (NWV + 3) / 4) void dht_inv_items_kernel(const float *__restrict__ E, const float *__restrict__ add_al,
                                                                               TO *__restrict__ out_al, DhtArgs a, unsigned shift0,
                                                                               int it_base, int it_rem, unsigned ldbc) {
    constexpr bool O16 = sizeof(TO) == 2;
    auto to_out = [](float v) -> TO {
        if constexpr (O16) return __builtin_bit_cast(unsigned short, (__bf16)v);
        else return v;
    };
    using G = ItemGeo<N1, N2>;
    constexpr int NP = G::NP, KM1 = 4, NT2 = G::NT2, J2 = G::J2, Js2 = G::Js2, Js1 = G::Js1;
    static_assert(NWV % NP == 0, "the items of a wave share X");
    constexpr int OBUF = G::OBUF, WSTRIDE = G::WSTRIDE, C1LO = G::C1LO;
    extern __shared__ float lds[];
    const DhtPlan &p = a.p;
    const Axis &a1 = p.ax[1];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, l15 = lane & 15;
    const int m1 = a1.m;
    float *obuf = lds + (size_t)wave * WSTRIDE, *scr = obuf + OBUF;
    const int bid = blockIdx.x;
    const int t_begin = bid * it_base + (bid < it_rem ? bid : it_rem);
    const int t_end = t_begin + it_base + (bid < it_rem ? 1 : 0);
    int t = t_begin + wave;
    const int X = __builtin_amdgcn_readfirstlane(t % NP);
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
    constexpr unsigned pe = N1 * N2;
    const int ne = 2 * p.CP;                                   // floats per intermediate plane
    const float ap = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE : 1.f;
    const float aq = a.act == HNO_ACT_SELU ? HNO_SELU_SCALE * HNO_SELU_ALPHA : 1.f;
    const bool lin = a.act == HNO_ACT_NONE;
    const bool q0 = q == 0;
    // chunk geometry of this wave's items: chunk 0 = rows [c0_r0, c0_r0 + c0_n) at LDS float 0, chunk 1 = the mirror rows
    // [c1_r0, c1_r0 + c1_n) at C1LO; the last item of a plane has its mirror rows directly behind its tile rows: one chunk
    const int hi = 16 + 16 * X < Js1 ? 16 + 16 * X : Js1;
    const bool merged = X == NP - 1;
    const int c0_r0 = X ? 1 + 16 * X : 0;
    const int c0_n = merged ? N1 - 16 * X - c0_r0 : hi + 1 - c0_r0;
    const int c1_r0 = N1 - hi, c1_n = merged ? 0 : hi - 16 * X;
    const int nch = merged ? 1 : 2;
    float erp[KM1], erm[KM1], eip[KM1], eim[KM1];
    auto load_e = [&](int tt) {
        const int pl = tt / NP;
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
    if (t < t_end) load_e(t);
    for (; t < t_end; t += NWV) {
        const int plane = t / NP;
        const int N0p = p.ax[0].N, bcv = plane / N0p, n0v = plane - bcv * N0p;
        const unsigned f0 = shift0 + (unsigned)bcv * ldbc + (unsigned)n0v * pe;
        if (n0v == N0p - 1 && X == 0 && ldbc > (unsigned)N0p * pe) {
            const unsigned npad = ldbc - (unsigned)N0p * pe;
            if ((unsigned)lane < npad) out_al[shift0 + (unsigned)bcv * ldbc + (unsigned)N0p * pe + lane] = to_out(0.f);
        }
        const unsigned fA = f0 + (unsigned)(c0_r0 * N2), fC = f0 + (unsigned)(c1_r0 * N2);
        const unsigned shA = fA & 3u, shC = fC & 3u;
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
        if (t + NWV < t_end) load_e(t + NWV);
        __builtin_amdgcn_sched_barrier(0);
        if (HAS_ADD) {
            // the residual rows go by LDS-DMA straight into the item's image (see dht_inv_item_kernel)
            const unsigned ob = (unsigned)(size_t)obuf;
#pragma unroll 1
            for (int ch = 0; ch < nch; ++ch) {
                const unsigned fch = ch ? fC : fA;
                const unsigned n = (unsigned)((ch ? c1_n : c0_n) * N2);
                const unsigned lim = (fch & 3u) + n, lo = ch ? (unsigned)C1LO : 0u;
                const int ng = (int)((lim + 255u) >> 8);
                const unsigned goff = ((fch & ~3u) + 4u * lane) * 4u;
#pragma unroll 1
                for (int j = 0; j < ng; ++j)
                    if (4u * (64u * j + lane) < lim)
                        dma_piece16(add_al, goff + 1024u * j, __builtin_amdgcn_readfirstlane(ob + (lo + 256u * j) * 4u));
            }
        }
        // ---- axis H, tile X
        f32x4 cRe = {0.f, 0.f, 0.f, 0.f}, cIm = cRe, sRe = cRe, sIm = cRe;
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
        // ---- LDS image of the item's rows: plane row n1 of chunk 0 at img0 + n1 N2, of chunk 1 at img1 + n1 N2
        float *img0 = obuf + shA - c0_r0 * N2, *img1 = merged ? img0 : obuf + C1LO + shC - c1_r0 * N2;
        const int n1p = 1 + 16 * X + 4 * q, n1m = N1 - n1p;     // rows of accumulator register 0 (plus tile ascending, mirror descending)
        float *rowP = img0 + n1p * N2, *rowM = img1 + n1m * N2;
        auto put = [&](float *dst, float v) {
            if (HAS_ADD) *dst = fmaf(v, a.scale, *dst);
            else *dst = v;
        };
        // ---- axis W: O[n1][n2] = sum_k2 FR cos - FI sin, mirror column N2 - n2 gets +; the A operands are the F registers
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) {
            f32x4 pc = {0.f, 0.f, 0.f, 0.f}, ps = pc, mc = pc, ms = pc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pc = mfma16(FRp[r], bwc[nt][r], pc);
                ps = mfma16(FIp[r], bws[nt][r], ps);
                mc = mfma16(FRm[r], bwc[nt][r], mc);
                ms = mfma16(FIm[r], bws[nt][r], ms);
            }
            const int n2 = 1 + 16 * nt + l15;
            // partial tiles: columns beyond J2 and rows beyond the item's last pair are not stored (they would land on valid elements);
            // the column N2 / 2 of an even N2 is its own mirror
            const bool okc = 16 * (nt + 1) <= J2 || n2 <= J2, okm = 16 * (nt + 1) <= Js2 || n2 <= Js2;
            if (HAS_ADD && nt == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the residual is in the image
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (n1p + r <= hi) {
                    if (okc) {
                        put(rowP + r * N2 + n2, pc[r] - ps[r]);
                        put(rowM - r * N2 + n2, mc[r] - ms[r]);
                    }
                    if (okm) {
                        put(rowP + r * N2 + N2 - n2, pc[r] + ps[r]);
                        put(rowM - r * N2 + N2 - n2, mc[r] + ms[r]);
                    }
                }
            }
        }
        // ---- output column n2 = 0: cos = 1, sin = 0 -> the plain sum of FR over k2 = over (q, r)
        {
            float zp = (FRp[0] + FRp[1]) + (FRp[2] + FRp[3]), zm = (FRm[0] + FRm[1]) + (FRm[2] + FRm[3]);
            zp += __shfl_xor(zp, 16);
            zm += __shfl_xor(zm, 16);
            zp += __shfl_xor(zp, 32);
            zm += __shfl_xor(zm, 32);
            if (q0 && 1 + 16 * X + l15 <= hi) {
                put(img0 + (1 + 16 * X + l15) * N2, zp);
                put(img1 + (N1 - 1 - 16 * X - l15) * N2, zm);
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
                    if (n2 <= J2) put(img0 + n2, c - s_);
                    if (n2 <= Js2) put(img0 + N2 - n2, c + s_);
                }
            }
            if (lane == 0) put(img0, z0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- epilogue: out = act(scale * O + residual) in aligned 16-byte groups (a loop: see dht_inv_item_kernel)
#pragma unroll 1
        for (int ch = 0; ch < nch; ++ch) {
            const unsigned fch = ch ? fC : fA;
            const unsigned n = (unsigned)((ch ? c1_n : c0_n) * N2);
            const float *limg = obuf + (ch ? C1LO : 0);
            const unsigned sh = fch & 3u;
            const int ng = (int)((sh + n + 255u) >> 8);
            TO *gbase = out_al + (fch & ~3u) + 4 * lane;
            const f32x2 sc = {HAS_ADD ? 1.f : a.scale, HAS_ADD ? 1.f : a.scale};
            f32x4 onext = *reinterpret_cast<const f32x4 *>(limg + 4 * lane);
#pragma unroll 1
            for (int j = 0; j < ng; ++j) {
                const int e0 = (int)(4 * (64u * j + lane)) - (int)sh;
                const f32x4 o = onext;
                if (j + 1 < ng) onext = *reinterpret_cast<const f32x4 *>(limg + 256 * (j + 1) + 4 * lane);
                f32x2 x0 = f32x2{o[0], o[1]} * sc;
                f32x2 x1 = f32x2{o[2], o[3]} * sc;
                if (!lin) {   // wave-uniform
                    x0 = selu_like_pk(x0, ap, aq);
                    x1 = selu_like_pk(x1, ap, aq);
                }
                const f32x4 v = {x0[0], x0[1], x1[0], x1[1]};
                if (e0 >= 0 && e0 + 3 < (int)n) {
                    if constexpr (O16) {
                        typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
                        *reinterpret_cast<u16x4 *>(gbase + 256 * j) = u16x4{to_out(v[0]), to_out(v[1]), to_out(v[2]), to_out(v[3])};
                    } else {
                        *reinterpret_cast<f32x4 *>(gbase + 256 * j) = v;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (e0 + c >= 0 && e0 + c < (int)n) gbase[256 * j + c] = to_out(v[c]);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------------- launchers
static const size_t kLdsMax = 160 * 1024;

// plane sizes the item kernels are built for: (N1, N2, waves per workgroup forward, inverse) -- the cubic working grids of 80^3 ... 256^3
// inputs, the 121 x 78 planes of 240 x 240 x 155 images and the 97 x 65 planes of 160 x 192 x 128 volumes (BASELINE cfg4's size) (from 105 up the forward kernel's registers overflow into AGPRs, no scratch).
// Waves per workgroup are measured: forward 121 x 78 26.9 us with 4, 30.0 with 6, 31.2 with 7; inverse 45.4 us with 12, 50.0 with 8,
// 73.5 with 4; the smaller cubes differ by < 3 %.  65 x 65 (the benchmark's planes): the forward kernel with SEVEN waves runs 17.8 us
// where dht_fwd_plane_dma_kernel (eight) runs 18.9 in the same step (6: 18.5, 5: 18.2, 4: 18.3); the inverse without residual 24.2
// against dht_inv_item_kernel's 25.4 (with residual 30.6 against 30.1: that one stays with the older kernel, see dht_inverse_launch).
#define HNO_ITEM_SIZES(X) X(65, 65, 7, 12) X(121, 78, 4, 12) X(97, 65, 6, 12) X(41, 41, 8, 12) X(49, 49, 8, 12) X(57, 57, 8, 12) X(73, 73, 6, 12) X(81, 81, 4, 12) X(89, 89, 4, 12) X(97, 97, 4, 12) X(105, 105, 4, 8) X(113, 113, 4, 8) X(121, 121, 4, 8) X(129, 129, 4, 8)

template <int N1, int N2, int NWV, typename T = float>
static int fwd_items_launch_t(const T *xal, float *ws, const DhtArgs &a, unsigned shift0, unsigned max_off, int planes, unsigned ldbc,
                              hipStream_t s) {
    using G = ItemGeo<N1, N2, (int)sizeof(T)>;
    const size_t lds = (size_t)NWV * 2 * G::slot_pieces() * 1024 + 128;
    static_assert((size_t)NWV * 2 * G::slot_pieces() * 1024 + 128 <= 160 * 1024, "ring exceeds LDS");
    static int attr = -1;
    if (attr != current_device()) {
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)dht_fwd_items_kernel<N1, N2, NWV, 0, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax));
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)dht_fwd_items_kernel<N1, N2, NWV, 1, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax));
        attr = current_device();
    }
    const int units = (planes + NWV - 1) / NWV;
    int gw = units < 256 ? units : 256;
    if (a.zl) hipLaunchKernelGGL((dht_fwd_items_kernel<N1, N2, NWV, 1, T>), dim3(gw), dim3(64 * NWV), lds, s, xal, ws, a, shift0, max_off, planes / gw, planes % gw, ldbc);
    else hipLaunchKernelGGL((dht_fwd_items_kernel<N1, N2, NWV, 0, T>), dim3(gw), dim3(64 * NWV), lds, s, xal, ws, a, shift0, max_off, planes / gw, planes % gw, ldbc);
    return 1;
}

// 1: launched; 0: no item kernel for this geometry (the caller falls back to the workgroup-per-plane kernels); < 0: error
int fwd_items_launch(const void *x, float *workspace, const DhtArgs &a, int BC, long long ldbc, hipStream_t s, int elem_bytes) {
    const DhtPlan &p = a.p;
    const int N0 = p.ax[0].N, N1 = p.ax[1].N, N2 = p.ax[2].N;
    const long long vol = (long long)N0 * N1 * N2;
    const size_t es = (size_t)elem_bytes;
    if (p.ax[1].KT != 1 || p.ax[2].KT != 1 || (double)BC * ldbc >= 1.0e9 || ((size_t)x & (es - 1))) return 0;
    const unsigned shift0 = (unsigned)(((size_t)x / es) & (16 / es - 1));      // elements between the 16-byte boundary below x and x
    const unsigned max_off = (unsigned)((((size_t)shift0 + (size_t)(BC - 1) * ldbc + (size_t)vol) * es - 1) & ~(size_t)15);
    const int planes = BC * N0;
    if (elem_bytes == 2) {      // bf16 planes: the benchmark's 65 x 65 (FNOSeg / HNOSeg under autocast)
        const unsigned short *xal = (const unsigned short *)x - shift0;
        if (N1 == 65 && N2 == 65) return fwd_items_launch_t<65, 65, 7, unsigned short>(xal, workspace, a, shift0, max_off, planes, (unsigned)ldbc, s);
        return 0;
    }
    const float *xal = (const float *)x - shift0;
    // co-scheduling experiments (round 6): HNO_ITEM_FWD_WAVES=4 runs the 65 x 65 planes with four waves per workgroup (half the LDS ring: a
    // second kernel's workgroup fits beside it on the compute unit)
    static const int fw_env = getenv("HNO_ITEM_FWD_WAVES") ? atoi(getenv("HNO_ITEM_FWD_WAVES")) : 0;
    if (N1 == 65 && N2 == 65 && fw_env == 4) return fwd_items_launch_t<65, 65, 4>(xal, workspace, a, shift0, max_off, planes, (unsigned)ldbc, s);
#define X(n1, n2, wf, wi) if (N1 == n1 && N2 == n2) return fwd_items_launch_t<n1, n2, wf>(xal, workspace, a, shift0, max_off, planes, (unsigned)ldbc, s);
    HNO_ITEM_SIZES(X)
#undef X
    return 0;
}

template <int N1, int N2, int NWV, bool HAS_ADD, typename TO = float>
static int inv_items_launch_t(const float *ws, const float *add_al, TO *out_al, const DhtArgs &a, unsigned shift0, int items, unsigned ldbc,
                              hipStream_t s) {
    using G = ItemGeo<N1, N2>;
    const size_t lds = sizeof(float) * ((size_t)NWV * G::WSTRIDE + 256);
    static_assert(sizeof(float) * ((size_t)NWV * G::WSTRIDE + 256) <= 160 * 1024, "images exceed LDS");
    static int attr = -1;
    if (attr != current_device()) {
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)dht_inv_items_kernel<N1, N2, HAS_ADD, NWV, false, TO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax));
        HNO_CHECK_HIP(hipFuncSetAttribute((const void *)dht_inv_items_kernel<N1, N2, HAS_ADD, NWV, true, TO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax));
        attr = current_device();
    }
    // whole multiples of the items per plane per workgroup start: a wave's X must not depend on the workgroup (t_begin is arbitrary,
    // X = (t_begin + wave) % NP is taken per wave, so any split works)
    const int gw = items < 256 * NWV ? (items + NWV - 1) / NWV : 256;
    if (a.zl) hipLaunchKernelGGL((dht_inv_items_kernel<N1, N2, HAS_ADD, NWV, true, TO>), dim3(gw), dim3(64 * NWV), lds, s, ws, add_al, out_al, a, shift0, items / gw, items % gw, ldbc);
    else hipLaunchKernelGGL((dht_inv_items_kernel<N1, N2, HAS_ADD, NWV, false, TO>), dim3(gw), dim3(64 * NWV), lds, s, ws, add_al, out_al, a, shift0, items / gw, items % gw, ldbc);
    return 1;
}

int inv_items_launch(const void *workspace, const float *addend, void *out, const DhtArgs &a, int BC, long long ldbc, hipStream_t s, int elem_bytes) {
    const DhtPlan &p = a.p;
    const int N0 = p.ax[0].N, N1 = p.ax[1].N, N2 = p.ax[2].N;
    const size_t es = (size_t)elem_bytes;
    // the residual rows are DMA'd into an LDS image laid out at the output's 4-element phase: both tensors must share it
    const unsigned shift0 = (unsigned)(((size_t)out / es) & 3);
    if (p.ax[1].KT != 1 || p.ax[2].KT != 1 || p.ax[1].KmP != 16 || p.ax[2].KmP > 16 || (double)BC * ldbc >= 1.0e9 || ((size_t)out & (es - 1)) ||
        (addend && (((size_t)addend & 3) || (unsigned)(((size_t)addend >> 2) & 3) != shift0)))
        return 0;
    const float *add_al = addend ? addend - shift0 : nullptr;
    const int planes = BC * N0;
    if (elem_bytes == 2) {      // bf16 output planes: the benchmark's 65 x 65 (FNOSeg / HNOSeg under autocast)
        unsigned short *o16 = (unsigned short *)out - shift0;
        if (N1 == 65 && N2 == 65) {
            const int items = planes * ItemGeo<65, 65>::NP;
            return addend ? inv_items_launch_t<65, 65, 12, true, unsigned short>((const float *)workspace, add_al, o16, a, shift0, items, (unsigned)ldbc, s)
                          : inv_items_launch_t<65, 65, 12, false, unsigned short>((const float *)workspace, add_al, o16, a, shift0, items, (unsigned)ldbc, s);
        }
        return 0;
    }
    float *out_al = (float *)out - shift0;
    static const int iw_env = getenv("HNO_ITEM_INV_WAVES") ? atoi(getenv("HNO_ITEM_INV_WAVES")) : 0;      // (see fwd_items_launch)
    if (N1 == 65 && N2 == 65 && iw_env == 6) {
        const int items = planes * ItemGeo<65, 65>::NP;
        return addend ? inv_items_launch_t<65, 65, 6, true>((const float *)workspace, add_al, out_al, a, shift0, items, (unsigned)ldbc, s)
                      : inv_items_launch_t<65, 65, 6, false>((const float *)workspace, add_al, out_al, a, shift0, items, (unsigned)ldbc, s);
    }
#define X(n1, n2, wf, wi)                                                                                                          \
    if (N1 == n1 && N2 == n2) {                                                                                                    \
        const int items = planes * ItemGeo<n1, n2>::NP;                                                                            \
        return addend ? inv_items_launch_t<n1, n2, wi, true>((const float *)workspace, add_al, out_al, a, shift0, items, (unsigned)ldbc, s) \
                      : inv_items_launch_t<n1, n2, wi, false>((const float *)workspace, add_al, out_al, a, shift0, items, (unsigned)ldbc, s); \
    }
    HNO_ITEM_SIZES(X)
#undef X
    return 0;
}

}  // namespace hno
