// gemm_epilogue.h — fused epilogues shared by the two bf16 GEMM kernels (gemm.hip: 16-wave one-barrier-per-K-tile kernel,
// gemm8.hip: 8-wave 8-phase kernel).  Both hand over fp32 accumulators in the v_mfma_f32_16x16x32_bf16 C layout.
//
// Epilogues reproduce the reference's rounding points exactly: every nn.Linear output is rounded to bf16 before
// anything else touches it (model/modeling_llada.py:925-927, 741-744, 962-970).
#pragma once
#include "kernels.h"

namespace gemm_detail {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// RoPE rotation in fp32 with separately rounded products (the reference evaluates t*cos and rotate_half(t)*sin
// as two tensors and then adds them: model/modeling_llada.py:408-409) — no FMA contraction allowed.
MM_DEVICE void rope_pair(float t1, float t2, float c, float s, float& o1, float& o2) {
#pragma clang fp contract(off)
    float a = t1 * c;
    float b = t2 * s;
    o1 = a - b;
    float e = t2 * c;
    float f = t1 * s;
    o2 = e + f;
}

MM_DEVICE float silu_bf16(float g) {
    // F.silu on a bf16 tensor: evaluated in fp32, rounded to bf16 (model/modeling_llada.py:477-480)
    return bfround(g / (1.0f + expf(-g)));
}

// SiLU as a table (round 4).  The SwiGLU epilogue applies silu_bf16 to a value that has just been rounded to bf16: a function of
// 16 bits.  Evaluating it costs ~25 VALU instructions per output (expf, an IEEE division, a rounding), 80 outputs per lane and
// tile; the table costs 5 + one 2-byte LDS read.  Only the exponents that occur are tabulated: |x| in [2^-14, 2^5), both signs —
// 19 x 128 x 2 = 4 864 entries, padded to 5 120 (10 KiB, ten 1-KiB LDS-DMA pieces), which fits beside the K-tile buffers of every
// tile configuration.  Anything else (tiny, huge, Inf, NaN: a wave-uniform test over four outputs) takes the evaluating path.
// The table is filled ON THE DEVICE by silu_bf16 itself (gemm.hip: silu_lut_kernel), so both paths return the same bits by
// construction; tests/test_gpu_kernels.py::test_swiglu_table_is_bit_identical sweeps all 65 536 gate values through both.
struct SiluLut {
    static constexpr unsigned E0 = 113u << 7;       // bf16 bits (sign cleared) of 2^-14
    static constexpr unsigned NKEY = 19u * 128u;    // [2^-14, 2^5)
    static constexpr int ENTRIES = 5120, BYTES = ENTRIES * 2, PIECES = BYTES / 1024;
    // entry index of a bf16 value, or >= 2 * NKEY when it is not tabulated
    static MM_DEVICE unsigned key(unsigned bits) { return (bits & 0x7fffu) - E0; }
    static MM_DEVICE unsigned index(unsigned bits) { return (key(bits) << 1) | (bits >> 15); }
};
typedef __attribute__((address_space(3))) const uint16_t* lds_u16_ptr;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
MM_DEVICE uint32_t lds_u16(int byte_off) { return *(lds_u16_ptr)(uint32_t)byte_off; }   // LDS addresses are plain integers here (gemm8.hip)
#pragma clang diagnostic pop

// Tile sequence number -> (row tile, column tile): grouped order.  The sequence is cut into column BANDS of GN column tiles; inside
// a band into GROUPS of GM row tiles x GN column tiles (GM <= 0 or >= ntm: the whole band), column tile fastest.  Workgroups with
// neighbouring sequence numbers run on one XCD at the same time (xcd_remap), so a group is what shares A and W panels in that
// XCD's L2: per XCD-round of 32 tiles the fabric delivers gm A panels + gn W panels (gm * gn = 32).  Default GM = all rows,
// GN = 1024 / BN; mmada_set_option("gemm_tile_order", ...) picks other shapes for the FETCH_SIZE sweep.
MM_DEVICE void tile_coords_g(int t, int ntm, int ntn, int GM, int GN, int& mt, int& nt) {
    const int band = GN * ntm;
    const int b = t / band, rem = t - b * band;
    const int gn = min(GN, ntn - b * GN);
    if (GM <= 0 || GM >= ntm) {
        mt = rem / gn;
        nt = b * GN + (rem - mt * gn);
        return;
    }
    const int gsz = GM * gn;
    const int rg = rem / gsz, rr = rem - rg * gsz;
    const int r = rr / gn;
    mt = rg * GM + r;
    nt = b * GN + (rr - r * gn);
}
template <int GN = 4>
MM_DEVICE void tile_coords(int t, int ntm, int ntn, int& mt, int& nt) { tile_coords_g(t, ntm, ntn, 0, GN, mt, nt); }

// m_lim: first row this tile does NOT write (g.M, or the end of a short row tile of gemm8.hip).
// Wave grid WM x WN over the block tile, a wave owns TM x TN outputs = FM x FN fragments of 16 x 16:
//     acc[mi][ni][r] = D[m][n],  m = m0 + wm*TM + mi*16 + (lane>>4)*4 + r,  n = n0 + wn*TN + ni*16 + (lane&15)
template <int EPI, int TM, int TN, int WN>
MM_DEVICE void gemm_epilogue(const GemmArgs& g, int m0, int n0, f32x4 (&acc)[TM / 16][TN / 16], int wave, int lane, int m_lim) {
    constexpr int FM = TM / 16, FN = TN / 16;
    static_assert(TN % 32 == 0, "fused epilogues pair adjacent 16-column fragments");
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, fq = lane >> 4;
    // ---- epilogue: acc[mi][ni][r] = D[m][n], m = m0+wm*TM+mi*16+fq*4+r, n = n0+wn*TN+ni*16+frow ----
    const int mrow0 = m0 + wm * TM + fq * 4;
    const int wcol0 = n0 + wn * TN;  // first column of this wave (wave-uniform)

    if constexpr (EPI == EPI_STORE || EPI == EPI_RESID) {
#pragma unroll
        for (int mi = 0; mi < FM; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mrow0 + mi * 16 + r;
                if (m >= m_lim) continue;
                // wave-uniform: the 16 rows of a fragment share one residual owner
                const bool add = EPI == EPI_RESID && (g.resid_mod == 1 || ((m >> 4) % g.resid_mod) == g.resid_rank);
                size_t rrow = (size_t)m;  // residual row (compact -> full layout when a row window is active)
                if (EPI == EPI_RESID && g.rwin) {
                    const int bb = m / g.rwin;
                    rrow = (size_t)bb * g.rlp + g.rbeg + (m - bb * g.rwin);
                }
#pragma unroll
                for (int ni = 0; ni < FN; ++ni) {
                    const int n = wcol0 + ni * 16 + frow;
                    if (n >= g.N) continue;
                    float v = acc[mi][ni][r];
                    if constexpr (EPI == EPI_RESID) {
                        v = bfround(v);
                        if (add) v = bf2f(g.resid[rrow * g.ldr + n]) + v;
                    }
                    g.C[(size_t)m * g.ldc + n] = f2bf(v);
                }
            }
    } else if constexpr (EPI == EPI_SWIGLU) {
        // columns come in 32-wide groups: [16 x ff_proj | 16 x up_proj] (see pack_gate_up); x = silu(ff_proj)*up
        if (wcol0 < g.N) {
            const int hcol0 = wcol0 / 2 + frow;
#pragma unroll
            for (int mi = 0; mi < FM; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mrow0 + mi * 16 + r;
                    if (m >= m_lim) continue;
#pragma unroll
                    for (int q2 = 0; q2 < FN / 2; ++q2) {
                        if (wcol0 + q2 * 32 >= g.N) continue;
                        const float gate = bfround(acc[mi][2 * q2][r]);
                        const float up = bfround(acc[mi][2 * q2 + 1][r]);
                        g.C[(size_t)m * g.ldc + hcol0 + q2 * 16] = f2bf(silu_bf16(gate) * up);
                    }
                }
        }
    } else {  // EPI_QKV: a wave's TN columns lie inside one 128-wide head
        const int head = wcol0 >> 7, c0 = wcol0 & 127;
        if (head < g.Hq + g.Hkv) {
            const bool isq = head < g.Hq;
            bf16_t* dst = isq ? g.q : g.k;
            const int hh = isq ? head : head - g.Hq;
            const int nh = isq ? g.Hq : g.Hkv;
#pragma unroll
            for (int mi = 0; mi < FM; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mrow0 + mi * 16 + r;
                    if (m >= m_lim) continue;
                    const int mg = m + g.m_base;  // row of the whole [B*Lp] stream
                    const int b = mg / g.Lp;
                    int l = mg - b * g.Lp;        // rotary position
                    int lrow = l, lstride = g.Lkv;  // destination row / rows per head
                    if (g.pos_map) {
                        const int pos = g.pos_map[mg];
                        if (isq) {
                            lrow = l; lstride = g.Lq;
                            l = g.q_pos_shift >= 0 ? l + g.q_pos_shift : (pos < 0 ? 0 : pos);
                        } else {
                            if (pos < 0) continue;  // pad row of the compact stream: never enters the cache
                            l = lrow = pos;
                        }
                    }
                    bf16_t* row = dst + ((size_t)(b * nh + hh) * lstride + lrow) * 128;
#pragma unroll
                    for (int q2 = 0; q2 < FN / 2; ++q2) {
                        // permuted column layout: fragments (2*q2, 2*q2+1) hold rotary partners i and i+64
                        const int i = (c0 / 32 + q2) * 16 + frow;
                        const float t1 = bfround(acc[mi][2 * q2][r]);
                        const float t2 = bfround(acc[mi][2 * q2 + 1][r]);
                        const float c = g.rope_cos[l * 64 + i], s = g.rope_sin[l * 64 + i];
                        float o1, o2;
                        rope_pair(t1, t2, c, s, o1, o2);
                        row[i] = f2bf(o1);
                        row[i + 64] = f2bf(o2);
                    }
                }
        } else if (head < g.Hq + 2 * g.Hkv) {
            const int hv = head - g.Hq - g.Hkv;
#pragma unroll
            for (int mi = 0; mi < FM; ++mi) {
                const int mb = mrow0 + mi * 16;  // multiple of 4; Lp is a multiple of 8 -> 4 rows share a batch
                if (mb >= m_lim) continue;
                const int mbg = mb + g.m_base;  // m_base is a multiple of 8: the 4 rows still share a batch element
                const int b = mbg / g.Lp, l0 = mbg - b * g.Lp;
                if (g.pos_map) {  // scattered rows: one 2-byte store per (row, d); only the computed rows of a cache step
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (mb + r >= m_lim) continue;
                        const int pos = g.pos_map[mbg + r];
                        if (pos < 0) continue;
                        const size_t kp = (size_t)vt_key_pos(pos & ~3) + (pos & 3);
#pragma unroll
                        for (int ni = 0; ni < FN; ++ni) {
                            const int d = c0 + ni * 16 + frow;
                            g.vT[((size_t)(b * g.Hkv + hv) * 128 + d) * g.Lkv + kp] = f2bf(acc[mi][ni][r]);
                        }
                    }
                    continue;
                }
#pragma unroll
                for (int ni = 0; ni < FN; ++ni) {
                    const int d = c0 + ni * 16 + frow;
                    u32x2 pk;
                    pk[0] = pack_bf2(acc[mi][ni][0], acc[mi][ni][1]);
                    pk[1] = pack_bf2(acc[mi][ni][2], acc[mi][ni][3]);
                    *(u32x2*)(g.vT + ((size_t)(b * g.Hkv + hv) * 128 + d) * g.Lkv + vt_key_pos(l0)) = pk;
                }
            }
        }
    }
}


// The same epilogues for TRANSPOSED accumulators: the 8-phase kernel issues its MFMAs with the operand roles swapped
// (W fragment as the A operand, activation fragment as the B operand), which computes the transposed 16 x 16 block with the
// same products in the same k order:
//     acc[mi][ni][r] = D[m][n],  m = m0 + wm*TM + mi*16 + (lane&15),  n = n0 + wn*TN + ni*16 + (lane>>4)*4 + r
// — a lane now owns FOUR CONSECUTIVE COLUMNS of one row, and after one half-row exchange between lanes (swap_halves16)
// EIGHT: every residual read and output write of a lane is one 16-byte access instead of eight 2-byte ones (a 160 x 64
// wave tile: 20 instead of 160 store instructions per lane; the epilogue of a short-K launch such as attn_out was ~30 %
// of its time, round-3 profile).
// The V columns of the QKV projection are the exception: vT is K-major, so there the untransposed layout (four consecutive
// KEYS of one feature per lane) is the coalesced one — gemm8 picks the operand order per wave (qkv_wave_is_v).
MM_DEVICE bool qkv_wave_is_v(const GemmArgs& g, int wcol0) { return (wcol0 >> 7) >= g.Hq + g.Hkv; }

// Two adjacent 16-column fragments of a lane's row, each held as 4 consecutive columns per lane (2 packed dwords), become ONE
// run of 8 consecutive columns per lane: v_permlane16_swap exchanges lanes 16-31 / 48-63 of the first register with lanes
// 0-15 / 32-47 of the second, so afterwards (lq = lane >> 4)
//     lq 0: {a, b} = columns 0-7 of the FIRST fragment        lq 1: columns 0-7  of the SECOND fragment
//     lq 2:          columns 8-15 of the first                lq 3: columns 8-15 of the second
// — one 16-byte access per lane instead of two 8-byte ones (the store tail of a tile is issue-bound: guide T21).
MM_DEVICE void swap_halves16(uint32_t& a, uint32_t& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
// column (inside the 32-column pair) of the 8-column run a lane holds after swap_halves16
MM_DEVICE int run8_col(int lq) { return (lq & 1) * 16 + (lq >> 1) * 8; }

template <int EPI, int TM, int TN, int WN>
MM_DEVICE void gemm_epilogue_t(const GemmArgs& g, int m0, int n0, f32x4 (&acc)[TM / 16][TN / 16], int wave, int lane, int m_lim,
                                int lut_lds = -1 /* LDS byte address of the SiLU table, or -1 */) {
    constexpr int FM = TM / 16, FN = TN / 16;
    static_assert(TN % 32 == 0, "fused epilogues pair adjacent 16-column fragments");
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 15, lq = lane >> 4;
    const int mrow0 = m0 + wm * TM + lrow;
    const int wcol0 = n0 + wn * TN;  // first column of this wave (wave-uniform)

    if constexpr ((EPI == EPI_STORE || EPI == EPI_RESID) && FM <= 5) {
        // The residual (the other buffer of the x / y ping-pong: never aliases C) is read for up to FIVE fragment rows at a time, all
        // their 16-byte loads in flight together; hipcc otherwise orders every fragment row's loads behind the previous row's stores
        // with `s_waitcnt vmcnt(0)` (it cannot prove that resid and C do not alias, and vmcnt counts stores): FM serial round trips
        // in the tail of every tile, which a one-round launch (attn_out, down at batch 1) exposes in full.  (The 320-row tiles, FM = 10, keep the row-by-row form below:
        // two batches of five made their short-K tensor-parallel launches — where the epilogue is half of a tile's time — 9 % slower,
        // and their launches at TP = 1 have two or more rounds, whose tails overlap other workgroups' main loops anyway.)
        constexpr int RB = FM;
#pragma unroll
        for (int mb = 0; mb < FM; mb += RB) {
            u32x4 rres[EPI == EPI_RESID ? RB : 1][EPI == EPI_RESID ? FN / 2 : 1];
            bool radd[RB];
            if constexpr (EPI == EPI_RESID) {
#pragma unroll
                for (int mj = 0; mj < RB; ++mj) {
                    const int mi = mb + mj;
                    if (mi >= FM) continue;
                    const int m = mrow0 + mi * 16;
                    // wave-uniform: the 16 rows of a fragment share one residual owner
                    radd[mj] = g.resid_mod == 1 || ((m >> 4) % g.resid_mod) == g.resid_rank;
                    size_t rrow = (size_t)min(m, m_lim - 1);  // residual row (compact -> full layout when a row window is active)
                    if (g.rwin) {
                        const int bb = (int)rrow / g.rwin;
                        rrow = (size_t)bb * g.rlp + g.rbeg + ((int)rrow - bb * g.rwin);
                    }
#pragma unroll
                    for (int ni = 0; ni < FN; ni += 2) {   // dead rows / columns read a valid address (row m_lim - 1, column clamped) and are not stored
                        const int n = min(wcol0 + ni * 16 + run8_col(lq), g.N - 8);
                        // wave-uniform branch: under tensor parallelism only the owner rank of a fragment row adds (and reads) the residual
                        if (radd[mj]) rres[mj][ni / 2] = *(const u32x4*)(g.resid + rrow * g.ldr + n);
                        else rres[mj][ni / 2] = u32x4{0u, 0u, 0u, 0u};
                    }
                }
                // ONE place where the loads are waited for: behind this statement the values are the asm's, not the loads', so hipcc
                // puts no (conservative, store-counting) vmcnt wait in front of their uses in the branches below
#pragma unroll
                for (int mj = 0; mj < RB; ++mj)
#pragma unroll
                    for (int nj = 0; nj < FN / 2; ++nj)
                        if (mb + mj < FM) asm volatile("" : "+v"(rres[mj][nj]));
            }
#pragma unroll
            for (int mj = 0; mj < RB; ++mj) {
                const int mi = mb + mj;
                if (mi >= FM) continue;
                const int m = mrow0 + mi * 16;
                const bool live = m < m_lim;  // the swaps below are wave-wide: every lane takes part, dead rows only skip memory
                const bool add = EPI == EPI_RESID && radd[mj];
#pragma unroll
                for (int ni = 0; ni < FN; ni += 2) {
                    // every nn.Linear output is rounded to bf16 first (exactly representable afterwards: the exchange is lossless)
                    uint32_t a0 = pack_bf2(acc[mi][ni][0], acc[mi][ni][1]), a1 = pack_bf2(acc[mi][ni][2], acc[mi][ni][3]);
                    uint32_t b0 = pack_bf2(acc[mi][ni + 1][0], acc[mi][ni + 1][1]), b1 = pack_bf2(acc[mi][ni + 1][2], acc[mi][ni + 1][3]);
                    swap_halves16(a0, b0);
                    swap_halves16(a1, b1);
                    const int n = wcol0 + ni * 16 + run8_col(lq);
                    if (!live || n >= g.N) continue;  // N is a multiple of 8 here: a lane's eight columns are all inside or all outside
                    u32x4 o{a0, a1, b0, b1};
                    if constexpr (EPI == EPI_RESID) {
                        if (add) {
                            const u32x4 rv = rres[mj][ni / 2];
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                o[e] = pack_bf2(__uint_as_float(rv[e] << 16) + __uint_as_float(o[e] << 16),
                                                __uint_as_float(rv[e] & 0xffff0000u) + __uint_as_float(o[e] & 0xffff0000u));
                        }
                    }
                    *(u32x4*)(g.C + (size_t)m * g.ldc + n) = o;
                }
            }
        }
    } else if constexpr (EPI == EPI_STORE || EPI == EPI_RESID) {
#pragma unroll
        for (int mi = 0; mi < FM; ++mi) {
            const int m = mrow0 + mi * 16;
            const bool live = m < m_lim;  // the swaps below are wave-wide: every lane takes part, dead rows only skip memory
            // wave-uniform: the 16 rows of a fragment share one residual owner
            const bool add = EPI == EPI_RESID && (g.resid_mod == 1 || ((m >> 4) % g.resid_mod) == g.resid_rank);
            size_t rrow = (size_t)m;  // residual row (compact -> full layout when a row window is active)
            if (EPI == EPI_RESID && g.rwin) {
                const int bb = m / g.rwin;
                rrow = (size_t)bb * g.rlp + g.rbeg + (m - bb * g.rwin);
            }
#pragma unroll
            for (int ni = 0; ni < FN; ni += 2) {
                // every nn.Linear output is rounded to bf16 first (exactly representable afterwards: the exchange is lossless)
                uint32_t a0 = pack_bf2(acc[mi][ni][0], acc[mi][ni][1]), a1 = pack_bf2(acc[mi][ni][2], acc[mi][ni][3]);
                uint32_t b0 = pack_bf2(acc[mi][ni + 1][0], acc[mi][ni + 1][1]), b1 = pack_bf2(acc[mi][ni + 1][2], acc[mi][ni + 1][3]);
                swap_halves16(a0, b0);
                swap_halves16(a1, b1);
                const int n = wcol0 + ni * 16 + run8_col(lq);
                if (!live || n >= g.N) continue;  // N is a multiple of 8 here: a lane's eight columns are all inside or all outside
                u32x4 o{a0, a1, b0, b1};
                if constexpr (EPI == EPI_RESID) {
                    if (add) {
                        const u32x4 rv = *(const u32x4*)(g.resid + rrow * g.ldr + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            o[e] = pack_bf2(__uint_as_float(rv[e] << 16) + __uint_as_float(o[e] << 16),
                                            __uint_as_float(rv[e] & 0xffff0000u) + __uint_as_float(o[e] & 0xffff0000u));
                    }
                }
                *(u32x4*)(g.C + (size_t)m * g.ldc + n) = o;
            }
        }
    } else if constexpr (EPI == EPI_SWIGLU) {
        // columns come in 32-wide groups: [16 x ff_proj | 16 x up_proj] (see pack_gate_up); x = silu(ff_proj)*up.  A 32-column
        // group yields 16 output columns, 4 per lane; two adjacent groups are joined into runs of 8 like two fragments above.
        static_assert((FN / 2) % 2 == 0, "SwiGLU epilogue joins the outputs of two 32-column groups");
        const int hbase = wcol0 / 2;
#pragma unroll
        for (int mi = 0; mi < FM; ++mi) {
            const int m = mrow0 + mi * 16;
            const bool live = m < m_lim;
#pragma unroll
            for (int q2 = 0; q2 < FN / 2; q2 += 2) {
                uint32_t pk[2][2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float o[4];
                    uint32_t gb[4];
                    bool untabulated = false;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        gb[r] = f2bf(acc[mi][2 * (q2 + u)][r]);
                        untabulated |= SiluLut::key(gb[r]) >= SiluLut::NKEY;
                    }
                    if (lut_lds >= 0 && !__any(untabulated)) {   // wave-uniform: all 256 gate values of this step are tabulated
                        uint32_t sv[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) sv[r] = lds_u16(lut_lds + (int)(SiluLut::index(gb[r]) << 1));
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = __uint_as_float(sv[r] << 16) * bfround(acc[mi][2 * (q2 + u) + 1][r]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = silu_bf16(__uint_as_float(gb[r] << 16)) * bfround(acc[mi][2 * (q2 + u) + 1][r]);
                    }
                    pk[u][0] = pack_bf2(o[0], o[1]);
                    pk[u][1] = pack_bf2(o[2], o[3]);
                }
                swap_halves16(pk[0][0], pk[1][0]);
                swap_halves16(pk[0][1], pk[1][1]);
                if (!live || wcol0 + q2 * 32 >= g.N) continue;  // N is a multiple of 64: both groups are inside or outside
                *(u32x4*)(g.C + (size_t)m * g.ldc + hbase + q2 * 16 + run8_col(lq)) = u32x4{pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
            }
        }
    } else {  // EPI_QKV, q and k heads (a wave's TN columns lie inside one 128-wide head; V waves never come here)
        static_assert((FN / 2) % 2 == 0, "QKV epilogue joins the outputs of two rotary groups");
        const int head = wcol0 >> 7, c0 = wcol0 & 127;
        const bool isq = head < g.Hq;
        bf16_t* dst = isq ? g.q : g.k;
        const int hh = isq ? head : head - g.Hq;
        const int nh = isq ? g.Hq : g.Hkv;
#pragma unroll
        for (int mi = 0; mi < FM; ++mi) {
            const int m = mrow0 + mi * 16;
            bool live = m < m_lim;
            const int mg = min(m, m_lim - 1) + g.m_base;  // row of the whole [B*Lp] stream (dead lanes: any valid row)
            const int b = mg / g.Lp;
            int l = mg - b * g.Lp;        // rotary position
            int lr = l, lstride = g.Lkv;  // destination row / rows per head
            if (g.pos_map) {
                const int pos = g.pos_map[mg];
                if (isq) {
                    lr = l; lstride = g.Lq;
                    l = g.q_pos_shift >= 0 ? l + g.q_pos_shift : (pos < 0 ? 0 : pos);
                } else {
                    if (pos < 0) live = false;  // pad row of the compact stream: never enters the cache
                    l = lr = max(pos, 0);
                }
            }
            bf16_t* row = dst + ((size_t)(b * nh + hh) * lstride + lr) * 128;
#pragma unroll
            for (int q2 = 0; q2 < FN / 2; q2 += 2) {
                // permuted column layout: fragments (2*q, 2*q+1) hold rotary partners i and i+64, i = (c0/32 + q)*16 + lq*4 + r
                uint32_t p1[2][2], p2[2][2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = (c0 / 32 + q2 + u) * 16 + lq * 4;
                    const f32x4 cv = *(const f32x4*)(g.rope_cos + l * 64 + i);
                    const f32x4 sv = *(const f32x4*)(g.rope_sin + l * 64 + i);
                    float o1[4], o2[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        rope_pair(bfround(acc[mi][2 * (q2 + u)][r]), bfround(acc[mi][2 * (q2 + u) + 1][r]), cv[r], sv[r], o1[r], o2[r]);
                    p1[u][0] = pack_bf2(o1[0], o1[1]); p1[u][1] = pack_bf2(o1[2], o1[3]);
                    p2[u][0] = pack_bf2(o2[0], o2[1]); p2[u][1] = pack_bf2(o2[2], o2[3]);
                }
                swap_halves16(p1[0][0], p1[1][0]); swap_halves16(p1[0][1], p1[1][1]);
                swap_halves16(p2[0][0], p2[1][0]); swap_halves16(p2[0][1], p2[1][1]);
                if (!live) continue;
                const int i8 = (c0 / 32 + q2) * 16 + run8_col(lq);
                *(u32x4*)(row + i8) = u32x4{p1[0][0], p1[0][1], p1[1][0], p1[1][1]};
                *(u32x4*)(row + i8 + 64) = u32x4{p2[0][0], p2[0][1], p2[1][0], p2[1][1]};
            }
        }
    }
}

// every workgroup whose C is read by OTHER agents / other XCDs' kernels polling a counter (tensor-parallel partials,
// csrc/tp_comm.hip) ends with a system-scope release so that its stores have left this XCD's L2
MM_DEVICE void gemm_publish(const GemmArgs& g, int wave) {
    if (g.publish) {
        __syncthreads();
        if (wave == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
}

}  // namespace gemm_detail

// gemm8.hip: the 8-phase kernel.  cfg = one of the GEMM8_* configurations; returns nonzero on a launch error.
enum Gemm8Cfg { GEMM8_320x256 = 0, GEMM8_256x256 = 1, GEMM8_160x256 = 2, GEMM8_320x128 = 3, GEMM8_NCFG = 4 };
bool gemm8_supports(const GemmArgs& g);  // shape contract of the 8-phase kernel (K % 128 == 0, K >= 256, M, N % 8 == 0, ...)
int launch_gemm8(int epi, int cfg, const GemmArgs& g, hipStream_t s);
// Measurement hook (tools/gemm_sweep.py, tests): force one configuration for every following launch of this process.
//   -1: automatic (default);  0..3: GEMM8_* configuration;  1000 + BM: the 16-wave kernel with that row-tile height.
void gemm_force_config(int code);
int gemm_plan_code(int M, int N, int K);  // the planner's pick for a plain product: 0..3 or 1000 + BM; -1: unsupported shape
// short row tiles of the 320-row configurations (gemm8.hip) on / off; on by default
void gemm8_set_short_tiles(int on);
// tile order of the 8-phase kernel: 0 / -1 = default (bands of 1024 columns, all row tiles per group); GM * 100 + GN = groups of
// GM row tiles x GN column tiles (sweeps: tools/gemm_sweep.py --order)
void gemm8_set_tile_order(int code);
// SiLU table of the SwiGLU epilogue (8-phase kernel) on / off; on by default
void gemm_set_silu_lut(int on);
// allocate + fill the per-device constants of the GEMM launchers now (zero rows, SiLU table) instead of at the first launch
int gemm_prepare_device();
