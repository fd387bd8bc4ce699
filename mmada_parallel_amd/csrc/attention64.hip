// attention64.hip — unmasked, non-causal flash attention forward for gfx950, 64 query rows per wave, ONE wave per SIMD
// (round 4).  Same contract as attention.hip (F.scaled_dot_product_attention as the reference calls it,
// model/modeling_llada.py:672-679; layouts of q / k / vT as the QKV GEMM epilogue writes them) and the SAME arithmetic per
// 32-row q-block — MFMA shapes and k order, 64-key tiles, the per-q-block deferred-rescale decision, the order of every
// sum — so its output is bit-identical to attn_fwd_kernel / attn4p_fwd_kernel (tests/test_gpu_kernels.py).
//
// What changes is the traffic and the issue structure (the matrix blocks of the 4 x 32-row kernels queue at
// the LDS — per 32-cycle MFMA a wave read 1 KB of K or vT beside the LDS-DMA of TWO workgroups per CU):
//   * one workgroup = 4 waves = one per SIMD, each with the whole 512-register file and NQ q-blocks of 32 query rows
//     (NQ = 2: a "full pass" over 256 rows; NQ = 1: a "half pass" over 128 rows that balances the grid, attn64_plan).
//     A K / vT fragment read from the LDS feeds the MFMAs of both q-blocks, and the CU streams the keys ONCE: half the
//     ds_read_b128 and half the LDS-DMA bytes per MFMA.
//   * hipcc cannot allocate more than 256 registers to MFMA operands without copying accumulators through
//     v_accvgpr_read/write (tried: 451 spilled registers), so the accumulator file is owned by hand (CDNA guide §5.7):
//         a[0:127]    O accumulators   o[qb][db] = a[64 qb + 16 db : +15]
//         a[128:191]  Q fragments      q[qb][s]  = a[128 + 32 qb + 4 s : +3]
//     and every MFMA is an asm statement naming them; the scores, the soft-max and all addressing live in the (at most 256)
//     architectural VGPRs the compiler manages.  tools/isa_check.py requires: no spill, no scratch, no compiler-issued
//     v_accvgpr_* in this kernel.
//   * the issue stream is placed by hand, one MFMA per "slot" with at most ~5 single-issue fillers behind it, pinned with
//     sched_barrier(0):  per key tile t (scores of two tiles live: sc[X] = S(t), sc[X^1] = S(t+1))
//         barrier ; LDS-DMA of K(t+2), vT(t+1)
//         phase A   4 NQ MFMAs  P·V(t-1, key group 3)   | row sums of t-1 (second half), first K(t+1) fragment reads
//         phase B  16 NQ MFMAs  S(t+1) = K(t+1)·Q^T      | row maxima of t, rescale decision, exponentials of key groups 0, 1
//         phase C  12 NQ MFMAs  P·V(t, key groups 0-2)   | exponentials of key groups 2, 3, row sums (first half)
//     One barrier per key tile; K runs one tile ahead of vT in the LDS ring ([K0][K1][vT0][vT1], 64 KiB).
// Wait states the compiler cannot see (guide §5.7 item 2) are kept by construction: an operand a VALU wrote is consumed by
// an MFMA at least one slot later; S(t+1) is read by the VALU one barrier after its last MFMA; O is read (rescale,
// epilogue) behind explicit s_nop padding.
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "attention.h"

namespace {

using namespace attn_detail;

template <int... Is, class F>
MM_DEVICE void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
MM_DEVICE void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

#define ATTN64_AGPRS                                                                                                          \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18",   \
        "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", \
        "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", \
        "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", \
        "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", \
        "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102",     \
        "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", \
        "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", \
        "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", \
        "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", \
        "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", \
        "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191"

#ifdef MMADA_TUNE
// cycle stamps of the tile phases (VAR bit 7, tuning builds): [workgroup][wave][8] summed s_memtime deltas
__device__ unsigned long long g_attn64_stamps[1024 * 4 * 8];
#endif

constexpr int A_O = 0, A_Q = 128;
constexpr int a_o(int qb, int db) { return A_O + 64 * qb + 16 * db; }
constexpr int a_q(int qb, int s) { return A_Q + 32 * qb + 4 * s; }

// PAD: the statement opens with `s_nop 1`.  An operand register the COMPILER wrote just before the statement (a v_mov at a
// control-flow merge, a v_accvgpr_read restoring a fragment it had parked in a[192:255] under register pressure) needs two
// wait states before an MFMA may read it, and hipcc pads hazards only for instructions it can see (guide §5.7 item 2).
// Found the hard way: the last tile of odd tile counts restored vT fragments right in front of the P·V MFMAs of q-block 0.
// The pad costs an issue slot per MFMA, and a lone wave is issue-bound: the steady-state tiles run WITHOUT it and
// tools/attn64_audit.py proves on the compiled code that no VALU writes an MFMA operand within two wait states of the MFMA;
// prologue, last tile and drain (cold, more register pressure) keep it.
// sc = K-fragment · Q[a AQ..AQ+3]  (first k-step: the accumulator starts at 0 without being read)
template <int AQ, bool PAD>
MM_DEVICE void mfma_s_first(f32x16& c, const bf16x8& ka) {
    if constexpr (PAD) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(c) : "v"(ka), "i"(AQ), "i"(AQ + 3) : ATTN64_AGPRS);
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(c) : "v"(ka), "i"(AQ), "i"(AQ + 3) : ATTN64_AGPRS);
}
template <int AQ, bool PAD>
MM_DEVICE void mfma_s(f32x16& c, const bf16x8& ka) {
    if constexpr (PAD) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(c) : "v"(ka), "i"(AQ), "i"(AQ + 3) : ATTN64_AGPRS);
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(c) : "v"(ka), "i"(AQ), "i"(AQ + 3) : ATTN64_AGPRS);
}
// O[a AO..AO+15] += vT-fragment · P-fragment
template <int AO, bool PAD>
MM_DEVICE void mfma_pv(const bf16x8& va, const bf16x8& pb) {
    if constexpr (PAD) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(va), "v"(pb), "i"(AO), "i"(AO + 15) : ATTN64_AGPRS);
    else asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(va), "v"(pb), "i"(AO), "i"(AO + 15) : ATTN64_AGPRS);
}
// diagnostic builds: stand-ins that keep the operands of a removed MFMA alive
MM_DEVICE void diag_touch(f32x16& c, const bf16x8& x) { asm volatile("" : "+v"(c) : "v"(x)); }
MM_DEVICE void diag_use(const bf16x8& x, const bf16x8& y) { asm volatile("" ::"v"(x), "v"(y)); }
template <int A>
MM_DEVICE void acc_write(uint32_t v) {
    asm volatile("v_accvgpr_write_b32 a[%c1], %0" ::"v"(v), "i"(A) : ATTN64_AGPRS);
}
template <int A>
MM_DEVICE float acc_read() {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(v) : "i"(A) : ATTN64_AGPRS);
    return v;
}
template <int A>
MM_DEVICE void acc_scale(float alpha) {
    float t;
    asm volatile("v_accvgpr_read_b32 %0, a[%c2]\n\ts_nop 0\n\tv_mul_f32 %0, %1, %0\n\ts_nop 0\n\tv_accvgpr_write_b32 a[%c2], %0"
                 : "=&v"(t)
                 : "v"(alpha), "i"(A)
                 : ATTN64_AGPRS);
}

// ---- the hand-placed schedule of one key tile, as compile-time tables -------------------------------------------------
// Unified MFMA slot u of a tile: phase A = [0, NA), phase B = [NA, NA + NBS), phase C = [NA + NBS, NT).
template <int NQ>
struct Sched64 {
    static constexpr int NA = 4 * NQ, NBS = 16 * NQ, NC = 12 * NQ, NT = NA + NBS + NC;
    static constexpr int NMAX = 6 * NQ;                 // slots of phase B that carry the row maxima + the decision
    static constexpr int NEXP = NBS - NMAX;             // slots of phase B behind the decision
    static constexpr int UNITS = 16 * NQ;               // exponential units (two scores each, per q-block) of a tile
    // LDS-DMA: piece p (0-3: K, 4-7: vT) is issued in unified slot 1 + p * DMA_STEP (spread out: an LDS-DMA instruction
    // holds the issuing wave for 60-180 cycles, guide constants, and one wave per SIMD has nobody to cover for it)
    static constexpr int DMA_STEP = NQ == 2 ? 5 : 3;
    // exponential pipeline, three stages one slot apart (a lone wave pays every VALU dependency in full):
    //   stage 1 (v_fma: s * scale - m) of unit U in E-slot pos1(U), stage 2 (v_exp) one E-slot later, stage 3 (v_cvt_pk)
    //   two later.  E-slots: e in [0, NEXP) = phase B behind the decision, e >= NEXP = phase C.
    // units [0, 8 NQ) = key groups 0, 1 (four in every five E-slots of phase B); units [8 NQ, 16 NQ) = key groups 2, 3
    static constexpr int pos1(int U) { return U < 8 * NQ ? U + U / 4 : NEXP + (U - 8 * NQ); }
    static constexpr int unit_at(int e) {               // the unit whose stage 1 sits in E-slot e, or -1
        for (int U = 0; U < UNITS; ++U)
            if (pos1(U) == e) return U;
        return -1;
    }
};

// bid: index inside this pass kind's part of the grid; per_pair: passes of this kind per (batch, head); qb_first: first
// q-block (32 rows) this kind covers.  VAR: 0 in the product; tuning builds (-DMMADA_TUNE, tools/attn_sweep.py) instantiate
// DIAGNOSTIC variants (wrong results, timing only): bit 0 no LDS-DMA in the tile loop, bit 1 no exponential pipeline,
// bit 2 no fragment reads in the tile loop, bit 3 no MFMAs.
template <int NQ, int VAR>
MM_DEVICE void attn64_body(const AttnArgs& a, int bid, int per_pair, int qb_first) {
    using SC = Sched64<NQ>;
    constexpr int NA = SC::NA, NBS = SC::NBS, NC = SC::NC, NMAX = SC::NMAX, NEXP = SC::NEXP;
    constexpr int KBASE0 = 0, VBASE0 = 2 * TILE_BYTES;
    constexpr bool NO_DMA = (VAR & 1) != 0, NO_EXP = (VAR & 2) != 0, NO_READ = (VAR & 4) != 0, NO_MFMA = (VAR & 8) != 0;
    // A/B switches (correct results): 16 LDS-DMA pieces all in phase A, 32 exponentials unpipelined
    constexpr bool DMA_EARLY = (VAR & 16) != 0, EXP_FLAT = (VAR & 32) != 0;
    constexpr bool STAMP = (VAR & 128) != 0;   // s_memtime at the phase boundaries of every tile, summed per wave
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0;
    auto stamp = [&](auto k_) {
        if constexpr (STAMP) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            st_acc[decltype(k_)::value] += now - st_prev;
            st_prev = now;
        }
    };
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hi = lane >> 5;
    // workgroup -> (batch, head, first q-block): XCD x takes whole (batch, head) pairs so a head's K / vT stays in one L2
    int pair, pi;
    if (a.xcd_pairs > 0) {
        const int xcd = bid & 7, i = bid >> 3;
        pair = xcd * a.xcd_pairs + i / per_pair;
        pi = i - (i / per_pair) * per_pair;
    } else {
        pair = bid / per_pair;
        pi = bid - pair * per_pair;
    }
    const int b = pair / a.Hq, h = pair - b * a.Hq;
    const int hkv = h / (a.Hq / a.Hkv);
    const bf16_t* Qp = a.q + (size_t)(b * a.Hq + h) * a.Lq_alloc * 128;
    const bf16_t* Kp = a.k + (size_t)(b * a.Hkv + hkv) * a.Lkv * 128;
    const bf16_t* Vp = a.vT + (size_t)(b * a.Hkv + hkv) * 128 * a.Lkv;
    const int qb0 = qb_first + pi * 4 * NQ + wave * NQ;  // q-block qb of this wave: rows q_begin + 32 (qb0 + qb) ..+32

    // ---- the accumulator file: O = 0, Q fragments ----
    asm volatile("s_nop 0" ::: ATTN64_AGPRS);  // a[0:191] belong to this kernel's asm statements from here on
    static_for<64 * NQ>([&](auto i) { acc_write<A_O + decltype(i)::value>(0u); });
    static_for<NQ>([&](auto qb_) {
        constexpr int qb = decltype(qb_)::value;
        const int q_ld = min(a.q_begin + (qb0 + qb) * 32 + ql, a.Lq_alloc - 1);
        static_for<8>([&](auto s_) {
            constexpr int s = decltype(s_)::value;
            const u32x4 w = *(const u32x4*)(Qp + (size_t)q_ld * 128 + s * 16 + hi * 8);
            acc_write<a_q(qb, s) + 0>(w[0]);
            acc_write<a_q(qb, s) + 1>(w[1]);
            acc_write<a_q(qb, s) + 2>(w[2]);
            acc_write<a_q(qb, s) + 3>(w[3]);
        });
    });
    float m_run[NQ], l_run[NQ], psum[NQ];
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) { m_run[qb] = -1e30f; l_run[qb] = 0.f; psum[qb] = 0.f; }

    // ---- LDS-DMA sources (the piece split of attention.hip: wave w moves K rows 16w..16w+15 and vT rows 32w..32w+31).
    // Piece i of a wave: K rows 16w + 4i + (lane >> 4), vT rows 32w + 8i + (lane >> 3).  The wave-uniform part of the source
    // offset goes into the scalar base; the lane part differs between a wave's four pieces only in the swizzle term, an
    // XOR with i << 6 (K) / (i & 1) << 6 (vT): ONE register per operand instead of four.
    unsigned klane, vlane;
    {
        const int r4 = lane >> 4, c = lane & 15;
        klane = (unsigned)(r4 * 256 + ((c ^ r4) << 4));
        const int r8 = lane >> 3;
        vlane = (unsigned)(r8 * a.Lkv * 2 + (((lane & 7) ^ (r8 >> 1)) << 4));
    }
    const int nkt = (a.L + KB - 1) / KB;
    // one 1-KiB piece of a K / vT tile into ring slot SLOT; a tile index past the end re-reads the last tile into a slot
    // nobody reads any more (harmless, at most two tiles per pass)
    auto dma_k = [&](auto slot_, auto i_, int kt) {
        constexpr int SLOT = decltype(slot_)::value, I = decltype(i_)::value;
        const char* kb = (const char*)Kp + (size_t)min(kt, nkt - 1) * KB * 256 + (size_t)(wave * 16 + I * 4) * 256;
        unsigned off = klane ^ (unsigned)(I << 6);
        dma16(kb, off, KBASE0 + SLOT * TILE_BYTES + wave * 4096 + I * 1024);
    };
    auto dma_v = [&](auto slot_, auto i_, int kt) {
        constexpr int SLOT = decltype(slot_)::value, I = decltype(i_)::value;
        const char* vb = (const char*)Vp + (size_t)min(kt, nkt - 1) * KB * 2 + (size_t)(wave * 32 + I * 8) * a.Lkv * 2;
        unsigned off = vlane ^ (unsigned)((I & 1) << 6);
        dma16(vb, off, VBASE0 + SLOT * TILE_BYTES + wave * 4096 + I * 1024);
    };
    int kro[8], vro[4];
    {
        const int ksw = ql & 15, vsw = (ql >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 8; ++s) kro[s] = KBASE0 + ql * 256 + (((2 * s + hi) ^ ksw) << 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) vro[j] = VBASE0 + ql * 128 + (((2 * j + hi) ^ vsw) << 4);
    }
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    f32x16 sc[2][NQ][2];   // scores / probabilities of two key tiles: [buffer][q-block][keys 0-31 | 32-63]
    // fragment registers.  NQ = 2 (128 score registers): K fragments of three k-steps, vT fragments of two key groups, each
    // requested right behind the first MFMA of the step / group before (every wait the compiler places in front of an asm
    // MFMA is lgkmcnt(0): the youngest request must be old by then).  NQ = 1 has the registers to hold a WHOLE K tile and
    // a whole vT tile (guide T16): K(kt + 1) is read in phase A, vT(kt) under the row maxima — one LDS wait per tile; with
    // the NQ = 2 rings a half pass, whose k-steps are only two MFMAs long, stalled at every step.
    constexpr int KR = NQ == 1 ? 8 : 3, VRG = NQ == 1 ? 4 : 2;
    bf16x8 ka[KR][2];      // K fragments of k-step s in ka[s % KR]
    bf16x8 va[VRG][4];     // vT fragments of key group j in va[j % VRG]
    bf16x8 pb[3][NQ];      // P fragments of key group j in pb[j % 3]: three groups are live at a time
    float mxa[NQ], mxb[NQ], mx[NQ];
    float et[3][2];        // the exponential pipeline: [stage slot][element] (fma result, then exp result)

    auto read_k = [&](auto slot_, auto s_) {
        constexpr int BASE = decltype(slot_)::value * TILE_BYTES, s = decltype(s_)::value;
        if constexpr (!NO_READ) {
            ka[s % KR][0] = lds_frag(kro[s] + BASE);
            ka[s % KR][1] = lds_frag(kro[s] + BASE + 8192);
        }
    };
    auto read_v_group = [&](auto slot_, auto j_) {
        constexpr int BASE = decltype(slot_)::value * TILE_BYTES, j = decltype(j_)::value;
        if constexpr (!NO_READ) static_for<4>([&](auto db_) { va[j % VRG][decltype(db_)::value] = lds_frag(vro[j] + BASE + decltype(db_)::value * 4096); });
    };
    // S MFMA number i of a tile: k-step s, q-block qb, key half hh (a K fragment feeds the q-blocks back to back)
    auto s_mfma = [&](auto x_, auto i_, auto pad_) {
        constexpr int X = decltype(x_)::value, i = decltype(i_)::value;
        constexpr bool PAD = decltype(pad_)::value;
        constexpr int s = i / (2 * NQ), qb = (i / 2) % NQ, hh = i & 1;
        if constexpr (NO_MFMA) diag_touch(sc[X][qb][hh], ka[s % KR][hh]);
        else if constexpr (s == 0) mfma_s_first<a_q(qb, s), PAD>(sc[X][qb][hh], ka[s % KR][hh]);
        else mfma_s<a_q(qb, s), PAD>(sc[X][qb][hh], ka[s % KR][hh]);
    };
    // P·V MFMA number i of key group j: vT fragment db, q-block qb
    auto pv_mfma = [&](auto j_, auto i_, auto pad_) {
        constexpr int j = decltype(j_)::value, i = decltype(i_)::value, db = i / NQ, qb = i % NQ;
        if constexpr (NO_MFMA) diag_use(va[j % VRG][db], pb[j % 3][qb]);
        else mfma_pv<a_o(qb, db), decltype(pad_)::value>(va[j % VRG][db], pb[j % 3][qb]);
    };
    // row-maximum operation number op (0..19) of q-block qb on the scores in sc[X] — the chains of attn_fwd_kernel
    auto max_op = [&](auto x_, auto qb_, auto op_) {
        constexpr int X = decltype(x_)::value, qb = decltype(qb_)::value, op = decltype(op_)::value;
        const f32x16& s0 = sc[X][qb][0];
        const f32x16& s1 = sc[X][qb][1];
        if constexpr (op == 0) mxa[qb] = fmax_nc(s0[0], s1[0]);
        else if constexpr (op == 1) mxb[qb] = fmax_nc(s0[1], s1[1]);
        else if constexpr (op < 16) {
            constexpr int r = (op / 2) * 2;  // op 2,3 -> r = 2; ... op 14,15 -> r = 14
            if constexpr ((op & 1) == 0) mxa[qb] = max3f(mxa[qb], s0[r], s1[r]);
            else mxb[qb] = max3f(mxb[qb], s0[r + 1], s1[r + 1]);
        } else if constexpr (op == 16) mx[qb] = fmax_nc(mxa[qb], mxb[qb]);
        else if constexpr (op == 17) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx[qb]), __float_as_uint(mx[qb]), false, false);
            mxa[qb] = __uint_as_float(sw[0]);
            mxb[qb] = __uint_as_float(sw[1]);
        } else if constexpr (op == 18) mx[qb] = fmax_nc(mxa[qb], mxb[qb]);
        else mx[qb] = mx[qb] * a.scale_log2e;
    };
    // E-slot e of the exponential pipeline on the tile in sc[X]: stage 3 of the unit that started two slots ago, stage 2 of
    // the one that started one slot ago, stage 1 of the one that starts here.  Unit U: key group j = U / (4 NQ), q-block
    // qb = (U / 4) % NQ, element pair p = U % 4; scores sc[X][qb][j >> 1][8 (j & 1) + 2 p (+1)].
    // exp2(s * scale_log2e - m) is attn_fwd_kernel's expression (one v_fma, one v_exp); the stages only pull it apart.
    auto exp_slot = [&](auto x_, auto e_) {
        constexpr int X = decltype(x_)::value, e = decltype(e_)::value;
        if constexpr (NO_EXP) return;
        if constexpr (EXP_FLAT) {
            constexpr int U = SC::unit_at(e);
            if constexpr (U >= 0) {
                constexpr int j = U / (4 * NQ), qb = (U / 4) % NQ, p = U % 4, r = 8 * (j & 1) + 2 * p;
                const float v0 = __builtin_amdgcn_exp2f(sc[X][qb][j >> 1][r] * a.scale_log2e - m_run[qb]);
                const float v1 = __builtin_amdgcn_exp2f(sc[X][qb][j >> 1][r + 1] * a.scale_log2e - m_run[qb]);
                sc[X][qb][j >> 1][r] = v0;
                sc[X][qb][j >> 1][r + 1] = v1;
                pb[j % 3][qb][2 * p] = (__bf16)v0;
                pb[j % 3][qb][2 * p + 1] = (__bf16)v1;
            }
            return;
        }
        constexpr int U3 = e >= 2 ? SC::unit_at(e - 2) : -1, U2 = e >= 1 ? SC::unit_at(e - 1) : -1, U1 = SC::unit_at(e);
        if constexpr (U3 >= 0) {
            constexpr int j = U3 / (4 * NQ), qb = (U3 / 4) % NQ, p = U3 % 4;
            pb[j % 3][qb][2 * p] = (__bf16)et[(e - 2) % 3][0];
            pb[j % 3][qb][2 * p + 1] = (__bf16)et[(e - 2) % 3][1];
        }
        if constexpr (U2 >= 0) {
            constexpr int j = U2 / (4 * NQ), qb = (U2 / 4) % NQ, p = U2 % 4, r = 8 * (j & 1) + 2 * p;
            const float v0 = __builtin_amdgcn_exp2f(et[(e - 1) % 3][0]), v1 = __builtin_amdgcn_exp2f(et[(e - 1) % 3][1]);
            et[(e - 1) % 3][0] = v0;
            et[(e - 1) % 3][1] = v1;
            sc[X][qb][j >> 1][r] = v0;
            sc[X][qb][j >> 1][r + 1] = v1;
        }
        if constexpr (U1 >= 0) {
            constexpr int j = U1 / (4 * NQ), qb = (U1 / 4) % NQ, p = U1 % 4, r = 8 * (j & 1) + 2 * p;
            et[e % 3][0] = sc[X][qb][j >> 1][r] * a.scale_log2e - m_run[qb];
            et[e % 3][1] = sc[X][qb][j >> 1][r + 1] * a.scale_log2e - m_run[qb];
        }
    };
    // attn_fwd_kernel's row sum: psum += s0[r] + s1[r] for r = 0 .. 15 in this order
    auto sum_r = [&](auto x_, auto qb_, auto r_) {
        constexpr int X = decltype(x_)::value, qb = decltype(qb_)::value, r = decltype(r_)::value;
        psum[qb] += sc[X][qb][0][r] + sc[X][qb][1][r];
    };

    // One key tile kt.  Entering: sc[X] = S(kt); K(kt + 1) in K slot X ^ 1 and vT(kt) in vT slot X requested one iteration
    // ago; for kt > 0 the P·V MFMAs of (kt - 1, key group 3) are pending with pb[0] and va[3 % VRG] in registers and the row sums
    // of kt - 1 half done.  LAST: no S(kt + 1), no LDS-DMA.
    auto tile = [&](auto x_, auto last_, int kt) {
        constexpr int X = decltype(x_)::value;
        constexpr bool LAST = decltype(last_)::value;
        using XC = std::integral_constant<int, X>;
        using XN = std::integral_constant<int, X ^ 1>;
        using PADC = std::integral_constant<bool, LAST>;   // the hazard pad of the MFMA statements: cold tiles only
        asm volatile("" : "+v"(kro[0]), "+v"(kro[1]), "+v"(kro[2]), "+v"(kro[3]), "+v"(kro[4]), "+v"(kro[5]), "+v"(kro[6]), "+v"(kro[7]));
        asm volatile("" : "+v"(vro[0]), "+v"(vro[1]), "+v"(vro[2]), "+v"(vro[3]));
        stamp(std::integral_constant<int, 5>{});   // phase C of the previous tile
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        A8_SB();
        stamp(std::integral_constant<int, 6>{});   // waiting for the LDS-DMA / fragment reads
        asm volatile("s_barrier" ::: "memory");
        A8_SB();
        stamp(std::integral_constant<int, 7>{});   // waiting at the barrier
        // what every unified slot u carries besides its MFMA and its soft-max share: LDS-DMA pieces (K(kt + 2) -> K slot X,
        // vT(kt + 1) -> vT slot X ^ 1)
        auto dma_slot = [&](auto u_) {
            constexpr int u = decltype(u_)::value;
            if constexpr (DMA_EARLY) {
                if constexpr (!LAST && !NO_DMA && u < NA) {
                    static_for<8 / NA>([&](auto k_) {
                        constexpr int p = u * (8 / NA) + decltype(k_)::value;
                        if constexpr (p < 4) dma_k(XC{}, std::integral_constant<int, p>{}, kt + 2);
                        else dma_v(XN{}, std::integral_constant<int, p - 4>{}, kt + 1);
                    });
                }
            } else if constexpr (!LAST && !NO_DMA && u >= 1 && (u - 1) % SC::DMA_STEP == 0 && (u - 1) / SC::DMA_STEP < 8) {
                constexpr int p = (u - 1) / SC::DMA_STEP;
                if constexpr (p < 4) dma_k(XC{}, std::integral_constant<int, p>{}, kt + 2);
                else dma_v(XN{}, std::integral_constant<int, p - 4>{}, kt + 1);
            }
        };
        // ---- phase A: P·V(kt - 1, key group 3) | second half of the row sums of kt - 1 | K(kt + 1) fragments of k-steps 0, 1
        auto phase_a_rest = [&](auto i_) {
            constexpr int i = decltype(i_)::value;
            dma_slot(i_);
            if constexpr (!LAST) {
                if constexpr (NQ == 1) {       // the whole K(kt + 1) tile: two k-steps per slot
                    read_k(XN{}, std::integral_constant<int, 2 * i>{});
                    read_k(XN{}, std::integral_constant<int, 2 * i + 1>{});
                } else if constexpr (i == 0 || i == 1) {
                    read_k(XN{}, i_);
                }
            }
        };
        if (kt > 0) {
            static_for<NA>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                pv_mfma(std::integral_constant<int, 3>{}, i_, PADC{});
                static_for<8 / NA>([&](auto k_) {   // NQ = 2: one r per slot; NQ = 1: two
                    constexpr int r = 8 + i * (8 / NA) + decltype(k_)::value;
                    static_for<NQ>([&](auto qb_) { sum_r(XN{}, qb_, std::integral_constant<int, r>{}); });
                });
                phase_a_rest(i_);
                A8_SB();
            });
#pragma unroll
            for (int qb = 0; qb < NQ; ++qb) { l_run[qb] += psum[qb]; psum[qb] = 0.f; }
        } else {
            static_for<NA>([&](auto i_) { phase_a_rest(i_); A8_SB(); });
        }
        stamp(std::integral_constant<int, 0>{});   // phase A
        if constexpr (LAST) {
            if (kt * KB + KB > a.L) {   // keys past L (only in the last tile) get -inf; select, not arithmetic
                const int kbase = kt * KB + 4 * hi;
#pragma unroll
                for (int qb = 0; qb < NQ; ++qb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kbase + (r & 3) + 8 * (r >> 2);
                        if (key >= a.L) sc[X][qb][0][r] = -INFINITY;
                        if (key + 32 >= a.L) sc[X][qb][1][r] = -INFINITY;
                    }
            }
        }
        A8_SB();
        // S MFMA i of phase B and the K fragment reads behind it: the fragments of k-step s + 2 are requested right behind
        // the FIRST MFMA of k-step s (whose operands the compiler has just waited for with lgkmcnt(0): the next wait is a
        // whole k-step away)
        auto s_slot = [&](auto i_) {
            constexpr int i = decltype(i_)::value, s = i / (2 * NQ);
            if constexpr (!LAST) {
                s_mfma(XN{}, i_, PADC{});
                if constexpr (NQ == 2 && i % (2 * NQ) == 0 && s + 2 < 8) read_k(XN{}, std::integral_constant<int, s + 2>{});
            }
            dma_slot(std::integral_constant<int, NA + i>{});
        };
        // ---- phase B, first part: S(kt + 1) | row maxima of kt ----
        constexpr int OPS = 20 / (NMAX - NQ);      // maximum operations per q-block and slot: NQ = 2: 2, NQ = 1: 4
        int okm = 0;                               // bit qb: q-block qb keeps its running maximum
        static_for<NMAX>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            s_slot(i_);
            if constexpr (NQ == 1 && i % 2 == 0) read_v_group(XC{}, std::integral_constant<int, i / 2>{});   // vT(kt), key groups 0-2
            static_for<OPS>([&](auto k_) {
                constexpr int op = i * OPS + decltype(k_)::value;
                if constexpr (op < 20) static_for<NQ>([&](auto qb_) { max_op(XC{}, qb_, std::integral_constant<int, op>{}); });
            });
            // the rescale test of attn_fwd_kernel, taken one slot before the branch that uses it (a lone wave would sit out
            // the VALU -> SALU latency of v_cmp / s_cmp right in front of the branch)
            if constexpr (i == NMAX - NQ) {
                static_for<NQ>([&](auto qb_) {
                    constexpr int qb = decltype(qb_)::value;
                    okm |= __builtin_amdgcn_readfirstlane(__all(mx[qb] - m_run[qb] <= DEFER_LOG2) ? 1 : 0) << qb;
                });
                asm volatile("" : "+s"(okm));
            }
            A8_SB();
        });
        stamp(std::integral_constant<int, 1>{});   // phase B, maxima
        // the deferred rescale of attn_fwd_kernel, per q-block (wave-uniform; rare once the running maxima have settled):
        // ONE branch for the q-blocks of the wave
        if (okm != (1 << NQ) - 1) {
#pragma unroll
            for (int qb = 0; qb < NQ; ++qb) {
                if (!((okm >> qb) & 1)) {
                    const float m_new = fmax_nc(m_run[qb], mx[qb]);
                    const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
                    l_run[qb] *= alpha;
                    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last P·V MFMAs have long retired; belt and braces
                    if (qb == 0) static_for<64>([&](auto i) { acc_scale<A_O + decltype(i)::value>(alpha); });
                    else static_for<64>([&](auto i) { acc_scale<A_O + 64 + decltype(i)::value>(alpha); });
                    asm volatile("s_nop 1" ::: "memory");
                    m_run[qb] = m_new;
                }
            }
        }
        A8_SB();
        stamp(std::integral_constant<int, 2>{});   // decision (+ rescale)
        // ---- phase B, second part: S(kt + 1) | exponentials of key groups 0 and 1 | vT fragments of key group 0 ----
        static_for<NEXP>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            s_slot(std::integral_constant<int, NMAX + i>{});
            exp_slot(XC{}, i_);
            if constexpr (NQ == 2 && i == NEXP - 4 * NQ) read_v_group(XC{}, I0{});
            if constexpr (NQ == 1 && i == 0) read_v_group(XC{}, std::integral_constant<int, 3>{});
            A8_SB();
        });
        stamp(std::integral_constant<int, 3>{});   // phase B, exponentials
        // ---- phase C: P·V(kt, key groups 0-2) | exponentials of key groups 2 and 3 | first half of the row sums ----
        static_for<NC>([&](auto i_) {
            constexpr int i = decltype(i_)::value, j = i / (4 * NQ), w = i % (4 * NQ);
            pv_mfma(std::integral_constant<int, j>{}, std::integral_constant<int, w>{}, PADC{});
            if constexpr (NQ == 2 && w == 0) read_v_group(XC{}, std::integral_constant<int, j + 1>{});   // behind the group's first MFMA
            exp_slot(XC{}, std::integral_constant<int, NEXP + i>{});
            if constexpr (i >= 8 * NQ + 2) {        // the exponentials end in slot 8 NQ + 1
                constexpr int n = 4 * NQ - 2, k0 = i - (8 * NQ + 2);     // 8 r values over n slots
                static_for<NQ>([&](auto qb_) {
                    static_for<8>([&](auto r_) {
                        constexpr int r = decltype(r_)::value;
                        if constexpr (r * n / 8 == k0) sum_r(XC{}, qb_, r_);
                    });
                });
            }
            dma_slot(std::integral_constant<int, NA + NBS + i>{});
            A8_SB();
        });
    };

    // ---- prologue: K(0), vT(0), K(1); S(0) ----
    static_for<4>([&](auto i_) { dma_k(I0{}, i_, 0); });
    static_for<4>([&](auto i_) { dma_v(I0{}, i_, 0); });
    static_for<4>([&](auto i_) { dma_k(I1{}, i_, 1); });
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    A8_SB();
    {
        auto rk = [&](auto s_) {   // prologue reads are never ablated
            constexpr int s = decltype(s_)::value;
            ka[s % KR][0] = lds_frag(kro[s] + KBASE0);
            ka[s % KR][1] = lds_frag(kro[s] + KBASE0 + 8192);
        };
        if constexpr (NQ == 1) static_for<8>([&](auto s_) { rk(s_); });
        else { rk(I0{}); rk(I1{}); }
        static_for<NBS>([&](auto i_) {
            constexpr int i = decltype(i_)::value, s = i / (2 * NQ);
            s_mfma(I0{}, i_, std::true_type{});
            if constexpr (NQ == 2 && i % (2 * NQ) == 0 && s + 2 < 8) rk(std::integral_constant<int, s + 2>{});
            A8_SB();
        });
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // S(0) is read by the VALU right behind the next barrier
    A8_SB();
    if constexpr (STAMP) st_prev = __builtin_amdgcn_s_memtime();
    using F = std::false_type;
    using T = std::true_type;
    int kt = 0;
    for (; kt + 2 < nkt; kt += 2) {
        tile(I0{}, F{}, kt);
        tile(I1{}, F{}, kt + 1);
    }
    bool odd;   // parity of the last tile's score buffer
    if (kt + 1 < nkt) {
        tile(I0{}, F{}, kt);
        tile(I1{}, T{}, kt + 1);
        odd = true;
    } else {
        tile(I0{}, T{}, kt);
        odd = false;
    }
#ifdef MMADA_TUNE
    if constexpr (STAMP) {
        if (lane == 0 && blockIdx.x < 1024)
            for (int i = 0; i < 8; ++i) g_attn64_stamps[(blockIdx.x * 4 + wave) * 8 + i] = st_acc[i];
    }
#endif
    // ---- drain: P·V(last tile, key group 3) and the rest of its row sums ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    A8_SB();
    static_for<NA>([&](auto i_) { pv_mfma(std::integral_constant<int, 3>{}, i_, std::true_type{}); A8_SB(); });
    if (odd) static_for<8>([&](auto r_) { static_for<NQ>([&](auto qb_) { sum_r(I1{}, qb_, std::integral_constant<int, 8 + decltype(r_)::value>{}); }); });
    else static_for<8>([&](auto r_) { static_for<NQ>([&](auto qb_) { sum_r(I0{}, qb_, std::integral_constant<int, 8 + decltype(r_)::value>{}); }); });
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) l_run[qb] += psum[qb];
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // MFMA results -> v_accvgpr_read

    // ---- normalise and store: lane holds O[q_row][d = db*32 + 8g + 4hi + j] ----
    static_for<NQ>([&](auto qb_) {
        constexpr int qb = decltype(qb_)::value;
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        const float inv = 1.0f / l_tot;
        const int q_row = a.q_begin + (qb0 + qb) * 32 + ql;
        bf16_t* orow = a.out + ((size_t)b * a.out_rows_per_batch + q_row - a.q_begin) * a.ld_out + h * 128;
        if (q_row < a.Lq_rows) {
            static_for<16>([&](auto g_) {
                constexpr int g = decltype(g_)::value, db = g / 4, g4 = g % 4;
                const float o0 = acc_read<a_o(qb, db) + 4 * g4 + 0>(), o1 = acc_read<a_o(qb, db) + 4 * g4 + 1>();
                const float o2 = acc_read<a_o(qb, db) + 4 * g4 + 2>(), o3 = acc_read<a_o(qb, db) + 4 * g4 + 3>();
                u32x2 pk;
                pk[0] = pack_bf2(o0 * inv, o1 * inv);
                pk[1] = pack_bf2(o2 * inv, o3 * inv);
                *(u32x2*)(orow + db * 32 + 8 * g4 + 4 * hi) = pk;
            });
        }
    });
}

template <int VAR>
__global__ __launch_bounds__(256, 1) void attn64_fwd_kernel(AttnArgs a) {
    if ((int)blockIdx.x < a.n_full) attn64_body<2, VAR>(a, blockIdx.x, a.full_per_pair, 0);
    else attn64_body<1, VAR>(a, blockIdx.x - a.n_full, a.half_per_pair, 8 * a.full_per_pair);
}

// Pass plan: every (batch, head) pair's nqb q-blocks are covered by F full passes (8 q-blocks each) and H half passes (4
// each).  A CU runs one pass at a time, the hardware hands out workgroups in grid order (full passes first), so the
// launch lasts as long as greedy list scheduling of {pairs * F jobs of length 1, pairs * H jobs of length T_HALF} on
// `cus` machines: pick the F that minimises it.  (B = 1, 32 heads, 77 q-blocks on 256 CUs: F = 8, H = 4 — one round of
// full passes, then half passes on half of the CUs.)
constexpr double T_HALF = 0.80;   // measured (tools/dbg/attn64_stamps.py): a half pass is issue-bound, 2296 vs 2877 cycles per key tile

double plan_makespan(int pairs, int F, int H, int cus) {
    const long long nf = (long long)pairs * F, nh = (long long)pairs * H;
    std::vector<double> load(cus);
    for (int i = 0; i < cus; ++i) load[i] = (double)(nf / cus + (i < nf % cus ? 1 : 0));
    for (long long j = 0; j < nh; ++j) {
        int best = 0;
        for (int i = 1; i < cus; ++i)
            if (load[i] < load[best] - 1e-9) best = i;
        load[best] += T_HALF;
    }
    double top = 0;
    for (int i = 0; i < cus; ++i) top = load[i] > top ? load[i] : top;
    return top;
}

// full passes per (batch, head) pair for `pairs` pairs of nqb q-blocks on `cus` CUs (cached: a handful of shapes per run)
int plan_full_passes(int pairs, int nqb, int cus) {
    static std::mutex mu;
    static std::map<std::pair<long long, int>, int> cache;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair((long long)pairs * 4096 + cus, nqb);   // the plan depends on the device's CU count too
    const auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int bestF = 0;
    double best = 1e30;
    for (int F = 0; F <= nqb / 8; ++F) {
        const int H = (nqb - 8 * F + 3) / 4;
        const double t = plan_makespan(pairs, F, H, cus);
        if (t < best - 1e-9) { best = t; bestF = F; }
    }
    cache[key] = bestF;
    return bestF;
}

}  // namespace

// var: 0 = the product kernel; 1..15 (-DMMADA_TUNE builds only) = diagnostic variants, see attn64_body
int launch_attention64(attn_detail::AttnArgs a, int B, hipStream_t s, int var) {
    typedef void (*kern_t)(AttnArgs);
    kern_t fn = attn64_fwd_kernel<0>;
#ifdef MMADA_TUNE
    switch (var) {
        case 1: fn = attn64_fwd_kernel<1>; break;
        case 2: fn = attn64_fwd_kernel<2>; break;
        case 4: fn = attn64_fwd_kernel<4>; break;
        case 8: fn = attn64_fwd_kernel<8>; break;
        case 7: fn = attn64_fwd_kernel<7>; break;
        case 14: fn = attn64_fwd_kernel<14>; break;
        case 16: fn = attn64_fwd_kernel<16>; break;
        case 32: fn = attn64_fwd_kernel<32>; break;
        case 128: fn = attn64_fwd_kernel<128>; break;
        case 129: fn = attn64_fwd_kernel<129>; break;
        case 130: fn = attn64_fwd_kernel<130>; break;
        case 132: fn = attn64_fwd_kernel<132>; break;
        case 136: fn = attn64_fwd_kernel<136>; break;
    }
#else
    if (var != 0) return mm_fail("attention64: diagnostic variant %d exists in -DMMADA_TUNE builds only", var);
#endif
    // 64 KiB of dynamic LDS: above the default limit, per device like every other launcher's attribute
    static MmOncePerDevice attr_set;
    MM_ONCE_PER_DEVICE(attr_set, MM_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES)));
    static std::atomic<int> cus_of[16];
    const int slot = mm_device_slot();
    int cus = slot >= 0 ? cus_of[slot].load(std::memory_order_relaxed) : 0;
    if (!cus) {
        int dev = 0;
        MM_CHECK_HIP(hipGetDevice(&dev));
        MM_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (slot >= 0) cus_of[slot].store(cus, std::memory_order_relaxed);
    }
    const int pairs = a.Hq * B;
    const int nqb = (a.Lq_rows - a.q_begin + 31) / 32;
    const int bestF = plan_full_passes(pairs, nqb, cus);
    a.full_per_pair = bestF;
    a.half_per_pair = (nqb - 8 * bestF + 3) / 4;
    a.n_full = pairs * a.full_per_pair;
    static const bool xcd_aware = [] { const char* e = getenv("MMADA_ATTN_XCD"); return !(e && e[0] == '0'); }();
    a.xcd_pairs = (xcd_aware && pairs % 8 == 0) ? pairs / 8 : 0;
    a.nq = 0;
    const int grid = a.n_full + pairs * a.half_per_pair;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 4 * TILE_BYTES, s, a);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}

#ifdef MMADA_TUNE
// tuning builds: the phase stamps of the last attention64 launch with VAR bit 7 -> host (1024 x 4 x 8 uint64)
extern "C" int mmada_tune_attn64_stamps(void* out_host) {
    MM_CHECK_HIP(hipDeviceSynchronize());
    MM_CHECK_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_attn64_stamps), sizeof(unsigned long long) * 1024 * 4 * 8));
    return 0;
}
#endif
