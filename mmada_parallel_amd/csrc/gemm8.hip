// gemm8.hip — the 8-phase bf16 MFMA GEMM  C[M,N] = A[M,K] · W[N,K]^T  for gfx950 (round 3; same contraction, same fused
// epilogues and the same K order — hence bit-identical results — as the 16-wave kernel of gemm.hip, 12-21 % faster on the
// projection shapes of the 8B block: profiles/r03_gemm8_sweep*.txt).
//
// It replaces F.linear in q/k/v_proj, attn_out, ff_proj/up_proj, ff_out (model/modeling_llada.py:925-927, 741-744, 962-970).
//
// Structure (CDNA guide §5 "256² 8-phase template", T3 + T4, rebuilt for several tile shapes):
//   * BM x BN x 64 tile, 8 waves = WM x WN, TWO waves per SIMD (256 registers each); a wave owns TM x TN outputs.
//   * a K-tile is FOUR phases; a phase multiplies one quadrant of the wave tile: one A half (FA fragments x 2 k-steps, held
//     in registers for two phases) against one W half (FB fragments x 2 k-steps):
//         P1 (A0,B0)   P2 (A0,B1)   P3 (A1,B1)   P4 (A1,B0)
//     phase = { LDS-DMA issue for a later K-tile ; ds_read_b128 of ONE operand half ; counted vmcnt ; barrier ;
//               MFMA cluster ; barrier }.  Balanced schedule (tile_bal: every shipped tile since round 5): P1 reads A0,
//     P2 B1, P3 A1 and P4 the NEXT K-tile's B0 (into the W registers P3 released), so no phase has to fetch two halves
//     while its SIMD partner is only 20 MFMAs long.  First schedule (tile: tuning builds only since round 5; the 320x256 tile
//     shipped it while its balanced RESID / QKV builds still spilled): P1 reads A0 and B0, P2 B1, P3 A1, P4 nothing.
//   * the LDS image of a K-tile is cut the same way into four HALF-TILES (A0, A1, B0, B1 = the rows every wave reads for
//     that half).  A half-tile slot is re-filled two phases after its last ds_read, with the data of the K-tile AFTER next:
//     four half-tiles (one whole K-tile, up to 72 KiB per CU) are always in flight, each issued at least four phases — one
//     K-tile of MFMA time — before it is needed, and the queue never drains: every wait is a COUNTED vmcnt that leaves four
//     half-tiles outstanding.  (The 16-wave kernel requests K-tile t+1 in one burst behind the barrier of K-tile t and
//     drains the queue at the next barrier: 41 % of its wave time was spent there, profiles/r02_pmc_sq_model.txt.)
//   * the two wave groups (waves 0-3 / 4-7: one of each on every SIMD) run ONE BARRIER APART: while one wave of a SIMD is
//     in its MFMA cluster its partner issues the ds_reads and LDS-DMA of its next phase (measured: -18 % without it).
// Hazard rules (guide: "read a staged buffer one phase AFTER the wait that retires it" / WAR two phases), checked for
// every K-tile count by tests/test_host_logic.py::test_gemm8_schedule (a program-order simulation of the queue):
//   RAW: a half-tile is read in the phase after the one whose (pre-barrier) counted wait retired it;
//   WAR: a slot is re-staged >= 2 phases after the phase that read it.
// The counted waits assume that NOTHING but LDS-DMA is in the vector-memory queue between the prologue and the last
// wait: tests/test_isa.py disassembles the library and checks that the main loops contain no scratch or global access.
#include <cstdlib>

#include "gemm_epilogue.h"

namespace {

using namespace gemm_detail;

constexpr int BK = 64;

#define G8_SB() __builtin_amdgcn_sched_barrier(0)
#define G8_BARRIER()                            \
    do {                                        \
        G8_SB();                                \
        asm volatile("s_barrier" ::: "memory"); \
        G8_SB();                                \
    } while (0)

// The kernel owns the whole LDS allocation and has no static __shared__ object, so the dynamic segment starts at LDS
// address 0 (tests/test_isa.py: group_segment_fixed_size == 0): LDS addresses are formed from plain integers, which
// spares the "base + offset" VALU add and its temporary per access (the 320-row tile has 24 registers to spare).
typedef __attribute__((address_space(3))) const bf16x8* lds_frag_ptr;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
#pragma clang diagnostic ignored "-Wint-to-void-pointer-cast"
MM_DEVICE lds_frag_ptr lds_frag(int byte_off) { return (lds_frag_ptr)(uint32_t)byte_off; }
MM_DEVICE lptr_t lds_at(int byte_off) { return (lptr_t)(uint32_t)byte_off; }
#pragma clang diagnostic pop

// LDS-DMA as an asm statement (round 4): behind the BUILTIN hipcc stops counting LDS reads — every wait in front of an MFMA
// cluster was lgkmcnt(0), the whole operand half had to land before the first MFMA (tools/dbg/wc reproduces it); behind a
// request it cannot see it keeps its counted lgkmcnt(N) and the cluster starts on the first fragments.  Nothing else
// changes: the vector-memory queue is ordered by the explicit counted waits of wait_vm, as before.
MM_DEVICE void dma16(const char* sbase, unsigned& voff, int lds_byte) {
    const int m0v = __builtin_amdgcn_readfirstlane(lds_byte);
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(m0v) : "memory");
}

template <int N>
MM_DEVICE void wait_vm() {
    G8_SB();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    G8_SB();
}

// SW: operand roles of the MFMAs.  0: (activation, weight) — the C layout of gemm_epilogue;  1: swapped, the transposed C
// layout of gemm_epilogue_t (a lane owns four consecutive columns of a row, eight after a half-row exchange: 16-byte epilogue accesses);  2: chosen per wave
// (`swap`): the QKV projection, whose V waves want the untransposed layout.  Same products, same k order: same bits.
// OPT: bit 0 = static s_setprio 1 for the late wave group (tuning builds only, tools/gemm_sweep.py);  bit 1 = balanced
// read schedule (tile_bal);  bits 2-4 and 6 are DIAGNOSTIC (tuning builds, wrong results, timing only): 4 = no MFMAs,
// 8 = no LDS-DMA, 16 = no ds_reads, 64 = every tile streams the operand panels of tile (0, 0).
template <int BM_, int BN_, int WM_, int WN_, int SW_, int OPT_ = 0>
struct Gemm8 {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, SW = SW_, OPT = OPT_;
    static constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
    static constexpr int FA0 = (FM + 1) / 2, FA1 = FM / 2, FB = FN / 2;  // fragments of the A halves / of a W half
    // LDS image: [A of buffer 0][A of buffer 1][W of buffer 0][W of buffer 1] — both buffers of an operand lie within the
    // 16-bit immediate of ds_read_b128 from ONE per-lane base (two bases per operand, one per k-step)
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, W_BASE = 2 * A_BYTES, LDS = 2 * (A_BYTES + B_BYTES);
    static_assert(A_BYTES + TM * 128 <= 65536 && B_BYTES + TN * 128 <= 65536, "second buffer within the ds_read immediate");
    // 1-KiB LDS-DMA pieces (8 rows x 128 B) per half-tile; wave w moves pieces w, w+8, w+16
    static constexpr int NPA0 = WM * FA0 * 2, NPA1 = WM * FA1 * 2, NPB = WN * FB * 2;
    static constexpr int RA0 = NPA0 % 8, RA1 = NPA1 % 8;  // waves below the remainder move one piece more
    static constexpr int REM = RA0 ? RA0 : RA1;           // the one wave-uniform predicate that separates the two code paths
    static constexpr int NB = NPB / 8;
    static_assert(WM * WN == 8 && TM % 16 == 0 && TN % 32 == 0 && FA1 >= 1, "wave tile");
    // SHORT ROW TILES (round 4).  The stream has M = B * 2440 rows: 152.5 / 305 fragments of 16 rows, which 320-row tiles
    // cover with 160 / 320 — 4.7 % of every MFMA cluster multiplies rows that do not exist.  305 = 15 * 19 + 20 and
    // 153 = 7 * 19 + 20: with a row-tile pitch of 304 every row tile but the last one is 19 fragments high, the last one
    // takes what is left (<= 320).  A short tile is the same tile whose LAST wave row leaves out its last A fragment
    // (DROP = 1: FA1 - 1 fragments in the second A half): waves 4-7 issue 16 instead of 20 MFMAs in P3 and P4, a SIMD 152
    // instead of 160 per K-tile.  Nothing else moves: same LDS image (the 16 rows are staged and not read), same barriers,
    // same k order for every output that is computed — the same bits — and the epilogue stops at the tile's last row.
    static constexpr bool CAN_DROP = WM == 2 && FA1 >= 2 && BM == 320;
    static constexpr int SHORT_BM = BM - 16;
    static_assert(NPB % 8 == 0 && (RA0 == 0 || RA1 == 0 || RA0 == RA1) && NPA0 <= 24 && NPA1 <= 24, "piece split");
    static_assert(LDS + SiluLut::BYTES <= 160 * 1024, "LDS (K-tile buffers + the SwiGLU epilogue's SiLU table)");

    const char* Ab;  // wave-uniform byte bases of the A / W panels of this output tile
    const char* Wb;
    const char* Zb;  // 8 zero rows of lda elements (source of A pieces that lie wholly beyond M), or null
    // LDS-DMA addressing: a piece's first row is wave-uniform and goes into the SCALAR base; the per-lane part (row inside
    // the piece, swizzled 16-B chunk) is ONE 32-bit register per operand: the swizzle of row r is (r >> 1) & 7 and a piece
    // starts at a multiple of 8 rows, so the lane part depends on the piece's parity only — which is the wave's parity
    // (a wave's pieces are 8 apart and every chunk of a half-tile holds an even number of pieces).
    unsigned alane, wlane;
    unsigned arow[2][3], wrow[2][2];  // wave-uniform source byte offsets of this wave's pieces, [half][piece]
    bool azero[2][3];                 // piece lies wholly beyond M: stream zeros (the products are discarded either way,
                                      // but MFMAs on zeros switch far less — this workload runs at the power limit)
    int alds[2][3], wlds[2][2];       // wave-uniform LDS byte offsets of those pieces in buffer 0
    int ra[2], rb[2];                 // per-lane LDS read offsets [k-step] of fragment 0 in buffer 0 (the other fragments
                                      // and the second buffer are immediates)
    const char* lut_src;              // SwiGLU builds: the SiLU table in device memory (null: none), staged behind the K-tile buffers
    int lut_wave;
    bf16x8 af[FA0][2], bf0[FB][2], bf1[FB][2];
    f32x4 acc[FM][FN];

    MM_DEVICE void init(const GemmArgs& g, int m0, int n0, int wave, int lane) {
        Ab = (const char*)(g.A + (size_t)((OPT & 64) ? 0 : m0) * g.lda);  // OPT 64 (diagnostic): every tile streams the panels of
        Wb = (const char*)(g.W + (size_t)((OPT & 64) ? 0 : n0) * g.ldw);  // tile (0, 0) — an operand stream without L2 misses
        Zb = (const char*)g.zero_row;
        const int wm = wave / WN, wn = wave % WN;
        const int prow = lane >> 3, psw = ((wave & 1) * 4 + (prow >> 1)) & 7;
        alane = (unsigned)prow * (unsigned)(g.lda * 2) + (((lane & 7) ^ psw) << 4);
        wlane = (unsigned)prow * (unsigned)(g.ldw * 2) + (((lane & 7) ^ psw) << 4);
        const int mrem = g.M - m0, nrem = g.N - n0;  // rows of this tile that exist (multiples of 8: gemm8_supports)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int fa = h ? FA1 : FA0, npa = h ? NPA1 : NPA0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int p = min(wave + 8 * i, npa - 1);
                const int row0 = (p / (fa * 2)) * TM + (h ? FA0 * 16 : 0) + (p % (fa * 2)) * 8;
                azero[h][i] = Zb != nullptr && row0 >= mrem;
                arow[h][i] = azero[h][i] ? 0u : (unsigned)min(row0, mrem - 8) * (unsigned)(g.lda * 2);
                alds[h][i] = row0 * 128;
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int p = wave + 8 * i;
                const int row0 = (p / (FB * 2)) * TN + h * FB * 16 + (p % (FB * 2)) * 8;
                wrow[h][i] = (unsigned)min(row0, nrem - 8) * (unsigned)(g.ldw * 2);
                wlds[h][i] = W_BASE + row0 * 128;
            }
        }
        const int frow = lane & 15, fq = lane >> 4, sw = (frow >> 1) & 7;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int lanepart = frow * 128 + (((kk * 4 + fq) ^ sw) << 4);
            ra[kk] = wm * TM * 128 + lanepart;
            rb[kk] = W_BASE + wn * TN * 128 + lanepart;
            asm volatile("" : "+v"(ra[kk]), "+v"(rb[kk]));  // opaque: two bases per operand, never re-derived
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (OPT & 16) {  // diagnostic build without ds_reads: operands that are not zeros (MFMAs on zeros draw less power)
            bf16x8 junk;
#pragma unroll
            for (int e = 0; e < 8; ++e) junk[e] = (__bf16)(0.37f + 0.01f * (float)((lane * 7 + e * 13) & 63));
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int i = 0; i < FA0; ++i) af[i][kk] = junk;
#pragma unroll
                for (int j = 0; j < FB; ++j) { bf0[j][kk] = junk; bf1[j][kk] = junk; }
            }
        }
    }

    template <int B, int H, int NA>
    MM_DEVICE void stage_a(int kt) {
        if (OPT & 8) return;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const char* src = (azero[H][i] ? Zb : Ab + arow[H][i]) + (size_t)kt * (BK * 2);
            // scalar base + 32-bit lane offset, zero-extended HERE: saddr form.  The lane offset is made opaque IN PLACE
            // (no copy: the 320-row tile has no register to spare for one).
            dma16(src, alane, B * A_BYTES + alds[H][i]);
        }
    }
    template <int B, int H>
    MM_DEVICE void stage_w(int kt) {
        if (OPT & 8) return;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const char* src = Wb + wrow[H][i] + (size_t)kt * (BK * 2);
            dma16(src, wlane, B * B_BYTES + wlds[H][i]);
        }
    }
    template <int B, int H, int DROP = 0>
    MM_DEVICE void read_a() {
        if (OPT & 16) return;
#pragma unroll
        for (int mi = 0; mi < (H ? FA1 - DROP : FA0); ++mi)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) af[mi][kk] = *lds_frag(ra[kk] + B * A_BYTES + ((H ? FA0 * 16 : 0) + mi * 16) * 128);
    }
    template <int B, int H>
    MM_DEVICE void read_b(bf16x8 (&bf)[FB][2]) {
        if (OPT & 16) return;
#pragma unroll
        for (int nj = 0; nj < FB; ++nj)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) bf[nj][kk] = *lds_frag(rb[kk] + B * B_BYTES + (H * FB * 16 + nj * 16) * 128);
    }
    template <int H, int NH, bool SWAP, int DROP = 0>
    MM_DEVICE void mma(bf16x8 (&bf)[FB][2]) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mi = 0; mi < (H ? FA1 - DROP : FA0); ++mi)
#pragma unroll
                for (int nj = 0; nj < FB; ++nj) {
                    f32x4& c = acc[(H ? FA0 : 0) + mi][NH * FB + nj];
                    if (OPT & 4) { asm volatile("" : "+v"(c) : "v"(af[mi][kk]), "v"(bf[nj][kk])); continue; }
                    c = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[nj][kk], af[mi][kk], c, 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi][kk], bf[nj][kk], c, 0, 0, 0);
                }
    }


    // One K-tile (index kt, LDS buffer B).  TAIL 0: steady state (kt + 2 < nk); 1: second-to-last; 2: last K-tile.
    // LDS-DMA queue, oldest first, at the top of P1(kt): B1(kt) A1(kt) A0(kt+1) B0(kt+1)  (A0, B0 of kt have landed).
    template <int B, int NA0, int NA1, int TAIL, bool SWAP, int DROP>
    MM_DEVICE void tile(int kt) {
        constexpr int FOUR = NA0 + NA1 + 2 * NB;  // this wave's LDS-DMA instructions of four half-tiles
        // ---- P1: quadrant (A0, B0) ----
        if (TAIL < 2) stage_w<B ^ 1, 1>(kt + 1);
        read_a<B, 0>();
        read_b<B, 0>(bf0);
        if (TAIL < 2) wait_vm<FOUR>(); else wait_vm<NA1>();  // B1(kt) has landed
        G8_BARRIER();
        mma<0, 0, SWAP>(bf0);
        G8_BARRIER();
        // ---- P2: (A0, B1) ----
        if (TAIL < 2) stage_a<B ^ 1, 1, NA1>(kt + 1);
        read_b<B, 1>(bf1);
        if (TAIL < 2) wait_vm<FOUR>(); else wait_vm<0>();    // A1(kt) has landed
        G8_BARRIER();
        mma<0, 1, SWAP>(bf1);
        G8_BARRIER();
        // ---- P3: (A1, B1) ----
        if (TAIL == 0) stage_a<B, 0, NA0>(kt + 2);
        read_a<B, 1, DROP>();
        G8_BARRIER();
        mma<1, 1, SWAP, DROP>(bf1);
        G8_BARRIER();
        // ---- P4: (A1, B0) ----
        if (TAIL == 0) stage_w<B, 0>(kt + 2);
        if (TAIL == 0) wait_vm<FOUR>();                      // A0, B0 of K-tile kt+1 have landed
        else if (TAIL == 1) wait_vm<NB + NA1>();
        G8_BARRIER();
        mma<1, 0, SWAP, DROP>(bf0);
        G8_BARRIER();
    }


    // The same K-tile with BALANCED reads (OPT bit 1): one half-tile staged, read and retired per phase.
    // LDS-DMA queue, oldest first, at the top of P1(kt): B1(kt) A1(kt) B0(kt+1) A0(kt+1)  (A0 of kt has landed; B0 of kt is
    // already in registers: it was read during P4 of the previous K-tile, into the W register set that P3 had just
    // released — the two sets swap roles from one K-tile to the next, which is the buffer parity B).  Every phase stages
    // one half-tile, reads one half-tile and retires one half-tile; four half-tiles stay in flight behind every wait.
    template <int B, int NA0, int NA1, int TAIL, bool SWAP, int DROP>
    MM_DEVICE void tile_bal(int kt) {
        constexpr int FOUR = NA0 + NA1 + 2 * NB;  // this wave's LDS-DMA instructions of four half-tiles
        bf16x8(&F)[FB][2] = B ? bf1 : bf0;        // holds W half 0 of this K-tile
        bf16x8(&S)[FB][2] = B ? bf0 : bf1;        // W half 1 of this K-tile, then W half 0 of the next one
        // ---- P1: quadrant (A0, B0) ----
        if (TAIL < 2) stage_w<B ^ 1, 1>(kt + 1);
        read_a<B, 0>();
        if (TAIL < 2) wait_vm<FOUR>(); else wait_vm<NA1>();  // B1(kt) has landed
        G8_BARRIER();
        mma<0, 0, SWAP>(F);
        G8_BARRIER();
        // ---- P2: (A0, B1) ----
        if (TAIL < 2) stage_a<B ^ 1, 1, NA1>(kt + 1);
        read_b<B, 1>(S);
        if (TAIL < 2) wait_vm<FOUR>(); else wait_vm<0>();    // A1(kt) has landed
        G8_BARRIER();
        mma<0, 1, SWAP>(S);
        G8_BARRIER();
        // ---- P3: (A1, B1) ----
        if (TAIL == 0) stage_w<B, 0>(kt + 2);
        read_a<B, 1, DROP>();
        if (TAIL == 0) wait_vm<FOUR>();                      // B0(kt+1) has landed
        else if (TAIL == 1) wait_vm<NA0 + NB + NA1>();
        G8_BARRIER();
        mma<1, 1, SWAP, DROP>(S);
        G8_BARRIER();
        // ---- P4: (A1, B0) ----
        if (TAIL == 0) stage_a<B, 0, NA0>(kt + 2);
        if (TAIL < 2) read_b<B ^ 1, 0>(S);
        if (TAIL == 0) wait_vm<FOUR>();                      // A0(kt+1) has landed
        else if (TAIL == 1) wait_vm<NB + NA1>();
        G8_BARRIER();
        mma<1, 0, SWAP, DROP>(F);
        G8_BARRIER();
    }

    template <int NA0, int NA1, bool SWAP, int DROP>
    MM_DEVICE void run(int nk, int grp) {
        constexpr bool BAL = (OPT & 2) != 0;
        // the vector-memory queue is empty here at run time; saying so keeps the static check of the counted waits
        // (tools/isa_check.py) independent of how the compiler lays the two code paths out
        wait_vm<0>();
        if (lut_src) {
            // the SiLU table of the SwiGLU epilogue: ten 1-KiB pieces requested BEFORE the pipeline's first piece — the oldest
            // entries of the vector-memory queue, so every counted wait below also covers them and none has to change; the
            // epilogue reads the table a whole main loop (and many barriers) later
            unsigned lane16 = (unsigned)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) * 16u;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int p = lut_wave + 8 * i;
                if (p < SiluLut::PIECES) dma16(lut_src + p * 1024, lane16, LDS + p * 1024);
            }
        }
        stage_a<0, 0, NA0>(0); stage_w<0, 0>(0); stage_w<0, 1>(0); stage_a<0, 1, NA1>(0);
        if (BAL) { stage_w<1, 0>(1); stage_a<1, 0, NA0>(1); }
        else { stage_a<1, 0, NA0>(1); stage_w<1, 0>(1); }
        wait_vm<NA0 + NA1 + 2 * NB>();
        G8_BARRIER();
        if (BAL) read_b<0, 0>(bf0);
        if (grp == 1) G8_BARRIER();  // waves 4-7 run one barrier behind waves 0-3
        if ((OPT & 1) && grp == 1) __builtin_amdgcn_s_setprio(1);
        if constexpr (BAL) {
            for (int kt = 0; kt + 2 < nk; kt += 2) {
                tile_bal<0, NA0, NA1, 0, SWAP, DROP>(kt);
                tile_bal<1, NA0, NA1, 0, SWAP, DROP>(kt + 1);
            }
            // the two tail K-tiles sit in a loop of ONE trip the compiler cannot count: as straight-line code hipcc picks the
            // three-address MFMA form there (destination != accumulator), which needs free register tuples the 320-row tile
            // does not have — 16 extra registers and a spilled accumulator; loop-carried accumulators stay in place
            int one = 1;
            asm volatile("" : "+s"(one));
#pragma nounroll
            for (int t = 0; t < one; ++t) {
                tile_bal<0, NA0, NA1, 1, SWAP, DROP>(nk - 2);
                tile_bal<1, NA0, NA1, 2, SWAP, DROP>(nk - 1);
            }
        } else {
            for (int kt = 0; kt + 2 < nk; kt += 2) {
                tile<0, NA0, NA1, 0, SWAP, DROP>(kt);
                tile<1, NA0, NA1, 0, SWAP, DROP>(kt + 1);
            }
            int one = 1;   // see above
            asm volatile("" : "+s"(one));
#pragma nounroll
            for (int t = 0; t < one; ++t) {
                tile<0, NA0, NA1, 1, SWAP, DROP>(nk - 2);
                tile<1, NA0, NA1, 2, SWAP, DROP>(nk - 1);
            }
        }
        if (grp == 0) G8_BARRIER();
    }
};

// mmada_set_option("gemm_short_tiles", 0) / MMADA_GEMM_SHORT_TILES=0: every row tile full height (the round-3 kernel; A/B timing)
std::atomic<int> g_short_tiles{-1};
bool short_tiles_on() {
    if (g_short_tiles < 0) {
        const char* e = getenv("MMADA_GEMM_SHORT_TILES");
        g_short_tiles = e && e[0] == '0' ? 0 : 1;
    }
    return g_short_tiles != 0;
}

std::atomic<int> g_tile_order{-1};  // -1: read MMADA_GEMM_TILE_ORDER once; 0: per-tile default (launch_cfg8); GM * 100 + GN (GM = 99: all row tiles)

template <int EPI, class G>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(GemmArgs g) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ntm = (g.M + G::BM - 1) / G::BM, ntn = (g.N + G::BN - 1) / G::BN;
    int mt, nt;
    tile_coords_g(xcd_remap(blockIdx.x, gridDim.x), ntm, ntn, g.tile_gm, g.tile_gn, mt, nt);
    // short row tiles (Gemm8::CAN_DROP): pitch 304, every row tile but the last one ends 16 rows early
    const bool short_tiles = G::CAN_DROP && g.row_drop != 0;
    const bool short_tile = short_tiles && mt < ntm - 1;
    const int m0 = mt * (short_tiles ? G::SHORT_BM : G::BM), n0 = nt * G::BN;
    const int m_lim = short_tile ? m0 + G::SHORT_BM : g.M;   // the epilogue's view: rows of this tile only
    const bool drop = short_tile && wave / G::WN == G::WM - 1;
    G k;
    // SwiGLU: the SiLU table (gemm_epilogue.h: SiluLut) goes into the LDS behind the K-tile buffers (run() requests it)
    const bool use_lut = EPI == EPI_SWIGLU && g.silu_lut != nullptr;
    const int lut_lds = use_lut ? G::LDS : -1;
    k.init(g, m0, n0, wave, threadIdx.x & 63);
    k.lut_src = use_lut ? (const char*)g.silu_lut : nullptr;
    k.lut_wave = wave;
    const int nk = g.K / BK;
    // Up to four self-contained code paths (a wave moves one LDS-DMA piece more per A half-tile or not; operand roles
    // swapped or not): they never rejoin with live accumulators — each runs its own epilogue, which recomputes the lane
    // id so that nothing of its addressing is live across the main loop.
    const bool extra = G::REM != 0 && wave < G::REM;
    const bool swap = G::SW == 1 || (G::SW == 2 && !qkv_wave_is_v(g, n0 + (wave % G::WN) * G::TN));
    constexpr int NA0 = G::NPA0 / 8, NA1 = G::NPA1 / 8, XA0 = NA0 + (G::RA0 ? 1 : 0), XA1 = NA1 + (G::RA1 ? 1 : 0);
#define G8_PATH1(A0, A1, SWP, DRP)                                                                  \
    do {                                                                                            \
        k.template run<A0, A1, SWP, DRP>(nk, wave >> 2);                                            \
        const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));        \
        if (SWP) gemm_epilogue_t<EPI, G::TM, G::TN, G::WN>(g, m0, n0, k.acc, wave, lane, m_lim, lut_lds); \
        else gemm_epilogue<EPI, G::TM, G::TN, G::WN>(g, m0, n0, k.acc, wave, lane, m_lim);               \
    } while (0)
#define G8_PATH(A0, A1, SWP)                                                                        \
    do {                                                                                            \
        if constexpr (G::CAN_DROP) {                                                                \
            if (drop) G8_PATH1(A0, A1, SWP, 1);                                                     \
            else G8_PATH1(A0, A1, SWP, 0);                                                          \
        } else G8_PATH1(A0, A1, SWP, 0);                                                            \
    } while (0)
    if (G::SW != 0 && swap) {
        if (extra) G8_PATH(XA0, XA1, true);
        else G8_PATH(NA0, NA1, true);
    } else if (G::SW != 1) {
        if (extra) G8_PATH(XA0, XA1, false);
        else G8_PATH(NA0, NA1, false);
    }
#undef G8_PATH
#undef G8_PATH1
    gemm_publish(g, wave);
}

template <int EPI, class G>
int launch_cfg8(const GemmArgs& g, hipStream_t s) {
    auto fn = gemm8_kernel<EPI, G>;
    static MmOncePerDevice attr_set;
    constexpr int LDS_BYTES = G::LDS + (EPI == EPI_SWIGLU ? SiluLut::BYTES : 0);
    MM_ONCE_PER_DEVICE(attr_set, MM_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES)));
    const int ntm = (g.M + G::BM - 1) / G::BM, ntn = (g.N + G::BN - 1) / G::BN;
    GemmArgs ga = g;
    // short row tiles when ntm - 1 tiles of 304 rows and one of <= 320 cover M (same tile count, 5 % fewer MFMAs in all
    // but the last row tile): M = 2440 -> 7 x 304 + 312, M = 4880 -> 15 x 304 + 320
    ga.row_drop = G::CAN_DROP && short_tiles_on() && ntm >= 2 && (ntm - 1) * G::SHORT_BM + G::BM >= g.M ? 1 : 0;
    if (g_tile_order < 0) {
        const char* e = getenv("MMADA_GEMM_TILE_ORDER");
        g_tile_order = e ? atoi(e) : 0;
    }
    // Default order: per XCD-round of 32 tiles the fabric delivers gm A panels (BM rows each) + gn W panels (BN rows each), gm x gn
    // = 32, least when gm * BM ~ gn * BN: 4 x 8 for the 320 x 256 tile (FETCH_SIZE -6 % against 8 x 4, gate/up +1.3 % at M = 2440,
    // every projection +0.6 ... 1.9 % at M = 4880: profiles/r05_tile_order_fetch.txt, r05_block_ab.txt), 8 x 4 bands of 1024
    // columns for the shorter tiles (there 4 x 8 measured -0.6 ... -2.6 %).  Any order gives the same bits.
    constexpr int DEF_GM = (G::BM == 320 && G::BN == 256) ? 4 : 0, DEF_GN = (G::BM == 320 && G::BN == 256) ? 8 : 1024 / G::BN;
    ga.tile_gm = g_tile_order > 0 ? g_tile_order / 100 : DEF_GM;
    ga.tile_gn = g_tile_order > 0 && g_tile_order % 100 > 0 ? g_tile_order % 100 : DEF_GN;
    hipLaunchKernelGGL(fn, dim3(ntm * ntn), dim3(512), LDS_BYTES, s, ga);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int EPI>
int launch_epi8(int cfg, const GemmArgs& g, hipStream_t s) {
    constexpr int SW = EPI == EPI_QKV ? 2 : 1;
    switch (cfg) {
        // the balanced read schedule (OPT 2) on every tile.  Rounds 3-4 kept the first schedule on 320 x 256 because its RESID /
        // QKV builds spilled inside the loop with the balanced one; with the tail K-tiles in a one-trip loop (run()) all four
        // epilogues compile spill-free (254-255 VGPRs) and the in-model A/B reads +0.4 ... 1.0 % on gate/up and QKV, +-0 on the
        // rest (profiles/r05_block_ab_balanced.txt; tuning build: configuration 6 = the first schedule)
        case GEMM8_320x256: return launch_cfg8<EPI, Gemm8<320, 256, 2, 4, SW, 2>>(g, s);
        case GEMM8_256x256: return launch_cfg8<EPI, Gemm8<256, 256, 2, 4, SW, 2>>(g, s);
        case GEMM8_160x256: return launch_cfg8<EPI, Gemm8<160, 256, 2, 4, SW, 2>>(g, s);
        case GEMM8_320x128: return launch_cfg8<EPI, Gemm8<320, 128, 4, 2, SW, 2>>(g, s);
#ifdef MMADA_TUNE
        case 4: return launch_cfg8<EPI, Gemm8<320, 256, 2, 4, SW, 3>>(g, s);    // balanced + static s_setprio for the late group
        case 5: return launch_cfg8<EPI, Gemm8<320, 128, 4, 2, SW, 0>>(g, s);
        case 6: return launch_cfg8<EPI, Gemm8<320, 256, 2, 4, SW, 0>>(g, s);    // the first read schedule (rounds 3-4 production)
        case 7: return launch_cfg8<EPI, Gemm8<256, 256, 2, 4, SW, 0>>(g, s);
        case 8: return launch_cfg8<EPI, Gemm8<160, 256, 2, 4, SW, 0>>(g, s);
        case 9: return launch_cfg8<EPI, Gemm8<320, 256, 2, 4, SW, 4>>(g, s);    // DIAG: no MFMA
        case 10: return launch_cfg8<EPI, Gemm8<320, 256, 2, 4, SW, 8>>(g, s);   // DIAG: no LDS-DMA
        case 11: return launch_cfg8<EPI, Gemm8<320, 256, 2, 4, SW, 16>>(g, s);  // DIAG: no ds_reads
        case 12: return launch_cfg8<EPI, Gemm8<320, 256, 2, 4, SW, 24>>(g, s);  // DIAG: MFMAs + barriers only
        case 13: return launch_cfg8<EPI, Gemm8<320, 256, 2, 4, SW, 20>>(g, s);  // DIAG: LDS-DMA + barriers only
        case 14: return launch_cfg8<EPI, Gemm8<320, 256, 2, 4, SW, 84>>(g, s);  // DIAG: LDS-DMA + barriers only, no L2 misses
        case 15: return launch_cfg8<EPI, Gemm8<320, 256, 2, 4, SW, 64>>(g, s);  // DIAG: whole kernel, no L2 misses
#endif
    }
    return mm_fail("gemm8: unknown configuration %d", cfg);
}

}  // namespace

bool gemm8_supports(const GemmArgs& g) {
    // two K-tiles per loop iteration and a two-tile tail; whole 8-row LDS-DMA pieces; 32-bit source offsets inside a tile
    // ... and 16-byte epilogue accesses: leading dimensions of C and of the residual in whole 8-element groups
    return g.K % 128 == 0 && g.K >= 256 && g.M % 8 == 0 && g.N % 8 == 0 && g.M >= 8 && g.N >= 8 &&
           (long long)g.lda * 2 * 328 < (1ll << 31) && (long long)g.ldw * 2 * 264 < (1ll << 31) &&
           g.ldc % 8 == 0 && (g.resid == nullptr || g.ldr % 8 == 0) && ((uintptr_t)g.C & 15) == 0 && ((uintptr_t)g.resid & 15) == 0;
}

void gemm8_set_short_tiles(int on) { g_short_tiles = on != 0 ? 1 : 0; }
void gemm8_set_tile_order(int code) { g_tile_order = code < 0 ? -1 : code; }

int launch_gemm8(int epi, int cfg, const GemmArgs& g, hipStream_t s) {
    if (!gemm8_supports(g)) return mm_fail("gemm8: unsupported shape M=%d N=%d K=%d", g.M, g.N, g.K);
    switch (epi) {
        case EPI_STORE: return launch_epi8<EPI_STORE>(cfg, g, s);
        case EPI_RESID: return launch_epi8<EPI_RESID>(cfg, g, s);
        case EPI_SWIGLU: return launch_epi8<EPI_SWIGLU>(cfg, g, s);
        case EPI_QKV: return launch_epi8<EPI_QKV>(cfg, g, s);
    }
    return mm_fail("gemm8: bad epilogue %d", epi);
}
