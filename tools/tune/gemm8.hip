// gemm8.hip — "8-phase" bf16 MFMA GEMM  C[M,N] = A[M,K] · W[N,K]^T  for gfx950 (tuning variant; plain bf16 store).
//
// Structure (CDNA guide §5 "256² 8-phase template", T3+T4+T5, rebuilt here for BM in {256, 320}):
//   * BM x 256 x 64 tile, 8 waves = 2 (M) x 4 (N), two waves per SIMD; a wave owns (BM/2) x 64 outputs.
//   * a K-tile is FOUR phases; a phase multiplies one quadrant of the wave tile — (BM/4 rows) x (32 columns) x 64 —
//     i.e. one A half (FMH fragments x 2 k-steps, kept in registers for two phases) against one W half:
//         P1 (A0,B0)   P2 (A0,B1)   P3 (A1,B1)   P4 (A1,B0)
//   * the LDS image of a K-tile is cut the same way into four HALF-TILES (A0, A1, B0, B1: the rows every wave reads
//     for that half); a half-tile slot is refilled by LDS-DMA two phases after its last ds_read, for the K-tile AFTER
//     next — so four half-tiles (one whole K-tile, 64-72 KiB per CU) are always in flight, every one issued at least
//     four phases (one K-tile of MFMA time) before it is needed, and the queue is never drained: every wait is a
//     COUNTED vmcnt that leaves four half-tiles outstanding (T4).  Two K-tile buffers = 128 / 144 KiB of LDS.
//   * the two wave rows run one barrier apart (wave row 1 starts with an extra s_barrier): while one wave of a SIMD is in
//     its MFMA cluster its partner issues ds_reads + LDS-DMA for the next phase (T3's role split; s_setprio around the
//     cluster, T5).
// Hazard rules followed (guide "Read a staged buffer one phase AFTER the wait that retires it" / WAR two phases):
//   RAW: a half-tile is read in the phase after the one whose (pre-barrier) counted wait retired it;
//   WAR: a slot is re-staged >= 2 phases after the phase that read it.
#include "kernels.h"

namespace {

constexpr int BN8 = 256, BK8 = 64;
typedef const __attribute__((address_space(1))) void* gptr8_t;
typedef __attribute__((address_space(3))) void* lptr8_t;

#define G8_SB() __builtin_amdgcn_sched_barrier(0)
#define G8_BARRIER()                            \
    do {                                        \
        G8_SB();                                \
        asm volatile("s_barrier" ::: "memory"); \
        G8_SB();                                \
    } while (0)

// The kernel owns the whole LDS allocation and has no static __shared__ object, so the dynamic segment starts at LDS
// address 0 (checked at build time: group_segment_fixed_size == 0): LDS addresses are formed from plain integers, which
// spares the "base + offset" VALU add and its temporary per access.
typedef __attribute__((address_space(3))) const bf16x8* lds_frag_ptr;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
#pragma clang diagnostic ignored "-Wint-to-void-pointer-cast"
MM_DEVICE lds_frag_ptr lds_frag(int byte_off) { return (lds_frag_ptr)(uint32_t)byte_off; }
MM_DEVICE lptr8_t lds_at(int byte_off) { return (lptr8_t)(uint32_t)byte_off; }
#pragma clang diagnostic pop

template <int N>
MM_DEVICE void g8_wait_vm() {
    G8_SB();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    G8_SB();
}

// SCHED 0: balanced (four half-tiles in flight, waits in P1 / P2 / P4); SCHED 1: the guide's order (three in flight, one
// wait per K-tile in P4, A0 re-staged one phase after its read with the lgkmcnt retired before the barrier).
template <int BM, bool PRIO, int SCHED, bool STAGGER>
struct Gemm8 {
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN8 * 128, STAGE = A_BYTES + B_BYTES, LDS = 2 * STAGE;
    static constexpr int HM = BM / 4;     // rows of one wave's A half
    static constexpr int FMH = HM / 16;   // A fragments per half (4 or 5)
    static constexpr int PPC = BM / 32;   // 1-KiB pieces (8 rows) per (wave row, half) chunk of A
    static constexpr int NPA = 2 * PPC;   // A pieces per half-tile
    static_assert(HM % 16 == 0 && LDS <= 160 * 1024, "tile");

    char* smem;
    const char* Ab;  // wave-uniform byte bases of the A / W panels of this output tile
    const char* Wb;
    const char* Zb;  // 8 zero rows of lda elements (source of A pieces that lie wholly beyond M), or null
    // LDS-DMA addressing: a 1-KiB piece is 8 rows x 128 B; its first row is wave-uniform and goes into the SCALAR base, the
    // per-lane part (row inside the piece, swizzled 16-B chunk) is ONE 32-bit VGPR per operand — the swizzle of row
    // r is (r >> 1) & 7 and a piece starts at a multiple of 8 rows, so the lane part only depends on the piece's parity,
    // which is the wave's parity (a wave's pieces are 8 apart).
    unsigned alane, wlane;
    unsigned arow[2][3], wrow[2][2];  // wave-uniform source byte offsets of this wave's pieces, [half][piece]
    bool azero[2][3];                 // piece lies wholly beyond M: stream zeros (their products are discarded)
    int alds[2][3], wlds[2][2];       // wave-uniform LDS byte offsets of those pieces inside a stage
    int ra[2][2], rb[2][2];           // per-lane LDS read offsets [buffer][k-step] of fragment 0 of half 0 (everything else
                                      // is an immediate; a second buffer's offsets exceed the 16-bit immediate range)
    bf16x8 af[FMH][2], bf0[2][2], bf1[2][2];
    f32x4 acc[2][FMH][4];

    MM_DEVICE void init(const GemmArgs& g, char* sm, int m0, int n0, int wave, int lane) {
        smem = sm;
        Ab = (const char*)(g.A + (size_t)m0 * g.lda);
        Wb = (const char*)(g.W + (size_t)n0 * g.ldw);
        Zb = (const char*)g.zero_row;
        const int wm = wave >> 2, wn = wave & 3;
        const int prow = lane >> 3, psw = ((wave & 1) * 4 + (prow >> 1)) & 7;
        alane = (unsigned)prow * (unsigned)(g.lda * 2) + (((lane & 7) ^ psw) << 4);
        wlane = (unsigned)prow * (unsigned)(g.ldw * 2) + (((lane & 7) ^ psw) << 4);
        const int mrem = g.M - m0, nrem = g.N - n0;  // rows of this tile that exist (multiples of 8 by the launch contract)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int p = min(wave + 8 * i, NPA - 1);
                const int row0 = (p / PPC) * (BM / 2) + h * HM + (p % PPC) * 8;
                azero[h][i] = Zb != nullptr && row0 >= mrem;
                arow[h][i] = azero[h][i] ? 0u : (unsigned)min(row0, mrem - 8) * (unsigned)(g.lda * 2);
                alds[h][i] = row0 * 128;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int p = wave + 8 * i;  // 16 pieces per W half: 4 wave columns x 4 pieces
                const int row0 = (p >> 2) * 64 + h * 32 + (p & 3) * 8;
                wrow[h][i] = (unsigned)min(row0, nrem - 8) * (unsigned)(g.ldw * 2);
                wlds[h][i] = A_BYTES + row0 * 128;
            }
        }
        const int frow = lane & 15, fq = lane >> 4, sw = (frow >> 1) & 7;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int lanepart = frow * 128 + (((kk * 4 + fq) ^ sw) << 4);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                ra[b][kk] = b * STAGE + wm * (BM / 2) * 128 + lanepart;
                rb[b][kk] = b * STAGE + A_BYTES + wn * 64 * 128 + lanepart;
                asm volatile("" : "+v"(ra[b][kk]), "+v"(rb[b][kk]));  // opaque: four bases per operand, no re-derivation
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < FMH; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    template <int B, int H, int NA>
    MM_DEVICE void stage_a(int kt) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const char* src = (azero[H][i] ? Zb : Ab + arow[H][i]) + (size_t)kt * (BK8 * 2);
            unsigned off = alane;
            asm volatile("" : "+s"(src), "+v"(off));  // scalar base + 32-bit lane offset, zero-extended HERE: saddr form
            __builtin_amdgcn_global_load_lds((gptr8_t)(src + off), lds_at(B * STAGE + alds[H][i]), 16, 0, 0);
        }
    }
    template <int B, int H>
    MM_DEVICE void stage_w(int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const char* src = Wb + wrow[H][i] + (size_t)kt * (BK8 * 2);
            unsigned off = wlane;
            asm volatile("" : "+s"(src), "+v"(off));
            __builtin_amdgcn_global_load_lds((gptr8_t)(src + off), lds_at(B * STAGE + wlds[H][i]), 16, 0, 0);
        }
    }
    template <int B, int H>
    MM_DEVICE void read_a() {
#pragma unroll
        for (int mi = 0; mi < FMH; ++mi)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                af[mi][kk] = *lds_frag(ra[B][kk] + (H * HM * 128 + mi * 2048));
    }
    template <int B, int H>
    MM_DEVICE void read_b(bf16x8 (&bf)[2][2]) {
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                bf[nj][kk] = *lds_frag(rb[B][kk] + (H * 32 * 128 + nj * 2048));
    }
    template <int H, int NH>
    MM_DEVICE void mma(bf16x8 (&bf)[2][2]) {
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mi = 0; mi < FMH; ++mi)
#pragma unroll
                for (int nj = 0; nj < 2; ++nj)
                    acc[H][mi][NH * 2 + nj] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi][kk], bf[nj][kk], acc[H][mi][NH * 2 + nj], 0, 0, 0);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    }
    MM_DEVICE void lgkm0_if_guide() {
        if (SCHED == 1) {
            G8_SB();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            G8_SB();
        }
    }

    // One K-tile (index kt, LDS buffer B).  TAIL 0: steady state (kt + 2 < nk); 1: second-to-last; 2: last K-tile.
    template <int B, int NA, int TAIL>
    MM_DEVICE void tile(int kt) {
        constexpr int FOUR = 2 * NA + 4;  // LDS-DMA instructions of four half-tiles (2 x A, 2 x W) of this wave
        // ---- P1: quadrant (A0, B0) ----
        if (SCHED == 0) { if (TAIL < 2) stage_w<B ^ 1, 1>(kt + 1); }
        else            { if (TAIL < 2) stage_a<B ^ 1, 1, NA>(kt + 1); }
        read_a<B, 0>();
        read_b<B, 0>(bf0);
        if (SCHED == 0) { if (TAIL < 2) g8_wait_vm<FOUR>(); else g8_wait_vm<NA>(); }  // B1(kt) has landed
        lgkm0_if_guide();
        G8_BARRIER();
        mma<0, 0>(bf0);
        G8_BARRIER();
        // ---- P2: (A0, B1) ----
        if (SCHED == 0) { if (TAIL < 2) stage_a<B ^ 1, 1, NA>(kt + 1); }
        else            { if (TAIL == 0) stage_a<B, 0, NA>(kt + 2); }
        read_b<B, 1>(bf1);
        if (SCHED == 0) { if (TAIL < 2) g8_wait_vm<FOUR>(); else g8_wait_vm<0>(); }  // A1(kt) has landed
        lgkm0_if_guide();
        G8_BARRIER();
        mma<0, 1>(bf1);
        G8_BARRIER();
        // ---- P3: (A1, B1) ----
        if (SCHED == 0) { if (TAIL == 0) stage_a<B, 0, NA>(kt + 2); }
        else            { if (TAIL == 0) stage_w<B, 0>(kt + 2); }
        read_a<B, 1>();
        lgkm0_if_guide();
        G8_BARRIER();
        mma<1, 1>(bf1);
        G8_BARRIER();
        // ---- P4: (A1, B0) ----
        if (SCHED == 0) {
            if (TAIL == 0) stage_w<B, 0>(kt + 2);
            if (TAIL == 0) g8_wait_vm<FOUR>();       // A0, B0 of K-tile kt+1 have landed
            else if (TAIL == 1) g8_wait_vm<NA + 2>();
        } else {
            if (TAIL == 0) stage_w<B, 1>(kt + 2);
            if (TAIL == 0) g8_wait_vm<NA + 4>();     // all of K-tile kt+1 has landed
            else if (TAIL == 1) g8_wait_vm<0>();
        }
        G8_BARRIER();
        mma<1, 0>(bf0);
        G8_BARRIER();
    }

    MM_DEVICE void store(const GemmArgs& g, int m0, int n0, int wm, int wn) {
        // lane id recomputed here: nothing of the epilogue's addressing stays live across the main loop
        const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mi = 0; mi < FMH; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * (BM / 2) + h * HM + mi * 16 + fq * 4 + r;
                    if (m >= g.M) continue;
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf) {
                        const int n = n0 + wn * 64 + nf * 16 + frow;
                        if (n < g.N) g.C[(size_t)m * g.ldc + n] = f2bf(acc[h][mi][nf][r]);
                    }
                }
    }

    template <int NA>
    MM_DEVICE void run(int nk, int wm) {
        // every wave's LDS reads of a previous use of this LDS are done (persistent callers); then the prologue
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (SCHED == 0) {
            stage_a<0, 0, NA>(0); stage_w<0, 0>(0); stage_w<0, 1>(0); stage_a<0, 1, NA>(0);
            stage_a<1, 0, NA>(1); stage_w<1, 0>(1);
            g8_wait_vm<2 * NA + 4>();
        } else {
            stage_a<0, 0, NA>(0); stage_w<0, 0>(0); stage_w<0, 1>(0); stage_a<0, 1, NA>(0);
            stage_a<1, 0, NA>(1); stage_w<1, 0>(1); stage_w<1, 1>(1);
            g8_wait_vm<NA + 4>();
        }
        G8_BARRIER();
        if (STAGGER && wm == 1) G8_BARRIER();  // wave row 1 runs one barrier behind wave row 0
        for (int kt = 0; kt + 2 < nk; kt += 2) {
            tile<0, NA, 0>(kt);
            tile<1, NA, 0>(kt + 1);
        }
        tile<0, NA, 1>(nk - 2);
        tile<1, NA, 2>(nk - 1);
        if (STAGGER && wm == 0) G8_BARRIER();
    }
};

template <int BM, bool PRIO, int SCHED, bool STAGGER>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(GemmArgs g) {
    using G = Gemm8<BM, PRIO, SCHED, STAGGER>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN8 - 1) / BN8;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    constexpr int GN = 4;
    const int gsize = GN * ntm;
    const int grp_t = id / gsize, rem = id - grp_t * gsize;
    const int gn = min(GN, ntn - grp_t * GN);
    const int mt = rem / gn, nt = grp_t * GN + (rem - (rem / gn) * gn);
    const int m0 = mt * BM, n0 = nt * BN8;
    const int wm = wave >> 2, wn = wave & 3;

    G k;
    k.init(g, smem, m0, n0, wave, lane);
    const int nk = g.K / BK8;
    // the two branches never rejoin with live accumulators: each runs its own epilogue
    if (G::NPA % 8 != 0 && wave < G::NPA % 8) {
        k.template run<G::NPA / 8 + 1>(nk, wm);
        k.store(g, m0, n0, wm, wn);
    } else {
        k.template run<G::NPA / 8>(nk, wm);
        k.store(g, m0, n0, wm, wn);
    }

}

template <int BM, bool PRIO, int SCHED, bool STAGGER>
int launch8(const GemmArgs& g, hipStream_t s) {
    using G = Gemm8<BM, PRIO, SCHED, STAGGER>;
    if (g.K % 128 || g.K < 256) return mm_fail("gemm8: K must be a multiple of 128 and >= 256");
    if (g.M % 8 || g.N % 8) return mm_fail("gemm8: M and N must be multiples of 8");
    auto fn = gemm8_kernel<BM, PRIO, SCHED, STAGGER>;
    static bool attr_set = false;
    if (!attr_set) {
        MM_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        attr_set = true;
    }
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN8 - 1) / BN8;
    hipLaunchKernelGGL(fn, dim3(ntm * ntn), dim3(512), G::LDS, s, g);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace

int launch_gemm8_variant(int variant, const GemmArgs& g, hipStream_t s) {
    switch (variant) {
        case 200: return launch8<256, true, 0, true>(g, s);
        case 201: return launch8<320, true, 0, true>(g, s);
        case 202: return launch8<256, false, 0, true>(g, s);
        case 203: return launch8<256, true, 1, true>(g, s);
        case 204: return launch8<256, true, 0, false>(g, s);
        case 205: return launch8<320, true, 1, true>(g, s);
        case 206: return launch8<320, false, 0, true>(g, s);
    }
    return mm_fail("gemm8: unknown variant %d", variant);
}
