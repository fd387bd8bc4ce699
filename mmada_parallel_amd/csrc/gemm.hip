// gemm.hip — bf16 MFMA GEMM  C[M,N] = A[M,K] · W[N,K]^T  for gfx950 with fused epilogues.
//
// This is the contraction behind q/k/v_proj, attn_out, ff_proj/up_proj, ff_out and the LM head of the reference
// (model/modeling_llada.py:925-927, 741-744, 962-970, 1399-1404: all nn.Linear without bias, bf16 storage).
//
// Structure (picked by measurement, tools/gemm_sweep.py + csrc/gemm_var.hip; numbers in DESIGN.md §3):
//   * block tile BM x 256 x 64 with 16 waves (1024 threads, 4 waves per SIMD, one workgroup per CU); each wave
//     owns a (BM/WM) x (256/WN) sub-tile of v_mfma_f32_16x16x32_bf16 fragments, fp32 accumulate.
//   * both operands are K-contiguous, so an MFMA fragment is one ds_read_b128.  Global->LDS staging uses the
//     gfx950 LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction).  The LDS image is lane-linear, so the
//     bank-conflict swizzle (16-B chunk c -> c ^ ((row>>1)&7) inside each 128-B row) is applied to the per-lane
//     SOURCE address and to the ds_read address (guide rule 21); reads are conflict-free.
//   * two LDS stages, one raw s_barrier per K-tile: wait(tile t landed) ; barrier ; issue tile t+1 ; multiply t.
//   * BM is chosen per launch from {128,160,192,224,256,320} so that the tile grid fills the 256 CUs with the fewest
//     row-waves (M = 2438 x N = 4096 is 160 tiles of 256x256 — 62 % of the CUs — but exactly 256 tiles of 160x256).
//   * round 3: the planner at the bottom of this file also offers the 8-phase kernel of gemm8.hip (8 waves, counted
//     vmcnt, half-tile slot recycling), which is faster wherever its shape contract holds; this kernel remains for
//     K % 128 != 0, M or N not a multiple of 8, and as a cost-model candidate.
//   * workgroup ids are remapped XCD-aware and in grouped order so each private L2 sees a compact patch of tiles.
//
// Epilogues reproduce the reference's rounding points exactly: every nn.Linear output is rounded to bf16 before
// anything else touches it.
#include <cstdlib>

#include <atomic>
#include <mutex>

#include "gemm_epilogue.h"

namespace {

using namespace gemm_detail;

constexpr int BN = 256, BK = 64, NWAVES = 16, NTHREADS = 1024;

MM_DEVICE void glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

// LDS stages: 3 where 3 x (BM + 256) x 128 B fits in the 160 KiB LDS (BM <= 160), else 2.  A 160-row K-tile is
// only ~0.7 us of MFMA work, less than an HBM round trip under load, so one tile of look-ahead is not enough there.
constexpr int stages_for(int bm) { return 3 * (bm + BN) * 128 <= 160 * 1024 ? 3 : 2; }

template <int BM, int WM, int WN>
struct Tile {
    static constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
    static_assert(WM * WN == NWAVES && TM % 16 == 0 && TN % 32 == 0, "wave tile");
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int PA_TOTAL = BM / 8;  // 1-KiB LDS-DMA pieces (8 rows x 128 B) of the A tile
    static constexpr int PA = (PA_TOTAL + NWAVES - 1) / NWAVES, PB = BN / 8 / NWAVES;
    static constexpr int STAGES = stages_for(BM);
    static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
};

// acc += A[m0.., k-tiles k0..k1) · W[n0.., same k-tiles)^T.  Two LDS stages, one raw s_barrier per K-tile:
//   wait(tile t landed, my reads of tile t-1 done) ; barrier ; issue tile t+1 ; multiply tile t.
// Rows of the last row tile that lie beyond M stream from a zero row instead of re-reading row M-1: their products are
// discarded either way, but MFMAs on zeros switch far less than on live data, and this workload runs at the package
// power limit — energy not spent there is clock for the rows that count.
constexpr int ZERO_ROW_ELEMS = 8 * 16384;  // 8 rows x 16384 elements (256 KiB)

template <int BM, int WM, int WN>
MM_DEVICE void mainloop(const GemmArgs& g, char* smem, int m0, int n0, int k0, int k1,
                        f32x4 (&acc)[Tile<BM, WM, WN>::FM][Tile<BM, WM, WN>::FN], int wave, int lane) {
    using T = Tile<BM, WM, WN>;
    const int wm = wave / WN, wn = wave % WN;
    // staging addresses: wave w issues A pieces w, w+16, ... and W pieces w*PB .. w*PB+PB-1
    const bf16_t* asrc[T::PA];
    const bf16_t* wsrc[T::PB];
#pragma unroll
    for (int i = 0; i < T::PA; ++i) {
        const int piece = min(wave + i * NWAVES, T::PA_TOTAL - 1);
        const int row = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);  // logical 16-B chunk this lane fetches
        asrc[i] = (m0 + row < g.M || !g.zero_row) ? g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + c * 8 : g.zero_row + c * 8;
    }
#pragma unroll
    for (int i = 0; i < T::PB; ++i) {
        const int row = (wave * T::PB + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        wsrc[i] = g.W + (size_t)min(n0 + row, g.N - 1) * g.ldw + c * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * T::STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < T::PA; ++i)
            if (T::PA_TOTAL % NWAVES == 0 || wave + i * NWAVES < T::PA_TOTAL)  // wave-uniform
                glds16(asrc[i] + kt * BK, base + (wave + i * NWAVES) * 1024);
#pragma unroll
        for (int i = 0; i < T::PB; ++i) glds16(wsrc[i] + kt * BK, base + T::A_BYTES + (wave * T::PB + i) * 1024);
    };
    const int frow = lane & 15;  // row inside a 16-row fragment
    const int fq = lane >> 4;    // which 8-element k group of the 32-wide MFMA step

    // every wave's LDS reads of the previous tile of this workgroup are done before stage 0 is overwritten
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // this wave's LDS-DMA pieces per K-tile (the A pieces do not always divide evenly over the 16 waves)
    const bool extra_a = (T::PA_TOTAL % NWAVES != 0) && (wave < T::PA_TOTAL % NWAVES);
    stage(0, k0);
    if (T::STAGES == 3 && k0 + 1 < k1) stage(1, k0 + 1);
    for (int kt = k0; kt < k1; ++kt) {
        const int cur = (kt - k0) % T::STAGES;
        // this wave's pieces of tile kt have landed (with 3 stages tile kt+1 may still be in flight) and its LDS
        // reads of tile kt-1 are done; then everyone's
        if (T::STAGES == 3 && kt + 1 < k1) {
            constexpr int PFULL = T::PA + T::PB, PLESS = T::PA - 1 + T::PB;
            if (T::PA_TOTAL % NWAVES == 0 || extra_a)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PFULL) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PLESS) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (kt + T::STAGES - 1 < k1) stage((cur + T::STAGES - 1) % T::STAGES, kt + T::STAGES - 1);
        const char* At = smem + cur * T::STAGE_BYTES;
        const char* Wt = At + T::A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[T::FM], b[T::FN];
#pragma unroll
            for (int mi = 0; mi < T::FM; ++mi) {
                const int row = wm * T::TM + mi * 16 + frow;
                a[mi] = *(const bf16x8*)(At + row * 128 + (((kk * 4 + fq) ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < T::FN; ++ni) {
                const int row = wn * T::TN + ni * 16 + frow;
                b[ni] = *(const bf16x8*)(Wt + row * 128 + (((kk * 4 + fq) ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int mi = 0; mi < T::FM; ++mi)
#pragma unroll
                for (int ni = 0; ni < T::FN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
    }
}

// Data-parallel launch: one workgroup per output tile.
template <int EPI, int BM, int WM, int WN>
__global__ __launch_bounds__(NTHREADS, 4) void gemm_bt_kernel(GemmArgs g) {
    using T = Tile<BM, WM, WN>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    int mt, nt;
    tile_coords(xcd_remap(blockIdx.x, gridDim.x), ntm, ntn, mt, nt);
    f32x4 acc[T::FM][T::FN];
#pragma unroll
    for (int i = 0; i < T::FM; ++i)
#pragma unroll
        for (int j = 0; j < T::FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    mainloop<BM, WM, WN>(g, smem, mt * BM, nt * BN, 0, g.K / BK, acc, wave, lane);
    gemm_epilogue<EPI, T::TM, T::TN, WN>(g, mt * BM, nt * BN, acc, wave, lane, g.M);
    gemm_publish(g, wave);
}

template <int EPI, int BM, int WM, int WN>
int launch_cfg(const GemmArgs& g, hipStream_t s) {
    constexpr int LDS = Tile<BM, WM, WN>::LDS_BYTES;
    static MmOncePerDevice attr_set;
    auto fn = gemm_bt_kernel<EPI, BM, WM, WN>;
    MM_ONCE_PER_DEVICE(attr_set, MM_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)));
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    hipLaunchKernelGGL(fn, dim3(ntm * ntn), dim3(NTHREADS), LDS, s, g);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- which kernel, which tile ------------------------------------------------------------------------------------
// One workgroup per CU, so a launch runs ceil(tiles / 256) rounds; the planner prices every candidate as
//     rounds x ( main loop of one tile + fixed per-round cost )            [us]
// and takes the cheapest.  Candidates:
//   * 16-wave kernel (this file), BM in {128..320} x 256:  t = A*BM*nk*h(BM) + C0 + C1*BM — a least-squares fit (rms 6 %)
//     to tools/bm_sweep.sh on MI355X, 240 (shape, M, BM) timings over K = 512..6144, N = 768..6144, M = 2440..19520
//     (h: per-row efficiency of the wave grid — fewer rows per wave = more LDS bytes per MFMA);
//   * 8-phase kernel (gemm8.hip), BM x BN in {320x256, 256x256, 160x256, 320x128}:  t = A8*(BM*BN/256)*nk*h8 + D0 + D1*BM*BN/256,
//     fitted to tools/gemm_sweep.py (profiles/r03_gemm8_sweep*.txt).
// Both kernels accumulate a K-tile at a time in the same order with the same MFMA, so the choice never changes a bit of
// the result (tests/test_gpu_kernels.py::test_gemm_configurations_are_bit_identical).
constexpr float OLD_SCALE = 1.0f;                    // 16-wave model vs this round's measurements
// 8-phase kernel: us per (row of 256 columns x K-tile), per round.  D0 / D1 re-fitted in round 4 (6.0 / 0.03 before): the epilogues
// lost half of their instructions with the hardware fp32 -> bf16 conversion (tools/fit_gemm8_cost.py on
// profiles/r04_gemm8_sweep_final.txt; with the old constants the short-K tensor-parallel shapes went to the slower 16-wave kernel)
constexpr float A8 = 0.00483f, D0 = 2.7f, D1 = 0.025f;
constexpr float H8[GEMM8_NCFG] = {1.0f, 1.0f, 1.08f, 1.08f};
struct Plan { bool p8; int code; };  // code: GEMM8_* configuration or the 16-wave kernel's BM

std::atomic<int> g_force{-2};  // -2: read MMADA_GEMM_CFG once; -1: automatic; else gemm_force_config's code

Plan plan(const GemmArgs& g) {
    if (g_force == -2) {
        const char* e = getenv("MMADA_GEMM_CFG");  // e.g. MMADA_GEMM_CFG=1 (GEMM8_256x256) or 1160 (16-wave, BM = 160)
        g_force = e ? atoi(e) : -1;
        const char* b = getenv("MMADA_GEMM_BM");   // round-2 spelling: the 16-wave kernel with this BM
        if (!e && b) g_force = 1000 + atoi(b);
    }
    const bool can8 = gemm8_supports(g);
#ifdef MMADA_TUNE
    if (g_force >= 0 && g_force < GEMM8_NCFG + 20 && can8) return {true, g_force};
#endif
    if (g_force >= 0 && g_force < GEMM8_NCFG && can8) return {true, g_force};
    if (g_force >= 1000) {
        const int bm = g_force - 1000;
        if (bm == 128 || bm == 160 || bm == 192 || bm == 224 || bm == 256 || bm == 320) return {false, bm};
    }
    static const bool no8 = [] { const char* e = getenv("MMADA_GEMM_NO8"); return e && e[0] == '1'; }();
    const int M = g.M, N = g.N, nk = g.K / BK;
    Plan best{false, 256};
    float best_cost = 1e30f;
    {
        const int cand[6] = {320, 256, 224, 192, 160, 128};
        const float h[6] = {1.0f, 1.0f, 1.096f, 1.125f, 1.277f, 1.236f};
        const float A = 0.00549f, C0 = 2.665f, C1 = 0.03107f;
        const int ntn = (N + BN - 1) / BN;
        for (int i = 0; i < 6; ++i) {
            const int bm = cand[i];
            const int tiles = ((M + bm - 1) / bm) * ntn;
            const float cost = (float)((tiles + 255) / 256) * (A * bm * nk * h[i] + C0 + C1 * bm) * OLD_SCALE;
            if (cost < best_cost) { best_cost = cost; best = {false, bm}; }
        }
    }
    if (can8 && !no8) {
        const int bm8[GEMM8_NCFG] = {320, 256, 160, 320}, bn8[GEMM8_NCFG] = {256, 256, 256, 128};
        for (int c = 0; c < GEMM8_NCFG; ++c) {
            const int tiles = ((M + bm8[c] - 1) / bm8[c]) * ((N + bn8[c] - 1) / bn8[c]);
            const float area = bm8[c] * (bn8[c] / 256.0f);  // in rows of a 256-column tile
            const float cost = (float)((tiles + 255) / 256) * (A8 * area * nk * H8[c] + D0 + D1 * area);
            if (cost < best_cost) { best_cost = cost; best = {true, c}; }
        }
    }
    return best;
}

template <int EPI>
int launch_t(const GemmArgs& g, hipStream_t s) {
    const Plan p = plan(g);
    if (p.p8) return launch_gemm8(EPI, p.code, g, s);
    switch (p.code) {
        case 320: return launch_cfg<EPI, 320, 4, 4>(g, s);
        case 256: return launch_cfg<EPI, 256, 4, 4>(g, s);
        case 192: return launch_cfg<EPI, 192, 4, 4>(g, s);
        case 128: return launch_cfg<EPI, 128, 4, 4>(g, s);
        case 160: return launch_cfg<EPI, 160, 2, 8>(g, s);
        default: return launch_cfg<EPI, 224, 2, 8>(g, s);
    }
}

}  // namespace

void gemm_force_config(int code) { g_force = code; }

// what the planner would launch for a plain [M, K] x [N, K]^T product (host arithmetic only): 0..3 = GEMM8_* configuration,
// 1000 + BM = the 16-wave kernel — tests/test_host_logic.py holds it against the committed sweep table
int gemm_plan_code(int M, int N, int K) {
    GemmArgs g{};
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N;
    if (M <= 0 || N <= 0 || K <= 0 || K % BK) return -1;
    const Plan p = plan(g);
    return p.p8 ? p.code : 1000 + p.code;
}

// 8 zero rows of up to ZERO_ROW_ELEMS / 8 elements per device (never freed: process lifetime): the source of the A rows of
// the last row tile that lie beyond M
static int zero_rows_for_device(const bf16_t** out) {
    static std::atomic<bf16_t*> rows[16];
    static std::mutex mu;
    static const bool enabled = [] { const char* e = getenv("MMADA_GEMM_ZEROPAD"); return !(e && e[0] == '0'); }();
    *out = nullptr;
    if (!enabled) return 0;
    int dev = 0;
    MM_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) return 0;
    bf16_t* have = rows[dev].load(std::memory_order_acquire);
    if (!have) {   // normally built by gemm_prepare_device (mmada_create); a bare mmada_gemm_bt call gets here
        std::lock_guard<std::mutex> lock(mu);
        have = rows[dev].load(std::memory_order_acquire);
        if (!have) {
            bf16_t* p = nullptr;
            MM_CHECK_HIP(hipMalloc(&p, ZERO_ROW_ELEMS * sizeof(bf16_t)));
            MM_CHECK_HIP(hipMemset(p, 0, ZERO_ROW_ELEMS * sizeof(bf16_t)));
            rows[dev].store(p, std::memory_order_release);
            have = p;
        }
    }
    *out = have;
    return 0;
}

// ---- the SiLU table of the SwiGLU epilogue (gemm_epilogue.h: SiluLut), one per device, filled by the function it replaces ----
__global__ void silu_lut_kernel(uint16_t* t) {
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (unsigned)gemm_detail::SiluLut::ENTRIES) return;
    const unsigned key = idx >> 1, sign = idx & 1u;
    uint16_t v = 0;
    if (key < gemm_detail::SiluLut::NKEY) {
        const unsigned bits = (sign << 15) | (key + gemm_detail::SiluLut::E0);
        v = f2bf(gemm_detail::silu_bf16(__uint_as_float(bits << 16)));
    }
    t[idx] = v;
}
static std::atomic<int> g_silu_lut{-1};  // -1: read MMADA_GEMM_SILU_LUT once (default on)
static int silu_lut_for_device(const uint16_t** out, hipStream_t s) {
    static std::atomic<uint16_t*> lut[16];
    static std::mutex mu;
    *out = nullptr;
    if (g_silu_lut.load(std::memory_order_relaxed) < 0) {
        const char* e = getenv("MMADA_GEMM_SILU_LUT");
        g_silu_lut.store(e && e[0] == '0' ? 0 : 1, std::memory_order_relaxed);
    }
    if (!g_silu_lut.load(std::memory_order_relaxed)) return 0;
    int dev = 0;
    MM_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) return 0;
    uint16_t* have = lut[dev].load(std::memory_order_acquire);
    if (!have) {   // first SwiGLU launch on this device: allocate and fill once (two host threads: one does it, the other waits)
        // A hipMalloc + synchronise cannot run under hipGraph capture (e.g. the option switched on after the eager warm-up
        // step): that launch evaluates SiLU instead — the same bits — and a later eager launch builds the table.
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            return 0;
        }
        std::lock_guard<std::mutex> lock(mu);
        have = lut[dev].load(std::memory_order_acquire);
        if (!have) {
            uint16_t* p = nullptr;
            MM_CHECK_HIP(hipMalloc(&p, gemm_detail::SiluLut::BYTES));
            hipLaunchKernelGGL(silu_lut_kernel, dim3((gemm_detail::SiluLut::ENTRIES + 255) / 256), dim3(256), 0, s, p);   // ordered before the GEMM
            MM_CHECK_HIP(hipGetLastError());
            MM_CHECK_HIP(hipStreamSynchronize(s));   // once per device: another stream's first SwiGLU launch must not overtake the fill
            lut[dev].store(p, std::memory_order_release);
            have = p;
        }
    }
    *out = have;
    return 0;
}
// Everything a GEMM launch would otherwise allocate lazily (the zero rows, the SiLU table: a hipMalloc and a stream
// synchronise at the FIRST launch on a device) built up front, from mmada_create: a launch then never blocks the host — a
// tensor-parallel rank group driven by one host thread deadlocks (until the hand-off timeout) if the first SwiGLU launch of the
// process synchronises rank 0's stream while rank 1's kernels are not enqueued yet.
int gemm_prepare_device() {
    const bf16_t* z = nullptr;
    if (zero_rows_for_device(&z)) return 1;
    const uint16_t* lut = nullptr;
    return silu_lut_for_device(&lut, (hipStream_t)0);
}

void gemm_set_silu_lut(int on) { g_silu_lut.store(on != 0 ? 1 : 0, std::memory_order_relaxed); }

int launch_gemm(int epi, const GemmArgs& g_in, hipStream_t s) {
    GemmArgs g = g_in;
    g.silu_lut = nullptr;
    if (epi == EPI_SWIGLU && silu_lut_for_device(&g.silu_lut, s)) return 1;
    // the 16-wave kernel reads K zeros, the 8-phase kernel a block of 8 rows x lda
    if (g.K <= ZERO_ROW_ELEMS / 8 && g.lda <= ZERO_ROW_ELEMS / 8 && zero_rows_for_device(&g.zero_row)) return 1;
    if (g.M <= 0 || g.N <= 0) return 0;
    if (g.K % BK != 0 || g.K <= 0) return mm_fail("gemm: K=%d must be a positive multiple of %d", g.K, BK);
    if ((g.lda % 8) || (g.ldw % 8)) return mm_fail("gemm: lda/ldw must be multiples of 8 elements");
    switch (epi) {
        case EPI_STORE: return launch_t<EPI_STORE>(g, s);
        case EPI_RESID: return launch_t<EPI_RESID>(g, s);
        case EPI_SWIGLU:
            if (g.N % 64) return mm_fail("gemm/swiglu: N must be a multiple of 64");
            return launch_t<EPI_SWIGLU>(g, s);
        case EPI_QKV:
            if (g.N != (g.Hq + 2 * g.Hkv) * 128) return mm_fail("gemm/qkv: N mismatch");
            if (g.Lp % 8 || g.m_base % 8) return mm_fail("gemm/qkv: Lp and m_base must be multiples of 8");
            return launch_t<EPI_QKV>(g, s);
    }
    return mm_fail("gemm: bad epilogue %d", epi);
}
