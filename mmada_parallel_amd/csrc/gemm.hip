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
//   * workgroup ids are remapped XCD-aware and in grouped order so each private L2 sees a compact patch of tiles.
//
// Epilogues reproduce the reference's rounding points exactly: every nn.Linear output is rounded to bf16 before
// anything else touches it.
#include <cstdlib>

#include "kernels.h"

namespace {

constexpr int BN = 256, BK = 64, NWAVES = 16, NTHREADS = 1024;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

MM_DEVICE void glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

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

// Tile sequence number -> (row tile, column tile): grouped order, GN column tiles x all row tiles per group, so
// workgroups with neighbouring sequence numbers (same XCD after xcd_remap) share A and W panels in their L2.
MM_DEVICE void tile_coords(int t, int ntm, int ntn, int& mt, int& nt) {
    constexpr int GN = 4;
    const int gsize = GN * ntm;
    const int grp = t / gsize, rem = t - grp * gsize;
    const int gn = min(GN, ntn - grp * GN);
    mt = rem / gn;
    nt = grp * GN + (rem - mt * gn);
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
constexpr int ZERO_ROW_ELEMS = 16384;

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

template <int EPI, int BM, int WM, int WN>
MM_DEVICE void epilogue(const GemmArgs& g, int m0, int n0,
                        f32x4 (&acc)[Tile<BM, WM, WN>::FM][Tile<BM, WM, WN>::FN], int wave, int lane) {
    using T = Tile<BM, WM, WN>;
    constexpr int TM = T::TM, TN = T::TN, FM = T::FM, FN = T::FN;
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
                if (m >= g.M) continue;
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
                    if (m >= g.M) continue;
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
                    if (m >= g.M) continue;
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
                if (mb >= g.M) continue;
                const int mbg = mb + g.m_base;  // m_base is a multiple of 8: the 4 rows still share a batch element
                const int b = mbg / g.Lp, l0 = mbg - b * g.Lp;
                if (g.pos_map) {  // scattered rows: one 2-byte store per (row, d); only the computed rows of a cache step
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (mb + r >= g.M) continue;
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
    epilogue<EPI, BM, WM, WN>(g, mt * BM, nt * BN, acc, wave, lane);
    if (g.publish) {  // hand-off to a peer that polls a counter instead of waiting on a HIP event (csrc/tp_comm.hip)
        __syncthreads();
        if (wave == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
}

template <int EPI, int BM, int WM, int WN>
int launch_cfg(const GemmArgs& g, hipStream_t s) {
    constexpr int LDS = Tile<BM, WM, WN>::LDS_BYTES;
    static bool attr_set = false;
    auto fn = gemm_bt_kernel<EPI, BM, WM, WN>;
    if (!attr_set) {
        MM_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    hipLaunchKernelGGL(fn, dim3(ntm * ntn), dim3(NTHREADS), LDS, s, g);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}

// Row-tile height.  One workgroup per CU, so a launch runs ceil(tiles / 256) rounds; a round of BM-row tiles costs
//   t(BM) = A * BM * (K/64) * h(BM)  +  C0 + C1 * BM      [us]
// (MFMA main loop, with the per-row efficiency h of the wave grid: fewer rows per wave = more LDS bytes per MFMA;
// plus pipeline fill and the BM x 256 output tile's HBM traffic, which nothing overlaps at one workgroup per CU).
// Constants are a least-squares fit (rms 6 %) to tools/bm_sweep.sh on MI355X: 240 (shape, M, BM) timings over
// K = 512..6144, N = 768..6144, M = 2440..19520; choosing by it is within 0.3 % of the best BM on that set.
// per-row cost of the 320-row tile relative to 256: fitted to tools/gemm_sweep.py on MI355X (round 2: 0.96-1.02 over
// gate/up, qkv, attn_out and down at M = 2438 / 4876)
constexpr float H320 = 1.0f;

int pick_bm(int M, int N, int K) {
    static const int forced = [] {  // MMADA_GEMM_BM=<128|160|192|224|256>: tests / sweeps force one configuration
        const char* e = getenv("MMADA_GEMM_BM");
        return e ? atoi(e) : 0;
    }();
    if (forced == 128 || forced == 160 || forced == 192 || forced == 224 || forced == 256 || forced == 320) return forced;
    // 320 x 256 (80 x 64 per wave, 2 stages = 144 KiB of LDS, 117 VGPRs): M = 2440 x N = 24576 is 768 tiles = exactly 3
    // rounds instead of 3.75 -> 4 rounds of 256-row tiles; M = 4880 x N = 4096 is 256 tiles = one round
    static const bool no320 = [] { const char* e = getenv("MMADA_GEMM_NO320"); return e && e[0] == '1'; }();
    const int cand[6] = {320, 256, 224, 192, 160, 128};
    const float h[6] = {H320, 1.0f, 1.096f, 1.125f, 1.277f, 1.236f};
    const float A = 0.00549f, C0 = 2.665f, C1 = 0.03107f;
    const int ntn = (N + BN - 1) / BN, nk = K / BK;
    int best = 256;
    float best_cost = 1e30f;
    for (int i = no320 ? 1 : 0; i < 6; ++i) {
        const int bm = cand[i];
        const int tiles = ((M + bm - 1) / bm) * ntn;
        const float cost = (float)((tiles + 255) / 256) * (A * bm * nk * h[i] + C0 + C1 * bm);
        if (cost < best_cost) { best_cost = cost; best = bm; }
    }
    return best;
}

template <int EPI>
int launch_t(const GemmArgs& g, hipStream_t s) {
    switch (pick_bm(g.M, g.N, g.K)) {
        case 320: return launch_cfg<EPI, 320, 4, 4>(g, s);
        case 256: return launch_cfg<EPI, 256, 4, 4>(g, s);
        case 192: return launch_cfg<EPI, 192, 4, 4>(g, s);
        case 128: return launch_cfg<EPI, 128, 4, 4>(g, s);
        case 160: return launch_cfg<EPI, 160, 2, 8>(g, s);
        default: return launch_cfg<EPI, 224, 2, 8>(g, s);
    }
}

}  // namespace

// one zero row per device (never freed: process lifetime), handed to every launch whose K fits
static int zero_row_for_device(const bf16_t** out) {
    static bf16_t* rows[16] = {};
    static const bool enabled = [] { const char* e = getenv("MMADA_GEMM_ZEROPAD"); return !(e && e[0] == '0'); }();
    *out = nullptr;
    if (!enabled) return 0;
    int dev = 0;
    MM_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) return 0;
    if (!rows[dev]) {
        MM_CHECK_HIP(hipMalloc(&rows[dev], ZERO_ROW_ELEMS * sizeof(bf16_t)));
        MM_CHECK_HIP(hipMemset(rows[dev], 0, ZERO_ROW_ELEMS * sizeof(bf16_t)));
    }
    *out = rows[dev];
    return 0;
}

int launch_gemm(int epi, const GemmArgs& g_in, hipStream_t s) {
    GemmArgs g = g_in;
    if (g.K <= ZERO_ROW_ELEMS && zero_row_for_device(&g.zero_row)) return 1;
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
