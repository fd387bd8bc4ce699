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
//   * BM is chosen per launch from {128,160,192,224,256} so that the tile grid fills the 256 CUs with the fewest
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

template <int EPI, int BM, int WM, int WN>
__global__ __launch_bounds__(NTHREADS, 4) void gemm_bt_kernel(GemmArgs g) {
    static_assert(WM * WN == NWAVES, "16 waves");
    constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
    static_assert(TM % 16 == 0 && TN % 32 == 0, "wave tile");
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int PA_TOTAL = BM / 8;               // 1-KiB LDS-DMA pieces (8 rows x 128 B) of the A tile
    constexpr int PA = (PA_TOTAL + NWAVES - 1) / NWAVES, PB = BN / 8 / NWAVES;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    // XCD-aware + grouped tile order: GN column tiles x all row tiles form one group
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    constexpr int GN = 4;
    const int gsize = GN * ntm;
    const int grp = id / gsize, rem = id - grp * gsize;
    const int gn = min(GN, ntn - grp * GN);
    const int mt = rem / gn, nt = grp * GN + (rem - (rem / gn) * gn);
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- staging addresses: wave w issues A pieces w, w+16, ... and W pieces w*PB .. w*PB+PB-1 ----
    const bf16_t* asrc[PA];
    const bf16_t* wsrc[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int piece = min(wave + i * NWAVES, PA_TOTAL - 1);
        const int row = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);  // logical 16-B chunk this lane fetches
        asrc[i] = g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + c * 8;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int row = (wave * PB + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        wsrc[i] = g.W + (size_t)min(n0 + row, g.N - 1) * g.ldw + c * 8;
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < PA; ++i)
            if (PA_TOTAL % NWAVES == 0 || wave + i * NWAVES < PA_TOTAL)  // wave-uniform
                glds16(asrc[i] + kt * BK, base + (wave + i * NWAVES) * 1024);
#pragma unroll
        for (int i = 0; i < PB; ++i) glds16(wsrc[i] + kt * BK, base + A_BYTES + (wave * PB + i) * 1024);
    };

    const int frow = lane & 15;  // row inside a 16-row fragment
    const int fq = lane >> 4;    // which 8-element k group of the 32-wide MFMA step

    const int nk = g.K / BK;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's pieces of tile kt have landed and its LDS reads of tile kt-1 are done; then everyone's
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
        const char* At = smem + (kt & 1) * STAGE_BYTES;
        const char* Wt = At + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[FM], b[FN];
#pragma unroll
            for (int mi = 0; mi < FM; ++mi) {
                const int row = wm * TM + mi * 16 + frow;
                a[mi] = *(const bf16x8*)(At + row * 128 + (((kk * 4 + fq) ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < FN; ++ni) {
                const int row = wn * TN + ni * 16 + frow;
                b[ni] = *(const bf16x8*)(Wt + row * 128 + (((kk * 4 + fq) ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int mi = 0; mi < FM; ++mi)
#pragma unroll
                for (int ni = 0; ni < FN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
    }

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
#pragma unroll
                for (int ni = 0; ni < FN; ++ni) {
                    const int n = wcol0 + ni * 16 + frow;
                    if (n >= g.N) continue;
                    float v = acc[mi][ni][r];
                    if constexpr (EPI == EPI_RESID) {
                        v = bfround(v);
                        if (g.add_resid) v = bf2f(g.resid[(size_t)m * g.ldr + n]) + v;
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
                    const int b = m / g.Lp, l = m - b * g.Lp;
                    bf16_t* row = dst + ((size_t)(b * nh + hh) * g.Lkv + l) * 128;
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
                const int b = mb / g.Lp, l0 = mb - b * g.Lp;
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

template <int EPI, int BM, int WM, int WN>
int launch_cfg(const GemmArgs& g, hipStream_t s) {
    constexpr int LDS = 2 * (BM + BN) * 128;
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

// Row-tile height: one workgroup per CU, so a launch costs ceil(tiles / 256) row-waves of BM rows each.
// 2x8 wave grids (BM = 160, 224) move ~25 % more LDS bytes per MFMA than 4x4: small handicap.
int pick_bm(int M, int N) {
    static const int forced = [] {  // MMADA_GEMM_BM=<128|160|192|224|256>: tests / sweeps force one configuration
        const char* e = getenv("MMADA_GEMM_BM");
        return e ? atoi(e) : 0;
    }();
    if (forced == 128 || forced == 160 || forced == 192 || forced == 224 || forced == 256) return forced;
    const int cand[5] = {256, 192, 128, 160, 224};
    const float handicap[5] = {1.0f, 1.0f, 1.04f, 1.08f, 1.08f};
    const int ntn = (N + BN - 1) / BN;
    int best = 256;
    float best_cost = 1e30f;
    for (int i = 0; i < 5; ++i) {
        const int bm = cand[i];
        const int tiles = ((M + bm - 1) / bm) * ntn;
        const float cost = (float)((tiles + 255) / 256) * bm * handicap[i];
        if (cost < best_cost) { best_cost = cost; best = bm; }
    }
    return best;
}

template <int EPI>
int launch_t(const GemmArgs& g, hipStream_t s) {
    switch (pick_bm(g.M, g.N)) {
        case 256: return launch_cfg<EPI, 256, 4, 4>(g, s);
        case 192: return launch_cfg<EPI, 192, 4, 4>(g, s);
        case 128: return launch_cfg<EPI, 128, 4, 4>(g, s);
        case 160: return launch_cfg<EPI, 160, 2, 8>(g, s);
        default: return launch_cfg<EPI, 224, 2, 8>(g, s);
    }
}

}  // namespace

int launch_gemm(int epi, const GemmArgs& g, hipStream_t s) {
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
            if (g.Lp % 8) return mm_fail("gemm/qkv: Lp must be a multiple of 8");
            return launch_t<EPI_QKV>(g, s);
    }
    return mm_fail("gemm: bad epilogue %d", epi);
}
