// gemm_var.hip — tile-shape / pipeline-depth variants of the bf16 MFMA GEMM, used to pick the production
// configuration by measurement (tools/gemm_sweep.py).  Plain C[M,N] = A[M,K]·W[N,K]^T with a bf16 store; the
// structure (LDS-DMA staging with source-side swizzle, XCD-aware grouped tile order) is the one of gemm.hip, made
// generic in <BM, BN, BK, waves, stages> with a counted-vmcnt multi-stage pipeline:
//     wait(tile t landed, (STAGES-2) tiles still in flight) ; barrier ; issue tile t+STAGES-1 ; multiply tile t
// One raw s_barrier per K-tile; the LDS-DMA loads stay in flight across barriers (guide T3/T4).
#include "kernels.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
MM_DEVICE void wait_vm_lgkm() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <int BM, int BN, int BK, int WM, int WN, int STAGES, int MINW>
__global__ __launch_bounds__(64 * WM * WN, MINW) void gemm_var_kernel(GemmArgs g) {
    constexpr int NW = WM * WN;
    constexpr int RB = BK * 2;          // bytes per LDS row
    constexpr int CPR = RB / 16;        // 16-B chunks per row
    constexpr int RPP = 64 / CPR;       // rows per 1-KiB LDS-DMA piece
    constexpr int ROWS_PER_BANKROW = 256 / RB;
    constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int PA = BM / RPP / NW, PB = BN / RPP / NW;  // pieces per wave per tile
    constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16, KK = BK / 32;
    static_assert(BM % (RPP * NW) == 0 && BN % (RPP * NW) == 0, "piece split");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    constexpr int GN = (BN >= 256) ? 4 : 8;
    const int gsize = GN * ntm;
    const int grp = id / gsize, rem = id - grp * gsize;
    const int gn = min(GN, ntn - grp * GN);
    const int mt = rem / gn, nt = grp * GN + (rem - (rem / gn) * gn);
    const int m0 = mt * BM, n0 = nt * BN;

    auto swz = [](int row) { return (row / ROWS_PER_BANKROW) % CPR; };

    const bf16_t* asrc[PA];
    const bf16_t* wsrc[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = (wave * PA + i) * RPP + lane / CPR;
        const int c = (lane % CPR) ^ swz(row);
        asrc[i] = g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + c * 8;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int row = (wave * PB + i) * RPP + lane / CPR;
        const int c = (lane % CPR) ^ swz(row);
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
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + kt * BK), (lptr_t)(base + (wave * PA + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < PB; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + kt * BK), (lptr_t)(base + A_BYTES + (wave * PB + i) * 1024),
                                             16, 0, 0);
    };

    const int frow = lane & 15, fq = lane >> 4;
    const int nk = g.K / BK;
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) stage(s, s);

    for (int kt = 0; kt < nk; ++kt) {
        // tile kt landed for this wave (tiles kt+1 .. kt+STAGES-2 may still be in flight), then everyone's did
        if (kt + STAGES - 2 < nk)
            wait_vm_lgkm<(STAGES - 2) * (PA + PB)>();
        else
            wait_vm_lgkm<0>();
        if (kt + STAGES - 1 < nk) stage((kt + STAGES - 1) % STAGES, kt + STAGES - 1);
        const char* At = smem + (kt % STAGES) * STAGE_BYTES;
        const char* Wt = At + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            bf16x8 a[FM], b[FN];
#pragma unroll
            for (int mi = 0; mi < FM; ++mi) {
                const int row = wm * TM + mi * 16 + frow;
                a[mi] = *(const bf16x8*)(At + row * RB + (((kk * 4 + fq) ^ swz(row)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < FN; ++ni) {
                const int row = wn * TN + ni * 16 + frow;
                b[ni] = *(const bf16x8*)(Wt + row * RB + (((kk * 4 + fq) ^ swz(row)) << 4));
            }
#pragma unroll
            for (int mi = 0; mi < FM; ++mi)
#pragma unroll
                for (int ni = 0; ni < FN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
    }

    const int mrow0 = m0 + wm * TM + fq * 4, ncol0 = n0 + wn * TN + frow;
#pragma unroll
    for (int mi = 0; mi < FM; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mrow0 + mi * 16 + r;
            if (m >= g.M) continue;
#pragma unroll
            for (int ni = 0; ni < FN; ++ni) {
                const int n = ncol0 + ni * 16;
                if (n < g.N) g.C[(size_t)m * g.ldc + n] = f2bf(acc[mi][ni][r]);
            }
        }
}

// Same pipeline with v_mfma_f32_32x32x16_bf16 (one ds_read_b128 per 32x16 operand fragment; C layout
// col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)).
template <int BM, int BN, int WM, int WN, int STAGES, int MINW>
__global__ __launch_bounds__(64 * WM * WN, MINW) void gemm_var32_kernel(GemmArgs g) {
    constexpr int BK = 64, NW = WM * WN, RB = 128, CPR = 8, RPP = 8;
    constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int PA = BM / RPP / NW, PB = BN / RPP / NW;
    constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 32, FN = TN / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    constexpr int GN = (BN >= 256) ? 4 : 8;
    const int gsize = GN * ntm;
    const int grp = id / gsize, rem = id - grp * gsize;
    const int gn = min(GN, ntn - grp * GN);
    const int mt = rem / gn, nt = grp * GN + (rem - (rem / gn) * gn);
    const int m0 = mt * BM, n0 = nt * BN;
    const bf16_t* asrc[PA];
    const bf16_t* wsrc[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = (wave * PA + i) * RPP + lane / CPR;
        const int c = (lane % CPR) ^ ((row >> 1) & 7);
        asrc[i] = g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + c * 8;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int row = (wave * PB + i) * RPP + lane / CPR;
        const int c = (lane % CPR) ^ ((row >> 1) & 7);
        wsrc[i] = g.W + (size_t)min(n0 + row, g.N - 1) * g.ldw + c * 8;
    }
    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < PA; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + kt * BK), (lptr_t)(base + (wave * PA + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < PB; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + kt * BK), (lptr_t)(base + A_BYTES + (wave * PB + i) * 1024),
                                             16, 0, 0);
    };
    const int frow = lane & 31, hi = lane >> 5;
    const int nk = g.K / BK;
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) stage(s, s);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + STAGES - 2 < nk)
            wait_vm_lgkm<(STAGES - 2) * (PA + PB)>();
        else
            wait_vm_lgkm<0>();
        if (kt + STAGES - 1 < nk) stage((kt + STAGES - 1) % STAGES, kt + STAGES - 1);
        const char* At = smem + (kt % STAGES) * STAGE_BYTES;
        const char* Wt = At + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 a[FM], b[FN];
#pragma unroll
            for (int mi = 0; mi < FM; ++mi) {
                const int row = wm * TM + mi * 32 + frow;
                a[mi] = *(const bf16x8*)(At + row * RB + (((ks * 2 + hi) ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < FN; ++ni) {
                const int row = wn * TN + ni * 32 + frow;
                b[ni] = *(const bf16x8*)(Wt + row * RB + (((ks * 2 + hi) ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int mi = 0; mi < FM; ++mi)
#pragma unroll
                for (int ni = 0; ni < FN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mi = 0; mi < FM; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * TM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (m >= g.M) continue;
#pragma unroll
            for (int ni = 0; ni < FN; ++ni) {
                const int n = n0 + wn * TN + ni * 32 + frow;
                if (n < g.N) g.C[(size_t)m * g.ldc + n] = f2bf(acc[mi][ni][r]);
            }
        }
}

template <int BM, int BN, int WM, int WN, int STAGES, int MINW>
int launch_var32(const GemmArgs& g, hipStream_t s) {
    constexpr int LDS = STAGES * (BM + BN) * 128;
    static bool attr_set = false;
    auto fn = gemm_var32_kernel<BM, BN, WM, WN, STAGES, MINW>;
    if (!attr_set) {
        MM_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    hipLaunchKernelGGL(fn, dim3(ntm * ntn), dim3(64 * WM * WN), LDS, s, g);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int BM, int BN, int BK, int WM, int WN, int STAGES, int MINW>
int launch_var(const GemmArgs& g, hipStream_t s) {
    constexpr int LDS = STAGES * (BM + BN) * BK * 2;
    static bool attr_set = false;
    auto fn = gemm_var_kernel<BM, BN, BK, WM, WN, STAGES, MINW>;
    if (!attr_set) {
        MM_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    hipLaunchKernelGGL(fn, dim3(ntm * ntn), dim3(64 * WM * WN), LDS, s, g);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace

int launch_gemm_variant(int variant, const GemmArgs& g, hipStream_t s) {
    if (g.K % 64) return mm_fail("gemm_variant: K must be a multiple of 64");
    switch (variant) {
        //                 BM   BN  BK WM WN ST MINW
        case 0: return launch_var<128, 128, 64, 2, 2, 2, 2>(g, s);
        case 1: return launch_var<128, 128, 32, 2, 2, 4, 2>(g, s);
        case 2: return launch_var<128, 128, 32, 2, 2, 3, 3>(g, s);
        case 3: return launch_var<256, 128, 64, 4, 2, 2, 2>(g, s);
        case 4: return launch_var<256, 256, 64, 2, 4, 2, 2>(g, s);
        case 5: return launch_var<256, 256, 32, 2, 4, 4, 2>(g, s);
        case 6: return launch_var<256, 256, 32, 2, 4, 3, 2>(g, s);
        case 7: return launch_var<256, 128, 32, 4, 2, 4, 2>(g, s);
        case 8: return launch_var<256, 128, 32, 2, 2, 3, 2>(g, s);
        case 9: return launch_var<128, 256, 32, 2, 2, 3, 2>(g, s);
        case 10: return launch_var<256, 256, 64, 4, 4, 2, 4>(g, s);
        case 11: return launch_var<256, 256, 32, 4, 4, 4, 4>(g, s);
        // 32x32x16 MFMA:    BM   BN  WM WN ST MINW
        case 12: return launch_var32<128, 128, 2, 2, 2, 2>(g, s);
        case 13: return launch_var32<256, 256, 4, 4, 2, 4>(g, s);
        case 14: return launch_var32<256, 256, 2, 4, 2, 2>(g, s);
        case 15: return launch_var32<256, 128, 4, 2, 2, 2>(g, s);
        case 16: return launch_var32<256, 128, 4, 4, 3, 4>(g, s);
        case 17: return launch_var<256, 128, 64, 4, 4, 3, 4>(g, s);
        case 18: return launch_var<256, 256, 64, 4, 2, 2, 2>(g, s);
    }
    return mm_fail("gemm_variant: unknown variant %d", variant);
}
