// gemm_var.hip — tile-shape / pipeline-depth variants of the bf16 MFMA GEMM, used to pick the production
// configuration by measurement (tools/gemm_sweep.py).  Plain C[M,N] = A[M,K]·W[N,K]^T with a bf16 store; the
// structure (LDS-DMA staging with source-side swizzle, XCD-aware grouped tile order) is the one of gemm.hip, made
// generic in <BM, BN, BK, waves, stages> with a counted-vmcnt multi-stage pipeline:
//     wait(tile t landed, (STAGES-2) tiles still in flight) ; barrier ; issue tile t+STAGES-1 ; multiply tile t
// One raw s_barrier per K-tile; the LDS-DMA loads stay in flight across barriers (guide T3/T4).
#include <algorithm>
#include "kernels.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
MM_DEVICE void wait_vm_lgkm() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// SPREAD != 0 (BK = 64 only): the LDS-DMA pieces of the next K-tile are not issued in one burst behind the barrier but
// one at a time between groups of FN MFMAs (1: during the first 32-deep half, 2: during the second half).
template <int BM, int BN, int BK, int WM, int WN, int STAGES, int MINW, bool NOSTORE = false, bool NODMA = false, bool NOLDS = false, int SPREAD = 0>
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
    // SPREAD == 3: waves 0..(BM+BN)/64-1 own 64 rows each of the A|W tile for the L2 touch (BK*2 = 128 B = one line/row)
    const bool pf_wave = SPREAD == 3 && wave < (BM + BN) / 64;
    const bf16_t* pf_ptr = g.A;
    float pf_sink = 0.f;
    if (pf_wave) {
        const int r = wave * 64 + lane;
        pf_ptr = r < BM ? g.A + (size_t)min(m0 + r, g.M - 1) * g.lda : g.W + (size_t)min(n0 + r - BM, g.N - 1) * g.ldw;
    }
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) stage(s, s);

    for (int kt = 0; kt < nk; ++kt) {
        // tile kt landed for this wave (tiles kt+1 .. kt+STAGES-2 may still be in flight), then everyone's did
        if (SPREAD == 3 && pf_wave && kt >= 1 && kt - 1 + STAGES < nk)
            wait_vm_lgkm<(STAGES - 2) * (PA + PB) + 1>();  // the newest outstanding op is last iteration's L2 touch
        else if (kt + STAGES - 2 < nk)
            wait_vm_lgkm<(STAGES - 2) * (PA + PB)>();
        else
            wait_vm_lgkm<0>();
        // NODMA probe: only the first tiles are ever fetched (wrong results; isolates the cost of the global->LDS stream)
        const bool more = kt + STAGES - 1 < nk;
        if ((SPREAD == 0 || SPREAD == 3) && more && (!NODMA || kt < 2)) stage((kt + STAGES - 1) % STAGES, kt + STAGES - 1);
        if (SPREAD == 3 && pf_wave && kt + STAGES < nk)  // touch the K-tile after the one being staged: one line per row
            asm volatile("global_load_dword %0, %1, off" : "+v"(pf_sink) : "v"(pf_ptr + (size_t)(kt + STAGES) * BK) : "memory");
        const char* At = smem + (kt % STAGES) * STAGE_BYTES;
        const char* Wt = At + A_BYTES;
        auto piece = [&](int i) {  // i-th LDS-DMA piece of this wave for K-tile kt + STAGES - 1
            char* base = smem + ((kt + STAGES - 1) % STAGES) * STAGE_BYTES;
            const int kn = kt + STAGES - 1;
            if (i < PA)
                __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + kn * BK), (lptr_t)(base + (wave * PA + i) * 1024), 16, 0, 0);
            else
                __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i - PA] + kn * BK),
                                                 (lptr_t)(base + A_BYTES + (wave * PB + i - PA) * 1024), 16, 0, 0);
        };
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
            for (int mi = 0; mi < FM; ++mi) {
#pragma unroll
                for (int ni = 0; ni < FN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
                if (SPREAD != 0 && kk == SPREAD - 1) {
                    // pieces mi, mi + FM, ... of this wave go out in the shadow of the MFMA group just issued
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) {
#pragma unroll
                        for (int i = mi; i < PA + PB; i += FM) piece(i);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
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

// 1.5 K-tiles of look-ahead inside 160 KiB at 256x256: the ring holds FIVE 32-deep half-tiles (5 x 32 KiB); every
// iteration multiplies two of them (one 64-deep K-tile, one barrier) while three are in flight or landed.
//   wait(halves 2t, 2t+1 landed; half 2t+2 may be in flight) ; barrier ; issue halves 2t+3, 2t+4 ; multiply
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, 4) void gemm_half_kernel(GemmArgs g) {
    constexpr int NW = WM * WN, BKH = 32, RB = 64, CPR = 4, RPP = 16, NSLOT = 5;
    constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB, SLOT_BYTES = A_BYTES + B_BYTES;
    constexpr int PA = BM / RPP / NW, PB = BN / RPP / NW;
    constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
    static_assert(BM % (RPP * NW) == 0 && BN % (RPP * NW) == 0, "piece split");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    constexpr int GN = 4;
    const int gsize = GN * ntm;
    const int grp = id / gsize, rem = id - grp * gsize;
    const int gn = min(GN, ntn - grp * GN);
    const int mt = rem / gn, nt = grp * GN + (rem - (rem / gn) * gn);
    const int m0 = mt * BM, n0 = nt * BN;
    auto swz = [](int row) { return (row >> 2) & 3; };  // 4 rows of 64 B per 256-B bank row
    const bf16_t* asrc[PA];
    const bf16_t* wsrc[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = (wave * PA + i) * RPP + lane / CPR;
        asrc[i] = g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + ((lane % CPR) ^ swz(row)) * 8;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int row = (wave * PB + i) * RPP + lane / CPR;
        wsrc[i] = g.W + (size_t)min(n0 + row, g.N - 1) * g.ldw + ((lane % CPR) ^ swz(row)) * 8;
    }
    auto issue = [&](int hf) {  // half-tile hf -> slot hf % 5
        char* base = smem + (hf % NSLOT) * SLOT_BYTES;
#pragma unroll
        for (int i = 0; i < PA; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + hf * BKH), (lptr_t)(base + (wave * PA + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < PB; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + hf * BKH), (lptr_t)(base + A_BYTES + (wave * PB + i) * 1024),
                                             16, 0, 0);
    };
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fq = lane >> 4;
    const int nh = g.K / BKH, nk = nh / 2;
    issue(0);
    issue(1);
    if (nh > 2) issue(2);
    for (int kt = 0; kt < nk; ++kt) {
        if (2 * kt + 2 < nh)
            wait_vm_lgkm<PA + PB>();
        else
            wait_vm_lgkm<0>();
        if (2 * kt + 3 < nh) issue(2 * kt + 3);
        if (2 * kt + 4 < nh) issue(2 * kt + 4);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const char* At = smem + ((2 * kt + hh) % NSLOT) * SLOT_BYTES;
            const char* Wt = At + A_BYTES;
            bf16x8 a[FM], b[FN];
#pragma unroll
            for (int mi = 0; mi < FM; ++mi) {
                const int row = wm * TM + mi * 16 + frow;
                a[mi] = *(const bf16x8*)(At + row * RB + ((fq ^ swz(row)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < FN; ++ni) {
                const int row = wn * TN + ni * 16 + frow;
                b[ni] = *(const bf16x8*)(Wt + row * RB + ((fq ^ swz(row)) << 4));
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

template <int BM, int BN, int WM, int WN>
int launch_half(const GemmArgs& g, hipStream_t s) {
    constexpr int LDS = 5 * (BM + BN) * 64;
    static bool attr_set = false;
    auto fn = gemm_half_kernel<BM, BN, WM, WN>;
    if (!attr_set) {
        MM_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    hipLaunchKernelGGL(fn, dim3(ntm * ntn), dim3(64 * WM * WN), LDS, s, g);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}

// Persistent data-parallel form of gemm_var_kernel: one workgroup per CU walks the tile sequence with stride gridDim
// (same XCD-contiguous, K-lockstep order as the one-tile-per-workgroup launch); with PREFETCH the first STAGES-1
// K-tiles of the NEXT output tile are requested before the epilogue stores of the current one.
template <int BM, int BN, int BK, int WM, int WN, int STAGES, int MINW, bool PREFETCH>
__global__ __launch_bounds__(64 * WM * WN, MINW) void gemm_pers_kernel(GemmArgs g) {
    constexpr int NW = WM * WN;
    constexpr int RB = BK * 2, CPR = RB / 16, RPP = 64 / CPR, ROWS_PER_BANKROW = 256 / RB;
    constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int PA = BM / RPP / NW, PB = BN / RPP / NW;
    constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16, KK = BK / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN, ntiles = ntm * ntn;
    constexpr int GN = (BN >= 256) ? 4 : 8;
    auto swz = [](int row) { return (row / ROWS_PER_BANKROW) % CPR; };
    const int frow = lane & 15, fq = lane >> 4;
    const int nk = g.K / BK;

    const bf16_t* asrc[PA];
    const bf16_t* wsrc[PB];
    int m0 = 0, n0 = 0;
    auto locate = [&](int tile, int& tm0, int& tn0) {
        const int id = xcd_remap(tile, ntiles);
        const int gsize = GN * ntm;
        const int grp = id / gsize, rem = id - grp * gsize;
        const int gn = min(GN, ntn - grp * GN);
        tm0 = (rem / gn) * BM;
        tn0 = (grp * GN + (rem - (rem / gn) * gn)) * BN;
    };
    auto point = [&](int tm0, int tn0) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int row = (wave * PA + i) * RPP + lane / CPR;
            asrc[i] = g.A + (size_t)min(tm0 + row, g.M - 1) * g.lda + ((lane % CPR) ^ swz(row)) * 8;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int row = (wave * PB + i) * RPP + lane / CPR;
            wsrc[i] = g.W + (size_t)min(tn0 + row, g.N - 1) * g.ldw + ((lane % CPR) ^ swz(row)) * 8;
        }
    };
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

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    locate(tile, m0, n0);
    point(m0, n0);
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) stage(s, s);

    while (true) {
        f32x4 acc[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + STAGES - 2 < nk && !(PREFETCH && kt == 0))
                wait_vm_lgkm<(STAGES - 2) * (PA + PB)>();
            else
                wait_vm_lgkm<0>();  // also drains the previous tile's epilogue stores (gfx9: one vmcnt for loads and stores)
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
        const int cm0 = m0, cn0 = n0;
        tile += gridDim.x;
        const bool more = tile < ntiles;
        __syncthreads();  // every wave has read the last K-tile: the ring may be refilled
        if (more) {
            locate(tile, m0, n0);
            point(m0, n0);
            if (PREFETCH) {
#pragma unroll
                for (int s = 0; s < STAGES - 1; ++s)
                    if (s < nk) stage(s, s);
            }
        }
        const int mrow0 = cm0 + wm * TM + fq * 4, ncol0 = cn0 + wn * TN + frow;
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
        if (!more) break;
        if (!PREFETCH) {
#pragma unroll
            for (int s = 0; s < STAGES - 1; ++s)
                if (s < nk) stage(s, s);
        }
    }
}

template <int BM, int BN, int BK, int WM, int WN, int STAGES, int MINW, bool PREFETCH>
int launch_pers(const GemmArgs& g, hipStream_t s) {
    constexpr int LDS = STAGES * (BM + BN) * BK * 2;
    static bool attr_set = false;
    auto fn = gemm_pers_kernel<BM, BN, BK, WM, WN, STAGES, MINW, PREFETCH>;
    if (!attr_set) {
        MM_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int ntiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    hipLaunchKernelGGL(fn, dim3(std::min(ntiles, 256)), dim3(64 * WM * WN), LDS, s, g);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}

// Same pipeline with v_mfma_f32_32x32x16_bf16 (one ds_read_b128 per 32x16 operand fragment; C layout
// col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)).
template <int BM, int BN, int WM, int WN, int STAGES, int MINW, bool PROBE = false>
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
        if (kt + STAGES - 1 < nk && (!PROBE || kt < 2)) stage((kt + STAGES - 1) % STAGES, kt + STAGES - 1);
        const char* At = smem + (kt % STAGES) * STAGE_BYTES;
        const char* Wt = At + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 a[FM], b[FN];
#pragma unroll
            for (int mi = 0; mi < FM; ++mi) {
                const int row = wm * TM + mi * 32 + frow;
                if (!PROBE || kt == 0) a[mi] = *(const bf16x8*)(At + row * RB + (((ks * 2 + hi) ^ ((row >> 1) & 7)) << 4));
                if (PROBE) asm volatile("" : "+v"(a[mi]));
            }
#pragma unroll
            for (int ni = 0; ni < FN; ++ni) {
                const int row = wn * TN + ni * 32 + frow;
                if (!PROBE || kt == 0) b[ni] = *(const bf16x8*)(Wt + row * RB + (((ks * 2 + hi) ^ ((row >> 1) & 7)) << 4));
                if (PROBE) asm volatile("" : "+v"(b[ni]));
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

template <int BM, int BN, int WM, int WN, int STAGES, int MINW, bool PROBE = false>
int launch_var32(const GemmArgs& g, hipStream_t s) {
    constexpr int LDS = STAGES * (BM + BN) * 128;
    static bool attr_set = false;
    auto fn = gemm_var32_kernel<BM, BN, WM, WN, STAGES, MINW, PROBE>;
    if (!attr_set) {
        MM_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    hipLaunchKernelGGL(fn, dim3(ntm * ntn), dim3(64 * WM * WN), LDS, s, g);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}

// "Big wave" structure: 256x256 block, 4 waves (one per SIMD, 512-register budget), each wave a 128x128 sub-tile
// (8x8 fragments, 256 accumulator registers); K streamed in 32-deep slices through a ring of RING LDS buffers with
// counted vmcnt; the fragments of slice t+1 are read into a second register set while slice t is multiplied.
template <int RING, bool PREFETCH>
__global__ __launch_bounds__(256, 1) void gemm_big_kernel(GemmArgs g) {
    constexpr int BM = 256, BN = 256, BK = 32, RB = 64, CPR = 4, RPP = 16, NW = 4;
    constexpr int A_BYTES = BM * RB, STAGE_BYTES = (BM + BN) * RB;  // 32 KiB per slice
    constexpr int PA = BM / RPP / NW, PB = BN / RPP / NW;           // 4 + 4 pieces per wave per slice
    constexpr int P = PA + PB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    constexpr int GN = 4;
    const int gsize = GN * ntm;
    const int grp = id / gsize, rem = id - grp * gsize;
    const int gn = min(GN, ntn - grp * GN);
    const int mt = rem / gn, nt = grp * GN + (rem - (rem / gn) * gn);
    const int m0 = mt * BM, n0 = nt * BN;
    auto swz = [](int row) { return (row >> 2) & 3; };
    const bf16_t* asrc[PA];
    const bf16_t* wsrc[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = (wave * PA + i) * RPP + lane / CPR;
        asrc[i] = g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + ((lane % CPR) ^ swz(row)) * 8;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int row = (wave * PB + i) * RPP + lane / CPR;
        wsrc[i] = g.W + (size_t)min(n0 + row, g.N - 1) * g.ldw + ((lane % CPR) ^ swz(row)) * 8;
    }
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto stage = [&](int kt) {
        char* base = smem + (kt % RING) * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < PA; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + kt * BK), (lptr_t)(base + (wave * PA + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < PB; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + kt * BK), (lptr_t)(base + A_BYTES + (wave * PB + i) * 1024),
                                             16, 0, 0);
    };
    const int frow = lane & 15, fq = lane >> 4;
    int aoff[8], boff[8];  // byte offsets of this lane's fragments inside a slice
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ra = wm * 128 + i * 16 + frow, rb = wn * 128 + i * 16 + frow;
        aoff[i] = ra * RB + ((fq ^ swz(ra)) << 4);
        boff[i] = A_BYTES + rb * RB + ((fq ^ swz(rb)) << 4);
    }
    auto load_frags = [&](int kt, bf16x8 (&a)[8], bf16x8 (&b)[8]) {
        const char* base = smem + (kt % RING) * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = *(const bf16x8*)(base + aoff[i]);
#pragma unroll
        for (int i = 0; i < 8; ++i) b[i] = *(const bf16x8*)(base + boff[i]);
    };
    auto mma = [&](bf16x8 (&a)[8], bf16x8 (&b)[8]) {
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < 8; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    };
    const int nk = g.K / BK;
    constexpr int AHEAD = RING - 1;  // slices in flight beyond the one being multiplied
#pragma unroll
    for (int s = 0; s < AHEAD; ++s)
        if (s < nk) stage(s);
    bf16x8 a0[8], b0[8], a1[8], b1[8];
    if constexpr (PREFETCH) {
        // certify slice 0, read its fragments
        if (AHEAD - 1 < nk) wait_vm_lgkm<(AHEAD - 1) * P>(); else wait_vm_lgkm<0>();
        load_frags(0, a0, b0);
        for (int kt = 0; kt < nk; kt += 2) {
            // step kt: slices <= kt+1 landed (own pieces), then everyone's; ring slot of slice kt-1 is free
            if (kt + AHEAD - 1 < nk) wait_vm_lgkm<(AHEAD - 2) * P>(); else wait_vm_lgkm<0>();
            if (kt + AHEAD < nk) stage(kt + AHEAD);
            if (kt + 1 < nk) load_frags(kt + 1, a1, b1);
            mma(a0, b0);
            if (kt + 1 >= nk) break;
            if (kt + AHEAD < nk) wait_vm_lgkm<(AHEAD - 2) * P>(); else wait_vm_lgkm<0>();
            if (kt + 1 + AHEAD < nk) stage(kt + 1 + AHEAD);
            if (kt + 2 < nk) load_frags(kt + 2, a0, b0);
            mma(a1, b1);
        }
    } else {
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + AHEAD - 1 < nk) wait_vm_lgkm<(AHEAD - 1) * P>(); else wait_vm_lgkm<0>();
            if (kt + AHEAD < nk) stage(kt + AHEAD);
            load_frags(kt, a0, b0);
            mma(a0, b0);
        }
    }
    const int mrow0 = m0 + wm * 128 + fq * 4, ncol0 = n0 + wn * 128 + frow;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mrow0 + mi * 16 + r;
            if (m >= g.M) continue;
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
                const int n = ncol0 + ni * 16;
                if (n < g.N) g.C[(size_t)m * g.ldc + n] = f2bf(acc[mi][ni][r]);
            }
        }
}

// "Ping-pong" structure: 256x256x64 tile, 8 waves = two groups of four (one wave of each group per SIMD).  Every
// K-tile is 8 barrier-separated phases per wave, alternating R (LDS fragment reads + LDS-DMA issue for the next
// K-tile) and C (16 MFMAs on one 64x32 quadrant of the wave's 128x64 sub-tile).  Group B runs one phase behind
// group A, so while one wave of a SIMD streams MFMAs its partner does its LDS traffic.
#define PP_BARRIER()                                   \
    do {                                               \
        __builtin_amdgcn_sched_barrier(0);             \
        asm volatile("s_barrier" ::: "memory");        \
        __builtin_amdgcn_sched_barrier(0);             \
    } while (0)
#define PP_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define PP_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

template <bool SETPRIO, int GRPMODE>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(GemmArgs g) {
    constexpr int BM = 256, BN = 256, BK = 64, A_BYTES = BM * 128, STAGE_BYTES = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 0: group A, 1: group B (one phase behind); the two groups must share SIMDs pairwise — which wave-id bit
    // separates SIMD partners is a dispatch detail, so it is a measured choice (GRPMODE)
    const int grp = GRPMODE == 0 ? (wave >> 2) : GRPMODE == 1 ? (wave & 1) : ((wave >> 1) & 1);
    const int wn = GRPMODE == 0 ? (wave & 3) : GRPMODE == 1 ? (wave >> 1) : ((wave & 1) | ((wave >> 2) << 1));
    const int wm = grp;  // wave tile: rows wm*128.., cols wn*64..
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    constexpr int GN = 4;
    const int gsize = GN * ntm;
    const int grp_t = id / gsize, rem = id - grp_t * gsize;
    const int gn = min(GN, ntn - grp_t * GN);
    const int mt = rem / gn, nt = grp_t * GN + (rem - (rem / gn) * gn);
    const int m0 = mt * BM, n0 = nt * BN;

    // LDS-DMA: 64 pieces (1 KiB = 8 rows x 128 B) per K-tile, wave w moves A pieces 4w..4w+3 and W pieces 4w..4w+3
    const bf16_t* asrc[4];
    const bf16_t* wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        asrc[i] = g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + c * 8;
        wsrc[i] = g.W + (size_t)min(n0 + row, g.N - 1) * g.ldw + c * 8;
    }
    auto dma_a = [&](int buf, int kt) {
        char* base = smem + buf * STAGE_BYTES + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + kt * BK), (lptr_t)(base + i * 1024), 16, 0, 0);
    };
    auto dma_w = [&](int buf, int kt) {
        char* base = smem + buf * STAGE_BYTES + A_BYTES + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + kt * BK), (lptr_t)(base + i * 1024), 16, 0, 0);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    // fragment byte offsets inside a stage (row*128 + swizzled chunk), kk adds (4 ^ ...) -> precompute both kk
    int aoff[8][2], boff[4][2];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int row = wm * 128 + mi * 16 + frow;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) aoff[mi][kk] = row * 128 + (((kk * 4 + fq) ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int row = wn * 64 + ni * 16 + frow;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) boff[ni][kk] = A_BYTES + row * 128 + (((kk * 4 + fq) ^ ((row >> 1) & 7)) << 4);
    }

    bf16x8 af[4][2], b0[2][2], b1[2][2];
    auto read_a = [&](const char* st, int mh) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) af[i][kk] = *(const bf16x8*)(st + aoff[mh * 4 + i][kk]);
    };
    auto read_b = [&](const char* st, int nh, bf16x8 (&b)[2][2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) b[i][kk] = *(const bf16x8*)(st + boff[nh * 2 + i][kk]);
    };
    auto mma = [&](int mh, int nh, bf16x8 (&b)[2][2]) {
        if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[mh * 4 + i][nh * 2 + j] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][kk], b[j][kk], acc[mh * 4 + i][nh * 2 + j], 0, 0, 0);
        if (SETPRIO) __builtin_amdgcn_s_setprio(0);
    };

    const int nk = g.K / BK;
    dma_a(0, 0);
    dma_w(0, 0);
    PP_VM0();
    PP_BARRIER();
    if (grp == 1) PP_BARRIER();  // group B starts one phase late

    for (int kt = 0; kt < nk; ++kt) {
        const char* st = smem + (kt & 1) * STAGE_BYTES;
        const bool more = kt + 1 < nk;
        // R0
        if (more) dma_a((kt + 1) & 1, kt + 1);
        read_a(st, 0);
        read_b(st, 0, b0);
        PP_LGKM0();
        PP_BARRIER();
        mma(0, 0, b0);  // C0
        PP_BARRIER();
        // R1
        if (more) dma_w((kt + 1) & 1, kt + 1);
        read_b(st, 1, b1);
        PP_LGKM0();
        PP_BARRIER();
        mma(0, 1, b1);  // C1
        PP_BARRIER();
        // R2
        read_a(st, 1);
        PP_LGKM0();
        PP_BARRIER();
        mma(1, 1, b1);  // C2
        PP_BARRIER();
        // R3 (nothing left to read: B half 0 is still resident); next K-tile must have landed one phase before
        // group A starts reading it
        PP_VM0();
        PP_BARRIER();
        mma(1, 0, b0);  // C3
        PP_VM0();
        PP_BARRIER();
    }
    if (grp == 0) PP_BARRIER();  // group A waits out group B's last phase

    const int mrow0 = m0 + wm * 128 + fq * 4, ncol0 = n0 + wn * 64 + frow;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mrow0 + mi * 16 + r;
            if (m >= g.M) continue;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = ncol0 + ni * 16;
                if (n < g.N) g.C[(size_t)m * g.ldc + n] = f2bf(acc[mi][ni][r]);
            }
        }
}

// Ping-pong with 4 phases per K-tile: R(kk) reads all 12 fragments of one 32-deep half (8 A + 4 B), C(kk) issues the
// 32 MFMAs of that half over the whole 128x64 wave tile; half as many barriers as gemm_pp_kernel.
template <int NPH>
__global__ __launch_bounds__(512, 2) void gemm_pp2_kernel(GemmArgs g) {
    constexpr int BM = 256, BN = 256, BK = 64, A_BYTES = BM * 128, STAGE_BYTES = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wm = grp, wn = wave & 3;
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    constexpr int GN = 4;
    const int gsize = GN * ntm;
    const int grp_t = id / gsize, rem = id - grp_t * gsize;
    const int gn = min(GN, ntn - grp_t * GN);
    const int mt = rem / gn, nt = grp_t * GN + (rem - (rem / gn) * gn);
    const int m0 = mt * BM, n0 = nt * BN;
    const bf16_t* asrc[4];
    const bf16_t* wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        asrc[i] = g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + c * 8;
        wsrc[i] = g.W + (size_t)min(n0 + row, g.N - 1) * g.ldw + c * 8;
    }
    auto dma = [&](int buf, int kt) {
        char* base = smem + buf * STAGE_BYTES + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + kt * BK), (lptr_t)(base + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + kt * BK), (lptr_t)(base + A_BYTES + i * 1024), 16, 0, 0);
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fq = lane >> 4;
    int aoff[8][2], boff[4][2];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int row = wm * 128 + mi * 16 + frow;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) aoff[mi][kk] = row * 128 + (((kk * 4 + fq) ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int row = wn * 64 + ni * 16 + frow;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) boff[ni][kk] = A_BYTES + row * 128 + (((kk * 4 + fq) ^ ((row >> 1) & 7)) << 4);
    }
    const int nk = g.K / BK;
    dma(0, 0);
    PP_VM0();
    PP_BARRIER();
    if (NPH == 2 && nk > 1) dma(1, 1);
    if (grp == 1) PP_BARRIER();
    if constexpr (NPH == 4) {
        bf16x8 af[8], bfr[4];
        auto rd = [&](const char* st, int kk) {
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = *(const bf16x8*)(st + aoff[i][kk]);
#pragma unroll
            for (int i = 0; i < 4; ++i) bfr[i] = *(const bf16x8*)(st + boff[i][kk]);
        };
        auto mma = [&]() {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        };
        for (int kt = 0; kt < nk; ++kt) {
            const char* st = smem + (kt & 1) * STAGE_BYTES;
            if (kt + 1 < nk) dma((kt + 1) & 1, kt + 1);  // R0: next K-tile + fragments of half 0
            rd(st, 0);
            PP_LGKM0();
            PP_BARRIER();
            mma();  // C0
            PP_BARRIER();
            rd(st, 1);  // R1
            PP_LGKM0();
            PP_VM0();   // group B: next K-tile must be complete one phase before group A reads it
            PP_BARRIER();
            mma();  // C1
            PP_VM0();
            PP_BARRIER();
        }
    } else {  // NPH == 2: the whole K-tile's 24 fragments are read in one phase, 64 MFMAs in the other.
        // All 8 waves issue the LDS-DMA of K-tile t+1 in the same global phase (group A at the top of its R(t),
        // group B at the top of its C(t-1)) and drain it one phase later (A: end of C(t), B: end of R(t)), i.e. one
        // phase before group A starts reading it.
        bf16x8 af[8][2], bfr[4][2];
        for (int kt = 0; kt < nk; ++kt) {
            const char* st = smem + (kt & 1) * STAGE_BYTES;
            if (grp == 0 && kt >= 1 && kt + 1 < nk) dma((kt + 1) & 1, kt + 1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int i = 0; i < 8; ++i) af[i][kk] = *(const bf16x8*)(st + aoff[i][kk]);
#pragma unroll
                for (int i = 0; i < 4; ++i) bfr[i][kk] = *(const bf16x8*)(st + boff[i][kk]);
            }
            PP_LGKM0();
            if (grp == 1) PP_VM0();
            PP_BARRIER();
            if (grp == 1 && kt + 2 < nk) dma(kt & 1, kt + 2);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][kk], bfr[j][kk], acc[i][j], 0, 0, 0);
            if (grp == 0) PP_VM0();
            PP_BARRIER();
        }
    }
    if (grp == 0) PP_BARRIER();
    const int mrow0 = m0 + wm * 128 + fq * 4, ncol0 = n0 + wn * 64 + frow;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mrow0 + mi * 16 + r;
            if (m >= g.M) continue;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = ncol0 + ni * 16;
                if (n < g.N) g.C[(size_t)m * g.ldc + n] = f2bf(acc[mi][ni][r]);
            }
        }
}

// 32x32x16 MFMA, 8 waves (2 x 4) of 128x64, register double-buffered fragments: the 6 ds_read_b128 of k-step s+1
// are issued before the 8 MFMAs of k-step s (software pipelining inside the wave), one barrier per K-tile.
template <int WAVES_M, int WAVES_N, int MINW>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, MINW) void gemm_swp32_kernel(GemmArgs g) {
    constexpr int BM = 256, BN = 256, BK = 64, NW = WAVES_M * WAVES_N, A_BYTES = BM * 128, STAGE_BYTES = (BM + BN) * 128;
    constexpr int TM = BM / WAVES_M, TN = BN / WAVES_N, FM = TM / 32, FN = TN / 32;
    constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    constexpr int GN = 4;
    const int gsize = GN * ntm;
    const int grp_t = id / gsize, rem = id - grp_t * gsize;
    const int gn = min(GN, ntn - grp_t * GN);
    const int mt = rem / gn, nt = grp_t * GN + (rem - (rem / gn) * gn);
    const int m0 = mt * BM, n0 = nt * BN;
    const bf16_t* asrc[PA];
    const bf16_t* wsrc[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = (wave * PA + i) * 8 + (lane >> 3);
        asrc[i] = g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int row = (wave * PB + i) * 8 + (lane >> 3);
        wsrc[i] = g.W + (size_t)min(n0 + row, g.N - 1) * g.ldw + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
    }
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
    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int frow = lane & 31, hi = lane >> 5;
    int aoff[FM], boff[FN];  // row*128 of this lane's fragment rows; chunk term added per k-step
    int asw[FM], bsw[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) { const int row = wm * TM + i * 32 + frow; aoff[i] = row * 128; asw[i] = (row >> 1) & 7; }
#pragma unroll
    for (int i = 0; i < FN; ++i) { const int row = wn * TN + i * 32 + frow; boff[i] = A_BYTES + row * 128; bsw[i] = (row >> 1) & 7; }
    bf16x8 fa[2][FM], fb[2][FN];
    auto rd = [&](const char* st, int ks, bf16x8 (&a)[FM], bf16x8 (&b)[FN]) {
#pragma unroll
        for (int i = 0; i < FM; ++i) a[i] = *(const bf16x8*)(st + aoff[i] + (((ks * 2 + hi) ^ asw[i]) << 4));
#pragma unroll
        for (int i = 0; i < FN; ++i) b[i] = *(const bf16x8*)(st + boff[i] + (((ks * 2 + hi) ^ bsw[i]) << 4));
    };
    auto mma = [&](bf16x8 (&a)[FM], bf16x8 (&b)[FN]) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    const int nk = g.K / BK;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
        const char* st = smem + (kt & 1) * STAGE_BYTES;
        rd(st, 0, fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        rd(st, 1, fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        rd(st, 2, fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        rd(st, 3, fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa[1], fb[1]);
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

template <int WAVES_M, int WAVES_N, int MINW>
int launch_swp32(const GemmArgs& g, hipStream_t s) {
    constexpr int LDS = 2 * 512 * 128;
    static bool attr_set = false;
    auto fn = gemm_swp32_kernel<WAVES_M, WAVES_N, MINW>;
    if (!attr_set) {
        MM_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int ntm = (g.M + 255) / 256, ntn = (g.N + 255) / 256;
    hipLaunchKernelGGL(fn, dim3(ntm * ntn), dim3(64 * WAVES_M * WAVES_N), LDS, s, g);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int NPH>
int launch_pp2(const GemmArgs& g, hipStream_t s) {
    constexpr int LDS = 2 * 512 * 128;
    static bool attr_set = false;
    auto fn = gemm_pp2_kernel<NPH>;
    if (!attr_set) {
        MM_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int ntm = (g.M + 255) / 256, ntn = (g.N + 255) / 256;
    hipLaunchKernelGGL(fn, dim3(ntm * ntn), dim3(512), LDS, s, g);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}

template <bool SETPRIO, int GRPMODE>
int launch_pp(const GemmArgs& g, hipStream_t s) {
    constexpr int LDS = 2 * 512 * 128;
    static bool attr_set = false;
    auto fn = gemm_pp_kernel<SETPRIO, GRPMODE>;
    if (!attr_set) {
        MM_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int ntm = (g.M + 255) / 256, ntn = (g.N + 255) / 256;
    hipLaunchKernelGGL(fn, dim3(ntm * ntn), dim3(512), LDS, s, g);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int RING, bool PREFETCH>
int launch_big(const GemmArgs& g, hipStream_t s) {
    constexpr int LDS = RING * 512 * 64;
    static bool attr_set = false;
    auto fn = gemm_big_kernel<RING, PREFETCH>;
    if (!attr_set) {
        MM_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int ntm = (g.M + 255) / 256, ntn = (g.N + 255) / 256;
    hipLaunchKernelGGL(fn, dim3(ntm * ntn), dim3(256), LDS, s, g);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int BM, int BN, int BK, int WM, int WN, int STAGES, int MINW, bool NOSTORE = false, bool NODMA = false, bool NOLDS = false, int SPREAD = 0>
int launch_var(const GemmArgs& g, hipStream_t s) {
    constexpr int LDS = STAGES * (BM + BN) * BK * 2;
    static bool attr_set = false;
    auto fn = gemm_var_kernel<BM, BN, BK, WM, WN, STAGES, MINW, NOSTORE, NODMA, NOLDS, SPREAD>;
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

int launch_gemm8_variant(int variant, const GemmArgs& g, hipStream_t s);  // gemm8.hip

int launch_gemm_variant(int variant, const GemmArgs& g, hipStream_t s) {
    if (variant >= 200 && variant < 300) return launch_gemm8_variant(variant, g, s);
    if (g.K % 64) return mm_fail("gemm_variant: K must be a multiple of 64");
    switch (variant) {
        case 100: return launch_gemm(EPI_STORE, g, s);  // the production kernel (BM picker included)
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
        case 40: return launch_var<256, 256, 64, 4, 4, 2, 4, true>(g, s);  // v10 without the epilogue stores
        case 41: return launch_var<256, 256, 64, 4, 4, 2, 4, false, true>(g, s);  // v10 without the global->LDS stream
        case 42: return launch_var<256, 256, 64, 4, 4, 2, 4, false, true, true>(g, s);  // ... and without LDS reads
        // 32x32x16 MFMA:    BM   BN  WM WN ST MINW
        case 12: return launch_var32<128, 128, 2, 2, 2, 2>(g, s);
        case 13: return launch_var32<256, 256, 4, 4, 2, 4>(g, s);
        case 43: return launch_var32<256, 256, 4, 4, 2, 4, true>(g, s);  // 32x32x16 MFMA-only probe
        case 44: return launch_var32<256, 256, 2, 4, 2, 2, true>(g, s);  // same, 8 waves of 128x64
        case 14: return launch_var32<256, 256, 2, 4, 2, 2>(g, s);
        case 15: return launch_var32<256, 128, 4, 2, 2, 2>(g, s);
        case 16: return launch_var32<256, 128, 4, 4, 3, 4>(g, s);
        case 17: return launch_var<256, 128, 64, 4, 4, 3, 4>(g, s);
        case 18: return launch_var<256, 256, 64, 4, 2, 2, 2>(g, s);
        case 30: return launch_pp<false, 0>(g, s);
        case 31: return launch_pp<true, 0>(g, s);
        case 50: return launch_swp32<2, 4, 2>(g, s);
        case 51: return launch_swp32<4, 4, 4>(g, s);
        case 34: return launch_pp2<4>(g, s);
        case 35: return launch_pp2<2>(g, s);
        case 32: return launch_pp<false, 1>(g, s);
        case 33: return launch_pp<false, 2>(g, s);
        case 70: return launch_var<256, 256, 64, 4, 4, 2, 4, false, false, false, 1>(g, s);  // v10, DMA pieces spread over the first half's MFMAs
        case 71: return launch_var<256, 256, 64, 4, 4, 2, 4, false, false, false, 2>(g, s);  // ... over the second half's
        case 72: return launch_var<256, 256, 64, 4, 4, 2, 4, false, false, false, 3>(g, s);  // v10 + L2 touch of the K-tile after next
        case 80: return launch_half<256, 256, 4, 4>(g, s);  // five 32-deep half-tiles in flight, one barrier per 64
        case 60: return launch_pers<256, 256, 64, 4, 4, 2, 4, false>(g, s);  // v10, persistent
        case 61: return launch_pers<256, 256, 64, 4, 4, 2, 4, true>(g, s);   // ... + next-tile prefetch under the epilogue
        case 20: return launch_big<4, false>(g, s);
        case 21: return launch_big<4, true>(g, s);
        case 22: return launch_big<5, true>(g, s);
    }
    return mm_fail("gemm_variant: unknown variant %d", variant);
}
