// gemm.hip — bf16 MFMA GEMM  C[M,N] = A[M,K] · W[N,K]^T  for gfx950 with fused epilogues.
//
// This is the contraction behind q/k/v_proj, attn_out, ff_proj/up_proj, ff_out and the LM head of the reference
// (model/modeling_llada.py:925-927, 741-744, 962-970, 1399-1404: all nn.Linear without bias, bf16 storage).
//
// Structure (v1): 128x128x64 block tile, 256 threads = 4 waves (2x2), each wave a 64x64 sub-tile as 4x4
// v_mfma_f32_16x16x32_bf16 fragments (fp32 accumulate).  Both operands are K-contiguous, so a fragment is one
// 16-byte LDS read.  Global->LDS staging uses the gfx950 LDS-DMA (global_load_lds_dwordx4): the LDS image is
// lane-linear, so the bank-conflict swizzle (16-byte chunk c -> c ^ ((row>>1)&7) inside each 128-byte row) is
// applied to the per-lane SOURCE address and to the ds_read address (guide rule 21).  Two LDS stages; the load of
// tile t+1 is in flight while tile t is multiplied.  Workgroup ids are remapped XCD-aware and in grouped order so
// the 8 private L2s see contiguous patches of the tile grid.
//
// Epilogues reproduce the reference's rounding points exactly: every nn.Linear output is rounded to bf16 before
// anything else touches it.
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 32 KiB
constexpr int LDS_BYTES = 2 * STAGE_BYTES;       // 64 KiB

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

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bt_128(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    // XCD-aware + grouped tile order: GN column tiles x all row tiles form one group
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    constexpr int GN = 8;
    const int gsize = GN * ntm;
    const int grp = id / gsize, rem = id - grp * gsize;
    const int gn = min(GN, ntn - grp * GN);
    const int mt = rem / gn, nt = grp * GN + (rem - (rem / gn) * gn);
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- staging addresses: wave w issues LDS-DMA pieces i = 4w..4w+3 of A and of W (8 rows x 128 B each) ----
    const bf16_t* asrc[4];
    const bf16_t* wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);  // logical 16-B chunk this lane fetches
        const int gm = min(m0 + row, g.M - 1);
        const int gw = min(n0 + row, g.N - 1);
        asrc[i] = g.A + (size_t)gm * g.lda + c * 8;
        wsrc[i] = g.W + (size_t)gw * g.ldw + c * 8;
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE_BYTES + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(asrc[i] + kt * BK, base + i * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(wsrc[i] + kt * BK, base + BM * BK * 2 + i * 1024);
    };

    // fragment read addresses (bytes inside an operand tile): row-major 128-B rows, swizzled chunk
    const int frow = lane & 15;  // row inside a 16-row fragment
    const int fq = lane >> 4;    // which 8-element k group of the 32-wide MFMA step

    const int nk = g.K / BK;
    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* At = smem + cur * STAGE_BYTES;
        const char* Wt = At + BM * BK * 2;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[4], b[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int row = wm * 64 + mi * 16 + frow;
                const int ch = (kk * 4 + fq) ^ ((row >> 1) & 7);
                a[mi] = *(const bf16x8*)(At + row * 128 + ch * 16);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int row = wn * 64 + ni * 16 + frow;
                const int ch = (kk * 4 + fq) ^ ((row >> 1) & 7);
                b[ni] = *(const bf16x8*)(Wt + row * 128 + ch * 16);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        __syncthreads();  // tile kt+1 landed (vmcnt(0) inside) and every wave is done reading tile kt
    }

    // ---- epilogue: acc[mi][ni][r] = D[m][n], m = m0+wm*64+mi*16+fq*4+r, n = n0+wn*64+ni*16+frow ----
    const int mrow0 = m0 + wm * 64 + fq * 4;
    const int ncol0 = n0 + wn * 64 + frow;

    if constexpr (EPI == EPI_STORE || EPI == EPI_RESID) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mrow0 + mi * 16 + r;
                if (m >= g.M) continue;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int n = ncol0 + ni * 16;
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
        const int hcol0 = (n0 + wn * 64) / 2 + frow;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mrow0 + mi * 16 + r;
                if (m >= g.M) continue;
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const float gate = bfround(acc[mi][2 * q2][r]);
                    const float up = bfround(acc[mi][2 * q2 + 1][r]);
                    g.C[(size_t)m * g.ldc + hcol0 + q2 * 16] = f2bf(silu_bf16(gate) * up);
                }
            }
    } else {  // EPI_QKV
        const int head = nt;  // BN == head_dim == 128: one head per column tile
        if (head < g.Hq + g.Hkv) {
            const bool isq = head < g.Hq;
            bf16_t* dst = isq ? g.q : g.k;
            const int hh = isq ? head : head - g.Hq;
            const int nh = isq ? g.Hq : g.Hkv;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mrow0 + mi * 16 + r;
                    if (m >= g.M) continue;
                    const int b = m / g.Lp, l = m - b * g.Lp;
                    bf16_t* row = dst + ((size_t)(b * nh + hh) * g.Lkv + l) * 128;
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2) {
                        // permuted column layout: fragments (2*q2, 2*q2+1) hold rotary partners i and i+64
                        const int i = (wn * 2 + q2) * 16 + frow;
                        const float t1 = bfround(acc[mi][2 * q2][r]);
                        const float t2 = bfround(acc[mi][2 * q2 + 1][r]);
                        const float c = g.rope_cos[l * 64 + i], s = g.rope_sin[l * 64 + i];
                        float o1, o2;
                        rope_pair(t1, t2, c, s, o1, o2);
                        row[i] = f2bf(o1);
                        row[i + 64] = f2bf(o2);
                    }
                }
        } else {
            const int hv = head - g.Hq - g.Hkv;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int mb = mrow0 + mi * 16;  // multiple of 4; Lp is a multiple of 8 -> 4 rows share a batch
                if (mb >= g.M) continue;
                const int b = mb / g.Lp, l0 = mb - b * g.Lp;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int d = wn * 64 + ni * 16 + frow;
                    u32x2 pk;
                    pk[0] = pack_bf2(acc[mi][ni][0], acc[mi][ni][1]);
                    pk[1] = pack_bf2(acc[mi][ni][2], acc[mi][ni][3]);
                    *(u32x2*)(g.vT + ((size_t)(b * g.Hkv + hv) * 128 + d) * g.Lkv + l0) = pk;
                }
            }
        }
    }
}

template <int EPI>
int launch_t(const GemmArgs& g, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        MM_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_128<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         LDS_BYTES));
        attr_set = true;
    }
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    hipLaunchKernelGGL(gemm_bt_128<EPI>, dim3(ntm * ntn), dim3(256), LDS_BYTES, s, g);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
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
            if (g.N % 128) return mm_fail("gemm/swiglu: N must be a multiple of 128");
            return launch_t<EPI_SWIGLU>(g, s);
        case EPI_QKV:
            if (g.N != (g.Hq + 2 * g.Hkv) * 128) return mm_fail("gemm/qkv: N mismatch");
            if (g.Lp % 8) return mm_fail("gemm/qkv: Lp must be a multiple of 8");
            return launch_t<EPI_QKV>(g, s);
    }
    return mm_fail("gemm: bad epilogue %d", epi);
}
