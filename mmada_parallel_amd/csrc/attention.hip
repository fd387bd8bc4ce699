// attention.hip — unmasked, non-causal flash attention forward for gfx950 (head_dim 128, bf16 in/out).
//
// Replaces F.scaled_dot_product_attention(q, k, v, attn_mask=None, is_causal=False) as called by the reference
// (model/modeling_llada.py:672-679 via :731-738; the attention-bias machinery around it is dead code).
//
// Layout contract (produced by the QKV GEMM epilogue): q [B,Hq,Lkv,128], k [B,Hkv,Lkv,128] row-major and V stored
// K-major as vT [B,Hkv,128,Lkv], so both MFMA operands of both products are contiguous along the contraction.
//
// One workgroup = 4 waves = 128 query rows; one wave owns 32 query rows and the whole 512-deep softmax state.
//   S^T = K·Q^T   : v_mfma_f32_32x32x16_bf16, A = K tile rows (LDS), B = Q rows (registers) -> each lane holds
//                   16 scores of ONE query (its column), so row max/sum are in-lane + one lane^32 exchange.
//   O^T = V^T·P^T : A = vT tile rows (LDS, 2 x ds_read_b64), B = P straight from the S accumulator registers —
//                   the k-slot order of the two operands is chosen to match the accumulator layout
//                   (key = (r&3) + 8*(r>>2) + 4*(lane>>5)), so P never moves between lanes.
// K/V tiles (64 keys) are double-buffered in LDS through registers (issue-early / write-late), K rows XOR-swizzled
// per 16-B chunk and vT rows per 8-B chunk so ds_read_b128 / ds_read_b64 are conflict-free.
#include "kernels.h"

namespace {

constexpr int QB = 128;  // query rows per workgroup
constexpr int KB = 64;   // keys per tile
constexpr int TILE_BYTES = KB * 128 * 2;  // 16 KiB (K tile == vT tile)
constexpr int ATT_LDS = 4 * TILE_BYTES;   // 2 stages x (K + vT)

struct AttnArgs {
    const bf16_t* q;
    const bf16_t* k;
    const bf16_t* vT;
    bf16_t* out;
    int Hq, Hkv, L, Lq_rows, Lkv, out_rows_per_batch, ld_out;
    float scale_log2e;
};

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 31, hi = lane >> 5;
    const int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int hkv = h / (a.Hq / a.Hkv);
    const bf16_t* Qp = a.q + (size_t)(b * a.Hq + h) * a.Lkv * 128;
    const bf16_t* Kp = a.k + (size_t)(b * a.Hkv + hkv) * a.Lkv * 128;
    const bf16_t* Vp = a.vT + (size_t)(b * a.Hkv + hkv) * 128 * a.Lkv;

    const int q_row = qb * QB + wave * 32 + ql;
    const int q_ld = min(q_row, a.Lkv - 1);
    bf16x8 qf[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) qf[s] = *(const bf16x8*)(Qp + (size_t)q_ld * 128 + s * 16 + hi * 8);

    f32x16 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    u32x4 kreg[4], vreg[4];
    // per-thread 32-bit element offsets inside a tile; the tile base stays wave-uniform (SGPR base + VGPR offset)
    const int koff = (tid >> 4) * 128 + (tid & 15) * 8;
    const int voff = (tid >> 3) * a.Lkv + (tid & 7) * 8;
    auto load_regs = [&](int kt) {
        const bf16_t* kb = Kp + (size_t)kt * KB * 128;
        const bf16_t* vb = Vp + kt * KB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            kreg[i] = *(const u32x4*)(kb + koff + i * 16 * 128);
            vreg[i] = *(const u32x4*)(vb + voff + i * 32 * a.Lkv);
        }
    };
    auto write_lds = [&](int buf) {
        char* Kt = smem + buf * 2 * TILE_BYTES;
        char* Vt = Kt + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = i * 256 + tid;
            const int kr = id >> 4, kc = id & 15;
            *(u32x4*)(Kt + kr * 256 + ((kc ^ (kr & 15)) << 4)) = kreg[i];
            const int d = id >> 3, c8 = (id & 7) * 2, sw = (d >> 1) & 15;
            *(u32x2*)(Vt + d * 128 + ((c8 ^ sw) << 3)) = u32x2{vreg[i][0], vreg[i][1]};
            *(u32x2*)(Vt + d * 128 + (((c8 + 1) ^ sw) << 3)) = u32x2{vreg[i][2], vreg[i][3]};
        }
    };

    const int nkt = (a.L + KB - 1) / KB;
    load_regs(0);
    write_lds(0);
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const char* Kt = smem + cur * 2 * TILE_BYTES;
        const char* Vt = Kt + TILE_BYTES;

        // ---- S^T = K · Q^T for keys [0,32) and [32,64) of the tile ----
        f32x16 s0, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int ch = ((2 * s + hi) ^ (ql & 15)) << 4;
            const bf16x8 ka0 = *(const bf16x8*)(Kt + ql * 256 + ch);
            const bf16x8 ka1 = *(const bf16x8*)(Kt + (32 + ql) * 256 + ch);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka0, qf[s], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka1, qf[s], s1, 0, 0, 0);
        }
        // keys past L (only in the last tile) get -inf; select, not arithmetic, so garbage K rows cannot leak NaN
        if (kt * KB + KB > a.L) {
            const int kbase = kt * KB + 4 * hi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + (r & 3) + 8 * (r >> 2);
                if (key >= a.L) s0[r] = -INFINITY;
                if (key + 32 >= a.L) s1[r] = -INFINITY;
            }
        }
        // ---- online softmax (fp32); lane and lane^32 share a query ----
        float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * a.scale_log2e);
        const float mc = m_new * a.scale_log2e;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(s0[r] * a.scale_log2e - mc);
            s1[r] = __builtin_amdgcn_exp2f(s1[r] * a.scale_log2e - mc);
            psum += s0[r] + s1[r];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;

        // P -> bf16 B-operand fragments: pb[t][s2] = P[q][keys of accumulator regs 8*s2 .. 8*s2+7 of tile t]
        bf16x8 pb[2][2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                pb[0][s2][j] = (__bf16)s0[8 * s2 + j];
                pb[1][s2][j] = (__bf16)s1[8 * s2 + j];
            }

        // next tile: global -> registers now (in flight under the PV MFMAs), registers -> LDS after them.
        // Issued here rather than at the top of the iteration so the 32 staging VGPRs are never live together
        // with the 32 score registers (keeps the kernel at 2 waves/SIMD without spills).
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nkt) load_regs(kt + 1);

        // ---- O^T += V^T · P^T ----
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            // row d = db*32 + ql; its swizzle ((d>>1)&15) does not depend on db, so the 8 chunk addresses are
            // shared by the four d-blocks up to an immediate offset
            const char* vrow = Vt + ql * 128 + db * 32 * 128;
            const int sw = (ql >> 1) & 15;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int ca = 8 * t + 4 * s2 + hi;
                    const bf16x4 lo = *(const bf16x4*)(vrow + ((ca ^ sw) << 3));
                    const bf16x4 hi4 = *(const bf16x4*)(vrow + (((ca + 2) ^ sw) << 3));
                    const bf16x8 va = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pb[t][s2], o[db], 0, 0, 0);
                }
        }

        if (kt + 1 < nkt) write_lds(cur ^ 1);
        __syncthreads();
    }

    // ---- normalise and store: lane holds O[q_row][d = db*32 + 8g + 4hi + j] ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_row < a.Lq_rows) {
        bf16_t* orow = a.out + ((size_t)b * a.out_rows_per_batch + q_row) * a.ld_out + h * 128;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                u32x2 pk;
                pk[0] = pack_bf2(o[db][4 * g4 + 0] * inv, o[db][4 * g4 + 1] * inv);
                pk[1] = pack_bf2(o[db][4 * g4 + 2] * inv, o[db][4 * g4 + 3] * inv);
                *(u32x2*)(orow + db * 32 + 8 * g4 + 4 * hi) = pk;
            }
    }
}

}  // namespace

int launch_attention(const bf16_t* q, const bf16_t* k, const bf16_t* vT, bf16_t* out, int B, int Hq, int Hkv, int L,
                     int Lq_rows, int Lkv, int out_rows_per_batch, int ld_out, hipStream_t s) {
    if (L <= 0 || B <= 0) return 0;
    if (Lkv % 64 || Lkv < L) return mm_fail("attention: Lkv=%d must be a multiple of 64 and >= L=%d", Lkv, L);
    if (Hq % Hkv) return mm_fail("attention: n_heads %% n_kv_heads != 0");
    static bool attr_set = false;
    if (!attr_set) {
        MM_CHECK_HIP(hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS));
        attr_set = true;
    }
    AttnArgs a;
    a.q = q; a.k = k; a.vT = vT; a.out = out;
    a.Hq = Hq; a.Hkv = Hkv; a.L = L; a.Lq_rows = Lq_rows; a.Lkv = Lkv;
    a.out_rows_per_batch = out_rows_per_batch; a.ld_out = ld_out;
    a.scale_log2e = 0.08838834764831845f * 1.4426950408889634f;  // 1/sqrt(128) * log2(e)
    hipLaunchKernelGGL(attn_fwd_kernel, dim3((Lq_rows + QB - 1) / QB, Hq, B), dim3(256), ATT_LDS, s, a);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}
