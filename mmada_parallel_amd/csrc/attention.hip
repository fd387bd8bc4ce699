// attention.hip — unmasked, non-causal flash attention forward for gfx950 (head_dim 128, bf16 in/out).
//
// Replaces F.scaled_dot_product_attention(q, k, v, attn_mask=None, is_causal=False) as called by the reference
// (model/modeling_llada.py:672-679 via :731-738; the attention-bias machinery around it is dead code).
//
// Layout contract (produced by the QKV GEMM epilogue): q [B,Hq,Lkv,128], k [B,Hkv,Lkv,128] row-major and V stored
// K-major as vT [B,Hkv,128,Lkv] with the keys of every 16-key group stored in the order [0-3, 8-11, 4-7, 12-15]
// (vt_key_pos in common.h), so both MFMA operands of both products are one contiguous 16-byte read.
//
// One workgroup = 4 waves = 128 query rows; one wave owns 32 query rows and the whole softmax state.
//   S^T = K·Q^T   : v_mfma_f32_32x32x16_bf16, A = K tile rows (LDS), B = Q rows (registers) -> each lane holds
//                   16 scores of ONE query (its column), so row max/sum are in-lane + one lane^32 exchange.
//   O^T = V^T·P^T : A = vT tile rows (LDS), B = P straight from the S accumulator registers — the key order of the
//                   stored vT rows is exactly the accumulator's key order (key = (r&3) + 8*(r>>2) + 4*(lane>>5)),
//                   so P never moves between lanes and V needs no transpose read.
// K/V tiles (64 keys) arrive by LDS-DMA (global_load_lds_dwordx4) into a 2-stage ring, one barrier per tile; the
// 16-B-chunk XOR swizzle is applied to the DMA source address and to the ds_read_b128 address (conflict-free).
// The O rescale is skipped while the running max grows by less than 2^DEFER_LOG2 (P stays <= 2^DEFER_LOG2).
#include "attention.h"

namespace {

using namespace attn_detail;

__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hi = lane >> 5;
    // Workgroup -> (query tile, head, batch).  With xcd_pairs > 0 the grid is 1-D and XCD-aware: hardware workgroup ids
    // round-robin over the 8 XCDs (observed, speed only), so XCD x takes the (batch, head) pairs [x*xcd_pairs, (x+1)*
    // xcd_pairs) and all their query tiles — a head's K / vT (1.25 MB at L = 2438) is then fetched into ONE private L2
    // instead of all eight.
    int qb, h, b;
    if (a.xcd_pairs > 0) {
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        const int pair = xcd * a.xcd_pairs + i / a.nq;
        qb = i - (i / a.nq) * a.nq;
        b = pair / a.Hq;
        h = pair - b * a.Hq;
    } else {
        qb = blockIdx.x; h = blockIdx.y; b = blockIdx.z;
    }
    const int hkv = h / (a.Hq / a.Hkv);
    const bf16_t* Qp = a.q + (size_t)(b * a.Hq + h) * a.Lq_alloc * 128;
    const bf16_t* Kp = a.k + (size_t)(b * a.Hkv + hkv) * a.Lkv * 128;
    const bf16_t* Vp = a.vT + (size_t)(b * a.Hkv + hkv) * 128 * a.Lkv;

    const int q_row = a.q_begin + qb * QB + wave * 32 + ql;
    const int q_ld = min(q_row, a.Lq_alloc - 1);
    bf16x8 qf[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) qf[s] = *(const bf16x8*)(Qp + (size_t)q_ld * 128 + s * 16 + hi * 8);

    f32x16 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;  // m_run in scaled log2 units

    // LDS-DMA sources: wave w moves K pieces 4w..4w+3 (4 rows x 256 B each) and vT pieces 4w..4w+3 (8 rows x 128 B)
    // byte offsets (unsigned 32-bit) from a wave-uniform tile base: the loads take the scalar-base + vector-offset form,
    // so advancing to the next tile is two scalar adds instead of eight 64-bit vector adds
    unsigned koff[4], voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kr = (wave * 4 + i) * 4 + (lane >> 4);
        koff[i] = (unsigned)(kr * 128 + (((lane & 15) ^ (kr & 15)) << 3)) * 2u;
        const int d = (wave * 4 + i) * 8 + (lane >> 3);
        voff[i] = (unsigned)(d * a.Lkv + (((lane & 7) ^ ((d >> 1) & 7)) << 3)) * 2u;
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * 2 * TILE_BYTES + wave * 4096;
        const char* kb = (const char*)Kp + (size_t)kt * KB * 256;
        const char* vb = (const char*)Vp + (size_t)kt * KB * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(kb + koff[i]), (lptr_t)(base + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(vb + voff[i]), (lptr_t)(base + TILE_BYTES + i * 1024), 16, 0, 0);
    };

    const int nkt = (a.L + KB - 1) / KB;
    const int ksw = ql & 15, vsw = (ql >> 1) & 7;
    stage(0, 0);

    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 1 < nkt) stage((kt + 1) & 1, kt + 1);
        const char* Kt = smem + (kt & 1) * 2 * TILE_BYTES;
        const char* Vt = Kt + TILE_BYTES;

        // ---- S^T = K · Q^T for keys [0,32) and [32,64) of the tile ----
        f32x16 s0, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int ch = ((2 * s + hi) ^ ksw) << 4;
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
        // ---- online softmax (fp32, log2 domain); lane and lane^32 share a query ----
        // two chains (shorter dependency depth), each STARTED by a compiler-visible fmaxf: hipcc inserts the wait states an
        // MFMA result needs before a VALU may read it only for instructions it can see (guide §5.7); the asm v_max3 ops depend
        // on these two and therefore come later.  (A loop order that put asm maxima right behind the MFMAs read stale scores.)
        float mxa = fmaxf(s0[0], s1[0]), mxb = fmaxf(s0[1], s1[1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) {
            mxa = max3f(mxa, s0[r], s1[r]);
            mxb = max3f(mxb, s0[r + 1], s1[r + 1]);
        }
        float mx = fmax_nc(mxa, mxb);
        mx = fmax_nc(mx, __shfl_xor(mx, 32, 64)) * a.scale_log2e;
        if (!__all(mx - m_run <= DEFER_LOG2)) {  // wave-uniform: rescale only when some row's max really grew
            const float m_new = fmax_nc(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
            m_run = m_new;
        }
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(s0[r] * a.scale_log2e - m_run);
            s1[r] = __builtin_amdgcn_exp2f(s1[r] * a.scale_log2e - m_run);
            psum += s0[r] + s1[r];
        }
        l_run += psum;

        // P -> bf16 B-operand fragments: pb[t][s2] = P[q][keys of accumulator regs 8*s2 .. 8*s2+7 of tile t]
        bf16x8 pb[2][2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                pb[0][s2][j] = (__bf16)s0[8 * s2 + j];
                pb[1][s2][j] = (__bf16)s1[8 * s2 + j];
            }

        // ---- O^T += V^T · P^T ----  (the four accumulators interleaved: consecutive MFMAs never wait for each other; each
        // accumulator still takes its key fragments in the order (t, s2) = (0,0) (0,1) (1,0) (1,1): bit-identical sums)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    const char* vrow = Vt + (db * 32 + ql) * 128;  // swizzle of row db*32+ql does not depend on db
                    const bf16x8 va = *(const bf16x8*)(vrow + (((4 * t + 2 * s2 + hi) ^ vsw) << 4));
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pb[t][s2], o[db], 0, 0, 0);
                }
    }

    // ---- normalise and store: lane holds O[q_row][d = db*32 + 8g + 4hi + j] ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_row < a.Lq_rows) {
        bf16_t* orow = a.out + ((size_t)b * a.out_rows_per_batch + q_row - a.q_begin) * a.ld_out + h * 128;
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


// ---- 4-wave kernel, pipelined matrix blocks (round 3) ---------------------------------------------------------------
// The arithmetic and its order are those of attn_fwd_kernel (bit-identical output); what changes is how the two matrix
// blocks of a key tile are issued.  hipcc scheduled them as  { 2 x ds_read_b128 ; s_waitcnt lgkmcnt(0) ; 2 x MFMA } x 8:
// every step exposed an LDS round trip in front of 64 cycles of MFMA work.  Here the fragments run THREE k-steps (S) / four
// MFMAs (PV) ahead of their use in registers, the order is pinned with sched_group_barrier, LDS addresses are per-lane
// offsets + immediates (the two ring stages are compile-time constants: loop unrolled by two), the four O accumulators
// are interleaved, the eight LDS-DMA pieces of the next tile are issued one per MFMA pair inside the S block instead
// of in one burst in front of it, the first vT fragments are requested before the soft-max and the half-row maximum is
// exchanged with v_permlane32_swap instead of ds_bpermute (no LDS round trip between the maximum and the exponentials).
// VAR (tuning builds only, tools/attn_sweep.py; the product instantiates VAR 0): 1-3 are DIAGNOSTIC (wrong results, timing
// only) — 1: no soft-max arithmetic, 2: no MFMAs, 3: no tile barrier; 4: static s_setprio 1 for the workgroup whose LDS
// allocation does not start at 0 (the second workgroup of the CU).
template <int VAR>
__global__ __launch_bounds__(256, 2) void attn4p_fwd_kernel(AttnArgs a) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hi = lane >> 5;
    if constexpr (VAR == 4) {
        // HW_REG_LDS_ALLOC (id 6): LDS_BASE in bits 7:0
        if (__builtin_amdgcn_s_getreg((6) | (0 << 6) | ((8 - 1) << 11)) != 0) __builtin_amdgcn_s_setprio(1);
    }
    int qb, h, b;
    if (a.xcd_pairs > 0) {
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        const int pair = xcd * a.xcd_pairs + i / a.nq;
        qb = i - (i / a.nq) * a.nq;
        b = pair / a.Hq;
        h = pair - b * a.Hq;
    } else {
        qb = blockIdx.x; h = blockIdx.y; b = blockIdx.z;
    }
    const int hkv = h / (a.Hq / a.Hkv);
    const bf16_t* Qp = a.q + (size_t)(b * a.Hq + h) * a.Lq_alloc * 128;
    const bf16_t* Kp = a.k + (size_t)(b * a.Hkv + hkv) * a.Lkv * 128;
    const bf16_t* Vp = a.vT + (size_t)(b * a.Hkv + hkv) * 128 * a.Lkv;

    const int q_row = a.q_begin + qb * QB + wave * 32 + ql;
    const int q_ld = min(q_row, a.Lq_alloc - 1);
    bf16x8 qf[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) qf[s] = *(const bf16x8*)(Qp + (size_t)q_ld * 128 + s * 16 + hi * 8);
    f32x16 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    unsigned koff[4], voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kr = (wave * 4 + i) * 4 + (lane >> 4);
        koff[i] = (unsigned)(kr * 128 + (((lane & 15) ^ (kr & 15)) << 3)) * 2u;
        const int d = (wave * 4 + i) * 8 + (lane >> 3);
        voff[i] = (unsigned)(d * a.Lkv + (((lane & 7) ^ ((d >> 1) & 7)) << 3)) * 2u;
    }
    auto stage = [&](auto buf_, int kt) {  // stage BUF: K at BUF * 32 KiB, vT 16 KiB behind it (the layout of attn_fwd_kernel)
        constexpr int BUF = decltype(buf_)::value;
        const char* kb = (const char*)Kp + (size_t)kt * KB * 256;
        const char* vb = (const char*)Vp + (size_t)kt * KB * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dma16(kb, koff[i], BUF * 2 * TILE_BYTES + wave * 4096 + i * 1024);   // asm form: attention.h
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dma16(vb, voff[i], BUF * 2 * TILE_BYTES + TILE_BYTES + wave * 4096 + i * 1024);
        }
    };
    int kro[8], vro[4];
    {
        const int ksw = ql & 15, vsw = (ql >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 8; ++s) kro[s] = ql * 256 + (((2 * s + hi) ^ ksw) << 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) vro[j] = TILE_BYTES + ql * 128 + (((2 * j + hi) ^ vsw) << 4);
    }
    const int nkt = (a.L + KB - 1) / KB;
    stage(std::integral_constant<int, 0>{}, 0);

    auto tile = [&](auto r_, int kt) {
        constexpr int R = decltype(r_)::value, BASE = R * 2 * TILE_BYTES;
        using Next = std::integral_constant<int, R ^ 1>;
        asm volatile("" : "+v"(kro[0]), "+v"(kro[1]), "+v"(kro[2]), "+v"(kro[3]), "+v"(kro[4]), "+v"(kro[5]), "+v"(kro[6]), "+v"(kro[7]));
        asm volatile("" : "+v"(vro[0]), "+v"(vro[1]), "+v"(vro[2]), "+v"(vro[3]));
        if constexpr (VAR == 3) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        A8_SB();
        const bool more = kt + 1 < nkt;
        // ---- S^T = K · Q^T ----
        bf16x8 ka[3][2];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            ka[s][0] = lds_frag(kro[s] + BASE);
            ka[s][1] = lds_frag(kro[s] + BASE + 8192);
        }
        if (more) stage(Next{}, kt + 1);
        f32x16 s0, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if constexpr (VAR == 2) {
                asm volatile("" : "+v"(s0), "+v"(s1) : "v"(ka[s % 3][0]), "v"(ka[s % 3][1]));
            } else {
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[s % 3][0], qf[s], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[s % 3][1], qf[s], s1, 0, 0, 0);
            }
            if (s + 3 < 8) {
                ka[s % 3][0] = lds_frag(kro[s + 3] + BASE);
                ka[s % 3][1] = lds_frag(kro[s + 3] + BASE + 8192);
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (VAR != 2) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            if (s + 3 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // one LDS-DMA piece of the next tile (none in the last tile)
        }
        A8_SB();
        // the first vT fragments are requested BEFORE the soft-max: their LDS round trip runs under it
        bf16x8 va[4];
#pragma unroll
        for (int db = 0; db < 4; ++db) va[db] = lds_frag(vro[0] + BASE + db * 4096);
        A8_SB();
        if (kt * KB + KB > a.L) {
            const int kbase = kt * KB + 4 * hi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + (r & 3) + 8 * (r >> 2);
                if (key >= a.L) s0[r] = -INFINITY;
                if (key + 32 >= a.L) s1[r] = -INFINITY;
            }
        }
        if constexpr (VAR != 1) {
        float mxa = fmaxf(s0[0], s1[0]), mxb = fmaxf(s0[1], s1[1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) {
            mxa = max3f(mxa, s0[r], s1[r]);
            mxb = max3f(mxb, s0[r + 1], s1[r + 1]);
        }
        float mx = fmax_nc(mxa, mxb);
        {  // the other half-row's maximum by v_permlane32_swap (VALU) instead of ds_bpermute (an LDS round trip on the critical path)
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmax_nc(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * a.scale_log2e;
        }
        if (!__all(mx - m_run <= DEFER_LOG2)) {
            const float m_new = fmax_nc(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[k][r] *= alpha;
            m_run = m_new;
        }
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(s0[r] * a.scale_log2e - m_run);
            s1[r] = __builtin_amdgcn_exp2f(s1[r] * a.scale_log2e - m_run);
            psum += s0[r] + s1[r];
        }
        l_run += psum;
        }
        bf16x8 pb[2][2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                pb[0][s2][j] = (__bf16)s0[8 * s2 + j];
                pb[1][s2][j] = (__bf16)s1[8 * s2 + j];
            }
        A8_SB();
        // ---- O^T += V^T · P^T ----
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const int j = n >> 2, db = n & 3;
            if constexpr (VAR == 2) asm volatile("" : "+v"(o[db]) : "v"(va[db]), "v"(pb[j >> 1][j & 1]));
            else o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[db], pb[j >> 1][j & 1], o[db], 0, 0, 0);
            if (n + 4 < 16) va[db] = lds_frag(vro[j + 1] + BASE + db * 4096);
        }
#pragma unroll
        for (int n = 0; n < 12; ++n) {
            if (VAR != 2) __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
        }
        if (VAR != 2) __builtin_amdgcn_sched_group_barrier(0x008, 4, 1);
        A8_SB();
    };
    int kt = 0;
    for (; kt + 1 < nkt; kt += 2) {
        tile(std::integral_constant<int, 0>{}, kt);
        tile(std::integral_constant<int, 1>{}, kt + 1);
    }
    if (kt < nkt) tile(std::integral_constant<int, 0>{}, kt);

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_row < a.Lq_rows) {
        bf16_t* orow = a.out + ((size_t)b * a.out_rows_per_batch + q_row - a.q_begin) * a.ld_out + h * 128;
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

static int attn4p_set_lds_limit() {
    MM_CHECK_HIP(hipFuncSetAttribute((const void*)attn4p_fwd_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS));
#ifdef MMADA_TUNE
    MM_CHECK_HIP(hipFuncSetAttribute((const void*)attn4p_fwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS));
    MM_CHECK_HIP(hipFuncSetAttribute((const void*)attn4p_fwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS));
    MM_CHECK_HIP(hipFuncSetAttribute((const void*)attn4p_fwd_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS));
    MM_CHECK_HIP(hipFuncSetAttribute((const void*)attn4p_fwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS));
#endif
    return 0;
}

static int g_attn_form = -1;  // -1: read MMADA_ATTN_FORM once
void attention_force_form(int form) { g_attn_form = form; }  // -1: back to MMADA_ATTN_FORM / default  // measurement / test hook: 0 = round-2 issue order, 1 = pipelined matrix blocks

int launch_attention(const bf16_t* q, const bf16_t* k, const bf16_t* vT, bf16_t* out, int B, int Hq, int Hkv, int L,
                     int Lq_rows, int Lkv, int out_rows_per_batch, int ld_out, hipStream_t s, int q_begin, int Lq_alloc) {
    if (L <= 0 || B <= 0) return 0;
    if (Lkv % 64 || Lkv < L) return mm_fail("attention: Lkv=%d must be a multiple of 64 and >= L=%d", Lkv, L);
    if (Hq % Hkv) return mm_fail("attention: n_heads %% n_kv_heads != 0");
    if (q_begin < 0 || (q_begin & 31) || q_begin >= Lq_rows) return mm_fail("attention: bad q_begin=%d", q_begin);
    static MmOncePerDevice attr_set;
    MM_ONCE_PER_DEVICE(attr_set, MM_CHECK_HIP(hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS)));
    AttnArgs a;
    a.q = q; a.k = k; a.vT = vT; a.out = out;
    a.Hq = Hq; a.Hkv = Hkv; a.L = L; a.Lq_rows = Lq_rows; a.Lkv = Lkv;
    a.out_rows_per_batch = out_rows_per_batch; a.ld_out = ld_out; a.q_begin = q_begin;
    a.Lq_alloc = Lq_alloc > 0 ? Lq_alloc : Lkv;
    if (Lq_rows > a.Lq_alloc) return mm_fail("attention: Lq_rows=%d exceeds the q allocation %d", Lq_rows, a.Lq_alloc);
    a.scale_log2e = 0.08838834764831845f * 1.4426950408889634f;  // 1/sqrt(128) * log2(e)
    const int pairs = Hq * B;
    static const bool xcd_aware = [] { const char* e = getenv("MMADA_ATTN_XCD"); return !(e && e[0] == '0'); }();
    if (g_attn_form < 0) {
        const char* e = getenv("MMADA_ATTN_FORM");  // 0: round-2 issue order; 1: pipelined matrix blocks
        g_attn_form = e ? atoi(e) : 1;
    }
    if (g_attn_form == 2 || g_attn_form >= 20) return mm_fail("attention: form %d (attention64) was removed in round 5", g_attn_form);
    const int nq = (Lq_rows - q_begin + QB - 1) / QB;
    auto fn = g_attn_form == 1 ? attn4p_fwd_kernel<0> : attn_fwd_kernel;
#ifdef MMADA_TUNE
    switch (g_attn_form) {  // 11-13: diagnostic (wrong results), 14: static priority for the CU's second workgroup
        case 11: fn = attn4p_fwd_kernel<1>; break;
        case 12: fn = attn4p_fwd_kernel<2>; break;
        case 13: fn = attn4p_fwd_kernel<3>; break;
        case 14: fn = attn4p_fwd_kernel<4>; break;
    }
#endif
    if (g_attn_form != 0) {
        static MmOncePerDevice attr4;
        MM_ONCE_PER_DEVICE(attr4, if (attn4p_set_lds_limit()) return 1);
    }
    if (xcd_aware && pairs % 8 == 0) {
        a.xcd_pairs = pairs / 8; a.nq = nq;
        hipLaunchKernelGGL(fn, dim3(nq * pairs), dim3(256), ATT_LDS, s, a);
    } else {
        a.xcd_pairs = 0; a.nq = nq;
        hipLaunchKernelGGL(fn, dim3(nq, Hq, B), dim3(256), ATT_LDS, s, a);
    }
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}
