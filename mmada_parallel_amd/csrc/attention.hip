// attention.hip — unmasked, non-causal flash attention forward for gfx950 on v_mfma_f32_16x16x32_bf16
// (head_dim 128, bf16 in / out).
//
// Replaces F.scaled_dot_product_attention(q, k, v, attn_mask=None, is_causal=False) as the reference calls it
// (MMaDA-Parallel-A/model/modeling_llada.py:672-679 via :731-738; the attention-bias machinery around it is dead code).
//
// Layout contract (written by the QKV GEMM epilogue): q [B,Hq,Lq,128], k [B,Hkv,Lkv,128] row-major, V K-major as
// vT [B,Hkv,128,Lkv] with the keys of every 32-key block stored in the order of vt_key_pos (common.h): position 8g+j holds
// key 4g+j (j < 4) or key 16+4g+(j-4) — exactly the eight keys lane quad g of a wave holds of two neighbouring 16-key
// score tiles, so P goes from the S accumulators into the P·V operand without leaving its lane and V needs no transpose.
//
// Work decomposition.  The unit is a GROUP of 16 consecutive query rows of one (batch, head) pair.  A pair's groups are cut
// into `chunks` contiguous runs, one workgroup (8 waves, one per CU) each; inside the workgroup a wave takes 0-3 groups,
// dealt so that the two waves of a SIMD (w, w+4) together carry that SIMD's share.  At L = 2438, 32 heads: 153 groups per
// head, 8 workgroups per head, 19-20 groups per CU, 5 per SIMD — ONE round of 256 workgroups with every SIMD within 5 % of
// the mean, where 128-row tiles were 640 workgroups on 512 slots (1.58 rounds for 1.19 of work).  A query row's arithmetic
// does not depend on which wave, slot or workgroup carries its group: results are invariant to batch and row window.
//
//   S^T = K·Q^T   : A = 16 keys x 32 features of the K tile (LDS, one ds_read_b128), B = the group's Q rows (registers);
//                   lane (quad g, column q) ends up with the scores of keys 4g..4g+3 of the tile for query q.
//   O^T = V^T·P^T : A = 16 features x 32 key positions of the vT tile (LDS), B = P straight from the S registers.
// One K / vT fragment feeds every group of the wave (up to three MFMAs per ds_read_b128).
//
// The two waves of a SIMD share its matrix pipe, and all eight share one barrier per key tile; to keep them out of lockstep
// (both in their matrix phases, then both in their soft-max) waves 4-7 run LATE: the P·V of a tile is issued one interval
// behind, so their loop is { P·V(t-1), S(t), soft-max(t) } beside { S(t), soft-max(t), P·V(t) } of waves 0-3 — a wave's
// soft-max then sits beside its partner's matrix phase.  vT tiles therefore live one interval longer (ring of three; K two).
// K / vT tiles (64 keys) arrive by LDS-DMA (global_load_lds_dwordx4, asm: attention.h) one tile ahead; the 16-byte XOR swizzle
// is applied to the DMA source address and to the read address (both products read conflict-free).
//
// Soft-max: fp32, log2 domain, per-lane partial row sums (combined over the four lanes of a row once, at the end).  The
// running maximum is only raised when some score of the GROUP exceeds it by more than 2^DEFER_LOG2 — a wave-wide vote on
// lane-local maxima, so the common path has no cross-lane exchange; the group is the unit of that decision, which is why a
// row's bits depend on its 16-row group (fixed by the row index) and on nothing else.
#include "attention.h"

namespace {

using namespace attn_detail;

constexpr int STAGE = 16384;           // one K or vT tile: 64 keys x 128 features, bf16
constexpr int LDS_V0 = 2 * STAGE;      // K stages at 0 / 16 KiB, vT stages at 32 / 48 / 64 KiB
constexpr int ATT16_LDS = 5 * STAGE;   // 80 KiB

MM_DEVICE f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

struct Wave16 {
    const bf16_t* Qp;   // q rows of this (batch, head)
    const char* Kp;     // k rows of the kv head
    const char* Vp;     // vT rows of the kv head
    bf16_t* out;        // a.out + batch offset + head column
    int row0;           // first query row of the wave's groups
    int wave, lane;
    int piece0, npiece; // this wave's LDS-DMA pieces of every K / vT tile: [piece0, piece0 + npiece), npiece <= 3
};

template <int NG, bool LATE>
MM_DEVICE void attn16_wave(const AttnArgs& a, const Wave16& w) {
    const int lane = w.lane, qi = lane & 15, quad = lane >> 4;
    constexpr int NGA = NG > 0 ? NG : 1;  // array extents (NG = 0: a wave that only moves tiles)

    bf16x8 qf[NGA][4];
    f32x4 o[8][NGA];
    float m_run[NGA], l_run[NGA];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int row = min(w.row0 + g * 16 + qi, a.Lq_alloc - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[g][ks] = *(const bf16x8*)(w.Qp + (size_t)row * 128 + ks * 32 + quad * 8);
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) o[dt][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        m_run[g] = -1e30f;
        l_run[g] = 0.f;
    }

    // LDS-DMA: wave w moves K pieces 2w, 2w+1 (4 key rows x 256 B each) and vT pieces 2w, 2w+1 (8 feature rows x 128 B)
    // LDS-DMA: a tile is 16 K pieces (4 key rows x 256 B) + 16 vT pieces (8 feature rows x 128 B); a SIMD's two waves move four of
    // each between them — two and two, or one and three when the first wave carries one group more than the second (the wave
    // with fewer groups has the time: a piece costs its issuer 60-185 cycles, MI355X guide)
    unsigned koff[3], voff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int p = w.piece0 + i;
        const int kr = p * 4 + (lane >> 4);
        koff[i] = (unsigned)(kr * 128 + (((lane & 15) ^ (kr & 15)) << 3)) * 2u;
        const int d = p * 8 + (lane >> 3);
        voff[i] = (unsigned)(d * a.Lkv + (((lane & 7) ^ ((d >> 1) & 7)) << 3)) * 2u;
    }
    auto stage = [&](int kt, int kst, int vst) {
        const char* kb = w.Kp + (size_t)kt * KB * 256;
        const char* vb = w.Vp + (size_t)kt * KB * 2;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < w.npiece) dma16(kb, koff[i], kst * STAGE + (w.piece0 + i) * 1024);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < w.npiece) dma16(vb, voff[i], LDS_V0 + vst * STAGE + (w.piece0 + i) * 1024);
    };
    // per-lane read offsets inside a tile: K row qi (+16 per score tile), 16-byte chunk (4 ks + quad) ^ row;
    //                                     vT row qi (+16 per feature tile), chunk (4 k2 + quad) ^ (row >> 1)
    int kro[4], vro[2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kro[ks] = qi * 256 + (((ks * 4 + quad) ^ qi) << 4);
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) vro[k2] = qi * 128 + (((k2 * 4 + quad) ^ (qi >> 1)) << 4);

    bf16x8 pb[NGA][2];
    const int nkt = (a.L + KB - 1) / KB;

    auto pv = [&](int vbase) {  // O^T += V^T · P^T of the tile whose vT sits at LDS byte vbase
        if constexpr (NG > 0) {
            A8_SB();
            int va[2];
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) va[k2] = vro[k2] + vbase;
            bf16x8 vf[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) vf[n] = lds_frag(va[0] + n * 2048);
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const int k2 = n >> 3, dt = n & 7;
#pragma unroll
                for (int g = 0; g < NG; ++g) o[dt][g] = mfma16(vf[n & 3], pb[g][k2], o[dt][g]);
                if (n + 4 < 16) vf[n & 3] = lds_frag(va[(n + 4) >> 3] + ((n + 4) & 7) * 2048);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 1);
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                __builtin_amdgcn_sched_group_barrier(0x008, NG, 1);
                if (n + 4 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
            }
            A8_SB();
        }
    };

    auto scores = [&](int kt, int kbase) {  // S(kt) and its soft-max -> pb
        if constexpr (NG > 0) {
            A8_SB();
            int ka[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ka[ks] = kro[ks] + kbase;
            f32x4 s[4][NGA];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int g = 0; g < NG; ++g) s[t][g] = f32x4{0.f, 0.f, 0.f, 0.f};
            bf16x8 kf[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) kf[n] = lds_frag(ka[0] + n * 4096);
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const int ks = n >> 2, t = n & 3;
#pragma unroll
                for (int g = 0; g < NG; ++g) s[t][g] = mfma16(kf[n & 3], qf[g][ks], s[t][g]);
                if (n + 4 < 16) kf[n & 3] = lds_frag(ka[(n + 4) >> 2] + ((n + 4) & 3) * 4096);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                __builtin_amdgcn_sched_group_barrier(0x008, NG, 0);
                if (n + 4 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            A8_SB();
            // keys past L (only in the last tile) get -inf; select, not arithmetic, so garbage K rows cannot leak NaN
            if (kt * KB + KB > a.L) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kt * KB + t * 16 + quad * 4 + r;
#pragma unroll
                        for (int g = 0; g < NG; ++g)
                            if (key >= a.L) s[t][g][r] = -INFINITY;
                    }
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                // lane-local maximum of the 16 scores: plain fmaxf (this unit is built without NaN semantics, so hipcc nests them
                // into v_max3_f32 without canonicalising v_max x,x, and — unlike an asm maximum — places the MFMA-result wait states)
                float mxa = fmaxf(s[0][g][0], s[1][g][0]), mxb = fmaxf(s[2][g][0], s[3][g][0]);
#pragma unroll
                for (int r = 1; r < 4; ++r) {
                    mxa = fmaxf(fmaxf(mxa, s[0][g][r]), s[1][g][r]);
                    mxb = fmaxf(fmaxf(mxb, s[2][g][r]), s[3][g][r]);
                }
                const float mx = fmaxf(mxa, mxb);
                if (!__all(mx * a.scale_log2e - m_run[g] <= DEFER_LOG2)) {  // wave-uniform, per group
                    float rm = fmaxf(mx, __shfl_xor(mx, 16, 64));
                    rm = fmaxf(rm, __shfl_xor(rm, 32, 64)) * a.scale_log2e;   // the row's maximum (4 lanes share a row)
                    const float m_new = fmaxf(m_run[g], rm);
                    const float alpha = __builtin_amdgcn_exp2f(m_run[g] - m_new);
                    l_run[g] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[dt][g][r] *= alpha;
                    m_run[g] = m_new;
                }
                float ps[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[t][g][r] = __builtin_amdgcn_exp2f(s[t][g][r] * a.scale_log2e - m_run[g]);
                    ps[t] = (s[t][g][0] + s[t][g][1]) + (s[t][g][2] + s[t][g][3]);
                }
                l_run[g] += (ps[0] + ps[1]) + (ps[2] + ps[3]);
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        pb[g][k2][j] = (__bf16)s[2 * k2][g][j];
                        pb[g][k2][4 + j] = (__bf16)s[2 * k2 + 1][g][j];
                    }
            }
        }
    };

    stage(0, 0, 0);
    // first use of the Q fragments HERE, in front of the loop: hipcc otherwise sinks their s_waitcnt vmcnt(N) into the loop,
    // where N counts only the loads it knows — the LDS-DMA requests in flight there would make that wait a stall per tile
#pragma unroll
    for (int g = 0; g < NG; ++g)
        asm volatile("" : "+v"(qf[g][0]), "+v"(qf[g][1]), "+v"(qf[g][2]), "+v"(qf[g][3]));
    int vst = 0;  // vT ring stage of tile kt
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int vnext = vst == 2 ? 0 : vst + 1;
        if (kt + 1 < nkt) stage(kt + 1, (kt + 1) & 1, vnext);
        if constexpr (LATE) {
            if (kt > 0) pv(LDS_V0 + (vst == 0 ? 2 : vst - 1) * STAGE);
        }
        scores(kt, (kt & 1) * STAGE);
        if constexpr (!LATE) pv(LDS_V0 + vst * STAGE);
        vst = vnext;
    }
    if constexpr (LATE) pv(LDS_V0 + (vst == 0 ? 2 : vst - 1) * STAGE);

    // ---- normalise and store: lane holds O[row][d = 16 dt + 4 quad + r]; two neighbouring feature tiles are exchanged between
    // lane quads (v_permlane16_swap) so that every lane stores 8 consecutive features = one 16-byte access (guide T21):
    // quad 0 / 2: features 0-7 / 8-15 of the even tile, quad 1 / 3: of the odd tile ----
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        float l = l_run[g];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        const int row = w.row0 + g * 16 + qi;
        bf16_t* orow = w.out + (size_t)(row - a.q_begin) * a.ld_out + (quad & 1) * 16 + (quad >> 1) * 8;
#pragma unroll
        for (int dp = 0; dp < 4; ++dp) {
            uint32_t a0 = pack_bf2(o[2 * dp][g][0] * inv, o[2 * dp][g][1] * inv), a1 = pack_bf2(o[2 * dp][g][2] * inv, o[2 * dp][g][3] * inv);
            uint32_t b0 = pack_bf2(o[2 * dp + 1][g][0] * inv, o[2 * dp + 1][g][1] * inv), b1 = pack_bf2(o[2 * dp + 1][g][2] * inv, o[2 * dp + 1][g][3] * inv);
            {
                const auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                a0 = r0[0]; b0 = r0[1]; a1 = r1[0]; b1 = r1[1];
            }
            if (row < a.Lq_rows) *(u32x4*)(orow + dp * 32) = u32x4{a0, a1, b0, b1};
        }
    }
}

__global__ __launch_bounds__(512, 2) void attn16_kernel(AttnArgs a) {
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup -> (pair, chunk).  XCD-aware form: hardware workgroup ids round-robin over the 8 XCDs (observed, speed only),
    // so XCD x takes the pairs [x * xcd_pairs, (x + 1) * xcd_pairs) with all their chunks: a head's K / vT is fetched into ONE L2.
    int pair, chunk;
    if (a.xcd_pairs > 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        pair = xcd * a.xcd_pairs + j / a.chunks;
        chunk = j % a.chunks;
    } else {
        pair = blockIdx.x / a.chunks;
        chunk = blockIdx.x % a.chunks;
    }
    const int b = pair / a.Hq, h = pair - b * a.Hq;
    const int hkv = h / (a.Hq / a.Hkv);
    // the chunk's groups [gbeg, gbeg + n); SIMD s (waves s, s+4) takes n/4 (+1) of them, the early wave the larger half
    const int cb = a.groups / a.chunks, cr = a.groups % a.chunks;
    const int gbeg = chunk * cb + min(chunk, cr), n = cb + (chunk < cr ? 1 : 0);
    const int simd = wave & 3, late = a.plain_order ? 0 : wave >> 2;
    const int sq = n >> 2, sr = n & 3;
    const int ls = sq + (simd < sr ? 1 : 0), sbeg = simd * sq + min(simd, sr);
    const int n_early = (ls + 1) >> 1;
    const int my_n = (wave >> 2) ? ls - n_early : n_early;
    const int my_beg = gbeg + sbeg + ((wave >> 2) ? n_early : 0);

    Wave16 w;
    w.Qp = a.q + (size_t)(b * a.Hq + h) * a.Lq_alloc * 128;
    w.Kp = (const char*)(a.k + (size_t)(b * a.Hkv + hkv) * a.Lkv * 128);
    w.Vp = (const char*)(a.vT + (size_t)(b * a.Hkv + hkv) * 128 * a.Lkv);
    w.out = a.out + (size_t)b * a.out_rows_per_batch * a.ld_out + h * 128;
    w.row0 = a.q_begin + my_beg * 16;
    w.wave = wave;
    w.lane = tid & 63;
    {   // the pair's four pieces: 2 + 2, or 1 + 3 when the first wave has a group more
        const int first = (n_early > ls - n_early) ? 1 : 2;
        w.piece0 = simd * 4 + ((wave >> 2) ? first : 0);
        w.npiece = (wave >> 2) ? 4 - first : first;
    }
    if (!late) {
        switch (my_n) {
            case 0: attn16_wave<0, false>(a, w); break;
            case 1: attn16_wave<1, false>(a, w); break;
            case 2: attn16_wave<2, false>(a, w); break;
            default: attn16_wave<3, false>(a, w); break;
        }
    } else {
        switch (my_n) {
            case 0: attn16_wave<0, false>(a, w); break;
            case 1: attn16_wave<1, true>(a, w); break;
            case 2: attn16_wave<2, true>(a, w); break;
            default: attn16_wave<3, true>(a, w); break;
        }
    }
}


}  // namespace

// Workgroups per (batch, head) pair: the smallest count with the least estimated time  rounds x (largest SIMD share x key tiles x
// T + F): T = 0.405 us per group and key tile, F = 5.7 us of prologue + epilogue per round of 256 workgroups (measured:
// profiles/r06_attention_time_vs_L.txt).  A workgroup holds at most 24 groups (8 waves x 3).
int attention_chunks(int pairs, int groups, int keys) {
    const double nkt = (keys + KB - 1) / KB;
    int best = 0;
    double best_cost = 0;
    for (int c = (groups + 23) / 24; c <= groups; ++c) {
        const int n = (groups + c - 1) / c;
        const long rounds = ((long)pairs * c + 255) / 256;
        const double cost = rounds * (((n + 3) / 4) * nkt * 0.405 + 5.7);
        if (!best || cost < best_cost - 1e-9) { best = c; best_cost = cost; }
        if (n == 1) break;
    }
    return best ? best : 1;
}

static std::atomic<int> g_attn_form{-1};  // -1: read MMADA_ATTN_FORM once
void attention_force_form(int form) { g_attn_form = form; }  // measurement / test hook; -1: back to MMADA_ATTN_FORM / default

// form 1 (default): waves 4-7 run LATE (P·V one interval behind); form 0: every wave in the plain order — the same arithmetic
// per query row, bit-identical output (tests/test_gpu_kernels.py::test_attention_forms_are_bit_identical).
int launch_attention(const bf16_t* q, const bf16_t* k, const bf16_t* vT, bf16_t* out, int B, int Hq, int Hkv, int L,
                     int Lq_rows, int Lkv, int out_rows_per_batch, int ld_out, hipStream_t s, int q_begin, int Lq_alloc) {
    if (L <= 0 || B <= 0) return 0;
    if (Lkv % 64 || Lkv < L) return mm_fail("attention: Lkv=%d must be a multiple of 64 and >= L=%d", Lkv, L);
    if (Hq % Hkv) return mm_fail("attention: n_heads %% n_kv_heads != 0");
    if (q_begin < 0 || (q_begin & 31) || q_begin >= Lq_rows) return mm_fail("attention: bad q_begin=%d", q_begin);
    int form = g_attn_form.load(std::memory_order_relaxed);
    if (form < 0) {
        const char* e = getenv("MMADA_ATTN_FORM");
        form = e ? atoi(e) : 1;
        g_attn_form.store(form, std::memory_order_relaxed);
    }
    if (form != 0 && form != 1) return mm_fail("attention: form %d does not exist (0: plain order, 1: late waves)", form);
    static MmOncePerDevice attr_set;
    MM_ONCE_PER_DEVICE(attr_set, MM_CHECK_HIP(hipFuncSetAttribute((const void*)attn16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ATT16_LDS)));
    AttnArgs a{};
    a.q = q; a.k = k; a.vT = vT; a.out = out;
    a.Hq = Hq; a.Hkv = Hkv; a.L = L; a.Lq_rows = Lq_rows; a.Lkv = Lkv;
    a.out_rows_per_batch = out_rows_per_batch; a.ld_out = ld_out; a.q_begin = q_begin;
    a.Lq_alloc = Lq_alloc > 0 ? Lq_alloc : Lkv;
    if (Lq_rows > a.Lq_alloc) return mm_fail("attention: Lq_rows=%d exceeds the q allocation %d", Lq_rows, a.Lq_alloc);
    a.scale_log2e = 0.08838834764831845f * 1.4426950408889634f;  // 1/sqrt(128) * log2(e)
    a.plain_order = form == 0;
    const int pairs = Hq * B;
    a.groups = (Lq_rows - q_begin + 15) / 16;
    a.chunks = attention_chunks(pairs, a.groups, L);
    a.xcd_pairs = (pairs % 8 == 0) ? pairs / 8 : 0;
    hipLaunchKernelGGL(attn16_kernel, dim3(pairs * a.chunks), dim3(512), ATT16_LDS, s, a);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}
