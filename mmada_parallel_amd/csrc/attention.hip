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
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace {

constexpr int QB = 128;  // query rows per workgroup
constexpr int KB = 64;   // keys per tile
constexpr int TILE_BYTES = KB * 128 * 2;  // 16 KiB (K tile == vT tile)
constexpr int ATT_LDS = 4 * TILE_BYTES;   // 2 stages x (K + vT)
constexpr float DEFER_LOG2 = 4.0f;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// 3-input / 2-input fp32 max as single instructions: hipcc wraps fmaxf() on MFMA outputs in canonicalising
// v_max_f32 x,x (one extra VALU op per score); scores are never signalling NaNs here.
MM_DEVICE float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
MM_DEVICE float fmax_nc(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

struct AttnArgs {
    const bf16_t* q;
    const bf16_t* k;
    const bf16_t* vT;
    bf16_t* out;
    int Hq, Hkv, L, Lq_rows, Lkv, out_rows_per_batch, ld_out;
    int q_begin;  // first query row (multiple of 32); output row of query r is b*out_rows_per_batch + r - q_begin
    int Lq_alloc; // rows per (batch, head) of q: Lkv, or the compact length of a cache step's queries
    float scale_log2e;
    int xcd_pairs, nq;  // XCD-aware 1-D grid: (batch, head) pairs per XCD and query tiles per pair (0: plain 3-D grid)
};

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

        // ---- O^T += V^T · P^T ----
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const char* vrow = Vt + (db * 32 + ql) * 128;  // swizzle of row db*32+ql does not depend on db
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 va = *(const bf16x8*)(vrow + (((4 * t + 2 * s2 + hi) ^ vsw) << 4));
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pb[t][s2], o[db], 0, 0, 0);
                }
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

// ---- 8-wave "ping-pong" form (round 3) -----------------------------------------------------------------------------
// Same arithmetic per wave, in the same order, as attn_fwd_kernel (bit-identical output), scheduled the way the 8-phase
// GEMM is: ONE workgroup of eight waves = 256 query rows per CU; waves 0-3 and waves 4-7 (one of each on every SIMD) run
// one barrier apart.  A wave's key-tile iteration is two barrier-separated phases
//     A_i : matrix block   S(i) = K(i)·Q^T  (16 MFMA)  +  O += V(i-1)^T·P(i-1)  (16 MFMA)     ~1024 MFMA cycles
//     B_i : soft-max of tile i (row max, rescale decision, 32 exp2, P -> bf16)                ~900 VALU cycles
// so while one wave of a SIMD streams MFMAs its partner does the VALU / transcendental work of its soft-max — the matrix
// pipe and the vector ALU of a SIMD are separate pipes — instead of the two meeting in the same phase by chance, as the two
// independent 4-wave workgroups of a CU did (a wave spent ~1000 of ~3400 cycles per tile parked at the tile barrier,
// profiles/r02_pmc_sq_model.txt).  K / vT tiles live in two 3-slot LDS rings (96 KiB) shared by all eight waves; a wave issues
// its LDS-DMA pieces of K(i+2) and vT(i+1) at the top of A_i and retires the previous iteration's pieces with a COUNTED
// vmcnt before the barrier that ends A_i, so every piece has a whole iteration (two phases) to land.
//   RAW: K(i+1) / vT(i) are first read in A_(i+1) of the early group, i.e. after the barrier that ends the late group's A_i;
//   WAR: slot (i+2) % 3 held K(i-1), last read in the late group's A_(i-1), which ended at least one barrier before the
//        early group's A_i — and a ds_read of a phase has completed when the phase's MFMAs that consume it have issued.
constexpr int QB8 = 256, NS8 = 3;
constexpr int ATT8_LDS = 2 * NS8 * TILE_BYTES;

// The kernel owns its whole LDS allocation and has no static __shared__ object: the dynamic segment starts at LDS address 0
// (tests/test_isa.py), so LDS addresses are plain integers — no "base + offset" VALU add per access.
typedef __attribute__((address_space(3))) const bf16x8* lds_frag_ptr;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
#pragma clang diagnostic ignored "-Wint-to-void-pointer-cast"
MM_DEVICE bf16x8 lds_frag(int byte_off) { return *(lds_frag_ptr)(uint32_t)byte_off; }
MM_DEVICE lptr_t lds_at(int byte_off) { return (lptr_t)(uint32_t)byte_off; }
#pragma clang diagnostic pop

#define A8_SB() __builtin_amdgcn_sched_barrier(0)
#define A8_BARRIER()                            \
    do {                                        \
        A8_SB();                                \
        asm volatile("s_barrier" ::: "memory"); \
        A8_SB();                                \
    } while (0)

__global__ __launch_bounds__(512, 2) void attn8_fwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;  // 0: early group, 1: one barrier behind
    const int ql = lane & 31, hi = lane >> 5;
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

    const int q_row = a.q_begin + qb * QB8 + wave * 32 + ql;
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

    // LDS-DMA: wave w moves K pieces 2w, 2w+1 (4 rows x 256 B each) and vT pieces 2w, 2w+1 (8 rows x 128 B)
    unsigned koff[2], voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int kr = (wave * 2 + i) * 4 + (lane >> 4);
        koff[i] = (unsigned)(kr * 128 + (((lane & 15) ^ (kr & 15)) << 3)) * 2u;
        const int d = (wave * 2 + i) * 8 + (lane >> 3);
        voff[i] = (unsigned)(d * a.Lkv + (((lane & 7) ^ ((d >> 1) & 7)) << 3)) * 2u;
    }
    // The ring slot of a tile is its index mod 3; the loop below is unrolled by three so that every slot is a
    // compile-time constant and every LDS address is a precomputed per-lane offset plus an immediate (no VALU address
    // arithmetic inside the matrix block: the partner wave of the SIMD is doing its soft-max on the same vector ALU).
    auto stage_k = [&](auto slot_, int kt) {
        constexpr int SLOT = decltype(slot_)::value;
        const char* kb = (const char*)Kp + (size_t)kt * KB * 256;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned off = koff[i];
            asm volatile("" : "+s"(kb), "+v"(off));  // scalar base + 32-bit lane offset, zero-extended here: saddr form
            __builtin_amdgcn_global_load_lds((gptr_t)(kb + off), lds_at(SLOT * TILE_BYTES + wave * 2048 + i * 1024), 16, 0, 0);
        }
    };
    auto stage_v = [&](auto slot_, int kt) {
        constexpr int SLOT = decltype(slot_)::value;
        const char* vb = (const char*)Vp + (size_t)kt * KB * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned off = voff[i];
            asm volatile("" : "+s"(vb), "+v"(off));
            __builtin_amdgcn_global_load_lds((gptr_t)(vb + off), lds_at((NS8 + SLOT) * TILE_BYTES + wave * 2048 + i * 1024), 16, 0, 0);
        }
    };
    // per-lane LDS read offsets: K fragment of k-step s (keys ql / 32+ql: + 8192), vT fragment (t, s2) of feature block db (+ db*4096)
    int kro[8], vro[4];
    {
        const int ksw = ql & 15, vsw = (ql >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 8; ++s) kro[s] = ql * 256 + (((2 * s + hi) ^ ksw) << 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) vro[j] = NS8 * TILE_BYTES + ql * 128 + (((2 * j + hi) ^ vsw) << 4);  // j = 2*t + s2
    }

    const int nkt = (a.L + KB - 1) / KB;
    stage_k(std::integral_constant<int, 0>{}, 0);
    if (nkt > 1) stage_k(std::integral_constant<int, 1>{}, 1);
    stage_v(std::integral_constant<int, 0>{}, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    A8_BARRIER();
    if (grp == 1) A8_BARRIER();  // waves 4-7 run one barrier behind waves 0-3

    f32x16 s0, s1;
    bf16x8 pb[2][2];
    // One key-tile iteration; R = i % 3.  FULL: 1 <= i and i + 2 < nkt — nothing is conditional and the matrix block is one
    // basic block whose issue order is pinned (sched_group_barrier); otherwise the guarded form for the first iteration
    // and the last three.  Returns true after the last matrix block (i == nkt).
    auto iteration = [&](auto r_, auto full_, int i) -> bool {
        constexpr int R = decltype(r_)::value;
        constexpr bool FULL = decltype(full_)::value;
        constexpr int KS = R * TILE_BYTES, VS = ((R + 2) % 3) * TILE_BYTES;  // slots read: K(i), vT(i-1)
        using KNext = std::integral_constant<int, (R + 2) % 3>;              // slots written: K(i+2), vT(i+1)
        using VNext = std::integral_constant<int, (R + 1) % 3>;
        // the read offsets are "redefined" here (no instruction): otherwise loop-invariant code motion materialises every
        // offset + slot constant of the three unrolled iterations in a register of its own (seen: 100 spilled registers);
        // like this each access is offset register + immediate
        asm volatile("" : "+v"(kro[0]), "+v"(kro[1]), "+v"(kro[2]), "+v"(kro[3]), "+v"(kro[4]), "+v"(kro[5]), "+v"(kro[6]), "+v"(kro[7]));
        asm volatile("" : "+v"(vro[0]), "+v"(vro[1]), "+v"(vro[2]), "+v"(vro[3]));
        // ======== A_i: matrix block — fragments are fetched three k-steps / four MFMAs ahead of their use ========
        const bool do_s = FULL || i < nkt, do_pv = FULL || i > 0;
        bf16x8 ka[3][2], va[4];
        if (do_s) {
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                ka[s][0] = lds_frag(kro[s] + KS);
                ka[s][1] = lds_frag(kro[s] + KS + 8192);
            }
        }
        int issued = 0;
        if (FULL || i + 2 < nkt) { stage_k(KNext{}, i + 2); issued += 2; }
        if (FULL || i + 1 < nkt) { stage_v(VNext{}, i + 1); issued += 2; }
        if (do_s) {  // S^T = K(i) · Q^T for keys [0,32) and [32,64) of the tile
#pragma unroll
            for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[s % 3][0], qf[s], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[s % 3][1], qf[s], s1, 0, 0, 0);
                if (s + 3 < 8) {
                    ka[s % 3][0] = lds_frag(kro[s + 3] + KS);
                    ka[s % 3][1] = lds_frag(kro[s + 3] + KS + 8192);
                } else if (do_pv && s < 7) {  // the first vT fragments, while the last S MFMAs run
                    va[2 * (s - 5)] = lds_frag(vro[2 * (s - 5)] + VS);
                    va[2 * (s - 5) + 1] = lds_frag(vro[2 * (s - 5) + 1] + VS);
                }
            }
        } else if (do_pv) {
#pragma unroll
            for (int j = 0; j < 4; ++j) va[j] = lds_frag(vro[j] + VS);
        }
        if (do_pv) {  // O^T += V(i-1)^T · P(i-1)^T : MFMA n = db*4 + j uses fragment j = 2*t + s2 of feature block db
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const int db = n >> 2, j = n & 3;
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[j], pb[j >> 1][j & 1], o[db], 0, 0, 0);
                if (n + 4 < 16) va[j] = lds_frag(vro[j] + VS + (db + 1) * 4096);
            }
        }
        if (FULL) {  // pin the issue order of the block: reads run ahead of the MFMAs that consume them
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);   // K fragments of k-steps 0-2
            __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);   // the four LDS-DMA pieces
#pragma unroll
            for (int s = 0; s < 7; ++s) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        // the pieces this wave issued in the PREVIOUS iteration (K(i+1), vT(i)) have landed; this iteration's stay in flight
        A8_SB();
        if (issued == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (issued == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        A8_BARRIER();
        if (!FULL && i == nkt) return true;

        // ======== B_i: online soft-max of tile i (fp32, log2 domain); lane and lane^32 share a query ========
        if (!FULL && i * KB + KB > a.L) {  // keys past L (only in the last tile) get -inf; select, so garbage K rows cannot leak NaN
            const int kbase = i * KB + 4 * hi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + (r & 3) + 8 * (r >> 2);
                if (key >= a.L) s0[r] = -INFINITY;
                if (key + 32 >= a.L) s1[r] = -INFINITY;
            }
        }
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
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                pb[0][s2][j] = (__bf16)s0[8 * s2 + j];
                pb[1][s2][j] = (__bf16)s1[8 * s2 + j];
            }
        A8_BARRIER();
        return false;
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    iteration(I0{}, std::false_type{}, 0);
    int i = 1;
    for (; i + 4 < nkt; i += 3) {  // i % 3 == 1 here; all three iterations have i + 2 < nkt
        iteration(I1{}, std::true_type{}, i);
        iteration(I2{}, std::true_type{}, i + 1);
        iteration(I0{}, std::true_type{}, i + 2);
    }
    for (;; i += 3) {              // the last (up to five) iterations, guarded
        if (iteration(I1{}, std::false_type{}, i)) break;
        if (iteration(I2{}, std::false_type{}, i + 1)) break;
        if (iteration(I0{}, std::false_type{}, i + 2)) break;
    }
    if (grp == 0) A8_BARRIER();

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

}  // namespace

static int g_attn_form = -1;  // -1: read MMADA_ATTN8 once
void attention_force_form(int form) { g_attn_form = form; }  // -1: back to MMADA_ATTN8 / default  // measurement / test hook: 0 = 4-wave, 1 = 8-wave ping-pong

int launch_attention(const bf16_t* q, const bf16_t* k, const bf16_t* vT, bf16_t* out, int B, int Hq, int Hkv, int L,
                     int Lq_rows, int Lkv, int out_rows_per_batch, int ld_out, hipStream_t s, int q_begin, int Lq_alloc) {
    if (L <= 0 || B <= 0) return 0;
    if (Lkv % 64 || Lkv < L) return mm_fail("attention: Lkv=%d must be a multiple of 64 and >= L=%d", Lkv, L);
    if (Hq % Hkv) return mm_fail("attention: n_heads %% n_kv_heads != 0");
    if (q_begin < 0 || (q_begin & 31) || q_begin >= Lq_rows) return mm_fail("attention: bad q_begin=%d", q_begin);
    static bool attr_set[16] = {};
    if (mm_first_use_on_device(attr_set)) {
        MM_CHECK_HIP(hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS));
    }
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
        const char* e = getenv("MMADA_ATTN8");  // 0: the 4-wave kernel (two independent workgroups per CU); 1: 8-wave ping-pong
        g_attn_form = e ? atoi(e) : 1;
    }
    if (g_attn_form == 1) {
        static bool attr8[16] = {};
        if (mm_first_use_on_device(attr8))
            MM_CHECK_HIP(hipFuncSetAttribute((const void*)attn8_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ATT8_LDS));
        const int nq = (Lq_rows - q_begin + QB8 - 1) / QB8;
        a.nq = nq;
        a.xcd_pairs = (xcd_aware && pairs % 8 == 0) ? pairs / 8 : 0;
        if (a.xcd_pairs) hipLaunchKernelGGL(attn8_fwd_kernel, dim3(nq * pairs), dim3(512), ATT8_LDS, s, a);
        else hipLaunchKernelGGL(attn8_fwd_kernel, dim3(nq, Hq, B), dim3(512), ATT8_LDS, s, a);
        MM_CHECK_HIP(hipGetLastError());
        return 0;
    }
    const int nq = (Lq_rows - q_begin + QB - 1) / QB;
    if (xcd_aware && pairs % 8 == 0) {
        a.xcd_pairs = pairs / 8; a.nq = nq;
        hipLaunchKernelGGL(attn_fwd_kernel, dim3(nq * pairs), dim3(256), ATT_LDS, s, a);
    } else {
        a.xcd_pairs = 0; a.nq = nq;
        hipLaunchKernelGGL(attn_fwd_kernel, dim3(nq, Hq, B), dim3(256), ATT_LDS, s, a);
    }
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}
