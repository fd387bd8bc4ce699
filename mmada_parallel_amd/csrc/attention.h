// attention.h — what the attention kernels of attention.hip (4 waves x 32 query rows) and attention64.hip (4 waves x 64
// query rows, one wave per SIMD) share: tile constants, launch arguments, single-instruction maxima, LDS addressing.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace attn_detail {

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
    // attn64 only (attn64_plan): the first n_full workgroups run full passes (256 query rows, `full_per_pair` per (batch,
    // head) pair, q-blocks [0, 8 * full_per_pair)), the others half passes (128 rows, `half_per_pair` per pair)
    int n_full, full_per_pair, half_per_pair;
    // attn4p only — key-split tail (round 5, attn_split_plan below).  Per (batch, head) pair the grid holds `nq_full` full
    // workgroups (rows q_begin + 128 i, all keys, limited to rows < full_end) and `nq_tail` x `nsplit` split workgroups (rows
    // tail_begin + 128 j < Lq_rows, key tiles [kt0, kt1) of split s); nsplit == 1: no tail region (nq_tail == 0)
    int nq_full, nq_tail, nsplit, full_end, tail_begin, pairs;
    int skip_idle;        // != 0: waves without a single live query row skip the matrix blocks and the soft-max
    float* part;          // [pair][tail tile][split][wave][66][64] fp32: O (64 registers), m_run, l_run per lane
    unsigned* counters;   // [pair][tail tile]: splits that have published; the last one combines and resets it to 0
};

// Key-split of the LAST query tiles of a sequence (round 5).  A batch-1 forward at L = 2438 is 20 query tiles x 32 heads = 640
// workgroups on 512 slots: one full round + a quarter round that runs one workgroup per CU at 0.58 of a round's time (1.58 rounds
// for 1.19 of work).  Rows >= r_full (a multiple of 128 x 16: the tiles that fill whole rounds for 32 heads) are therefore cut
// `nsplit` ways along the KEYS: each split workgroup runs the same online soft-max over its key tiles and publishes (O, m, l) in
// fp32; the last one to arrive combines them in split order (deterministic) and stores.  The plan is a function of L ALONE — not
// of the batch, the window or the head count — so a row's arithmetic never depends on what shares its launch (batch invariance,
// consumed-row window bit-identity: tests/test_gpu_fullsize.py).  nsplit == 1: no split.
struct AttnSplitPlan { int r_full, nsplit; };
inline AttnSplitPlan attn_split_plan(int L) {
    const int nq = (L + QB - 1) / QB, nkt = (L + KB - 1) / KB;
    const int t_full = nq / 16 * 16, nt = nq - t_full;
    AttnSplitPlan p{0, 1};
    if (t_full == 0 || nt == 0) return p;
    int s = 16 / nt;
    if (s > 8) s = 8;
    while (s >= 2 && nkt / s < 4) --s;
    if (s < 2) return p;
    p.r_full = t_full * QB;
    p.nsplit = s;
    return p;
}
constexpr size_t ATTN_SPLIT_HDR = 65536;                       // counters (16 384 of them), at the very start of the workspace
constexpr size_t ATTN_PART_FLOATS = 4 * 66 * 64;               // one split workgroup's partial


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

// One 16-byte-per-lane LDS-DMA request as an asm statement.  With the BUILTIN (__builtin_amdgcn_global_load_lds) hipcc places
// `s_waitcnt lgkmcnt(0)` in front of the next asm statement that reads a register an LDS read is still writing — it gives up
// counting once an LDS-writing VMEM operation is in flight next to an asm it cannot see into (tools/dbg/wc reproduces that in
// 20 lines) — so every K / vT fragment wait in the tile loop drained the whole read queue: 24 % of the wave's cycles parked
// (SQ_WAIT_ANY, profiles/r04_pmc_sq_attention_forms.txt).  A request the compiler does not know leaves its own, COUNTED
// lgkmcnt(N) waits in place.  The tile barrier's explicit `s_waitcnt vmcnt(0)` is what orders the DMA, as before; the
// "memory" clobber keeps the compiler's LDS reads on their side of the statement.  M0: hipcc reserves it and ignores it in a
// clobber list; nothing else in these kernels reads it (tools/isa_check.py: check_m0, on the compiled code).
MM_DEVICE void dma16(const char* sbase, unsigned voff, int lds_byte) {
    const int m0v = __builtin_amdgcn_readfirstlane(lds_byte);
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(m0v) : "memory");
}

}  // namespace attn_detail

// attention64.hip
int launch_attention64(attn_detail::AttnArgs a, int B, hipStream_t s, int var = 0);
