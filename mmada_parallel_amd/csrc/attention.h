// attention.h — tile constants, launch arguments and LDS addressing of the attention kernel (attention.hip: 8 waves per
// workgroup, 0-3 groups of 16 query rows per wave, v_mfma_f32_16x16x32_bf16).  Rounds 2-5 shipped a 4-wave x 32-row kernel on
// v_mfma_f32_32x32x16_bf16 (and, in round 4, a 64-row one-wave-per-SIMD form); profiles/HISTORY.md keeps what they taught.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace attn_detail {

constexpr int KB = 64;   // keys per tile
constexpr float DEFER_LOG2 = 4.0f;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct AttnArgs {
    const bf16_t* q;
    const bf16_t* k;
    const bf16_t* vT;
    bf16_t* out;
    int Hq, Hkv, L, Lq_rows, Lkv, out_rows_per_batch, ld_out;
    int q_begin;  // first query row (multiple of 32); output row of query r is b*out_rows_per_batch + r - q_begin
    int Lq_alloc; // rows per (batch, head) of q: Lkv, or the compact length of a cache step's queries
    float scale_log2e;
    int xcd_pairs;      // XCD-aware grid: (batch, head) pairs per XCD (0: pair-major grid)
    int groups, chunks; // 16-row query groups per (batch, head) pair and workgroups the pair is cut into
    int plain_order;    // 1: no late waves (attention form 0)
};

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
