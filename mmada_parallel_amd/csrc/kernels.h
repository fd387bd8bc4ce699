// kernels.h — host-side launcher declarations shared between the .hip translation units and api.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

enum GemmEpilogue { EPI_STORE = 0, EPI_RESID = 1, EPI_SWIGLU = 2, EPI_QKV = 3 };

struct GemmArgs {
    const bf16_t* A;   // [M, lda]   activations, K-contiguous
    const bf16_t* W;   // [N, ldw]   nn.Linear weight layout (out, in), K-contiguous
    bf16_t* C;         // [M, ldc]   (EPI_STORE / EPI_RESID / EPI_SWIGLU: ldc = N/2)
    int M, N, K;
    int lda, ldw, ldc;
    // EPI_RESID: C = bf16(resid + bf16(acc)) on the rows this rank owns the residual of, else bf16(acc).
    // Row m's residual is added by rank (m >> 4) % resid_mod, so that under tensor parallelism every rank reads
    // 1/tp of the residual stream instead of rank 0 reading all of it (resid_mod = 1: always add).
    const bf16_t* resid;
    int ldr;
    int resid_mod, resid_rank;
    // Row map of the residual read (0 = identity): C/A rows are compact [b*rwin + i], the residual is read from the
    // full-layout row b*rlp + rbeg + i (last block of a forward whose consumer only reads a row window).
    int rwin, rlp, rbeg;
    // >= K zeros: source of the A rows of the last row tile that lie beyond M (set by launch_gemm; null = re-read row M-1)
    const bf16_t* zero_row;
    // EPI_QKV: rows are (b, l) with m = b*Lp + l; columns are [q heads | k heads | v heads] x 128
    bf16_t* q;         // [B, Hq , Lkv, 128]
    bf16_t* k;         // [B, Hkv, Lkv, 128]
    bf16_t* vT;        // [B, Hkv, 128, Lkv]
    const float* rope_cos;  // [max_seq, 64]
    const float* rope_sin;  // [max_seq, 64]
    int Lp, Lkv, Hq, Hkv;
    int publish;       // != 0: C is read by OTHER agents / other XCDs' kernels polling a counter (tensor-parallel partials):
                       // every workgroup ends with a system-scope release so its stores have left this XCD's L2
    int m_base;        // EPI_QKV on a row chunk: A/M describe rows [m_base, m_base+M) of the [B*Lp] stream (b, l from m_base+m)
    // EPI_QKV with a position map (compute-mask forward of the dLLM cache, model/modeling_llada.py:929-937): stream row
    // m = b*Lp + i is the token at sequence position pos_map[m] (< 0: pad row).  q goes to the COMPACT row i of
    // q [B, Hq, Lq, 128]; k / v are scattered to row pos of the cache-resident k / vT (stride Lkv); nothing of a pad row is
    // kept.  Rotary position: k always pos; q pos, or i + q_pos_shift when q_pos_shift >= 0 (the reference's rotary
    // q_mask quirk with caching() off, :714-716,416-428).
    const int32_t* pos_map;
    int Lq, q_pos_shift;
    // EPI_SWIGLU in the 8-phase kernel: SiLU of a bf16 value as a table (gemm_epilogue.h: SiluLut) in device memory, or null =
    // evaluate it (set by launch_gemm)
    const uint16_t* silu_lut;
    int row_drop;      // set by gemm8's launcher only: row tiles of pitch BM - 16 (gemm8.hip, "short row tiles")
    int tile_gm, tile_gn;  // set by gemm8's launcher only: tile-order groups of tile_gm row tiles (0: all) x tile_gn column tiles
};

int launch_gemm(int epi, const GemmArgs& g, hipStream_t s);

// elementwise.hip
// norm_w != null: also xn = RMSNorm(x) * norm_w (the first norm of the forward, fused: SURVEY §2.3 K1)
int launch_embed(const int64_t* ids, const bf16_t* wte, bf16_t* x, int B, int L, int Lp, int d, int vocab, hipStream_t s,
                 const bf16_t* norm_w = nullptr, bf16_t* xn = nullptr, float eps = 0.f);
int launch_rmsnorm(const bf16_t* x, const bf16_t* w, bf16_t* out, int rows, int d, float eps, hipStream_t s);
// gathered rmsnorm: out[r] = rmsnorm(x[map(rows[r])]) where rows[r] = b*L + l and x rows are b*Lp + l
// x row of the flat index b*L + l is b*row_stride + l - row_off (row_stride = Lp, row_off = 0 for the full layout)
int launch_rmsnorm_gather(const bf16_t* x, const bf16_t* w, bf16_t* out, const int32_t* rows, int R, int L, int row_stride,
                          int d, float eps, hipStream_t s, int row_off = 0, int nflat = 0);
int launch_rope_table(float* cos_t, float* sin_t, const float* inv_freq_dev, int max_seq, hipStream_t s);
int launch_unpad_rows(const bf16_t* x, bf16_t* out, int B, int L, int Lp, int d, hipStream_t s);
int launch_iota_rows(int32_t* rows, int n, hipStream_t s);
// dLLM cache helpers: posmap[b*Lp + i] = pos[b*Tc + i] (i < Tc) or -1; dst[b*Lp_dst + pos] = src[b*Lp + i] for mapped rows
int launch_expand_pos(const int32_t* pos, int32_t* posmap, int B, int Tc, int Lp, int L, hipStream_t s);
int launch_scatter_rows(const bf16_t* src, bf16_t* dst, const int32_t* posmap, int M, int Lp, int Lp_dst, int d, hipStream_t s);
// weight repack
int launch_pack_qkv(const bf16_t* wq, const bf16_t* wk, const bf16_t* wv, bf16_t* out, int d, int Hq, int Hkv, int tp_rank,
                    int tp_size, hipStream_t s);
int launch_pack_gate_up(const bf16_t* gate, const bf16_t* up, bf16_t* out, int d, int F, int tp_rank, int tp_size,
                        hipStream_t s);
int launch_pack_cols(const bf16_t* w, bf16_t* out, int rows, int cols, int tp_rank, int tp_size, hipStream_t s);
// [B,H,L,128] -> padded q/k layout or K-major V copy (for the standalone mmada_sdpa entry point)
int launch_pad_heads(const bf16_t* in, bf16_t* out, int BH, int L, int Lkv, hipStream_t s);
int launch_transpose_v(const bf16_t* v, bf16_t* vT, int BH, int L, int Lkv, hipStream_t s);
int launch_lfq_gather(const int64_t* idx, void* out, int B, int N, int nbits, int f32, hipStream_t s);

// attention.hip:  q [B,Hq,Lkv,128], k [B,Hkv,Lkv,128], vT [B,Hkv,128,Lkv] -> out rows (b*Lp_out + l) x (Hq*128)
// Lq_alloc > 0: q is [B,Hq,Lq_alloc,128] (compact queries against longer cached keys); 0: q shares the keys' Lkv
int launch_attention(const bf16_t* q, const bf16_t* k, const bf16_t* vT, bf16_t* out, int B, int Hq, int Hkv, int L,
                     int Lq_rows, int Lkv, int out_row_stride_per_batch, int ld_out, hipStream_t s, int q_begin = 0,
                     int Lq_alloc = 0);

int attention_chunks(int pairs, int groups, int keys);  // workgroups per (batch, head) pair (the launch plan; host arithmetic)
void attention_force_form(int form);  // 0: every wave in the plain order, 1: late waves (default); tests compare the two

// sampler.hip
struct TextStat { float lmax; int32_t arg; double sum; };  // one rank's record of a text row (vocabulary-parallel head)
int launch_text_stats_partial(const bf16_t* logits, int B, int T, int Vl, int ld, int col_off, const int64_t* ids, int L,
                              int text_start, int mask_id, TextStat* out, hipStream_t s);
int launch_text_commit(const void* scratch, int B, int T, int64_t* ids, int L, int text_start, const int32_t* k, hipStream_t s);
int launch_text_select(const bf16_t* logits, const bf16_t* noisy, const bf16_t* unc, float text_cfg, const int32_t* x0_in,
                       int B, int T, int V, int ld, int64_t* ids, int L, int text_start, const int32_t* k, void* scratch,
                       int mask_id, hipStream_t s, const float* rand_conf = nullptr);
int launch_image_probs(const bf16_t* cond, const bf16_t* ut, const bf16_t* ui, int B, int N, int CB, float cfg_scale,
                       float cfg_img, bf16_t* probs_out, int32_t* argmax_out, bf16_t* pmax_out, int mvar, hipStream_t s);
int launch_image_commit(int64_t* ids, int B, int L, const int32_t* pos_map, int N, const int32_t* sampled_in,
                        const bf16_t* p_in, const bf16_t* noise, float remask_temp, const int32_t* mask_len_sched,
                        int mask_id, int text_vocab, int codebook, int mvar, hipStream_t s);

// probe.hip: MFMA-only diagnostic (attainable roof at the sustained clock on random data)
void mfma_probe_set_variant(int v);
int launch_f2bf_probe(const float* in, bf16_t* out, long long n, hipStream_t s);
int launch_mfma_probe(const bf16_t* data, float* sink, int iters, int launches, hipStream_t s, double* tflops_out, double* ms_out);
