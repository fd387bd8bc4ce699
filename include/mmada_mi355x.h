/*
 * mmada_mi355x.h — C-ABI of libmmada_mi355x.so: the MI355X (gfx950) denoising hot path of the
 * MMaDA-Parallel parallel text+image sampler.
 *
 * The reference (tyfeld/MMaDA-Parallel, 100 % Python) has no FFI layer; the path sits behind two Python call
 * contracts.  Each entry point below names the reference code it replaces (paths relative to
 * /root/reference/MMaDA-Parallel-A unless prefixed).  INTEGRATION.md shows the ctypes stub a reference
 * maintainer would add.
 *
 * Conventions
 *   - every pointer documented "device" is a device pointer owned by the caller (normally a torch tensor's
 *     data_ptr()); the library never frees caller memory.  bf16 data are raw uint16 bit patterns.
 *   - every function returns 0 on success, non-zero on error; mmada_last_error() returns a thread-local message.
 *   - all work is enqueued on the `stream` argument (a hipStream_t passed as void*); no hidden host syncs
 *     (mmada_bind_* and mmada_create allocate and are not graph-capturable; the forward / select calls are).
 *   - one handle per (process, device); a handle is used from one host thread at a time.
 *   - process-global state: kernel attributes (dynamic LDS size) and the per-device constants of the GEMM launchers (zero rows,
 *     SiLU table: mmada_create builds them) sit behind per-device atomic latches that are set only after the call succeeded —
 *     two host threads at worst repeat an idempotent call; the measurement switches of mmada_set_option are process-global
 *     atomics read once per launch.  The supported model stays ONE process per GPU with launches from one host thread per
 *     handle; several handles of one process (the ranks of a test group) may be driven from different threads.
 */
#ifndef MMADA_MI355X_H
#define MMADA_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: only what this header declares is exported. */
#pragma GCC visibility push(default)

typedef struct mmada_handle mmada_handle;

/* Model hyper-parameters: read from the checkpoint's config.json by the caller, never hard-coded
 * (model/configuration_llada.py:129-384 ModelConfig; model/modeling_llada.py:1063-1142 LLaDAModel.__init__). */
typedef struct mmada_cfg {
    int32_t d_model;        /* 4096 */
    int32_t n_layers;       /* 32 */
    int32_t n_heads;        /* 32 */
    int32_t n_kv_heads;     /* 32 (== n_heads unless GQA) */
    int32_t head_dim;       /* must be 128 */
    int32_t mlp_hidden;     /* 12288 (ff_proj / up_proj out features) */
    int32_t vocab;          /* embedding_size = rows of wte / lm head (>= 134548) */
    int32_t max_seq;        /* RoPE table length (max_sequence_length) */
    float   rms_eps;        /* rms_norm_eps */
    float   rope_theta;     /* rope_theta */
    int32_t tp_rank;        /* tensor-parallel rank  (0 when tp_size == 1) */
    int32_t tp_size;        /* tensor-parallel degree (1, 2, 4, 8) */
    int32_t mask_token_id;  /* 126336 (inference.py:22-31) */
    int32_t text_vocab_size;/* 126356 = image-token offset (inference.py:87-89) */
    int32_t codebook_size;  /* 8192 */
    int32_t reserved;
} mmada_cfg;

/* ---- lifetime ------------------------------------------------------------------------------------------------- */

/* Replaces LLaDAModel.__init__ (model/modeling_llada.py:1063-1142) incl. the RotaryEmbedding cache warm-up
 * (:363-400).  `inv_freq_host` (64 floats, host) may be NULL: then inv_freq[i] = theta^(-2i/128) is computed by
 * the library; pass the torch-computed vector to reproduce the reference's fp32 table bit-for-bit. */
int mmada_create(const mmada_cfg* cfg, const float* inv_freq_host, mmada_handle** out);
int mmada_destroy(mmada_handle* h);
/* Second activation context over the SAME bound weights (borrowed from `h`, which must outlive the clone): lets a
 * caller keep two micro-batches in flight, e.g. to overlap one micro-batch's tensor-parallel all-reduce with the
 * other's GEMMs.  The clone needs its own workspace; bind_* on a clone is an error. */
int mmada_clone_shared(mmada_handle* h, mmada_handle** out);
const char* mmada_last_error(void);
/* ABI version of this header (bumped on any signature change). */
int mmada_abi_version(void);

/* ---- weights (state-dict keys of SURVEY.md §5.4; model/modeling_llada.py:1097-1131, 864-893, 564-575) ---------- */

/* model.transformer.wte.weight [vocab,d], model.transformer.ln_f.weight [d], model.transformer.ff_out.weight
 * [vocab,d]; bf16 device pointers, BORROWED (must outlive the handle). */
int mmada_bind_globals(mmada_handle* h, const void* wte, const void* ln_f, const void* lm_head);

/* model.transformer.blocks.{layer}.*: all nine tensors in checkpoint layout (nn.Linear [out,in], bf16, device,
 * UNSHARDED).  The library repacks them into its own MI355X layout (fused+RoPE-permuted QKV, 16-column
 * interleaved gate/up, TP slices by cfg.tp_rank) on `stream`; the caller may free the originals once the
 * stream has drained. */
int mmada_bind_layer(mmada_handle* h, int layer,
                     const void* attn_norm, const void* ff_norm,
                     const void* q_proj, const void* k_proj, const void* v_proj, const void* attn_out,
                     const void* ff_proj, const void* up_proj, const void* ff_out,
                     void* stream);

/* ---- workspace ---------------------------------------------------------------------------------------------- */

/* Bytes of activation workspace needed for a forward of B sequences of length L (all equal length: the
 * reference never masks padding). */
size_t mmada_workspace_bytes(const mmada_handle* h, int B, int L);
/* Caller-owned device buffer (>= mmada_workspace_bytes for the largest (B,L) used), 256-byte aligned. */
int mmada_set_workspace(mmada_handle* h, void* ws, size_t bytes);

/* ---- transformer forward -------------------------------------------------------------------------------------
 * Replaces LLaDAForMultiModalGeneration.forward(infer=True) → LLaDAModelLM.forward → LLaDAModel.forward
 * (model/modeling_xllmx_dimoo.py:41-72, model/modeling_llada.py:1462-1511, 1201-1415) without the dead
 * attention-bias plumbing. */

/* Embedding + all n_layers blocks; leaves the final residual stream resident in the workspace.  tp_size > 1: needs a
 * connected transport (mmada_comm_*), see "tensor-parallel exchange" below. */
int mmada_forward_body(mmada_handle* h, const int64_t* ids /*device [B,L]*/, int B, int L, void* stream);

/* ln_f + LM head on a row subset (model/modeling_llada.py:1392,1399-1404):
 * logits_out[r, :] = ff_out.weight[col_begin:col_end] · ln_f(x[rows[r]]) ; rows[r] = b*L + l (device int32 [R]);
 * logits_out bf16 device [R, col_end-col_begin].  Covers the reference's slices
 * cond_logits[:, text_start:text_end, :] and [:, pos, 126356:134548] (generators/parallel_generator.py:185,236-239,267-274). */
int mmada_head_rows(mmada_handle* h, const int32_t* rows, int R, int col_begin, int col_end,
                    void* logits_out, void* stream);

/* Declare which residual-stream rows the caller will read after the forwards that follow: only l in
 * [row_begin, row_end) of every sequence (e.g. the image + text span of generate_ti2ti; the prompt and the input image
 * are never decoded).  The LAST block then runs attention queries, attn_out and the MLP on those rows only — every
 * earlier block, and the last block's keys/values, still cover the whole sequence, so the consumed rows are
 * bit-identical to a full forward.  mmada_head_rows must then only be given rows inside the window; mmada_read_stream
 * and mmada_forward refuse to run.  row_begin == row_end clears the window.  The setting persists across forwards. */
int mmada_set_consumed_rows(mmada_handle* h, int row_begin, int row_end);

/* ---- dLLM cache: LLaDAModel.forward(input_ids, use_cache=True, to_compute_mask=..., cat=...) ---------------------------
 * (model/modeling_llada.py:1244-1245 token gather, :929-940 per-block k / v cache keyed by `cat`, :714-716,416-428 rotary
 * positions of the computed queries, :1406-1413 logit cache; caching()/empty_cache() :598-600,1417-1426.)
 * A slot is one `cat` key: caller-owned device memory that keeps, for B sequences of length L, every block's keys and
 * values and the residual stream after the last block.  A compute-mask step embeds only the Tc selected tokens of each
 * sequence, runs every block on those rows — their fresh k / v replace the slot's rows at their positions, their queries
 * attend to ALL L cached keys — and replaces their rows of the final stream; everything else is reused.  The reference
 * keeps k / v un-rotated and re-rotates the whole cache every call; the slot keeps them rotated (same values: the rotation
 * of a row depends on its position only).  The reference also keeps a [B, L, vocab] logit cache; a logit row is a
 * function of its final-stream row alone, so mmada_cache_head_rows computes any rows on demand instead. */

/* Bytes of one slot for B sequences of length L. */
size_t mmada_cache_bytes(const mmada_handle* h, int B, int L);
/* Attach `mem` (256-byte aligned, >= mmada_cache_bytes) as slot `slot` in [0,16) and zero it on `stream` — the reference
 * creates a cache with torch.zeros_like (:930-932,1407-1408), so never-computed positions have zero keys, values and
 * logits.  mem == NULL forgets the slot (empty_cache()).  Tensor parallel (round 5): with the library's exchange connected a slot
 * holds THIS rank's heads of every block's keys / values, and the final rows it keeps are already ln_f-normalised (the last
 * exchange of a tensor-parallel forward applies ln_f on the owners' rows and all-gathers them). */
int mmada_cache_bind(mmada_handle* h, int slot, void* mem, size_t bytes, int B, int L, void* stream);
/* One forward through slot `slot`.  pos == NULL: every token is computed (ids [B,L]); keys / values / final stream of the
 * whole sequence are stored (the reference's use_cache=True, to_compute_mask=None call).  pos != NULL: ids [B,Tc] are
 * the tokens at the ascending positions pos [B,Tc] (device int32; the reference's input_ids[to_compute_mask] and
 * to_compute_mask.nonzero()); q_pos_from_map != 0 rotates the queries by their positions (block.use_cache on, i.e.
 * caching(True) was called), 0 reproduces the reference's fallback of positions L-Tc..L-1 (:421-425).  Needs the
 * workspace of (B, L).  Leaves no plain forward resident (mmada_head_rows fails until the next mmada_forward_body). */
int mmada_forward_cached(mmada_handle* h, int slot, const int64_t* ids, const int32_t* pos, int B, int L, int Tc,
                         int q_pos_from_map, void* stream);
/* ln_f + LM head on rows (b*L + l) of the slot's final stream: the rows of logit_cache[cat] (:1409-1413). */
int mmada_cache_head_rows(mmada_handle* h, int slot, const int32_t* rows, int R, int col_begin, int col_end,
                          void* logits_out, void* stream);

/* Drop-in full forward: logits_out bf16 device [B, L, vocab] (generators/parallel_generator.py:178,263,264). */
int mmada_forward(mmada_handle* h, const int64_t* ids, int B, int L, void* logits_out, void* stream);

/* Segment entry points for tensor parallelism (the caller interleaves the RCCL all-reduce of the partial
 * buffer returned by mmada_partial_ptr between them):
 *   mmada_embed → for each layer { mmada_attn_partial, [all-reduce], mmada_mlp_partial, [all-reduce] }.
 * *_partial writes  (own(m) ? x[m] : 0) + local partial of the row-parallel GEMM  into the partial buffer and makes
 * it the new residual stream, where row m's residual is owned by rank (m >> 4) % tp_size (each rank reads 1/tp of
 * the old stream); summing it over ranks yields the reference's x + attn_out(...) / x + ff_out(...). */
int mmada_embed(mmada_handle* h, const int64_t* ids, int B, int L, void* stream);
int mmada_attn_partial(mmada_handle* h, int layer, void* stream);
int mmada_mlp_partial(mmada_handle* h, int layer, void* stream);
/* Device pointer / byte size of the current residual-stream buffer ([B*Lp, d_model] bf16, Lp = L rounded up to 8). */
void* mmada_stream_ptr(mmada_handle* h);
size_t mmada_stream_bytes(const mmada_handle* h);

/* Debug / parity taps: copy the residual stream rows [b*L+l] (pad rows dropped) to out bf16 [B*L, d]. */
int mmada_read_stream(mmada_handle* h, void* out, void* stream);
/* Parity tap on the intermediates of the most recent block: which = 0 xn [B*Lp,d] (last RMSNorm output),
 * 1 q [B,Hq,Lkv,128] (after RoPE), 2 k [B,Hkv,Lkv,128] (after RoPE), 3 vT [B,Hkv,128,Lkv], 4 att [B*Lp,Hq*128]
 * (SDPA output, heads merged), 5 h [B*Lp,F] (silu(ff_proj)*up_proj).  Returns the device pointer and the padded
 * dims {Lp, Lkv} so tests can slice.  The buffers live in the caller's workspace. */
int mmada_debug_buffer(mmada_handle* h, int which, void** ptr_out, int32_t* lp_out, int32_t* lkv_out);

/* ---- sampler math (generators/parallel_generator.py) ---------------------------------------------------------- */

/* Text step, generators/parallel_generator.py:181-217 (low_confidence remasking):
 *   x0 = argmax(logits [+ gumbel]) (first index on ties); p = softmax(float64(logits)); conf = masked ? p[x0] : -inf;
 *   the k[b] highest-confidence masked positions of row b get ids[b, text_start + t] = x0.
 * logits: bf16 device [B,T,ld_logits] (row stride ld_logits elements, V columns used).
 * noisy:  NULL (text_temperature == 0) or bf16 device [B,T,ld_logits] = add_gumbel_noise(logits) (:8-20) — argmax
 *         is taken over it, confidence over `logits`.
 * k:      device int32 [B] (num_transfer_tokens[:, step], :78-99).   ids: device int64 [B,L], updated in place.
 * scratch: device, >= B*T*16 bytes. */
int mmada_text_select(mmada_handle* h, const void* logits, const void* noisy, int B, int T, int V, int ld_logits,
                      int64_t* ids, int L, int text_start, const int32_t* k, void* scratch, void* stream);

/* The same step with remasking='random' (generators/parallel_generator.py:194-198, inference.py --remasking random): the
 * confidence that ranks the masked positions is `uniform` (device fp32 [B,T], the caller's torch.rand draw) instead of the
 * soft-max probability; x0 is still the arg-max of `noisy` / `logits`. */
int mmada_text_select_random(mmada_handle* h, const void* logits, const void* noisy, const float* uniform, int B, int T,
                             int V, int ld_logits, int64_t* ids, int L, int text_start, const int32_t* k, void* scratch,
                             void* stream);

/* Image step part 1, generators/parallel_generator.py:282-295,311: dual-CFG combine with the reference's per-op
 * bf16 rounding, softmax → bf16 probabilities, first-index argmax, probability of the argmax.
 * cond/unc_text/unc_img: bf16 device [B,N,CB] (unc_* may be NULL when the matching scale == 0).
 * probs_out: NULL or bf16 device [B,N,CB] (needed only for temperature > 0, torch.multinomial :297-302).
 * argmax_out: device int32 [B,N];  pmax_out: bf16 device [B,N]. */
int mmada_image_probs(mmada_handle* h, const void* cond, const void* unc_text, const void* unc_img,
                      int B, int N, int CB, float cfg_scale, float cfg_img,
                      void* probs_out, int32_t* argmax_out, void* pmax_out, void* stream);

/* Image step part 2, generators/parallel_generator.py:221-233,304-344 + mask_by_random_topk :23-70:
 *   vq = ids[b, pos_map[n]] (MASK → unknown); sampled = unknown ? sampled_in : clamp(vq - text_vocab);
 *   conf = log(bf16(unknown ? p_in : bf16_max) + 1e-10) [+ temp·noise] (bf16 arithmetic);
 *   mask_len = clamp(max(1, min(unknown_count-1, *mask_len_sched)), 0, N-1); the mask_len lowest-confidence
 *   positions (stable order) are re-masked, all others get ids = sampled + text_vocab.
 * sampled_in: device int32 [B,N]; p_in: bf16 device [B,N]; noise: NULL or bf16 device [B,N] (randn, :30-33);
 * mask_len_sched: device int32 [1] = floor(N·cos(ratio·π/2)) for this step (:318-321);
 * text_vocab_size / codebook_size: the reference's call arguments of the same name (:124-125). */
int mmada_image_commit(mmada_handle* h, int64_t* ids, int B, int L, const int32_t* pos_map, int N,
                       const int32_t* sampled_in, const void* p_in, const void* noise, float remask_temp,
                       const int32_t* mask_len_sched, int text_vocab_size, int codebook_size, void* stream);

/* ---- (M variant) sampler math of MMadaModelLM.interleave_generate, MMaDA-Parallel-M/models/modeling_mmada.py:117-248 --
 * Text step (:179-207): logits = cond + text_cfg * (uncond - cond) with bf16 rounding per op, then exactly
 * mmada_text_select on the combined logits.  x0_in (optional, device int32 [B,T]) supplies an externally sampled x0
 * (the reference's float64 Gumbel-max when text_temperature > 0, :49-60); NULL = first-index argmax. */
int mmada_text_select_cfg(mmada_handle* h, const void* cond, const void* uncond, float text_cfg, const int32_t* x0_in,
                          int B, int T, int V, int ld_logits, int64_t* ids, int L, int text_start, const int32_t* k,
                          void* scratch, void* stream);
/* Image logits (:216): (1 + image_cfg) * cond - image_cfg * uncond (bf16 per op) -> softmax -> bf16 probabilities
 * (probs_out, consumed by torch.multinomial :220-222), first-index argmax and its probability. */
int mmada_image_probs_m(mmada_handle* h, const void* cond, const void* uncond, int B, int N, int CB, float image_cfg,
                        void* probs_out, int32_t* argmax_out, void* pmax_out, void* stream);
/* Re-mask + write-back (:224-241, MMaDA-Parallel-M/models/sampling.py:31-36): confidence = log(clamp(p,1e-20)) +
 * temp * gumbel (bf16 ops; gumbel: bf16 device [B,N], required); masked <=> confidence < sorted[mask_len]
 * (ties with the cut-off are NOT masked); known ids are kept unclamped. */
int mmada_image_commit_m(mmada_handle* h, int64_t* ids, int B, int L, const int32_t* pos_map, int N,
                         const int32_t* sampled_in, const void* p_in, const void* gumbel, float remask_temp,
                         const int32_t* mask_len_sched, int text_vocab_size, void* stream);

/* (A, text-to-image) re-mask + write-back of generate_image (generators/image_generation_generator.py:99-103,178-207,
 * utils/generation_utils.py:47-64): as mmada_image_commit_m, but the number of tokens that stay masked is
 * keep_n[0].clamp(0, unknown-1) with no floor of 1 (keep_n = 0 on the last step masks nothing).  pos_map lists the
 * image slots (positions masked in the prompt); already-known slots keep their ids. */
int mmada_image_commit_g(mmada_handle* h, int64_t* ids, int B, int L, const int32_t* pos_map, int N,
                         const int32_t* sampled_in, const void* p_in, const void* gumbel, float remask_temp,
                         const int32_t* keep_n, int text_vocab_size, void* stream);

/* (M variant) LFQ codebook gather, MMaDA-Parallel-M/models/modeling_magvitv2.py:186-194,208-221:
 * out[b, c, n] = 2·bit_c(idx[b,n]) − 1 as bf16/f32, c in [0,nbits) with bit 0 = most significant
 * (mask = 2^(nbits-1-c)).  idx: device int64 [B,N]; out: device [B,nbits,N] (dtype_f32 ? float : bf16). */
int mmada_lfq_gather(mmada_handle* h, const int64_t* idx, int B, int N, int nbits, int dtype_f32, void* out,
                     void* stream);

/* ---- MAGVITv2 token -> pixel decode of MMaDA-Parallel-M, fp32 like the reference -----------
 * Replaces  vq_model.decode_code(output_image_ids)  (MMaDA-Parallel-M/inference.py:127, models/modeling_mmada.py:840):
 * MAGVITv2.decode_code models/modeling_magvitv2.py:429-433 = LFQuantizer.get_codebook_entry :208-221 +
 * VQGANDecoder.forward :369-406 (ResnetBlock / AttnBlock / Upsample / GroupNorm(32) / swish,
 * models/common_modules.py:337-357,187-211,36-40,16-24).  Independent of mmada_handle (own weights, own workspace). */
typedef struct mmada_vq mmada_vq;
typedef struct mmada_vq_cfg {      /* VQGANDecoder.__init__ arguments, modeling_magvitv2.py:278-287 */
    int32_t ch;                    /* 128; must be a multiple of 128 */
    int32_t n_levels;              /* len(ch_mult) <= 8; the decoder upsamples by 2^(n_levels-1) */
    int32_t ch_mult[8];            /* [1,1,2,2,4]; index 0 = full resolution */
    int32_t num_res_blocks[8];     /* [4,4,3,4,3] */
    int32_t z_channels;            /* 13 = LFQ codebook_dim (codebook size 2^13) */
    int32_t out_ch;                /* 3 */
} mmada_vq_cfg;
int mmada_vq_create(const mmada_vq_cfg* cfg, mmada_vq** out);
void mmada_vq_destroy(mmada_vq* h);
/* Bind one tensor of the decoder's state dict by its checkpoint key ("conv_in.weight", "mid.block_1.norm1.bias",
 * "up.3.block.0.nin_shortcut.weight", "up.2.upsample.conv.weight", ...; an optional "decoder." prefix is accepted).
 * data: device fp32 in the nn.Module layout ([Cout,Cin,k,k] for convolutions); the library keeps its own repacked
 * copy ([Cout][tap][Cin]).  Unknown keys and wrong element counts are errors. */
int mmada_vq_bind(mmada_vq* h, const char* name, const float* data, int64_t numel, void* stream);
int mmada_vq_num_unbound(const mmada_vq* h);   /* 0 when every expected tensor has been bound */
size_t mmada_vq_workspace_bytes(const mmada_vq* h, int B, int hz, int wz);
/* indices: device int64 [B, hz*wz] codebook ids; out: device fp32 [B, out_ch, hz*2^(n_levels-1), wz*2^(n_levels-1)]
 * (NCHW, unclamped, exactly what decode_code returns).  workspace: device, 256-byte aligned,
 * >= mmada_vq_workspace_bytes(h,B,hz,wz).  hz*wz must be a multiple of 32. */
int mmada_vq_decode_code(mmada_vq* h, const int64_t* indices, int B, int hz, int wz, void* workspace,
                         size_t workspace_bytes, float* out, void* stream);
/* Encoder direction: replaces  vq_model.get_code(image)  (MMaDA-Parallel-M/inference.py:79; MAGVITv2.get_code
 * models/modeling_magvitv2.py:422-427 = VQGANEncoder.forward :143-171 + the sign quantisation / get_indices of
 * LFQuantizer :201-206,241-243).  cfg as for the decoder with the encoder's ch_mult / num_res_blocks
 * ([1,2,2,4,4] / [4,3,4,3,4], :63-66) and out_ch = the image channel count (3).  Bind "encoder.*" / bare keys
 * ("conv_in.weight", "down.1.block.0.nin_shortcut.weight", "down.0.downsample.conv.weight", "quant_conv.weight", ...)
 * with mmada_vq_bind; workspace from mmada_vq_workspace_bytes(h, B, H/2^(n_levels-1), W/2^(n_levels-1)). */
int mmada_vq_create_encoder(const mmada_vq_cfg* cfg, mmada_vq** out);
/* pixel_values: device fp32 [B, out_ch, H, W] (NCHW, normalised to [-1,1] by the caller);  indices_out: device int64
 * [B, (H/f)*(W/f)], f = 2^(n_levels-1), bit (z_channels-1-c) of an index = (z_c > 0);  z_out (optional): device fp32
 * [B, (H/f)*(W/f), z_channels], the pre-quantisation encoder output (parity tap). */
int mmada_vq_get_code(mmada_vq* h, const float* pixel_values, int B, int H, int W, void* workspace,
                      size_t workspace_bytes, int64_t* indices_out, float* z_out, void* stream);
/* ---- A variant: diffusers.VQModel (MMaDA-Parallel-A/utils/image_utils.py:13-75 decode_vq_to_image, :159-173
 * encode_img_with_breaks; loaded at inference.py:94-96).  The class itself lives in the third-party package
 * diffusers==0.34.0 (requirements pin), which is NOT vendored under /root/reference: this restates its published
 * architecture (autoencoders/vq_model.py VQModel, autoencoders/vae.py Encoder / Decoder / VectorQuantizer, ResnetBlock2D,
 * UNetMidBlock2D + Attention with one head, Downsample2D(padding=0) / Upsample2D(nearest), GroupNorm eps 1e-6, SiLU) on the
 * kernels of the MAGVITv2 path — parity UNPINNED (no golden vectors can be produced offline).  Handles from
 * mmada_vq_create_vqmodel are used with mmada_vq_bind (checkpoint keys of VQModel: "decoder.up_blocks.0.resnets.1.
 * conv_shortcut.weight", "encoder.down_blocks.0.downsamplers.0.conv.bias", "decoder.mid_block.attentions.0.to_q.weight",
 * "quant_conv.weight", "post_quant_conv.bias", "quantize.embedding.weight", ...), mmada_vq_workspace_bytes,
 * mmada_vq_decode_code (decode(indices, force_not_quantize=True, shape=...) with lookup_from_codebook: codebook rows ->
 * post_quant_conv -> Decoder; hz*wz need not be a multiple of 32 without mid-block attention) and mmada_vq_get_code
 * (encode -> latents, then VectorQuantizer's nearest codebook row: z_out = the latents [B, hz*wz, vq_embed_dim]). */
typedef struct mmada_vqmodel_cfg {   /* VQModel config.json */
    int32_t n_levels;                /* len(block_out_channels) <= 8; scale factor 2^(n_levels-1) */
    int32_t block_out_channels[8];   /* multiples of 128, <= 1024 */
    int32_t layers_per_block;
    int32_t latent_channels;
    int32_t vq_embed_dim;            /* = latent_channels when the config leaves it null */
    int32_t num_vq_embeddings;
    int32_t image_channels;          /* in_channels == out_channels */
    int32_t mid_block_add_attention;
    int32_t norm_num_groups;         /* must be 32 */
} mmada_vqmodel_cfg;
int mmada_vq_create_vqmodel(const mmada_vqmodel_cfg* cfg, int encoder, mmada_vq** out);
/* VectorQuantizer.forward's index: argmin_j cdist(z, embedding)[., j] (first minimum) for n latent rows z [n, vq_embed_dim]
 * (NHWC); either network's handle works once "quantize.embedding.weight" is bound. */
int mmada_vq_nearest_code(mmada_vq* h, const float* z_nhwc, int64_t n, int64_t* indices_out, void* stream);

/* Kernel-level entry points (parity tests).  NHWC fp32; w_packed = weight.permute(0,2,3,1) ([Cout][k*k][Cin]);
 * upsample > 0 folds F.interpolate(scale_factor=2, mode="nearest") in front of the convolution (Upsample.forward),
 * upsample < 0 is Downsample.forward (F.pad(x,(0,1,0,1)) + 3x3 stride 2, common_modules.py:83-90);
 * resid (optional, may alias out) is added after the bias.  out: [B, Ho, Wo, Cout]. */
int mmada_vq_conv2d(const float* in_nhwc, const float* w_packed, const float* bias, const float* resid, float* out,
                    int B, int Hi, int Wi, int Cin, int Cout, int ksize, int upsample, void* stream);
/* GroupNorm(32, C, eps=1e-6, affine) over NHWC [B, HW, C] (+ x*sigmoid(x) when swish != 0); C multiple of 128.
 * scratch: device, >= mmada_vq_group_norm_scratch_bytes(B). */
int mmada_vq_group_norm(const float* x_nhwc, const float* gamma, const float* beta, float* out, void* scratch,
                        int B, int HW, int C, int swish, void* stream);
size_t mmada_vq_group_norm_scratch_bytes(int B);

/* ---- live kernel timing (bench.py's roofline leg) ---------------------------------------------------------------
 * While enabled, the five hot kernels of block `layer` (0 qkv GEMM+RoPE, 1 attention, 2 attn_out GEMM, 3 gate/up
 * GEMM+SiLU·mul, 4 down GEMM) are bracketed by hipEvents on the launch stream.  mmada_profile_end synchronises
 * the events and returns, per kernel class, the launch count, the summed duration (ms) and the summed ALGORITHMIC
 * flops (2·M·N·K with M = B·L real rows; attention 4·B·H·L²·128). */
int mmada_profile_begin(mmada_handle* h, int layer);
int mmada_profile_end(mmada_handle* h, int32_t* count_out /*[5]*/, double* ms_out /*[5]*/, double* flops_out /*[5]*/);

/* ---- measurement / test switches (process-wide; no reference counterpart) --------------------------------------------
 * Every choice below is between kernels that produce BIT-IDENTICAL results (tests/test_gpu_kernels.py); the switches exist
 * so that sweeps and A/B tests can pin one.
 *   "gemm_config"    -1 automatic (default: the cost model of csrc/gemm.hip); 0..3 the 8-phase kernel's tile configuration
 *                    (320x256, 256x256, 160x256, 320x128); 1000 + BM the 16-wave kernel with that row-tile height
 *   "attention_form" -1 automatic (default: MMADA_ATTN_FORM or 1); 0 every wave of a workgroup in the plain order
 *                    { S, soft-max, P·V }; 1 waves 4-7 accumulate P·V one key tile late (bit-identical output)
 *   "gemm_silu_lut"  1 (default): the 8-phase SwiGLU epilogue reads SiLU of the bf16 gate value from a 10-KiB table in the LDS (filled on
 *                    the device by the function it replaces; untabulated values are evaluated); 0: always evaluate
 *   "gemm_short_tiles" 1 (default): the 320-row configurations use a row-tile pitch of 304 when ntm - 1 tiles of 304 rows and
 *                    one of <= 320 cover M (M = B * 2440: 5 % fewer MFMAs in all but the last row tile); 0: full height
 *   "probe_variant"  MFMA shape / occupancy of mmada_mfma_probe (tools/probe_variants.py)
 *   "gemm_tile_order" 0 (default): the 8-phase kernel's per-tile default — groups of 4 row tiles x 8 column tiles for the 320x256
 *                    tile, bands of 1024 columns x all row tiles for the others; GM * 100 + GN: groups of GM row tiles x GN column
 *                    tiles per XCD-round (9904 = the round-4 order; the FETCH_SIZE sweep, tools/tile_order_fetch.py)
 *   "tp_allow_single_rank" 1: mmada_comm_create accepts tp_size == 1 and mmada_forward_body runs a connected one-rank handle through
 *                    the tensor-parallel path (every line of the RCCL / pull transports executes; bit-identical to the plain
 *                    forward: tests/test_gpu_tp.py).  0 (default): tp_size must be 2..8 */
int mmada_set_option(const char* name, int value);
/* The tile configuration the GEMM planner picks for a plain [M, K] x [N, K]^T product (host arithmetic, no launch): 0..3 = the
 * 8-phase configurations in the order above, 1000 + BM = the 16-wave kernel, -1 = unsupported shape.  Honours "gemm_config". */
int mmada_gemm_plan(int M, int N, int K);
/* The attention kernel's launch plan (host arithmetic, no launch): workgroups per (batch, head) pair for `pairs` pairs of `groups`
 * 16-row query groups each against `keys` keys — the smallest count with the least estimated time, rounds (of 256 workgroups) x
 * (largest per-SIMD share x key tiles x 0.405 us + 5.7 us per round); a workgroup holds at most 24 groups.  L = 2438 with 32
 * heads: 153 groups -> 8 workgroups per head, one round at batch 1. */
int mmada_attention_plan(int pairs, int groups, int keys);

/* ---- attainable-MFMA probe (bench.py's roofline.attainable_tflops; measurement only, no reference counterpart) -----
 * Runs `launches` launches of an MFMA-only kernel (v_mfma_f32_16x16x32_bf16 on register-resident operand fragments taken
 * from `data`, eight waves per CU, no memory / LDS / barrier traffic) on `stream`, synchronises, and returns the achieved
 * dense TFLOP/s and the elapsed milliseconds.  `data`: device, >= mmada_mfma_probe_bytes() bytes of bf16 the caller
 * filled with RANDOM values (zeros clock ~20 % higher: MI355X runs this load at its package power limit); `sink`: 4 bytes
 * of device scratch.  iters = 32768 is ~50 ms per launch. */
size_t mmada_mfma_probe_bytes(void);
int mmada_mfma_probe(const void* data, void* sink, int iters, int launches, void* stream, double* tflops_out, double* ms_out);

/* ---- CU-mask probe (measurement only): where the workgroups of a grid run on a stream created with `mask` (words x 32 bits, bit i
 * = "CU i" of hipExtStreamCreateWithCUMask): out_host[b] = XCC_ID | HW_ID << 8 of workgroup b.  tools/cu_mask_probe.py maps mask bits
 * to (XCD, SE, CU) with it — the layout mmada_comm_set_partition relies on. */
int mmada_probe_cu_mask(const uint32_t* mask, int words, uint32_t* out_host, int n_blocks);

/* ---- rounding probe (tests only) -----------------------------------------------------------------------------------------
 * out[i] = the library's fp32 -> bf16 conversion (f2bf of csrc/common.h, used by every epilogue and elementwise kernel) of
 * in[i]; tests/test_gpu_kernels.py sweeps all 2^32 bit patterns against torch's conversion. */
int mmada_probe_f2bf(const float* in, uint16_t* out, int64_t n, void* stream);

/* ---- tensor-parallel exchange inside the library (SURVEY.md §8b mmada_allreduce_init, §8e) ----------------------------
 * New design; the reference runs one replica (inference.py:83-85).  With tp_size > 1 the two row-parallel GEMMs of a block
 * (attn_out, ff_out; model/modeling_llada.py:741-744, 968-970) leave a partial [B*Lp, d] sum on every rank.  One exchange
 * = reduce-scatter -> residual add + RMSNorm on the rows this rank owns -> all-gather of the normalised rows
 * (csrc/tp_comm.hip): the residual stream stays sharded by rows, the all-gathered tensor is the next GEMM's input.
 * Once a transport is connected, mmada_forward_body runs the whole tensor-parallel forward (exchanges on a second,
 * high-priority stream, two row chunks in flight so an exchange runs under the neighbouring GEMM also at batch 1);
 * mmada_head_rows / mmada_read_stream work as at tp_size == 1.  mmada_set_consumed_rows is ignored (every row is kept).
 *
 *   mmada_comm_create        allocate the published buffers for up to max_rows (= B*Lp) stream rows; export_out receives
 *                            mmada_comm_export_bytes() bytes to hand to the other ranks (hipIpc handles), or NULL
 *   mmada_comm_connect_ipc   peers in OTHER processes: exports = [tp_size][export_bytes] in rank order (own slot ignored)
 *   mmada_comm_connect_local peers in THIS process (one host driving several devices with peer access, or tests)
 *   mmada_comm_connect_rccl  RCCL transport (ncclReduceScatter / ncclAllGather issued by the library); unique_id128 from
 *                            mmada_comm_unique_id on rank 0, distributed by the host; librccl_path NULL = "librccl.so"
 *                            (pass the path of the library the process already has loaded, e.g. torch/lib/librccl.so)
 *   mmada_comm_set_mode      switch between connected transports (1 pull, 2 RCCL, 4 copy engines: the mapped peer buffers of
 *                            the pull transport, bytes moved by hipMemcpyAsync / SDMA into local staging, the owner kernel
 *                            reads local memory only, the all-gather half is copies); 3 = DIAGNOSTIC "no exchange": the
 *                            forward runs its owner-side kernels on this rank's own partial sums only (no peer traffic,
 *                            no hand-off; wrong values) — bench.py times it to report the exposed exchange time
 *   mmada_comm_set_partition exchange_cus > 0 (a multiple of 8): the exchange stream owns that many CUs (the same number on
 *                            every XCD, hipExtStreamCreateWithCUMask) and the forward's compute kernels run on a library
 *                            stream masked to the other CUs, forked from / joined to the caller's stream by events — the
 *                            exchange is concurrent with the GEMMs by construction instead of by stream priority; 0 removes
 *                            the partition.  mmada_comm_partition returns the current value, mmada_comm_streams the two
 *                            library streams (probes)
 *   mmada_comm_rccl_nranks   ranks of the RCCL communicator this handle created (ncclCommCount), 0 if none
 *   mmada_comm_status        transport in use, sticky error (a peer never arrived within MMADA_TP_TIMEOUT_S, default 20 s:
 *                            the device is never hung, the results are void), whether the counters are fine-grained
 *   mmada_comm_exchange      ONE exchange over the rows of the resident carve with the caller's partials
 *                            (mmada_comm_part_ptr) — self-tests and bandwidth probes
 * Every rank must issue the same sequence of forwards / exchanges. */
int mmada_comm_export_bytes(void);
int mmada_comm_create(mmada_handle* h, int max_rows, void* export_out);
int mmada_comm_connect_ipc(mmada_handle* h, const void* exports);
int mmada_comm_connect_local(mmada_handle* h, mmada_handle* const* ranks);
int mmada_comm_unique_id(void* out128, const char* librccl_path);
int mmada_comm_connect_rccl(mmada_handle* h, const void* unique_id128, const char* librccl_path);
int mmada_comm_set_mode(mmada_handle* h, int mode);
int mmada_comm_set_partition(mmada_handle* h, int exchange_cus);
int mmada_comm_partition(mmada_handle* h);
int mmada_comm_streams(mmada_handle* h, void** exchange_out, void** compute_out);
int mmada_comm_rccl_nranks(mmada_handle* h);
/* Hand-off timeout of the pull transport in seconds (<= 0: MMADA_TP_TIMEOUT_S or 20 s); clears a sticky error. */
int mmada_comm_set_timeout(mmada_handle* h, double seconds);
int mmada_comm_status(mmada_handle* h, int* mode_out, int* err_out, int* finegrained_out, void* stream);
void* mmada_comm_part_ptr(mmada_handle* h);
int mmada_comm_exchange(mmada_handle* h, const void* norm_w, void* stream);
/* Vocabulary-parallel text step after a tensor-parallel forward (generators/parallel_generator.py:185-217 at
 * text_temperature == 0; LM head model/modeling_llada.py:1399-1404): every rank multiplies the consumed ln_f rows by ITS
 * vocab/tp_size columns of ff_out.weight, reduces each row to {max, first arg-max, fp64 sum-exp} (16 B), the records are
 * exchanged and combined (max of maxima; lowest column wins ties = torch.argmax; rescaled sum), and the k[b] most confident
 * masked positions are committed identically on every rank.  rows: device int32 [B*T] = b*L + text_start + t;
 * scratch: device, >= B*T*16 bytes. */
int mmada_text_select_tp(mmada_handle* h, const int32_t* rows, int B, int T, int64_t* ids, int L, int text_start,
                         const int32_t* k, void* scratch, void* stream);
int mmada_comm_destroy(mmada_handle* h);

/* ---- hipGraph capture of a launch sequence ---------------------------------------------------------------------
 * One denoise step of generate_ti2ti is a fixed sequence of launches over fixed buffers: the per-step text k, the image
 * mask_len and which forwards run are schedule-determined (generators/parallel_generator.py:78-99,157-159,318-327;
 * SURVEY.md A.5), and the reference's ~3 k `.item()` syncs per image step (:223-230,339-344) do not exist here.  Between
 * mmada_graph_begin and mmada_graph_end every call documented "capturable" (forward_body, head_rows, the select /
 * probs / commit kernels, plain device-to-device copies of the caller) issued on `stream` is recorded instead of run;
 * mmada_graph_launch replays the whole step with one host call.  Requirements: a created (non-default) stream; every
 * entry point used inside has run once eagerly before (first calls set kernel attributes and carve the workspace);
 * no allocation and no mmada_profile_begin window inside the capture.  BASELINE.json configs[4]. */
typedef struct mmada_graph mmada_graph;
int mmada_graph_begin(void* stream);
int mmada_graph_end(void* stream, mmada_graph** out);
/* Leave capture mode after a call inside the sequence failed; nothing is kept. */
int mmada_graph_abort(void* stream);
int mmada_graph_launch(mmada_graph* g, void* stream);
/* Kernel / memset / copy nodes in the captured step (== launches the replay saves the host). */
int mmada_graph_num_nodes(const mmada_graph* g);
int mmada_graph_destroy(mmada_graph* g);

/* ---- low-level kernels exposed for parity tests and profiling -------------------------------------------------- */

/* C[M,N] = A[M,K] · W[N,K]^T, bf16 in / fp32 accumulate / bf16 out (F.linear without bias). */
int mmada_gemm_bt(const void* A, const void* W, void* C, int M, int N, int K, void* stream);
/* C[M, N/2] = silu(bf16(A·Wg^T)) * bf16(A·Wu^T) with W in the packed gate/up row order of the library (rows in groups of 32:
 * 16 rows of ff_proj, then the matching 16 rows of up_proj — model/modeling_llada.py:962-968 fused); tests of the SwiGLU epilogue. */
int mmada_gemm_swiglu_bt(const void* A, const void* W, void* C, int M, int N, int K, void* stream);
/* RMSLayerNorm.forward (model/modeling_llada.py:301-329): out = w * bf16(x * rsqrt(mean(x²)+eps)). */
int mmada_rmsnorm(const void* x, const void* w, void* out, int rows, int d, float eps, void* stream);
/* Unmasked non-causal SDPA over [B,H,L,128] q/k/v (bf16, contiguous) → out [B,L,H*128]
 * (model/modeling_llada.py:643-679,731-744). Uses the handle's workspace for the padded q / k and the K-major V copy. */
int mmada_sdpa(mmada_handle* h, const void* q, const void* k, const void* v, void* out, int B, int H, int Hkv, int L,
               void* stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* MMADA_MI355X_H */
