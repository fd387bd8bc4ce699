// api.hip — the C-ABI of include/mmada_mi355x.h: handle, weight repack, workspace carving and the launch sequence
// of one denoiser forward (embedding → n_layers × [RMSNorm → QKV+RoPE GEMM → flash attention → attn_out GEMM +
// residual → RMSNorm → gate/up GEMM + SiLU·mul → down GEMM + residual] → RMSNorm → LM-head rows).
// Host code only; every kernel lives in gemm.hip / attention.hip / elementwise.hip / sampler.hip.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <utility>
#include <vector>

#include "../../include/mmada_mi355x.h"
#include "handle.h"
#include "gemm_epilogue.h"

static thread_local char g_err[1024] = "";

int mm_fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

extern "C" {

int mmada_abi_version(void) { return 1; }
const char* mmada_last_error(void) { return g_err; }

int mmada_create(const mmada_cfg* cfg, const float* inv_freq_host, mmada_handle** out) {
    if (!cfg || !out) return mm_fail("mmada_create: null argument");
    if (cfg->head_dim != 128) return mm_fail("mmada_create: head_dim must be 128 (got %d)", cfg->head_dim);
    if (cfg->d_model != cfg->n_heads * cfg->head_dim) return mm_fail("mmada_create: d_model != n_heads*head_dim");
    if (cfg->n_kv_heads <= 0 || cfg->n_heads % cfg->n_kv_heads) return mm_fail("mmada_create: bad n_kv_heads");
    if (cfg->tp_size < 1 || cfg->tp_rank < 0 || cfg->tp_rank >= cfg->tp_size) return mm_fail("mmada_create: bad tp rank/size");
    if (cfg->n_kv_heads % cfg->tp_size || cfg->n_heads % cfg->tp_size) return mm_fail("mmada_create: heads not divisible by tp_size");
    if (cfg->mlp_hidden % (64 * cfg->tp_size)) return mm_fail("mmada_create: mlp_hidden must be a multiple of 64*tp_size");
    if (cfg->d_model % 64) return mm_fail("mmada_create: d_model must be a multiple of 64");
    if (cfg->max_seq <= 0 || cfg->n_layers <= 0 || cfg->vocab <= 0) return mm_fail("mmada_create: bad sizes");
    if (gemm_prepare_device()) return 1;   // no launch of this handle ever allocates or synchronises (gemm.hip)
    mmada_handle* h = new mmada_handle();
    h->cfg = *cfg;
    h->hq_l = cfg->n_heads / cfg->tp_size;
    h->hkv_l = cfg->n_kv_heads / cfg->tp_size;
    h->f_l = cfg->mlp_hidden / cfg->tp_size;
    h->layers.resize(cfg->n_layers);
    // RoPE tables (model/modeling_llada.py:391-397)
    float inv[64];
    for (int i = 0; i < 64; ++i)
        inv[i] = inv_freq_host ? inv_freq_host[i] : (float)(1.0 / pow((double)cfg->rope_theta, (double)(2 * i) / 128.0));
    float* inv_dev = nullptr;
    MM_CHECK_HIP(hipMalloc(&inv_dev, sizeof(inv)));
    MM_CHECK_HIP(hipMemcpy(inv_dev, inv, sizeof(inv), hipMemcpyHostToDevice));
    MM_CHECK_HIP(hipMalloc(&h->rope_cos, (size_t)cfg->max_seq * 64 * 4));
    MM_CHECK_HIP(hipMalloc(&h->rope_sin, (size_t)cfg->max_seq * 64 * 4));
    if (launch_rope_table(h->rope_cos, h->rope_sin, inv_dev, cfg->max_seq, 0)) return 1;
    MM_CHECK_HIP(hipDeviceSynchronize());
    MM_CHECK_HIP(hipFree(inv_dev));
    *out = h;
    return 0;
}

int mmada_clone_shared(mmada_handle* h, mmada_handle** out) {
    if (!h || !out) return mm_fail("mmada_clone_shared: null argument");
    mmada_handle* c = new mmada_handle();
    c->cfg = h->cfg;
    c->hq_l = h->hq_l; c->hkv_l = h->hkv_l; c->f_l = h->f_l;
    c->rope_cos = h->rope_cos; c->rope_sin = h->rope_sin;
    c->wte = h->wte; c->ln_f = h->ln_f; c->lm_head = h->lm_head;
    c->layers = h->layers;  // pointer copies
    c->owns_weights = false;
    *out = c;
    return 0;
}

int mmada_destroy(mmada_handle* h) {
    if (!h) return 0;
    for (auto& r : h->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto& e : h->prof_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    tp_comm_free(h);
    if (!h->owns_weights) { delete h; return 0; }
    for (auto& lw : h->layers) {
        (void)hipFree(lw.wqkv); (void)hipFree(lw.wo); (void)hipFree(lw.wgu);
        (void)hipFree(lw.wdown); (void)hipFree(lw.attn_norm); (void)hipFree(lw.ff_norm);
    }
    (void)hipFree(h->rope_cos);
    (void)hipFree(h->rope_sin);
    delete h;
    return 0;
}

int mmada_bind_globals(mmada_handle* h, const void* wte, const void* ln_f, const void* lm_head) {
    if (!h || !wte || !ln_f || !lm_head) return mm_fail("mmada_bind_globals: null argument");
    h->wte = (const bf16_t*)wte;
    h->ln_f = (const bf16_t*)ln_f;
    h->lm_head = (const bf16_t*)lm_head;
    return 0;
}

int mmada_bind_layer(mmada_handle* h, int layer, const void* attn_norm, const void* ff_norm, const void* q_proj,
                     const void* k_proj, const void* v_proj, const void* attn_out, const void* ff_proj,
                     const void* up_proj, const void* ff_out, void* stream) {
    if (!h) return mm_fail("mmada_bind_layer: null handle");
    if (!h->owns_weights) return mm_fail("mmada_bind_layer: handle is a shared clone");
    if (layer < 0 || layer >= h->cfg.n_layers) return mm_fail("mmada_bind_layer: layer %d out of range", layer);
    if (!attn_norm || !ff_norm || !q_proj || !k_proj || !v_proj || !attn_out || !ff_proj || !up_proj || !ff_out)
        return mm_fail("mmada_bind_layer: null weight pointer");
    hipStream_t s = (hipStream_t)stream;
    const mmada_cfg& c = h->cfg;
    const int d = c.d_model;
    LayerWeights& lw = h->layers[layer];
    const size_t nqkv = (size_t)(h->hq_l + 2 * h->hkv_l) * 128;
    if (!lw.wqkv) {
        MM_CHECK_HIP(hipMalloc(&lw.wqkv, nqkv * d * 2));
        MM_CHECK_HIP(hipMalloc(&lw.wo, (size_t)d * h->hq_l * 128 * 2));
        MM_CHECK_HIP(hipMalloc(&lw.wgu, (size_t)2 * h->f_l * d * 2));
        MM_CHECK_HIP(hipMalloc(&lw.wdown, (size_t)d * h->f_l * 2));
        MM_CHECK_HIP(hipMalloc(&lw.attn_norm, (size_t)d * 2));
        MM_CHECK_HIP(hipMalloc(&lw.ff_norm, (size_t)d * 2));
    }
    if (launch_pack_qkv((const bf16_t*)q_proj, (const bf16_t*)k_proj, (const bf16_t*)v_proj, lw.wqkv, d, c.n_heads,
                        c.n_kv_heads, c.tp_rank, c.tp_size, s)) return 1;
    if (launch_pack_cols((const bf16_t*)attn_out, lw.wo, d, c.n_heads * 128, c.tp_rank, c.tp_size, s)) return 1;
    if (launch_pack_gate_up((const bf16_t*)ff_proj, (const bf16_t*)up_proj, lw.wgu, d, c.mlp_hidden, c.tp_rank,
                            c.tp_size, s)) return 1;
    if (launch_pack_cols((const bf16_t*)ff_out, lw.wdown, d, c.mlp_hidden, c.tp_rank, c.tp_size, s)) return 1;
    MM_CHECK_HIP(hipMemcpyAsync(lw.attn_norm, attn_norm, (size_t)d * 2, hipMemcpyDeviceToDevice, s));
    MM_CHECK_HIP(hipMemcpyAsync(lw.ff_norm, ff_norm, (size_t)d * 2, hipMemcpyDeviceToDevice, s));
    lw.bound = true;
    return 0;
}

size_t mmada_workspace_bytes(const mmada_handle* h, int B, int L) {
    if (!h || B <= 0 || L <= 0) return 0;
    return carve_for(h, B, L).total;
}

int mmada_set_workspace(mmada_handle* h, void* ws, size_t bytes) {
    if (!h || !ws) return mm_fail("mmada_set_workspace: null argument");
    if (((uintptr_t)ws) & 255) return mm_fail("mmada_set_workspace: workspace must be 256-byte aligned");
    h->ws = (char*)ws;
    h->ws_bytes = bytes;
    h->B = h->L = 0;
    return 0;
}

static int apply_carve(mmada_handle* h, int B, int L, hipStream_t s) {
    if (B <= 0 || L <= 0) return mm_fail("forward: bad shape B=%d L=%d", B, L);
    if (L > h->cfg.max_seq) return mm_fail("forward: L=%d exceeds max_seq=%d", L, h->cfg.max_seq);
    const Carve c = carve_for(h, B, L);
    if (!h->ws || c.total > h->ws_bytes)
        return mm_fail("forward: workspace too small (%zu needed, %zu set)", c.total, h->ws_bytes);
    h->B = B; h->L = L; h->Lp = c.Lp; h->Lkv = c.Lkv; h->M = c.M;
    h->x = (bf16_t*)(h->ws + c.x); h->y = (bf16_t*)(h->ws + c.y); h->xn = (bf16_t*)(h->ws + c.xn);
    h->att = (bf16_t*)(h->ws + c.att); h->hbuf = (bf16_t*)(h->ws + c.h); h->q = (bf16_t*)(h->ws + c.q);
    h->k = (bf16_t*)(h->ws + c.k); h->vT = (bf16_t*)(h->ws + c.vT); h->xg = (bf16_t*)(h->ws + c.xg);
    h->rows_all = (int32_t*)(h->ws + c.rows);
    h->posmap = (int32_t*)(h->ws + c.posmap);
    // vT columns never written by the QKV epilogue (keys >= Lp; the key order inside a 32-key block is permuted, so
    // start at the last block boundary) are multiplied by P == 0: keep them finite
    const int z0 = c.Lp & ~31;
    if (c.Lkv > z0)
        MM_CHECK_HIP(hipMemset2DAsync(h->vT + z0, (size_t)c.Lkv * 2, 0, (size_t)(c.Lkv - z0) * 2,
                                      (size_t)B * h->hkv_l * 128, s));
    return 0;
}

static int check_bound(const mmada_handle* h) {
    if (!h->wte) return mm_fail("forward: mmada_bind_globals was not called");
    for (int i = 0; i < h->cfg.n_layers; ++i)
        if (!h->layers[i].bound) return mm_fail("forward: layer %d not bound", i);
    return 0;
}

int mmada_embed(mmada_handle* h, const int64_t* ids, int B, int L, void* stream) {
    if (!h || !ids) return mm_fail("mmada_embed: null argument");
    hipStream_t s = (hipStream_t)stream;
    if (check_bound(h)) return 1;
    if (apply_carve(h, B, L, s)) return 1;
    h->cur_W = 0; h->cur_beg = 0; h->Mcur = h->M;
    h->xn_is_final = false;
    h->xn_is_layer0 = true;
    return launch_embed(ids, h->wte, h->x, B, L, h->Lp, h->cfg.d_model, h->cfg.vocab, s, h->layers[0].attn_norm, h->xn,
                        h->cfg.rms_eps);
}

int mmada_attn_partial(mmada_handle* h, int layer, void* stream) {
    if (!h || h->M == 0) return mm_fail("mmada_attn_partial: call mmada_embed first");
    if (layer < 0 || layer >= h->cfg.n_layers) return mm_fail("mmada_attn_partial: bad layer");
    hipStream_t s = (hipStream_t)stream;
    const LayerWeights& lw = h->layers[layer];
    const int d = h->cfg.d_model;
    if (layer == 0 && h->xn_is_layer0) {
        h->xn_is_layer0 = false;  // the embedding kernel normalised its rows already
    } else if (launch_rmsnorm(h->x, lw.attn_norm, h->xn, h->M, d, h->cfg.rms_eps, s)) return 1;
    GemmArgs g{};
    g.A = h->xn; g.W = lw.wqkv; g.C = nullptr;
    g.M = h->M; g.N = (h->hq_l + 2 * h->hkv_l) * 128; g.K = d;
    g.lda = d; g.ldw = d; g.ldc = 0;
    g.q = h->q; g.k = h->k; g.vT = h->vT; g.rope_cos = h->rope_cos; g.rope_sin = h->rope_sin;
    g.Lp = h->Lp; g.Lkv = h->Lkv; g.Hq = h->hq_l; g.Hkv = h->hkv_l;
    const CacheSlot* cc = h->cc;  // dLLM cache step: this block's keys / values live in (and are written to) the slot
    if (cc) {
        g.k = cc->K(layer); g.vT = cc->vT(layer); g.Lkv = cc->Lkv;
        g.pos_map = h->cc_pos; g.Lq = h->Lkv; g.q_pos_shift = h->cc_qshift;
    }
    const double rows = (double)h->B * h->L;
    {
        ProfScope p(h, layer, 0, 2.0 * rows * g.N * g.K, s);
        if (launch_gemm(EPI_QKV, g, s)) return 1;
    }
    // last block + consumed-row window: only the rows the caller will read are attended / projected (bit-identical on
    // them: the window start is rounded down to the 32-query wave granule, so every wave sees the queries it saw before)
    // The window END is rounded up to a multiple of 8 rows (inside the Lp-padded stream): the compact panel then meets the
    // shape contract of the 8-phase GEMM (whole 8-row LDS-DMA pieces); the up to 7 extra rows are computed like any other.
    int wbeg = 0, wend = 0, W = 0;
    if (!cc && layer == h->cfg.n_layers - 1 && h->win_end > h->win_beg) {
        if (h->win_end > h->L) return mm_fail("forward: consumed rows [%d,%d) exceed L=%d", h->win_beg, h->win_end, h->L);
        wbeg = h->win_beg & ~31;
        wend = std::min((h->win_end + 7) & ~7, h->Lp);
        W = wend - wbeg;
        if (W >= h->Lp) { wbeg = 0; W = 0; }  // nothing to skip
    }
    const int Mo = W ? h->B * W : h->M;
    const double orows = W ? (double)h->B * W : rows;
    {
        ProfScope p(h, layer, 1, 4.0 * h->hq_l * orows * (cc ? cc->L : h->L) * 128.0, s);
        if (cc) {  // compact (or all) queries of this call against the slot's keys / values of the whole sequence
            if (launch_attention(h->q, g.k, g.vT, h->att, h->B, h->hq_l, h->hkv_l, cc->L, h->Lp, cc->Lkv, h->Lp,
                                 h->hq_l * 128, s, 0, h->Lkv)) return 1;
        } else if (W) {
            if (launch_attention(h->q, h->k, h->vT, h->att, h->B, h->hq_l, h->hkv_l, h->L, wend, h->Lkv, W,
                                 h->hq_l * 128, s, wbeg)) return 1;
        } else if (launch_attention(h->q, h->k, h->vT, h->att, h->B, h->hq_l, h->hkv_l, h->L, h->Lp, h->Lkv, h->Lp,
                                    h->hq_l * 128, s)) return 1;
    }
    GemmArgs o{};
    o.A = h->att; o.W = lw.wo; o.C = h->y;
    o.M = Mo; o.N = d; o.K = h->hq_l * 128;
    o.lda = o.K; o.ldw = o.K; o.ldc = d;
    o.resid = h->x; o.ldr = d; o.resid_mod = h->cfg.tp_size; o.resid_rank = h->cfg.tp_rank;
    if (W) { o.rwin = W; o.rlp = h->Lp; o.rbeg = wbeg; }
    {
        ProfScope p(h, layer, 2, 2.0 * orows * o.N * o.K, s);
        if (launch_gemm(EPI_RESID, o, s)) return 1;
    }
    std::swap(h->x, h->y);
    if (W) { h->cur_W = W; h->cur_beg = wbeg; h->Mcur = Mo; }
    return 0;
}

int mmada_mlp_partial(mmada_handle* h, int layer, void* stream) {
    if (!h || h->M == 0) return mm_fail("mmada_mlp_partial: call mmada_embed first");
    if (layer < 0 || layer >= h->cfg.n_layers) return mm_fail("mmada_mlp_partial: bad layer");
    hipStream_t s = (hipStream_t)stream;
    const LayerWeights& lw = h->layers[layer];
    const int d = h->cfg.d_model;
    if (launch_rmsnorm(h->x, lw.ff_norm, h->xn, h->Mcur, d, h->cfg.rms_eps, s)) return 1;
    GemmArgs g{};
    g.A = h->xn; g.W = lw.wgu; g.C = h->hbuf;
    g.M = h->Mcur; g.N = 2 * h->f_l; g.K = d;
    g.lda = d; g.ldw = d; g.ldc = h->f_l;
    const double rows = h->cur_W ? (double)h->Mcur : (double)h->B * h->L;
    {
        ProfScope p(h, layer, 3, 2.0 * rows * g.N * g.K, s);
        if (launch_gemm(EPI_SWIGLU, g, s)) return 1;
    }
    GemmArgs o{};
    o.A = h->hbuf; o.W = lw.wdown; o.C = h->y;
    o.M = h->Mcur; o.N = d; o.K = h->f_l;
    o.lda = h->f_l; o.ldw = h->f_l; o.ldc = d;
    o.resid = h->x; o.ldr = d; o.resid_mod = h->cfg.tp_size; o.resid_rank = h->cfg.tp_rank;
    {
        ProfScope p(h, layer, 4, 2.0 * rows * o.N * o.K, s);
        if (launch_gemm(EPI_RESID, o, s)) return 1;
    }
    std::swap(h->x, h->y);
    return 0;
}

int mmada_profile_begin(mmada_handle* h, int layer) {
    if (!h) return mm_fail("mmada_profile_begin: null handle");
    for (auto& r : h->prof) h->prof_pool.push_back({r.a, r.b});
    h->prof.clear();
    h->prof_layer = layer;
    return 0;
}

int mmada_profile_end(mmada_handle* h, int32_t* count_out, double* ms_out, double* flops_out) {
    if (!h || !count_out || !ms_out || !flops_out) return mm_fail("mmada_profile_end: null argument");
    for (int i = 0; i < 5; ++i) { count_out[i] = 0; ms_out[i] = 0.0; flops_out[i] = 0.0; }
    for (auto& r : h->prof) {
        MM_CHECK_HIP(hipEventSynchronize(r.b));
        float ms = 0.f;
        MM_CHECK_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        count_out[r.kind] += 1; ms_out[r.kind] += ms; flops_out[r.kind] += r.flops;
        h->prof_pool.push_back({r.a, r.b});
    }
    h->prof.clear();
    h->prof_layer = -1;
    return 0;
}

int mmada_set_option(const char* name, int value) {
    if (!name) return mm_fail("mmada_set_option: null name");
    if (!strcmp(name, "gemm_config")) { gemm_force_config(value); return 0; }
    if (!strcmp(name, "gemm_silu_lut")) { gemm_set_silu_lut(value); return 0; }
    if (!strcmp(name, "gemm_short_tiles")) { gemm8_set_short_tiles(value); return 0; }
    if (!strcmp(name, "gemm_tile_order")) { gemm8_set_tile_order(value); return 0; }
    if (!strcmp(name, "attention_form")) { attention_force_form(value); return 0; }
    if (!strcmp(name, "probe_variant")) { mfma_probe_set_variant(value); return 0; }
    if (!strcmp(name, "tp_allow_single_rank")) { tp_allow_single_rank(value); return 0; }
    return mm_fail("mmada_set_option: unknown option '%s'", name);
}

int mmada_gemm_swiglu_bt(const void* A, const void* W, void* C, int M, int N, int K, void* stream) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0 || N % 64) return mm_fail("mmada_gemm_swiglu_bt: bad argument");
    GemmArgs g{};
    g.A = (const bf16_t*)A; g.W = (const bf16_t*)W; g.C = (bf16_t*)C;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N / 2;
    return launch_gemm(EPI_SWIGLU, g, (hipStream_t)stream);
}

int mmada_gemm_plan(int M, int N, int K) { return gemm_plan_code(M, N, K); }
int mmada_attention_plan(int pairs, int groups, int keys) {
    if (pairs <= 0 || groups <= 0 || keys <= 0) return mm_fail("mmada_attention_plan: bad argument"), -1;
    return attention_chunks(pairs, groups, keys);
}

int mmada_probe_f2bf(const float* in, uint16_t* out, int64_t n, void* stream) {
    if (!in || !out || n < 0) return mm_fail("mmada_probe_f2bf: bad argument");
    return launch_f2bf_probe(in, out, (long long)n, (hipStream_t)stream);
}

size_t mmada_mfma_probe_bytes(void) { return (size_t)64 * 8 * 16 * 64 * 16; }

int mmada_mfma_probe(const void* data, void* sink, int iters, int launches, void* stream, double* tflops_out, double* ms_out) {
    if (!data || !sink || iters <= 0 || launches <= 0) return mm_fail("mmada_mfma_probe: bad argument");
    return launch_mfma_probe((const bf16_t*)data, (float*)sink, iters, launches, (hipStream_t)stream, tflops_out, ms_out);
}

void* mmada_stream_ptr(mmada_handle* h) { return h ? (void*)h->x : nullptr; }
size_t mmada_stream_bytes(const mmada_handle* h) { return h ? (size_t)h->Mcur * h->cfg.d_model * 2 : 0; }

int mmada_forward_body(mmada_handle* h, const int64_t* ids, int B, int L, void* stream) {
    if (!h) return mm_fail("mmada_forward_body: null handle");
    if (h->cfg.tp_size != 1 || tp_comm_connected(h)) {   // (a connected one-rank group: the tp_allow_single_rank test switch)
        // tensor parallel: the exchange step lives in the library (tp_comm.hip); the residual stream stays sharded by rows
        if (!h->tp)
            return mm_fail("mmada_forward_body: tp_size=%d needs a connected collective (mmada_comm_create + "
                           "mmada_comm_connect_*) or the host-issued all-reduce of the segment API", h->cfg.tp_size);
        if (mmada_embed(h, ids, B, L, stream)) return 1;
        return tp_forward_body(h, (hipStream_t)stream);
    }
    if (mmada_embed(h, ids, B, L, stream)) return 1;
    for (int i = 0; i < h->cfg.n_layers; ++i) {
        if (mmada_attn_partial(h, i, stream)) return 1;
        if (mmada_mlp_partial(h, i, stream)) return 1;
    }
    return 0;
}

// ---- dLLM cache (model/modeling_llada.py:593-600,929-940,1244-1245,1406-1426) -----------------------------------------
static void slot_layout(const mmada_handle* h, int B, int L, CacheSlot& c) {
    c.B = B; c.L = L; c.Lp = ceil_to(L, 8); c.Lkv = ceil_to(L, 64);
    c.kv_bytes = align_up((size_t)B * h->hkv_l * c.Lkv * 128 * 2, 256);
    c.layer_stride = 2 * c.kv_bytes;
    c.bytes = (size_t)h->cfg.n_layers * c.layer_stride + align_up((size_t)B * c.Lp * h->cfg.d_model * 2, 256);
}

size_t mmada_cache_bytes(const mmada_handle* h, int B, int L) {
    if (!h || B <= 0 || L <= 0) return 0;
    CacheSlot c;
    slot_layout(h, B, L, c);
    return c.bytes;
}

int mmada_cache_bind(mmada_handle* h, int slot, void* mem, size_t bytes, int B, int L, void* stream) {
    if (!h) return mm_fail("mmada_cache_bind: null handle");
    if (slot < 0 || slot >= MMADA_CACHE_SLOTS) return mm_fail("mmada_cache_bind: slot %d outside [0,%d)", slot, MMADA_CACHE_SLOTS);
    if (!mem) {  // release
        h->slots[slot] = CacheSlot{};
        return 0;
    }
    if (h->cfg.tp_size != 1 && !tp_comm_connected(h))
        return mm_fail("mmada_cache_bind: tp_size=%d needs the library's exchange connected (mmada_comm_create + mmada_comm_connect_*)", h->cfg.tp_size);
    if (B <= 0 || L <= 0 || L > h->cfg.max_seq) return mm_fail("mmada_cache_bind: bad shape B=%d L=%d", B, L);
    if (((uintptr_t)mem) & 255) return mm_fail("mmada_cache_bind: memory must be 256-byte aligned");
    CacheSlot c;
    slot_layout(h, B, L, c);
    if (bytes < c.bytes) return mm_fail("mmada_cache_bind: %zu bytes given, %zu needed", bytes, c.bytes);
    c.mem = (char*)mem;
    // the reference starts a cache at zeros (torch.zeros_like, :930-932,1407-1408): a never-computed position has zero
    // keys / values (a zero score, a zero value row) and zero logits (ln_f(0) = 0)
    MM_CHECK_HIP(hipMemsetAsync(mem, 0, c.bytes, (hipStream_t)stream));
    h->slots[slot] = c;
    return 0;
}

int mmada_forward_cached(mmada_handle* h, int slot, const int64_t* ids, const int32_t* pos, int B, int L, int Tc,
                         int q_pos_from_map, void* stream) {
    if (!h || !ids) return mm_fail("mmada_forward_cached: null argument");
    if (slot < 0 || slot >= MMADA_CACHE_SLOTS || !h->slots[slot].mem) return mm_fail("mmada_forward_cached: slot %d is not bound", slot);
    const CacheSlot& c = h->slots[slot];
    if (c.B != B || c.L != L) return mm_fail("mmada_forward_cached: slot holds B=%d L=%d, call has B=%d L=%d", c.B, c.L, B, L);
    const bool tp = h->cfg.tp_size != 1 || tp_comm_connected(h);
    if (tp && !tp_comm_connected(h)) return mm_fail("mmada_forward_cached: tp_size=%d needs the library's exchange connected", h->cfg.tp_size);
    if (!pos) Tc = L;
    if (Tc <= 0 || Tc > L) return mm_fail("mmada_forward_cached: Tc=%d outside (0,%d]", Tc, L);
    hipStream_t s = (hipStream_t)stream;
    if (check_bound(h)) return 1;
    // buffers are carved for the whole (B, L) shape — mmada_cache_head_rows may ask for any row — and the blocks then run on
    // the compact [B, ceil8(Tc)] stream of the computed tokens
    if (apply_carve(h, B, L, s)) return 1;
    if (pos) {
        h->L = Tc; h->Lp = ceil_to(Tc, 8); h->Lkv = ceil_to(Tc, 64); h->M = B * h->Lp;
        if (launch_expand_pos(pos, h->posmap, B, Tc, h->Lp, L, s)) { h->M = 0; return 1; }
    }
    h->cur_W = 0; h->cur_beg = 0; h->Mcur = h->M;
    h->xn_is_final = false;
    h->xn_is_layer0 = true;
    const int d = h->cfg.d_model;
    if (launch_embed(ids, h->wte, h->x, B, h->L, h->Lp, d, h->cfg.vocab, s, h->layers[0].attn_norm, h->xn, h->cfg.rms_eps)) {
        h->M = 0;
        return 1;
    }
    h->cc = &c;
    h->cc_pos = pos ? h->posmap : nullptr;
    h->cc_qshift = (pos && !q_pos_from_map) ? L - Tc : -1;
    int rc = 0;
    if (tp) {
        // tensor parallel: the blocks, their exchanges and the cache hooks of this rank's heads run in tp_forward_body; its last
        // exchange leaves xn = ln_f(x) on EVERY row of every rank, which is what the slot keeps (CacheSlot::normalized)
        rc = tp_forward_body(h, s);
    } else {
        for (int i = 0; i < h->cfg.n_layers && !rc; ++i) {
            rc = mmada_attn_partial(h, i, stream);
            if (!rc) rc = mmada_mlp_partial(h, i, stream);
        }
    }
    h->cc = nullptr; h->cc_pos = nullptr; h->cc_qshift = -1;
    if (rc) { h->M = 0; return 1; }
    // the rows just computed replace theirs in the slot's final residual stream (the reference scatters the logits,
    // :1409-1411; a logit row is a function of its residual row alone, so the head runs on demand: mmada_cache_head_rows)
    bf16_t* xfin = c.xfin(h->cfg.n_layers);
    const bf16_t* fin = tp ? h->xn : h->x;
    h->slots[slot].normalized = tp;
    if (pos) {
        if (launch_scatter_rows(fin, xfin, h->posmap, h->M, h->Lp, c.Lp, d, s)) { h->M = 0; return 1; }
    } else {
        MM_CHECK_HIP(hipMemcpyAsync(xfin, fin, (size_t)h->M * d * 2, hipMemcpyDeviceToDevice, s));
    }
    h->xn_is_final = false;
    h->M = 0;  // no plain forward is resident: mmada_head_rows / mmada_read_stream must not read the compact stream
    return 0;
}

int mmada_cache_head_rows(mmada_handle* h, int slot, const int32_t* rows, int R, int col_begin, int col_end,
                          void* logits_out, void* stream) {
    if (!h || !rows || !logits_out) return mm_fail("mmada_cache_head_rows: null argument");
    if (slot < 0 || slot >= MMADA_CACHE_SLOTS || !h->slots[slot].mem) return mm_fail("mmada_cache_head_rows: slot %d is not bound", slot);
    const CacheSlot& c = h->slots[slot];
    if (R <= 0) return 0;
    if (R > c.B * c.L) return mm_fail("mmada_cache_head_rows: R=%d exceeds B*L=%d", R, c.B * c.L);
    if (col_begin < 0 || col_end > h->cfg.vocab || col_begin >= col_end) return mm_fail("mmada_cache_head_rows: bad column range");
    hipStream_t s = (hipStream_t)stream;
    const int d = h->cfg.d_model;
    // staging rows for ln_f: the gather buffer of the workspace.  While a plain forward is resident its carve is live, so
    // its own gather buffer (B*L rows of that forward) is the only region that may be written; otherwise the carve of the
    // slot's shape applies
    bf16_t* xg;
    if (h->M != 0) {
        if ((size_t)R > (size_t)h->B * h->L)
            return mm_fail("mmada_cache_head_rows: %d rows do not fit the resident forward's gather buffer (%d x %d rows); "
                           "ask for fewer rows per call", R, h->B, h->L);
        xg = h->xg;
    } else {
        const Carve cv = carve_for(h, c.B, c.L);
        if (!h->ws || cv.total > h->ws_bytes) return mm_fail("mmada_cache_head_rows: workspace too small (%zu needed)", cv.total);
        xg = (bf16_t*)(h->ws + cv.xg);
    }
    if (c.normalized) {   // rows written by a tensor-parallel forward: ln_f already applied by the owners
        if (tp_gather_rows(c.xfin(h->cfg.n_layers), rows, R, c.L, c.Lp, d, c.B * c.L, xg, s)) return 1;
    } else if (launch_rmsnorm_gather(c.xfin(h->cfg.n_layers), h->ln_f, xg, rows, R, c.L, c.Lp, d, h->cfg.rms_eps, s, 0, c.B * c.L))
        return 1;
    GemmArgs g{};
    g.A = xg; g.W = h->lm_head + (size_t)col_begin * d; g.C = (bf16_t*)logits_out;
    g.M = R; g.N = col_end - col_begin; g.K = d;
    g.lda = d; g.ldw = d; g.ldc = g.N;
    return launch_gemm(EPI_STORE, g, s);
}

int mmada_head_rows(mmada_handle* h, const int32_t* rows, int R, int col_begin, int col_end, void* logits_out,
                    void* stream) {
    if (!h || h->M == 0) return mm_fail("mmada_head_rows: no forward resident");
    if (!rows || !logits_out) return mm_fail("mmada_head_rows: null argument");
    if (R <= 0) return 0;
    if (R > h->B * h->L) return mm_fail("mmada_head_rows: R=%d exceeds B*L=%d", R, h->B * h->L);
    if (col_begin < 0 || col_end > h->cfg.vocab || col_begin >= col_end) return mm_fail("mmada_head_rows: bad column range");
    hipStream_t s = (hipStream_t)stream;
    const int d = h->cfg.d_model;
    // a windowed forward left the stream compact: row (b, l) sits at b*cur_W + l - cur_beg; rows outside the window the
    // caller declared with mmada_set_consumed_rows were never computed and must not be requested
    if (h->xn_is_final) {  // tensor-parallel forward: the last exchange already applied ln_f on the owners' rows
        if (tp_head_gather(h, rows, R, s)) return 1;
    } else if (launch_rmsnorm_gather(h->x, h->ln_f, h->xg, rows, R, h->L, h->cur_W ? h->cur_W : h->Lp, d, h->cfg.rms_eps, s,
                                     h->cur_beg, h->B * h->L)) return 1;
    GemmArgs g{};
    g.A = h->xg; g.W = h->lm_head + (size_t)col_begin * d; g.C = (bf16_t*)logits_out;
    g.M = R; g.N = col_end - col_begin; g.K = d;
    g.lda = d; g.ldw = d; g.ldc = g.N;
    return launch_gemm(EPI_STORE, g, s);
}

int mmada_set_consumed_rows(mmada_handle* h, int row_begin, int row_end) {
    if (!h) return mm_fail("mmada_set_consumed_rows: null handle");
    if (row_begin < 0 || row_end < row_begin) return mm_fail("mmada_set_consumed_rows: bad range [%d,%d)", row_begin, row_end);
    h->win_beg = row_begin;
    h->win_end = row_end;  // row_begin == row_end: no window (every row is computed)
    return 0;
}

int mmada_forward(mmada_handle* h, const int64_t* ids, int B, int L, void* logits_out, void* stream) {
    if (h && h->win_end > h->win_beg) return mm_fail("mmada_forward: returns every row; clear mmada_set_consumed_rows first");
    if (mmada_forward_body(h, ids, B, L, stream)) return 1;
    if (launch_iota_rows(h->rows_all, B * L, (hipStream_t)stream)) return 1;
    return mmada_head_rows(h, h->rows_all, B * L, 0, h->cfg.vocab, logits_out, stream);
}

int mmada_read_stream(mmada_handle* h, void* out, void* stream) {
    if (!h || h->M == 0 || !out) return mm_fail("mmada_read_stream: no forward resident");
    if (h->cur_W) return mm_fail("mmada_read_stream: the resident stream only holds rows [%d,%d) of each sequence", h->cur_beg, h->cur_beg + h->cur_W);
    if (h->xn_is_final) {  // rows of the residual stream live on their owners: collect them (parity tap only)
        if (tp_gather_stream(h, h->y, (hipStream_t)stream)) return 1;
        return launch_unpad_rows(h->y, (bf16_t*)out, h->B, h->L, h->Lp, h->cfg.d_model, (hipStream_t)stream);
    }
    return launch_unpad_rows(h->x, (bf16_t*)out, h->B, h->L, h->Lp, h->cfg.d_model, (hipStream_t)stream);
}

int mmada_debug_buffer(mmada_handle* h, int which, void** ptr_out, int32_t* lp_out, int32_t* lkv_out) {
    if (!h || h->M == 0 || !ptr_out) return mm_fail("mmada_debug_buffer: no forward resident");
    bf16_t* tab[6] = {h->xn, h->q, h->k, h->vT, h->att, h->hbuf};
    if (which < 0 || which > 5) return mm_fail("mmada_debug_buffer: which=%d", which);
    *ptr_out = tab[which];
    if (lp_out) *lp_out = h->Lp;
    if (lkv_out) *lkv_out = h->Lkv;
    return 0;
}

int mmada_text_select(mmada_handle* h, const void* logits, const void* noisy, int B, int T, int V, int ld_logits,
                      int64_t* ids, int L, int text_start, const int32_t* k, void* scratch, void* stream) {
    if (!h || !logits || !ids || !k || !scratch) return mm_fail("mmada_text_select: null argument");
    if (text_start < 0 || text_start + T > L) return mm_fail("mmada_text_select: text span outside the sequence");
    return launch_text_select((const bf16_t*)logits, (const bf16_t*)noisy, nullptr, 0.f, nullptr, B, T, V, ld_logits, ids, L,
                              text_start, k, scratch, h->cfg.mask_token_id, (hipStream_t)stream);
}

int mmada_text_select_random(mmada_handle* h, const void* logits, const void* noisy, const float* uniform, int B, int T,
                             int V, int ld_logits, int64_t* ids, int L, int text_start, const int32_t* k, void* scratch,
                             void* stream) {
    if (!h || !logits || !uniform || !ids || !k || !scratch) return mm_fail("mmada_text_select_random: null argument");
    if (text_start < 0 || text_start + T > L) return mm_fail("mmada_text_select_random: text span outside the sequence");
    return launch_text_select((const bf16_t*)logits, (const bf16_t*)noisy, nullptr, 0.f, nullptr, B, T, V, ld_logits, ids, L,
                              text_start, k, scratch, h->cfg.mask_token_id, (hipStream_t)stream, uniform);
}

int mmada_text_select_cfg(mmada_handle* h, const void* cond, const void* uncond, float text_cfg, const int32_t* x0_in,
                          int B, int T, int V, int ld_logits, int64_t* ids, int L, int text_start, const int32_t* k,
                          void* scratch, void* stream) {
    if (!h || !cond || !uncond || !ids || !k || !scratch) return mm_fail("mmada_text_select_cfg: null argument");
    if (text_start < 0 || text_start + T > L) return mm_fail("mmada_text_select_cfg: text span outside the sequence");
    return launch_text_select((const bf16_t*)cond, nullptr, (const bf16_t*)uncond, text_cfg, x0_in, B, T, V, ld_logits, ids,
                              L, text_start, k, scratch, h->cfg.mask_token_id, (hipStream_t)stream);
}

int mmada_image_probs_m(mmada_handle* h, const void* cond, const void* uncond, int B, int N, int CB, float image_cfg,
                        void* probs_out, int32_t* argmax_out, void* pmax_out, void* stream) {
    if (!h || !cond || !uncond || !argmax_out || !pmax_out) return mm_fail("mmada_image_probs_m: null argument");
    const float one_plus = (float)(1.0 + (double)image_cfg);
    return launch_image_probs((const bf16_t*)cond, (const bf16_t*)uncond, nullptr, B, N, CB, image_cfg, one_plus,
                              (bf16_t*)probs_out, argmax_out, (bf16_t*)pmax_out, 1, (hipStream_t)stream);
}

int mmada_image_commit_m(mmada_handle* h, int64_t* ids, int B, int L, const int32_t* pos_map, int N,
                         const int32_t* sampled_in, const void* p_in, const void* gumbel, float remask_temp,
                         const int32_t* mask_len_sched, int text_vocab_size, void* stream) {
    if (!h || !ids || !pos_map || !sampled_in || !p_in || !gumbel || !mask_len_sched)
        return mm_fail("mmada_image_commit_m: null argument");
    return launch_image_commit(ids, B, L, pos_map, N, sampled_in, (const bf16_t*)p_in, (const bf16_t*)gumbel, remask_temp,
                               mask_len_sched, h->cfg.mask_token_id, text_vocab_size, 0, 1, (hipStream_t)stream);
}

int mmada_image_commit_g(mmada_handle* h, int64_t* ids, int B, int L, const int32_t* pos_map, int N,
                         const int32_t* sampled_in, const void* p_in, const void* gumbel, float remask_temp,
                         const int32_t* keep_n, int text_vocab_size, void* stream) {
    if (!h || !ids || !pos_map || !sampled_in || !p_in || !gumbel || !keep_n)
        return mm_fail("mmada_image_commit_g: null argument");
    return launch_image_commit(ids, B, L, pos_map, N, sampled_in, (const bf16_t*)p_in, (const bf16_t*)gumbel, remask_temp,
                               keep_n, h->cfg.mask_token_id, text_vocab_size, 0, 2, (hipStream_t)stream);
}

int mmada_image_probs(mmada_handle* h, const void* cond, const void* unc_text, const void* unc_img, int B, int N, int CB,
                      float cfg_scale, float cfg_img, void* probs_out, int32_t* argmax_out, void* pmax_out,
                      void* stream) {
    if (!h || !cond || !argmax_out || !pmax_out) return mm_fail("mmada_image_probs: null argument");
    return launch_image_probs((const bf16_t*)cond, (const bf16_t*)unc_text, (const bf16_t*)unc_img, B, N, CB, cfg_scale,
                              cfg_img, (bf16_t*)probs_out, argmax_out, (bf16_t*)pmax_out, 0, (hipStream_t)stream);
}

int mmada_image_commit(mmada_handle* h, int64_t* ids, int B, int L, const int32_t* pos_map, int N,
                       const int32_t* sampled_in, const void* p_in, const void* noise, float remask_temp,
                       const int32_t* mask_len_sched, int text_vocab_size, int codebook_size, void* stream) {
    if (!h || !ids || !pos_map || !sampled_in || !p_in || !mask_len_sched) return mm_fail("mmada_image_commit: null argument");
    return launch_image_commit(ids, B, L, pos_map, N, sampled_in, (const bf16_t*)p_in, (const bf16_t*)noise, remask_temp,
                               mask_len_sched, h->cfg.mask_token_id, text_vocab_size, codebook_size, 0, (hipStream_t)stream);
}

int mmada_lfq_gather(mmada_handle* h, const int64_t* idx, int B, int N, int nbits, int dtype_f32, void* out,
                     void* stream) {
    (void)h;
    if (!idx || !out) return mm_fail("mmada_lfq_gather: null argument");
    if (nbits <= 0 || nbits > 62) return mm_fail("mmada_lfq_gather: bad nbits");
    return launch_lfq_gather(idx, out, B, N, nbits, dtype_f32, (hipStream_t)stream);
}

int mmada_gemm_bt(const void* A, const void* W, void* C, int M, int N, int K, void* stream) {
    if (!A || !W || !C) return mm_fail("mmada_gemm_bt: null argument");
    GemmArgs g{};
    g.A = (const bf16_t*)A; g.W = (const bf16_t*)W; g.C = (bf16_t*)C;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N;
    return launch_gemm(EPI_STORE, g, (hipStream_t)stream);
}

int mmada_rmsnorm(const void* x, const void* w, void* out, int rows, int d, float eps, void* stream) {
    if (!x || !w || !out) return mm_fail("mmada_rmsnorm: null argument");
    return launch_rmsnorm((const bf16_t*)x, (const bf16_t*)w, (bf16_t*)out, rows, d, eps, (hipStream_t)stream);
}

int mmada_sdpa(mmada_handle* h, const void* q, const void* k, const void* v, void* out, int B, int H, int Hkv, int L,
               void* stream) {
    if (!h || !q || !k || !v || !out) return mm_fail("mmada_sdpa: null argument");
    hipStream_t s = (hipStream_t)stream;
    const int Lkv = ceil_to(L, 64);
    const size_t qb = align_up((size_t)B * H * Lkv * 128 * 2, 256), kb = align_up((size_t)B * Hkv * Lkv * 128 * 2, 256);
    if (!h->ws || qb + 2 * kb > h->ws_bytes) return mm_fail("mmada_sdpa: workspace too small (%zu needed)", qb + 2 * kb);
    bf16_t* qp = (bf16_t*)h->ws;
    bf16_t* kp = (bf16_t*)(h->ws + qb);
    bf16_t* vt = (bf16_t*)(h->ws + qb + kb);
    h->M = 0;  // the resident forward (if any) is clobbered
    if (launch_pad_heads((const bf16_t*)q, qp, B * H, L, Lkv, s)) return 1;
    if (launch_pad_heads((const bf16_t*)k, kp, B * Hkv, L, Lkv, s)) return 1;
    if (launch_transpose_v((const bf16_t*)v, vt, B * Hkv, L, Lkv, s)) return 1;
    return launch_attention(qp, kp, vt, (bf16_t*)out, B, H, Hkv, L, L, Lkv, L, H * 128, s);
}

}  // extern "C"
