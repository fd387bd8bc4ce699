// handle.h — the library's private state behind the opaque mmada_handle (shared by api.hip and tp_comm.hip).
#pragma once
#include <utility>
#include <vector>

#include "../../include/mmada_mi355x.h"
#include "kernels.h"

struct TpComm;  // tp_comm.hip

struct LayerWeights {
    bf16_t* wqkv = nullptr;   // [(Hq_l + 2 Hkv_l) * 128, d]   fused, rotary-partner permuted
    bf16_t* wo = nullptr;     // [d, Hq_l * 128]
    bf16_t* wgu = nullptr;    // [2 F_l, d]                    16-row interleaved ff_proj / up_proj
    bf16_t* wdown = nullptr;  // [d, F_l]
    bf16_t* attn_norm = nullptr;  // [d]
    bf16_t* ff_norm = nullptr;    // [d]
    bool bound = false;
};

// One dLLM-cache slot (the reference's `cat` key: model/modeling_llada.py:593-597,929-940,1406-1413): caller-owned device
// memory holding, for B sequences of length L, every block's keys (rotated, attention layout) and K-major values plus
// the residual stream after the last block; see mmada_cache_bind.
struct CacheSlot {
    char* mem = nullptr;
    size_t bytes = 0, layer_stride = 0, kv_bytes = 0;
    int B = 0, L = 0, Lp = 0, Lkv = 0;
    bool normalized = false;   // the final rows are ALREADY ln_f-normalised (written by a tensor-parallel forward, whose last
                               // exchange applies ln_f on the owners' rows): mmada_cache_head_rows gathers them without a norm
    bf16_t* K(int layer) const { return (bf16_t*)(mem + (size_t)layer * layer_stride); }
    bf16_t* vT(int layer) const { return (bf16_t*)(mem + (size_t)layer * layer_stride + kv_bytes); }
    bf16_t* xfin(int n_layers) const { return (bf16_t*)(mem + (size_t)n_layers * layer_stride); }
};
constexpr int MMADA_CACHE_SLOTS = 16;

struct mmada_handle {
    mmada_cfg cfg;
    int hq_l, hkv_l, f_l;  // per-rank heads / mlp columns
    float* rope_cos = nullptr;
    float* rope_sin = nullptr;
    const bf16_t* wte = nullptr;
    const bf16_t* ln_f = nullptr;
    const bf16_t* lm_head = nullptr;
    std::vector<LayerWeights> layers;
    bool owns_weights = true;  // false for mmada_clone_shared handles
    // workspace
    char* ws = nullptr;
    size_t ws_bytes = 0;
    // current carve
    int B = 0, L = 0, Lp = 0, Lkv = 0, M = 0;
    bf16_t *x = nullptr, *y = nullptr, *xn = nullptr, *att = nullptr, *hbuf = nullptr, *q = nullptr, *k = nullptr,
           *vT = nullptr, *xg = nullptr;
    int32_t* rows_all = nullptr;
    int32_t* posmap = nullptr;  // [B*Lp] sequence position of every compact stream row (compute-mask forward)
    // dLLM cache: slots, and the one a forward in flight writes its keys / values into (cc != null only inside
    // mmada_forward_cached); cc_pos: position map of a compute-mask step (null: every row is computed)
    CacheSlot slots[MMADA_CACHE_SLOTS];
    const CacheSlot* cc = nullptr;
    const int32_t* cc_pos = nullptr;
    int cc_qshift = -1;
    // consumed-row window (mmada_set_consumed_rows): requested [win_beg, win_end) per sequence; while a forward whose
    // last block ran windowed is resident, the stream is compact: cur_W rows per sequence starting at row cur_beg
    int win_beg = 0, win_end = 0;
    int cur_W = 0, cur_beg = 0, Mcur = 0;
    // live timing (mmada_profile_begin/end)
    int prof_layer = -1;
    struct ProfRec { int kind; hipEvent_t a, b; double flops; };
    std::vector<ProfRec> prof;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pool;
    // tensor-parallel collective engine (tp_comm.hip; mmada_comm_*).  xn_is_final: the resident xn already holds
    // ln_f(x) for every row (the last reduce-scatter of a tensor-parallel forward applies ln_f on the owned rows)
    TpComm* tp = nullptr;
    bool xn_is_final = false;
    bool xn_is_layer0 = false;  // mmada_embed already wrote xn = RMSNorm(x) * blocks[0].attn_norm (fused, K1)
};

// tp_comm.hip
int tp_forward_body(mmada_handle* h, hipStream_t s);              // all blocks of a tensor-parallel forward, after mmada_embed
int tp_gather_stream(mmada_handle* h, bf16_t* full_out, hipStream_t s);  // residual stream rows of every owner -> [M, d]
void tp_comm_free(mmada_handle* h);
void tp_allow_single_rank(int on);   // test switch: mmada_comm_create accepts tp_size == 1
bool tp_comm_connected(const mmada_handle* h);   // a transport (or the no-exchange diagnostic) is active on this handle
int tp_head_gather(mmada_handle* h, const int32_t* rows, int R, hipStream_t s);  // xg[r] = xn[row r] (xn already = ln_f(x))
// out[r] = src[b * Lp + l] for rows[r] = b * L + l: the plain row gather behind tp_head_gather, on any [B * Lp, d] buffer
int tp_gather_rows(const bf16_t* src, const int32_t* rows, int R, int L, int Lp, int d, int nflat, bf16_t* out, hipStream_t s);

struct ProfScope {
    mmada_handle* h; hipStream_t s; bool on; hipEvent_t a{}, b{}; int kind; double flops;
    ProfScope(mmada_handle* h_, int layer, int kind_, double flops_, hipStream_t s_)
        : h(h_), s(s_), on(h_->prof_layer == layer), kind(kind_), flops(flops_) {
        if (on) {  // a stream under hipGraph capture records nothing: event timing only exists for eager launches
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) on = false;
        }
        if (!on) return;
        if (h->prof_pool.empty()) {
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
        } else {
            a = h->prof_pool.back().first; b = h->prof_pool.back().second;
            h->prof_pool.pop_back();
        }
        (void)hipEventRecord(a, s);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(b, s);
        h->prof.push_back({kind, a, b, flops});
    }
};

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int ceil_to(int v, int a) { return (v + a - 1) / a * a; }

struct Carve {
    size_t x, y, xn, att, h, q, k, vT, xg, rows, posmap, total;
    int Lp, Lkv, M;
};

static Carve carve_for(const mmada_handle* h, int B, int L) {
    Carve c;
    const int d = h->cfg.d_model;
    c.Lp = ceil_to(L, 8);
    c.Lkv = ceil_to(L, 64);
    c.M = B * c.Lp;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = align_up(off + bytes, 256);
        return o;
    };
    // tensor parallel: a row chunk is split into tp equal owner slices of a multiple of 8 rows; the last chunk's slices
    // may reach past M (equal counts for the RCCL reduce-scatter / all-gather), so the stream buffers carry pad rows
    const size_t mrows = (size_t)c.M + (h->cfg.tp_size > 1 ? 8 * h->cfg.tp_size : 0);
    c.x = take(mrows * d * 2);
    c.y = take(mrows * d * 2);
    c.xn = take(mrows * d * 2);
    c.att = take((size_t)c.M * h->hq_l * 128 * 2);
    c.h = take((size_t)c.M * h->f_l * 2);
    c.q = take((size_t)B * h->hq_l * c.Lkv * 128 * 2);
    c.k = take((size_t)B * h->hkv_l * c.Lkv * 128 * 2);
    c.vT = take((size_t)B * h->hkv_l * 128 * c.Lkv * 2);
    c.xg = take((size_t)B * L * d * 2);
    c.rows = take((size_t)B * L * 4);
    c.posmap = take((size_t)c.M * 4);
    c.total = off;
    return c;
}

