/*
 * CPU ORACLE (test infrastructure — never linked, imported or executed by the product path).
 *
 * Plain-C restatement of the tensor math of generate_ti2ti, integer/tie-break exact.
 * Follows /root/reference/MMaDA-Parallel-A/generators/parallel_generator.py:
 *     oracle_text_select    :181-217  (argmax, float64 softmax confidence, per-row top-k, masked scatter)
 *     oracle_image_probs    :282-295, :311  (dual-CFG combine with bf16 rounding per op, bf16 softmax, argmax)
 *     oracle_image_commit   :221-233, :304-344 and mask_by_random_topk :23-70
 * and /root/reference/MMaDA-Parallel-M/models/modeling_magvitv2.py:186-194,208-221 (oracle_lfq_gather).
 *
 * Parity is PINNED: tests/test_oracle_golden.py checks these functions against fixtures produced by running the
 * reference's own Python (oracle/gen_golden.py -> tests/golden/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; outputs oracle/_build/liboracle.so)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float bf2f(uint16_t b) {
    uint32_t u = ((uint32_t)b) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline uint16_t f2bf(float f) { /* round-to-nearest-even, like torch's .to(bfloat16) */
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bfround(float f) { return bf2f(f2bf(f)); }

/* ---- text step ---------------------------------------------------------------------------------------------- */

typedef struct { double c; int idx; } conf_idx;

static int cmp_desc(const void* a, const void* b) { /* confidence descending, index ascending on ties */
    const conf_idx *x = (const conf_idx*)a, *y = (const conf_idx*)b;
    if (x->c > y->c) return -1;
    if (x->c < y->c) return 1;
    return x->idx - y->idx;
}

/* conf_out [B*T] (may be NULL), x0_out [B*T] (may be NULL) expose the intermediate values for tests. */
void oracle_text_select(const uint16_t* logits, const uint16_t* noisy, int B, int T, int V, int ld, int64_t* ids, int L,
                        int text_start, const int32_t* k, int mask_id, double* conf_out, int32_t* x0_out) {
    conf_idx* ci = (conf_idx*)malloc(sizeof(conf_idx) * (size_t)T);
    int32_t* x0 = (int32_t*)malloc(sizeof(int32_t) * (size_t)T);
    for (int b = 0; b < B; ++b) {
        for (int t = 0; t < T; ++t) {
            const size_t row = (size_t)b * T + t;
            const uint16_t* l = logits + row * ld;
            const uint16_t* a = noisy ? noisy + row * ld : l;
            const int masked = ids[(size_t)b * L + text_start + t] == (int64_t)mask_id;
            /* x0 = argmax(logits_with_noise) (:189): first maximal index */
            int best = 0;
            float bv = bf2f(a[0]);
            for (int i = 1; i < V; ++i) {
                const float v = bf2f(a[i]);
                if (v > bv) { bv = v; best = i; }
            }
            /* p = softmax(text_logits.to(float64)); x0_p = p[x0] (:193-194) */
            double mx = (double)bf2f(l[0]);
            for (int i = 1; i < V; ++i) {
                const double v = (double)bf2f(l[i]);
                if (v > mx) mx = v;
            }
            double sum = 0.0;
            for (int i = 0; i < V; ++i) sum += exp((double)bf2f(l[i]) - mx);
            const double p = exp((double)bf2f(l[best]) - mx) / sum;
            x0[t] = best;
            ci[t].c = masked ? p : -INFINITY; /* confidence = where(mask, x0_p, -inf) (:205) */
            ci[t].idx = t;
            if (conf_out) conf_out[row] = ci[t].c;
            if (x0_out) x0_out[row] = masked ? best : 0;
        }
        /* top-k of the confidences, then scatter x0 at the selected (masked) positions (:207-217) */
        const int kk = k[b];
        if (kk > 0) {
            qsort(ci, (size_t)T, sizeof(conf_idx), cmp_desc);
            for (int j = 0; j < kk && j < T; ++j) {
                if (ci[j].c == -INFINITY) continue; /* x0 = where(mask, x0, ids): unmasked positions keep their id */
                ids[(size_t)b * L + text_start + ci[j].idx] = (int64_t)x0[ci[j].idx];
            }
        }
    }
    free(ci);
    free(x0);
}

/* ---- image step part 1 ---------------------------------------------------------------------------------------- */

void oracle_image_probs(const uint16_t* cond, const uint16_t* ut, const uint16_t* ui, int B, int N, int CB,
                        float cfg_scale, float cfg_img, uint16_t* probs_out, int32_t* argmax_out, uint16_t* pmax_out) {
    float* lg = (float*)malloc(sizeof(float) * (size_t)CB);
    float* e = (float*)malloc(sizeof(float) * (size_t)CB);
    const int use_t = cfg_scale != 0.0f && ut != NULL, use_i = cfg_img != 0.0f && ui != NULL;
    for (size_t row = 0; row < (size_t)B * N; ++row) {
        const uint16_t *c = cond + row * CB, *t = ut ? ut + row * CB : NULL, *im = ui ? ui + row * CB : NULL;
        float mx = -INFINITY;
        for (int i = 0; i < CB; ++i) {
            /* image_logits = cond; += cfg_scale*(cond-ut); += cfg_img*(cond-ui), each op rounded to bf16 (:285-289) */
            const float cv = bf2f(c[i]);
            float l = cv;
            if (use_t) l = bfround(l + bfround(cfg_scale * bfround(cv - bf2f(t[i]))));
            if (use_i) l = bfround(l + bfround(cfg_img * bfround(cv - bf2f(im[i]))));
            lg[i] = l;
            if (l > mx) mx = l;
        }
        /* probs = F.softmax(image_logits) in bf16 (:292): fp32 exp / fp32 sum, output rounded to bf16 */
        double sum = 0.0;
        for (int i = 0; i < CB; ++i) {
            e[i] = (float)exp((double)(lg[i] - mx));
            sum += (double)e[i];
        }
        const float fsum = (float)sum;
        int best = 0;
        float bp = -1.0f;
        for (int i = 0; i < CB; ++i) {
            const uint16_t pb = f2bf(e[i] / fsum);
            if (probs_out) probs_out[row * CB + i] = pb;
            const float p = bf2f(pb);
            if (p > bp) { bp = p; best = i; } /* probs.argmax: first maximal index (:294-295) */
        }
        argmax_out[row] = best;
        pmax_out[row] = f2bf(bp);
    }
    free(lg);
    free(e);
}

/* ---- image step part 2 ---------------------------------------------------------------------------------------- */

typedef struct { float c; int idx; } fconf_idx;

static void stable_sort_asc(fconf_idx* a, fconf_idx* tmp, int n) { /* bottom-up merge sort: stable */
    for (int w = 1; w < n; w *= 2) {
        for (int lo = 0; lo < n; lo += 2 * w) {
            int mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int i = lo, j = mid, o = lo;
            while (i < mid && j < hi) tmp[o++] = (a[j].c < a[i].c) ? a[j++] : a[i++];
            while (i < mid) tmp[o++] = a[i++];
            while (j < hi) tmp[o++] = a[j++];
        }
        memcpy(a, tmp, sizeof(fconf_idx) * (size_t)n);
    }
}

void oracle_image_commit(int64_t* ids, int B, int L, const int32_t* pos_map, int N, const int32_t* sampled_in,
                         const uint16_t* p_in, const uint16_t* noise, float remask_temp, int mask_len_sched, int mask_id,
                         int text_vocab, int codebook) {
    fconf_idx* ci = (fconf_idx*)malloc(sizeof(fconf_idx) * (size_t)N);
    fconf_idx* tmp = (fconf_idx*)malloc(sizeof(fconf_idx) * (size_t)N);
    int32_t* samp = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
    uint8_t* masking = (uint8_t*)malloc((size_t)N);
    for (int b = 0; b < B; ++b) {
        int64_t* row = ids + (size_t)b * L;
        int unknown_count = 0;
        for (int n = 0; n < N; ++n) {
            /* vq list (:221-233): MASK -> -1 (unknown) else clamp(token - text_vocab) */
            const int64_t tok = row[pos_map[n]];
            const int unknown = tok == (int64_t)mask_id;
            int64_t vq = tok - text_vocab;
            if (vq < 0) vq = 0;
            if (vq > codebook - 1) vq = codebook - 1;
            /* sampled = where(unknown, sampled, known); clamp (:305-308) */
            int s = unknown ? sampled_in[(size_t)b * N + n] : (int)vq;
            if (s < 0) s = 0;
            if (s > codebook - 1) s = codebook - 1;
            samp[n] = s;
            /* selected_probs = where(unknown, probs[sampled], finfo(bf16).max) (:311-315) */
            const float p = unknown ? bf2f(p_in[(size_t)b * N + n]) : bf2f(0x7f7f);
            /* confidence = log(probs + 1e-10) + temperature * noise, bf16 tensor ops (:36).  torch rounds the
             * Python scalar of a bf16 `tensor + scalar` to bf16 first (but multiplies by a scalar in fp32) —
             * measured against torch 2.10 CPU, see tests/test_oracle_golden.py::test_log_conf_table */
            float c = bfround((float)log((double)bfround(p + bfround(1e-10f))));
            if (noise) c = bfround(c + bfround(remask_temp * bf2f(noise[(size_t)b * N + n])));
            ci[n].c = c;
            ci[n].idx = n;
            unknown_count += unknown;
        }
        /* mask_len = max(1, min(unknown-1, floor(N*mask_ratio))) (:318-324); clamp(mask_len, 0, N-1) (:43) */
        int k = unknown_count - 1 < mask_len_sched ? unknown_count - 1 : mask_len_sched;
        if (k < 1) k = 1;
        if (k < 0) k = 0;
        if (k > N - 1) k = N - 1;
        /* torch.sort ascending (stable on CPU); the first k sorted indices stay masked (:39, :49-57) */
        stable_sort_asc(ci, tmp, N);
        memset(masking, 0, (size_t)N);
        for (int j = 0; j < k; ++j) masking[ci[j].idx] = 1;
        /* write back (:335-344) */
        for (int n = 0; n < N; ++n) row[pos_map[n]] = masking[n] ? (int64_t)mask_id : (int64_t)(samp[n] + text_vocab);
    }
    free(ci);
    free(tmp);
    free(samp);
    free(masking);
}

/* ==== M variant: MMaDA-Parallel-M/models/modeling_mmada.py:117-248, models/sampling.py:31-36 ================== */

/* text step (:179-207): logits = cond + text_cfg * (uncond - cond) (bf16 per op), then the A text step on it;
 * x0_in != NULL replaces the argmax (float64 Gumbel-max computed by the caller, :181-182). */
void oracle_text_select_cfg(const uint16_t* cond, const uint16_t* unc, float text_cfg, const int32_t* x0_in, int B, int T,
                            int V, int ld, int64_t* ids, int L, int text_start, const int32_t* k, int mask_id,
                            double* conf_out, int32_t* x0_out) {
    uint16_t* comb = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)B * T * ld);
    for (size_t r = 0; r < (size_t)B * T; ++r)
        for (int i = 0; i < V; ++i) {
            const float c = bf2f(cond[r * ld + i]), u = bf2f(unc[r * ld + i]);
            comb[r * ld + i] = f2bf(c + bfround(text_cfg * bfround(u - c)));
        }
    if (!x0_in) {
        oracle_text_select(comb, NULL, B, T, V, ld, ids, L, text_start, k, mask_id, conf_out, x0_out);
    } else {
        /* same as oracle_text_select with x0 given: confidence = softmax_f64(comb)[x0] */
        conf_idx* ci = (conf_idx*)malloc(sizeof(conf_idx) * (size_t)T);
        for (int b = 0; b < B; ++b) {
            for (int t = 0; t < T; ++t) {
                const size_t row = (size_t)b * T + t;
                const uint16_t* l = comb + row * ld;
                const int masked = ids[(size_t)b * L + text_start + t] == (int64_t)mask_id;
                double mx = (double)bf2f(l[0]);
                for (int i = 1; i < V; ++i) if ((double)bf2f(l[i]) > mx) mx = (double)bf2f(l[i]);
                double sum = 0.0;
                for (int i = 0; i < V; ++i) sum += exp((double)bf2f(l[i]) - mx);
                const int x0 = x0_in[row];
                ci[t].c = masked ? exp((double)bf2f(l[x0]) - mx) / sum : -INFINITY;
                ci[t].idx = t;
                if (conf_out) conf_out[row] = ci[t].c;
                if (x0_out) x0_out[row] = masked ? x0 : 0;
            }
            if (k[b] > 0) {
                qsort(ci, (size_t)T, sizeof(conf_idx), cmp_desc);
                for (int j = 0; j < k[b] && j < T; ++j)
                    if (ci[j].c != -INFINITY)
                        ids[(size_t)b * L + text_start + ci[j].idx] = (int64_t)x0_in[(size_t)b * T + ci[j].idx];
            }
        }
        free(ci);
    }
    free(comb);
}

/* image logits (:216): (1 + image_cfg) * cond - image_cfg * uncond, bf16 per op; softmax -> bf16 probs */
void oracle_image_probs_m(const uint16_t* cond, const uint16_t* unc, int B, int N, int CB, float image_cfg,
                          uint16_t* probs_out, int32_t* argmax_out, uint16_t* pmax_out) {
    float* lg = (float*)malloc(sizeof(float) * (size_t)CB);
    float* e = (float*)malloc(sizeof(float) * (size_t)CB);
    const float one_plus = (float)(1.0 + (double)image_cfg);
    for (size_t row = 0; row < (size_t)B * N; ++row) {
        float mx = -INFINITY;
        for (int i = 0; i < CB; ++i) {
            const float l = bfround(bfround(one_plus * bf2f(cond[row * CB + i])) - bfround(image_cfg * bf2f(unc[row * CB + i])));
            lg[i] = l;
            if (l > mx) mx = l;
        }
        double sum = 0.0;
        for (int i = 0; i < CB; ++i) {
            e[i] = (float)exp((double)(lg[i] - mx));
            sum += (double)e[i];
        }
        const float fsum = (float)sum;
        int best = 0;
        float bp = -1.0f;
        for (int i = 0; i < CB; ++i) {
            const uint16_t pb = f2bf(e[i] / fsum);
            if (probs_out) probs_out[row * CB + i] = pb;
            if (bf2f(pb) > bp) { bp = bf2f(pb); best = i; }
        }
        argmax_out[row] = best;
        pmax_out[row] = f2bf(bp);
    }
    free(lg);
    free(e);
}

static int cmp_float_asc(const void* a, const void* b) {
    const float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

/* re-mask + write-back (:224-241; sampling.py:31-36): masking = confidence < sorted(confidence)[mask_len] */
void oracle_image_commit_m(int64_t* ids, int B, int L, const int32_t* pos_map, int N, const int32_t* sampled_in,
                           const uint16_t* p_in, const uint16_t* gumbel, float remask_temp, int mask_len_sched, int mask_id,
                           int text_vocab) {
    float* conf = (float*)malloc(sizeof(float) * (size_t)N);
    float* sorted = (float*)malloc(sizeof(float) * (size_t)N);
    int64_t* samp = (int64_t*)malloc(sizeof(int64_t) * (size_t)N);
    for (int b = 0; b < B; ++b) {
        int64_t* row = ids + (size_t)b * L;
        int unknown_count = 0;
        for (int n = 0; n < N; ++n) {
            const int64_t tok = row[pos_map[n]];
            const int unknown = tok == (int64_t)mask_id;
            samp[n] = unknown ? (int64_t)sampled_in[(size_t)b * N + n] : tok - text_vocab; /* :214,:225 (no clamp) */
            const float p = unknown ? bf2f(p_in[(size_t)b * N + n]) : bf2f(0x7f7f);         /* :233 */
            float c = bfround((float)log((double)(p < 1e-20f ? bfround(1e-20f) : p)));      /* log(t.clamp(min=1e-20)) */
            c = bfround(c + bfround(remask_temp * bf2f(gumbel[(size_t)b * N + n])));
            conf[n] = sorted[n] = c;
            unknown_count += unknown;
        }
        int k = unknown_count - 1 < mask_len_sched ? unknown_count - 1 : mask_len_sched; /* :234-235 */
        if (k < 1) k = 1;
        if (k > N - 1) k = N - 1;
        qsort(sorted, (size_t)N, sizeof(float), cmp_float_asc);
        const float cut = sorted[k];
        for (int n = 0; n < N; ++n) row[pos_map[n]] = conf[n] < cut ? (int64_t)mask_id : samp[n] + text_vocab; /* :239 */
    }
    free(conf);
    free(sorted);
    free(samp);
}

/* log-confidence of a single bf16 probability (exposed for the exhaustive 2^15-value table test) */
uint16_t oracle_log_conf(uint16_t p_bits) {
    return f2bf((float)log((double)bfround(bf2f(p_bits) + bfround(1e-10f))));
}

/* ---- (M) LFQ codebook gather --------------------------------------------------------------------------------- */
/* modeling_magvitv2.py:186-194,208-221: mask = 2^arange(nbits-1,-1,-1); out[b,c,n] = ((idx & mask[c]) != 0)*2 - 1 */
void oracle_lfq_gather(const int64_t* idx, int B, int N, int nbits, float* out) {
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < nbits; ++c)
            for (int n = 0; n < N; ++n)
                out[((size_t)b * nbits + c) * N + n] = ((idx[(size_t)b * N + n] >> (nbits - 1 - c)) & 1) ? 1.0f : -1.0f;
}
