// sampler.hip — the tensor math of generate_ti2ti (generators/parallel_generator.py) as fused row kernels.
//
// These kernels are HBM-bound (one pass over [rows, vocab] bf16 logits) and integer/tie-break exact:
//   * argmax resolves ties to the LOWEST index (torch.argmax on CPU);
//   * the re-mask selection is a STABLE ascending order (torch.sort(stable) semantics observed on CPU);
//   * every bf16 rounding point of the reference is reproduced (A.7): CFG combine per op, bf16 probabilities,
//     bf16 log-confidence.
// Transcendentals are evaluated in fp64 and rounded once to the precision the reference computes in, which makes
// the device result independent of libm flavour (the CPU oracle in oracle/sampler_oracle.c does the same).
#include "kernels.h"

namespace {

constexpr int TB = 256;

// ---------------------------------------------------------------------------------------------------------------
// Text step part 1: per (b,t) row — x0 = argmax(noisy or logits), conf = softmax_f64(logits)[x0]
// (generators/parallel_generator.py:185-205).  Rows whose token is not MASK are skipped (conf = -inf).
// ---------------------------------------------------------------------------------------------------------------
// M variant (MMaDA-Parallel-M/models/modeling_mmada.py:168-199): the text logits are first combined with the
// unconditional branch, logits = cond + text_cfg * (uncond - cond), each op rounded to bf16; `unc` == nullptr keeps
// the A behaviour.  `x0_in` (optional) supplies an externally sampled x0 (float64 Gumbel-max, text_temperature > 0).
MM_DEVICE float text_cfg_combine(float c, float u, float s) { return bfround(c + bfround(s * bfround(u - c))); }

template <bool CFG>
__global__ __launch_bounds__(TB) void text_row_stats_kernel(const bf16_t* __restrict__ logits,
                                                            const bf16_t* __restrict__ noisy,
                                                            const bf16_t* __restrict__ unc, float text_cfg,
                                                            const int32_t* __restrict__ x0_in, int T, int V, int ld,
                                                            const int64_t* __restrict__ ids, int L, int text_start,
                                                            int mask_id, double* __restrict__ conf_out,
                                                            int32_t* __restrict__ x0_out) {
    __shared__ float s_val[TB / 64];
    __shared__ int s_idx[TB / 64];
    __shared__ float s_max[TB / 64];
    __shared__ double s_sum[TB / 64];
    const int row = blockIdx.x;
    const int b = row / T, t = row - b * T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (ids[(size_t)b * L + text_start + t] != (int64_t)mask_id) {
        if (tid == 0) {
            conf_out[row] = -INFINITY;
            x0_out[row] = 0;
        }
        return;
    }
    const bf16_t* lrow = logits + (size_t)row * ld;
    const bf16_t* arow = noisy ? noisy + (size_t)row * ld : lrow;
    const bf16_t* urow = CFG ? unc + (size_t)row * ld : nullptr;
    const int nchunk = V >> 3;
    // value of logit i of the (possibly CFG-combined) row
    auto lval = [&](uint32_t cbits, uint32_t ubits) -> float {
        const float c = __uint_as_float(cbits);
        if constexpr (CFG) return text_cfg_combine(c, __uint_as_float(ubits), text_cfg);
        return c;
    };

    // pass 1: first-index argmax of arow, max of lrow
    float best = -INFINITY, lmax = -INFINITY;
    int bidx = 0x7fffffff;
    for (int c = tid; c < nchunk; c += TB) {
        const u32x4 lv = ((const u32x4*)lrow)[c];
        u32x4 uv = lv;
        if constexpr (CFG) uv = ((const u32x4*)urow)[c];
        u32x4 av = lv;
        if (noisy) av = ((const u32x4*)arow)[c];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float l0 = lval(lv[j] << 16, uv[j] << 16), l1 = lval(lv[j] & 0xffff0000u, uv[j] & 0xffff0000u);
            lmax = fmaxf(lmax, fmaxf(l0, l1));
            const float lo = noisy ? __uint_as_float(av[j] << 16) : l0;
            const float hi = noisy ? __uint_as_float(av[j] & 0xffff0000u) : l1;
            if (lo > best) { best = lo; bidx = c * 8 + 2 * j; }
            if (hi > best) { best = hi; bidx = c * 8 + 2 * j + 1; }
        }
    }
    for (int i = (nchunk << 3) + tid; i < V; i += TB) {  // tail (V % 8)
        const float l = lval((uint32_t)lrow[i] << 16, CFG ? (uint32_t)urow[i] << 16 : 0u);
        lmax = fmaxf(lmax, l);
        const float a = noisy ? bf2f(arow[i]) : l;
        if (a > best || (a == best && i < bidx)) { best = a; bidx = i; }
    }
    // wave reduce (value desc, index asc)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    lmax = wave_max(lmax);
    if (lane == 0) { s_val[wave] = best; s_idx[wave] = bidx; s_max[wave] = lmax; }
    __syncthreads();
    best = s_val[0]; bidx = s_idx[0]; lmax = s_max[0];
#pragma unroll
    for (int w = 1; w < TB / 64; ++w) {
        if (s_val[w] > best || (s_val[w] == best && s_idx[w] < bidx)) { best = s_val[w]; bidx = s_idx[w]; }
        lmax = fmaxf(lmax, s_max[w]);
    }
    if (bidx == 0x7fffffff) bidx = 0;  // all -inf / NaN row
    if (x0_in) bidx = x0_in[row];

    // pass 2: sum exp(l - max) in fp64 (F.softmax(text_logits.to(torch.float64)), :193)
    const double dmax = (double)lmax;
    double sum = 0.0;
    for (int c = tid; c < nchunk; c += TB) {
        const u32x4 lv = ((const u32x4*)lrow)[c];
        u32x4 uv = lv;
        if constexpr (CFG) uv = ((const u32x4*)urow)[c];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sum += exp((double)lval(lv[j] << 16, uv[j] << 16) - dmax);
            sum += exp((double)lval(lv[j] & 0xffff0000u, uv[j] & 0xffff0000u) - dmax);
        }
    }
    for (int i = (nchunk << 3) + tid; i < V; i += TB)
        sum += exp((double)lval((uint32_t)lrow[i] << 16, CFG ? (uint32_t)urow[i] << 16 : 0u) - dmax);
    sum = wave_sum_d(sum);
    if (lane == 0) s_sum[wave] = sum;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int w = 0; w < TB / 64; ++w) tot += s_sum[w];
        const float lsel = lval((uint32_t)lrow[bidx] << 16, CFG ? (uint32_t)urow[bidx] << 16 : 0u);
        conf_out[row] = exp((double)lsel - dmax) / tot;
        x0_out[row] = bidx;
    }
}

// Vocabulary-parallel form of part 1 (tensor parallelism, text_temperature == 0): this rank holds the logits of columns
// [col_off, col_off + Vl) only.  Per masked row it publishes {local max, global index of its first local maximum,
// sum_j exp(l_j - local max) in fp64}; csrc/tp_comm.hip combines the tp records (max of maxima, the lowest rank holding it
// = the lowest column index = torch.argmax's first index, rescaled sum) into the same conf / x0 the one-rank kernel writes.
__global__ __launch_bounds__(TB) void text_row_stats_partial_kernel(const bf16_t* __restrict__ logits, int T, int Vl, int ld,
                                                                    int col_off, const int64_t* __restrict__ ids, int L,
                                                                    int text_start, int mask_id,
                                                                    TextStat* __restrict__ out) {
    __shared__ float s_val[TB / 64];
    __shared__ int s_idx[TB / 64];
    __shared__ double s_sum[TB / 64];
    const int row = blockIdx.x;
    const int b = row / T, t = row - b * T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (ids[(size_t)b * L + text_start + t] != (int64_t)mask_id) {
        if (tid == 0) out[row] = TextStat{-INFINITY, 0, 0.0};
        return;
    }
    const bf16_t* lrow = logits + (size_t)row * ld;
    const int nchunk = Vl >> 3;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    for (int c = tid; c < nchunk; c += TB) {
        const u32x4 lv = ((const u32x4*)lrow)[c];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = __uint_as_float(lv[j] << 16), hi = __uint_as_float(lv[j] & 0xffff0000u);
            if (lo > best) { best = lo; bidx = c * 8 + 2 * j; }
            if (hi > best) { best = hi; bidx = c * 8 + 2 * j + 1; }
        }
    }
    for (int i = (nchunk << 3) + tid; i < Vl; i += TB) {
        const float a = bf2f(lrow[i]);
        if (a > best || (a == best && i < bidx)) { best = a; bidx = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    if (lane == 0) { s_val[wave] = best; s_idx[wave] = bidx; }
    __syncthreads();
    best = s_val[0]; bidx = s_idx[0];
#pragma unroll
    for (int w = 1; w < TB / 64; ++w)
        if (s_val[w] > best || (s_val[w] == best && s_idx[w] < bidx)) { best = s_val[w]; bidx = s_idx[w]; }
    if (bidx == 0x7fffffff) bidx = 0;
    const double dmax = (double)best;
    double sum = 0.0;
    for (int c = tid; c < nchunk; c += TB) {
        const u32x4 lv = ((const u32x4*)lrow)[c];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sum += exp((double)__uint_as_float(lv[j] << 16) - dmax);
            sum += exp((double)__uint_as_float(lv[j] & 0xffff0000u) - dmax);
        }
    }
    for (int i = (nchunk << 3) + tid; i < Vl; i += TB) sum += exp((double)bf2f(lrow[i]) - dmax);
    sum = wave_sum_d(sum);
    if (lane == 0) s_sum[wave] = sum;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int w = 0; w < TB / 64; ++w) tot += s_sum[w];
        out[row] = TextStat{best, col_off + bidx, tot};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // published: read by the other ranks (csrc/tp_comm.hip)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// Text step part 2: per batch row, unmask the k highest-confidence masked positions (:207-217).
__global__ __launch_bounds__(TB) void text_commit_kernel(const double* __restrict__ conf, const int32_t* __restrict__ x0,
                                                         int T, int64_t* __restrict__ ids, int L, int text_start,
                                                         const int32_t* __restrict__ kptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sc = (double*)smem;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int k = kptr[b];
    if (k <= 0) return;
    for (int t = tid; t < T; t += TB) sc[t] = conf[(size_t)b * T + t];
    __syncthreads();
    for (int t = tid; t < T; t += TB) {
        const double c = sc[t];
        if (!(c > -INFINITY)) continue;  // only masked positions carry a finite confidence
        int rank = 0;
        for (int j = 0; j < T; ++j) {
            const double cj = sc[j];
            rank += (cj > c) || (cj == c && j < t);
        }
        if (rank < k) ids[(size_t)b * L + text_start + t] = (int64_t)x0[(size_t)b * T + t];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Image step part 1: dual-CFG combine + softmax(bf16) + first-index argmax, one workgroup per (b,n) row
// (generators/parallel_generator.py:282-295, 311).
// ---------------------------------------------------------------------------------------------------------------
MM_DEVICE float cfg_combine(float c, float ut, float ui, bool use_t, bool use_i, float st, float si) {
    // image_logits = c; += cfg_scale*(c - ut); += cfg_img*(c - ui) — every op rounds to bf16 (:285-289)
    float l = c;
    if (use_t) l = bfround(l + bfround(st * bfround(c - ut)));
    if (use_i) l = bfround(l + bfround(si * bfround(c - ui)));
    return l;
}

// M variant (modeling_mmada.py:216): image_logits = (1 + image_cfg) * cond - image_cfg * uncond, bf16 per op.
MM_DEVICE float cfg_combine_m(float c, float u, float one_plus, float s) {
    return bfround(bfround(one_plus * c) - bfround(s * u));
}

template <bool MVAR>
__global__ __launch_bounds__(TB) void image_probs_kernel(const bf16_t* __restrict__ cond, const bf16_t* __restrict__ ut,
                                                         const bf16_t* __restrict__ ui, int CB, float cfg_scale,
                                                         float cfg_img, bf16_t* __restrict__ probs_out,
                                                         int32_t* __restrict__ argmax_out,
                                                         bf16_t* __restrict__ pmax_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* e = (float*)smem;  // [CB]
    __shared__ float s_f[TB / 64];
    __shared__ double s_d[TB / 64];
    __shared__ int s_i[TB / 64];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool use_t = MVAR || (cfg_scale != 0.0f && ut != nullptr), use_i = !MVAR && cfg_img != 0.0f && ui != nullptr;
    const bf16_t* crow = cond + (size_t)row * CB;
    const bf16_t* trow = use_t ? ut + (size_t)row * CB : crow;
    const bf16_t* irow = use_i ? ui + (size_t)row * CB : crow;
    const int nchunk = CB >> 3;

    float mx = -INFINITY;
    for (int c = tid; c < nchunk; c += TB) {
        const u32x4 cv = ((const u32x4*)crow)[c];
        u32x4 tv = cv, iv = cv;
        if (use_t) tv = ((const u32x4*)trow)[c];
        if (use_i) iv = ((const u32x4*)irow)[c];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float l0, l1;
            if constexpr (MVAR) {  // ut = uncond branch, cfg_scale = image_cfg, cfg_img = 1 + image_cfg (host-computed)
                l0 = cfg_combine_m(__uint_as_float(cv[j] << 16), __uint_as_float(tv[j] << 16), cfg_img, cfg_scale);
                l1 = cfg_combine_m(__uint_as_float(cv[j] & 0xffff0000u), __uint_as_float(tv[j] & 0xffff0000u), cfg_img,
                                   cfg_scale);
            } else {
                l0 = cfg_combine(__uint_as_float(cv[j] << 16), __uint_as_float(tv[j] << 16),
                                 __uint_as_float(iv[j] << 16), use_t, use_i, cfg_scale, cfg_img);
                l1 = cfg_combine(__uint_as_float(cv[j] & 0xffff0000u), __uint_as_float(tv[j] & 0xffff0000u),
                                 __uint_as_float(iv[j] & 0xffff0000u), use_t, use_i, cfg_scale, cfg_img);
            }
            e[c * 8 + 2 * j] = l0;
            e[c * 8 + 2 * j + 1] = l1;
            mx = fmaxf(mx, fmaxf(l0, l1));
        }
    }
    mx = wave_max(mx);
    if (lane == 0) s_f[wave] = mx;
    __syncthreads();
    mx = s_f[0];
#pragma unroll
    for (int w = 1; w < TB / 64; ++w) mx = fmaxf(mx, s_f[w]);

    // e_i = exp(l_i - max) rounded to fp32; the row sum is accumulated in fp64 (order-independent to fp32 precision)
    double sum = 0.0;
    for (int i = tid; i < CB; i += TB) {
        const float ev = (float)exp((double)(e[i] - mx));
        e[i] = ev;
        sum += (double)ev;
    }
    sum = wave_sum_d(sum);
    if (lane == 0) s_d[wave] = sum;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < TB / 64; ++w) tot += s_d[w];
    const float fsum = (float)tot;

    // p_i = bf16(e_i / sum); argmax over the bf16 values, lowest index wins
    float best = -1.0f;
    int bidx = 0x7fffffff;
    for (int i = tid; i < CB; i += TB) {
        const bf16_t pb = f2bf(e[i] / fsum);
        if (probs_out) probs_out[(size_t)row * CB + i] = pb;
        const float p = bf2f(pb);
        if (p > best) { best = p; bidx = i; }  // i ascends per thread -> first index kept
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    __syncthreads();
    if (lane == 0) { s_f[wave] = best; s_i[wave] = bidx; }
    __syncthreads();
    if (tid == 0) {
        best = s_f[0]; bidx = s_i[0];
        for (int w = 1; w < TB / 64; ++w)
            if (s_f[w] > best || (s_f[w] == best && s_i[w] < bidx)) { best = s_f[w]; bidx = s_i[w]; }
        if (bidx == 0x7fffffff) bidx = 0;
        argmax_out[row] = bidx;
        pmax_out[row] = f2bf(best < 0.f ? 0.f : best);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Image step part 2: keep known tokens, confidence, stable lowest-k re-mask, write back
// (generators/parallel_generator.py:221-233, 304-344; mask_by_random_topk :23-70).
// ---------------------------------------------------------------------------------------------------------------
// MVAR (MMaDA-Parallel-M/models/sampling.py:31-36 + modeling_mmada.py:225-241): no clamp of known ids,
// confidence = log(clamp(p, 1e-20)) + temperature * gumbel (caller supplies the gumbel tensor), and the re-mask rule is
// the cut-off compare  confidence < sorted[mask_len]  (ties with the cut-off value are NOT masked).
template <bool MVAR>
__global__ __launch_bounds__(1024) void image_commit_kernel(int64_t* __restrict__ ids, int L,
                                                            const int32_t* __restrict__ pos_map, int N,
                                                            const int32_t* __restrict__ sampled_in,
                                                            const bf16_t* __restrict__ p_in,
                                                            const bf16_t* __restrict__ noise, float remask_temp,
                                                            const int32_t* __restrict__ mask_len_sched, int mask_id,
                                                            int text_vocab, int codebook, int keep_rule) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* conf = (float*)smem;          // [N] bf16-valued
    int* samp = (int*)(conf + N);        // [N]
    __shared__ int s_unknown;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) s_unknown = 0;
    __syncthreads();
    int64_t* row = ids + (size_t)b * L;
    int my_unknown = 0;
    for (int n = tid; n < N; n += blockDim.x) {
        const long long tok = row[pos_map[n]];
        const bool unknown = tok == (long long)mask_id;
        long long vq = tok - text_vocab;
        if constexpr (!MVAR) vq = vq < 0 ? 0 : (vq > codebook - 1 ? codebook - 1 : vq);
        int sidx = unknown ? sampled_in[(size_t)b * N + n] : (int)vq;
        if constexpr (!MVAR) sidx = sidx < 0 ? 0 : (sidx > codebook - 1 ? codebook - 1 : sidx);
        // selected_probs = unknown ? probs[sampled] : finfo(bf16).max  (:311-315)
        const float p = unknown ? bf2f(p_in[(size_t)b * N + n]) : bf2f((bf16_t)0x7f7f);
        // confidence = log(probs + 1e-10) + temperature * noise, each op rounded to bf16 (:36); torch rounds the
        // scalar of a bf16 `tensor + scalar` to bf16 first, while `scalar * tensor` multiplies in fp32
        float c;
        if constexpr (MVAR) {
            c = bfround((float)log((double)(p < 1e-20f ? bfround(1e-20f) : p)));  // log(t.clamp(min=1e-20)) in bf16
            c = bfround(c + bfround(remask_temp * bf2f(noise[(size_t)b * N + n])));
        } else {
            c = bfround((float)log((double)bfround(p + bfround(1e-10f))));
            if (noise) c = bfround(c + bfround(remask_temp * bf2f(noise[(size_t)b * N + n])));
        }
        conf[n] = c;
        samp[n] = sidx;
        my_unknown += unknown;
    }
    if (my_unknown) atomicAdd(&s_unknown, my_unknown);
    __syncthreads();
    // mask_len = max(1, min(unknown-1, floor(N*ratio))) (:318-324), then clamp(…, 0, N-1) (:43)
    // keep_rule (generate_image, generators/image_generation_generator.py:99-103 + utils/generation_utils.py:62):
    // k = keep_n.clamp(0, unknown-1) with keep_n supplied as is (0 on the last step) — no floor of 1
    int k = min(s_unknown - 1, mask_len_sched[0]);
    k = keep_rule ? max(0, k) : max(1, k);
    k = max(0, min(k, N - 1));
    for (int n = tid; n < N; n += blockDim.x) {
        const float c = conf[n];
        int rank = 0;
        if constexpr (MVAR) {
            // conf < sorted[k]  <=>  at most k elements are <= conf (its own tie class sits below position k)
            for (int j = 0; j < N; ++j) rank += conf[j] <= c;
            row[pos_map[n]] = (rank <= k) ? (int64_t)mask_id : (int64_t)((long long)samp[n] + text_vocab);
        } else {
            for (int j = 0; j < N; ++j) {
                const float cj = conf[j];
                rank += (cj < c) || (cj == c && j < n);
            }
            row[pos_map[n]] = (rank < k) ? (int64_t)mask_id : (int64_t)(samp[n] + text_vocab);
        }
    }
}

}  // namespace

// remasking == 'random' (generators/parallel_generator.py:194-198): the confidence of a masked position is a uniform draw
// instead of its soft-max probability; already-unmasked positions keep -inf (:203)
__global__ void text_random_conf_kernel(double* __restrict__ conf, const float* __restrict__ u, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && conf[i] != -INFINITY) conf[i] = (double)u[i];
}

int launch_text_select(const bf16_t* logits, const bf16_t* noisy, const bf16_t* unc, float text_cfg, const int32_t* x0_in,
                       int B, int T, int V, int ld, int64_t* ids, int L, int text_start, const int32_t* k, void* scratch,
                       int mask_id, hipStream_t s, const float* rand_conf) {
    if (B <= 0 || T <= 0) return 0;
    if (ld % 8) return mm_fail("text_select: ld_logits must be a multiple of 8");
    if (T > 8192) return mm_fail("text_select: T=%d too large", T);
    double* conf = (double*)scratch;
    int32_t* x0 = (int32_t*)((char*)scratch + (size_t)B * T * 8);
    if (unc)
        hipLaunchKernelGGL(text_row_stats_kernel<true>, dim3(B * T), dim3(TB), 0, s, logits, noisy, unc, text_cfg, x0_in, T,
                           V, ld, ids, L, text_start, mask_id, conf, x0);
    else
        hipLaunchKernelGGL(text_row_stats_kernel<false>, dim3(B * T), dim3(TB), 0, s, logits, noisy, unc, text_cfg, x0_in,
                           T, V, ld, ids, L, text_start, mask_id, conf, x0);
    MM_CHECK_HIP(hipGetLastError());
    if (rand_conf) {
        hipLaunchKernelGGL(text_random_conf_kernel, dim3((B * T + 255) / 256), dim3(256), 0, s, conf, rand_conf, B * T);
        MM_CHECK_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(text_commit_kernel, dim3(B), dim3(TB), (size_t)T * 8, s, conf, x0, T, ids, L, text_start, k);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_image_probs(const bf16_t* cond, const bf16_t* ut, const bf16_t* ui, int B, int N, int CB, float cfg_scale,
                       float cfg_img, bf16_t* probs_out, int32_t* argmax_out, bf16_t* pmax_out, int mvar, hipStream_t s) {
    if (B * N <= 0) return 0;
    if (CB % 8 || CB > 16384) return mm_fail("image_probs: codebook=%d must be a multiple of 8 and <= 16384", CB);
    static MmOncePerDevice attr_set;
    MM_ONCE_PER_DEVICE(attr_set,
        MM_CHECK_HIP(hipFuncSetAttribute((const void*)image_probs_kernel<false>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4));
        MM_CHECK_HIP(hipFuncSetAttribute((const void*)image_probs_kernel<true>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4)));
    if (mvar)
        hipLaunchKernelGGL(image_probs_kernel<true>, dim3(B * N), dim3(TB), (size_t)CB * 4, s, cond, ut, ui, CB, cfg_scale,
                           cfg_img, probs_out, argmax_out, pmax_out);
    else
        hipLaunchKernelGGL(image_probs_kernel<false>, dim3(B * N), dim3(TB), (size_t)CB * 4, s, cond, ut, ui, CB, cfg_scale,
                           cfg_img, probs_out, argmax_out, pmax_out);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_image_commit(int64_t* ids, int B, int L, const int32_t* pos_map, int N, const int32_t* sampled_in,
                        const bf16_t* p_in, const bf16_t* noise, float remask_temp, const int32_t* mask_len_sched,
                        int mask_id, int text_vocab, int codebook, int mvar, hipStream_t s) {
    if (B <= 0 || N <= 0) return 0;
    if (N > 8192) return mm_fail("image_commit: N=%d too large", N);
    static MmOncePerDevice attr_set;  // N * 8 B of dynamic LDS + the kernel's static LDS: above the 64 KiB default near N = 8192
    MM_ONCE_PER_DEVICE(attr_set,
        MM_CHECK_HIP(hipFuncSetAttribute((const void*)image_commit_kernel<true>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8));
        MM_CHECK_HIP(hipFuncSetAttribute((const void*)image_commit_kernel<false>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8)));
    if (mvar) {
        if (!noise) return mm_fail("image_commit (M variant): the gumbel tensor is required");
        hipLaunchKernelGGL(image_commit_kernel<true>, dim3(B), dim3(1024), (size_t)N * 8, s, ids, L, pos_map, N, sampled_in,
                           p_in, noise, remask_temp, mask_len_sched, mask_id, text_vocab, codebook, mvar == 2);
    } else
        hipLaunchKernelGGL(image_commit_kernel<false>, dim3(B), dim3(1024), (size_t)N * 8, s, ids, L, pos_map, N, sampled_in,
                           p_in, noise, remask_temp, mask_len_sched, mask_id, text_vocab, codebook, 0);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_text_stats_partial(const bf16_t* logits, int B, int T, int Vl, int ld, int col_off, const int64_t* ids, int L,
                              int text_start, int mask_id, TextStat* out, hipStream_t s) {
    if (B * T <= 0) return 0;
    if (ld % 8) return mm_fail("text_stats_partial: ld must be a multiple of 8");
    hipLaunchKernelGGL(text_row_stats_partial_kernel, dim3(B * T), dim3(TB), 0, s, logits, T, Vl, ld, col_off, ids, L,
                       text_start, mask_id, out);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_text_commit(const void* scratch, int B, int T, int64_t* ids, int L, int text_start, const int32_t* k, hipStream_t s) {
    if (T > 8192) return mm_fail("text_commit: T=%d too large", T);
    const double* conf = (const double*)scratch;
    const int32_t* x0 = (const int32_t*)((const char*)scratch + (size_t)B * T * 8);
    hipLaunchKernelGGL(text_commit_kernel, dim3(B), dim3(TB), (size_t)T * 8, s, conf, x0, T, ids, L, text_start, k);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}
