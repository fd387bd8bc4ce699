// elementwise.hip — HBM-bound row kernels of the denoiser: embedding gather, RMSNorm, RoPE table, weight repack.
// All bf16 traffic is 16 bytes per lane (guide G13); one wave64 owns one row so the reduction is shuffle-only.
#include "kernels.h"

namespace {

// ---- RMSLayerNorm.forward (model/modeling_llada.py:315-329) ------------------------------------------------------
//   x32 = x.float(); var = mean(x32^2); y = bf16(x32 * rsqrt(var + eps)); out = bf16(w * y)   (cast-then-scale)
MM_DEVICE void rmsnorm_row(const bf16_t* __restrict__ xr, const bf16_t* __restrict__ w, bf16_t* __restrict__ outr,
                           int d, float eps, int lane) {
    const int nchunk = d >> 3;
    float ss = 0.f;
    for (int c = lane; c < nchunk; c += 64) {
        const u32x4 v = ((const u32x4*)xr)[c];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = __uint_as_float(v[j] << 16), hi = __uint_as_float(v[j] & 0xffff0000u);
            ss += lo * lo;
            ss += hi * hi;
        }
    }
    ss = wave_sum(ss);
    const float var = ss / (float)d;
    const float rs = 1.0f / sqrtf(var + eps);  // torch.rsqrt in fp32; IEEE div+sqrt keeps us within 1 ulp of it
    for (int c = lane; c < nchunk; c += 64) {
        const u32x4 v = ((const u32x4*)xr)[c];  // second read hits L1/L2 (row <= 16 KiB)
        const u32x4 wv = ((const u32x4*)w)[c];
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = __uint_as_float(v[j] << 16), hi = __uint_as_float(v[j] & 0xffff0000u);
            const float wl = __uint_as_float(wv[j] << 16), wh = __uint_as_float(wv[j] & 0xffff0000u);
            const float yl = bfround(lo * rs), yh = bfround(hi * rs);
            o[j] = pack_bf2(wl * yl, wh * yh);
        }
        ((u32x4*)outr)[c] = o;
    }
}

// d == 512*NCH fast path: the whole row lives in registers (NCH x 16 B per lane, all loads in flight at once), one
// HBM read + one write per element.  Same arithmetic as rmsnorm_row.
template <int NCH>
MM_DEVICE void rmsnorm_row_regs(const bf16_t* __restrict__ xr, const bf16_t* __restrict__ w, bf16_t* __restrict__ outr,
                                float eps, int lane) {
    u32x4 v[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) v[i] = ((const u32x4*)xr)[lane + 64 * i];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = __uint_as_float(v[i][j] << 16), hi = __uint_as_float(v[i][j] & 0xffff0000u);
            ss += lo * lo;
            ss += hi * hi;
        }
    ss = wave_sum(ss);
    const float var = ss / (float)(NCH * 512);
    const float rs = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const u32x4 wv = ((const u32x4*)w)[lane + 64 * i];
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = __uint_as_float(v[i][j] << 16), hi = __uint_as_float(v[i][j] & 0xffff0000u);
            const float wl = __uint_as_float(wv[j] << 16), wh = __uint_as_float(wv[j] & 0xffff0000u);
            o[j] = pack_bf2(wl * bfround(lo * rs), wh * bfround(hi * rs));
        }
        ((u32x4*)outr)[lane + 64 * i] = o;
    }
}

// ---- embedding gather: x[b*Lp + l] = wte[ids[b*L + l]] (model/modeling_llada.py:1265), pad rows zero ----------
// Fused with the first RMSNorm of the forward (block 0's attn_norm, :924): the wave that gathered the row normalises it
// while it is still in L1 — same row routine as rmsnorm_kernel, hence the same bits.
__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ ids, const bf16_t* __restrict__ wte,
                                                    bf16_t* __restrict__ x, int B, int L, int Lp, int d, int vocab,
                                                    const bf16_t* __restrict__ norm_w, bf16_t* __restrict__ xn, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B * Lp) return;
    const int b = row / Lp, l = row - b * Lp;
    u32x4* dst = (u32x4*)(x + (size_t)row * d);
    const int nchunk = d >> 3;
    if (l < L) {
        long long id = ids[(size_t)b * L + l];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // torch would raise; clamp keeps the device safe
        const u32x4* src = (const u32x4*)(wte + (size_t)id * d);
        for (int c = lane; c < nchunk; c += 64) dst[c] = src[c];
    } else {
        for (int c = lane; c < nchunk; c += 64) dst[c] = u32x4{0, 0, 0, 0};
    }
    if (norm_w) {  // every lane re-reads exactly the chunks it wrote
        __threadfence_block();
        if (d == 4096)
            rmsnorm_row_regs<8>(x + (size_t)row * d, norm_w, xn + (size_t)row * d, eps, lane);
        else
            rmsnorm_row(x + (size_t)row * d, norm_w, xn + (size_t)row * d, d, eps, lane);
    }
}

__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                      bf16_t* __restrict__ out, int rows, int d, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    if (d == 4096)
        rmsnorm_row_regs<8>(x + (size_t)row * d, w, out + (size_t)row * d, eps, threadIdx.x & 63);
    else
        rmsnorm_row(x + (size_t)row * d, w, out + (size_t)row * d, d, eps, threadIdx.x & 63);
}

__global__ __launch_bounds__(256) void rmsnorm_gather_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                             bf16_t* __restrict__ out, const int32_t* __restrict__ rows,
                                                             int R, int L, int Lp, int d, float eps, int row_off, int nflat) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int flat = min(max(rows[r], 0), nflat - 1);  // a bad row index must not become an out-of-bounds read
    const int b = flat / L;
    const int l = min(max(flat - b * L - row_off, 0), Lp - 1);
    if (d == 4096)
        rmsnorm_row_regs<8>(x + ((size_t)b * Lp + l) * d, w, out + (size_t)r * d, eps, threadIdx.x & 63);
    else
        rmsnorm_row(x + ((size_t)b * Lp + l) * d, w, out + (size_t)r * d, d, eps, threadIdx.x & 63);
}

// ---- RoPE table (model/modeling_llada.py:391-397): freqs = seq (x) inv_freq in fp32; sin/cos of the fp32 angle.
// The angle product is a single IEEE fp32 multiply (bit-identical to torch); sin/cos are evaluated in fp64 and
// rounded once, i.e. the correctly rounded fp32 value torch's libm sin/cos aim at.
__global__ void rope_table_kernel(float* __restrict__ cos_t, float* __restrict__ sin_t,
                                  const float* __restrict__ inv_freq, int max_seq) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= max_seq * 64) return;
    const int l = idx >> 6, i = idx & 63;
    const float ang = (float)l * inv_freq[i];
    cos_t[idx] = (float)cos((double)ang);
    sin_t[idx] = (float)sin((double)ang);
}

__global__ __launch_bounds__(256) void unpad_rows_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int B,
                                                         int L, int Lp, int d) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * L) return;
    const int b = row / L, l = row - b * L;
    const u32x4* src = (const u32x4*)(x + ((size_t)b * Lp + l) * d);
    u32x4* dst = (u32x4*)(out + (size_t)row * d);
    for (int c = threadIdx.x & 63; c < (d >> 3); c += 64) dst[c] = src[c];
}

__global__ void iota_kernel(int32_t* rows, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rows[i] = i;
}

// ---- dLLM cache helpers (compute-mask forward, model/modeling_llada.py:929-937,1244-1245,1409-1411) ----------------
// posmap[b*Lp + i] = position of the i-th computed token of sequence b, -1 on the pad rows of the compact stream;
// a position outside [0, L) also becomes -1 (torch would raise; the device stays safe)
__global__ void expand_pos_kernel(const int32_t* __restrict__ pos, int32_t* __restrict__ posmap, int B, int Tc, int Lp, int L) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= B * Lp) return;
    const int b = m / Lp, i = m - b * Lp;
    int p = -1;
    if (i < Tc) {
        p = pos[b * Tc + i];
        if (p < 0 || p >= L) p = -1;
    }
    posmap[m] = p;
}

// dst[b*Lp_dst + posmap[m]] = src[m] for the mapped rows of the compact stream (one wave per row)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                           const int32_t* __restrict__ posmap, int M, int Lp, int Lp_dst,
                                                           int d) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int p = posmap[m];
    if (p < 0) return;
    const int b = m / Lp;
    const u32x4* s = (const u32x4*)(src + (size_t)m * d);
    u32x4* o = (u32x4*)(dst + ((size_t)b * Lp_dst + p) * d);
    for (int c = threadIdx.x & 63; c < (d >> 3); c += 64) o[c] = s[c];
}

// ---- weight repack (one wave copies one output row of d bf16) --------------------------------------------------
// Fused QKV with the rotary-partner permutation: inside each q/k head, packed column c = 32*p + w holds original
// feature i = 16*p + (w & 15) + (w >= 16 ? 64 : 0), so the two 16-wide MFMA fragments 2p and 2p+1 of a lane are
// features i and i+64 — rotate_half partners (model/modeling_llada.py:402-406).  q·k is invariant under the
// permutation because q and k use the same one.  TP: rank r owns heads [r*H/tp, (r+1)*H/tp).
__global__ __launch_bounds__(256) void pack_qkv_kernel(const bf16_t* __restrict__ wq, const bf16_t* __restrict__ wk,
                                                       const bf16_t* __restrict__ wv, bf16_t* __restrict__ out, int d,
                                                       int Hq, int Hkv, int tp_rank, int tp_size) {
    const int hq_l = Hq / tp_size, hkv_l = Hkv / tp_size;
    const int nrows = (hq_l + 2 * hkv_l) * 128;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int head = row >> 7, c = row & 127;
    const bf16_t* src;
    if (head < hq_l + hkv_l) {
        const int p = c >> 5, w = c & 31;
        const int i = 16 * p + (w & 15) + (w >= 16 ? 64 : 0);
        if (head < hq_l)
            src = wq + ((size_t)(tp_rank * hq_l + head) * 128 + i) * d;
        else
            src = wk + ((size_t)(tp_rank * hkv_l + head - hq_l) * 128 + i) * d;
    } else {
        src = wv + ((size_t)(tp_rank * hkv_l + head - hq_l - hkv_l) * 128 + c) * d;
    }
    u32x4* dst = (u32x4*)(out + (size_t)row * d);
    for (int ch = threadIdx.x & 63; ch < (d >> 3); ch += 64) dst[ch] = ((const u32x4*)src)[ch];
}

// ff_proj (gate) and up_proj rows interleaved in blocks of 16: packed row n = 32*q + w -> feature j = 16*q + (w&15)
// of ff_proj (w < 16) or up_proj (w >= 16), so fragments 2q, 2q+1 of a lane are silu-input and multiplier.
__global__ __launch_bounds__(256) void pack_gate_up_kernel(const bf16_t* __restrict__ gate, const bf16_t* __restrict__ up,
                                                           bf16_t* __restrict__ out, int d, int F, int tp_rank,
                                                           int tp_size) {
    const int f_l = F / tp_size;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= 2 * f_l) return;
    const int q = row >> 5, w = row & 31;
    const int j = tp_rank * f_l + 16 * q + (w & 15);
    const bf16_t* src = (w < 16 ? gate : up) + (size_t)j * d;
    u32x4* dst = (u32x4*)(out + (size_t)row * d);
    for (int ch = threadIdx.x & 63; ch < (d >> 3); ch += 64) dst[ch] = ((const u32x4*)src)[ch];
}

// row-parallel weights (attn_out [d, Hq*128], ff_out [d, F]): rank r keeps columns [r*cols/tp, (r+1)*cols/tp)
__global__ __launch_bounds__(256) void pack_cols_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ out, int rows,
                                                        int cols, int tp_rank, int tp_size) {
    const int c_l = cols / tp_size;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const u32x4* src = (const u32x4*)(w + (size_t)row * cols + (size_t)tp_rank * c_l);
    u32x4* dst = (u32x4*)(out + (size_t)row * c_l);
    for (int ch = threadIdx.x & 63; ch < (c_l >> 3); ch += 64) dst[ch] = src[ch];
}

// [BH, L, 128] -> [BH, Lkv, 128] (rows >= L zero)
__global__ __launch_bounds__(256) void pad_heads_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int BH,
                                                        int L, int Lkv) {
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4);  // 16 lanes x 16 B = one 128-element row
    if (row >= BH * Lkv) return;
    const int bh = row / Lkv, l = row - bh * Lkv;
    u32x4 v = u32x4{0, 0, 0, 0};
    if (l < L) v = ((const u32x4*)(in + ((size_t)bh * L + l) * 128))[threadIdx.x & 15];
    ((u32x4*)(out + (size_t)row * 128))[threadIdx.x & 15] = v;
}

// [BH, L, 128] -> [BH, 128, Lkv] (columns >= L zero); 64x64 LDS tile transpose
__global__ __launch_bounds__(256) void transpose_v_kernel(const bf16_t* __restrict__ v, bf16_t* __restrict__ vT, int BH,
                                                          int L, int Lkv) {
    __shared__ bf16_t tile[64][66];
    const int bh = blockIdx.z, l0 = blockIdx.x * 64, d0 = blockIdx.y * 64;
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int r = e >> 6, c = e & 63;
        tile[r][c] = (l0 + r < L) ? v[((size_t)bh * L + l0 + r) * 128 + d0 + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int r = e >> 6, c = e & 63;  // r: d, c: l
        if (l0 + c < Lkv) vT[((size_t)bh * 128 + d0 + r) * Lkv + vt_key_pos(l0 + c)] = tile[c][r];
    }
}

// LFQ codebook entry (MMaDA-Parallel-M/models/modeling_magvitv2.py:186-194,208-221):
// mask = 2^arange(nbits-1,-1,-1); bits = (idx & mask) != 0; out = bits*2 - 1, laid out [B, nbits, N]
__global__ void lfq_gather_kernel(const int64_t* __restrict__ idx, void* __restrict__ out, int B, int N, int nbits,
                                  int f32) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * N) return;
    const int b = i / N, n = i - b * N;
    const long long v = idx[i];
    for (int c = 0; c < nbits; ++c) {
        const float val = ((v >> (nbits - 1 - c)) & 1) ? 1.0f : -1.0f;
        const size_t o = ((size_t)b * nbits + c) * N + n;
        if (f32)
            ((float*)out)[o] = val;
        else
            ((bf16_t*)out)[o] = f2bf(val);
    }
}

}  // namespace

#define LAUNCH_CHECK()                  \
    MM_CHECK_HIP(hipGetLastError()); \
    return 0

int launch_embed(const int64_t* ids, const bf16_t* wte, bf16_t* x, int B, int L, int Lp, int d, int vocab, hipStream_t s,
                 const bf16_t* norm_w, bf16_t* xn, float eps) {
    hipLaunchKernelGGL(embed_kernel, dim3((B * Lp + 3) / 4), dim3(256), 0, s, ids, wte, x, B, L, Lp, d, vocab, norm_w, xn, eps);
    LAUNCH_CHECK();
}
int launch_rmsnorm(const bf16_t* x, const bf16_t* w, bf16_t* out, int rows, int d, float eps, hipStream_t s) {
    if (d % 8) return mm_fail("rmsnorm: d must be a multiple of 8");
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(rmsnorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, w, out, rows, d, eps);
    LAUNCH_CHECK();
}
int launch_rmsnorm_gather(const bf16_t* x, const bf16_t* w, bf16_t* out, const int32_t* rows, int R, int L, int Lp, int d,
                          float eps, hipStream_t s, int row_off, int nflat) {
    if (R <= 0) return 0;
    if (nflat <= 0) nflat = 0x7fffffff;
    hipLaunchKernelGGL(rmsnorm_gather_kernel, dim3((R + 3) / 4), dim3(256), 0, s, x, w, out, rows, R, L, Lp, d, eps, row_off, nflat);
    LAUNCH_CHECK();
}
int launch_rope_table(float* cos_t, float* sin_t, const float* inv_freq_dev, int max_seq, hipStream_t s) {
    hipLaunchKernelGGL(rope_table_kernel, dim3((max_seq * 64 + 255) / 256), dim3(256), 0, s, cos_t, sin_t, inv_freq_dev,
                       max_seq);
    LAUNCH_CHECK();
}
int launch_unpad_rows(const bf16_t* x, bf16_t* out, int B, int L, int Lp, int d, hipStream_t s) {
    hipLaunchKernelGGL(unpad_rows_kernel, dim3((B * L + 3) / 4), dim3(256), 0, s, x, out, B, L, Lp, d);
    LAUNCH_CHECK();
}
int launch_iota_rows(int32_t* rows, int n, hipStream_t s) {
    hipLaunchKernelGGL(iota_kernel, dim3((n + 255) / 256), dim3(256), 0, s, rows, n);
    LAUNCH_CHECK();
}
int launch_expand_pos(const int32_t* pos, int32_t* posmap, int B, int Tc, int Lp, int L, hipStream_t s) {
    hipLaunchKernelGGL(expand_pos_kernel, dim3((B * Lp + 255) / 256), dim3(256), 0, s, pos, posmap, B, Tc, Lp, L);
    LAUNCH_CHECK();
}
int launch_scatter_rows(const bf16_t* src, bf16_t* dst, const int32_t* posmap, int M, int Lp, int Lp_dst, int d,
                        hipStream_t s) {
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, s, src, dst, posmap, M, Lp, Lp_dst, d);
    LAUNCH_CHECK();
}
int launch_pack_qkv(const bf16_t* wq, const bf16_t* wk, const bf16_t* wv, bf16_t* out, int d, int Hq, int Hkv, int tp_rank,
                    int tp_size, hipStream_t s) {
    const int nrows = (Hq / tp_size + 2 * (Hkv / tp_size)) * 128;
    hipLaunchKernelGGL(pack_qkv_kernel, dim3((nrows + 3) / 4), dim3(256), 0, s, wq, wk, wv, out, d, Hq, Hkv, tp_rank,
                       tp_size);
    LAUNCH_CHECK();
}
int launch_pack_gate_up(const bf16_t* gate, const bf16_t* up, bf16_t* out, int d, int F, int tp_rank, int tp_size,
                        hipStream_t s) {
    const int nrows = 2 * (F / tp_size);
    hipLaunchKernelGGL(pack_gate_up_kernel, dim3((nrows + 3) / 4), dim3(256), 0, s, gate, up, out, d, F, tp_rank, tp_size);
    LAUNCH_CHECK();
}
int launch_pack_cols(const bf16_t* w, bf16_t* out, int rows, int cols, int tp_rank, int tp_size, hipStream_t s) {
    hipLaunchKernelGGL(pack_cols_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, w, out, rows, cols, tp_rank, tp_size);
    LAUNCH_CHECK();
}
int launch_pad_heads(const bf16_t* in, bf16_t* out, int BH, int L, int Lkv, hipStream_t s) {
    hipLaunchKernelGGL(pad_heads_kernel, dim3((BH * Lkv + 15) / 16), dim3(256), 0, s, in, out, BH, L, Lkv);
    LAUNCH_CHECK();
}
int launch_transpose_v(const bf16_t* v, bf16_t* vT, int BH, int L, int Lkv, hipStream_t s) {
    hipLaunchKernelGGL(transpose_v_kernel, dim3(Lkv / 64, 2, BH), dim3(256), 0, s, v, vT, BH, L, Lkv);
    LAUNCH_CHECK();
}
int launch_lfq_gather(const int64_t* idx, void* out, int B, int N, int nbits, int f32, hipStream_t s) {
    hipLaunchKernelGGL(lfq_gather_kernel, dim3((B * N + 255) / 256), dim3(256), 0, s, idx, out, B, N, nbits, f32);
    LAUNCH_CHECK();
}
