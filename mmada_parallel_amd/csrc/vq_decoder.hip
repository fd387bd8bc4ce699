// vq_decoder.hip — MAGVITv2 token -> pixel decode of MMaDA-Parallel-M on gfx950, fp32 like the
// reference (MMaDA-Parallel-M/inference.py:56-59 keeps the VQ model in fp32):
//   MAGVITv2.decode_code          models/modeling_magvitv2.py:429-433
//   LFQuantizer.get_codebook_entry  :208-221      VQGANDecoder.forward  :369-406
//   ResnetBlock / AttnBlock / Upsample / Normalize / swish   models/common_modules.py:337-357,187-211,36-40,16-24
// Layout: activations are NHWC fp32 ([B, H, W, C], channels contiguous), so a 3x3 convolution is an implicit GEMM
// with M = B*H*W pixels, N = Cout, K = 9*Cin whose A-tile rows are 128-byte channel runs of shifted pixels; conv
// weights are repacked once at bind time to [Cout][tap][Cin].  The 2x nearest upsample is folded into the A-tile
// addressing of the convolution that follows it, bias / residual adds into the epilogue.
// Kernels: conv_mfma_kernel (v_mfma_f32_32x32x2_f32, 128x128x32 tiles, register-prefetched LDS staging),
// conv_direct_kernel (Cin = 13: scalar-weight FMA), conv_thin_kernel (conv_out: 3 channels out), gn_stats/gn_apply (GroupNorm(32) + swish, fp64
// moments, deterministic two-level reduction), softmax_rows, transpose, lfq_nhwc.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/mmada_mi355x.h"
#include "common.h"

namespace {

constexpr int GN_GROUPS = 32;
constexpr int GN_MAX_CHUNKS = 256;

struct ConvArgs {
    const float* in;     // [B, Hi, Wi, Cin]
    const float* w;      // [Cout][taps][Cin]
    const float* bias;   // [Cout] or null
    const float* resid;  // [B, Ho, Wo, Cout] or null (may alias out)
    float* out;          // [B, Ho, Wo, Cout]  (nchw_out: [B, Cout, Ho, Wo], direct kernel only)
    int B, Hi, Wi, Cin, Cout, Ho, Wo, taps, ups, down, nchw_out;
    long long M;         // B * Ho * Wo
};

// Input pixel of output pixel (oh, ow) for filter tap `tap`; false = zero padding.
//   default: stride 1, padding k/2 (Conv2d(k, 1, k//2))
//   ups:     the same on the 2x nearest-upsampled input (Upsample.forward, common_modules.py:36-40)
//   down:    Downsample.forward (common_modules.py:83-90): F.pad(x, (0,1,0,1)) then Conv2d(3, stride 2, padding 0)
MM_DEVICE bool conv_src(const ConvArgs& g, int oh, int ow, int tap, int& ih, int& iw) {
    if (g.down) {
        ih = 2 * oh + tap / 3;
        iw = 2 * ow + tap % 3;
        return ih < g.Hi && iw < g.Wi;
    }
    const int uh = oh + (g.taps == 9 ? tap / 3 - 1 : 0), uw = ow + (g.taps == 9 ? tap % 3 - 1 : 0);
    ih = uh >> g.ups;
    iw = uw >> g.ups;
    return uh >= 0 && uh < g.Ho && uw >= 0 && uw < g.Wo;
}

// ---------------------------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution on the fp32 matrix cores.  256 threads = 4 waves (2x2), wave tile 64x64 = 2x2 MFMA
// 32x32 blocks, K step 32 channels of one tap.  Global -> registers (next chunk) overlaps the MFMAs of the current
// chunk; LDS rows are padded to 36 floats so both the float4 stores and the float4 fragment reads are conflict-free.
constexpr int CBM = 128, CBN = 128, CBK = 32, CLD = CBK + 4;

__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvArgs g) {
    __shared__ __attribute__((aligned(16))) float As[CBM * CLD];
    __shared__ __attribute__((aligned(16))) float Bs[CBN * CLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long long m0 = (long long)blockIdx.x * CBM;
    const int n0 = blockIdx.y * CBN;
    const int lc = tid & 7, lr = tid >> 3;

    // per-thread loader rows: 4 pixel rows and 4 weight rows, 32 apart
    int pb[4], poh[4], pow_[4];
    bool pv[4];
    const float* wrow[4];
    bool wv[4];
    const int HoWo = g.Ho * g.Wo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long m = m0 + lr + 32 * i;
        pv[i] = m < g.M;
        const long long mm = pv[i] ? m : 0;
        pb[i] = (int)(mm / HoWo);
        const int r = (int)(mm - (long long)pb[i] * HoWo);
        poh[i] = r / g.Wo;
        pow_[i] = r - poh[i] * g.Wo;
        const int n = n0 + lr + 32 * i;
        wv[i] = n < g.Cout;
        wrow[i] = g.w + (size_t)(wv[i] ? n : 0) * g.taps * g.Cin + 4 * lc;
    }
    const int cchunks = g.Cin / CBK, nch = g.taps * cchunks;
    f32x4 ra[4], rb[4];
    auto gload = [&](int ch) {
        const int tap = ch / cchunks, c0 = (ch - tap * cchunks) * CBK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int ih, iw;
            const bool ok = conv_src(g, poh[i], pow_[i], tap, ih, iw) && pv[i];
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *(const f32x4*)(g.in + (((size_t)pb[i] * g.Hi + ih) * g.Wi + iw) * g.Cin + c0 + 4 * lc);
            ra[i] = v;
            f32x4 wv4 = {0.f, 0.f, 0.f, 0.f};
            if (wv[i]) wv4 = *(const f32x4*)(wrow[i] + (size_t)tap * g.Cin + c0);
            rb[i] = wv4;
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, kh = lane >> 5;
    gload(0);
    for (int ch = 0; ch < nch; ++ch) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(f32x4*)&As[(lr + 32 * i) * CLD + 4 * lc] = ra[i];
            *(f32x4*)&Bs[(lr + 32 * i) * CLD + 4 * lc] = rb[i];
        }
        __syncthreads();
        if (ch + 1 < nch) gload(ch + 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k4 = s * 8 + kh * 4;
            f32x4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *(const f32x4*)&As[(wm * 64 + i * 32 + frow) * CLD + k4];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *(const f32x4*)&Bs[(wn * 64 + j * 32 + frow) * CLD + k4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // D[m][n]: n = lane & 31, m = 8*(r>>2) + 4*(lane>>5) + (r&3)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long m = m0 + wm * 64 + i * 32 + 8 * (r >> 2) + 4 * kh + (r & 3);
            if (m >= g.M) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + frow;
                if (n >= g.Cout) continue;
                float v = acc[i][j][r];
                if (g.bias) v += g.bias[n];
                if (g.resid) v += g.resid[(size_t)m * g.Cout + n];
                g.out[(size_t)m * g.Cout + n] = v;
            }
        }
}

// Direct convolution for the thin ends of the network (z_channels = 13 in, 3 / 13 out): one thread per pixel and
// CO output channels; the weight index does not depend on the lane, so the weights come through the scalar cache.
template <int CO>
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvArgs g) {
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    const int co0 = blockIdx.y * CO;
    if (m >= g.M) return;
    const int HoWo = g.Ho * g.Wo;
    const int b = (int)(m / HoWo), r = (int)(m - (long long)b * HoWo), oh = r / g.Wo, ow = r - (r / g.Wo) * g.Wo;
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.f;
    const size_t wstride = (size_t)g.taps * g.Cin;
    for (int tap = 0; tap < g.taps; ++tap) {
        int ih, iw;
        if (!conv_src(g, oh, ow, tap, ih, iw)) continue;
        const float* src = g.in + (((size_t)b * g.Hi + ih) * g.Wi + iw) * g.Cin;
        const float* wt = g.w + (size_t)co0 * wstride + (size_t)tap * g.Cin;
        for (int ci = 0; ci < g.Cin; ++ci) {
            const float x = src[ci];
#pragma unroll
            for (int c = 0; c < CO; ++c)
                if (co0 + c < g.Cout) acc[c] = fmaf(x, wt[(size_t)c * wstride + ci], acc[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < CO; ++c) {
        const int n = co0 + c;
        if (n >= g.Cout) continue;
        float v = acc[c];
        if (g.bias) v += g.bias[n];
        if (g.nchw_out) {
            g.out[((size_t)b * g.Cout + n) * HoWo + r] = v;
        } else {
            if (g.resid) v += g.resid[(size_t)m * g.Cout + n];
            g.out[(size_t)m * g.Cout + n] = v;
        }
    }
}

// conv_out (wide in, <= 4 out): eight lanes share a pixel, each owning every 8th float4 of the channel run, so a tap
// is one coalesced 128-byte-per-pixel read; weights sit in LDS; the eight partial sums meet in a 3-step butterfly.
constexpr int THIN_CO = 4;
__global__ __launch_bounds__(256) void conv_thin_kernel(ConvArgs g) {
    extern __shared__ __attribute__((aligned(16))) float wsh[];  // [Cout][taps][Cin]
    const int tid = threadIdx.x, sub = tid & 7;
    const int wtotal = g.Cout * g.taps * g.Cin;
    for (int i = tid * 4; i < wtotal; i += 1024) *(f32x4*)&wsh[i] = *(const f32x4*)&g.w[i];
    __syncthreads();
    const long long m = (long long)blockIdx.x * 32 + (tid >> 3);
    const bool live = m < g.M;
    const long long mm = live ? m : 0;
    const int HoWo = g.Ho * g.Wo;
    const int b = (int)(mm / HoWo), r = (int)(mm - (long long)b * HoWo), oh = r / g.Wo, ow = r - (r / g.Wo) * g.Wo;
    float acc[THIN_CO] = {0.f, 0.f, 0.f, 0.f};
    const int wstride = g.taps * g.Cin;
    for (int tap = 0; tap < g.taps; ++tap) {
        int ih, iw;
        if (!conv_src(g, oh, ow, tap, ih, iw) || !live) continue;
        const float* src = g.in + (((size_t)b * g.Hi + ih) * g.Wi + iw) * g.Cin;
        for (int c = sub * 4; c < g.Cin; c += 32) {
            const f32x4 x = *(const f32x4*)(src + c);
#pragma unroll
            for (int co = 0; co < THIN_CO; ++co) {
                if (co >= g.Cout) break;
                const f32x4 w4 = *(const f32x4*)&wsh[co * wstride + tap * g.Cin + c];
                acc[co] = fmaf(x[0], w4[0], fmaf(x[1], w4[1], fmaf(x[2], w4[2], fmaf(x[3], w4[3], acc[co]))));
            }
        }
    }
#pragma unroll
    for (int co = 0; co < THIN_CO; ++co)
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) acc[co] += __shfl_xor(acc[co], o, 64);
    if (!live || sub >= g.Cout) return;
    float v = sub == 0 ? acc[0] : sub == 1 ? acc[1] : sub == 2 ? acc[2] : acc[3];
    if (g.bias) v += g.bias[sub];
    if (g.nchw_out) {
        g.out[((size_t)b * g.Cout + sub) * HoWo + r] = v;
    } else {
        if (g.resid) v += g.resid[(size_t)m * g.Cout + sub];
        g.out[(size_t)m * g.Cout + sub] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// GroupNorm(32 groups, eps 1e-6) over NHWC: pass 1 writes fp64 (sum, sum of squares) per (batch, group, pixel chunk);
// pass 2 sums the chunks in a fixed order, normalises, applies the affine and (optionally) swish.
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, double* __restrict__ partial, int HW,
                                                       int C, int nchunks) {
    __shared__ double sh_s[256], sh_q[256];
    const int C4 = C >> 2, PL = 256 / C4;
    const int tid = threadIdx.x, col = tid % C4, pl = tid / C4;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int per = (HW + nchunks - 1) / nchunks;
    const int p0 = chunk * per, p1 = min(HW, p0 + per);
    double s = 0.0, q = 0.0;
    if (pl < PL) {
        const float* base = x + (size_t)b * HW * C + 4 * col;
        for (int p = p0 + pl; p < p1; p += PL) {
            const f32x4 v = *(const f32x4*)(base + (size_t)p * C);
            s += (double)v[0] + (double)v[1] + (double)v[2] + (double)v[3];
            q += (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2] + (double)v[3] * v[3];
        }
    }
    sh_s[tid] = s;
    sh_q[tid] = q;
    __syncthreads();
    if (tid < GN_GROUPS) {
        const int cpg4 = C4 / GN_GROUPS;
        double ts = 0.0, tq = 0.0;
        for (int l = 0; l < PL; ++l)
            for (int c = 0; c < cpg4; ++c) {
                ts += sh_s[l * C4 + tid * cpg4 + c];
                tq += sh_q[l * C4 + tid * cpg4 + c];
            }
        double* o = partial + (((size_t)b * GN_GROUPS + tid) * nchunks + chunk) * 2;
        o[0] = ts;
        o[1] = tq;
    }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const double* __restrict__ partial,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ out, int HW, int C, int nchunks, int swish) {
    __shared__ float sh_mean[GN_GROUPS], sh_rstd[GN_GROUPS];
    const int b = blockIdx.y, tid = threadIdx.x, C4 = C >> 2, cpg4 = C4 / GN_GROUPS;
    if (tid < GN_GROUPS) {
        const double* p = partial + ((size_t)b * GN_GROUPS + tid) * nchunks * 2;
        double s = 0.0, q = 0.0;
        for (int c = 0; c < nchunks; ++c) {
            s += p[2 * c];
            q += p[2 * c + 1];
        }
        const double n = (double)HW * (C / GN_GROUPS);
        const double mean = s / n, var = fmax(q / n - mean * mean, 0.0);
        sh_mean[tid] = (float)mean;
        sh_rstd[tid] = (float)(1.0 / sqrt(var + 1e-6));
    }
    __syncthreads();
    const size_t total = (size_t)HW * C4;
    const float* xb = x + (size_t)b * HW * C;
    float* ob = out + (size_t)b * HW * C;
    for (size_t e = (size_t)blockIdx.x * 256 + tid; e < total; e += (size_t)gridDim.x * 256) {
        const int col = (int)(e % C4), grp = col / cpg4;
        const f32x4 v = *(const f32x4*)(xb + e * 4);
        const f32x4 ga = *(const f32x4*)(gamma + 4 * col), be = *(const f32x4*)(beta + 4 * col);
        const float mean = sh_mean[grp], rstd = sh_rstd[grp];
        f32x4 y;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t = (v[i] - mean) * rstd * ga[i] + be[i];
            if (swish) t = t / (1.0f + expf(-t));
            y[i] = t;
        }
        *(f32x4*)(ob + e * 4) = y;
    }
}

// softmax over the rows of S [rows, n] in place, after scaling (AttnBlock: w_ * c^-0.5 then softmax(dim=2))
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ S, int n, float scale) {
    __shared__ float red[4];
    float* row = S + (size_t)blockIdx.x * n;
    const int tid = threadIdx.x;
    float mx = -INFINITY;
    for (int i = tid; i < n; i += 256) mx = fmaxf(mx, row[i] * scale);
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int i = tid; i < n; i += 256) {
        const float e = expf(row[i] * scale - mx);
        row[i] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int i = tid; i < n; i += 256) row[i] *= inv;
}

// out[c][r] = in[r][c]
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cc) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < R && c0 + tx < Cc) tile[j][tx] = in[(size_t)(r0 + j) * Cc + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < Cc && r0 + tx < R) out[(size_t)(c0 + j) * R + r0 + tx] = tile[tx][j];
}

// LFQuantizer.get_codebook_entry: bit (nbits-1-c) of the index -> +-1 in channel c; NHWC
__global__ void lfq_nhwc_kernel(const int64_t* __restrict__ idx, float* __restrict__ out, long long n, int nbits) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long v = idx[i];
    for (int c = 0; c < nbits; ++c) out[i * nbits + c] = ((v >> (nbits - 1 - c)) & 1) ? 1.0f : -1.0f;
}

// LFQuantizer.get_indices (modeling_magvitv2.py:201-206) of the sign quantisation (:241-243): bit (nbits-1-c) = z_c > 0
__global__ void lfq_index_kernel(const float* __restrict__ z, int64_t* __restrict__ idx, long long n, int nbits) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    long long v = 0;
    for (int c = 0; c < nbits; ++c) v |= (long long)(z[i * nbits + c] > 0.f) << (nbits - 1 - c);
    idx[i] = v;
}

// pixel_values [B, C, HW] -> [B, HW, C]
// ---- diffusers VQModel quantizer (A variant): a learned codebook [n_embed, D] instead of the lookup-free bit code ----
// VectorQuantizer.get_codebook_entry: z_q = embedding(indices).view(B, h, w, D) — which IS the NHWC layout used here
__global__ void codebook_gather_kernel(const int64_t* __restrict__ idx, const float* __restrict__ cb, float* __restrict__ out,
                                       long long n, int D, int n_embed) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * D) return;
    const long long p = i / D;
    long long id = idx[p];
    id = id < 0 ? 0 : (id >= n_embed ? n_embed - 1 : id);  // torch would raise; keep the device safe
    out[i] = cb[id * D + (i - p * D)];
}

// VectorQuantizer.forward: min_encoding_indices = argmin_j cdist(z, E)[., j].  torch.cdist (p = 2, > 25 rows) evaluates
// sqrt(clamp_min(|z|^2 + |e_j|^2 - 2 z·e_j, 0)) through one matmul; this kernel forms the same three terms in fp32 and
// takes the FIRST index of the minimum like torch.argmin.  One workgroup per latent row, codes strided over the threads.
__global__ __launch_bounds__(256) void codebook_argmin_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                              int64_t* __restrict__ idx, int D, int n_embed) {
    extern __shared__ float zrow[];  // D floats, then 256 (dist, index) pairs
    float* sd = zrow + D;
    int* si = (int*)(sd + 256);
    const long long row = blockIdx.x;
    float xn = 0.f;
    for (int c = threadIdx.x; c < D; c += 256) zrow[c] = z[row * D + c];
    __syncthreads();
    for (int c = 0; c < D; ++c) xn = fmaf(zrow[c], zrow[c], xn);
    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int j = threadIdx.x; j < n_embed; j += 256) {
        const float* e = cb + (size_t)j * D;
        float dot = 0.f, yn = 0.f;
        for (int c = 0; c < D; ++c) {
            const float ev = e[c];
            dot = fmaf(-2.0f * zrow[c], ev, dot);
            yn = fmaf(ev, ev, yn);
        }
        const float d = sqrtf(fmaxf(dot + xn + yn, 0.f));
        if (d < best) { best = d; bi = j; }  // j ascending per thread: the first minimum is kept
    }
    sd[threadIdx.x] = best; si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const float d2 = sd[threadIdx.x + o];
            const int i2 = si[threadIdx.x + o];
            if (d2 < sd[threadIdx.x] || (d2 == sd[threadIdx.x] && i2 < si[threadIdx.x])) { sd[threadIdx.x] = d2; si[threadIdx.x] = i2; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) idx[row] = si[0];
}

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int C, long long HW, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long b = i / (HW * C), r = i - b * HW * C, p = r / C;
    const int c = (int)(r - p * C);
    out[i] = in[(b * C + c) * HW + p];
}

// [Cout][Cin][k][k] (nn.Conv2d) -> [Cout][k*k][Cin]
__global__ void repack_conv_kernel(const float* __restrict__ src, float* __restrict__ dst, int co, int ci, int kk) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)co * ci * kk;
    if (i >= total) return;
    const int c = (int)(i % ci);
    const int t = (int)((i / ci) % kk);
    const size_t o = i / ((size_t)ci * kk);
    dst[i] = src[(o * ci + c) * kk + t];
}

int launch_conv(const ConvArgs& g, hipStream_t s) {
    if (g.M <= 0) return 0;
    if (g.taps != 1 && g.taps != 9) return mm_fail("vq conv: taps must be 1 or 9");
    const bool mfma = (g.Cin % CBK == 0) && g.Cout > 16 && !g.nchw_out;
    if (mfma) {
        hipLaunchKernelGGL(conv_mfma_kernel, dim3((unsigned)((g.M + CBM - 1) / CBM), (g.Cout + CBN - 1) / CBN), dim3(256), 0, s, g);
    } else {
        const unsigned gx = (unsigned)((g.M + 255) / 256);
        const size_t wbytes = (size_t)g.Cout * g.taps * g.Cin * sizeof(float);
        if (g.Cout <= THIN_CO && g.Cin % 32 == 0 && wbytes <= 64 * 1024)
            hipLaunchKernelGGL(conv_thin_kernel, dim3((unsigned)((g.M + 31) / 32)), dim3(256), wbytes, s, g);
        else if (g.Cout <= 3)
            hipLaunchKernelGGL(conv_direct_kernel<3>, dim3(gx, 1), dim3(256), 0, s, g);
        else if (g.Cout <= 13)
            hipLaunchKernelGGL(conv_direct_kernel<13>, dim3(gx, 1), dim3(256), 0, s, g);
        else
            hipLaunchKernelGGL(conv_direct_kernel<16>, dim3(gx, (g.Cout + 15) / 16), dim3(256), 0, s, g);
    }
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}

int gn_chunks(int HW) { return std::max(1, std::min(GN_MAX_CHUNKS, HW / 256)); }

int launch_group_norm(const float* x, const float* gamma, const float* beta, float* out, double* partial, int B, int HW,
                      int C, int swish, hipStream_t s) {
    if (C % 128 || C > 1024) return mm_fail("vq group_norm: C=%d must be a multiple of 128 and <= 1024", C);
    const int nch = gn_chunks(HW);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nch, B), dim3(256), 0, s, x, partial, HW, C, nch);
    const size_t total = (size_t)HW * (C / 4);
    const unsigned gx = (unsigned)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(gx, B), dim3(256), 0, s, x, partial, gamma, beta, out, HW, C, nch, swish);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}

struct ConvP {
    float *w = nullptr, *b = nullptr;
    int co = 0, ci = 0, k = 0;
};
struct NormP {
    float *g = nullptr, *b = nullptr;
    int c = 0;
};
struct ResP {
    NormP n1, n2;
    ConvP c1, c2, nin;
    bool has_nin = false;
};
struct Slot {  // one expected state-dict tensor
    float** dst;
    long long numel;
    int co, ci, kk;  // conv weight: repack; otherwise kk = 0
    bool bound;
};

}  // namespace

struct mmada_vq {
    mmada_vq_cfg cfg;
    ConvP post_quant, conv_in, conv_out, aq, ak, av, aproj;
    NormP norm_out, attn_norm;
    ResP mid1, mid2;
    std::vector<std::vector<ResP>> up;  // [level][block]   (decoder: up.*, encoder: down.*)
    std::vector<ConvP> upsample;        // [level] decoder: up.{l}.upsample.conv (l > 0); encoder: down.{l}.downsample.conv
    bool encoder = false;               // encoder: conv_in takes cfg.out_ch image channels, post_quant = quant_conv
    // diffusers VQModel flavour (A variant, mmada_vq_create_vqmodel): learned codebook, optional mid-block attention
    bool vqmodel = false, mid_attn = true;
    float* codebook = nullptr;          // [n_embed, embed_dim]
    int n_embed = 0, embed_dim = 0;
    std::map<std::string, Slot> slots;
    std::vector<float*> owned;
};

namespace {

void reg_conv(mmada_vq* h, const std::string& p, ConvP& c, int co, int ci, int k) {
    c.co = co; c.ci = ci; c.k = k;
    h->slots[p + ".weight"] = Slot{&c.w, (long long)co * ci * k * k, co, ci, k * k, false};
    h->slots[p + ".bias"] = Slot{&c.b, co, 0, 0, 0, false};
}
void reg_norm(mmada_vq* h, const std::string& p, NormP& n, int c) {
    n.c = c;
    h->slots[p + ".weight"] = Slot{&n.g, c, 0, 0, 0, false};
    h->slots[p + ".bias"] = Slot{&n.b, c, 0, 0, 0, false};
}
void reg_res(mmada_vq* h, const std::string& p, ResP& r, int ci, int co, const char* shortcut = "nin_shortcut") {
    reg_norm(h, p + ".norm1", r.n1, ci);
    reg_conv(h, p + ".conv1", r.c1, co, ci, 3);
    reg_norm(h, p + ".norm2", r.n2, co);
    reg_conv(h, p + ".conv2", r.c2, co, co, 3);
    r.has_nin = ci != co;
    if (r.has_nin) reg_conv(h, p + "." + shortcut, r.nin, co, ci, 1);
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Plan {
    size_t act_bytes;   // one activation buffer (largest [B, H, W, C] of the network)
    size_t attn_bytes;  // q, k, v, v^T ([T, C] each) + S [T, T], per batch element handled one at a time
    size_t gn_bytes;
    size_t total;
};

Plan plan_for(const mmada_vq* h, int B, int hz, int wz) {
    const mmada_vq_cfg& c = h->cfg;
    // largest [H, W, C] of the network: level l lives at hz * 2^(L-1-l); a tensor at that resolution has the channel
    // count of level l or of a neighbouring level (first block of a level / tensor just after a resample)
    size_t max_elems = 0;
    for (int lvl = 0; lvl < c.n_levels; ++lvl) {
        int m = c.ch_mult[lvl];
        if (lvl > 0) m = std::max(m, c.ch_mult[lvl - 1]);
        if (lvl + 1 < c.n_levels) m = std::max(m, c.ch_mult[lvl + 1]);
        const size_t f = (size_t)1 << (c.n_levels - 1 - lvl);
        max_elems = std::max(max_elems, (size_t)hz * f * wz * f * c.ch * m);
    }
    Plan p;
    p.act_bytes = align256(max_elems * B * sizeof(float));
    const size_t T = (size_t)hz * wz, C = (size_t)c.ch * c.ch_mult[c.n_levels - 1];
    p.attn_bytes = align256((4 * T * C * B + T * T) * sizeof(float));
    p.gn_bytes = align256((size_t)B * GN_GROUPS * GN_MAX_CHUNKS * 2 * sizeof(double));
    p.total = 3 * p.act_bytes + p.attn_bytes + p.gn_bytes;
    return p;
}

struct Runner {
    hipStream_t s;
    double* gn;
    int B;
    // resample: 0 same size, 1 = 2x nearest upsample folded in front, -1 = Downsample (pad right/bottom, stride 2)
    int conv(const ConvP& p, const float* in, float* out, const float* resid, int Hi, int Wi, int resample, int nchw = 0) {
        ConvArgs g{};
        g.in = in; g.w = p.w; g.bias = p.b; g.resid = resid; g.out = out;
        g.B = B; g.Hi = Hi; g.Wi = Wi; g.Cin = p.ci; g.Cout = p.co;
        g.ups = resample > 0; g.down = resample < 0;
        g.Ho = g.down ? (Hi - 2) / 2 + 1 : Hi << g.ups;
        g.Wo = g.down ? (Wi - 2) / 2 + 1 : Wi << g.ups;
        g.taps = p.k * p.k; g.nchw_out = nchw;
        g.M = (long long)B * g.Ho * g.Wo;
        return launch_conv(g, s);
    }
    // AttnBlock.forward (common_modules.py:187-211): x += proj_out(softmax(q k^T / sqrt(C)) v), single head over H*W
    int attn(const mmada_vq* h, float* x, float* t1, float* scratch, int H, int W) {
        const int T = H * W, C = h->attn_norm.c;
        float* q = scratch;
        float* k = q + (size_t)B * T * C;
        float* v = k + (size_t)B * T * C;
        float* vt = v + (size_t)B * T * C;   // one batch element at a time: [C, T]
        float* S = vt + (size_t)B * T * C;   // [T, T]
        if (norm(h->attn_norm, x, t1, T, 0)) return 1;
        if (conv(h->aq, t1, q, nullptr, H, W, 0)) return 1;
        if (conv(h->ak, t1, k, nullptr, H, W, 0)) return 1;
        if (conv(h->av, t1, v, nullptr, H, W, 0)) return 1;
        for (int b = 0; b < B; ++b) {
            ConvArgs g{};
            g.in = q + (size_t)b * T * C; g.w = k + (size_t)b * T * C; g.out = S;
            g.B = 1; g.Hi = g.Ho = T; g.Wi = g.Wo = 1; g.Cin = C; g.Cout = T; g.taps = 1; g.M = T;
            if (launch_conv(g, s)) return 1;  // S[i][j] = sum_c q[i][c] k[j][c]
            hipLaunchKernelGGL(softmax_rows_kernel, dim3(T), dim3(256), 0, s, S, T, 1.0f / sqrtf((float)C));
            hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (T + 31) / 32), dim3(256), 0, s,
                               v + (size_t)b * T * C, vt, T, C);
            MM_CHECK_HIP(hipGetLastError());
            ConvArgs o{};
            o.in = S; o.w = vt; o.out = t1 + (size_t)b * T * C;
            o.B = 1; o.Hi = o.Ho = T; o.Wi = o.Wo = 1; o.Cin = T; o.Cout = C; o.taps = 1; o.M = T;
            if (launch_conv(o, s)) return 1;  // h_[i][c] = sum_j softmax(S)[i][j] v[j][c]
        }
        return conv(h->aproj, t1, x, x, H, W, 0);
    }
    int norm(const NormP& n, const float* in, float* out, int HW, int swish) {
        return launch_group_norm(in, n.g, n.b, out, gn, B, HW, n.c, swish, s);
    }
    // common_modules.py:337-357; x is updated in place (its channel count becomes r.c2.co)
    int res(const ResP& r, float* x, float* t1, float* t2, int H, int W) {
        if (norm(r.n1, x, t1, H * W, 1)) return 1;
        if (conv(r.c1, t1, t2, nullptr, H, W, 0)) return 1;
        if (norm(r.n2, t2, t1, H * W, 1)) return 1;
        if (r.has_nin) {
            if (conv(r.nin, x, t2, nullptr, H, W, 0)) return 1;
            return conv(r.c2, t1, x, t2, H, W, 0);
        }
        return conv(r.c2, t1, x, x, H, W, 0);
    }
};

int launch_nearest_code(const mmada_vq* h, const float* z, long long n, int64_t* idx, hipStream_t s) {
    const size_t lds = (size_t)h->embed_dim * sizeof(float) + 256 * (sizeof(float) + sizeof(int));
    hipLaunchKernelGGL(codebook_argmin_kernel, dim3((unsigned)n), dim3(256), lds, s, z, h->codebook, idx, h->embed_dim, h->n_embed);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" {

static int vq_check_cfg(const mmada_vq_cfg* cfg, mmada_vq** out) {
    if (!cfg || !out) return mm_fail("mmada_vq_create: null argument");
    if (cfg->n_levels < 1 || cfg->n_levels > 8) return mm_fail("mmada_vq_create: n_levels must be 1..8");
    if (cfg->ch <= 0 || cfg->ch % 128) return mm_fail("mmada_vq_create: ch must be a positive multiple of 128 (GroupNorm(32) over float4 columns)");
    if (cfg->z_channels <= 0 || cfg->z_channels > 62 || cfg->out_ch <= 0 || cfg->out_ch > 16)
        return mm_fail("mmada_vq_create: bad z_channels / out_ch");
    for (int i = 0; i < cfg->n_levels; ++i)
        if (cfg->ch_mult[i] <= 0 || cfg->num_res_blocks[i] <= 0 || cfg->ch * cfg->ch_mult[i] > 1024)
            return mm_fail("mmada_vq_create: bad ch_mult / num_res_blocks at level %d", i);
    return 0;
}

int mmada_vq_create(const mmada_vq_cfg* cfg, mmada_vq** out) {
    if (vq_check_cfg(cfg, out)) return 1;
    mmada_vq* h = new mmada_vq();
    h->cfg = *cfg;
    const int L = cfg->n_levels;
    int block_in = cfg->ch * cfg->ch_mult[L - 1];
    reg_conv(h, "post_quant_conv", h->post_quant, cfg->z_channels, cfg->z_channels, 1);
    reg_conv(h, "conv_in", h->conv_in, block_in, cfg->z_channels, 3);
    reg_res(h, "mid.block_1", h->mid1, block_in, block_in);
    reg_norm(h, "mid.attn_1.norm", h->attn_norm, block_in);
    reg_conv(h, "mid.attn_1.q", h->aq, block_in, block_in, 1);
    reg_conv(h, "mid.attn_1.k", h->ak, block_in, block_in, 1);
    reg_conv(h, "mid.attn_1.v", h->av, block_in, block_in, 1);
    reg_conv(h, "mid.attn_1.proj_out", h->aproj, block_in, block_in, 1);
    reg_res(h, "mid.block_2", h->mid2, block_in, block_in);
    h->up.resize(L);
    h->upsample.resize(L);
    for (int lvl = L - 1; lvl >= 0; --lvl) {
        const int block_out = cfg->ch * cfg->ch_mult[lvl];
        h->up[lvl].resize(cfg->num_res_blocks[lvl]);
        for (int b = 0; b < cfg->num_res_blocks[lvl]; ++b) {
            reg_res(h, "up." + std::to_string(lvl) + ".block." + std::to_string(b), h->up[lvl][b], block_in, block_out);
            block_in = block_out;
        }
        if (lvl != 0) reg_conv(h, "up." + std::to_string(lvl) + ".upsample.conv", h->upsample[lvl], block_in, block_in, 3);
    }
    reg_norm(h, "norm_out", h->norm_out, block_in);
    reg_conv(h, "conv_out", h->conv_out, cfg->out_ch, block_in, 3);
    *out = h;
    return 0;
}

/* VQGANEncoder.__init__ (modeling_magvitv2.py:62-141); cfg->out_ch carries in_ch (image channels) */
int mmada_vq_create_encoder(const mmada_vq_cfg* cfg, mmada_vq** out) {
    if (vq_check_cfg(cfg, out)) return 1;
    mmada_vq* h = new mmada_vq();
    h->cfg = *cfg;
    h->encoder = true;
    const int L = cfg->n_levels;
    reg_conv(h, "conv_in", h->conv_in, cfg->ch, cfg->out_ch, 3);
    h->up.resize(L);
    h->upsample.resize(L);
    int block_in = cfg->ch;
    for (int lvl = 0; lvl < L; ++lvl) {
        const int block_out = cfg->ch * cfg->ch_mult[lvl];
        h->up[lvl].resize(cfg->num_res_blocks[lvl]);
        for (int b = 0; b < cfg->num_res_blocks[lvl]; ++b) {
            reg_res(h, "down." + std::to_string(lvl) + ".block." + std::to_string(b), h->up[lvl][b], block_in, block_out);
            block_in = block_out;
        }
        if (lvl != L - 1) reg_conv(h, "down." + std::to_string(lvl) + ".downsample.conv", h->upsample[lvl], block_in, block_in, 3);
    }
    reg_res(h, "mid.block_1", h->mid1, block_in, block_in);
    reg_norm(h, "mid.attn_1.norm", h->attn_norm, block_in);
    reg_conv(h, "mid.attn_1.q", h->aq, block_in, block_in, 1);
    reg_conv(h, "mid.attn_1.k", h->ak, block_in, block_in, 1);
    reg_conv(h, "mid.attn_1.v", h->av, block_in, block_in, 1);
    reg_conv(h, "mid.attn_1.proj_out", h->aproj, block_in, block_in, 1);
    reg_res(h, "mid.block_2", h->mid2, block_in, block_in);
    reg_norm(h, "norm_out", h->norm_out, block_in);
    reg_conv(h, "conv_out", h->conv_out, cfg->z_channels, block_in, 3);
    reg_conv(h, "quant_conv", h->post_quant, cfg->z_channels, cfg->z_channels, 1);
    *out = h;
    return 0;
}

void mmada_vq_destroy(mmada_vq* h) {
    if (!h) return;
    for (float* p : h->owned) (void)hipFree(p);
    delete h;
}

int mmada_vq_bind(mmada_vq* h, const char* name, const float* data, int64_t numel, void* stream) {
    if (!h || !name || !data) return mm_fail("mmada_vq_bind: null argument");
    std::string key(name);
    const std::string prefix = h->encoder ? "encoder." : "decoder.";
    if (key.rfind(prefix, 0) == 0) key = key.substr(prefix.size());
    auto it = h->slots.find(key);
    if (it == h->slots.end()) return mm_fail("mmada_vq_bind: unexpected tensor '%s'", name);
    Slot& sl = it->second;
    if (numel != sl.numel) return mm_fail("mmada_vq_bind: '%s' has %lld elements, expected %lld", name, (long long)numel, sl.numel);
    hipStream_t s = (hipStream_t)stream;
    if (!*sl.dst) {
        float* p = nullptr;
        MM_CHECK_HIP(hipMalloc(&p, (size_t)numel * sizeof(float)));
        h->owned.push_back(p);
        *sl.dst = p;
    }
    if (sl.kk > 1) {
        hipLaunchKernelGGL(repack_conv_kernel, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, s, data, *sl.dst, sl.co, sl.ci, sl.kk);
        MM_CHECK_HIP(hipGetLastError());
    } else {
        MM_CHECK_HIP(hipMemcpyAsync(*sl.dst, data, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    sl.bound = true;
    return 0;
}

int mmada_vq_num_unbound(const mmada_vq* h) {
    if (!h) return -1;
    int n = 0;
    for (const auto& kv : h->slots) n += kv.second.bound ? 0 : 1;
    return n;
}

size_t mmada_vq_workspace_bytes(const mmada_vq* h, int B, int hz, int wz) {
    if (!h || B <= 0 || hz <= 0 || wz <= 0) return 0;
    return plan_for(h, B, hz, wz).total;  // encoder: the same buffers (largest activation is at the image resolution)
}

int mmada_vq_decode_code(mmada_vq* h, const int64_t* indices, int B, int hz, int wz, void* workspace,
                         size_t workspace_bytes, float* out, void* stream) {
    if (!h || !indices || !workspace || !out) return mm_fail("mmada_vq_decode_code: null argument");
    if (h->encoder) return mm_fail("mmada_vq_decode_code: this handle is an encoder");
    if (B <= 0 || hz <= 0 || wz <= 0) return mm_fail("mmada_vq_decode_code: bad shape");
    if (h->mid_attn && (hz * wz) % 32) return mm_fail("mmada_vq_decode_code: hz*wz must be a multiple of 32 (attention K tiles)");
    for (const auto& kv : h->slots)
        if (!kv.second.bound) return mm_fail("mmada_vq_decode_code: tensor '%s' was never bound", kv.first.c_str());
    const Plan pl = plan_for(h, B, hz, wz);
    if (workspace_bytes < pl.total) return mm_fail("mmada_vq_decode_code: workspace too small (%zu < %zu)", workspace_bytes, pl.total);
    if ((uintptr_t)workspace & 255) return mm_fail("mmada_vq_decode_code: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float* x = (float*)ws;
    float* t1 = (float*)(ws + pl.act_bytes);
    float* t2 = (float*)(ws + 2 * pl.act_bytes);
    float* attn = (float*)(ws + 3 * pl.act_bytes);
    Runner r{s, (double*)(ws + 3 * pl.act_bytes + pl.attn_bytes), B};
    const mmada_vq_cfg& c = h->cfg;
    int H = hz, W = wz;
    const long long npix = (long long)B * H * W;

    // get_codebook_entry (:208-221) -> post_quant_conv -> conv_in (:374-377)
    if (h->vqmodel) {  // VectorQuantizer.get_codebook_entry: rows of the learned codebook, already NHWC
        const long long tot = npix * h->embed_dim;
        hipLaunchKernelGGL(codebook_gather_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, indices, h->codebook, t1,
                           npix, h->embed_dim, h->n_embed);
    } else {
        hipLaunchKernelGGL(lfq_nhwc_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, indices, t1, npix, c.z_channels);
    }
    MM_CHECK_HIP(hipGetLastError());
    if (r.conv(h->post_quant, t1, t2, nullptr, H, W, 0)) return 1;
    if (r.conv(h->conv_in, t2, x, nullptr, H, W, 0)) return 1;
    // middle (:380-382)
    if (r.res(h->mid1, x, t1, t2, H, W)) return 1;
    if (h->mid_attn && r.attn(h, x, t1, attn, H, W)) return 1;
    if (r.res(h->mid2, x, t1, t2, H, W)) return 1;
    // upsampling (:385-391)
    for (int lvl = c.n_levels - 1; lvl >= 0; --lvl) {
        for (const ResP& rb : h->up[lvl])
            if (r.res(rb, x, t1, t2, H, W)) return 1;
        if (lvl != 0) {
            if (r.conv(h->upsample[lvl], x, t1, nullptr, H, W, 1)) return 1;
            std::swap(x, t1);
            H *= 2; W *= 2;
        }
    }
    // end (:398-400); output NCHW like the reference
    if (r.norm(h->norm_out, x, t1, H * W, 1)) return 1;
    return r.conv(h->conv_out, t1, out, nullptr, H, W, 0, 1);
}

/* MAGVITv2.get_code (modeling_magvitv2.py:422-427): VQGANEncoder.forward (:143-171) -> sign quantisation -> indices */
int mmada_vq_get_code(mmada_vq* h, const float* pixel_values, int B, int H, int W, void* workspace,
                      size_t workspace_bytes, int64_t* indices_out, float* z_out, void* stream) {
    if (!h || !pixel_values || !workspace || !indices_out) return mm_fail("mmada_vq_get_code: null argument");
    if (!h->encoder) return mm_fail("mmada_vq_get_code: this handle is a decoder");
    const mmada_vq_cfg& c = h->cfg;
    const int f = 1 << (c.n_levels - 1);
    if (B <= 0 || H <= 0 || W <= 0 || H % f || W % f) return mm_fail("mmada_vq_get_code: H, W must be multiples of %d", f);
    const int hz = H / f, wz = W / f;
    if (h->mid_attn && (hz * wz) % 32) return mm_fail("mmada_vq_get_code: (H/%d)*(W/%d) must be a multiple of 32 (attention K tiles)", f, f);
    for (const auto& kv : h->slots)
        if (!kv.second.bound) return mm_fail("mmada_vq_get_code: tensor '%s' was never bound", kv.first.c_str());
    const Plan pl = plan_for(h, B, hz, wz);
    if (workspace_bytes < pl.total) return mm_fail("mmada_vq_get_code: workspace too small (%zu < %zu)", workspace_bytes, pl.total);
    if ((uintptr_t)workspace & 255) return mm_fail("mmada_vq_get_code: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float* x = (float*)ws;
    float* t1 = (float*)(ws + pl.act_bytes);
    float* t2 = (float*)(ws + 2 * pl.act_bytes);
    float* attn = (float*)(ws + 3 * pl.act_bytes);
    Runner r{s, (double*)(ws + 3 * pl.act_bytes + pl.attn_bytes), B};
    const long long total = (long long)B * H * W * c.out_ch;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pixel_values, t1,
                       c.out_ch, (long long)H * W, total);
    MM_CHECK_HIP(hipGetLastError());
    if (r.conv(h->conv_in, t1, x, nullptr, H, W, 0)) return 1;
    int Hc = H, Wc = W;
    for (int lvl = 0; lvl < c.n_levels; ++lvl) {  // downsampling (:148-156); hs[-1] is always the running tensor
        for (const ResP& rb : h->up[lvl])
            if (r.res(rb, x, t1, t2, Hc, Wc)) return 1;
        if (lvl != c.n_levels - 1) {
            if (r.conv(h->upsample[lvl], x, t1, nullptr, Hc, Wc, -1)) return 1;
            std::swap(x, t1);
            Hc /= 2; Wc /= 2;
        }
    }
    if (r.res(h->mid1, x, t1, t2, Hc, Wc)) return 1;  // middle (:159-162)
    if (h->mid_attn && r.attn(h, x, t1, attn, Hc, Wc)) return 1;
    if (r.res(h->mid2, x, t1, t2, Hc, Wc)) return 1;
    if (r.norm(h->norm_out, x, t1, Hc * Wc, 1)) return 1;  // end (:165-169)
    if (r.conv(h->conv_out, t1, t2, nullptr, Hc, Wc, 0)) return 1;
    if (r.conv(h->post_quant, t2, t1, nullptr, Hc, Wc, 0)) return 1;  // quant_conv
    const long long npix = (long long)B * Hc * Wc;
    if (h->vqmodel) {  // VQModel.encode -> latents [npix, embed_dim]; VectorQuantizer: nearest codebook row
        if (launch_nearest_code(h, t1, npix, indices_out, s)) return 1;
        if (z_out) MM_CHECK_HIP(hipMemcpyAsync(z_out, t1, (size_t)npix * h->embed_dim * sizeof(float), hipMemcpyDeviceToDevice, s));
        return 0;
    }
    hipLaunchKernelGGL(lfq_index_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, t1, indices_out, npix, c.z_channels);
    MM_CHECK_HIP(hipGetLastError());
    if (z_out) MM_CHECK_HIP(hipMemcpyAsync(z_out, t1, (size_t)npix * c.z_channels * sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
}

/* ---- diffusers VQModel (A variant) ------------------------------------------------------------------------------------ */
int mmada_vq_create_vqmodel(const mmada_vqmodel_cfg* cfg, int encoder, mmada_vq** out) {
    if (!cfg || !out) return mm_fail("mmada_vq_create_vqmodel: null argument");
    const int L = cfg->n_levels;
    if (L < 1 || L > 8) return mm_fail("mmada_vq_create_vqmodel: 1..8 blocks, got %d", L);
    if (cfg->norm_num_groups != GN_GROUPS) return mm_fail("mmada_vq_create_vqmodel: norm_num_groups must be %d", GN_GROUPS);
    if (cfg->layers_per_block < 1 || cfg->layers_per_block > 16) return mm_fail("mmada_vq_create_vqmodel: bad layers_per_block");
    if (cfg->latent_channels <= 0 || cfg->latent_channels > 1024 || cfg->vq_embed_dim <= 0 || cfg->vq_embed_dim > 1024 ||
        cfg->num_vq_embeddings <= 0 || cfg->image_channels <= 0 || cfg->image_channels > 16)
        return mm_fail("mmada_vq_create_vqmodel: bad latent_channels / vq_embed_dim / num_vq_embeddings / image_channels");
    for (int i = 0; i < L; ++i)
        if (cfg->block_out_channels[i] <= 0 || cfg->block_out_channels[i] % 128 || cfg->block_out_channels[i] > 1024)
            return mm_fail("mmada_vq_create_vqmodel: block_out_channels[%d]=%d must be a multiple of 128 up to 1024 "
                           "(GroupNorm(32) over float4 columns)", i, cfg->block_out_channels[i]);
    mmada_vq* h = new mmada_vq();
    h->vqmodel = true;
    h->encoder = encoder != 0;
    h->mid_attn = cfg->mid_block_add_attention != 0;
    h->n_embed = cfg->num_vq_embeddings;
    h->embed_dim = cfg->vq_embed_dim;
    // the shared runner sizes its buffers from (ch, ch_mult): level l of the taming numbering = diffusers block l
    h->cfg.ch = 128;
    h->cfg.n_levels = L;
    for (int i = 0; i < L; ++i) {
        h->cfg.ch_mult[i] = cfg->block_out_channels[i] / 128;
        h->cfg.num_res_blocks[i] = cfg->layers_per_block + (encoder ? 0 : 1);
    }
    h->cfg.z_channels = cfg->latent_channels;
    h->cfg.out_ch = cfg->image_channels;
    h->slots["quantize.embedding.weight"] = Slot{&h->codebook, (long long)h->n_embed * h->embed_dim, 0, 0, 0, false};
    h->up.resize(L);
    h->upsample.resize(L);
    auto reg_mid = [&](int C) {
        reg_res(h, "mid_block.resnets.0", h->mid1, C, C, "conv_shortcut");
        if (h->mid_attn) {  // diffusers Attention with one head: GroupNorm, Linear q / k / v / out (= 1x1 convolutions), residual
            reg_norm(h, "mid_block.attentions.0.group_norm", h->attn_norm, C);
            reg_conv(h, "mid_block.attentions.0.to_q", h->aq, C, C, 1);
            reg_conv(h, "mid_block.attentions.0.to_k", h->ak, C, C, 1);
            reg_conv(h, "mid_block.attentions.0.to_v", h->av, C, C, 1);
            reg_conv(h, "mid_block.attentions.0.to_out.0", h->aproj, C, C, 1);
        } else {
            h->attn_norm.c = C;
        }
        reg_res(h, "mid_block.resnets.1", h->mid2, C, C, "conv_shortcut");
    };
    if (!encoder) {  // Decoder (autoencoders/vae.py): conv_in, mid_block, up_blocks (lowest resolution first), norm, conv_out
        reg_conv(h, "post_quant_conv", h->post_quant, cfg->latent_channels, cfg->vq_embed_dim, 1);
        int block_in = cfg->block_out_channels[L - 1];
        reg_conv(h, "conv_in", h->conv_in, block_in, cfg->latent_channels, 3);
        reg_mid(block_in);
        for (int i = 0; i < L; ++i) {
            const int lvl = L - 1 - i, block_out = cfg->block_out_channels[lvl];
            h->up[lvl].resize(cfg->layers_per_block + 1);
            for (int j = 0; j <= cfg->layers_per_block; ++j) {
                reg_res(h, "up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), h->up[lvl][j], block_in, block_out,
                        "conv_shortcut");
                block_in = block_out;
            }
            if (i != L - 1)
                reg_conv(h, "up_blocks." + std::to_string(i) + ".upsamplers.0.conv", h->upsample[lvl], block_in, block_in, 3);
        }
        reg_norm(h, "conv_norm_out", h->norm_out, block_in);
        reg_conv(h, "conv_out", h->conv_out, cfg->image_channels, block_in, 3);
    } else {  // Encoder: conv_in, down_blocks, mid_block, norm, conv_out; then VQModel.quant_conv
        int block_in = cfg->block_out_channels[0];
        reg_conv(h, "conv_in", h->conv_in, block_in, cfg->image_channels, 3);
        for (int i = 0; i < L; ++i) {
            const int block_out = cfg->block_out_channels[i];
            h->up[i].resize(cfg->layers_per_block);
            for (int j = 0; j < cfg->layers_per_block; ++j) {
                reg_res(h, "down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), h->up[i][j], block_in, block_out,
                        "conv_shortcut");
                block_in = block_out;
            }
            if (i != L - 1)
                reg_conv(h, "down_blocks." + std::to_string(i) + ".downsamplers.0.conv", h->upsample[i], block_in, block_in, 3);
        }
        reg_mid(block_in);
        reg_norm(h, "conv_norm_out", h->norm_out, block_in);
        reg_conv(h, "conv_out", h->conv_out, cfg->latent_channels, block_in, 3);
        reg_conv(h, "quant_conv", h->post_quant, cfg->vq_embed_dim, cfg->latent_channels, 1);
    }
    *out = h;
    return 0;
}

int mmada_vq_nearest_code(mmada_vq* h, const float* z_nhwc, int64_t n, int64_t* indices_out, void* stream) {
    if (!h || !z_nhwc || !indices_out) return mm_fail("mmada_vq_nearest_code: null argument");
    if (!h->vqmodel || !h->codebook) return mm_fail("mmada_vq_nearest_code: needs a VQModel handle with its codebook bound");
    if (n <= 0) return 0;
    return launch_nearest_code(h, z_nhwc, n, indices_out, (hipStream_t)stream);
}

/* kernel-level entry points (parity tests) */
int mmada_vq_conv2d(const float* in_nhwc, const float* w_packed, const float* bias, const float* resid, float* out,
                    int B, int Hi, int Wi, int Cin, int Cout, int ksize, int upsample, void* stream) {
    if (!in_nhwc || !w_packed || !out) return mm_fail("mmada_vq_conv2d: null argument");
    if (ksize != 1 && ksize != 3) return mm_fail("mmada_vq_conv2d: ksize must be 1 or 3");
    ConvArgs g{};
    g.in = in_nhwc; g.w = w_packed; g.bias = bias; g.resid = resid; g.out = out;
    g.B = B; g.Hi = Hi; g.Wi = Wi; g.Cin = Cin; g.Cout = Cout; g.ups = upsample > 0; g.down = upsample < 0;
    if (g.down && ksize != 3) return mm_fail("mmada_vq_conv2d: the stride-2 mode is 3x3 only");
    g.Ho = g.down ? (Hi - 2) / 2 + 1 : Hi << g.ups;
    g.Wo = g.down ? (Wi - 2) / 2 + 1 : Wi << g.ups;
    g.taps = ksize * ksize;
    g.M = (long long)B * g.Ho * g.Wo;
    return launch_conv(g, (hipStream_t)stream);
}

int mmada_vq_group_norm(const float* x_nhwc, const float* gamma, const float* beta, float* out, void* scratch,
                        int B, int HW, int C, int swish, void* stream) {
    if (!x_nhwc || !gamma || !beta || !out || !scratch) return mm_fail("mmada_vq_group_norm: null argument");
    return launch_group_norm(x_nhwc, gamma, beta, out, (double*)scratch, B, HW, C, swish, (hipStream_t)stream);
}

size_t mmada_vq_group_norm_scratch_bytes(int B) { return (size_t)B * GN_GROUPS * GN_MAX_CHUNKS * 2 * sizeof(double); }

}  // extern "C"
