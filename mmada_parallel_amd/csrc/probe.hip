// probe.hip — the attainable-MFMA probe behind mmada_mfma_probe (bench.py's `roofline.attainable_tflops`).
//
// Measurement only; nothing of the reference corresponds to it.  The dense bf16 peak of gfx950 (2.5 PFLOP/s) assumes the
// 2.4 GHz engine clock; under a sustained MFMA load on RANDOM operands the part runs at its package power limit and
// clocks lower (guide: DVFS give-back; zero-filled operands run ~20 % faster at equal cycles).  This kernel is what a
// GEMM would be with every memory, LDS and barrier cost removed: eight waves per CU (two per SIMD, the occupancy of the
// 8-phase GEMM) issue nothing but v_mfma_f32_16x16x32_bf16 — the production instruction — on operand fragments of
// random bf16 data held in registers, 16 independent accumulators per wave.  Its rate, measured right after the timed
// region of a bench run while the chip is still warm, is the roof a real kernel could at best approach on this part at
// that moment; `roofline.frac` stays quoted against the 2.5 PFLOP/s datasheet peak.
#include "../../include/mmada_mi355x.h"
#include "kernels.h"

namespace {

constexpr int PROBE_WAVES = 8, PROBE_FRAGS = 4, MFMA_PER_ITER = PROBE_FRAGS * PROBE_FRAGS * 2;

__global__ __launch_bounds__(PROBE_WAVES * 64, 2) void mfma_probe_kernel(const bf16_t* data, int iters, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // 16 fragments (8 per operand) of this wave: 8 KiB of random bf16 per wave, different for every wave of the workgroup
    const bf16x8* src = (const bf16x8*)data + (size_t)((blockIdx.x % 64) * PROBE_WAVES + wave) * 16 * 64 + lane;
    bf16x8 a[2][PROBE_FRAGS], b[2][PROBE_FRAGS];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < PROBE_FRAGS; ++i) {
            a[s][i] = src[(s * 8 + i) * 64];
            b[s][i] = src[(s * 8 + 4 + i) * 64];
        }
    f32x4 acc[PROBE_FRAGS][PROBE_FRAGS];
#pragma unroll
    for (int i = 0; i < PROBE_FRAGS; ++i)
#pragma unroll
        for (int j = 0; j < PROBE_FRAGS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 2; ++s)  // two operand sets alternate, so consecutive MFMAs of a pipe see changing inputs
#pragma unroll
            for (int i = 0; i < PROBE_FRAGS; ++i)
#pragma unroll
                for (int j = 0; j < PROBE_FRAGS; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < PROBE_FRAGS; ++i)
#pragma unroll
        for (int j = 0; j < PROBE_FRAGS; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 123.456f) sink[0] = t;  // keeps the accumulators live; practically never true
}


// Variants (mmada_set_option("probe_variant", v); measurement only): v & 1 = v_mfma_f32_32x32x16_bf16 instead of the
// production 16x16x32 (twice the flops per instruction, half the operand-register reads per flop: does the part hold a higher
// clock on it at its power limit?), v & 2 = four waves per CU (one per SIMD) instead of eight.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void mfma_probe32_kernel(const bf16_t* data, int iters, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bf16x8* src = (const bf16x8*)data + (size_t)((blockIdx.x % 64) * PROBE_WAVES + wave) * 16 * 64 + lane;
    bf16x8 a[2][2], b[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < 2; ++i) a[s][i] = src[(s * 8 + i) * 64];
#pragma unroll
        for (int i = 0; i < 4; ++i) b[s][i] = src[(s * 8 + 4 + i) * 64];
    }
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    if (t == 123.456f) sink[0] = t;
}

__global__ __launch_bounds__(4 * 64, 2) void mfma_probe16w4_kernel(const bf16_t* data, int iters, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bf16x8* src = (const bf16x8*)data + (size_t)((blockIdx.x % 64) * PROBE_WAVES + wave) * 16 * 64 + lane;
    bf16x8 a[2][PROBE_FRAGS], b[2][PROBE_FRAGS];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < PROBE_FRAGS; ++i) {
            a[s][i] = src[(s * 8 + i) * 64];
            b[s][i] = src[(s * 8 + 4 + i) * 64];
        }
    f32x4 acc[PROBE_FRAGS][PROBE_FRAGS];
#pragma unroll
    for (int i = 0; i < PROBE_FRAGS; ++i)
#pragma unroll
        for (int j = 0; j < PROBE_FRAGS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < PROBE_FRAGS; ++i)
#pragma unroll
                for (int j = 0; j < PROBE_FRAGS; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < PROBE_FRAGS; ++i)
#pragma unroll
        for (int j = 0; j < PROBE_FRAGS; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 123.456f) sink[0] = t;
}

}  // namespace

// data: >= 64 * 8 * 16 * 64 * 16 bytes (8 MiB) of bf16 values the caller filled (random: never zeros — see above).
// Runs `launches` back-to-back launches of `iters` iterations on `stream` and returns the achieved dense TFLOP/s.
static std::atomic<int> g_probe_variant{0};
void mfma_probe_set_variant(int v) { g_probe_variant = v & 3; }

int launch_mfma_probe(const bf16_t* data, float* sink, int iters, int launches, hipStream_t s, double* tflops_out, double* ms_out) {
    typedef void (*probe_fn)(const bf16_t*, int, float*);
    const int v = g_probe_variant;
    const probe_fn fn = v == 0 ? (probe_fn)mfma_probe_kernel : v == 1 ? (probe_fn)mfma_probe32_kernel<8>
                      : v == 2 ? (probe_fn)mfma_probe16w4_kernel : (probe_fn)mfma_probe32_kernel<4>;
    const int waves = (v & 2) ? 4 : PROBE_WAVES;
    const double flops_per_iter = (v & 1) ? 16 * (2.0 * 32 * 32 * 16) : MFMA_PER_ITER * (2.0 * 16 * 16 * 32);
    int dev = 0, cus = 0;
    MM_CHECK_HIP(hipGetDevice(&dev));
    MM_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    hipEvent_t e0, e1;
    MM_CHECK_HIP(hipEventCreate(&e0));
    MM_CHECK_HIP(hipEventCreate(&e1));
    hipLaunchKernelGGL(fn, dim3(cus), dim3(waves * 64), 0, s, data, 64, sink);  // warm (code, clocks)
    MM_CHECK_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < launches; ++i)
        hipLaunchKernelGGL(fn, dim3(cus), dim3(waves * 64), 0, s, data, iters, sink);
    MM_CHECK_HIP(hipEventRecord(e1, s));
    MM_CHECK_HIP(hipGetLastError());
    MM_CHECK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    MM_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    const double flops = (double)launches * cus * waves * (double)iters * flops_per_iter;
    if (tflops_out) *tflops_out = flops / (ms * 1e-3) / 1e12;
    if (ms_out) *ms_out = ms;
    return 0;
}

// ---- rounding probe: out[i] = f2bf(in[i]) (tests: every fp32 bit pattern against torch) --------------------------------
namespace {
__global__ __launch_bounds__(256) void f2bf_probe_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = f2bf(in[i]);
}
}  // namespace
int launch_f2bf_probe(const float* in, bf16_t* out, long long n, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(f2bf_probe_kernel, dim3(8192), dim3(256), 0, s, in, out, n);
    MM_CHECK_HIP(hipGetLastError());
    return 0;
}


// ---- which CUs a stream's CU mask selects (round 5: the tensor-parallel exchange's CU partition, tp_comm.hip) -------------------
// Every workgroup records where it ran: out[b] = XCC_ID | HW_ID << 8 (HW_REG_XCC_ID bits 3:0; HW_REG_HW_ID: cu_id 11:8, sh_id 12,
// se_id 15:13).  Launched on a stream created with the given mask; tools/cu_mask_probe.py walks the mask bits.
__global__ void where_kernel(uint32_t* out) {
    if (threadIdx.x == 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));     // XCC_ID[3:0]
        const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_ID
        out[blockIdx.x] = (xcc & 15u) | (hw << 8);
    }
    __builtin_amdgcn_s_sleep(60);   // stay resident a little: the dispatcher spreads the grid over every allowed CU
}

extern "C" int mmada_probe_cu_mask(const uint32_t* mask, int words, uint32_t* out_host, int n_blocks) {
    if (!mask || !out_host || words <= 0 || n_blocks <= 0) return mm_fail("mmada_probe_cu_mask: bad argument");
    hipStream_t st = nullptr;
    MM_CHECK_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask));
    uint32_t* dev = nullptr;
    MM_CHECK_HIP(hipMalloc(&dev, (size_t)n_blocks * 4));
    hipLaunchKernelGGL(where_kernel, dim3(n_blocks), dim3(256), 0, st, dev);
    hipError_t e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipMemcpy(out_host, dev, (size_t)n_blocks * 4, hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    (void)hipStreamDestroy(st);
    if (e != hipSuccess) return mm_fail("mmada_probe_cu_mask: %s", hipGetErrorString(e));
    return 0;
}
