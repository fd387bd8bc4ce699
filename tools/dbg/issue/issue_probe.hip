// issue_probe.hip — how much vector-ALU work hides beside MFMAs on a gfx950 SIMD?  (the attention kernel's time is
// matrix-pipe busy + VALU busy; this probe asks the hardware directly.)  Stand-alone: hipcc --offload-arch=gfx950 -O3 issue_probe.hip
//
// Every workgroup = 8 waves = two per SIMD, one workgroup per CU (256 workgroups).  Modes:
//   same  k : every wave runs  { 1 MFMA ; k VALU } x 8 per iteration (the VALU in the MFMA's shadow of the SAME wave)
//   split k : waves 0-3 run MFMAs only, waves 4-7 run k VALU per MFMA slot only (the partner wave's VALU beside the MFMAs)
// MFMA = v_mfma_f32_16x16x32_bf16 (or 32x32x16 with -DBIG), 8 independent accumulators; VALU = v_fma_f32 (or v_exp_f32 with
// TRANS) on 8 independent registers.  Prints time per MFMA slot per SIMD in cycles at the measured kernel clock (wall clock x
// nominal: the probe reports ns per slot; read ratios).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int K, bool TRANS, int OFF = 0>
__device__ __forceinline__ void valu(float (&v)[8], float c) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
        if (TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(i + OFF) & 7]));
        else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[(i + OFF) & 7]) : "v"(c));
    }
}

template <int MODE /*0 same, 1 split*/, int K, bool TRANS, bool BIG>
__global__ __launch_bounds__(512, 2) void probe(const bf16x8* __restrict__ in, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16x8 a = in[lane], b = in[64 + lane];
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 1.0f + lane * 1e-3f + i;
    const float c = 0.999f;
    f32x4 acc[8];
    f32x16 accb[4];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
    // MODE 0: both in one stream.  MODE 1: two separate loops picked once per wave (no per-slot branch in either stream).
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (BIG) accb[s & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, accb[s & 3], 0, 0, 0);
                else acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[s], 0, 0, 0);
                valu<K, TRANS>(v, c);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (__builtin_amdgcn_readfirstlane(wave) < 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (BIG) accb[s & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, accb[s & 3], 0, 0, 0);
                else acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[s], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                valu<K, TRANS>(v, c);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += v[i] + acc[i][0] + acc[i][3];
    for (int i = 0; i < 4; ++i) r += accb[i][0] + accb[i][15];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int MODE, int K, bool TRANS, bool BIG>
static void run(const char* name, const bf16x8* in, float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<MODE, K, TRANS, BIG>), dim3(256), dim3(512), 0, 0, in, out, iters);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<MODE, K, TRANS, BIG>), dim3(256), dim3(512), 0, 0, in, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // per SIMD: MODE 0: 2 waves x iters x 8 slots; MODE 1: 1 MFMA wave x iters x 8 slots (+ 1 VALU wave)
    const double slots = (MODE == 0 ? 2.0 : 1.0) * iters * 8.0;
    printf("%-44s %8.3f ms  %7.2f ns per MFMA slot per SIMD\n", name, best, best * 1e6 / slots);
}

#define ROW(MODE, K, TRANS, BIG, NAME) run<MODE, K, TRANS, BIG>(NAME, in, out, iters)
int main() {
    bf16x8* in; float* out;
    hipMalloc(&in, 128 * sizeof(bf16x8)); hipMalloc(&out, 256 * 512 * sizeof(float));
    unsigned short h[128 * 8];
    srand(1);
    for (int i = 0; i < 128 * 8; ++i) h[i] = 0x3f00 + (rand() & 0xff);   // random bf16 near 0.5-1 (never bench on zeros)
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 40000;
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL((probe<0, 0, false, false>), dim3(256), dim3(512), 0, 0, in, out, iters);  // settle the clock
    hipDeviceSynchronize();
    printf("# 16x16x32 MFMA, v_fma_f32 fillers\n");
    ROW(0, 0, false, false, "same wave: MFMA only (2 waves/SIMD)");
    ROW(0, 1, false, false, "same wave: MFMA + 1 fma");
    ROW(0, 2, false, false, "same wave: MFMA + 2 fma");
    ROW(0, 3, false, false, "same wave: MFMA + 3 fma");
    ROW(0, 4, false, false, "same wave: MFMA + 4 fma");
    ROW(0, 6, false, false, "same wave: MFMA + 6 fma");
    ROW(1, 0, false, false, "split: MFMA wave alone (1 wave/SIMD busy)");
    ROW(1, 1, false, false, "split: MFMA wave | partner 1 fma per slot");
    ROW(1, 2, false, false, "split: MFMA wave | partner 2 fma per slot");
    ROW(1, 3, false, false, "split: MFMA wave | partner 3 fma per slot");
    ROW(1, 4, false, false, "split: MFMA wave | partner 4 fma per slot");
    ROW(1, 6, false, false, "split: MFMA wave | partner 6 fma per slot");
    printf("# 16x16x32 MFMA, v_exp_f32 fillers\n");
    ROW(0, 1, true, false, "same wave: MFMA + 1 exp");
    ROW(0, 2, true, false, "same wave: MFMA + 2 exp");
    ROW(1, 1, true, false, "split: MFMA wave | partner 1 exp per slot");
    ROW(1, 2, true, false, "split: MFMA wave | partner 2 exp per slot");
    printf("# 32x32x16 MFMA, v_fma_f32 fillers\n");
    ROW(0, 0, false, true, "same wave: MFMA only (2 waves/SIMD)");
    ROW(0, 2, false, true, "same wave: MFMA + 2 fma");
    ROW(0, 4, false, true, "same wave: MFMA + 4 fma");
    ROW(0, 6, false, true, "same wave: MFMA + 6 fma");
    ROW(0, 8, false, true, "same wave: MFMA + 8 fma");
    ROW(1, 0, false, true, "split: MFMA wave alone");
    ROW(1, 4, false, true, "split: MFMA wave | partner 4 fma per slot");
    ROW(1, 8, false, true, "split: MFMA wave | partner 8 fma per slot");
    printf("# VALU wave alone beside idle partners: (split with the MFMA loop removed is not built; see the split rows: time = max of the two streams when they overlap)\n");
    return 0;
}
