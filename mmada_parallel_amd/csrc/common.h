// common.h — shared device helpers for the gfx950 kernels (bf16 bit handling, wave64 reductions).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define MM_DEVICE __device__ __forceinline__

// bf16 <-> f32, round-to-nearest-even (what torch's .to(bfloat16) does).  NaN kept quiet.
MM_DEVICE float bf2f(bf16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
// fp32 -> bf16, round to nearest even: gfx950 has the instruction (v_cvt_pk_bf16_f32; the cast below selects it).  Rounds 1-3
// did this with integer arithmetic (u += 0x7fff + lsb, a NaN branch): ~8 VALU + 6 scalar instructions per value, 4 values per
// SwiGLU output — the same bits for every finite and infinite input (tests/test_gpu_kernels.py::test_f2bf_matches_torch).
MM_DEVICE bf16_t f2bf(float f) {
    const __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
// round an f32 to the nearest bf16 value, returned as f32
MM_DEVICE float bfround(float f) { return bf2f(f2bf(f)); }
MM_DEVICE uint32_t pack_bf2(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }

// wave64 butterfly reductions
MM_DEVICE float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
MM_DEVICE double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
MM_DEVICE float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware, bijective remap of a linear workgroup id: consecutive hardware ids round-robin over the 8 XCDs
// (observed, speed only), so give each XCD a contiguous chunk of the logical tile sequence.
MM_DEVICE int xcd_remap(int bid, int nwg) {
    const int NX = 8;
    int xcd = bid % NX, idx = bid / NX;
    int q = nwg / NX, r = nwg % NX;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Position of key l inside the K-major V buffer: inside every 32-key block the eight 4-key chunks are stored in the order
// [0, 4, 1, 5, 2, 6, 3, 7] — position 8g+j holds key 4g+j (j < 4) or key 16+4g+(j-4), the eight keys lane quad g of a wave
// holds of two neighbouring 16-key score tiles (v_mfma_f32_16x16x32_bf16: lane (quad g, column q) = keys 4g..4g+3 of query q),
// so P goes from the score accumulators into the P·V operand without leaving its lane and the attention PV operand is one
// 16-byte read (attention.hip).  (Rounds 1-5: 16-key blocks in the key order of a 32x32 accumulator column.)
MM_DEVICE int vt_key_pos(int l) {
    const int c = (l >> 2) & 7;
    return (l & ~28) | ((c & 3) << 3) | ((c >> 2) << 2);
}

#define MM_CHECK_HIP(expr)                                                                 \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) return mm_fail("%s:%d %s: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
    } while (0)

int mm_fail(const char* fmt, ...);

// Function attributes (the dynamic-LDS limit of a kernel) are per DEVICE: a process that drives several devices (the ranks of
// a tensor-parallel group as handles of one process) must set them on each.  MM_ONCE_PER_DEVICE(once, stmts) runs `stmts`
// the first time a launcher runs on the current device and marks the device only AFTER every statement succeeded (an
// MM_CHECK_HIP inside returns from the launcher first, so a failed attribute call is retried by the next launch instead of
// turning into an LDS-size launch error); host threads driving two handles at worst set the same attribute twice.
struct MmOncePerDevice {
    std::atomic<bool> seen[16];
};
inline int mm_device_slot() {
    int dev = 0;
    return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16) ? dev : -1;
}
#define MM_ONCE_PER_DEVICE(once, ...)                                                      \
    do {                                                                                   \
        const int _slot = mm_device_slot();                                                \
        if (_slot < 0 || !(once).seen[_slot].load(std::memory_order_acquire)) {            \
            __VA_ARGS__;                                                                   \
            if (_slot >= 0) (once).seen[_slot].store(true, std::memory_order_release);     \
        }                                                                                  \
    } while (0)
