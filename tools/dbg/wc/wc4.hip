#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(float* out, const char* g, int n) {
    extern __shared__ char lds[];
    const int l = threadIdx.x * 16;
    bf16x8 a[6];
    for (int i = 0; i < 6; ++i) a[i] = *(bf16x8*)(lds + l + i * 4096);
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], a[1], acc, 0, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(g + l), (lptr_t)(lds + 32768 + (threadIdx.x / 64) * 1024), 16, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], a[3], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[4], a[5], acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) out[threadIdx.x * 16 + i] = acc[i];
}
