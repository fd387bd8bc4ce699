#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void k(float* out, int n) {
    extern __shared__ char lds[];
    const int l = threadIdx.x * 16;
    bf16x8 a[6];
    for (int i = 0; i < 6; ++i) a[i] = *(bf16x8*)(lds + l + i * 4096);
    float x = out[threadIdx.x];
#ifdef CLOB
#define CL : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15"
#else
#define CL
#endif
    asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" :: "v"(a[0]), "v"(a[1]) CL);
    x = x * 2.f + 1.f;
    asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" :: "v"(a[2]), "v"(a[3]) CL);
    x = x * 2.f + 1.f;
    asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" :: "v"(a[4]), "v"(a[5]) CL);
    out[threadIdx.x] = x;
}
