// tune_api.hip — C entry of tools/libmmada_tune.so: the rejected / experimental GEMM variants of gemm_var.hip, kept out
// of the product library (libmmada_mi355x.so) and built only by tools/gemm_sweep.py for re-measurement.
#include <cstdarg>
#include <cstdio>

#include "gemm_epilogue.h"

static thread_local char g_err[512] = "";

int mm_fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

int launch_gemm_variant(int variant, const GemmArgs& g, hipStream_t s);  // gemm_var.hip

extern "C" {
const char* mmada_tune_last_error(void) { return g_err; }

// variant 100 = the production kernel (csrc/gemm.hip compiled into this library too, BM picker included)
int mmada_gemm_variant(int variant, const void* A, const void* W, void* C, int M, int N, int K, void* stream) {
    if (!A || !W || !C) return mm_fail("mmada_gemm_variant: null argument");
    GemmArgs g{};
    g.A = (const bf16_t*)A; g.W = (const bf16_t*)W; g.C = (bf16_t*)C;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N;
    // 100: production planner; 300 + c: production with the 8-phase configuration c forced; 1000 + BM: production with the
    // 16-wave kernel's row tile forced
    if (variant == 100 || (variant >= 300 && variant < 300 + GEMM8_NCFG + 20) || variant >= 1000) {
        gemm_force_config(variant == 100 ? -1 : variant >= 1000 ? variant : variant - 300);
        const int rc = launch_gemm(EPI_STORE, g, (hipStream_t)stream);
        gemm_force_config(-1);
        return rc;
    }
    return launch_gemm_variant(variant, g, (hipStream_t)stream);
}
}
