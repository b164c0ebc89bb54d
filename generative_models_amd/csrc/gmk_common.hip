// Error plumbing of the C ABI (include/gmk.h): per-thread message, no exceptions.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "gmk_common.h"

static thread_local char g_err[512] = "";

void gmk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int gmk_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gmk_set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

extern "C" int gmk_version(void) { return 1; }
extern "C" const char* gmk_last_error(void) { return g_err; }

// which kernel the last gmk_conv_igemm / gmk_conv_wgrad call of this thread launched (profiling aid, see bench.py)
static thread_local int g_last_kernel = 0;
void gmk_note_kernel(int id) { g_last_kernel = id; }
extern "C" int gmk_last_kernel(void) { return g_last_kernel; }

// kernel-selection overrides (development / A-B measurement): -1 = unset (environment variable, then automatic)
static int g_choice[4] = {-1, -1, -1, -1};
extern "C" int gmk_set_kernel_choice(int conv, int wgrad, int gn) {
    g_choice[0] = conv; g_choice[1] = wgrad; g_choice[2] = gn;
    return 0;
}
extern "C" int gmk_set_dev_variant(int v) { g_choice[3] = v; return 0; }
int gmk_kernel_choice(int which, const char* env) {
    if (g_choice[which] >= 0) return g_choice[which];
    const char* v = getenv(env);
    return v ? atoi(v) : 0;
}

// Workgroups the persistent kernels (one 512-thread, 160 KiB-LDS workgroup per CU) may occupy.  256 = the whole chip; a
// data-parallel run lowers it so that RCCL's all-reduce kernels find free CUs beside the backward pass instead of queueing
// behind a chip-filling persistent grid (generative_models_amd/parallel.py; GMK_CU_LIMIT overrides).
static int g_cu_limit = -1;
extern "C" int gmk_set_cu_limit(int n) {
    if (n < 8 || n > 256) { gmk_set_error("gmk_set_cu_limit: %d not in [8, 256]", n); return GMK_ERR_ARG; }
    g_cu_limit = n;
    return 0;
}
int gmk_cu_limit(void) {
    if (g_cu_limit > 0) return g_cu_limit;
    const char* v = getenv("GMK_CU_LIMIT");
    const int n = v ? atoi(v) : 256;
    return n >= 8 && n <= 256 ? n : 256;
}
extern "C" int gmk_get_cu_limit(void) { return gmk_cu_limit(); }

// fp32 mode arithmetic of the convolutions and weight gradients: exact fp32 MFMA chains (default: the parity mode) or, with
// GMK_FP32_SPLIT=1 / gmk_set_fp32_exact(0), operands split into bf16 hi + lo and three bf16 MFMAs per product (round 6; conv_igemm.hip split_bf16)
static int g_fp32_exact = -1;
extern "C" int gmk_set_fp32_exact(int exact) { g_fp32_exact = exact ? 1 : 0; return 0; }
bool fp32_split(void) {
    if (g_fp32_exact < 0) {
        const char* v = getenv("GMK_FP32_SPLIT");
        g_fp32_exact = (v && atoi(v) != 0) ? 0 : 1;
    }
    return g_fp32_exact == 0;
}
extern "C" int gmk_fp32_split(void) { return fp32_split() ? 1 : 0; }

