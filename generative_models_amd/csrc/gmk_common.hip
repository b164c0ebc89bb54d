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
