// Error plumbing and version for the C ABI (include/adamml_hip.h).
#include "common.h"
#include "../../include/adamml_hip.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

int adamml_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int adamml_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return adamml_set_error(ADAMML_ELAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
    return ADAMML_OK;
}

extern "C" {
int adamml_det_set_conv_gemm(int), adamml_det_set_conv3x3_c64(int), adamml_det_set_conv1x1_stream(int), adamml_det_set_conv_stem(int),
    adamml_det_set_dwconv(int), adamml_det_set_elementwise(int);
}
static int g_det = 1;        // exact integer-bin accumulation across workgroups (common.h); 0: fp64 slot atomics (A/B aid)
int adamml_deterministic_enabled(void) { return g_det; }

extern "C" int adamml_set_deterministic(int on) {
    on = on ? 1 : 0;
    int rc = adamml_det_set_conv_gemm(on) | adamml_det_set_conv3x3_c64(on) | adamml_det_set_conv1x1_stream(on) |
             adamml_det_set_conv_stem(on) | adamml_det_set_dwconv(on) | adamml_det_set_elementwise(on);
    if (rc) return adamml_set_error(ADAMML_ELAUNCH, "set_deterministic: hipMemcpyToSymbol failed (%d)", rc);
    g_det = on;
    return ADAMML_OK;
}
extern "C" int adamml_get_deterministic(void) { return g_det; }

extern "C" int adamml_version(void) { return 101; }
extern "C" const char* adamml_last_error_string(void) { return g_err; }
