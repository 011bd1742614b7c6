// ABI version + thread-local error string of libtokenflow_hip.so.
#include <stdarg.h>

#include "tf_common.h"

static thread_local char g_err[512] = "";

void tf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int tf_abi_version(void) { return TF_ABI_VERSION; }
extern "C" const char* tf_last_error(void) { return g_err; }
