// sos_abi.hip -- error plumbing shared by every entry point of libsos_hip.so.
#include "sos_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void sos_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int sos_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        sos_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return SOS_ELAUNCH;
    }
    return SOS_OK;
}

extern "C" int sos_abi_version(void) { return 10; }
// sizeof() of the descriptor structs the caller fills (0: sos_view, 1: sos_conv_desc, 2: sos_wgrad_desc): a binding whose mirror
// of a struct has drifted from the header finds out when it loads the library, not through a corrupted launch
extern "C" int sos_struct_size(int which) {
    switch (which) {
        case 0: return (int)sizeof(sos_view);
        case 1: return (int)sizeof(sos_conv_desc);
        case 2: return (int)sizeof(sos_wgrad_desc);
    }
    return -1;
}
extern "C" const char* sos_storage_dtype(void) { return SOS_STORAGE_NAME; }
extern "C" const char* sos_last_error(void) { return g_err; }
