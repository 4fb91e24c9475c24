// Error plumbing of the C ABI: one message buffer per calling thread, no exceptions.
#include "equss_common.h"

namespace equss {

char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace equss

extern "C" int equss_abi_version(void) { return EQUSS_ABI_VERSION; }
extern "C" const char* equss_last_error(void) { return equss::error_buffer(); }
