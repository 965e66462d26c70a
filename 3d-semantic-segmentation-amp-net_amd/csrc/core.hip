// core.hip -- error plumbing and version entry points of the C ABI (include/ampnet_hip.h).
#include "common.h"

namespace ampnet {

char *err_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace ampnet

extern "C" int ampnet_abi_version(void) { return AMPNET_ABI_VERSION; }
extern "C" const char *ampnet_last_error(void) { return ampnet::err_buf(); }
