// Library-wide pieces of the C ABI: version and the thread-local error text.
#include "aq_common.h"
#include "../../include/aq_hip.h"

namespace aq {

char *error_buffer()
{
    static thread_local char buf[512] = "";
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

static char *kernel_buffer()
{
    static thread_local char buf[192] = "";
    return buf;
}

void note_kernel(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(kernel_buffer(), 192, fmt, ap);
    va_end(ap);
}

} // namespace aq

extern "C" {
int aq_abi_version(void) { return AQ_ABI_VERSION; }
const char *aq_last_error(void) { return aq::error_buffer(); }
const char *aq_last_kernel(void) { return aq::kernel_buffer(); }
}
