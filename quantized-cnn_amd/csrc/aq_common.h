// Host-side helpers shared by the C-ABI translation units: thread-local error text and
// argument checking.  No HIP allocation or synchronisation happens anywhere in the library.
#pragma once
#include <hip/hip_runtime.h>
#include <cerrno>
#include <cstdarg>
#include <cstdio>

namespace aq {

char *error_buffer();           // thread-local, 512 bytes
int fail(int code, const char *fmt, ...);
// name (with template arguments) of the main kernel of the most recent call on this thread, for aq_last_kernel()
void note_kernel(const char *fmt, ...);

inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-EIO, "%s: launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

} // namespace aq

#define AQ_REQUIRE(cond, code, ...)                       \
    do {                                                  \
        if (!(cond)) return ::aq::fail(code, __VA_ARGS__); \
    } while (0)
