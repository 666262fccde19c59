// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

namespace nadm {

inline char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
inline int fail(const char* msg) {
    snprintf(err_buf(), 512, "%s", msg);
    return 1;
}
inline int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(err_buf(), 512, "%s: launch failed: %s", what, hipGetErrorString(e));
        return 2;
    }
    return 0;
}

}  // namespace nadm
