// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <math.h>

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

// step-dependent scalars of the Adam update (bias corrections folded in): step_size = lr / (1 - b1^t), inv_bc2 = 1 / sqrt(1 - b2^t)
inline void adam_scalars(float lr, int step, float* step_size, float* inv_bc2) {
    const double bc1 = 1.0 - pow(0.9, (double)step);
    const double bc2 = 1.0 - pow(0.95, (double)step);
    *step_size = (float)((double)lr / bc1);
    *inv_bc2 = (float)(1.0 / sqrt(bc2));
}

}  // namespace nadm
