// Optional per-kernel timing with HIP events on the launch stream (used by bench.py for the roofline line).
// Off by default; when off a ProfScope costs one branch.
#pragma once
#include <hip/hip_runtime.h>

namespace vn {
struct ProfScope {
    ProfScope(const char *name, hipStream_t st);
    ~ProfScope();
    int idx;
    hipStream_t st;
};
}  // namespace vn
