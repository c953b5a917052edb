// TEST INFRASTRUCTURE: the mock "device" allocator behind tests/hostmock/hip/hip_runtime.h (calloc with an optional byte limit, so that the driver can walk the
// engine's out-of-memory paths).
#include <hip/hip_runtime.h>

#include <map>

extern "C" { size_t mock_hip_mem_limit = 0; size_t mock_hip_mem_in_use = 0; int mock_hip_device_count = 1; }
static std::map<void*, size_t>& live() { static std::map<void*, size_t> m; return m; }

hipError_t hipMalloc(void** p, size_t bytes) {
    if (mock_hip_mem_limit && mock_hip_mem_in_use + bytes > mock_hip_mem_limit) { *p = nullptr; return hipErrorOutOfMemory; }
    *p = calloc(bytes ? bytes : 1, 1);
    if (!*p) return hipErrorOutOfMemory;
    live()[*p] = bytes; mock_hip_mem_in_use += bytes;
    return hipSuccess;
}
hipError_t hipFree(void* p) {
    if (!p) return hipSuccess;
    auto it = live().find(p);
    if (it == live().end()) return hipErrorInvalidValue;          // a pointer the mock never handed out (double free: ASan reports the free below too)
    mock_hip_mem_in_use -= it->second; live().erase(it);
    free(p);
    return hipSuccess;
}
extern "C" size_t mock_hip_live_allocations() { return live().size(); }
