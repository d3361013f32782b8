// crt_static.hip -- the per-GPU static scratch behind crt_render_static (include/crt_hip.h), the entry point for
// callers that cannot allocate device memory.
//
// `scratch` is ONE file-scope device array per GPU (every GPU the library's code object is loaded on holds its own
// copy, at its own address, until the process ends).  Its name is a host-side stand-in in host code, not a device
// address: the address is obtained once per device with hipGetSymbolAddress, with that device current, and only that
// pointer is handed to hipMemsetAsync and to the kernels -- never the array's name.
//
// Layout (unsigned long long units):
//   [0, CRT_STATIC_COUNTER_WORDS)                  work-queue counter, statistics, census (crt_device.h `counters`)
//   [CRT_STATIC_COUNTER_WORDS, + 3 * MAX_PIXELS)   int64 2^-40 fixed-point RGB sums of the frame
#include <hip/hip_runtime.h>

#include <mutex>

#include "crt_launch.h"

static_assert((CRT_STATIC_COUNTER_WORDS * sizeof(unsigned long long)) % 256 == 0, "the sums must start 256-byte aligned");
static __device__ __align__(256) unsigned long long scratch[CRT_STATIC_COUNTER_WORDS + 3ull * CRT_STATIC_MAX_PIXELS_];

namespace {
constexpr int MAX_DEVICES = 64;
unsigned long long* g_ptr[MAX_DEVICES] = {};  // per device: the address of ITS copy of `scratch`
std::mutex g_mutex;                            // contexts of several host threads may make their first call at once
}

// The device pointer of `device`'s scratch array.  First call per device (the first crt_render_static on it makes it):
// resolves the symbol with that device current; this is also what loads this file's code object -- and allocates
// the array -- on that GPU.
hipError_t crt_static_scratch(int device, unsigned long long** out) {
    if (device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_mutex);
    if (!g_ptr[device]) {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return e;
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_ptr[device] = (unsigned long long*)p;
    }
    *out = g_ptr[device];
    return hipSuccess;
}
