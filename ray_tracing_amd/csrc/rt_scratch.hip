// rt_scratch.hip -- the library's one piece of statically allocated device memory, for callers that must not
// allocate (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).
//
// The render kernels need a small block of device memory beyond their inputs and outputs: the launch-wide
// counters (path segments traced, next pixel unit to hand out).  rt_render_tiles_device keeps that block in
// memory the context got from hipMalloc; rt_render_tiles_device_static keeps it in `scratch` below, a
// file-scope device array that every GPU this library runs on holds from the moment the library is loaded
// until the process ends.
//
// In host code the name `scratch` is a host-side stand-in, not a device address: passed to a kernel or to
// hipMemsetAsync it would compile and then fault on the GPU.  The device address is looked up once per GPU
// with hipGetSymbolAddress while that GPU is current (each GPU has its own copy of the array at its own
// address) and only that pointer is handed out.  Built by hipcc as an ordinary HIP object (the array lives
// in the object's own embedded gfx950 image, not in rt_kernels.hsaco) and linked into librt_hip.so.
#include <hip/hip_runtime.h>

#include <mutex>

#include "rt_scratch.h"

static __device__ __align__(256) unsigned char scratch[RT_SCRATCH_BYTES];

namespace {
std::mutex g_mu;
void* g_ptr[RT_SCRATCH_MAX_DEVICES];        // device address of `scratch` on GPU i, once looked up
}  // namespace

extern "C" int rt_scratch_lookup(int device, void** out, size_t* bytes) {
    if (!out) return (int)hipErrorInvalidValue;
    *out = nullptr;
    if (bytes) *bytes = RT_SCRATCH_BYTES;
    if (device < 0 || device >= RT_SCRATCH_MAX_DEVICES) return (int)hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_ptr[device]) {
        int current = -1;
        hipError_t e = hipGetDevice(&current);
        if (e != hipSuccess) return (int)e;
        if (current != device) return (int)hipErrorInvalidDevice;    // the caller makes `device` current first
        void* ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return (int)e;
        if (!ptr) return (int)hipErrorInvalidSymbol;
        g_ptr[device] = ptr;
    }
    *out = g_ptr[device];
    return 0;
}
