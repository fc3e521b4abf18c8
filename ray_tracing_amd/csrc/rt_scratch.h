// rt_scratch.h -- the statically allocated device scratch of librt_hip.so (rt_scratch.hip) as seen by the host
// launcher (rt_hip.cpp).  Not part of the public interface.
#ifndef RT_SCRATCH_H
#define RT_SCRATCH_H

#include <stddef.h>

// Size of the static device array, per GPU.  The render kernels' counter block (RT_COUNTER_BYTES in
// rt_device.h) must fit; it does not grow with the frame, the scene or the sample count.
#define RT_SCRATCH_BYTES 256
// GPUs (device ordinals) the per-device address table covers.
#define RT_SCRATCH_MAX_DEVICES 64

#ifdef __cplusplus
extern "C" {
#endif

// Device address of the static scratch array on GPU `device`, which must be the calling thread's current
// device.  The first call per GPU asks the runtime (hipGetSymbolAddress); later calls return the address kept
// from then.  Returns 0 or a hipError_t value.  Allocates nothing.
int rt_scratch_lookup(int device, void** out, size_t* bytes);

#ifdef __cplusplus
}
#endif
#endif
