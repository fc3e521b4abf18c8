/* hip_alloc_count.c -- TEST INFRASTRUCTURE: counts the HIP runtime's allocation calls made by libraries loaded after
 * it.  tests/test_zz_no_alloc.py loads this object with RTLD_GLOBAL before the HIP runtime and librt_hip.so in a child
 * process, so that the library's references to hipMalloc, hipFree, ... bind here (symbol lookup goes through the
 * global scope in load order); each wrapper counts and forwards to the real runtime (HIP_ALLOC_COUNT_RUNTIME names
 * its file).  No HIP header is included on purpose: the wrappers carry the runtime's C ABI (hipError_t is an int).
 * Nothing in the product links or loads this file. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stddef.h>
#include <stdlib.h>

static long n_calls[2];   /* [0] allocations, [1] frees */

static void *real(const char *name) {
    static void *handle;
    if (!handle) {
        const char *path = getenv("HIP_ALLOC_COUNT_RUNTIME");
        handle = dlopen(path && *path ? path : "libamdhip64.so.7", RTLD_NOW | RTLD_GLOBAL);
        if (!handle) abort();
    }
    void *f = dlsym(handle, name);
    if (!f) abort();
    return f;
}

#define COUNT(which) __atomic_add_fetch(&n_calls[which], 1, __ATOMIC_RELAXED)

int hipMalloc(void **p, size_t n) { COUNT(0); return ((int (*)(void **, size_t))real("hipMalloc"))(p, n); }
int hipFree(void *p) { COUNT(1); return ((int (*)(void *))real("hipFree"))(p); }
int hipMallocAsync(void **p, size_t n, void *s) { COUNT(0); return ((int (*)(void **, size_t, void *))real("hipMallocAsync"))(p, n, s); }
int hipFreeAsync(void *p, void *s) { COUNT(1); return ((int (*)(void *, void *))real("hipFreeAsync"))(p, s); }
int hipExtMallocWithFlags(void **p, size_t n, unsigned f) { COUNT(0); return ((int (*)(void **, size_t, unsigned))real("hipExtMallocWithFlags"))(p, n, f); }
int hipMallocManaged(void **p, size_t n, unsigned f) { COUNT(0); return ((int (*)(void **, size_t, unsigned))real("hipMallocManaged"))(p, n, f); }
int hipHostMalloc(void **p, size_t n, unsigned f) { COUNT(0); return ((int (*)(void **, size_t, unsigned))real("hipHostMalloc"))(p, n, f); }
int hipHostFree(void *p) { COUNT(1); return ((int (*)(void *))real("hipHostFree"))(p); }
int hipMallocPitch(void **p, size_t *pitch, size_t w, size_t h) { COUNT(0); return ((int (*)(void **, size_t *, size_t, size_t))real("hipMallocPitch"))(p, pitch, w, h); }

long hip_alloc_count(int which) { return which == 0 || which == 1 ? __atomic_load_n(&n_calls[which], __ATOMIC_RELAXED) : -1; }
