// Checks what global_load_lds_dwordx4 does on gfx950: every active lane of a wavefront loads 16 bytes from its own
// global address into LDS at (wave-uniform base) + lane * 16; inactive lanes leave their slots alone; the data is there
// after s_waitcnt vmcnt(0) (+ a barrier for other wavefronts).
// Build: hipcc --offload-arch=gfx950 -O3 lds_async.hip -o lds_async
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern __shared__ float4 lds[];
__global__ void k(const float4* __restrict__ g, float4* out, int count) {
    for (int i = threadIdx.x; i < (int)blockDim.x; i += blockDim.x) lds[i] = make_float4(-1.f, -1.f, -1.f, -1.f);
    __syncthreads();
    for (int i = threadIdx.x; i < count; i += blockDim.x) {
        const int wave_first = i - (int)(threadIdx.x & 63u);
        __builtin_amdgcn_global_load_lds(g + i, (__attribute__((address_space(3))) void*)(lds + wave_first), 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // gfx9 encoding: vmcnt(0) (bits 3:0 and 15:14 = 0), expcnt 7 (bits 6:4) and lgkmcnt 15 (bits 11:8) left alone
    __syncthreads();
    out[threadIdx.x] = lds[threadIdx.x];
}
int main() {
    const int n = 256, count = 200;
    std::vector<float4> h(n);
    for (int i = 0; i < n; i++) h[i] = make_float4(i, i + 0.25f, i + 0.5f, i + 0.75f);
    float4 *g, *o;
    hipMalloc(&g, n * sizeof(float4)); hipMalloc(&o, n * sizeof(float4));
    hipMemcpy(g, h.data(), n * sizeof(float4), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(n), n * sizeof(float4), 0, g, o, count);
    std::vector<float4> r(n);
    hipMemcpy(r.data(), o, n * sizeof(float4), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) {
        const float want = i < count ? (float)i : -1.f;
        if (r[i].x != want || (i < count && r[i].w != i + 0.75f)) { if (bad < 5) printf("slot %d: %g %g %g %g\n", i, r[i].x, r[i].y, r[i].z, r[i].w); bad++; }
    }
    printf("%s: %d wrong of %d\n", bad ? "FAIL" : "ok", bad, n);
    return bad != 0;
}
