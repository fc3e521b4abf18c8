// valu_rate.hip -- measures what one MI355X sustains for plain and packed fp32 FMA, to fix the
// denominator of the path tracer's VALU roofline.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float float2_t __attribute__((ext_vector_type(2)));

template <int ACC>
__global__ void __launch_bounds__(256) k_fma(float* out, float a, float b, int iters) {
    float x[ACC];
#pragma unroll
    for (int i = 0; i < ACC; i++) x[i] = (float)threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ACC; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < ACC; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ACC>
__global__ void __launch_bounds__(256) k_pk_fma(float* out, float a, float b, int iters) {
    float2_t x[ACC];
    float2_t va = {a, a * 1.0001f}, vb = {b, b * 0.999f};
#pragma unroll
    for (int i = 0; i < ACC; i++) { x[i].x = (float)threadIdx.x * 1e-3f + i; x[i].y = x[i].x + 0.5f; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ACC; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(va), "v"(vb));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < ACC; i++) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; r++) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("device: %s %s, CUs %d, clockRate %d kHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate);
    const int iters = 65536, ACC = 16;
    float* out;
    for (int waves_per_simd = 1; waves_per_simd <= 8; waves_per_simd *= 2) {
        int blocks = p.multiProcessorCount * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD per block
        hipMalloc(&out, (size_t)blocks * 256 * 4);
        double ms1 = time_ms([&] { hipLaunchKernelGGL(k_fma<ACC>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 1e-4f, iters); }, 5);
        double ms2 = time_ms([&] { hipLaunchKernelGGL(k_pk_fma<ACC>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 1e-4f, iters); }, 5);
        double fl1 = 2.0 * blocks * 256.0 * iters * ACC, fl2 = 2.0 * fl1;
        printf("waves/SIMD %d: v_fma_f32 %.3f ms = %.1f TFLOP/s | v_pk_fma_f32 %.3f ms = %.1f TFLOP/s\n",
               waves_per_simd, ms1, fl1 / ms1 * 1e-9, ms2, fl2 / ms2 * 1e-9);
        hipFree(out);
    }
    return 0;
}
