// vgpr_bank.hip -- does the source-register assignment of v_fma_f32 change its issue rate on gfx950?
// Sixteen independent accumulators, the three sources of every instruction placed by hand:
//   spread : a = v1, b = v2, accumulators in v[16..31]          (banks mixed, as a compiler would leave them)
//   apart  : a = v1, b = v2, accumulators only in registers = 0 or 3 (mod 4)
//   same   : a = v4, b = v8, accumulators only in registers = 0 (mod 4)
// Build: hipcc --offload-arch=gfx950 -O3 vgpr_bank.hip -o vgpr_bank
#include <hip/hip_runtime.h>
#include <cstdio>

#define FMA(acc, a, b) "v_fma_f32 v" #acc ", v" #a ", v" #b ", v" #acc "\n"
#define INIT(r) "v_mov_b32 v" #r ", 1.0\n"

__global__ void __launch_bounds__(256) k_spread(float* out, int iters) {
    asm volatile("v_mov_b32 v1, 0x3f800347\n v_mov_b32 v2, 0x38d1b717\n"
                 INIT(16) INIT(17) INIT(18) INIT(19) INIT(20) INIT(21) INIT(22) INIT(23)
                 INIT(24) INIT(25) INIT(26) INIT(27) INIT(28) INIT(29) INIT(30) INIT(31) ::: "v1", "v2",
                 "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31");
    for (int it = 0; it < iters; it++)
        asm volatile(FMA(16,1,2) FMA(17,1,2) FMA(18,1,2) FMA(19,1,2) FMA(20,1,2) FMA(21,1,2) FMA(22,1,2) FMA(23,1,2)
                     FMA(24,1,2) FMA(25,1,2) FMA(26,1,2) FMA(27,1,2) FMA(28,1,2) FMA(29,1,2) FMA(30,1,2) FMA(31,1,2) :::
                     "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31");
    float r;
    asm volatile("v_add_f32 %0, v16, v31" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

__global__ void __launch_bounds__(256) k_apart(float* out, int iters) {
    asm volatile("v_mov_b32 v1, 0x3f800347\n v_mov_b32 v2, 0x38d1b717\n"
                 INIT(16) INIT(19) INIT(20) INIT(23) INIT(24) INIT(27) INIT(28) INIT(31)
                 INIT(32) INIT(35) INIT(36) INIT(39) INIT(40) INIT(43) INIT(44) INIT(47) ::: "v1", "v2",
                 "v16","v19","v20","v23","v24","v27","v28","v31","v32","v35","v36","v39","v40","v43","v44","v47");
    for (int it = 0; it < iters; it++)
        asm volatile(FMA(16,1,2) FMA(19,1,2) FMA(20,1,2) FMA(23,1,2) FMA(24,1,2) FMA(27,1,2) FMA(28,1,2) FMA(31,1,2)
                     FMA(32,1,2) FMA(35,1,2) FMA(36,1,2) FMA(39,1,2) FMA(40,1,2) FMA(43,1,2) FMA(44,1,2) FMA(47,1,2) :::
                     "v16","v19","v20","v23","v24","v27","v28","v31","v32","v35","v36","v39","v40","v43","v44","v47");
    float r;
    asm volatile("v_add_f32 %0, v16, v47" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

__global__ void __launch_bounds__(256) k_same(float* out, int iters) {
    asm volatile("v_mov_b32 v4, 0x3f800347\n v_mov_b32 v8, 0x38d1b717\n"
                 INIT(12) INIT(16) INIT(20) INIT(24) INIT(28) INIT(32) INIT(36) INIT(40)
                 INIT(44) INIT(48) INIT(52) INIT(56) INIT(60) INIT(64) INIT(68) INIT(72) ::: "v4", "v8",
                 "v12","v16","v20","v24","v28","v32","v36","v40","v44","v48","v52","v56","v60","v64","v68","v72");
    for (int it = 0; it < iters; it++)
        asm volatile(FMA(12,4,8) FMA(16,4,8) FMA(20,4,8) FMA(24,4,8) FMA(28,4,8) FMA(32,4,8) FMA(36,4,8) FMA(40,4,8)
                     FMA(44,4,8) FMA(48,4,8) FMA(52,4,8) FMA(56,4,8) FMA(60,4,8) FMA(64,4,8) FMA(68,4,8) FMA(72,4,8) :::
                     "v12","v16","v20","v24","v28","v32","v36","v40","v44","v48","v52","v56","v60","v64","v68","v72");
    float r;
    asm volatile("v_add_f32 %0, v12, v72" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// two sources only (a * b + inline constant is not encodable; use v_mul_f32 acc = acc * a): 2 VGPR reads
__global__ void __launch_bounds__(256) k_mul2(float* out, int iters) {
    asm volatile("v_mov_b32 v1, 0x3f800001\n"
                 INIT(16) INIT(17) INIT(18) INIT(19) INIT(20) INIT(21) INIT(22) INIT(23)
                 INIT(24) INIT(25) INIT(26) INIT(27) INIT(28) INIT(29) INIT(30) INIT(31) ::: "v1",
                 "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31");
#define MUL(acc) "v_mul_f32 v" #acc ", v1, v" #acc "\n"
    for (int it = 0; it < iters; it++)
        asm volatile(MUL(16) MUL(17) MUL(18) MUL(19) MUL(20) MUL(21) MUL(22) MUL(23)
                     MUL(24) MUL(25) MUL(26) MUL(27) MUL(28) MUL(29) MUL(30) MUL(31) :::
                     "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31");
    float r;
    asm volatile("v_add_f32 %0, v16, v31" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// fma with an SGPR and an inline constant among the sources: one VGPR read
__global__ void __launch_bounds__(256) k_fma1(float* out, int iters) {
    asm volatile(INIT(16) INIT(17) INIT(18) INIT(19) INIT(20) INIT(21) INIT(22) INIT(23)
                 INIT(24) INIT(25) INIT(26) INIT(27) INIT(28) INIT(29) INIT(30) INIT(31) :::
                 "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31");
#define FM1(acc) "v_fma_f32 v" #acc ", v" #acc ", 0.5, 0.5\n"
    for (int it = 0; it < iters; it++)
        asm volatile(FM1(16) FM1(17) FM1(18) FM1(19) FM1(20) FM1(21) FM1(22) FM1(23)
                     FM1(24) FM1(25) FM1(26) FM1(27) FM1(28) FM1(29) FM1(30) FM1(31) :::
                     "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31");
    float r;
    asm volatile("v_add_f32 %0, v16, v31" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <typename K>
static double time_ms(K kern, int blocks, float* out, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int iters = 400000;
    for (int wps : {4, 8}) {
        int blocks = p.multiProcessorCount * wps;
        float* out;
        (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
        double n = (double)blocks * 4 * iters * 16;   // wave-instructions
        struct { const char* name; double ms; } r[] = {
            {"spread", time_ms(k_spread, blocks, out, iters)}, {"apart", time_ms(k_apart, blocks, out, iters)},
            {"same", time_ms(k_same, blocks, out, iters)}, {"mul 2 src", time_ms(k_mul2, blocks, out, iters)},
            {"fma 1 vgpr src", time_ms(k_fma1, blocks, out, iters)}};
        for (auto& x : r)
            printf("waves/SIMD %d %-15s %8.2f ms  %.3f G wave-instr/s/SIMD  (%.1f TFLOP/s as FMA)\n", wps, x.name, x.ms,
                   n / (p.multiProcessorCount * 4.0) / x.ms * 1e-6, n * 128 / x.ms * 1e-9);
        (void)hipFree(out);
    }
    return 0;
}
