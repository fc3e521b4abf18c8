// scan_rate.hip -- isolates the sphere-scan inner loop of the path tracer (9 VALU operations per sphere and
// lane, sphere data broadcast from LDS) and times it in scalar fp32 and in packed fp32 (two spheres per
// register pair, v_pk_fma_f32).  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scan_rate.hip -o scan_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float float2_t __attribute__((ext_vector_type(2)));
#define N_SPH 488

__global__ void __launch_bounds__(256) k_scalar(const float4* geom, uint32_t* out, int reps) {
    __shared__ float4 geo[N_SPH];
    for (int i = threadIdx.x; i < N_SPH; i += blockDim.x) geo[i] = geom[i];
    __syncthreads();
    float wx = threadIdx.x * 0.01f, wy = 0.3f, wz = -0.2f, ux = 0.6f, uy = 0.0f, uz = 0.8f, e = 0.1f;
    uint32_t acc = 0;
    for (int r = 0; r < reps; r++) {
        float k0 = -1.0f - (float)r;
        for (int blk = 0; blk < N_SPH; blk += 32) {
            int k = N_SPH - blk < 32 ? N_SPH - blk : 32;
            uint32_t miss = 0;
            for (int i = 0; i < k; i += 8) {
                float4 s[8];
#pragma unroll
                for (int u = 0; u < 8; u++) s[u] = geo[blk + i + u];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    float t = fmaf(wz, s[u].z, fmaf(wy, s[u].y, fmaf(wx, s[u].x, k0)));
                    float g = fmaf(uz, s[u].z, fmaf(uy, s[u].y, fmaf(ux, s[u].x, e)));
                    float key = fmaf(g, g, t) + s[u].w;
                    miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(key), 31u);
                }
            }
            acc ^= miss;
        }
        wx += 0.001f;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// pair layout: pa[j] = {cx(2j), cx(2j+1), cy(2j), cy(2j+1)}, pb[j] = {cz(2j), cz(2j+1), m(2j), m(2j+1)}
__global__ void __launch_bounds__(256) k_packed(const float4* geom, uint32_t* out, int reps) {
    __shared__ float4 pa[N_SPH / 2], pb[N_SPH / 2];
    for (int j = threadIdx.x; j < N_SPH / 2; j += blockDim.x) {
        float4 a = geom[2 * j], b = geom[2 * j + 1];
        pa[j] = make_float4(a.x, b.x, a.y, b.y);
        pb[j] = make_float4(a.z, b.z, a.w, b.w);
    }
    __syncthreads();
    float wx = threadIdx.x * 0.01f;
    float2_t wy = {0.3f, 0.3f}, wz = {-0.2f, -0.2f}, ux = {0.6f, 0.6f}, uy = {0.0f, 0.0f}, uz = {0.8f, 0.8f}, e = {0.1f, 0.1f};
    uint32_t acc = 0;
    for (int r = 0; r < reps; r++) {
        float2_t k0 = {-1.0f - (float)r, -1.0f - (float)r};
        float2_t wx2 = {wx, wx};
        for (int blk = 0; blk < N_SPH; blk += 32) {
            int k = N_SPH - blk < 32 ? N_SPH - blk : 32;
            uint32_t miss = 0;
            for (int i = 0; i < k; i += 8) {
                float4 a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { a[u] = pa[(blk + i) / 2 + u]; b[u] = pb[(blk + i) / 2 + u]; }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    float2_t cx = {a[u].x, a[u].y}, cy = {a[u].z, a[u].w}, cz = {b[u].x, b[u].y}, m = {b[u].z, b[u].w};
                    float2_t t, g, q, key;
                    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(wx2), "v"(cx), "v"(k0));
                    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(wy), "v"(cy), "v"(t));
                    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(wz), "v"(cz), "v"(t));
                    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(g) : "v"(ux), "v"(cx), "v"(e));
                    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(g) : "v"(uy), "v"(cy), "v"(g));
                    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(g) : "v"(uz), "v"(cz), "v"(g));
                    asm("v_pk_fma_f32 %0, %1, %1, %2" : "=v"(q) : "v"(g), "v"(t));
                    asm("v_pk_add_f32 %0, %1, %2" : "=v"(key) : "v"(q), "v"(m));
                    miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(key.x), 31u);
                    miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(key.y), 31u);
                }
            }
            acc ^= miss;
        }
        wx += 0.001f;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// camera-ray scan: 5 operations per sphere (t + m precomputed per sphere), 10 LDS reads per 8 spheres
__global__ void __launch_bounds__(256) k_primary(const float4* geom, uint32_t* out, int reps) {
    __shared__ float4 geo[N_SPH];
    __shared__ float4 tm0[N_SPH / 4];
    for (int i = threadIdx.x; i < N_SPH; i += blockDim.x) { geo[i] = geom[i]; ((float*)tm0)[i] = geom[i].w - 1.0f; }
    __syncthreads();
    float ux = 0.6f + threadIdx.x * 1e-4f, uy = 0.0f, uz = 0.8f;
    uint32_t acc = 0;
    for (int r = 0; r < reps; r++) {
        float e = 0.1f + (float)r;
        for (int blk = 0; blk < N_SPH; blk += 32) {
            int k = N_SPH - blk < 32 ? N_SPH - blk : 32;
            uint32_t miss = 0;
            for (int i = 0; i < k; i += 8) {
                float4 s[8];
#pragma unroll
                for (int u = 0; u < 8; u++) s[u] = geo[blk + i + u];
                float4 ta = tm0[(blk + i) >> 2], tb = tm0[((blk + i) >> 2) + 1];
                float tm[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    float g = fmaf(uz, s[u].z, fmaf(uy, s[u].y, fmaf(ux, s[u].x, e)));
                    float key = fmaf(g, g, tm[u]);
                    miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(key), 31u);
                }
            }
            acc ^= miss;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// the same with two rays per lane: every sphere read serves both
__global__ void __launch_bounds__(256) k_primary2(const float4* geom, uint32_t* out, int reps) {
    __shared__ float4 geo[N_SPH];
    __shared__ float4 tm0[N_SPH / 4];
    for (int i = threadIdx.x; i < N_SPH; i += blockDim.x) { geo[i] = geom[i]; ((float*)tm0)[i] = geom[i].w - 1.0f; }
    __syncthreads();
    float ux = 0.6f + threadIdx.x * 1e-4f, uy = 0.0f, uz = 0.8f, vx = 0.5f, vy = 0.1f + threadIdx.x * 1e-4f, vz = 0.7f;
    uint32_t acc = 0;
    for (int r = 0; r < reps; r++) {
        float e = 0.1f + (float)r, f = 0.2f + (float)r;
        for (int blk = 0; blk < N_SPH; blk += 32) {
            int k = N_SPH - blk < 32 ? N_SPH - blk : 32;
            uint32_t miss = 0, miss2 = 0;
            for (int i = 0; i < k; i += 8) {
                float4 s[8];
#pragma unroll
                for (int u = 0; u < 8; u++) s[u] = geo[blk + i + u];
                float4 ta = tm0[(blk + i) >> 2], tb = tm0[((blk + i) >> 2) + 1];
                float tm[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    float g = fmaf(uz, s[u].z, fmaf(uy, s[u].y, fmaf(ux, s[u].x, e)));
                    float key = fmaf(g, g, tm[u]);
                    miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(key), 31u);
                    float g2 = fmaf(vz, s[u].z, fmaf(vy, s[u].y, fmaf(vx, s[u].x, f)));
                    float key2 = fmaf(g2, g2, tm[u]);
                    miss2 = __builtin_amdgcn_alignbit(miss2, __float_as_uint(key2), 31u);
                }
            }
            acc ^= miss ^ (miss2 * 3u);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// camera-ray scan with t + m in the sphere's fourth word: 8 LDS reads per 8 spheres
__global__ void __launch_bounds__(256) k_primary8(const float4* geom, uint32_t* out, int reps) {
    __shared__ float4 geo[N_SPH];
    for (int i = threadIdx.x; i < N_SPH; i += blockDim.x) { geo[i] = geom[i]; geo[i].w -= 1.0f; }
    __syncthreads();
    float ux = 0.6f + threadIdx.x * 1e-4f, uy = 0.0f, uz = 0.8f;
    uint32_t acc = 0;
    for (int r = 0; r < reps; r++) {
        float e = 0.1f + (float)r;
        for (int blk = 0; blk < N_SPH; blk += 32) {
            int k = N_SPH - blk < 32 ? N_SPH - blk : 32;
            uint32_t miss = 0;
            for (int i = 0; i < k; i += 8) {
                float4 s[8];
#pragma unroll
                for (int u = 0; u < 8; u++) s[u] = geo[blk + i + u];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    float g = fmaf(uz, s[u].z, fmaf(uy, s[u].y, fmaf(ux, s[u].x, e)));
                    float key = fmaf(g, g, s[u].w);
                    miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(key), 31u);
                }
            }
            acc ^= miss;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// two rays per lane, t + m in the fourth word
__global__ void __launch_bounds__(256) k_primary28(const float4* geom, uint32_t* out, int reps) {
    __shared__ float4 geo[N_SPH];
    for (int i = threadIdx.x; i < N_SPH; i += blockDim.x) { geo[i] = geom[i]; geo[i].w -= 1.0f; }
    __syncthreads();
    float ux = 0.6f + threadIdx.x * 1e-4f, uy = 0.0f, uz = 0.8f, vx = 0.5f, vy = 0.1f + threadIdx.x * 1e-4f, vz = 0.7f;
    uint32_t acc = 0;
    for (int r = 0; r < reps; r++) {
        float e = 0.1f + (float)r, f = 0.2f + (float)r;
        for (int blk = 0; blk < N_SPH; blk += 32) {
            int k = N_SPH - blk < 32 ? N_SPH - blk : 32;
            uint32_t miss = 0, miss2 = 0;
            for (int i = 0; i < k; i += 8) {
                float4 s[8];
#pragma unroll
                for (int u = 0; u < 8; u++) s[u] = geo[blk + i + u];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    float g = fmaf(uz, s[u].z, fmaf(uy, s[u].y, fmaf(ux, s[u].x, e)));
                    float key = fmaf(g, g, s[u].w);
                    miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(key), 31u);
                    float g2 = fmaf(vz, s[u].z, fmaf(vy, s[u].y, fmaf(vx, s[u].x, f)));
                    float key2 = fmaf(g2, g2, s[u].w);
                    miss2 = __builtin_amdgcn_alignbit(miss2, __float_as_uint(key2), 31u);
                }
            }
            acc ^= miss ^ (miss2 * 3u);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// LDS side alone: the same eight broadcast ds_read_b128 per step, nothing computed on them
__global__ void __launch_bounds__(256) k_lds(const float4* geom, uint32_t* out, int reps) {
    __shared__ float4 geo[N_SPH];
    for (int i = threadIdx.x; i < N_SPH; i += blockDim.x) geo[i] = geom[i];
    __syncthreads();
    float4 s[8];
    for (int u = 0; u < 8; u++) s[u] = make_float4(0, 0, 0, 0);
    uint32_t base = (uint32_t)(uintptr_t)geo;
    for (int r = 0; r < reps; r++) {
        for (int i = 0; i < N_SPH; i += 8) {
            uint32_t addr = base + (uint32_t)i * 16u;
#pragma unroll
            for (int u = 0; u < 8; u++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(s[u]) : "v"(addr), "n"(u * 16));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    float acc = 0;
    for (int u = 0; u < 8; u++) acc += s[u].x + s[u].y + s[u].z + s[u].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_uint(acc);
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
    std::vector<float4> h(N_SPH);
    srand(1);
    for (auto& s : h) {
        float x = rand() % 2000 * 0.01f - 10, y = rand() % 100 * 0.01f, z = rand() % 2000 * 0.01f - 10;
        s = make_float4(x, y, z, 0.04f - (x * x + y * y + z * z));
    }
    float4* geom; uint32_t* out;
    hipMalloc(&geom, N_SPH * sizeof(float4));
    hipMemcpy(geom, h.data(), N_SPH * sizeof(float4), hipMemcpyHostToDevice);
    const int reps = 2000;
    for (int wps = 2; wps <= 8; wps += (wps < 4 ? 2 : (wps == 4 ? 1 : 3))) {
        int blocks = p.multiProcessorCount * wps;
        hipMalloc(&out, (size_t)blocks * 256 * 4);
        std::vector<uint32_t> r1((size_t)blocks * 256), r2((size_t)blocks * 256);
        double ms1 = time_ms([&] { hipLaunchKernelGGL(k_scalar, dim3(blocks), dim3(256), 0, 0, geom, out, reps); }, 3);
        hipMemcpy(r1.data(), out, r1.size() * 4, hipMemcpyDeviceToHost);
        double ms2 = time_ms([&] { hipLaunchKernelGGL(k_packed, dim3(blocks), dim3(256), 0, 0, geom, out, reps); }, 3);
        hipMemcpy(r2.data(), out, r2.size() * 4, hipMemcpyDeviceToHost);
        double ms3 = time_ms([&] { hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), 0, 0, geom, out, reps); }, 3);
        double ms4 = time_ms([&] { hipLaunchKernelGGL(k_primary, dim3(blocks), dim3(256), 0, 0, geom, out, reps); }, 3);
        double ms5 = time_ms([&] { hipLaunchKernelGGL(k_primary2, dim3(blocks), dim3(256), 0, 0, geom, out, reps); }, 3);
        double ms6 = time_ms([&] { hipLaunchKernelGGL(k_primary8, dim3(blocks), dim3(256), 0, 0, geom, out, reps); }, 3);
        double ms7 = time_ms([&] { hipLaunchKernelGGL(k_primary28, dim3(blocks), dim3(256), 0, 0, geom, out, reps); }, 3);
        printf("waves/SIMD %d: camera-ray scan, t+m in .w: %.2f ms; two rays per lane %.2f ms = %.2f per ray\n", wps, ms6, ms7, ms7 / 2);
        printf("waves/SIMD %d: camera-ray scan (5 ops) %.2f ms; two rays per lane %.2f ms = %.2f ms per ray\n", wps, ms4, ms5, ms5 / 2);
        size_t diff = 0;
        for (size_t i = 0; i < r1.size(); i++) diff += r1[i] != r2[i];
        double tests = (double)blocks * 256 * reps * N_SPH;
        printf("waves/SIMD %d: scalar %.2f ms = %.1f Gtests/s (%.1f TFLOP/s at 18/test) | packed %.2f ms = %.1f Gtests/s (%.1f) | LDS reads alone %.2f ms | mismatching lanes %zu\n",
               wps, ms1, tests / ms1 * 1e-6, tests * 18 / ms1 * 1e-9, ms2, tests / ms2 * 1e-6, tests * 18 / ms2 * 1e-9, ms3, diff);
        hipFree(out);
    }
    return 0;
}
