// Microbenchmark: LDS atomic throughput per CU (f32 add, u32 add, u64 add, f32 with 1/8 of the lanes active, plain ds_write),
// 1024-thread workgroups owning a 128 KiB array, random slots.  Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, uint32_t iters, uint32_t hot_mask) {
    __shared__ float acc[32768];
    for (uint32_t j = threadIdx.x; j < 32768; j += 1024) acc[j] = 0.f;
    __syncthreads();
    uint32_t h = hash32(blockIdx.x * 1024 + threadIdx.x + 1);
    for (uint32_t i = 0; i < iters; ++i) {
        h = hash32(h + i);
        uint32_t idx = h & 32767u;
        if (hot_mask && (h >> 31)) idx &= hot_mask;                      // half of the adds on a small hot set
        if (MODE == 0) atomicAdd(&acc[idx], 1.0f);
        if (MODE == 1) atomicAdd((uint32_t*)&acc[idx], 1u);
        if (MODE == 2) atomicAdd((unsigned long long*)&acc[idx & 32766u], 1ull);
        if (MODE == 3 && (h & 0x7000000u) == 0) atomicAdd(&acc[idx], 1.0f);   // 1/8 of the lanes
        if (MODE == 4) acc[idx] = 1.0f;
        if (MODE == 5 && (h & 0x7000000u) == 0) atomicAdd((uint32_t*)&acc[idx], 1u);
    }
    __syncthreads();
    float s = 0.f;
    for (uint32_t j = threadIdx.x; j < 32768; j += 1024) s += acc[j];
    if (s == 123.456f) out[0] = s;
}

template <int MODE>
int run(const char* name, float* out, uint32_t hot_mask) {
    const uint32_t iters = 2048, blocks = 256;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    k<MODE><<<blocks, 1024>>>(out, iters, hot_mask);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    k<MODE><<<blocks, 1024>>>(out, iters, hot_mask);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    double lane_ops = (double)blocks * 1024 * iters * ((MODE == 3 || MODE == 5) ? 0.125 : 1.0);
    printf("%-34s hot_mask %5u : %8.3f ms  %8.1f G lane-ops/s chip  %6.2f lane-ops/cycle/CU (2.4 GHz)\n", name, hot_mask, ms,
           lane_ops / (ms * 1e-3) / 1e9, lane_ops / (ms * 1e-3) / 256 / 2.4e9);
    return 0;
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    float* out; CHECK(hipMalloc(&out, 4));
    for (uint32_t hot : {0u, 1023u, 31u, 0u + 0}) {
        if (run<0>("ds_add_f32", out, hot)) return 1;
        if (run<1>("ds_add_u32", out, hot)) return 1;
        if (run<2>("ds_add_u64", out, hot)) return 1;
        if (run<3>("ds_add_f32, 1/8 lanes active", out, hot)) return 1;
        if (run<5>("ds_add_u32, 1/8 lanes active", out, hot)) return 1;
        if (run<4>("ds_write_b32", out, hot)) return 1;
        if (hot == 0 && false) break;
    }
    return 0;
}
