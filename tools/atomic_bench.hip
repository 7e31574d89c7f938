// Microbenchmark: what does a random 4-byte scatter-add / gather cost on MI355X for tables of the n-tuple sizes?
// Informs the design of k_td_update.  Build: hipcc --offload-arch=gfx950 -O3 -o atomic_bench atomic_bench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// mode 0: f32 atomic add no-return; 1: f32 atomic add returning; 2: u32 atomic add; 3: plain gather; 4: plain store;
// 5: LDS f32 atomic (table slice in LDS, 32K entries); 6: f32 atomic, skewed (zipf-ish: half of the adds to 1/256 of the table)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* table, uint32_t mask, uint32_t per_thread, float* sink) {
    __shared__ float lds[MODE == 5 ? 32768 : 1];
    uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (MODE == 5) {
        for (uint32_t i = threadIdx.x; i < 32768; i += 256) lds[i] = 0.f;
        __syncthreads();
    }
    float acc = 0.f;
    uint32_t h = hash32(t * 2654435761u + 12345u);
    for (uint32_t j = 0; j < per_thread; ++j) {
        h = hash32(h + j);
        uint32_t idx = h & mask;
        if (MODE == 6 && (h >> 31)) idx &= (mask >> 8);
        if (MODE == 0 || MODE == 6) __hip_atomic_fetch_add(&table[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 1) acc += __hip_atomic_fetch_add(&table[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 2) __hip_atomic_fetch_add((uint32_t*)&table[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 3) acc += table[idx];
        if (MODE == 4) table[idx] = 1.0f;
        if (MODE == 5) atomicAdd(&lds[idx & 32767u], 1.0f);
    }
    if (MODE == 5) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < 32768; i += 256) acc += lds[i];
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int MODE>
void run(const char* name, float* table, size_t entries, float* sink) {
    const uint32_t threads = 1u << 23, per_thread = 21;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    printf("[%s] table=%p entries=%zu sink=%p\n", name, (void*)table, entries, (void*)sink);
    k<MODE><<<threads / 256, 256>>>(table, (uint32_t)entries - 1, per_thread, sink);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 3; ++r) k<MODE><<<threads / 256, 256>>>(table, (uint32_t)entries - 1, per_thread, sink);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    double ops = 3.0 * threads * per_thread;
    printf("%-28s table %8.1f MB : %8.3f ms/launch  %8.2f Gop/s\n", name, entries * 4 / 1e6, ms / 3, ops / (ms * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
    setvbuf(stdout, NULL, _IONBF, 0);
    int only_lds = argc > 1;
    float* sink; CHECK(hipMalloc(&sink, 4));
    size_t sizes[] = {1u << 13, 1u << 18, 1u << 20, 1u << 22, 1u << 23, 1u << 25, 1u << 27};   // 32 KB .. 512 MB
    for (size_t e : sizes) {
        if (only_lds) break;
        float* table; CHECK(hipMalloc(&table, e * 4)); CHECK(hipMemset(table, 0, e * 4));
        run<0>("f32 atomic add (no return)", table, e, sink);
        run<6>("f32 atomic add, skewed", table, e, sink);
        run<1>("f32 atomic add (returning)", table, e, sink);
        run<2>("u32 atomic add", table, e, sink);
        run<3>("gather 4B", table, e, sink);
        run<4>("scatter store 4B", table, e, sink);
        CHECK(hipFree(table));
    }
    float* table; CHECK(hipMalloc(&table, 1 << 20)); 
    run<5>("LDS f32 atomic (32K slice)", table, 32768, sink);
    return 0;
}
