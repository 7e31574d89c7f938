// What does k_td_play's lane I/O cost on its own?  Same arrays, widths and access pattern (one lane per thread, SoA), no
// game logic: 5 loads (57 B) and 14 stores (113 B) per lane.  hipcc --offload-arch=gfx950 -O3 -o io_pattern_bench io_pattern_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Arrays {
    uint4* boards; ulonglong2* rng; int* scores; float* label; uint8_t* flags;
    uint4* prev; uint2* q; uint4* x; uint16_t* c; float* dw1; uint16_t* last_move;
};

template <int MODE>      // 0: all stores, 1: without the six orbit-index stores, 2: only the 16-byte stores, 3: loads only (+4 B)
__global__ __launch_bounds__(256) void k_io(Arrays a, uint32_t B, uint32_t salt) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < B; i += gridDim.x * 256) {
        uint4 b = a.boards[i];
        ulonglong2 g = a.rng[i];
        int s = a.scores[i];
        float l = a.label[i];
        uint8_t f = a.flags[i];
        b.x ^= salt; g.x += b.y; s += (int)b.z; l += 1.0f; f ^= 1;
        a.boards[i] = b;
        a.rng[i] = g;
        if (MODE == 3) { a.dw1[i] = l + (float)s + (float)f; continue; }
        a.prev[i] = make_uint4(b.y, b.z, b.w, b.x);
        if (MODE <= 1) {
            a.scores[i] = s;
            a.label[i] = l;
            a.flags[i] = f;
            a.dw1[i] = l * 0.5f;
            a.last_move[i] = (uint16_t)b.x;
        }
        if (MODE == 0) {
            for (int v = 0; v < 4; ++v) a.q[(size_t)v * B + i] = make_uint2(b.x + v, b.y);
            a.x[i] = make_uint4(b.w, b.z, b.y, b.x);
            a.c[i] = (uint16_t)b.y;
        }
    }
}

int main() {
    const uint32_t B = 1u << 20;
    Arrays a;
    CK(hipMalloc(&a.boards, B * 16)); CK(hipMalloc(&a.rng, B * 16)); CK(hipMalloc(&a.scores, B * 4)); CK(hipMalloc(&a.label, B * 4));
    CK(hipMalloc(&a.flags, B)); CK(hipMalloc(&a.prev, B * 16)); CK(hipMalloc(&a.q, (size_t)B * 32)); CK(hipMalloc(&a.x, B * 16));
    CK(hipMalloc(&a.c, B * 2)); CK(hipMalloc(&a.dw1, B * 4)); CK(hipMalloc(&a.last_move, B * 2));
    CK(hipMemset(a.boards, 1, B * 16)); CK(hipMemset(a.rng, 2, B * 16)); CK(hipMemset(a.scores, 0, B * 4)); CK(hipMemset(a.label, 0, B * 4)); CK(hipMemset(a.flags, 0, B));
    // something between the launches that evicts the lane state from the caches, as the update kernels do
    float* big; CK(hipMalloc(&big, 256u << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {768, 1536, 4096}) {
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9f, sum = 0;
            for (int rep = 0; rep < 12; ++rep) {
                CK(hipMemsetAsync(big, rep, 64u << 20, 0));
                CK(hipEventRecord(e0, 0));
                switch (mode) {
                    case 0: k_io<0><<<grid, 256>>>(a, B, rep); break;
                    case 1: k_io<1><<<grid, 256>>>(a, B, rep); break;
                    case 2: k_io<2><<<grid, 256>>>(a, B, rep); break;
                    default: k_io<3><<<grid, 256>>>(a, B, rep); break;
                }
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 2) { best = ms < best ? ms : best; sum += ms; }
            }
            printf("grid %4d mode %d: best %.1f us, mean %.1f us\n", grid, mode, best * 1e3f, sum / 10 * 1e3f);
        }
    }
    return 0;
}
