import sys
p=sys.argv[1]; s=open(p).read()
old='''template <int NI, bool FB, bool FIXED, uint32_t BIAS = 0u>
__device__ __forceinline__ void own_accum_idx('''
new='''__device__ unsigned long long g_own_phase[1024 * 8];
template <int NI, bool FB, bool FIXED, uint32_t BIAS = 0u>
__device__ __forceinline__ void own_accum_idx('''
assert old in s; s=s.replace(old,new,1)
old='''    {   // terminal queue
        const uint32_t q = *recs.qcount;'''
new='''    if (threadIdx.x == 0) g_own_phase[8 * blockIdx.x + 1] = wall_clock64();
    if (threadIdx.x == 960) g_own_phase[8 * blockIdx.x + 6] = wall_clock64();
    {   // terminal queue
        const uint32_t q = *recs.qcount;'''
assert old in s; s=s.replace(old,new,1)
old='''    own_dispatch<N, 0>(acc, s, recs, B, hits, dst, cdst, fb_hits, scale, cbits);
    __syncthreads();'''
new='''    if (threadIdx.x == 0) g_own_phase[8 * blockIdx.x] = wall_clock64();
    own_dispatch<N, 0>(acc, s, recs, B, hits, dst, cdst, fb_hits, scale, cbits);
    if (threadIdx.x == 0) g_own_phase[8 * blockIdx.x + 2] = wall_clock64();
    __syncthreads();
    if (threadIdx.x == 0) g_own_phase[8 * blockIdx.x + 3] = wall_clock64();'''
assert old in s; s=s.replace(old,new,1)
old='''    if (threadIdx.x == 0) wg_clock[2 * blockIdx.x + 1] = wall_clock64();'''
new='''    if (threadIdx.x == 0) g_own_phase[8 * blockIdx.x + 4] = wall_clock64();
    if (threadIdx.x == 0) wg_clock[2 * blockIdx.x + 1] = wall_clock64();'''
assert old in s; s=s.replace(old,new,1)
old='''int g2048_debug_owner_plan(g2048_ctx* c, uint64_t* out, uint32_t capacity, uint32_t* count) {'''
new='''int g2048_debug_owner_phases(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_own_phase), sizeof(g_own_phase)) == hipSuccess ? 0 : -2;
}

int g2048_debug_owner_plan(g2048_ctx* c, uint64_t* out, uint32_t capacity, uint32_t* count) {'''
assert old in s; s=s.replace(old,new,1)
open(p,'w').write(s)
