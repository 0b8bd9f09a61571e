// fetch_calib.hip — development aid: known byte counts under rocprofv3 --pmc FETCH_SIZE, to calibrate the counter for the
// access pattern of sdust_w64 (MI355X_MICROARCH.md: the x2 rule is measured for wide coalesced 16 B/lane streams only).
//   hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib ;  rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib
// Three kernels over the same 4 GiB (far beyond the 256 MiB Infinity Cache), each reading every byte exactly once:
//   calib_coalesced   16 B per lane, consecutive lanes consecutive addresses (the guide's pattern: counter = bytes / 2)
//   calib_lane32      every lane walks its OWN contiguous region in 32-byte requests (2 x dwordx4), the way a lane of sdust_w64
//                     pulls one half of a 64-byte block of its chunk at a time, the two halves 16 steps apart
//   calib_lane64      the same with 64 bytes per request group (4 x dwordx4 back to back)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void calib_coalesced(const uint4 *p, size_t n16, unsigned *sink)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x1234567u) *sink = acc;
}

template <int GROUP>   // 16-byte loads issued back to back per visit: 2 = 32 bytes, 4 = 64 bytes
__global__ void calib_lane(const uint4 *p, size_t region16, unsigned *sink)
{
    const size_t lane_id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint4 *q = p + lane_id * region16;
    unsigned acc = 0;
    for (size_t i = 0; i + GROUP <= region16; i += 4) {
        // first GROUP/… of the 64-byte block now, the rest of it a little later (other lanes' requests in between)
#pragma unroll
        for (int g = 0; g < GROUP; ++g) { const uint4 v = q[i + g]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
        if (GROUP < 4) {
            __builtin_amdgcn_s_sleep(8);
#pragma unroll
            for (int g = GROUP; g < 4; ++g) { const uint4 v = q[i + g]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
        }
    }
    if (acc == 0x1234567u) *sink = acc;
}

int main()
{
    const size_t bytes = 4ull << 30;
    uint4 *p = nullptr;
    unsigned *sink = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
    (void)hipMemset(p, 1, bytes);
    (void)hipDeviceSynchronize();
    const size_t n16 = bytes / 16;
    calib_coalesced<<<256 * 16, 256>>>(p, n16, sink);
    (void)hipDeviceSynchronize();
    // 19 waves per CU x 256 CUs x 64 lanes, like the production launch; the regions tile the buffer exactly
    const size_t lanes = 19 * 256 * 64;
    const size_t region16 = (n16 / lanes) & ~(size_t)3;
    calib_lane<2><<<19 * 256, 64>>>(p, region16, sink);
    (void)hipDeviceSynchronize();
    calib_lane<4><<<19 * 256, 64>>>(p, region16, sink);
    (void)hipDeviceSynchronize();
    printf("bytes read by calib_coalesced %zu\nbytes read by each calib_lane %zu (%zu lanes x %zu bytes)\n", n16 * 16, lanes * region16 * 16, lanes, region16 * 16);
    return 0;
}
