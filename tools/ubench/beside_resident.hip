// beside_resident.hip — development probe: how long does a cov_blocks-shaped streaming kernel (256 threads, 72 registers, 12.8 KB of LDS, sixteen
// 16-byte loads per thread and tile, four staged phases with two barriers each) take beside N resident one-wave workgroups per CU with sd_sift's
// footprint (72 registers, 6400 bytes of LDS), idle (s_sleep) or busy (a dependent chain of vector instructions and LDS reads)?  bench.py sees
// the coverage kernel go from 3.1 to 3.55 ms between 19 and 19.5 resident sdust waves per CU; which resource is it?
//   hipcc --offload-arch=gfx950 -O3 -o beside_resident beside_resident.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// BUSY: 0 idle; 1 a dependent chain of vector instructions and LDS reads; 2 four independent chains of vector instructions (the vector ALU of a
// SIMD saturated by five such waves); 3 the same + a random LDS atomic and three random LDS reads per 16 vector instructions (sd_sift's tile loop);
// 4 scalar work as well (a ballot and scalar arithmetic per 8 vector instructions); 5-8 the vector work of kind 2 as 1-4 x 256 unrolled blocks: code of
// about 16-64 KB that every wave runs through again and again (the instruction cache)
template <int BUSY>
__global__ __launch_bounds__(64) void resident(unsigned long long *out, long long cycles)
{
    extern __shared__ unsigned int lds[];
    asm volatile("v_mov_b32 v71, 0" ::: "v71");            // 72 registers allocated
    const long long t0 = clock64();
    unsigned x = threadIdx.x, acc = 0, y = x * 3 + 1, z = x * 5 + 2, w = x * 7 + 3;
    unsigned long long sacc = 0;
    lds[threadIdx.x] = x;
    lds[64 + threadIdx.x] = x;
    while (clock64() - t0 < cycles) {
        if (BUSY == 1) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                x = x * 1664525u + 1013904223u;
                acc += lds[(x >> 8) & 63] ^ x;
            }
        } else if (BUSY >= 5) {
            // the same vector work as kind 2 as a LONG stretch of code (BUSY x 16 KB: the instruction cache is 64 KB per two CUs)
#pragma unroll
            for (int i = 0; i < (BUSY - 4) * 256; ++i) {
                x = (x << 1 | x >> 31) ^ (0x9E3779B9u + i); y = (y << 3 | y >> 29) + (0x7F4A7C15u ^ i); z = (z >> 5 | z << 27) ^ x; w = (w << 7 | w >> 25) + y;
                x += z & 0xFF; y ^= w >> 3; z += x | 5; w ^= y & 0x3F3F;
            }
        } else if (BUSY >= 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                x = (x << 1 | x >> 31) ^ 0x9E3779B9u; y = (y << 3 | y >> 29) + 0x7F4A7C15u; z = (z >> 5 | z << 27) ^ x; w = (w << 7 | w >> 25) + y;
                x += z & 0xFF; y ^= w >> 3; z += x | 5; w ^= y & 0x3F3F;
                if (BUSY >= 3 && (i & 1) == 1) {
                    __hip_atomic_fetch_or(&lds[(x >> 4) & 63], 1u << (threadIdx.x & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    acc += lds[(y >> 4) & 63] + lds[64 + ((z >> 4) & 63)] + lds[(w >> 4) & 127];
                }
                if (BUSY >= 4) {
                    const unsigned long long m = __ballot((x & 7) == 3);
                    sacc += __popcll(m) * 3 + (m >> 7);
                }
            }
        } else {
            __builtin_amdgcn_s_sleep(32);
        }
    }
    if (acc + x + y + z + w + (unsigned)sacc == 0x12345678u) out[0] = acc;
}

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_like(const u4 *src, unsigned *dst, long long n_tiles)
{
    __builtin_amdgcn_s_setprio(3);
    extern __shared__ __attribute__((aligned(16))) unsigned int sv[];      // 12800 bytes
    asm volatile("v_mov_b32 v71, 0" ::: "v71");
    const int t = threadIdx.x;
    const u4 *p = src + (size_t)blockIdx.x * 3200;          // 51200 bytes per tile
    u4 pre[4][4];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = t + k * 256;
            pre[ph][k] = v < 800 ? p[ph * 800 + v] : u4{0, 0, 0, 0};
        }
    unsigned f = 0;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
        if (ph) __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = t + k * 256;
            if (v < 800) reinterpret_cast<u4 *>(sv)[v] = pre[ph][k];
        }
        __syncthreads();
        if ((t >> 7) == (ph & 1)) {
            const unsigned *wd = sv + (t & 127) * 25;
#pragma unroll
            for (int i = 0; i < 25; ++i) f += wd[i];
        }
    }
    dst[(size_t)blockIdx.x * 256 + t] = f;
}

int main(int argc, char **argv)
{
    const long long n_tiles = 246884;                        // 3.16 G positions / 50 / 256
    const size_t bytes = (size_t)n_tiles * 51200;
    u4 *src; unsigned *dst; unsigned long long *out;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, (size_t)n_tiles * 256 * 4)); CK(hipMalloc(&out, 64));
    CK(hipMemset(src, 1, bytes));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&stream_like), hipFuncAttributeMaxDynamicSharedMemorySize, 12800));
    const int lds_list[] = {6400, 5120};
    const double per_cu[] = {0, 16, 18, 19, 19.25, 19.5, 20, 21, 23};
    for (int busy = (argc > 1 ? atoi(argv[1]) : 0); busy < 9; ++busy)
        for (int li = 0; li < (busy < 2 ? 2 : 1); ++li)
            for (double pc : per_cu) {
                const unsigned nres = (unsigned)(pc * 256);
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipDeviceSynchronize());
                    if (nres) {
                        if (busy == 1) resident<1><<<dim3(nres), dim3(64), lds_list[li], sa>>>(out, 14000000);     // ~6 ms
                        else if (busy == 2) resident<2><<<dim3(nres), dim3(64), lds_list[li], sa>>>(out, 14000000);
                        else if (busy == 3) resident<3><<<dim3(nres), dim3(64), lds_list[li], sa>>>(out, 14000000);
                        else if (busy == 4) resident<4><<<dim3(nres), dim3(64), lds_list[li], sa>>>(out, 14000000);
                        else if (busy == 5) resident<5><<<dim3(nres), dim3(64), lds_list[li], sa>>>(out, 14000000);
                        else if (busy == 6) resident<6><<<dim3(nres), dim3(64), lds_list[li], sa>>>(out, 14000000);
                        else if (busy == 7) resident<7><<<dim3(nres), dim3(64), lds_list[li], sa>>>(out, 14000000);
                        else if (busy == 8) resident<8><<<dim3(nres), dim3(64), lds_list[li], sa>>>(out, 14000000);
                        else resident<0><<<dim3(nres), dim3(64), lds_list[li], sa>>>(out, 14000000);
                    }
                    // (let the resident waves land first)
                    auto t0 = std::chrono::steady_clock::now();
                    while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 60.0) { }
                    CK(hipEventRecord(e0, sb));
                    stream_like<<<dim3((unsigned)n_tiles), dim3(256), 12800, sb>>>(src, dst, n_tiles);
                    CK(hipEventRecord(e1, sb));
                    CK(hipStreamSynchronize(sb));
                    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                printf("resident waves of kind %d, %d B of LDS each, %5.2f per CU: streaming kernel %.3f ms\n", busy, lds_list[li], pc, best);
                fflush(stdout);
            }
    CK(hipDeviceSynchronize());
    return 0;
}
