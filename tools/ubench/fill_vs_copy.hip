// fill_vs_copy.hip — development probe: how long does a small piece of work on one stream take while a large device -> pinned-host copy
// runs on another stream (the lazy result copies of a bench step)?  Host-timed from the launch to the end of the small stream's work.
//   hipcc --offload-arch=gfx950 -O3 -o fill_vs_copy fill_vs_copy.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

static __global__ void tiny(unsigned *p) { if (threadIdx.x == 0) p[0] += 1; }
static __global__ void mid(unsigned *p, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 3 + 1;
}
static __global__ void resident(unsigned long long *p, long long cycles)
{
    long long t0 = clock64();
    while (clock64() - t0 < cycles) { }
    if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = 1;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t big = 60u << 20;
    void *d_big, *h_big; unsigned *d_small, *d_mid; unsigned long long *d_res;
    CK(hipMalloc(&d_big, big)); CK(hipHostMalloc(&h_big, big, hipHostMallocDefault));
    CK(hipMalloc(&d_small, 4096)); CK(hipMalloc(&d_mid, 64u << 20)); CK(hipMalloc(&d_res, 64));
    hipStream_t sa, sb, sc; hipEvent_t ev;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CK(hipMemset(d_big, 1, big)); CK(hipMemset(d_mid, 1, 64u << 20)); CK(hipDeviceSynchronize());
    const char *names[] = {"memset 64 B", "tiny kernel", "memset + tiny kernel", "16M-thread kernel", "memset + 16M-thread kernel", "tiny kernel + 64 B to the host", "memset + kernel + 64 B to the host", "the same between two timed events"};
    hipEvent_t te0, te1; CK(hipEventCreate(&te0)); CK(hipEventCreate(&te1));
    unsigned *p_small;
    CK(hipHostMalloc(&p_small, 4096, hipHostMallocDefault));
    for (int with_res = 0; with_res < 2; ++with_res)
    for (int with_copy = 0; with_copy < 2; ++with_copy)
        for (int what = 0; what < 8; ++what) {
            double best = 1e30, sum = 0;
            for (int rep = 0; rep < 12; ++rep) {
                CK(hipDeviceSynchronize());
                if (with_res) resident<<<dim3(256 * 5), dim3(256), 0, sc>>>(d_res, 4000000);      // ~2 ms of waves on 5/8 of the slots
                if (with_copy) {
                    CK(hipEventRecord(ev, sb));
                    CK(hipStreamWaitEvent(sa, ev, 0));
                    CK(hipMemcpyAsync(h_big, d_big, big, hipMemcpyDeviceToHost, sa));
                }
                const double t0 = now();
                if (what == 0 || what == 2 || what == 4) CK(hipMemsetAsync(d_small, 0, 64, sb));
                if (what == 1 || what == 2) tiny<<<1, 64, 0, sb>>>(d_small);
                if (what == 3 || what == 4) mid<<<dim3((16u << 20) / 256), dim3(256), 0, sb>>>(d_mid, 16u << 20);
                if (what == 6 || what == 7) CK(hipMemsetAsync(d_small, 0, 64, sb));
                if (what == 7) CK(hipEventRecord(te0, sb));
                if (what == 5 || what == 6 || what == 7) {
                    tiny<<<1, 64, 0, sb>>>(d_small);
                    if (what == 7) CK(hipEventRecord(te1, sb));
                    CK(hipMemcpyAsync(p_small, d_small, 64, hipMemcpyDeviceToHost, sb));
                }
                CK(hipStreamSynchronize(sb));
                const double t = now() - t0;
                if (rep >= 2) { sum += t; if (t < best) best = t; }
            }
            printf("%-14s %-10s %-28s best %8.1f us  mean %8.1f us\n", with_res ? "resident waves" : "idle GPU", with_copy ? "copy 60 MB" : "no copy", names[what], best, sum / 10);
        }
    CK(hipDeviceSynchronize());
    return 0;
}
