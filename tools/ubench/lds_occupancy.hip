// How many workgroups does a CU of MI355X hold at once as a function of the LDS a workgroup declares?  (development aid)
// Every workgroup stamps its start, spins 300 us, stamps its end; the host counts the workgroups running at start + 150 us.
//   hipcc --offload-arch=gfx950 -O2 -o lds_occupancy lds_occupancy.hip && ./lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
extern __shared__ unsigned char dyn[];
__global__ void occ(unsigned long long *t, int spin_ticks)
{
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) dyn[0] = 1;
    while (wall_clock64() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = t0; t[2 * blockIdx.x + 1] = wall_clock64(); }
}
int main()
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipFuncSetAttribute(reinterpret_cast<const void *>(occ), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int sizes[] = {0, 4096, 7680, 8192, 8448, 8704, 8960, 9216, 9728, 10240, 10752, 16384, 16896, 17408, 18432, 20480, 32768, 33792, 34816, 36864};
    for (int threads : {64, 128, 256})
        for (int lds : sizes) {
            const int nb = cus * 48;
            unsigned long long *d;
            hipMalloc(&d, sizeof(unsigned long long) * 2 * nb);
            hipMemset(d, 0, sizeof(unsigned long long) * 2 * nb);
            hipLaunchKernelGGL(occ, dim3(nb), dim3(threads), lds, 0, d, 30000);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed lds %d\n", lds); continue; }
            std::vector<unsigned long long> t(2 * nb);
            hipMemcpy(t.data(), d, sizeof(unsigned long long) * 2 * nb, hipMemcpyDeviceToHost);
            unsigned long long first = ~0ull;
            for (int i = 0; i < nb; ++i) first = std::min(first, t[2 * i]);
            int running = 0;
            for (int i = 0; i < nb; ++i) running += t[2 * i] <= first + 15000 && t[2 * i + 1] > first + 15000;
            int api = 0;
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, occ, threads, lds);
            printf("threads %3d lds %6d: %5d workgroups at once = %.2f per CU = %.2f waves per CU (occupancy API: %d workgroups)\n", threads, lds, running, (double)running / cus,
                   (double)running / cus * threads / 64, api);
            hipFree(d);
        }
    return 0;
}
