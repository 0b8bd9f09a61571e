// valu_rate.hip — development aid: issue cost (cycles per wave-instruction per SIMD) of the instruction classes the
// sdust kernel is made of, on gfx950, at 1..8 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
// Every test: each wave runs ITER iterations of a block of 16 instructions of one class on 8 independent register chains
// (so that a single wave is never limited by a dependent chain of less than 8), stamps s_memtime around the loop; the host
// reports (max end - min start) / (waves per SIMD * instructions per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>

#define ITER 2048

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define R16(X) R8(X) R8(X)

template <int KIND>
__global__ __launch_bounds__(64) void k(unsigned long long *t0, unsigned long long *t1, unsigned *sink, int sarg)
{
    __shared__ __attribute__((aligned(16))) unsigned lds[64 * 17];
    unsigned v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 2654435761u + i * 40503u + sarg;
    lds[threadIdx.x] = 0;
    for (int i = 0; i < 16; ++i) lds[threadIdx.x + 64 * i] = i;
    unsigned s0 = sarg, s1 = sarg + 1, s2 = sarg + 2, s3 = sarg + 3;
    unsigned long long sm = 0;
    __syncthreads();
    const unsigned long long a = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
        if (KIND == 0) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
            R16(X)
#undef X
        } else if (KIND == 1) {
#define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
            R16(X)
#undef X
        } else if (KIND == 2) {
#define X(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(v[i]));
            R16(X)
#undef X
        } else if (KIND == 3) {
#define X(i) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(v[i]));
            R16(X)
#undef X
        } else if (KIND == 4) {
#define X(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 7]), "v"(v[(i + 2) & 7]));
            R16(X)
#undef X
        } else if (KIND == 5) {
#define X(i) asm volatile("v_alignbyte_b32 %0, %0, %1, 3" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
            R16(X)
#undef X
        } else if (KIND == 6) {
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 7]), "v"(v[(i + 2) & 7]));
            R16(X)
#undef X
        } else if (KIND == 7) {   // compare into vcc + select
#define X(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(v[(i + 1) & 7]) : "vcc");
            R8(X)
#undef X
        } else if (KIND == 8) {   // compare into an SGPR pair (ballot)
#define X(i) asm volatile("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(sm) : "v"(v[i]), "v"(v[(i + 1) & 7]));
            R16(X)
#undef X
        } else if (KIND == 9) {   // readlane
#define X(i) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s0) : "v"(v[i]));
            R16(X)
#undef X
        } else if (KIND == 10) {  // DPP add
#define X(i) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
            R16(X)
#undef X
        } else if (KIND == 11) {  // VALU with an SGPR operand
#define X(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(v[i]) : "s"(s1));
            R16(X)
#undef X
        } else if (KIND == 12) {  // 64-bit shift
            unsigned long long w[4] = {((unsigned long long)v[0] << 32) | v[1], ((unsigned long long)v[2] << 32) | v[3], ((unsigned long long)v[4] << 32) | v[5], ((unsigned long long)v[6] << 32) | v[7]};
#define X(i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(w[(i) & 3]));
            R16(X)
#undef X
            v[0] ^= (unsigned)w[0]; v[1] ^= (unsigned)w[1]; v[2] ^= (unsigned)w[2]; v[3] ^= (unsigned)w[3];
        } else if (KIND == 13) {  // SALU
#define X(i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc"); asm volatile("s_and_b32 %0, %0, %1" : "+s"(s2) : "s"(s3) : "scc");
            R8(X)
#undef X
        } else if (KIND == 14) {  // LDS returning atomic (conflict-free: [k][lane] dwords)
#define X(i) { unsigned addr = (threadIdx.x + 64 * ((v[i] >> 7) & 15)) * 4, r; asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr), "v"(1u) : "memory"); v[i] += r; }
            R8(X)
#undef X
        } else if (KIND == 15) {  // VALU + SALU interleaved 1:1 (do they co-issue?)
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7])); asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
            R16(X)
#undef X
        } else if (KIND == 16) {  // VALU -> SGPR -> SALU -> branch-free dependent use (ballot + s_and + s_cmp): the kernel's idiom
#define X(i) asm volatile("v_cmp_lt_u32_e64 %0, %1, %2\n\ts_and_b64 %0, %0, exec\n\ts_cmp_lg_u64 %0, 0\n\ts_cselect_b32 %3, 1, 0" : "=&s"(sm), "+v"(v[i]) : "v"(v[(i + 1) & 7]), "s"(s0) : "scc");
            R8(X)
#undef X
        } else if (KIND == 17) {  // v_bfi / 3-input logic
#define X(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 7]), "v"(v[(i + 2) & 7]));
            R16(X)
#undef X
        } else if (KIND == 18) {  // ds_read_u8 + ds_write_b8 pair
#define X(i) { unsigned addr = threadIdx.x * 68 + ((v[i] >> 7) & 63), r; asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory"); r += 1; asm volatile("ds_write_b8 %0, %1" :: "v"(addr), "v"(r) : "memory"); v[i] += r; }
            R8(X)
#undef X
        } else if (KIND == 19) {  // v_mbcnt pair
#define X(i) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, 0\n\tv_mbcnt_hi_u32_b32 %0, %2, %0" : "+v"(v[i]) : "s"(s1), "s"(s2));
            R8(X)
#undef X
        } else if (KIND == 20) {  // writelane
#define X(i) asm volatile("v_writelane_b32 %0, %1, 7" : "+v"(v[i]) : "s"(s1));
            R16(X)
#undef X
        } else if (KIND == 21) {  // 32 x 32 -> high 32 (the ratio key of the dp tiles, round 4)
#define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
            R16(X)
#undef X
        } else if (KIND == 22) {  // 24 x 24 -> low 32
#define X(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
            R16(X)
#undef X
        } else if (KIND == 23) {  // 24 x 24 -> high 16
#define X(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
            R16(X)
#undef X
        } else if (KIND == 24) {  // 32 x 32 -> low 32
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
            R16(X)
#undef X
        } else if (KIND == 25) {  // v_bcnt accumulate
#define X(i) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
            R16(X)
#undef X
        } else if (KIND == 26) {  // v_alignbit
#define X(i) asm volatile("v_alignbit_b32 %0, %0, %1, 13" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
            R16(X)
#undef X
        } else if (KIND == 27) {  // LDS gather at random dword addresses of a 64-entry table (the equal-word tables of sd_sift), 4 in flight
#define X(i) { unsigned addr = ((v[i] >> 9) & 63) * 4, r; asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(addr) : "memory"); v[i] += r + 0x9e3779b9u; }
            R8(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 28) {  // the same gather with lane-indexed (conflict-free) addresses
#define X(i) { unsigned addr = (threadIdx.x & 63) * 4, r; asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(addr) : "memory"); v[i] += r + 0x9e3779b9u; }
            R8(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 29) {  // ds_or_b32 (no return) at random entries of a 64-entry table
#define X(i) { unsigned addr = ((v[i] >> 9) & 63) * 4; asm volatile("ds_or_b32 %0, %1" :: "v"(addr), "v"(1u << (threadIdx.x & 31)) : "memory"); v[i] = v[i] * 1664525u + 1013904223u; }
            R8(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 30) {  // ds_bpermute_b32 from random lanes
#define X(i) { unsigned addr = ((v[i] >> 9) & 63) * 4, r; asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(r) : "v"(addr), "v"(v[(i + 1) & 7]) : "memory"); v[i] += r + 0x9e3779b9u; }
            R8(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 31) {  // ds_bpermute_b32 from the neighbouring lane (no two lanes read the same source)
#define X(i) { unsigned addr = ((threadIdx.x + 1) & 63) * 4, r; asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(r) : "v"(addr), "v"(v[(i + 1) & 7]) : "memory"); v[i] += r + 0x9e3779b9u; }
            R8(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 32) {  // ds_read_b64 gather at random 8-byte entries of a 64-entry table
#define X(i) { unsigned addr = ((v[i] >> 9) & 63) * 8; unsigned long long r; asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(addr) : "memory"); v[i] += (unsigned)r + (unsigned)(r >> 32) + 0x9e3779b9u; }
            R8(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 33) {  // ds_read_b128 gather at random 16-byte entries of a 64-entry table
#define X(i) { unsigned addr = ((v[i] >> 9) & 63) * 16; uint4 r; asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr) : "memory"); v[i] += r.x + r.y + r.z + r.w + 0x9e3779b9u; }
            R8(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 34) {  // v_dot4_u32_u8 (round 6: the bit planes of tf_scan)
#define X(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(v[i]) : "v"(v[(i + 1) & 7]), "s"(s1));
            R16(X)
#undef X
        } else if (KIND == 35) {  // v_bitop3_b32, two vector registers and a scalar one
#define X(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x90" : "+v"(v[i]) : "v"(v[(i + 1) & 7]), "s"(s1));
            R16(X)
#undef X
        } else if (KIND == 36) {  // v_dot4_u32_u8 with the accumulator chained (dependent pairs, as the planes make them)
#define X(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, 0\n\tv_dot4_u32_u8 %0, %3, %4, %0" : "+v"(v[i]) : "v"(v[(i + 1) & 7]), "s"(s1), "v"(v[(i + 2) & 7]), "s"(s2));
            R8(X)
#undef X
        }
    }
    const unsigned long long b = __builtin_amdgcn_s_memtime();
    unsigned acc = s0 ^ s2 ^ (unsigned)sm;
    for (int i = 0; i < 8; ++i) acc ^= v[i];
    if (acc == 0x12345u) sink[0] = acc + lds[threadIdx.x];
    if (threadIdx.x == 0) { t0[blockIdx.x] = a; t1[blockIdx.x] = b; }
}

struct Test { const char *name; int per_iter; void (*fn)(unsigned long long *, unsigned long long *, unsigned *, int); };

int main(int argc, char **argv)
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const Test tests[] = {
        {"v_add_u32", 16, k<0>}, {"v_and_b32", 16, k<1>}, {"v_lshlrev_b32", 16, k<2>}, {"v_bfe_u32", 16, k<3>}, {"v_perm_b32", 16, k<4>},
        {"v_alignbyte_b32", 16, k<5>}, {"v_mad_u32_u24", 16, k<6>}, {"v_cmp(vcc)+v_cndmask", 16, k<7>}, {"v_cmp_e64 -> sgpr", 16, k<8>},
        {"v_readlane_b32", 16, k<9>}, {"v_add_u32 dpp row_shr", 16, k<10>}, {"v_add_u32 sgpr operand", 16, k<11>}, {"v_lshlrev_b64", 16, k<12>},
        {"s_add+s_and (SALU)", 16, k<13>}, {"ds_add_rtn_u32 + wait + use", 8, k<14>}, {"v_add + s_add interleaved (pairs)", 16, k<15>},
        {"ballot idiom (v_cmp,s_and,s_cmp,s_cselect)", 8, k<16>}, {"v_and_or_b32", 16, k<17>}, {"ds_read_u8+wait+ds_write_b8", 8, k<18>},
        {"v_mbcnt lo+hi (pairs)", 8, k<19>}, {"v_writelane_b32", 16, k<20>},
        {"v_mul_hi_u32", 16, k<21>}, {"v_mul_u32_u24", 16, k<22>}, {"v_mul_hi_u32_u24", 16, k<23>}, {"v_mul_lo_u32", 16, k<24>}, {"v_bcnt_u32_b32", 16, k<25>},
        {"v_alignbit_b32", 16, k<26>}, {"ds_read_b32 random of 64 entries (8 in flight)", 8, k<27>}, {"ds_read_b32 lane-indexed (8 in flight)", 8, k<28>},
        {"ds_or_b32 random of 64 entries (8 in flight)", 8, k<29>}, {"ds_bpermute_b32 random lanes (8 in flight)", 8, k<30>},
        {"ds_bpermute_b32 neighbour lane (8 in flight)", 8, k<31>}, {"ds_read_b64 random of 64 entries", 8, k<32>}, {"ds_read_b128 random of 64 entries", 8, k<33>},
        {"v_dot4_u32_u8", 16, k<34>}, {"v_bitop3_b32 (v, v, s)", 16, k<35>}, {"v_dot4_u32_u8 chained pairs", 16, k<36>},
    };
    unsigned long long *t0, *t1; unsigned *sink;
    const int maxb = cus * 32;
    hipMalloc(&t0, maxb * 8); hipMalloc(&t1, maxb * 8); hipMalloc(&sink, 64);
    std::vector<unsigned long long> h0(maxb), h1(maxb);
    printf("%-44s", "cycles per wave-instruction per SIMD at waves/SIMD =");
    const int wps[] = {1, 2, 4, 5, 8};
    for (int w : wps) printf(" %6d", w);
    printf("\n");
    for (const Test &t : tests) {
        printf("%-44s", t.name);
        for (int w : wps) {
            const int nb = cus * 4 * w;
            t.fn<<<nb, 64>>>(t0, t1, sink, 1);   // warm
            hipDeviceSynchronize();
            t.fn<<<nb, 64>>>(t0, t1, sink, 1);
            hipDeviceSynchronize();
            hipMemcpy(h0.data(), t0, nb * 8, hipMemcpyDeviceToHost);
            hipMemcpy(h1.data(), t1, nb * 8, hipMemcpyDeviceToHost);
            // median per-wave duration: co-resident waves share the SIMD for (nearly) the whole of it
            std::vector<double> d(nb);
            for (int i = 0; i < nb; ++i) d[i] = (double)(h1[i] - h0[i]);
            std::sort(d.begin(), d.end());
            const double dur = d[nb / 2];
            printf(" %6.2f", dur / ((double)w * ITER * t.per_iter));
        }
        printf("\n");
    }
    return 0;
}
