// What one SIMD of gfx950 issues per cycle, by waves per SIMD and by the independent chains inside a wave (diagnostic; not part of the library):
//   hipcc --offload-arch=gfx950 -O3 experiments/issue_rate.hip -o gpurun_out/issue_rate && gpurun_out/issue_rate
// Every workgroup is 256 threads = one wave per SIMD of its CU; dynamic LDS of 160 KiB / k lets exactly k workgroups share a CU, so a grid of
// 256 k workgroups runs k waves per SIMD.  Cycles are s_memtime ticks converted with the measured ratio to the wall clock (100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// OP 0: v_add_u32   1: v_alignbit_b32   2: v_cmp + v_cndmask (two instructions)   3: v_min3_u32   4: v_mul_lo_u32  5: v_ffbl (inline asm)  6: v_perm_b32
template <int OP, int ILP>
__global__ void k_valu(unsigned *o, unsigned n, unsigned long long *cyc) {
    extern __shared__ unsigned lds[];
    unsigned a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 7 + i * 13 + 1;
    const unsigned b = threadIdx.x | 1;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (unsigned i = 0; i < n; i++) {
#pragma unroll
        for (int r = 0; r < 16 / ILP; r++) {
#pragma unroll
            for (int c = 0; c < ILP; c++) {
                if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
                else if (OP == 1) asm volatile("v_alignbit_b32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(b));
                else if (OP == 2) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(b) : "vcc");
                else if (OP == 3) asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(b));
                else if (OP == 4) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
                else if (OP == 5) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a[c]));
                else asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(b));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned x = 0;
    for (int i = 0; i < 8; i++) x ^= a[i];
    o[blockIdx.x * blockDim.x + threadIdx.x] = x + lds[threadIdx.x & 15];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// LDS reads at random dword-aligned addresses inside a 32 KiB window (the match step's pattern).  KIND 0: ds_read_b32, 1: ds_read2_b32 (8 bytes),
// 2: four consecutive dwords as 2 x ds_read2_b32, 3: ds_read_b128 at a 16-byte aligned random address
template <int KIND>
__global__ void k_lds(unsigned *o, unsigned n, unsigned long long *cyc) {
    extern __shared__ unsigned lds[];
    for (unsigned i = threadIdx.x; i < 8192 + 16; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    unsigned addr = (threadIdx.x * 2654435761u) >> 19;              // dword index < 8192
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (unsigned i = 0; i < n; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const unsigned ad = (addr + r * 977u) & 8191u;
            if (KIND == 0) acc += lds[ad];
            else if (KIND == 1) { acc += lds[ad] ^ lds[ad + 1]; }
            else if (KIND == 2) { acc += lds[ad] ^ lds[ad + 1] ^ lds[ad + 2] ^ lds[ad + 3]; }
            else if (KIND == 3) { const uint4 v = *(const uint4 *)&lds[ad & ~3u]; acc += v.x ^ v.y ^ v.z ^ v.w; }
        }
        addr = (addr * 5u + acc) & 8191u;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    o[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

static unsigned *d_o; static unsigned long long *d_c;
template <typename K> static void run(const char *name, K kern, int k, unsigned n, double per_iter) {
    const int lds_bytes = (160 * 1024) / k - 512;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(256 * k), dim3(256), lds_bytes, 0, d_o, n / 50, d_c);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(256 * k), dim3(256), lds_bytes, 0, d_o, n, d_c);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long cyc = 0; CK(hipMemcpy(&cyc, d_c, 8, hipMemcpyDeviceToHost));
    // per SIMD: k waves x n x per_iter instructions in `ms`
    const double inst = (double)k * n * per_iter;
    printf("%-34s waves/SIMD %d  %8.3f ms  %6.2f ns per wave-instruction per SIMD  (%.2f cycles at 2.4 GHz; counter ticks per instruction of one wave %.2f)\n",
           name, k, ms, ms * 1e6 / inst, ms * 1e6 / inst * 2.4, (double)cyc / ((double)n * per_iter));
}

int main() {
    CK(hipMalloc(&d_o, 256 * 8 * 256 * 4)); CK(hipMalloc(&d_c, 8));
    const unsigned n = 20000;
    for (int k : {1, 2, 4, 8}) {
        run("v_add_u32 ILP1", k_valu<0, 1>, k, n, 16);
        run("v_add_u32 ILP4", k_valu<0, 4>, k, n, 16);
        run("v_add_u32 ILP8", k_valu<0, 8>, k, n, 16);
    }
    for (int k : {4, 8}) {
        run("v_alignbit_b32 ILP4", k_valu<1, 4>, k, n, 16);
        run("v_cmp+v_cndmask ILP4 (2 inst)", k_valu<2, 4>, k, n, 32);
        run("v_min3_u32 ILP4", k_valu<3, 4>, k, n, 16);
        run("v_mul_lo_u32 ILP4", k_valu<4, 4>, k, n, 16);
        run("v_ffbl_b32 ILP4", k_valu<5, 4>, k, n, 16);
        run("v_perm_b32 ILP4", k_valu<6, 4>, k, n, 16);
    }
    for (int k : {1, 4}) {
        run("ds_read_b32 random", k_lds<0>, k, 4000, 8);
        run("ds_read2_b32 random (8 B)", k_lds<1>, k, 4000, 8);
        run("2 x ds_read2_b32 random (16 B)", k_lds<2>, k, 4000, 8);
        run("ds_read_b128 random aligned", k_lds<3>, k, 4000, 8);
    }
    return 0;
}
