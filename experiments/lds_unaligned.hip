// LDS reads of 16 bytes at a random BYTE address inside a 32 KiB window, 4 waves per SIMD (k_lzm's occupancy): the five aligned dwords + v_alignbit of rounds 2 - 4 against the
// unaligned ds_read_b128 / b64 / b32 that hipcc emits for byte-aligned pointers on gfx950 (diagnostic; not part of the library):
//   hipcc --offload-arch=gfx950 -O3 experiments/lds_unaligned.hip -o gpurun_out/lds_unaligned && gpurun_out/lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned u32u __attribute__((aligned(1)));
struct __attribute__((packed, aligned(1))) U2 { unsigned x, y; };
struct __attribute__((packed, aligned(1))) U3 { unsigned x, y, z; };
struct __attribute__((packed, aligned(1))) U4 { unsigned x, y, z, w; };
typedef const __attribute__((address_space(3))) unsigned lds_cu32;

// KIND 0: 5 aligned dwords (ds_read2_b32 x 2 + ds_read_b32) + 4 v_alignbit   1: unaligned b128   2: 2 x unaligned b64   3: 4 x unaligned b32   4: unaligned b96 + b32
// 5: KIND 0 on byte addresses that are multiples of 4 (no conflicts from the shift)   6: aligned b128 at multiples of 16    7: unaligned b64 only (8 bytes)  8: 3 aligned dwords + 2 alignbit (8 bytes)
template <int KIND, int DENS>
__global__ void k_lds(unsigned *o, unsigned n, unsigned *chk) {
    extern __shared__ unsigned char lds[];
    for (unsigned i = threadIdx.x; i < 8192 + 16; i += blockDim.x) ((unsigned *)lds)[i] = i * 2654435761u + (i >> 3);
    __syncthreads();
    unsigned addr = (threadIdx.x * 2654435761u) >> 17;              // byte address < 32768
    unsigned acc = 0;
    const bool act = ((threadIdx.x * 40503u) >> 8 & 127u) < (unsigned)DENS;   // DENS of 128 lanes take part
    for (unsigned i = 0; i < n; i++) {
        if (act) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            unsigned ad = (addr + r * 9777u) & 32767u;
            if (KIND == 5) ad &= ~3u;
            if (KIND == 6) ad &= ~15u;
            unsigned x0, x1, x2, x3;
            if (KIND == 0 || KIND == 5) {
                lds_cu32 *p = (lds_cu32 *)(size_t)(ad & ~3u);
                const unsigned d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4], sh = ad << 3;
                x0 = __builtin_amdgcn_alignbit(d1, d0, sh); x1 = __builtin_amdgcn_alignbit(d2, d1, sh); x2 = __builtin_amdgcn_alignbit(d3, d2, sh); x3 = __builtin_amdgcn_alignbit(d4, d3, sh);
            } else if (KIND == 1 || KIND == 6) { const U4 v = *(const U4 *)(lds + ad); x0 = v.x; x1 = v.y; x2 = v.z; x3 = v.w; }
            else if (KIND == 2) { const U2 a = *(const U2 *)(lds + ad), b = *(const U2 *)(lds + ad + 8); x0 = a.x; x1 = a.y; x2 = b.x; x3 = b.y; }
            else if (KIND == 3) { x0 = *(const u32u *)(lds + ad); x1 = *(const u32u *)(lds + ad + 4); x2 = *(const u32u *)(lds + ad + 8); x3 = *(const u32u *)(lds + ad + 12); }
            else if (KIND == 4) { const U3 a = *(const U3 *)(lds + ad); x0 = a.x; x1 = a.y; x2 = a.z; x3 = *(const u32u *)(lds + ad + 12); }
            else if (KIND == 7) { const U2 a = *(const U2 *)(lds + ad); x0 = a.x; x1 = a.y; x2 = x3 = 0; }
            else { lds_cu32 *p = (lds_cu32 *)(size_t)(ad & ~3u); const unsigned d0 = p[0], d1 = p[1], d2 = p[2], sh = ad << 3;
                   x0 = __builtin_amdgcn_alignbit(d1, d0, sh); x1 = __builtin_amdgcn_alignbit(d2, d1, sh); x2 = x3 = 0; }
            acc += x0 ^ (x1 * 3u) ^ (x2 * 5u) ^ (x3 * 7u);
        }
        }
        addr = (addr * 5u + acc + 1u) & 32767u;
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (blockIdx.x == 0 && threadIdx.x < 64) chk[threadIdx.x] = acc;
}
static unsigned *d_o, *d_chk;
template <int KIND, int DENS> static void run(const char *name) {
    const int k = 4, lds_bytes = (160 * 1024) / k - 512;
    const unsigned n = 4000;
    CK(hipFuncSetAttribute((const void *)k_lds<KIND, DENS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_lds<KIND, DENS>), dim3(256 * k), dim3(256), lds_bytes, 0, d_o, n / 50, d_chk);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_lds<KIND, DENS>), dim3(256 * k), dim3(256), lds_bytes, 0, d_o, n, d_chk);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned chk[64]; CK(hipMemcpy(chk, d_chk, sizeof chk, hipMemcpyDeviceToHost));
    unsigned h = 0; for (int i = 0; i < 64; i++) h = h * 31 + chk[i];
    printf("%-64s lanes %3d/128  %7.3f ms  %6.1f ns per 16-byte fetch of a wave, per CU   check %08x\n", name, DENS, ms, ms * 1e6 / ((double)k * 4 * n * 4), h);
}
int main() {
    CK(hipMalloc(&d_o, 256 * 8 * 256 * 4)); CK(hipMalloc(&d_chk, 256));
    run<0, 128>("5 aligned dwords + 4 alignbit (byte address)");
    run<1, 128>("unaligned ds_read_b128");
    run<2, 128>("2 x unaligned ds_read_b64");
    run<3, 128>("4 x unaligned ds_read_b32");
    run<4, 128>("unaligned ds_read_b96 + b32");
    run<5, 128>("5 aligned dwords + 4 alignbit (dword address)");
    run<6, 128>("aligned ds_read_b128 (16-byte address)");
    run<7, 128>("unaligned ds_read_b64 (8 bytes only)");
    run<8, 128>("3 aligned dwords + 2 alignbit (8 bytes only)");
    run<0, 70>("5 aligned dwords + 4 alignbit (byte address)");
    run<1, 70>("unaligned ds_read_b128");
    run<2, 70>("2 x unaligned ds_read_b64");
    run<3, 70>("4 x unaligned ds_read_b32");
    run<4, 70>("unaligned ds_read_b96 + b32");
    run<7, 70>("unaligned ds_read_b64 (8 bytes only)");
    run<8, 70>("3 aligned dwords + 2 alignbit (8 bytes only)");
    return 0;
}
