// Issue cost of the vector instructions the LZ kernels are made of, on one SIMD of gfx950 at 4 waves per SIMD (k_lzm's occupancy) with four independent
// chains per wave (diagnostic; not part of the library):
//   hipcc --offload-arch=gfx950 -O3 experiments/op_cost.hip -o gpurun_out/op_cost && gpurun_out/op_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define OPS(X) \
    X(0,  "v_add_u32 %0, %0, %1") \
    X(1,  "v_sub_u32 %0, %0, %1") \
    X(2,  "v_and_b32 %0, %0, %1") \
    X(3,  "v_or_b32 %0, %0, %1") \
    X(4,  "v_xor_b32 %0, %0, %1") \
    X(5,  "v_lshlrev_b32 %0, 3, %0") \
    X(6,  "v_lshrrev_b32 %0, %1, %0") \
    X(7,  "v_min_u32 %0, %0, %1") \
    X(8,  "v_max_u32 %0, %0, %1") \
    X(9,  "v_mov_b32 %0, %1") \
    X(10, "v_lshl_or_b32 %0, %0, 3, %1") \
    X(11, "v_lshl_add_u32 %0, %0, 3, %1") \
    X(12, "v_and_or_b32 %0, %0, %1, %1") \
    X(13, "v_or3_b32 %0, %0, %1, %1") \
    X(14, "v_add3_u32 %0, %0, %1, %1") \
    X(15, "v_bfe_u32 %0, %0, 3, 5") \
    X(16, "v_alignbit_b32 %0, %0, %1, %1") \
    X(17, "v_alignbyte_b32 %0, %0, %1, 1") \
    X(18, "v_perm_b32 %0, %0, %1, %1") \
    X(19, "v_min3_u32 %0, %0, %1, %1") \
    X(20, "v_med3_u32 %0, %0, %1, %1") \
    X(21, "v_ffbl_b32 %0, %0") \
    X(22, "v_ffbh_u32 %0, %0") \
    X(23, "v_bcnt_u32_b32 %0, %0, %1") \
    X(24, "v_mbcnt_lo_u32_b32 %0, %0, %1") \
    X(25, "v_mul_lo_u32 %0, %0, %1") \
    X(26, "v_mul_hi_u32 %0, %0, %1") \
    X(27, "v_mul_u32_u24 %0, %0, %1") \
    X(28, "v_mad_u32_u24 %0, %0, %1, %1") \
    X(29, "v_cmp_lt_u32 vcc, %0, %1") \
    X(30, "v_cndmask_b32 %0, %0, %1, vcc") \
    X(31, "v_cmp_lt_u32 s[20:21], %0, %1") \
    X(32, "v_cndmask_b32 %0, %0, %1, s[20:21]") \
    X(33, "v_add_u32 %0, %0, %1 clamp") \
    X(34, "v_add_u32_e64 %0, %0, %1") \
    X(35, "v_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1") \
    X(36, "v_add_u32_dpp %0, %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1") \
    X(37, "v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD") \
    X(38, "v_bitop3_b32 %0, %0, %1, %1 bitop3:0x48") \
    X(39, "v_xad_u32 %0, %0, %1, %1") \
    X(40, "v_sad_u32 %0, %0, %1, %1") \
    X(41, "v_sad_u8 %0, %0, %1, %1") \
    X(42, "v_lshlrev_b64 %0, 3, %0") \
    X(43, "v_pk_add_u16 %0, %0, %1") \
    X(44, "v_pk_min_u16 %0, %0, %1") \
    X(45, "v_pk_sub_u16 %0, %0, %1") \
    X(46, "v_cmp_eq_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc") \
    X(47, "v_add_u32 %0, %0, %1\n\tv_alignbit_b32 %0, %0, %1, %1") \
    X(48, "v_fma_f32 %0, %0, %1, %1") \
    X(49, "v_add_f32 %0, %0, %1") \
    X(50, "v_pk_add_f32 %0, %0, %0") \
    X(51, "v_readfirstlane_b32 s20, %0") \
    X(52, "s_nop 0")

template <int OP>
__global__ void k_op(unsigned *o, unsigned n) {
    extern __shared__ unsigned lds[];
    unsigned long long a[4];
    for (int i = 0; i < 4; i++) a[i] = threadIdx.x * 7 + i * 13 + 1;
    const unsigned b = threadIdx.x | 1;
    for (unsigned i = 0; i < n; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
#define X(ID, STR) if (OP == ID) { if (ID == 42 || ID == 50) asm volatile(STR : "+v"(a[c]) : "v"(b) : "vcc", "s20", "s21"); else asm volatile(STR : "+v"(*(unsigned *)&a[c]) : "v"(b) : "vcc", "s20", "s21"); }
                OPS(X)
#undef X
            }
        }
    }
    unsigned x = 0;
    for (int i = 0; i < 4; i++) x ^= (unsigned)a[i] ^ (unsigned)(a[i] >> 32);
    o[blockIdx.x * blockDim.x + threadIdx.x] = x + lds[threadIdx.x & 15];
}
static unsigned *d_o;
static double g_base = 0;
template <int OP> static void run(const char *name) {
    const int k = 4, lds_bytes = (160 * 1024) / k - 512;
    const unsigned n = 8000;
    CK(hipFuncSetAttribute((const void *)k_op<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_op<OP>, dim3(256 * k), dim3(256), lds_bytes, 0, d_o, n / 50);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_op<OP>, dim3(256 * k), dim3(256), lds_bytes, 0, d_o, n);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double ns = ms * 1e6 / ((double)k * n * 32);
    if (OP == 0) g_base = ns;
    printf("%2d %-100.100s %6.3f ns per wave-instruction(-group) per SIMD = %.2f x v_add_u32\n", OP, name, ns, ns / g_base);
}
int main() {
    CK(hipMalloc(&d_o, 256 * 8 * 256 * 4));
#define X(ID, STR) run<ID>(STR);
    OPS(X)
#undef X
    return 0;
}
