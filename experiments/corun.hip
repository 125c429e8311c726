// Can an LDS-free, vector-bound kernel live on the issue slots the match kernel leaves idle?  A stand-in for "k_lzp without LDS": workgroups of one wave, no LDS,
// `regs` live registers, a stream of integer instructions (half of them of the slow kind) with four independent chains, `iters` x 64 instructions per wave.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC experiments/corun.hip -o build/libcorun.so ; scripts/corun.py runs it beside the library's LZ stage.
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(64) void k_spin(unsigned *o, unsigned iters) {
    unsigned a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 7 + i * 13 + blockIdx.x;
    const unsigned b = threadIdx.x | 1;
    for (unsigned i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                asm volatile("v_add_u32 %0, %0, %1\n\tv_alignbit_b32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(b));
            }
        }
    }
    unsigned x = 0;
    for (int i = 0; i < 8; i++) x ^= a[i];
    if (x == 0x12345678u) o[0] = x;
}
extern "C" int corun_launch(void *stream, unsigned wgs, unsigned iters, unsigned *d_out) {
    hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(64), 0, (hipStream_t)stream, d_out, iters);
    return (int)hipGetLastError();
}
