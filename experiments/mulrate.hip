// Issue rate of 32-bit integer multiplies against adds and 24-bit multiplies on gfx950 (diagnostic; not part of the library):
// hipcc --offload-arch=gfx950 -O3 experiments/mulrate.hip -o gpurun_out/mulrate && gpurun_out/mulrate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void k(unsigned *o, unsigned n) {
    unsigned a0 = threadIdx.x + 1, a1 = a0 + 7, a2 = a0 + 13, a3 = a0 + 29, a4 = a0 + 31, a5 = a0 + 37, a6 = a0 + 41, a7 = a0 + 43;
    for (unsigned i = 0; i < n; i++) {
#define STEP(a) if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(a0 | 1)); \
                else if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(a0 | 1)); \
                else if (OP == 2) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(a0 | 1)); \
                else asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(a0 | 1));
        STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
        STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int OP> static float run(unsigned *d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256 * 8), dim3(256), 0, 0, d, 100u);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(256 * 8), dim3(256), 0, 0, d, 20000u);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    unsigned *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    const char *names[4] = {"v_add_u32", "v_mul_lo_u32", "v_mul_u32_u24", "v_mul_hi_u32"};
    float t[4] = {run<0>(d), run<1>(d), run<2>(d), run<3>(d)};
    for (int i = 0; i < 4; i++) printf("%-14s %8.3f ms  x%.2f of add\n", names[i], t[i], t[i] / t[0]);
    return 0;
}
