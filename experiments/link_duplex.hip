// link_duplex.hip -- what the host link of an MI355X box gives to the two directions at once (round 3).
// Builds stand-alone: hipcc --offload-arch=gfx950 -O2 -o link_duplex link_duplex.hip -lpthread
// Cases: H2D and D2H with the copy engines (hipMemcpyAsync from / to page-locked memory), alone and together on two streams; the same
// with ONE direction done by a kernel that reads / writes the page-locked host buffer through its device mapping (so that the two
// directions cannot meet in one SDMA engine); the pageable -> page-locked staging copy on T host threads, alone and next to the link traffic.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
using clk = std::chrono::steady_clock;
static double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }

__global__ void k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main(int argc, char **argv) {
    const size_t MB = argc > 1 ? (size_t)atol(argv[1]) : 1024;
    const size_t N = MB << 20;
    const int reps = 5;
    uint8_t *h_in, *h_out, *d_a, *d_b;
    CK(hipHostMalloc((void **)&h_in, N, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&h_out, N, hipHostMallocDefault));
    CK(hipMalloc((void **)&d_a, N)); CK(hipMalloc((void **)&d_b, N));
    memset(h_in, 1, N); memset(h_out, 2, N);
    CK(hipMemset(d_a, 3, N)); CK(hipMemset(d_b, 4, N));
    uint8_t *m_in, *m_out;                                       // device views of the host buffers
    CK(hipHostGetDevicePointer((void **)&m_in, h_in, 0)); CK(hipHostGetDevicePointer((void **)&m_out, h_out, 0));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    auto run = [&](const char *name, auto f1, auto f2, double bytes1, double bytes2) {
        double best = 1e9;
        for (int r = 0; r < reps; r++) {
            CK(hipDeviceSynchronize());
            const auto t0 = clk::now();
            f1(); f2();
            CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
            best = std::min(best, secs(t0, clk::now()));
        }
        printf("%-64s %7.2f ms  in %6.1f GB/s  out %6.1f GB/s  sum %6.1f GB/s\n", name, best * 1e3, bytes1 / best / 1e9, bytes2 / best / 1e9, (bytes1 + bytes2) / best / 1e9);
        fflush(stdout);
    };
    auto none = [] {};
    auto h2d = [&] { CK(hipMemcpyAsync(d_a, h_in, N, hipMemcpyHostToDevice, s1)); };
    auto d2h = [&] { CK(hipMemcpyAsync(h_out, d_b, N, hipMemcpyDeviceToHost, s2)); };
    auto d2h_half = [&] { CK(hipMemcpyAsync(h_out, d_b, N * 3 / 8, hipMemcpyDeviceToHost, s2)); };
    auto k_d2h = [&] { hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, s2, (const uint4 *)d_b, (uint4 *)m_out, N / 16); };
    auto k_d2h_half = [&] { hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, s2, (const uint4 *)d_b, (uint4 *)m_out, N * 3 / 8 / 16); };
    auto k_h2d = [&] { hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, s1, (const uint4 *)m_in, (uint4 *)d_a, N / 16); };
    printf("buffers of %zu MiB, best of %d\n", MB, reps);
    run("H2D copy engine alone", h2d, none, (double)N, 0);
    run("D2H copy engine alone", none, d2h, 0, (double)N);
    run("H2D || D2H, copy engines, two streams", h2d, d2h, (double)N, (double)N);
    run("H2D || D2H of 3/8 the size (the archive's share), copy engines", h2d, d2h_half, (double)N, (double)N * 3 / 8);
    run("D2H by kernel stores into mapped host memory, alone", none, k_d2h, 0, (double)N);
    run("H2D by kernel loads from mapped host memory, alone", k_h2d, none, (double)N, 0);
    run("H2D copy engine || D2H by kernel stores", h2d, k_d2h, (double)N, (double)N);
    run("H2D copy engine || D2H (3/8) by kernel stores", h2d, k_d2h_half, (double)N, (double)N * 3 / 8);
    run("H2D by kernel loads || D2H copy engine", k_h2d, d2h, (double)N, (double)N);
    run("H2D by kernel loads || D2H by kernel stores", k_h2d, k_d2h, (double)N, (double)N);
    // D2H of 3/8 the size by kernel stores with a limited grid next to a full H2D on the copy engine: each side's own duration
    {
        hipEvent_t a1, b1, a2, b2; CK(hipEventCreate(&a1)); CK(hipEventCreate(&b1)); CK(hipEventCreate(&a2)); CK(hipEventCreate(&b2));
        for (int G : {2, 4, 8, 16, 32, 64, 256, 1024}) {
            float t1 = 1e9f, t2 = 1e9f;
            for (int r = 0; r < reps; r++) {
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(a1, s1)); CK(hipMemcpyAsync(d_a, h_in, N, hipMemcpyHostToDevice, s1)); CK(hipEventRecord(b1, s1));
                CK(hipEventRecord(a2, s2)); hipLaunchKernelGGL(k_copy, dim3(G), dim3(256), 0, s2, (const uint4 *)d_b, (uint4 *)m_out, N * 3 / 8 / 16); CK(hipEventRecord(b2, s2));
                CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
                float x1, x2; CK(hipEventElapsedTime(&x1, a1, b1)); CK(hipEventElapsedTime(&x2, a2, b2));
                t1 = std::min(t1, x1); t2 = std::min(t2, x2);
            }
            printf("H2D copy engine (1x) || D2H kernel stores (3/8x) with %4d workgroups: H2D %6.2f ms = %5.1f GB/s, D2H %6.2f ms = %5.1f GB/s\n", G, t1, N / t1 / 1e6, t2, N * 3.0 / 8 / t2 / 1e6);
            fflush(stdout);
        }
    }
    // the staging copy (pageable -> page-locked) on T threads, alone and next to an H2D + D2H pair
    uint8_t *pg = (uint8_t *)malloc(N); memset(pg, 5, N);
    for (unsigned T : {4u, 8u, 16u}) {
        auto stage = [&] { std::vector<std::thread> th; for (unsigned t = 0; t < T; t++) th.emplace_back([&, t] { const size_t a = N * t / T, b = N * (t + 1) / T; memcpy(h_in + a, pg + a, b - a); }); for (auto &x : th) x.join(); };
        double best = 1e9;
        for (int r = 0; r < reps; r++) { const auto t0 = clk::now(); stage(); best = std::min(best, secs(t0, clk::now())); }
        printf("staging memcpy pageable -> page-locked, %2u threads, alone            %7.2f ms  %6.1f GB/s\n", T, best * 1e3, N / best / 1e9);
        best = 1e9; double bl = 0;
        for (int r = 0; r < reps; r++) {
            CK(hipDeviceSynchronize());
            const auto t0 = clk::now();
            CK(hipMemcpyAsync(d_a, h_out, N, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(h_out + 0, d_b, 0, hipMemcpyDeviceToHost, s2));
            stage();
            const double ts = secs(t0, clk::now());
            CK(hipStreamSynchronize(s1));
            const double tl = secs(t0, clk::now());
            if (ts < best) { best = ts; bl = tl; }
        }
        printf("  ... next to an H2D copy of the same size                            %7.2f ms  %6.1f GB/s  (the H2D copy took %.2f ms = %.1f GB/s)\n", best * 1e3, N / best / 1e9, bl * 1e3, N / bl / 1e9);
        fflush(stdout);
    }
    return 0;
}
