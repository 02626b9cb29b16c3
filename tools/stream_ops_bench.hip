// Wall time of short stream-operation sequences on one HIP stream (each ended by a
// hipStreamSynchronize): what the host side of a small call is made of.
//   hipcc -O2 --offload-arch=gfx950 tools/stream_ops_bench.hip -o stream_ops_bench && ./stream_ops_bench
// MI355X, ROCm 7.2 (round 4): kernel 11.9 us, each further kernel +2.9, a 32 KB pinned H2D or D2H
// copy around a kernel +8..9 each, a memset +2.7 (32 KB .. 8 MB alike), a kernel reading and
// writing 32 KB of mapped pinned host memory 13.3; the skeleton of a one-vector solve (H2D, 2
// memsets, kernel, 2 memsets, kernel, D2H) 46.9 with empty kernels.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k(double *p, size_t n)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n)
        p[i] += 1.0;
}
template <class F> double timeit(hipStream_t s, F f)
{
    for (int i = 0; i < 10; ++i) { f(); hipStreamSynchronize(s); }
    auto t0 = std::chrono::steady_clock::now();
    const int reps = 300;
    for (int i = 0; i < reps; ++i) { f(); hipStreamSynchronize(s); }
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
}
int main()
{
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const size_t n = 4096;
    double *d, *h;
    hipMalloc(&d, 1 << 20);
    hipHostMalloc(&h, 1 << 20);
    auto K = [&]() { hipLaunchKernelGGL(k, dim3(16), dim3(256), 0, s, d, n); };
    printf("kernel: %.1f us\n", timeit(s, [&]() { K(); }));
    printf("2 kernels: %.1f us\n", timeit(s, [&]() { K(); K(); }));
    printf("H2D 32K: %.1f us\n", timeit(s, [&]() { hipMemcpyAsync(d, h, 32768, hipMemcpyHostToDevice, s); }));
    printf("D2H 32K: %.1f us\n", timeit(s, [&]() { hipMemcpyAsync(h, d, 32768, hipMemcpyDeviceToHost, s); }));
    printf("H2D + kernel: %.1f us\n", timeit(s, [&]() { hipMemcpyAsync(d, h, 32768, hipMemcpyHostToDevice, s); K(); }));
    printf("kernel + D2H: %.1f us\n", timeit(s, [&]() { K(); hipMemcpyAsync(h, d, 32768, hipMemcpyDeviceToHost, s); }));
    printf("H2D + kernel + D2H: %.1f us\n", timeit(s, [&]() { hipMemcpyAsync(d, h, 32768, hipMemcpyHostToDevice, s); K(); hipMemcpyAsync(h, d, 32768, hipMemcpyDeviceToHost, s); }));
    printf("H2D + 2 memset + k + 2 memset + k + D2H: %.1f us\n", timeit(s, [&]() {
        hipMemcpyAsync(d, h, 32768, hipMemcpyHostToDevice, s);
        hipMemsetAsync(d + 8192, 0xFF, 229 << 10, s); hipMemsetAsync(d + 65536, 0xFF, 32768, s); K();
        hipMemsetAsync(d + 8192, 0xFF, 229 << 10, s); hipMemsetAsync(d + 65536, 0xFF, 32768, s); K();
        hipMemcpyAsync(h, d, 32768, hipMemcpyDeviceToHost, s); }));
    // device reads / writes the pinned buffer itself
    double *hd;
    hipHostGetDevicePointer((void **)&hd, h, 0);
    auto KH = [&]() { hipLaunchKernelGGL(k, dim3(16), dim3(256), 0, s, hd, n); };
    printf("kernel on mapped host memory (32K read + write): %.1f us\n", timeit(s, [&]() { KH(); }));
    printf("D2H 8 B: %.1f us\n", timeit(s, [&]() { hipMemcpyAsync(h, d, 8, hipMemcpyDeviceToHost, s); }));
    return 0;
}
