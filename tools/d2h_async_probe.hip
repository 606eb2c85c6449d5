// Does hipMemcpyAsync(device -> page-locked host) return before the stream reaches it?  (tools/: developer probe)
// Enqueue ~2 ms of kernel work, then a small D2H copy, and time the host side of each call.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(double* x, long iters) {
    double v = x[threadIdx.x];
    for (long i = 0; i < iters; ++i) v = v * 1.0000001 + 1e-9;
    x[threadIdx.x] = v;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    double *d, *h; hipMalloc(&d, 1 << 20); hipHostMalloc(&h, 1 << 20, hipHostMallocDefault);
    double* hp = (double*)malloc(1 << 20);
    for (int rep = 0; rep < 3; ++rep) {
        for (size_t bytes : {8ul, 8000ul, 60000ul, 1ul << 20}) {
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, d, 400000L); hipStreamSynchronize(st);
            double t0 = now();
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, d, 400000L);
            double t1 = now();
            hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st);
            double t2 = now();
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, d, 1000L);
            double t3 = now();
            hipStreamSynchronize(st);
            double t4 = now();
            printf("pinned  %8zu B: launch %.1f us, memcpyAsync call %.1f us, next launch %.1f us, sync %.1f us\n", bytes, t1 - t0, t2 - t1, t3 - t2, t4 - t3);
        }
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, d, 400000L);
        double t1 = now();
        hipMemcpyAsync(hp, d, 8000, hipMemcpyDeviceToHost, st);
        double t2 = now();
        hipStreamSynchronize(st);
        printf("pageable    8000 B: memcpyAsync call %.1f us\n", t2 - t1);
    }
    return 0;
}
