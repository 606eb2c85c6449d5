// mfma_f64_probe.hip — measures what the north-star's "MFMA for the batched X^T W X" alternative would cost on
// gfx950: issue rate of v_mfma_f64_16x16x4_f64 (the only fp64 matrix shape that takes a 16-wide tile) against the
// fp64 vector FMA rate, in the units the per-gene kernels care about (cycles per 64 samples of X^T W X / X^T dW X).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>

#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, double a0, double b0) {
    d4 acc[NACC];
    for (int k = 0; k < NACC; ++k) acc[k] = d4{0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
    }
    double s = 0.0;
    for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_fma(double* out, int iters, double a0, double b0) {
    double acc[NACC];
    for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = __builtin_fma(a, b, acc[k]);
        a += 1e-30;  // keep the loop body from being hoisted
    }
    double s = 0.0;
    for (int k = 0; k < NACC; ++k) s += acc[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate * 1e-6;
    double* out;
    const int blocks = cus * 8;  // 8 workgroups of 4 waves per CU: every SIMD holds 8 waves
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(double));
    const int iters = 20000;
    printf("device %s  CUs %d  clock %.2f GHz\n", p.gcnArchName, cus, ghz);
    // total instructions per SIMD = waves per SIMD (8) * iters * NACC
    auto report = [&](const char* name, int nacc, double ms, double flop_per_inst) {
        const double inst_per_simd = 8.0 * iters * nacc;
        const double cyc = ms * 1e-3 * ghz * 1e9 / inst_per_simd;
        const double tflops = flop_per_inst * (double)blocks * 4 * iters * nacc / (ms * 1e-3) / 1e12;
        printf("%-28s acc=%d  %.3f ms  %.2f cycles/instr/SIMD  %.1f TFLOP/s\n", name, nacc, ms, cyc, tflops);
    };
    report("v_mfma_f64_16x16x4_f64", 1, time_ms([&] { hipLaunchKernelGGL(k_mfma<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1.0); }), 2048.0);
    report("v_mfma_f64_16x16x4_f64", 4, time_ms([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1.0); }), 2048.0);
    report("v_fma_f64 (wave64)", 8, time_ms([&] { hipLaunchKernelGGL(k_fma<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1.0); }), 128.0);
    report("v_fma_f64 (wave64)", 16, time_ms([&] { hipLaunchKernelGGL(k_fma<16>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1.0); }), 128.0);
    printf("X^T W X + X^T dW X for 64 samples of one gene at p = 8:\n"
           "  vector path : 72 fp64 FMA + 16 mul per lane (each lane one sample, symmetric half only)\n"
           "  matrix path : 16 x v_mfma_f64_16x16x4 on [Xw ; Xdw]^T X (4 samples per instruction, 16 x 8 of the 16 x 16 tile used)\n"
           "  => cycles per 64 samples = 88 * (cycles per FMA)  vs  16 * (cycles per MFMA): see the measured rates above\n");
    hipFree(out);
    return 0;
}
