// fp64 issue-rate microbenchmark for gfx950: v_mfma_f64_16x16x4_f64, v_mfma_f64_4x4x4_4b_f64 and
// v_fma_f64, each as chains of independent accumulators long enough to hide the pipeline
// latency.  Prints achieved TFLOP/s for the whole chip (every CU, 4 waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o build/fp64_peak scripts/microbench/fp64_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double v4d __attribute__((ext_vector_type(4)));

#define CHK(x)                                                                     \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma16(double *out, int iters, double a0, double b0)
{
    v4d acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = {0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma4(double *out, int iters, double a0, double b0)
{
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void __launch_bounds__(256) k_fma(double *out, int iters, double a0, double b0)
{
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    const double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the same chain without contraction: v_mul_f64 + v_add_f64, what -ffp-contract=off (the
// reference's arithmetic) issues for a * b + c
template <int NACC>
__global__ void __launch_bounds__(256) k_muladd(double *out, int iters, double a0, double b0)
{
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    const double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            double m;
            asm volatile("v_mul_f64 %0, %1, %2" : "=v"(m) : "v"(acc[i]), "v"(a));
            asm volatile("v_add_f64 %0, %1, %2" : "=v"(acc[i]) : "v"(m), "v"(b));
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
static double run(K kern, double *out, int blocks, int iters, double flop_per_thread_iter,
                  const char *name)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters / 10, 1.0000001, 1e-9);
    CHK(hipDeviceSynchronize());
    double best = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9);
        CHK(hipEventRecord(e1, 0));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        const double tf = flop_per_thread_iter * iters * blocks * 256.0 / (ms * 1e-3) / 1e12;
        if (tf > best) best = tf;
    }
    printf("{\"kernel\": \"%s\", \"TFLOPs\": %.2f}\n", name, best);
    return best;
}

int main()
{
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const int blocks = cus * 4;  // 4 blocks of 4 waves per CU: 4 waves per SIMD
    double *out;
    CHK(hipMalloc(&out, sizeof(double) * blocks * 256));
    printf("{\"device\": \"%s\", \"CUs\": %d, \"clock_MHz\": %d}\n", p.gcnArchName, cus, p.clockRate / 1000);
    const int iters = 20000;
    // per wave instruction: 16x16x4 MFMA = 2*16*16*4 = 2048 flop = 32 per lane;
    // 4x4x4 x 4 blocks = 512 flop = 8 per lane; FMA = 2 per lane; mul + add = 2 per lane
    run(k_mfma16<4>, out, blocks, iters, 4 * 32.0, "v_mfma_f64_16x16x4_f64");
    run(k_mfma4<8>, out, blocks, iters, 8 * 8.0, "v_mfma_f64_4x4x4_4b_f64");
    run(k_fma<16>, out, blocks, iters, 16 * 2.0, "v_fma_f64");
    run(k_muladd<16>, out, blocks, iters, 16 * 2.0, "v_mul_f64+v_add_f64 (no contraction)");
    CHK(hipFree(out));
    return 0;
}
