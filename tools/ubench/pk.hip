// Micro-benchmark: does packed fp32 (v_pk_mul_f32 / v_pk_add_f32) issue at the rate of one plain VALU instruction on gfx950?
// hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o pk pk.hip && ./pk
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int PK>
__global__ __launch_bounds__(256) void k(float *out, float a, float b, int iters)
{
    v2f x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = v2f{(float) threadIdx.x + i, (float) i};
    const v2f va = {a, a}, vb = {b, b};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (PK) { x[i] = x[i] * va; x[i] = x[i] + vb; }
                else {
                    float lo = x[i].x, hi = x[i].y;
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(lo), "v"(a));
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(hi), "v"(a));
                    asm volatile("v_add_f32 %0, %1, %2" : "=v"(lo) : "v"(lo), "v"(b));
                    asm volatile("v_add_f32 %0, %1, %2" : "=v"(hi) : "v"(hi), "v"(b));
                    x[i] = v2f{lo, hi};
                }
            }
        }
    }
    v2f s = {0, 0};
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

int main()
{
    float *d; hipMalloc(&d, 256 * 2048 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int pk = 0; pk < 2; pk++) for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (pk) k<1><<<2048, 256>>>(d, 1.0001f, 0.5f, iters); else k<0><<<2048, 256>>>(d, 1.0001f, 0.5f, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double lane_ops = 2048.0 * 256 * iters * 8 * 8 * 4;      // 2 mul + 2 add per element pair
        printf("%s: %.3f ms, %.1f T lane-ops/s\n", pk ? "packed" : "plain ", ms, lane_ops / ms * 1e-9);
    }
    return 0;
}
