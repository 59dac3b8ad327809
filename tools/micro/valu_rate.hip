// Micro-benchmark: issue rate of scalar-f32 vs packed-f32 VALU instructions at full occupancy (8 waves/SIMD).
// Prints cycles per wave-instruction per SIMD for each variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {r0, r1}, p1 = {r2, r3}, p2 = {r4, r5}, p3 = {r6, r7}, pa = {a, b};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // 8 independent v_mul_f32 x16
            REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                               "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));)
        } else if (MODE == 1) {  // 4 independent v_pk_mul_f32 x16 (same number of f32 results as MODE 0 / 2)
            REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                               "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa));)
        } else if (MODE == 2) {  // min3/max3
            REP16(asm volatile("v_max3_f32 %0, %0, %8, %1\n v_min3_f32 %1, %1, %8, %2\n v_max3_f32 %2, %2, %8, %3\n v_min3_f32 %3, %3, %8, %4\n"
                               "v_max3_f32 %4, %4, %8, %5\n v_min3_f32 %5, %5, %8, %6\n v_max3_f32 %6, %6, %8, %7\n v_min3_f32 %7, %7, %8, %0\n"
                               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));)
        } else if (MODE == 3) {  // v_pk_add_f32
            REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                               "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa));)
        } else if (MODE == 4) {  // compare + select pairs
            REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %2, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %4, vcc\n"
                               "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %6, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %0, vcc\n"
                               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a) : "vcc");)
        } else if (MODE == 5) {  // v_sub + v_mul pairs (what the slab test uses)
            REP16(asm volatile("v_sub_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_mul_f32 %1, %1, %8\n"
                               "v_sub_f32 %2, %2, %8\n v_mul_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_mul_f32 %3, %3, %8\n"
                               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
double run(float* out, int blocks_per_cu, int n_cu, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(n_cu * blocks_per_cu), dim3(256), 0, 0, out, 8, 1.0001f, 0.9999f);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(n_cu * blocks_per_cu), dim3(256), 0, 0, out, iters, 1.0001f, 0.9999f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int n_cu = p.multiProcessorCount;
    double ghz = p.clockRate / 1e6;
    float* out;
    hipMalloc(&out, size_t(n_cu) * 8 * 256 * 4);
    const int iters = 2000;
    const char* names[] = {"v_mul_f32", "v_pk_mul_f32", "v_min3/max3_f32", "v_pk_add_f32", "v_cmp+v_cndmask", "v_sub+v_mul dependent pairs"};
    for (int bpc : {8, 4, 2, 1}) {
        double ms[6] = {run<0>(out, bpc, n_cu, iters), run<1>(out, bpc, n_cu, iters), run<2>(out, bpc, n_cu, iters),
                        run<3>(out, bpc, n_cu, iters), run<4>(out, bpc, n_cu, iters), run<5>(out, bpc, n_cu, iters)};
        for (int m = 0; m < 6; ++m) {
            double instr_per_wave = double(iters) * 16 * 8;
            double waves_per_simd = bpc;  // 4 waves per block, 4 SIMDs per CU
            double cyc = ms[m] * 1e-3 * ghz * 1e9 / (instr_per_wave * waves_per_simd);
            printf("waves/SIMD %d  %-28s %.3f ms  %.2f cycles per wave-instruction per SIMD (clock %.2f GHz)\n", bpc, names[m], ms[m], cyc, ghz);
        }
    }
    return 0;
}
