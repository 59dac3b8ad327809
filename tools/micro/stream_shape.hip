// What does a streaming kernel of the shade stage's shape reach on this chip?  R float4 input arrays -> W float4 output arrays,
// n records each, grid-stride loop, for several grid sizes (256-thread workgroups per CU). Prints GB/s per shape.
// usage: stream_shape [n_records_millions]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Ptrs { const float4* in[4]; float4* out[6]; };
template <int R, int W>
__global__ __launch_bounds__(256) void k_stream(Ptrs p, unsigned n) {
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float4 a = p.in[0][i];
        if (R > 1) { float4 b = p.in[1][i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        if (R > 2) { float4 b = p.in[2][i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        if (R > 3) { float4 b = p.in[3][i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
#pragma unroll
        for (int w = 0; w < W; ++w) p.out[w][i] = a;
    }
}
template <int R, int W>
int run(Ptrs p, unsigned n, int blocks_per_cu, int n_cu) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const unsigned grid = blocks_per_cu > 0 ? unsigned(blocks_per_cu * n_cu) : (n + 255) / 256;
    hipLaunchKernelGGL((k_stream<R, W>), dim3(grid), dim3(256), 0, 0, p, n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k_stream<R, W>), dim3(grid), dim3(256), 0, 0, p, n);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= 5;
    printf("  reads %d writes %d  blocks/CU %3d (grid %8u): %7.3f ms  %7.1f GB/s\n", R, W, blocks_per_cu, grid, ms, double(n) * 16.0 * (R + W) / ms / 1e6);
    return 0;
}
int main(int argc, char** argv) {
    const unsigned n = unsigned((argc > 1 ? atof(argv[1]) : 40.0) * 1e6);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    Ptrs p;
    for (int k = 0; k < 4; ++k) { void* q; CK(hipMalloc(&q, size_t(n) * 16)); CK(hipMemset(q, 0, size_t(n) * 16)); p.in[k] = (const float4*)q; }
    for (int k = 0; k < 6; ++k) { void* q; CK(hipMalloc(&q, size_t(n) * 16)); p.out[k] = (float4*)q; }
    printf("%u records per array, %d CUs\n", n, n_cu);
    for (int bpc : {2, 4, 8, 16, 0}) {
        if (run<4, 0>(p, n, bpc, n_cu)) return 1;
        if (run<4, 3>(p, n, bpc, n_cu)) return 1;
        if (run<4, 6>(p, n, bpc, n_cu)) return 1;
        if (run<1, 1>(p, n, bpc, n_cu)) return 1;
    }
    return 0;
}
