// Micro-benchmark: vector-memory pipeline cost of fetching one random 64-byte record per lane.
//   mode 0: each lane issues 4 x dwordx4 for its own record (what a per-lane BVH node fetch does)
//   mode 1: a quad of lanes fetches one record per instruction (lane j takes bytes 16j..16j+15), 4 instructions
//           cover the 4 records of the quad
//   mode 2: each lane issues 1 x dwordx4 (a quarter of the bytes of mode 0, same number of distinct lines)
//   mode 3: each lane issues 2 x dwordx4 from its record
// Prints cycles per wave-level load instruction per CU and records fetched per second.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }

template <int MODE>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ recs, uint32_t mask, int iters, float* out) {
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    const uint32_t lane = threadIdx.x & 63u, j = lane & 3u;
    const uint32_t sq = ((blockIdx.x * 256 + threadIdx.x) >> 2) * 2654435761u + 977u;
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        uint32_t idx = (lcg(s) >> 8) & mask;          // this lane's record
        if (MODE == 0) {
            const float4* p = recs + size_t(idx) * 4;
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t other = ((sq * 2246822519u + uint32_t(it * 4 + i) * 3266489917u) >> 8) & mask;   // same for the 4 lanes of a quad
                float4 a = recs[size_t(other) * 4 + j];
                acc += a.x + a.w;
            }
        } else if (MODE == 2) {
            float4 a = recs[size_t(idx) * 4];
            acc += a.x + a.w;
        } else if (MODE == 3) {
            const float4* p = recs + size_t(idx) * 4;
            float4 a = p[0], b = p[1];
            acc += a.x + b.y;
        } else if (MODE == 4) {          // 2 x dwordx4 at offsets 0 and 32
            const float4* p = recs + size_t(idx) * 4;
            float4 a = p[0], b = p[2];
            acc += a.x + b.y;
        } else if (MODE == 5) {          // 4 x dwordx4, only even lanes active
            if ((lane & 1u) == 0u) {
                const float4* p = recs + size_t(idx) * 4;
                float4 a = p[0], b = p[1], c = p[2], d = p[3];
                acc += a.x + b.y + c.z + d.w;
            }
        } else if (MODE == 6) {          // 4 x dwordx4, only lanes 0..31 active
            if (lane < 32u) {
                const float4* p = recs + size_t(idx) * 4;
                float4 a = p[0], b = p[1], c = p[2], d = p[3];
                acc += a.x + b.y + c.z + d.w;
            }
        } else if (MODE == 7) {          // 8 x dwordx2
            const float2* p = reinterpret_cast<const float2*>(recs + size_t(idx) * 4);
            float2 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6], h = p[7];
            acc += a.x + b.y + c.x + d.y + e.x + f.y + g.x + h.y;
        } else if (MODE == 12) {         // 4 x global_load_lds_dwordx4 (record gathered into LDS), then one ds_read_b128 back per quarter
            __shared__ float4 stage[4][4][64];
            const uint32_t w = threadIdx.x >> 6;
            const float4* p = recs + size_t(idx) * 4;
            typedef __attribute__((address_space(1))) const void* gptr_t;
            typedef __attribute__((address_space(3))) void* lptr_t;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)(p + q), (lptr_t)&stage[w][q][0], 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            float4 a = stage[w][0][lane], b = stage[w][1][lane], c = stage[w][2][lane], d = stage[w][3][lane];
            acc += a.x + b.y + c.z + d.w;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (MODE == 9) {          // 4 x dwordx4, chunk order rotated by lane
            const float4* p = recs + size_t(idx) * 4;
            float4 a = p[(0 + lane) & 3], b = p[(1 + lane) & 3], c = p[(2 + lane) & 3], d = p[(3 + lane) & 3];
            acc += a.x + b.y + c.z + d.w;
        } else if (MODE == 10) {         // 4 x dwordx4, chunk order rotated by lane>>1 (8 combos with bit 6)
            const float4* p = recs + size_t(idx) * 4;
            uint32_t r = lane >> 1;
            float4 a = p[(0 + r) & 3], b = p[(1 + r) & 3], c = p[(2 + r) & 3], d = p[(3 + r) & 3];
            acc += a.x + b.y + c.z + d.w;
        } else if (MODE == 11) {         // 2 x dwordx4 rotated
            const float4* p = recs + size_t(idx) * 4;
            float4 a = p[(0 + lane) & 3], b = p[(1 + lane) & 3];
            acc += a.x + b.y;
        } else {                         // 4 x dwordx4 from 128-byte-aligned records (index doubled)
            const float4* p = recs + size_t(idx & (mask >> 1)) * 8;
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
float run(const float4* recs, uint32_t mask, int grid, int iters, float* out) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, recs, mask, 4, out);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, recs, mask, iters, out);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int n_cu = p.multiProcessorCount, grid = n_cu * 8, iters = 4000;
    float4* recs; float* out;
    const size_t max_recs = 1u << 22;                      // 256 MiB of records
    (void)hipMalloc(&recs, max_recs * 64);
    (void)hipMemset(recs, 0, max_recs * 64);
    (void)hipMalloc(&out, size_t(grid) * 256 * 4);
    const char* names[] = {"4 x dwordx4 per lane", "quad-cooperative 4 x dwordx4", "1 x dwordx4 per lane", "2 x dwordx4 per lane", "2 x dwordx4 offsets 0,32",
                           "4 x dwordx4 even lanes only", "4 x dwordx4 lanes 0-31 only", "8 x dwordx2 per lane", "4 x dwordx4, 128-B aligned", "4 x dwordx4 rotated by lane", "4 x dwordx4 rotated by lane>>1", "2 x dwordx4 rotated by lane", "4 x global_load_lds_dwordx4 + ds_read"};
    const int n_instr[] = {4, 4, 1, 2, 2, 4, 4, 8, 4, 4, 4, 2, 4};
    for (uint32_t log2n : {8u, 14u}) {          // 64 KiB, 1 MiB, 16 MiB, 256 MiB working sets
        uint32_t mask = (1u << log2n) - 1u;
        float ms[13] = {run<0>(recs, mask, grid, iters, out), run<1>(recs, mask, grid, iters, out), run<2>(recs, mask, grid, iters, out), run<3>(recs, mask, grid, iters, out),
                       run<4>(recs, mask, grid, iters, out), run<5>(recs, mask, grid, iters, out), run<6>(recs, mask, grid, iters, out), run<7>(recs, mask, grid, iters, out), run<8>(recs, mask, grid, iters, out), run<9>(recs, mask, grid, iters, out), run<10>(recs, mask, grid, iters, out), run<11>(recs, mask, grid, iters, out), run<12>(recs, mask, grid, iters, out)};
        for (int m = 0; m < 13; ++m) {
            double wave_instr_per_cu = double(iters) * n_instr[m] * 32;          // 32 waves per CU
            double cyc = ms[m] * 1e-3 * 2.4e9 / wave_instr_per_cu;
            double recs_per_s = double(grid) * 256 * iters / (ms[m] * 1e-3);
            printf("records %8u (%6.1f MiB)  %-30s %8.3f ms  %6.1f cyc@2.4GHz per wave-load per CU  %7.1f G records/s\n", mask + 1, (mask + 1) * 64.0 / 1048576.0, names[m], ms[m], cyc, recs_per_s / 1e9);
        }
    }
    return 0;
}
