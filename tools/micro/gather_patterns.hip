// Micro-benchmark: how fast can a CU gather 64-byte BVH-node-like records, one per lane, under the access shapes a
// traversal kernel can choose between. Complements tcp_rate.hip (independent loads) with
//   * dependent chains (the next record index comes out of the record just fetched, as in a tree walk),
//   * lanes sharing records (coherent rays visit the same node),
//   * 56-byte and 48-byte fetches from the 64-byte record,
//   * records resident in LDS (the hot top of a tree) with plain and lane-rotated chunk order,
//   * wave-uniform records fetched through the scalar cache,
//   * a mix: a fraction of the steps served from LDS, the rest from global memory.
// Prints cycles per wave-step per CU (2.4 GHz nominal) and records per second per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstring>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const v4f lds_f4;
__device__ __forceinline__ float4 ld(lds_f4* p) { v4f v = *p; return make_float4(v.x, v.y, v.z, v.w); }

__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }
__device__ __forceinline__ void eat(float4 v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }
__device__ __forceinline__ void eat2(float2 v) { asm volatile("" ::"v"(v.x), "v"(v.y)); }

enum Mode {
    G_IND_64 = 0,      // independent, per-lane random, 4 x dwordx4
    G_IND_56,          // independent, 3 x dwordx4 + 1 x dwordx2
    G_IND_48,          // independent, 3 x dwordx4
    G_IND_UNIFORM,     // independent, all lanes of the wave fetch the same record
    G_IND_SHARE4,      // independent, 4 adjacent lanes fetch the same record (same addresses)
    G_IND_SHARE16,     // independent, 16 adjacent lanes fetch the same record
    G_DEP_64,          // dependent chain, per-lane, 4 x dwordx4
    G_DEP_56,          // dependent chain, 3 x dwordx4 + dwordx2
    G_DEP_HALF,        // dependent chain, only even lanes active
    G_DEP_UNIFORM,     // dependent chain, all lanes share one chain
    L_IND_64,          // LDS table, independent, 4 x ds_read_b128
    L_IND_ROT,         // LDS table, chunk order rotated by lane
    L_DEP_64,          // LDS table, dependent chain
    L_DEP_ROT,         // LDS table, dependent chain, rotated chunks
    S_DEP_UNIFORM,     // dependent chain, wave-uniform, scalar loads (s_load_dwordx16)
    MIX_DEP_30,        // dependent chain, ~30 % of steps from the LDS table (rotated), the rest global
    MIX_DEP_50,        // dependent chain, ~50 % of steps from LDS
    N_MODES
};

constexpr uint32_t kLdsRecs = 1024;   // 64 KiB LDS table

template <int MODE>
__global__ __launch_bounds__(1024) void k(const float4* __restrict__ recs, uint32_t mask, int iters, float* out) {
    extern __shared__ v4f s_tab[];
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    constexpr bool kUsesLds = MODE >= L_IND_64 && MODE != S_DEP_UNIFORM;
    if (kUsesLds) {
        for (uint32_t i = threadIdx.x; i < kLdsRecs * 4; i += blockDim.x) { float4 v = recs[i]; s_tab[i] = v4f{v.x, v.y, v.z, v.w}; }
        __syncthreads();
    }
    lds_f4* tab = (lds_f4*)s_tab;
    uint32_t s = tid * 2654435761u + 12345u;
    uint32_t ws = (tid >> 6) * 2654435761u + 777u;                  // wave-uniform stream
    uint32_t idx = (lcg(s) >> 8) & mask;
    if (MODE == G_DEP_UNIFORM || MODE == S_DEP_UNIFORM) idx = (lcg(ws) >> 8) & mask;
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == G_IND_64 || MODE == G_IND_UNIFORM || MODE == G_IND_SHARE4 || MODE == G_IND_SHARE16) {
            uint32_t r;
            if (MODE == G_IND_64) r = (lcg(s) >> 8) & mask;
            else if (MODE == G_IND_UNIFORM) r = (lcg(ws) >> 8) & mask;
            else {
                uint32_t g = MODE == G_IND_SHARE4 ? (tid >> 2) : (tid >> 4);
                r = ((g * 2246822519u + uint32_t(it) * 3266489917u) >> 8) & mask;
            }
            const float4* p = recs + size_t(r) * 4;
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            eat(a); eat(b); eat(c); eat(d);
        } else if (MODE == G_IND_56) {
            uint32_t r = (lcg(s) >> 8) & mask;
            const float4* p = recs + size_t(r) * 4;
            float4 a = p[0], b = p[1], c = p[2];
            float2 d = *reinterpret_cast<const float2*>(p + 3);
            eat(a); eat(b); eat(c); eat2(d);
        } else if (MODE == G_IND_48) {
            uint32_t r = (lcg(s) >> 8) & mask;
            const float4* p = recs + size_t(r) * 4;
            float4 a = p[0], b = p[1], c = p[2];
            eat(a); eat(b); eat(c);
        } else if (MODE == G_DEP_64 || MODE == G_DEP_UNIFORM) {
            const float4* p = recs + size_t(idx) * 4;
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            eat(a); eat(b); eat(c);
            idx = (__float_as_uint(d.w) + uint32_t(it)) & mask;
        } else if (MODE == G_DEP_56) {
            const float4* p = recs + size_t(idx) * 4;
            float4 a = p[0], b = p[1], c = p[2];
            float2 d = *reinterpret_cast<const float2*>(p + 3);
            eat(a); eat(b); eat(c);
            idx = (__float_as_uint(d.y) + uint32_t(it)) & mask;
        } else if (MODE == G_DEP_HALF) {
            if ((lane & 1u) == 0u) {
                const float4* p = recs + size_t(idx) * 4;
                float4 a = p[0], b = p[1], c = p[2], d = p[3];
                eat(a); eat(b); eat(c);
                idx = (__float_as_uint(d.w) + uint32_t(it)) & mask;
            }
        } else if (MODE == L_IND_64 || MODE == L_IND_ROT) {
            uint32_t r = (lcg(s) >> 8) & (kLdsRecs - 1u);
            lds_f4* p = tab + r * 4;
            uint32_t rot = MODE == L_IND_ROT ? lane : 0u;
            float4 a = ld(p + ((0 + rot) & 3)), b = ld(p + ((1 + rot) & 3)), c = ld(p + ((2 + rot) & 3)), d = ld(p + ((3 + rot) & 3));
            eat(a); eat(b); eat(c); eat(d);
        } else if (MODE == L_DEP_64 || MODE == L_DEP_ROT) {
            lds_f4* p = tab + (idx & (kLdsRecs - 1u)) * 4;
            uint32_t rot = MODE == L_DEP_ROT ? lane : 0u;
            float4 a = ld(p + ((0 + rot) & 3)), b = ld(p + ((1 + rot) & 3)), c = ld(p + ((2 + rot) & 3)), d = ld(p + ((3 + rot) & 3));
            eat(a); eat(b); eat(c);
            // with rotation the "last" chunk sits in a lane-dependent register: pick it without branching
            uint32_t w0 = __float_as_uint(a.w), w1 = __float_as_uint(b.w), w2 = __float_as_uint(c.w), w3 = __float_as_uint(d.w);
            uint32_t k3 = (3u - rot) & 3u;     // which of a,b,c,d holds chunk 3
            uint32_t nxt = k3 == 0 ? w0 : (k3 == 1 ? w1 : (k3 == 2 ? w2 : w3));
            idx = (nxt + uint32_t(it)) & mask;
        } else if (MODE == S_DEP_UNIFORM) {
            uint32_t u = __builtin_amdgcn_readfirstlane(idx);
            const float4* p = recs + size_t(u) * 4;
            float4 a = p[0], b = p[1], c = p[2], d = p[3];      // uniform address: scalar loads
            acc += (a.x + a.y + a.z + a.w) * float(lane) + (b.x + b.y + b.z + b.w) + (c.x + c.y + c.z + c.w) * acc + d.x + d.y + d.z;               // use with per-lane data so values are needed in VALU
            idx = (__float_as_uint(d.w) + uint32_t(it)) & mask;
        } else if (MODE == MIX_DEP_30 || MODE == MIX_DEP_50) {
            // the record's own low index bits decide whether it is "hot" (LDS copy) or not
            const uint32_t thresh = MODE == MIX_DEP_30 ? 77u : 128u;    // of 256
            bool hot = ((idx * 2654435761u) >> 24) < thresh;
            float4 a, b, c, d;
            if (hot) {
                lds_f4* p = tab + (idx & (kLdsRecs - 1u)) * 4;
                a = ld(p + ((0 + lane) & 3)); b = ld(p + ((1 + lane) & 3)); c = ld(p + ((2 + lane) & 3)); d = ld(p + ((3 + lane) & 3));
                uint32_t k3 = (3u - lane) & 3u;
                uint32_t w0 = __float_as_uint(a.w), w1 = __float_as_uint(b.w), w2 = __float_as_uint(c.w), w3 = __float_as_uint(d.w);
                uint32_t nxt = k3 == 0 ? w0 : (k3 == 1 ? w1 : (k3 == 2 ? w2 : w3));
                idx = (nxt + uint32_t(it)) & mask;
            } else {
                const float4* p = recs + size_t(idx) * 4;
                a = p[0]; b = p[1]; c = p[2]; d = p[3];
                idx = (__float_as_uint(d.w) + uint32_t(it)) & mask;
            }
            eat(a); eat(b); eat(c);
        }
    }
    out[tid] = acc + float(idx);
}

struct Result { float ms; };

template <int MODE>
float run(const float4* recs, uint32_t mask, int grid, int block, int iters, float* out) {
    constexpr bool kUsesLds = MODE >= L_IND_64 && MODE != S_DEP_UNIFORM;
    size_t lds = kUsesLds ? kLdsRecs * 64 : 0;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(block), lds, 0, recs, mask, 8, out);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(block), lds, 0, recs, mask, iters, out);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { printf("mode %d: %s\n", MODE, hipGetErrorString(e)); return -1.0f; }
    return ms;
}

template <int MODE>
void report(const char* name, const float4* recs, uint32_t mask, int n_cu, int waves_per_cu, int iters, float* out, double lanes_frac = 1.0) {
    // workgroups of 1024 when the LDS table is used (2 x 64 KiB per CU at most), otherwise 256
    constexpr bool kUsesLds = MODE >= L_IND_64 && MODE != S_DEP_UNIFORM;
    int block = kUsesLds ? 1024 : 256;
    int grid = n_cu * waves_per_cu * 64 / block;
    float ms = run<MODE>(recs, mask, grid, block, iters, out);
    double steps_per_cu = double(iters) * waves_per_cu;
    double cyc = ms * 1e-3 * 2.4e9 / steps_per_cu;
    double recs_per_s_cu = steps_per_cu * 64.0 * lanes_frac / (ms * 1e-3);
    printf("table %7u recs  waves/CU %2d  %-44s %8.3f ms  %7.1f cyc per wave-step per CU  %6.3f G rec/s/CU\n", mask + 1, waves_per_cu, name, ms,
           cyc, recs_per_s_cu / 1e9);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int n_cu = p.multiProcessorCount, iters = 3000;
    const size_t max_recs = 1u << 18;                      // 16 MiB of records
    float4* recs; float* out;
    (void)hipMalloc(&recs, max_recs * 64);
    std::vector<uint32_t> h(max_recs * 16);
    uint32_t s = 99u;
    for (auto& w : h) { s = s * 1664525u + 1013904223u; w = s >> 4; }
    (void)hipMemcpy(recs, h.data(), max_recs * 64, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, size_t(n_cu) * 32 * 64 * 4);
    for (uint32_t log2n : {14u, 17u}) {          // 1 MiB (L2-resident) and 8 MiB (about the node table of a 262k-triangle scene)
        uint32_t mask = (1u << log2n) - 1u;
        for (int wpc : {32, 16}) {
            report<G_IND_64>("global ind 4x16B", recs, mask, n_cu, wpc, iters, out);
            report<G_IND_56>("global ind 3x16B+8B", recs, mask, n_cu, wpc, iters, out);
            report<G_IND_48>("global ind 3x16B", recs, mask, n_cu, wpc, iters, out);
            report<G_IND_UNIFORM>("global ind wave-uniform record", recs, mask, n_cu, wpc, iters, out);
            report<G_IND_SHARE4>("global ind 4 lanes share a record", recs, mask, n_cu, wpc, iters, out);
            report<G_IND_SHARE16>("global ind 16 lanes share a record", recs, mask, n_cu, wpc, iters, out);
            report<G_DEP_64>("global dep 4x16B", recs, mask, n_cu, wpc, iters, out);
            report<G_DEP_56>("global dep 3x16B+8B", recs, mask, n_cu, wpc, iters, out);
            report<G_DEP_HALF>("global dep 4x16B even lanes", recs, mask, n_cu, wpc, iters, out, 0.5);
            report<G_DEP_UNIFORM>("global dep wave-uniform", recs, mask, n_cu, wpc, iters, out);
            report<S_DEP_UNIFORM>("scalar dep wave-uniform", recs, mask, n_cu, wpc, iters, out);
            report<L_IND_64>("lds ind 4x16B", recs, mask, n_cu, wpc, iters, out);
            report<L_IND_ROT>("lds ind 4x16B rotated", recs, mask, n_cu, wpc, iters, out);
            report<L_DEP_64>("lds dep 4x16B", recs, mask, n_cu, wpc, iters, out);
            report<L_DEP_ROT>("lds dep 4x16B rotated", recs, mask, n_cu, wpc, iters, out);
            report<MIX_DEP_30>("mix dep 30% lds", recs, mask, n_cu, wpc, iters, out);
            report<MIX_DEP_50>("mix dep 50% lds", recs, mask, n_cu, wpc, iters, out);
        }
    }
    return 0;
}
