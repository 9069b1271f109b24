// copy_proto.hip -- what a streaming read + write of 2 GiB + 2 GiB really sustains on this box, by access shape (round 4).
// The wire-format kernels can be no faster than a copy with their access shape; MI355X_MICROARCH.md quotes 6.29 TB/s for a
// float4 copy, the first probe here gave 5.4.  Sweep: 16-byte pieces per thread per trip (V), all loads of a trip issued
// before its stores; trips grid-strided over the whole buffer (S) or each block walking its own contiguous slice (C);
// plain or non-temporal; grid sized for K trips per thread or persistent (8 blocks per CU).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o build_tools/copy_proto tools/copy_proto.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int kBlk = 256;

template <bool NT>
__device__ __forceinline__ uint4 ld16(const uint4 *p) {
    if constexpr (NT) {
        uint4 v;
        v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y);
        v.z = __builtin_nontemporal_load(&p->z); v.w = __builtin_nontemporal_load(&p->w);
        return v;
    } else {
        return *p;
    }
}
template <bool NT>
__device__ __forceinline__ void st16(uint4 *p, uint4 v) {
    if constexpr (NT) {
        __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
        __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);
    } else {
        *p = v;
    }
}

// n16 = number of 16-byte pieces.  CONTIG: block b owns pieces [b * per_block, (b + 1) * per_block) and walks them
// kBlk * V at a time; else a trip of the whole grid covers gridDim * kBlk * V consecutive pieces.
template <int V, bool CONTIG, bool NT>
__global__ void __launch_bounds__(kBlk) k_copy(const uint4 *in, uint4 *out, size_t n16, size_t per_block) {
    size_t base, end, step;
    if constexpr (CONTIG) {
        base = (size_t)blockIdx.x * per_block;
        end = base + per_block < n16 ? base + per_block : n16;
        step = (size_t)kBlk * V;
    } else {
        base = (size_t)blockIdx.x * kBlk * V;
        end = n16;
        step = (size_t)gridDim.x * kBlk * V;
    }
    for (size_t i = base; i < end; i += step) {
        uint4 r[V];
#pragma unroll
        for (int v = 0; v < V; v++) {
            const size_t idx = i + (size_t)v * kBlk + threadIdx.x;
            if (idx < end) r[v] = ld16<NT>(in + idx);
        }
#pragma unroll
        for (int v = 0; v < V; v++) {
            const size_t idx = i + (size_t)v * kBlk + threadIdx.x;
            if (idx < end) st16<NT>(out + idx, r[v]);
        }
    }
}

struct Var {
    const char *name;
    void (*fn)(const uint4 *, uint4 *, size_t, size_t);
    int v;
    bool contig;
};
#define VV(V, C, NT) {"V" #V "/contig=" #C "/nt=" #NT, k_copy<V, C, NT>, V, C}

int main(int argc, char **argv) {
    const size_t bytes = (size_t)2 << 30, n16 = bytes / 16;
    uint4 *a, *b;
    CHECK(hipMalloc(&a, bytes + (1 << 20)));
    CHECK(hipMalloc(&b, bytes + (1 << 20)));
    CHECK(hipMemset(a, 1, bytes));
    CHECK(hipMemset(b, 2, bytes));
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<Var> vs = {VV(1, false, false), VV(2, false, false), VV(4, false, false), VV(8, false, false),
                           VV(2, true, false),  VV(4, true, false),  VV(8, true, false),  VV(2, false, true),
                           VV(4, false, true),  VV(4, true, true),   VV(8, true, true)};
    // out-buffer offsets: whether read and write streams meeting in the same channels matters
    for (size_t shift : {(size_t)0, (size_t)4096 + 256, (size_t)(1 << 19) + 4096}) {
        uint4 *dst = (uint4 *)((uint8_t *)b + shift);
        for (const Var &v : vs) {
            for (int trips : {1, 4, 16, 0}) {
                unsigned grid;
                size_t per_block = 0;
                if (trips == 0)
                    grid = (unsigned)cus * 8;
                else
                    grid = (unsigned)(n16 / ((size_t)kBlk * v.v * trips));
                if (v.contig) per_block = (n16 + grid - 1) / grid, per_block = (per_block + kBlk * v.v - 1) / (kBlk * v.v) * (kBlk * v.v);
                if (shift && !(trips == 4 || trips == 0)) continue;
                std::vector<float> ts;
                for (int rep = 0; rep < 6; rep++) {
                    CHECK(hipEventRecord(e0));
                    hipLaunchKernelGGL(v.fn, dim3(grid), dim3(kBlk), 0, 0, a, dst, n16, per_block);
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep) ts.push_back(ms);
                }
                std::sort(ts.begin(), ts.end());
                printf("shift %7zu %-20s trips %2d grid %8u  median %6.3f ms  %7.1f GB/s\n", shift, v.name, trips, grid, ts[2],
                       2.0 * bytes / (ts[2] * 1e-3) / 1e9);
            }
        }
    }
    return 0;
}
