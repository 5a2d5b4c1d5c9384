// Device-to-device copy laboratory: which form of a float4 streaming copy reaches the box's HBM ceiling
// (MI355X_MICROARCH.md quotes 6.29 TB/s).  Build + run on the GPU box: tools/gpu.sh lab copy_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_k(f32x4 *__restrict__ d, const f32x4 *__restrict__ s, long n4) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NTL ? __builtin_nontemporal_load(s + i + u * stride) : s[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NTS) __builtin_nontemporal_store(v[u], d + i + u * stride);
            else d[i + u * stride] = v[u];
        }
    }
    for (; i < n4; i += stride) d[i] = s[i];
}

// contiguous chunk per block (each block streams its own range; U loads in flight per lane)
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_chunk_k(f32x4 *__restrict__ d, const f32x4 *__restrict__ s, long n4) {
    const long per = (n4 + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
    long i = lo + threadIdx.x;
    for (; i + (U - 1) * 256 < hi; i += U * 256) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NTL ? __builtin_nontemporal_load(s + i + u * 256) : s[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NTS) __builtin_nontemporal_store(v[u], d + i + u * 256);
            else d[i + u * 256] = v[u];
        }
    }
    for (; i < hi; i += 256) d[i] = s[i];
}

template <class F>
static float time_ms(F f, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < reps; ++r) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const long n = 256l << 20;           // floats: 1 GiB
    float *s, *d;
    hipMalloc(&s, n * 4);
    hipMalloc(&d, n * 4);
    hipMemset(s, 1, n * 4);
    const long n4 = n / 4;
    const double gb = 2.0 * n * 4 / 1e9;
#define RUN(NAME, KERN, GRID)                                                                     \
    do {                                                                                          \
        float ms = time_ms([&] { hipLaunchKernelGGL(KERN, dim3(GRID), dim3(256), 0, 0, (f32x4 *)d, (const f32x4 *)s, n4); }, 10); \
        printf("%-44s grid %6d  %.3f ms  %.0f GB/s\n", NAME, (int)(GRID), ms, gb / ms * 1e3);     \
    } while (0)
    {
        float ms = time_ms([&] { hipMemcpyAsync(d, s, n * 4, hipMemcpyDeviceToDevice, 0); }, 10);
        printf("%-44s              %.3f ms  %.0f GB/s\n", "hipMemcpyAsync D2D", ms, gb / ms * 1e3);
    }
    for (int g : {256 * 4, 256 * 8, 256 * 16, 256 * 32, 256 * 64}) {
        RUN("strided U1 plain", (copy_k<1, false, false>), g);
        RUN("strided U4 plain", (copy_k<4, false, false>), g);
        RUN("strided U4 nt-load nt-store", (copy_k<4, true, true>), g);
        RUN("strided U4 plain-load nt-store", (copy_k<4, false, true>), g);
        RUN("strided U8 plain", (copy_k<8, false, false>), g);
        RUN("chunk   U4 plain", (copy_chunk_k<4, false, false>), g);
        RUN("chunk   U4 nt nt", (copy_chunk_k<4, true, true>), g);
        RUN("chunk   U8 plain", (copy_chunk_k<8, false, false>), g);
    }
    // one-shot: a thread per float4 (no loop)
    RUN("one float4 per thread, plain", (copy_k<1, false, false>), (int)(n4 / 256));
    RUN("one float4 per thread, nt nt", (copy_k<1, true, true>), (int)(n4 / 256));
    return 0;
}
