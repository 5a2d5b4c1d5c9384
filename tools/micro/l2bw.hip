// Microbenchmark: aggregate rate of 1 KiB-per-wave row fetches (the gather's access shape) out of L2 / MALL / HBM.
// Every wave walks rows of a region of `region_mb` MiB with a per-wave stride pattern, 8 fetches in flight.
//   hipcc --offload-arch=gfx950 -O3 -o l2bw l2bw.hip && ./l2bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void walk(const f32x4 *__restrict__ base, unsigned n_rows, int iters, float *out) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * WAVES + (threadIdx.x >> 6);
    unsigned r = (wave * 2654435761u) % n_rows;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
        f32x4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            v[q] = base[(size_t)r * 64 + lane];
            r += 977;                       // a stride that walks the whole region
            if (r >= n_rows) r -= n_rows;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += v[q];
    }
    if (acc.x == 1.2345f) out[wave] = acc.y;
}

int main() {
    const size_t max_bytes = 2048ull << 20;
    float *buf, *out;
    hipMalloc(&buf, max_bytes);
    hipMalloc(&out, 1 << 20);
    hipMemset(buf, 0, max_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int region_mb[] = {1, 2, 3, 8, 32, 128, 1024, 2048};
    for (int waves : {4, 8, 16}) {
        for (int mb : region_mb) {
            const unsigned n_rows = (unsigned)(((size_t)mb << 20) / 1024);
            const int iters = 2000;
            auto launch = [&]() {
                if (waves == 4) hipLaunchKernelGGL(walk<4>, dim3(256), dim3(256), 0, 0, (const f32x4 *)buf, n_rows, iters, out);
                else if (waves == 8) hipLaunchKernelGGL(walk<8>, dim3(256), dim3(512), 0, 0, (const f32x4 *)buf, n_rows, iters, out);
                else hipLaunchKernelGGL(walk<16>, dim3(256), dim3(1024), 0, 0, (const f32x4 *)buf, n_rows, iters, out);
            };
            launch();
            hipDeviceSynchronize();
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double bytes = 256.0 * waves * iters * 8 * 1024;
            printf("waves/CU %2d region %5d MiB: %.3f ms  %.2f TB/s\n", waves, mb, ms, bytes / ms / 1e9);
        }
    }
    return 0;
}
