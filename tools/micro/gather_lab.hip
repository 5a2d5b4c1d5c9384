// Ceiling of a random ROW gather on MI355X: Y[r] = sum_j X[col[r][j]] with 1 KiB rows (H = 256 fp32), uniformly random
// columns - the access pattern of the CSR SpMM on a graph without locality (BASELINE configs 2 and 3) with everything but
// the fetches stripped away (no values, no index arithmetic beyond one load, sums only to keep the fetches alive).
//   ./gather_lab            -> table: panel size (Infinity-Cache resident 100 MB / HBM resident 1 GB) x fetches in flight per
//                              wave x waves per CU -> TB/s of gathered rows
// What it answers: is ndcn's gather (C2: 7.0 TB/s of fabric traffic, C3: 5.5 TB/s) below what the memory system gives ANY
// kernel for this pattern?  The plateau of this table is that ceiling.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int INFLIGHT>
__global__ __launch_bounds__(256) void gather_kernel(const float *__restrict__ X, const int *__restrict__ col, float *__restrict__ Y,
                                                     int n_rows, int deg) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < n_rows; r += n_waves) {
        const int *c = col + (size_t)r * deg;
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < deg; j += INFLIGHT) {
            f4 v[INFLIGHT];
#pragma unroll
            for (int u = 0; u < INFLIGHT; ++u) {
                const int cj = c[j + u < deg ? j + u : deg - 1];
                v[u] = reinterpret_cast<const f4 *>(X + (size_t)cj * 256)[lane];
            }
#pragma unroll
            for (int u = 0; u < INFLIGHT; ++u)
                if (j + u < deg) acc += v[u];
        }
        __builtin_nontemporal_store(acc, reinterpret_cast<f4 *>(Y + (size_t)r * 256) + lane);
    }
}

template <int INFLIGHT>
static double run(const float *X, const int *col, float *Y, int n, int deg, int waves_per_cu) {
    const int blocks = 256 * waves_per_cu / 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(gather_kernel<INFLIGHT>, dim3(blocks), dim3(256), 0, 0, X, col, Y, n, deg);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gather_kernel<INFLIGHT>, dim3(blocks), dim3(256), 0, 0, X, col, Y, n, deg);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    // hot_rows / hot_pct (round 5, BASELINE config 3's last lever): hot_pct % of the indices fall into the first hot_rows rows - the
    // hub rows of a Barabasi-Albert graph packed in front by --layout degree (256 MB = the Infinity Cache's size): does the memory
    // system serve a skewed gather faster than a uniform one?
    struct Case { const char *name; int n, deg, hot_rows, hot_pct; } cases[] = {
        {"C2-like: 100k rows x 41 (X = 102 MB, Infinity-Cache resident)", 100000, 41, 0, 0},
        {"C3-like: 1M rows x 11 (X = 1 GB, HBM resident)", 1000000, 11, 0, 0},
        {"C3-like, 50 % of the gathers into the first 250k rows (256 MB)", 1000000, 11, 250000, 50},
        {"C3-like, 50 % of the gathers into the first 125k rows (128 MB)", 1000000, 11, 125000, 50},
        {"C3-like, 75 % of the gathers into the first 125k rows (128 MB)", 1000000, 11, 125000, 75},
        {"C3-like, 50 % of the gathers into the first 30k rows (31 MB: the L2s)", 1000000, 11, 30000, 50}};
    for (const Case &cs : cases) {
        float *X, *Y;
        int *col;
        hipMalloc(&X, (size_t)cs.n * 1024);
        hipMalloc(&Y, (size_t)cs.n * 1024);
        hipMalloc(&col, (size_t)cs.n * cs.deg * 4);
        hipMemset(X, 0, (size_t)cs.n * 1024);
        std::vector<int> h((size_t)cs.n * cs.deg);
        unsigned long long s = 88172645463325252ull;
        for (auto &v : h) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            const bool hot = cs.hot_rows > 0 && (int)((s >> 40) % 100) < cs.hot_pct;
            v = (int)((s & 0xffffffffffull) % (unsigned long long)(hot ? cs.hot_rows : cs.n));
        }
        hipMemcpy(col, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        const double gathered = (double)cs.n * cs.deg * 1024.0, written = (double)cs.n * 1024.0;
        printf("%s\n  gathered %.2f GB + written %.2f GB per launch\n", cs.name, gathered / 1e9, written / 1e9);
        printf("  %-18s %10s %10s %10s %10s\n", "waves per CU", "1 in fl.", "4 in fl.", "8 in fl.", "16 in fl.");
        for (int w : {4, 8, 16, 32}) {
            const double t1 = run<1>(X, col, Y, cs.n, cs.deg, w), t4 = run<4>(X, col, Y, cs.n, cs.deg, w),
                         t8 = run<8>(X, col, Y, cs.n, cs.deg, w), t16 = run<16>(X, col, Y, cs.n, cs.deg, w);
            printf("  %-18d %7.3f ms %7.3f ms %7.3f ms %7.3f ms   -> best %.2f TB/s gathered (+%.2f written)\n", w, t1, t4, t8, t16,
                   gathered / (1e9 * fmin(fmin(t1, t4), fmin(t8, t16))), written / (1e9 * fmin(fmin(t1, t4), fmin(t8, t16))));
        }
        hipFree(X); hipFree(Y); hipFree(col);
    }
    return 0;
}
