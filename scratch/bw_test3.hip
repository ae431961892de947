// Memory-pattern ceiling of the fused NN-TV pass: K images x (NR read + NW written) fields, 256-pixel chunks,
// no arithmetic. Variants: loads in flight per thread (BATCH images at once), waves per SIMD (LDS ballast), 1 or 2
// pixels per thread.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int K = 16, NR = 12, NW = 9;
template <int BATCH, int LDSKB, int MINW>
__global__ __launch_bounds__(256, MINW) void k_pat(int npix, int m, size_t fs, const double* __restrict__ in, double* __restrict__ out) {
    __shared__ double ballast[LDSKB * 128];
    if (threadIdx.x == 0) ballast[0] = 0.0;
    for (int base = blockIdx.x * 256; base < npix; base += gridDim.x * 256) {
        const unsigned off = 8u * (unsigned)(base + threadIdx.x + m);
#pragma unroll 1
        for (int k0 = 0; k0 < K; k0 += BATCH) {
            double v[BATCH][NR];
#pragma unroll
            for (int kk = 0; kk < BATCH; ++kk)
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    v[kk][r] = *(const double*)((const char*)(in + (size_t)(r * K + k0 + kk) * fs) + off);
#pragma unroll
            for (int kk = 0; kk < BATCH; ++kk) {
                double s = 0.0;
#pragma unroll
                for (int r = 0; r < NR; ++r) s += v[kk][r];
#pragma unroll
                for (int w = 0; w < NW; ++w)
                    *(double*)((char*)(out + (size_t)(w * K + k0 + kk) * fs) + off) = s + w;
            }
        }
    }
    if (ballast[0] == 1.5) out[0] = 1.0;
}
template <int TW, int LDSKB, int MINW>
__global__ __launch_bounds__(256, MINW) void k_pat2d(int m, int n, size_t fs, const double* __restrict__ in, double* __restrict__ out) {
    __shared__ double ballast[LDSKB * 128];
    if (threadIdx.x == 0) ballast[0] = 0.0;
    constexpr int TH = 256 / TW;
    const int tiles_i = m / TH, ntiles = tiles_i * (n / TW);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int ti = t % tiles_i, tj = t / tiles_i;
        const int i = ti * TH + (threadIdx.x % TH), j = tj * TW + (threadIdx.x / TH);
        const unsigned off = 8u * (unsigned)((j + 1) * m + i);
#pragma unroll 1
        for (int k = 0; k < K; ++k) {
            double v[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) v[r] = *(const double*)((const char*)(in + (size_t)(r * K + k) * fs) + off);
            double s = 0.0;
#pragma unroll
            for (int r = 0; r < NR; ++r) s += v[r];
#pragma unroll
            for (int w = 0; w < NW; ++w) *(double*)((char*)(out + (size_t)(w * K + k) * fs) + off) = s + w;
        }
    }
    if (ballast[0] == 1.5) out[0] = 1.0;
}
int main() {
    const int m = 4096, n = 4096, npix = m * n;
    const size_t fs = (size_t)m * (n + 2);
    double *in, *out;
    CHK(hipMalloc(&in, fs * NR * K * 8)); CHK(hipMalloc(&out, fs * NW * K * 8));
    CHK(hipMemset(in, 0, fs * NR * K * 8)); CHK(hipMemset(out, 0, fs * NW * K * 8));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const double bytes = 8.0 * (NR + NW) * K * (double)npix;
    auto run = [&](const char* name, auto launch) {
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-44s %.3f ms  %.0f GB/s\n", name, best, bytes / best / 1e6); fflush(stdout);
    };
#define RUN(B, L, W, G) run("batch " #B " lds " #L "KB minw " #W " grid " #G, [&] { hipLaunchKernelGGL((k_pat<B, L, W>), dim3(G), dim3(256), 0, 0, npix, m, fs, (const double*)in, out); });
    RUN(1, 70, 2, 512) RUN(1, 70, 2, 4096) RUN(1, 70, 2, 65536)
#define RUN2(TW, G) run("tile w" #TW " grid " #G, [&] { hipLaunchKernelGGL((k_pat2d<TW, 70, 2>), dim3(G), dim3(256), 0, 0, m, n, fs, (const double*)in, out); });
    RUN2(1, 512) RUN2(1, 4096) RUN2(1, 65536)
    RUN2(2, 512) RUN2(2, 4096) RUN2(2, 65536)
    RUN2(4, 512) RUN2(4, 4096) RUN2(4, 65536)
    RUN2(8, 4096) RUN2(8, 65536)
    CHK(hipDeviceSynchronize());
    return 0;
}
