// Micro-benchmark: what HBM rate does the dual pass' access pattern admit, independent of its arithmetic?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename T> struct V;   // vector width
// pattern of k_mf_dual<0,0>: per image 15 unique loads + 8 neighbour loads, 6 stores
template <int W>   // W doubles per lane (1 or 2)
__global__ __launch_bounds__(256) void k_pattern(int m, int npix, int fs, int K, const double* __restrict__ xc, const double* __restrict__ xo,
                          const double* __restrict__ pin, double* __restrict__ pout, const double* __restrict__ g, const double* __restrict__ b, int neigh) {
    typedef double vec __attribute__((ext_vector_type(W)));
    const int nchunk = npix / (256 * W);
    for (int c = blockIdx.x; c < nchunk; c += gridDim.x) {
        const int q = (c * 256 + threadIdx.x) * W;
        const int e = q + m;
        for (int k = 0; k < K; ++k) {
            vec acc = 0.0;
            #pragma unroll
            for (int f = 0; f < 3; ++f) {
                const int fld = (f < 2) ? 2 * k + f : 2 * K + k;
                acc += *(const vec*)(xc + (size_t)fld * fs + e);
                acc += *(const vec*)(xo + (size_t)fld * fs + e);
                if (neigh && f < 2) {
                    acc += *(const vec*)(xc + (size_t)fld * fs + e + m);
                    acc += *(const vec*)(xo + (size_t)fld * fs + e + m);
                    if (W == 1) { acc += *(const vec*)(xc + (size_t)fld * fs + e + 1); acc += *(const vec*)(xo + (size_t)fld * fs + e + 1); }
                }
            }
            acc += *(const vec*)(g + (size_t)(2 * k) * fs + e);
            acc += *(const vec*)(g + (size_t)(2 * k + 1) * fs + e);
            acc += *(const vec*)(b + (size_t)k * fs + e);
            vec v[6];
            v[0] = *(const vec*)(pin + (size_t)k * fs + e);
            #pragma unroll
            for (int f = 0; f < 4; ++f) v[1 + f] = *(const vec*)(pin + (size_t)(K + 4 * k + f) * fs + e);
            v[5] = *(const vec*)(pin + (size_t)(5 * K + k) * fs + e);
            *(vec*)(pout + (size_t)k * fs + e) = v[0] + acc;
            #pragma unroll
            for (int f = 0; f < 4; ++f) *(vec*)(pout + (size_t)(K + 4 * k + f) * fs + e) = v[1 + f] * acc;
            *(vec*)(pout + (size_t)(5 * K + k) * fs + e) = v[5] - acc;
        }
    }
}

template <int W>
__global__ __launch_bounds__(256) void k_copy(size_t n, const double* __restrict__ a, double* __restrict__ o) {
    typedef double vec __attribute__((ext_vector_type(W)));
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * W; i < n; i += (size_t)gridDim.x * 256 * W)
        *(vec*)(o + i) = *(const vec*)(a + i);
}

int main() {
    const int m = 4096, n = 4096, K = 16, npix = m * n, fs = m * (n + 2);
    double *xc, *xo, *pin, *pout, *g, *b;
    CHK(hipMalloc(&xc, sizeof(double) * (size_t)48 * fs)); CHK(hipMalloc(&xo, sizeof(double) * (size_t)48 * fs));
    CHK(hipMalloc(&pin, sizeof(double) * (size_t)96 * fs)); CHK(hipMalloc(&pout, sizeof(double) * (size_t)96 * fs));
    CHK(hipMalloc(&g, sizeof(double) * (size_t)32 * fs)); CHK(hipMalloc(&b, sizeof(double) * (size_t)16 * fs));
    CHK(hipMemset(xc, 0, sizeof(double) * (size_t)48 * fs)); CHK(hipMemset(xo, 0, sizeof(double) * (size_t)48 * fs));
    CHK(hipMemset(pin, 0, sizeof(double) * (size_t)96 * fs)); CHK(hipMemset(g, 0, sizeof(double) * (size_t)32 * fs));
    CHK(hipMemset(b, 0, sizeof(double) * (size_t)16 * fs));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const double bytes = 21.0 * 8 * K * npix;
    int grids[] = {768, 1024, 2048, 4096};
    for (int neigh = 0; neigh < 2; ++neigh)
    for (int gi = 0; gi < 4; ++gi) {
        for (int w = 1; w <= 2; ++w) {
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                CHK(hipEventRecord(e0));
                if (w == 1) hipLaunchKernelGGL(k_pattern<1>, dim3(grids[gi]), dim3(256), 0, 0, m, npix, fs, K, xc, xo, pin, pout, g, b, neigh);
                else hipLaunchKernelGGL(k_pattern<2>, dim3(grids[gi]), dim3(256), 0, 0, m, npix, fs, K, xc, xo, pin, pout, g, b, neigh);
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            printf("pattern neigh=%d grid=%d W=%d: %.3f ms  %.0f GB/s (algorithmic 21 elems)\n", neigh, grids[gi], w, best, bytes / best / 1e6);
        }
    }
    const size_t ncopy = (size_t)96 * fs;
    for (int w = 1; w <= 2; ++w) {
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            CHK(hipEventRecord(e0));
            if (w == 1) hipLaunchKernelGGL(k_copy<1>, dim3(2048), dim3(256), 0, 0, ncopy, pin, pout);
            else hipLaunchKernelGGL(k_copy<2>, dim3(2048), dim3(256), 0, 0, ncopy, pin, pout);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("copy W=%d: %.3f ms %.0f GB/s\n", w, best, 2.0 * 8 * ncopy / best / 1e6);
    }
    return 0;
}
