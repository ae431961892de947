#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(size_t n2, const d2* __restrict__ a, d2* __restrict__ o) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n2; i += U * stride) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], o + i + u * stride); else o[i + u * stride] = v[u]; }
    }
    for (; i < n2; i += stride) o[i] = a[i];
}
template <int U>
__global__ __launch_bounds__(256) void k_read(size_t n2, const d2* __restrict__ a, double* __restrict__ o) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    d2 acc = 0.0;
    for (; i + (U - 1) * stride < n2; i += U * stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc += a[i + u * stride];
    }
    if (acc.x + acc.y == 12345.678) o[0] = acc.x;
}
__global__ __launch_bounds__(256) void k_write(size_t n2, d2* __restrict__ o) {
    const size_t stride = (size_t)gridDim.x * 256;
    d2 v = {1.0, 2.0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) o[i] = v;
}
// contiguous-chunk-per-block copy (each block streams its own contiguous range)
__global__ __launch_bounds__(256) void k_copy_chunk(size_t n2, const d2* __restrict__ a, d2* __restrict__ o) {
    const size_t per = n2 / gridDim.x;
    const size_t s = per * blockIdx.x, e = s + per;
    for (size_t i = s + threadIdx.x; i + 768 < e; i += 1024) {
        d2 v0 = a[i], v1 = a[i + 256], v2 = a[i + 512], v3 = a[i + 768];
        o[i] = v0; o[i + 256] = v1; o[i + 512] = v2; o[i + 768] = v3;
    }
}
int main() {
    const size_t n = (size_t)1 << 30;   // doubles: 8 GiB per buffer
    double *a, *o;
    CHK(hipMalloc(&a, n * 8)); CHK(hipMalloc(&o, n * 8));
    CHK(hipMemset(a, 0, n * 8)); CHK(hipMemset(o, 0, n * 8));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto run = [&](const char* name, double bytes, auto launch) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-40s %.3f ms  %.0f GB/s\n", name, best, bytes / best / 1e6);
    };
    const size_t n2 = n / 2;
    for (int grid : {1024, 2048, 4096, 8192, 65536}) {
        char nm[64];
        snprintf(nm, 64, "copy U1 grid %d", grid); run(nm, 16.0 * n, [&] { hipLaunchKernelGGL((k_copy<1, false>), dim3(grid), dim3(256), 0, 0, n2, (const d2*)a, (d2*)o); });
        snprintf(nm, 64, "copy U4 grid %d", grid); run(nm, 16.0 * n, [&] { hipLaunchKernelGGL((k_copy<4, false>), dim3(grid), dim3(256), 0, 0, n2, (const d2*)a, (d2*)o); });
        snprintf(nm, 64, "copy U4 NT grid %d", grid); run(nm, 16.0 * n, [&] { hipLaunchKernelGGL((k_copy<4, true>), dim3(grid), dim3(256), 0, 0, n2, (const d2*)a, (d2*)o); });
        snprintf(nm, 64, "copy chunk grid %d", grid); run(nm, 16.0 * n, [&] { hipLaunchKernelGGL(k_copy_chunk, dim3(grid), dim3(256), 0, 0, n2, (const d2*)a, (d2*)o); });
        snprintf(nm, 64, "read U4 grid %d", grid); run(nm, 8.0 * n, [&] { hipLaunchKernelGGL((k_read<4>), dim3(grid), dim3(256), 0, 0, n2, (const d2*)a, o); });
        snprintf(nm, 64, "write grid %d", grid); run(nm, 8.0 * n, [&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, n2, (d2*)o); });
    }
    run("hipMemcpyDtoD", 16.0 * n, [&] { hipMemcpyAsync(o, a, n * 8, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
