// Streaming-copy variants: which shape reproduces the ~6.3 TB/s float4 copy of /opt/skills/guides/MI355X_MICROARCH.md:36
// (cp_stream_copy_gbps measured 5.5 TB/s with one element in flight per thread, grid-stride).
// hipcc --offload-arch=gfx950 -O3 -o scratch/bw_copy scratch/bw_copy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename V, bool NTL, bool NTS>
__device__ __forceinline__ void cp1(const V *s, V *d, size_t q) {
    V v = NTL ? __builtin_nontemporal_load(s + q) : s[q];
    if (NTS) __builtin_nontemporal_store(v, d + q); else d[q] = v;
}
// A: grid-stride, one element in flight per thread
template <typename V, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_a(size_t nv, const V *__restrict__ s, V *__restrict__ d) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < nv; q += stride) cp1<V, NTL, NTS>(s, d, q);
}
// B: one element per thread, no loop
template <typename V, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_b(size_t nv, const V *__restrict__ s, V *__restrict__ d) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nv) cp1<V, NTL, NTS>(s, d, q);
}
// C: U loads in flight per thread, a workgroup owns a contiguous chunk of U * blockDim elements, grid-stride over chunks
template <typename V, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_c(size_t nv, const V *__restrict__ s, V *__restrict__ d) {
    const size_t chunk = (size_t)U * blockDim.x;
    for (size_t base = (size_t)blockIdx.x * chunk; base < nv; base += (size_t)gridDim.x * chunk) {
        V v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t q = base + (size_t)u * blockDim.x + threadIdx.x;
            v[u] = NTL ? __builtin_nontemporal_load(s + q) : s[q];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t q = base + (size_t)u * blockDim.x + threadIdx.x;
            if (NTS) __builtin_nontemporal_store(v[u], d + q); else d[q] = v[u];
        }
    }
}
typedef double d2 __attribute__((ext_vector_type(2)));
int main() {
    const size_t bytes = (size_t)2 << 30;
    char *a, *b;
    CHK(hipMalloc(&a, bytes)); CHK(hipMalloc(&b, bytes));
    CHK(hipMemset(a, 1, bytes)); CHK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-52s %.3f ms  %.0f GB/s\n", name, best, 2.0 * bytes / best / 1e6); fflush(stdout);
    };
#define RA(V, L, S, G) run("A grid-stride " #V " ntl" #L " nts" #S " grid " #G, [&] { hipLaunchKernelGGL((k_a<V, L, S>), dim3(G), dim3(256), 0, 0, bytes / sizeof(V), (const V *)a, (V *)b); });
#define RB(V, L, S) run("B one-per-thread " #V " ntl" #L " nts" #S, [&] { hipLaunchKernelGGL((k_b<V, L, S>), dim3((unsigned)(bytes / sizeof(V) / 256)), dim3(256), 0, 0, bytes / sizeof(V), (const V *)a, (V *)b); });
#define RC(V, U, L, S, G) run("C chunk U" #U " " #V " ntl" #L " nts" #S " grid " #G, [&] { hipLaunchKernelGGL((k_c<V, U, L, S>), dim3(G), dim3(256), 0, 0, bytes / sizeof(V), (const V *)a, (V *)b); });
    RA(double, true, true, 4096) RA(d2, true, true, 4096) RA(d2, false, false, 4096) RA(d2, false, true, 4096)
    RB(double, true, true) RB(d2, true, true) RB(d2, false, false) RB(d2, false, true) RB(double, false, false)
    RC(d2, 4, true, true, 2048) RC(d2, 4, false, false, 2048) RC(d2, 4, false, true, 2048) RC(d2, 8, false, false, 2048)
    RC(d2, 4, false, false, 8192) RC(d2, 4, false, false, 65536) RC(d2, 2, false, false, 65536)
    RC(double, 4, false, false, 2048) RC(double, 8, false, false, 2048) RC(double, 8, true, true, 2048) RC(double, 8, false, false, 65536)
    RC(d2, 4, true, false, 2048) RC(d2, 4, true, false, 65536)
    CHK(hipDeviceSynchronize());
    return 0;
}
