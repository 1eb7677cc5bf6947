// vmm_bench -- does the way a large array is ALLOCATED decide how fast it streams?  (the 512^3 product
// runs at 1 650 / 1 715 / 1 787 / 1 843 us by process, a property of the value stream's allocation.)
// A 7.5 GB read stream (16 B per lane, non-temporal, grid 1792 as the format-5 product) + a 1 GB written vector, with the
// big array obtained by  (a) hipMalloc,  (b) hipMemCreate / hipMemMap in chunks of 2 MiB, 64 MiB, 1 GiB or one handle.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void stream(const d2* __restrict__ a, long n2, double* __restrict__ y, long ny) {
    const long S = (long)gridDim.x * 256;
    double s = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += S) {
        const d2 v = __builtin_nontemporal_load(a + i);
        s += v.x + v.y;
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ny; i += S) __builtin_nontemporal_store(s, y + i);
}
struct Vmm { void* va = nullptr; size_t size = 0; std::vector<hipMemGenericAllocationHandle_t> h; };
static bool vmm_alloc(Vmm& v, size_t bytes, size_t chunk) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) return false;
    if (chunk == 0) chunk = bytes;
    chunk = (chunk + gran - 1) / gran * gran;
    v.size = (bytes + chunk - 1) / chunk * chunk;
    if (hipMemAddressReserve(&v.va, v.size, (size_t)1 << 30, nullptr, 0) != hipSuccess) return false;
    for (size_t off = 0; off < v.size; off += chunk) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) return false;
        if (hipMemMap((char*)v.va + off, chunk, 0, h, 0) != hipSuccess) return false;
        v.h.push_back(h);
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    return hipMemSetAccess(v.va, v.size, &acc, 1) == hipSuccess;
}
static void vmm_free(Vmm& v) {
    if (!v.va) return;
    hipMemUnmap(v.va, v.size);
    for (auto h : v.h) hipMemRelease(h);
    hipMemAddressFree(v.va, v.size);
    v = Vmm();
}
static float timeit(const d2* a, long n2, double* y, long ny, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(stream, dim3(1792), dim3(256), 0, 0, a, n2, y, ny); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(stream, dim3(1792), dim3(256), 0, 0, a, n2, y, ny);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps * 1e3f;
}
int main(int argc, char** argv) {
    const size_t bytes = (size_t)7500 << 20;
    const long ny = 134217728;
    const int rounds = argc > 1 ? atoi(argv[1]) : 3;
    // something big first, like the CSR arrays of the real run (11.8 GB)
    void* ballast; CK(hipMalloc(&ballast, (size_t)11800 << 20)); CK(hipMemset(ballast, 1, (size_t)11800 << 20));
    double* y; CK(hipMalloc(&y, 8 * ny));
    size_t gran = 0; { hipMemAllocationProp p = {}; p.type = hipMemAllocationTypePinned; p.location.type = hipMemLocationTypeDevice; hipMemGetAllocationGranularity(&gran, &p, hipMemAllocationGranularityRecommended); size_t gmin = 0; hipMemGetAllocationGranularity(&gmin, &p, hipMemAllocationGranularityMinimum); printf("VMM granularity: recommended %zu, minimum %zu\n", gran, gmin); }
    for (int r = 0; r < rounds; ++r) {
        { void* a; CK(hipMalloc(&a, bytes)); CK(hipMemset(a, 0, bytes)); printf("round %d hipMalloc          : %7.1f us\n", r, timeit((d2*)a, bytes / 16, y, ny, 10)); CK(hipFree(a)); }
        for (size_t chunk : {(size_t)2 << 20, (size_t)64 << 20, (size_t)1 << 30, (size_t)0}) {
            Vmm v;
            if (!vmm_alloc(v, bytes, chunk)) { printf("round %d VMM chunk %zu MiB: allocation failed (%s)\n", r, chunk >> 20, hipGetErrorString(hipGetLastError())); vmm_free(v); continue; }
            CK(hipMemset(v.va, 0, bytes));
            printf("round %d VMM chunk %5zu MiB : %7.1f us\n", r, chunk >> 20, timeit((d2*)v.va, bytes / 16, y, ny, 10));
            vmm_free(v);
        }
        // a different y each round as well
        CK(hipFree(y)); void* pad; CK(hipMalloc(&pad, (size_t)(100 + 300 * r) << 20)); CK(hipMalloc(&y, 8 * ny)); CK(hipFree(pad));
    }
    return 0;
}
