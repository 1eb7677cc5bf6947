// xcdbar -- does an XCD-local soft barrier work?  2048 workgroups (8 per CU); workgroup b bumps counter[b & 7] and spins
// (bounded) until it sees gridDim/8.  Variants: scope of the atomic (workgroup: stays in the XCD's L2; agent), scope of the load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1);} } while (0)
template <int ASCOPE, int LSCOPE>
__global__ __launch_bounds__(256, 8) void k(unsigned* cnt, unsigned target, int* out, long long* wait, int byxcc) {
    if (threadIdx.x == 0) {
        unsigned xcc = blockIdx.x & 7;
        if (byxcc) { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); xcc = v & 0xf; }
        unsigned* c = cnt + xcc * 32;
        __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, ASCOPE);
        const long long t0 = wall_clock64();
        unsigned v = 0;
        while ((v = __hip_atomic_load(c, __ATOMIC_RELAXED, LSCOPE)) < target && wall_clock64() - t0 < 5000) __builtin_amdgcn_s_sleep(2);
        out[blockIdx.x] = (int)v;
        wait[blockIdx.x] = wall_clock64() - t0;
        out[gridDim.x + blockIdx.x] = (int)xcc;
    }
}
template <int A, int L> void run(const char* nm, int byxcc) {
    const int G = 2048;
    unsigned* c; int* o; long long* w;
    CK(hipMalloc(&c, 4096)); CK(hipMalloc(&o, 8 * G)); CK(hipMalloc(&w, 8 * G));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(c, 0, 4096));
        hipLaunchKernelGGL((k<A, L>), dim3(G), dim3(256), 0, 0, c, G / 8, o, w, byxcc);
        CK(hipDeviceSynchronize());
    }
    std::vector<int> ho(2 * G); std::vector<long long> hw(G); std::vector<unsigned> hc(1024);
    CK(hipMemcpy(ho.data(), o, 8 * G, hipMemcpyDeviceToHost)); CK(hipMemcpy(hw.data(), w, 8 * G, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hc.data(), c, 4096, hipMemcpyDeviceToHost));
    int ok = 0; long long mx = 0; double av = 0; int mism = 0; int per[16] = {0};
    for (int b = 0; b < G; ++b) { ok += ho[b] >= G / 8; mx = std::max(mx, hw[b]); av += hw[b]; mism += (ho[G + b] != (b & 7)); per[ho[G + b] & 15]++; }
    printf("%-28s byxcc=%d: %4d / %d saw the target; wait mean %.2f us max %.2f us; final counters", nm, byxcc, ok, G, av / G / 100.0, mx / 100.0);
    for (int e = 0; e < 8; ++e) printf(" %u", hc[e * 32]);
    printf("; xcc != b&7 for %d workgroups; per xcc:", mism);
    for (int e = 0; e < 8; ++e) printf(" %d", per[e]);
    printf("\n");
}
int main() {
    run<__HIP_MEMORY_SCOPE_WORKGROUP, __HIP_MEMORY_SCOPE_AGENT>("atomic wg / load agent", 0);
    run<__HIP_MEMORY_SCOPE_AGENT, __HIP_MEMORY_SCOPE_AGENT>("atomic agent / load agent", 0);
    run<__HIP_MEMORY_SCOPE_WORKGROUP, __HIP_MEMORY_SCOPE_WORKGROUP>("atomic wg / load wg", 0);
    run<__HIP_MEMORY_SCOPE_WORKGROUP, __HIP_MEMORY_SCOPE_AGENT>("atomic wg / load agent", 1);
    run<__HIP_MEMORY_SCOPE_AGENT, __HIP_MEMORY_SCOPE_AGENT>("atomic agent / load agent", 1);
    run<__HIP_MEMORY_SCOPE_SYSTEM, __HIP_MEMORY_SCOPE_SYSTEM>("atomic system / load system", 0);
    return 0;
}
