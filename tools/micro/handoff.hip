// Micro-benchmark: one-way latency of a flag hand-off between two workgroups of one kernel on gfx950, for
// different store / load flavours and placements (same XCD or not).  Build: hipcc --offload-arch=gfx950 -O3 handoff.hip -o handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define N_IT 4000
__device__ __forceinline__ unsigned long long ld(const unsigned long long* p, int mode) {
    switch (mode) {
        case 0: return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        case 1: case 2: return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        case 3: return __hip_atomic_fetch_add((unsigned long long*)p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        case 4: return __hip_atomic_fetch_add((unsigned long long*)p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        case 5: { unsigned long long v; asm volatile("global_load_dwordx2 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
        case 6: { unsigned long long v; asm volatile("buffer_inv sc0\n global_load_dwordx2 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
        case 7: { unsigned long long v; asm volatile("global_load_dwordx2 %0, %1, off sc0 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
        default: return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__device__ __forceinline__ void st(unsigned long long* p, unsigned long long v, int mode) {
    switch (mode) {
        case 0: case 1: case 4: __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
        case 2: case 3: case 5: case 6: case 7: __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break;
        default: __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void pingpong(unsigned long long* flags, int ida, int idb, int mode, int* xcc, int* fail) {
    int my; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(my));
    if (threadIdx.x == 0) xcc[blockIdx.x] = my;
    if (threadIdx.x != 0) return;
    const int b = blockIdx.x;
    if (b != ida && b != idb) return;
    unsigned long long* f0 = flags;        // a -> b
    unsigned long long* f1 = flags + 64;   // b -> a
    for (unsigned long long i = 1; i <= N_IT; ++i) {
        if (b == ida) {
            st(f0, i, mode);
            int spin = 0;
            while (ld(f1, mode) != i) if (++spin > (1 << 20)) { *fail = 1; return; }
        } else {
            int spin = 0;
            while (ld(f0, mode) != i) if (++spin > (1 << 20)) { *fail = 1; return; }
            st(f1, i, mode);
        }
    }
}
int main() {
    unsigned long long* flags; int *xcc, *fail;
    hipMalloc(&flags, 1024); hipMalloc(&xcc, 64 * 4); hipMalloc(&fail, 4);
    int hx[64];
    const int pairs[3][2] = {{0, 8}, {0, 1}, {0, 16}};
    const char* names[] = {"st agent / ld agent", "st agent / ld wg", "st wg / ld wg", "st wg / rmw wg", "st agent / rmw agent", "st wg / ld nt", "st wg / inv sc0 + ld", "st wg / ld sc0 nt", "system / system"};
    for (int pr = 0; pr < 3; ++pr)
        for (int mode = 0; mode < 9; ++mode) {
            hipMemset(flags, 0, 1024); hipMemset(fail, 0, 4);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(pingpong, dim3(32), dim3(64), 0, 0, flags, pairs[pr][0], pairs[pr][1], mode, xcc, fail);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            int hf; hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost); hipMemcpy(hx, xcc, 32 * 4, hipMemcpyDeviceToHost);
            printf("wg %2d (xcc %d) <-> wg %2d (xcc %d)  %-24s : %s one-way %.0f ns\n", pairs[pr][0], hx[pairs[pr][0]], pairs[pr][1], hx[pairs[pr][1]],
                   names[mode], hf ? "TIMED OUT" : "ok", hf ? 0.0 : ms * 1e6 / (2.0 * N_IT));
        }
    printf("xcc of wg 0..15:"); for (int i = 0; i < 16; ++i) printf(" %d", hx[i]); printf("\n");
    return 0;
}
