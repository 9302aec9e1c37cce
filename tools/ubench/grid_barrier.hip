// Micro-benchmark: cost of a device-wide barrier between co-resident workgroups (one per CU) on MI355X, as a
// persistent decode kernel would use between the phases of a layer.  Release/acquire at agent scope (the compiler emits the
// L2 write-back / invalidate that cross-XCD visibility needs).  Also checks that data written before the barrier by other
// workgroups is visible after it.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gb tools/ubench/grid_barrier.hip && timeout 60 /tmp/gb
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned& target, unsigned nwg) {
    __syncthreads();
    if (threadIdx.x == 0) {
        target += nwg;
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k(unsigned* counter, float* buf, int iters, unsigned long long* cyc, int* errors) {
    unsigned target = 0;
    const unsigned nwg = gridDim.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int bad = 0;
    for (int it = 0; it < iters; it++) {
        // every workgroup writes its slot, then after the barrier reads the slot of another workgroup (other XCD)
        if (threadIdx.x == 0) buf[(it & 1) * nwg + blockIdx.x] = (float)(it * 1000 + blockIdx.x);
        grid_barrier(counter, target, nwg);
        const unsigned other = (blockIdx.x * 37 + 11) % nwg;
        if (threadIdx.x == 0 && buf[(it & 1) * nwg + other] != (float)(it * 1000 + other)) bad++;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; if (bad) atomicAdd(errors, bad); }
}

int main() {
    unsigned* counter; float* buf; unsigned long long* cyc; int* err;
    hipMalloc(&counter, 4); hipMalloc(&buf, 2 * 256 * 4); hipMalloc(&cyc, 256 * 8); hipMalloc(&err, 4);
    for (int nwg : {64, 128, 256}) {
        hipMemset(counter, 0, 4); hipMemset(err, 0, 4);
        const int iters = 2000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<<<nwg, 256>>>(counter, buf, iters, cyc, err);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        int herr; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
        printf("%3d workgroups: %.2f us per barrier (%d iterations), stale reads: %d\n", nwg, ms * 1e3 / iters, iters, herr);
    }
    return 0;
}
