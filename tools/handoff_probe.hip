// PROBE ONLY: what a layer-pipelined persistent backward would hand between CU groups.  In that design CU groups own
// layers and keep their dW accumulators in registers; a group hands its dY tile to the group of the next layer through L2 / MALL instead of
// writing it to HBM for a later kernel.  The question is the AGGREGATE hand-off bandwidth: 128 producer workgroups each stream 64 KiB tiles to
// one partner workgroup (a ring of 4 slots per pair in global memory; release store of a flag at agent scope after the tile, acquire poll on the
// other side, an acknowledgement counter back), all 256 workgroups co-resident (one per CU).
//   mode 0: partner on the SAME XCD (workgroup b -> b + 8; consecutive workgroup ids go round the 8 XCDs)
//   mode 1: partner on the NEXT XCD (b -> b + 1)
//   modes 3 / 4: modes 0 / 1 with the tile's stores and loads marked sc0 sc1 (write-through / read-around the XCD's non-coherent L2 lines), so that
//   the release has no dirty L2 to write back and the acquire nothing to invalidate -- the cheapest coherent path the ISA offers
//   mode 2: no partner -- every workgroup writes its tiles and reads them back itself (what the memory system gives without any hand-off protocol)
//   hipcc --offload-arch=gfx950 -O3 tools/handoff_probe.hip -o tools/bin/handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int TILE_F4 = 4096;          // float4 per tile: 64 KiB
constexpr int RING = 4;
typedef float f4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k_handoff(float4 *buf, unsigned *flags, unsigned *acks, unsigned *xcc, float *sink, int iters, int mode) {
    extern __shared__ float lds[];     // (sized to keep one workgroup per CU)
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[b] = id & 0xf;
    }
    const bool wt = mode >= 3;
    if (wt) mode -= 3;
    int pair, producer;
    if (mode == 0) { const int x = b & 7, s = b >> 3; pair = (s >> 1) * 8 + x; producer = !(s & 1); }      // (s, s + 1) on XCD x
    else if (mode == 1) { pair = b >> 1; producer = !(b & 1); }
    else { pair = b; producer = 1; }
    float4 *ring = buf + (size_t)pair * RING * TILE_F4;
    unsigned *flag = flags + pair * 32, *ack = acks + pair * 32;
    float acc = 0.f;
    for (int t = 0; t < iters; ++t) {
        float4 *tile = ring + (size_t)(t % RING) * TILE_F4;
        if (producer) {
            if (mode != 2 && t >= RING) {
                if (tid == 0) while (__hip_atomic_load(ack, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(t - RING + 1)) __builtin_amdgcn_s_sleep(1);
                __syncthreads();
            }
            const float v = 1e-3f * (float)(t + tid);
#pragma unroll
            for (int i = 0; i < TILE_F4 / 512; ++i) {
                const float4 x = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
                const f4v xv = {x.x, x.y, x.z, x.w};
                if (wt) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(tile + tid + 512 * i), "v"(xv) : "memory");
                else tile[tid + 512 * i] = x;
            }
            if (mode != 2) {
                if (wt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(flag, (unsigned)(t + 1), wt ? __ATOMIC_RELAXED : __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (!producer || mode == 2) {
            if (mode != 2) {
                if (tid == 0) while (__hip_atomic_load(flag, wt ? __ATOMIC_RELAXED : __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(t + 1)) __builtin_amdgcn_s_sleep(1);
                __syncthreads();
                if (!wt) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            } else __syncthreads();
            if (wt) {
                f4v x[TILE_F4 / 512];
#pragma unroll
                for (int i = 0; i < TILE_F4 / 512; ++i) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(x[i]) : "v"(tile + tid + 512 * i) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < TILE_F4 / 512; ++i) acc += x[i].x + x[i].w;
            } else {
#pragma unroll
            for (int i = 0; i < TILE_F4 / 512; ++i) { const float4 x = tile[tid + 512 * i]; acc += x.x + x.w; }
            }
            if (mode != 2) {
                __syncthreads();
                if (tid == 0) __hip_atomic_store(ack, (unsigned)(t + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (acc == 123.456f) sink[b] = acc;
    lds[tid] = acc;
}

int main() {
    const int wgs = 256, iters = 1500;
    float4 *buf; unsigned *flags, *acks, *xcc; float *sink;
    hipMalloc(&buf, (size_t)wgs * RING * TILE_F4 * sizeof(float4));
    hipMalloc(&flags, wgs * 32 * 4); hipMalloc(&acks, wgs * 32 * 4); hipMalloc(&xcc, wgs * 4); hipMalloc(&sink, wgs * 4);
    (void)hipFuncSetAttribute((const void *)k_handoff, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[5] = {"hand-off to a workgroup on the SAME XCD", "hand-off to a workgroup on the NEXT XCD", "no hand-off: write, then read back in place",
                            "SAME XCD, sc0 sc1 stores / loads, relaxed flags", "NEXT XCD, sc0 sc1 stores / loads, relaxed flags"};
    for (int mode = 0; mode < 5; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(flags, 0, wgs * 32 * 4); hipMemset(acks, 0, wgs * 32 * 4);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_handoff, dim3(wgs), dim3(512), 96 * 1024, 0, buf, flags, acks, xcc, sink, iters, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const int streams = mode == 2 ? wgs : wgs / 2;
            const double gb = (double)streams * iters * TILE_F4 * 16 / 1e9;
            if (rep == 1) printf("mode %d (%s): %d streams x %d tiles of 64 KiB = %.1f GB handed over in %.3f ms = %.2f TB/s (%.1f GB/s per stream)\n",
                                 mode, names[mode], streams, iters, gb, ms, gb / ms, gb / ms * 1e3 / streams);
        }
        if (mode == 0) {
            unsigned h[256]; hipMemcpy(h, xcc, sizeof(h), hipMemcpyDeviceToHost);
            int ok = 0; for (int b = 0; b + 8 < 256; ++b) ok += h[b] == h[b + 8];
            printf("  XCC_ID of workgroups 0..15: "); for (int b = 0; b < 16; ++b) printf("%u ", h[b]);
            printf(" | workgroups b and b + 8 on the same XCD: %d of 248\n", ok);
        }
    }
    return 0;
}
