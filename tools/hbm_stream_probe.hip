// HBM streaming plateaus of one MI355X for the access shapes the 16-bit training kernels use (docs/HISTORY.md 4.1c):
//   read   : 16 B per lane, wave-contiguous 1 KiB pieces (what k_wg16's DMA and every fragment load does)
//   write  : 16 B per lane nontemporal stores, 1 KiB per wave instruction (store_t16)
//   copy   : both at once (a kernel that reads as much as it writes)
//   dma    : HBM -> LDS by global_load_lds_dwordx4 (k_wg16's operand stream, the MLP kernels' weight stream): every wave keeps RING x 4 KiB
//            in flight in its own LDS ring, nothing consumes the data; with 128 KiB of LDS per workgroup (one per CU, like k_wg16) and 32 KiB (four per CU)
// Buffers are 4 GiB (far beyond the 256 MiB Infinity Cache); every workgroup streams a contiguous slice.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_stream_probe.hip -o /tmp/hbm_probe && /tmp/hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>   // 0 read, 1 write, 2 copy
__global__ __launch_bounds__(512) void k_stream(const u4 *src, u4 *dst, size_t n_per_wg, unsigned *sink) {
    const size_t base = (size_t)blockIdx.x * n_per_wg;
    u4 acc = {0u, 0u, 0u, 0u};
    for (size_t i = threadIdx.x; i < n_per_wg; i += 4 * 512) {
        u4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t k = base + i + (size_t)u * 512;
            if (MODE != 1) v[u] = __builtin_nontemporal_load(src + k);
            else v[u] = (u4){(unsigned)k, 1u, 2u, 3u};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t k = base + i + (size_t)u * 512;
            if (MODE != 0) __builtin_nontemporal_store(v[u], dst + k);
            else acc ^= v[u];
        }
    }
    if (MODE == 0 && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1u;
}

template <int RING>   // 4 KiB bursts in flight per wave
__global__ __launch_bounds__(512) void k_dma(const char *src, size_t bytes_per_wg) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t per_wave = bytes_per_wg / 8;
    const char *g = src + (size_t)blockIdx.x * bytes_per_wg + (size_t)wave * per_wave + lane * 16;
    const unsigned l0 = (unsigned)(size_t)(lds + wave * RING * 4096);
    const int n = (int)(per_wave / 4096);
    for (int i = 0; i < n; ++i) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(l0 + (unsigned)(i % RING) * 4096u);
        unsigned keep;
        if (i >= RING) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (RING - 1)) : "memory");       // the burst that used this slot has landed
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\t"
                     "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(g + (size_t)i * 4096), "s"(dst) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// registers -> LDS: what a register-staged operand stream does (global_load_dwordx4 + ds_write_b128 of the same lane-linear image); MIX: every
// second 4 KiB burst goes by DMA instead
template <int DEPTH, bool MIX>   // 4 KiB bursts in flight per wave
__global__ __launch_bounds__(512) void k_regs_lds(const char *src, size_t bytes_per_wg) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t per_wave = bytes_per_wg / 8;
    const char *g = src + (size_t)blockIdx.x * bytes_per_wg + (size_t)wave * per_wave + lane * 16;
    char *l = lds + wave * DEPTH * 4096;
    const int n = (int)(per_wave / 4096);
    u4 v[DEPTH][4];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int u = 0; u < 4; ++u) v[d][u] = (MIX && (d & 1)) ? (u4){0u, 0u, 0u, 0u} : __builtin_nontemporal_load((const u4 *)(g + (size_t)d * 4096 + u * 1024));
    for (int i = 0; i + DEPTH <= n; i += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (MIX && (d & 1)) {            // DMA burst for this slot (of the NEXT round), its predecessor drained by the vmcnt of the register loads behind it
                const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(l + d * 4096));
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                             "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\t"
                             "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(g + (size_t)(i + d) * 4096), "s"(dst) : "memory");
                continue;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) *(u4 *)(l + d * 4096 + u * 1024 + lane * 16) = v[d][u];        // waits for this burst's loads (compiler's vmcnt)
            if (i + DEPTH + d < n)
#pragma unroll
                for (int u = 0; u < 4; ++u) v[d][u] = __builtin_nontemporal_load((const u4 *)(g + (size_t)(i + DEPTH + d) * 4096 + u * 1024));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int main() {
    const size_t bytes = (size_t)4 << 30, n = bytes / 16;
    u4 *a, *b; unsigned *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grids[] = {256, 512, 1024, 4096, 16384};
    for (int mode = 0; mode < 3; ++mode)
        for (int g : grids) {
            const size_t per = n / g;       // multiples of 2048 for these grids
            float best = 1e30f;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k_stream<0>, dim3(g), dim3(512), 0, 0, a, b, per, sink);
                if (mode == 1) hipLaunchKernelGGL(k_stream<1>, dim3(g), dim3(512), 0, 0, a, b, per, sink);
                if (mode == 2) hipLaunchKernelGGL(k_stream<2>, dim3(g), dim3(512), 0, 0, a, b, per, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            const double moved = (mode == 2 ? 2.0 : 1.0) * (double)bytes;
            printf("%-5s grid %5d: %.3f ms  %.2f TB/s%s\n", mode == 0 ? "read" : mode == 1 ? "write" : "copy", g, best, moved / best * 1e-9,
                   mode == 2 ? " (read + written)" : "");
        }
    for (int ring = 1; ring <= 4; ring *= 2)
        for (int g : {256, 1024, 4096}) {
            const size_t per = bytes / g;
            float best = 1e30f;
            const size_t lds = (size_t)8 * ring * 4096;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0));
                if (ring == 1) hipLaunchKernelGGL(k_dma<1>, dim3(g), dim3(512), lds, 0, (const char *)a, per);
                if (ring == 2) hipLaunchKernelGGL(k_dma<2>, dim3(g), dim3(512), lds, 0, (const char *)a, per);
                if (ring == 4) hipLaunchKernelGGL(k_dma<4>, dim3(g), dim3(512), lds, 0, (const char *)a, per);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            printf("dma   ring %d x 4 KiB per wave (%3zu KiB LDS per workgroup), grid %5d: %.3f ms  %.2f TB/s\n", ring, lds >> 10, g, best, (double)bytes / best * 1e-9);
        }
    for (int mix = 0; mix < 2; ++mix)
        for (int depth = 2; depth <= 4; depth *= 2)
            for (int g : {256, 1024}) {
                const size_t per = bytes / g;
                float best = 1e30f;
                const size_t lds = (size_t)8 * depth * 4096;
                for (int rep = 0; rep < 6; ++rep) {
                    CK(hipEventRecord(e0));
                    if (!mix && depth == 2) hipLaunchKernelGGL((k_regs_lds<2, false>), dim3(g), dim3(512), lds, 0, (const char *)a, per);
                    if (!mix && depth == 4) hipLaunchKernelGGL((k_regs_lds<4, false>), dim3(g), dim3(512), lds, 0, (const char *)a, per);
                    if (mix && depth == 2) hipLaunchKernelGGL((k_regs_lds<2, true>), dim3(g), dim3(512), lds, 0, (const char *)a, per);
                    if (mix && depth == 4) hipLaunchKernelGGL((k_regs_lds<4, true>), dim3(g), dim3(512), lds, 0, (const char *)a, per);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep > 0 && ms < best) best = ms;
                }
                printf("%s depth %d x 4 KiB per wave (%3zu KiB LDS per workgroup), grid %5d: %.3f ms  %.2f TB/s\n", mix ? "regs+dma -> LDS" : "regs -> LDS    ", depth,
                       lds >> 10, g, best, (double)bytes / best * 1e-9);
            }
    return 0;
}
