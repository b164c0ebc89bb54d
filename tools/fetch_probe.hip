// Calibration of rocprofv3's FETCH_SIZE on this box, per access type and pattern (round-3 verdict item 5): every kernel below reads a KNOWN
// number of bytes of a 1-GiB buffer exactly once; the ratio counter / bytes is the correction factor for that kind of read.
//   reg_full / dma_full   fully coalesced streaming reads, 16 B per lane: global loads to registers / buffer_load ... lds (LDS-DMA)
//   reg_half / dma_half   the weight-gradient kernels' pattern: 128 B (one 64-channel half) of every 256-B pixel row, 8 lanes per row
//   reg_half_twice        the same half rows read by TWO workgroups on DIFFERENT XCDs at about the same time
//   pair_half<reg | dma>  ... by two co-resident workgroups of the SAME XCD in step (the two output-channel tiles of a weight-gradient split
//                         share the X operand this way), the second one started 0 / 2 / 4 us late
// build + run on the GPU box:   hipcc -O3 --offload-arch=gfx950 tools/fetch_probe.hip -o /tmp/fetch_probe
//                               rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fp -o fp -- /tmp/fetch_probe
// then tools/fetch_probe_parse.py /tmp/fp  (prints bytes read, counter x 1024, ratio per kernel)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define LDSP __attribute__((address_space(3)))
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr size_t kBytes = 1ull << 30;
constexpr int kGrid = 256 * 8, kBlock = 256;

__global__ void reg_full(const u32x4* __restrict__ src, unsigned* sink, size_t nvec) {
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) acc ^= src[i];
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

__global__ void dma_full(const void* src, unsigned nbytes) {
    __shared__ __attribute__((aligned(16))) char lds[4 * 1024];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(src), 0, (int)nbytes, 0x00020000);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    LDSP char* dst = (LDSP char*)(lds + wave * 1024);
    const unsigned waves = gridDim.x * (blockDim.x >> 6);
    for (unsigned w = blockIdx.x * (blockDim.x >> 6) + wave; w < nbytes / 1024; w += waves)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDSP void*)dst, 16, w * 1024u + lane * 16u, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// half rows: instruction k of the grid covers rows 8k .. 8k+7 (256 B apart), lane = (row = lane >> 3, 16-B chunk = lane & 7), half h
__global__ void reg_half(const char* __restrict__ src, unsigned* sink, unsigned nrows, int h, int twice) {
    u32x4 acc = {0, 0, 0, 0};
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned waves = (gridDim.x >> twice) * (blockDim.x >> 6);
    for (unsigned k = (blockIdx.x >> twice) * (blockDim.x >> 6) + wave; k < nrows / 8; k += waves)
        acc ^= *reinterpret_cast<const u32x4*>(src + (size_t)(8 * k + (lane >> 3)) * 256 + h * 128 + (lane & 7) * 16);
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

__global__ void dma_half(const void* src, unsigned nbytes, int h) {
    __shared__ __attribute__((aligned(16))) char lds[4 * 1024];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(src), 0, (int)nbytes, 0x00020000);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    LDSP char* dst = (LDSP char*)(lds + wave * 1024);
    const unsigned waves = gridDim.x * (blockDim.x >> 6);
    for (unsigned k = blockIdx.x * (blockDim.x >> 6) + wave; k < nbytes / 256 / 8; k += waves)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDSP void*)dst, 16, (8 * k + (lane >> 3)) * 256u + h * 128u + (lane & 7) * 16u, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Same-XCD sharing: 512 co-resident workgroups (2 per CU); blocks b and b + 256 (same XCD: ids 256 apart) walk the SAME half rows in step, like
// the two output-channel tiles of a weight-gradient split that share the X operand.  delay: the second block of a pair starts ~`delay` us late.
// kAux: cache-policy bits of the LDS-DMA load (gfx950: 1 = sc0, 2 = nt, 16 = sc1) - does any of them make the L2 serve the second reader?
template <bool kDma, int kAux = 0>
__global__ void pair_half(const void* src, unsigned* sink, unsigned nbytes, int delay) {
    __shared__ __attribute__((aligned(16))) char lds[4 * 1024];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(src), 0, (int)nbytes, 0x00020000);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned b = blockIdx.x & 255;
    if (blockIdx.x >= 256) for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(32);      // ~ 1 us per iteration
    LDSP char* dst = (LDSP char*)(lds + wave * 1024);
    u32x4 acc = {0, 0, 0, 0};
    const unsigned waves = 256 * (blockDim.x >> 6);
    for (unsigned k = b * (blockDim.x >> 6) + wave; k < nbytes / 256 / 8; k += waves) {
        const unsigned off = (8 * k + (lane >> 3)) * 256u + (lane & 7) * 16u;
        if (kDma) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDSP void*)dst, 16, off, 0, 0, kAux);
        else acc ^= __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

__global__ void store_full(u32x4* dst, size_t nvec) {
    const u32x4 v = {1, 2, 3, 4};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}

int main() {
    char *a, *b; unsigned* sink;
    if (hipMalloc(&a, kBytes) != hipSuccess || hipMalloc(&b, kBytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(a, 1, kBytes); (void)hipMemset(b, 2, kBytes);
    const unsigned lim = (unsigned)(kBytes - 65536);        // buffer descriptors: 32-bit sizes
    // every measured kernel is preceded by a 1-GiB read of the OTHER buffer, so nothing of its own buffer is left in the 256-MiB Infinity Cache
    auto flush = [&]() { reg_full<<<kGrid, kBlock>>>((const u32x4*)b, sink, kBytes / 16); };
    for (int rep = 0; rep < 2; ++rep) {
        flush(); reg_full<<<kGrid, kBlock>>>((const u32x4*)a, sink, kBytes / 16);
        flush(); dma_full<<<kGrid, kBlock>>>(a, lim);
        flush(); reg_half<<<kGrid, kBlock>>>(a, sink, (unsigned)(kBytes / 256), 0, 0);
        flush(); dma_half<<<kGrid, kBlock>>>(a, lim, 0);
        flush(); reg_half<<<kGrid, kBlock>>>(a, sink, (unsigned)(kBytes / 256), 1, 1);
        for (int delay = 0; delay <= 4; delay += 2) {
            flush(); pair_half<false><<<512, kBlock>>>(a, sink, lim, delay);
            flush(); pair_half<true><<<512, kBlock>>>(a, sink, lim, delay);
        }
        flush(); pair_half<true, 1><<<512, kBlock>>>(a, sink, lim, 0);
        flush(); pair_half<true, 2><<<512, kBlock>>>(a, sink, lim, 0);
        flush(); pair_half<true, 16><<<512, kBlock>>>(a, sink, lim, 0);
        flush(); pair_half<true, 17><<<512, kBlock>>>(a, sink, lim, 0);
        flush(); pair_half<true, 3><<<512, kBlock>>>(a, sink, lim, 0);
        flush(); store_full<<<kGrid, kBlock>>>((u32x4*)a, kBytes / 16);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    printf("bytes: reg_full %zu dma_full %u reg_half %zu dma_half %u reg_half(twice: each half row by two workgroups) %zu store_full %zu\n",
           kBytes, lim / 1024 * 1024, kBytes / 2, lim / 256 / 8 * 8 * 128, kBytes / 2, kBytes);
    return 0;
}
