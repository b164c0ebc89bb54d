// The data gradient of the up-path ResBlocks' 1x1 skip convolution over the concatenated input (reference: autograd of
// `skip_connection = nn.Conv2d(2C, C, 1)`, gms/diffusion/simple_unet.py:176,186): out_a | out_b [pixel][128] = dout[pixel][128] . Wd[128][256],
// a [pixels x 128] x [128 x 256] GEMM that moves 3 N bytes for 128 k of work per output - bound by memory, not by the matrix cores.
// The general LDS-DMA convolution kernel runs it at 3.2 TB/s (two K-steps per 256-pixel tile: its pipeline never fills).  Here, as in the
// stem kernels: a persistent workgroup per CU keeps the 256 x 128 weight block in LDS for the whole launch, streams
// tiles of 128 pixels - the next tile's 32 KiB arrive by dense 16-byte loads while the current one is computed - and every wave multiplies its
// 32 pixels with v_mfma_f32_32x32x16 (A = weight rows from LDS, B = the pixel rows from LDS, D[channel][pixel]: a lane ends up with 4
// consecutive channels of one pixel), packs to 16 bits and leaves through a per-wave LDS tile so that a store instruction writes 1 KiB of
// contiguous memory.
#include "gmk_common.h"

namespace {

constexpr int kRow = 272;                       // LDS row of 128 16-bit values: 256 B + 16 B (the rows of consecutive lanes fall on different banks)

template <typename T>
__global__ __launch_bounds__(256) void conv1x1_pair_stream_kernel(const T* __restrict__ src, const T* __restrict__ w, T* __restrict__ out_a,
                                                                 T* __restrict__ out_b, unsigned npix, unsigned ntiles, unsigned src_bytes) {
    fp16_saturating_stores<T>();
    typedef typename Frag16<T>::type frag_t;
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4s;
    constexpr unsigned kBadOff = 0xFFFFFF00u;
    __shared__ __attribute__((aligned(16))) char wl[256 * kRow];          // weight rows n = 0 .. 255 (128 k each)
    __shared__ __attribute__((aligned(16))) char tile[128 * kRow];        // 128 pixel rows
    __shared__ __attribute__((aligned(16))) char trans[4 * 32 * kRow];    // per wave: 32 pixels x 256 B of results
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const __amdgpu_buffer_rsrc_t rss = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(src), 0, (int)src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(out_a, 0, (int)src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(out_b, 0, (int)src_bytes, 0x00020000);
    // weights: 256 rows x 16 chunks of 16 B
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = tid + 256 * i;
        *reinterpret_cast<u32x4s*>(wl + (c >> 4) * kRow + (c & 15) * 16) = *reinterpret_cast<const u32x4s*>(reinterpret_cast<const char*>(w) + (size_t)c * 16);
    }
    auto load_tile = [&](unsigned t, u32x4s (&br)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {          // chunk c = tid + 256 i of the tile's 128 x 16 sixteen-byte chunks
            const unsigned c = (unsigned)tid + 256u * i, px = t * 128u + (c >> 4);
            br[i] = __builtin_amdgcn_raw_buffer_load_b128(rss, t < ntiles && px < npix ? t * 32768u + c * 16u : kBadOff, 0, 0);
        }
    };
    u32x4s br[8];
    load_tile(blockIdx.x, br);
    char* tr = trans + wave * (32 * kRow);
    for (unsigned t = blockIdx.x; t < ntiles; t += gridDim.x) {
        __syncthreads();                                  // every wave is done with the previous tile (and, first time, the weights are parked)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = tid + 256 * i;
            *reinterpret_cast<u32x4s*>(tile + (c >> 4) * kRow + (c & 15) * 16) = br[i];
        }
        __syncthreads();
        load_tile(t + gridDim.x, br);                     // in flight under this tile's work
        // B operands: this lane's pixel wave * 32 + r, k = 16 kk + 8 h .. + 7
        frag_t bf[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) bf[kk] = *reinterpret_cast<const frag_t*>(tile + (wave * 32 + r) * kRow + (16 * kk + 8 * h) * 2);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const frag_t af = *reinterpret_cast<const frag_t*>(wl + (half * 128 + nb * 32 + r) * kRow + (16 * kk + 8 * h) * 2);
                    acc = mfma_32x32x16<T>(af, bf[kk], acc);
                }
                // lane (pixel r, half h) holds channels 32 nb + 8 q4 + 4 h + (0..3), q4 = 0..3
                unsigned pk[4][2];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    pk[q4][0] = pack_pair<T>(sat16<T>(acc[4 * q4]), sat16<T>(acc[4 * q4 + 1]));
                    pk[q4][1] = pack_pair<T>(sat16<T>(acc[4 * q4 + 2]), sat16<T>(acc[4 * q4 + 3]));
                }
#pragma unroll
                for (int q4 = 0; q4 < 4; q4 += 2) {       // after the swap: lanes < 32 hold channels 8 q4 .. 8 q4 + 7, lanes >= 32 the next 8
                    const auto s0 = __builtin_amdgcn_permlane32_swap(pk[q4][0], pk[q4 + 1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(pk[q4][1], pk[q4 + 1][1], false, false);
                    const u32x4s o = {s0[0], s1[0], s0[1], s1[1]};
                    *reinterpret_cast<u32x4s*>(tr + r * kRow + (nb * 4 + q4 + h) * 16) = o;
                }
            }
            // the wave's own tile, LDS operations of one wave complete in order: no barrier.  Lane -> (row 4 i + lane / 16, 16-byte chunk lane % 16)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = 4 * i + (lane >> 4);
                const u32x4s v = *reinterpret_cast<const u32x4s*>(tr + row * kRow + (lane & 15) * 16);
                const unsigned px = t * 128u + (unsigned)(wave * 32 + row);
                __builtin_amdgcn_raw_buffer_store_b128(v, half ? rsb : rsa, px < npix ? px * 256u + (unsigned)(lane & 15) * 16u : kBadOff, 0, 0);
            }
        }
    }
}

}  // namespace

// 1 if the launch was taken (16-bit, 128 input channels, enough tiles for one per CU and 32-bit offsets), 0 otherwise
int gmk_conv1x1_pair_stream_try(const void* src, int64_t npix, const void* w_rows256, void* out_a, void* out_b, int dtype, hipStream_t stream) {
    if (!gmk_is16(dtype) || gmk_kernel_choice(3, "GMK_DEV_VARIANT") == 46) return 0;
    const int64_t bytes = npix * 256;
    if (bytes >= 0xFFFFFF00ll) return 0;
    const int64_t ntiles = (npix + 127) / 128;
    const int ncu = gmk_cu_limit();
    if (ntiles < 2 * ncu) return 0;                       // small problems: the general kernel (or two plain launches)
    const unsigned grid = (unsigned)ncu;
    if (dtype == GMK_BF16)
        conv1x1_pair_stream_kernel<bf16_t><<<grid, 256, 0, stream>>>((const bf16_t*)src, (const bf16_t*)w_rows256, (bf16_t*)out_a, (bf16_t*)out_b,
                                                                    (unsigned)npix, (unsigned)ntiles, (unsigned)bytes);
    else
        conv1x1_pair_stream_kernel<f16_t><<<grid, 256, 0, stream>>>((const f16_t*)src, (const f16_t*)w_rows256, (f16_t*)out_a, (f16_t*)out_b,
                                                                   (unsigned)npix, (unsigned)ntiles, (unsigned)bytes);
    return 1;
}
