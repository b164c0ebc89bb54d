// The data gradient of the up-path ResBlocks' 1x1 skip convolution over the concatenated input (reference: autograd of
// `skip_connection = nn.Conv2d(2C, C, 1)`, gms/diffusion/simple_unet.py:176,186): out_a | out_b [pixel][128] = dout[pixel][128] . Wd[128][256],
// a [pixels x 128] x [128 x 256] GEMM that moves 3 N bytes for 128 k of work per output - bound by memory, not by the matrix cores.
// The general LDS-DMA convolution kernel runs it at 3.2 TB/s (two K-steps per 256-pixel tile: its pipeline never fills).  Here, as in the
// stem kernels (docs/EXPERIMENTS.md section 7b.4): a persistent workgroup per CU keeps the 256 x 128 weight block in LDS for the whole launch, streams
// tiles of 128 pixels - the next tile's 32 KiB arrive by dense 16-byte loads while the current one is computed - and every wave multiplies its
// 32 pixels with v_mfma_f32_32x32x16 (A = weight rows from LDS, B = the pixel rows from LDS, D[channel][pixel]: a lane ends up with 4
// consecutive channels of one pixel), packs to 16 bits and leaves through a per-wave LDS tile so that a store instruction writes 1 KiB of
// contiguous memory.
#include <type_traits>

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

// ---------------------------------------------------------------------------------------------------------------------------------------
// The WEIGHT gradient of the same skip convolution: dWs[co][ci] = sum_pixels dout[pixel][co] * x_cat[pixel][ci], x_cat = [src0 | src1] (128 + 128
// channels), again 3 N bytes for little work.  The general im2col kernel (conv_wgrad_kernel) gives every 128-channel input block its own
// workgroup per pixel range, so dout goes through the vector-memory path twice (the second time from L2): 4 N at the CUs for 3 N of operands.
// Here one persistent 8-wave workgroup per CU owns a pixel range and the whole 128 x 256 result: per 64-pixel step the 16 KiB of dout and the
// 2 x 16 KiB of x_cat arrive once (dense 16-byte loads, two steps in flight in registers), are parked in the LDS layout of conv_wgrad_kernel and read
// back transposed (`ds_read_b64_tr_b16`: the pixel is the contraction index), wave (wm, wq) accumulates 64 co x 64 ci of input block wq / 2.
// fp16 activations are re-rounded to bf16 on their way into LDS, as there.  One fp32 slab per workgroup, the deterministic reduce of conv_igemm.hip.
template <typename TX>
__global__ __launch_bounds__(512) void conv1x1_wgrad_stream_kernel(const bf16_t* __restrict__ dy, int dy_cstride, const TX* __restrict__ src0,
                                                                  const TX* __restrict__ src1, float* __restrict__ slab, int M, int chunk) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4s;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    __shared__ __attribute__((aligned(16))) char smem[2 * 3 * 16384];     // [2 buffers][dY | X0 | X1][64 px][256 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wq = wave & 3, cib = wq >> 1, wn = wq & 1;
    const int pix_begin = blockIdx.x * chunk;
    const int pix_end = min(pix_begin + chunk, M);
    const int nk = pix_end > pix_begin ? (pix_end - pix_begin + 63) / 64 : 0;
    // staging: chunk c = tid + 512 j (j < 2) of a tile's 64 x 16 sixteen-byte chunks: row c / 16, the swizzle of conv_wgrad_kernel
    int lds_w[2], srow[2];
    const int sc = tid & 15;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (tid >> 4) + 32 * j;
        srow[j] = row;
        lds_w[j] = row * 256 + ((((sc >> 2) ^ (row & 3)) << 2 | (sc & 3)) << 4);
    }
    u32x4s ry[2][2], rx0[2][2], rx1[2][2];
    auto load_step = [&](int ks, u32x4s (&y)[2], u32x4s (&x0)[2], u32x4s (&x1)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = pix_begin + ks * 64 + srow[j];
            u32x4s vy = {0u, 0u, 0u, 0u}, v0 = vy, v1 = vy;
            if (m < pix_end) {
                vy = *reinterpret_cast<const u32x4s*>(dy + (int64_t)m * dy_cstride + sc * 8);
                v0 = *reinterpret_cast<const u32x4s*>(src0 + (int64_t)m * 128 + sc * 8);
                v1 = *reinterpret_cast<const u32x4s*>(src1 + (int64_t)m * 128 + sc * 8);
            }
            y[j] = vy; x0[j] = v0; x1[j] = v1;
        }
    };
    auto round_x = [&](u32x4s v) {
        if constexpr (!__is_same(TX, bf16_t)) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                float lo, hi;
                unpack_pair<TX>(v[d], lo, hi);
                v[d] = pack_pair<bf16_t>(lo, hi);
            }
        }
        return v;
    };
    auto write_step = [&](int buf, const u32x4s (&y)[2], const u32x4s (&x0)[2], const u32x4s (&x1)[2]) {
        char* b = smem + buf * 49152;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            *reinterpret_cast<u32x4s*>(b + lds_w[j]) = y[j];
            *reinterpret_cast<u32x4s*>(b + 16384 + lds_w[j]) = round_x(x0[j]);
            *reinterpret_cast<u32x4s*>(b + 32768 + lds_w[j]) = round_x(x1[j]);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // transposed-read lane geometry: 16-lane group gg, lane-in-group 4 q + pp
    const int gg = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
    const int hh = gg >> 1, cblk = gg & 1;
    const int r = lane & 31, h = lane >> 5;
    if (nk > 0) {
        load_step(0, ry[0], rx0[0], rx1[0]);
        if (nk > 1) load_step(1, ry[1], rx0[1], rx1[1]);
        write_step(0, ry[0], rx0[0], rx1[0]);
    }
    __syncthreads();
    auto k_step = [&](int ks, auto par) {
        constexpr int PAR = decltype(par)::value;          // ks & 1: LDS buffer of this step, register set of step ks + 2
        if (ks + 2 < nk) load_step(ks + 2, ry[PAR], rx0[PAR], rx1[PAR]);
        const char* Yb = smem + PAR * 49152;
        const char* Xb = Yb + 16384 + cib * 16384;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 a[2], b[2];
            const int pxl = kk * 16 + 8 * hh + q;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int wa = (((wm * 2 + i) ^ q) << 6) + 32 * cblk + 8 * pp;
                const int wb = (((wn * 2 + i) ^ q) << 6) + 32 * cblk + 8 * pp;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)(Yb + pxl * 256 + wa));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)(Yb + (pxl + 4) * 256 + wa));
                const s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                a[i] = __builtin_bit_cast(bf16x8, t);
                lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)(Xb + pxl * 256 + wb));
                hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)(Xb + (pxl + 4) * 256 + wb));
                const s16x8 u = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                b[i] = __builtin_bit_cast(bf16x8, u);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (ks + 1 < nk) write_step(PAR ^ 1, ry[PAR ^ 1], rx0[PAR ^ 1], rx1[PAR ^ 1]);
        __syncthreads();
    };
    for (int ks = 0; ks < nk; ks += 2) {
        k_step(ks, std::integral_constant<int, 0>{});
        if (ks + 1 < nk) k_step(ks + 1, std::integral_constant<int, 1>{});
    }
    // slab[workgroup][co][ci] (ci over 256)
    float* out = slab + (int64_t)blockIdx.x * (128 * 256) + cib * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int ci = wn * 64 + j * 32 + r;
                out[co * 256 + ci] = acc[i][j][e];
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

// Number of slabs written (> 0) if the launch was taken: bf16 gradients, 128 + 128 input channels, 128 output channels, enough 64-pixel steps
int gmk_conv1x1_wgrad_stream_try(const void* dy, int dy_cstride, const void* src0, const void* src1, int64_t npix, float* slab, int64_t slab_bytes,
                                 bool x_f16, hipStream_t stream) {
    if (gmk_kernel_choice(3, "GMK_DEV_VARIANT") == 47) return 0;
    const int ncu = gmk_cu_limit();
    if (npix < (int64_t)ncu * 64 * 8 || npix >= (1ll << 31) - 64) return 0;        // at least 8 steps per workgroup
    if (slab_bytes < (int64_t)ncu * 128 * 256 * 4) return 0;
    int64_t chunk = (npix + ncu - 1) / ncu;
    chunk = (chunk + 63) / 64 * 64;
    const int ns = (int)((npix + chunk - 1) / chunk);
    if (x_f16)
        conv1x1_wgrad_stream_kernel<f16_t><<<ns, 512, 0, stream>>>((const bf16_t*)dy, dy_cstride, (const f16_t*)src0, (const f16_t*)src1, slab, (int)npix, (int)chunk);
    else
        conv1x1_wgrad_stream_kernel<bf16_t><<<ns, 512, 0, stream>>>((const bf16_t*)dy, dy_cstride, (const bf16_t*)src0, (const bf16_t*)src1, slab, (int)npix, (int)chunk);
    return ns;
}
