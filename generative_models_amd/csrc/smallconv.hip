// Stem (C_in <= 4 -> C) and head (C -> C_out <= 4) 3x3 convolutions with their backward passes.
// Reference: gms/diffusion/simple_unet.py:92-94 (Downsample(1, channels, 1)) and :41 (Conv2d(channels, 1, 3, padding=1)).
// Degenerate GEMM shapes (K = 9 or N = 1): HBM-bound streaming kernels, no MFMA.  Image-side tensors are NCHW
// fp32 as the reference passes them; the network side is NHWC in `dtype`.
//
// All five kernels move the C-channel tensor exactly once (16 B per lane, a wave covers whole 256-B pixel rows) and
// take the 1-channel image through L1/L2; a thread is (8-channel vector, pixel lane) and walks the flat pixel range of
// its block with 32-bit counters (one division per thread, then increments — per-pixel 64-bit div/mod had made the
// first version of these kernels 5x slower than their HBM time).  A thread's weights / accumulators for ALL image channels live in
// registers: 8 network channels per thread for 1-2 image channels, 4 for 3-4 (CIFAR-shape configs).
#include "gmk_common.h"

namespace {

constexpr int kMaxSmall = 4;   // max image channels handled (1 in the reference, 3 for the CIFAR-shape configs)

__device__ __forceinline__ int div_small(int a, float inv) { return (int)(((float)a + 0.5f) * inv); }

// flat pixel g = b*HW + p, advanced by `step` (< HW or not) with increments only
struct PixWalk {
    int b, p;
    __device__ __forceinline__ PixWalk(unsigned g, int HW) { b = (int)(g / (unsigned)HW); p = (int)(g - (unsigned)b * (unsigned)HW); }
    __device__ __forceinline__ void advance(int step, int HW) {
        p += step;
        while (p >= HW) { p -= HW; ++b; }
    }
};

// sum over the 16 lanes of a DPP row; every lane of the row ends with the total
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

template <int VW> struct VecIO;
template <> struct VecIO<8> {
    template <typename T> static __device__ __forceinline__ void load(const T* p, float (&v)[8]) { load8(p, v); }
    template <typename T> static __device__ __forceinline__ void store(T* p, const float (&v)[8]) { store8(p, v); }
};
template <> struct VecIO<4> {
    template <typename T> static __device__ __forceinline__ void load(const T* p, float (&v)[4]) { load4(p, v); }
    template <typename T> static __device__ __forceinline__ void store(T* p, const float (&v)[4]) { store4(p, v); }
};

// ---- 1 -> C "expand" convolution: stem forward (FLIP = false) and head data gradient (FLIP = true) ---------
//   stem:  y[b,p,c]  = bias[c] + sum_{s,t} x[b,s,p + off(t)]    * w[c][s][t]        w: [C][cs][3][3]
//   head:  da[b,p,c] =           sum_{s,t} dout[b,s,p - off(t)] * w[s][c][t]        w: [cs][C][3][3]
// CSN image channels, VW network channels per thread (8 for 1-2 image channels, 4 for 3-4: the CSN x 9 x VW weights of a
// thread always live in registers)
template <typename T, int CSN, int VW, bool FLIP>
__global__ __launch_bounds__(256) void expand3x3_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                       const float* __restrict__ bias, T* __restrict__ out, int H, int W, int C,
                                                       float inv_w, unsigned npix, int ppb) {
    fp16_saturating_stores<T>();
    const int tid = threadIdx.x;
    const int nvec = C / VW, planes = 256 / nvec;
    const int vec = tid % nvec, pl = tid / nvec;
    const int HW = H * W;
    float wr[CSN][9][VW], bs[VW];
#pragma unroll
    for (int s = 0; s < CSN; ++s)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < VW; ++i) {
                const int c = vec * VW + i;
                wr[s][t][i] = w[FLIP ? (s * C + c) * 9 + t : (c * CSN + s) * 9 + t];
            }
#pragma unroll
    for (int i = 0; i < VW; ++i) bs[i] = bias ? bias[vec * VW + i] : 0.f;

    const unsigned g0 = blockIdx.x * (unsigned)ppb;
    const unsigned g1 = min(g0 + (unsigned)ppb, npix);
    if (g0 + pl >= g1) return;
    PixWalk pw(g0 + pl, HW);
    for (unsigned g = g0 + pl; g < g1; g += planes) {
        const int oy = div_small(pw.p, inv_w), ox = pw.p - oy * W;
        float acc[VW];
#pragma unroll
        for (int i = 0; i < VW; ++i) acc[i] = bs[i];
#pragma unroll
        for (int s = 0; s < CSN; ++s) {
            const float* ip = in + (size_t)(pw.b * CSN + s) * HW;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = FLIP ? oy + 1 - ky : oy + ky - 1;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = FLIP ? ox + 1 - kx : ox + kx - 1;
                    const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                    const float xv = ok ? ip[iy * W + ix] : 0.f;
#pragma unroll
                    for (int i = 0; i < VW; ++i) acc[i] = fmaf(xv, wr[s][ky * 3 + kx][i], acc[i]);
                }
            }
        }
        VecIO<VW>::store(out + (size_t)g * C + vec * VW, acc);
        pw.advance(planes, HW);
    }
}


// ---- the same "expand" convolution on the matrix cores, exact fp32 --------------------------------------------------------
// y^T[c][px] = bias[c] + sum_k W^T[c][k] * patch^T[k][px],  k = s * 9 + t  (K = 9 x image channels = 9 ... 36) as a chain of
// v_mfma_f32_32x32x2_f32 (an exact fp32 fma chain in k order; the bias is one more k entry, added last).  Why:
// the VALU kernel issues 27 image loads and 108 FMAs per 4-channel vector of a 3-channel image and ran at 0.9 TB/s of its only HBM
// traffic, the 128-channel tensor it writes (1.18 ms at 64 x 64, B = 1024).  Here a wave owns 32 pixels x all 128 channels: lane
// (pixel r, half h) loads ONE image value per k pair (14 loads per 32 pixels for 3 channels instead of 27 per 2), the weights and
// the bias sit in registers as MFMA operands for the whole launch, and the D[channel][pixel] orientation lets every lane store 16
// contiguous bytes after one v_permlane32_swap per dword (bf16) or directly (fp32).
template <typename T, int CSN, bool FLIP>
__global__ __launch_bounds__(256) void expand3x3_mfma_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                            const float* __restrict__ bias, T* __restrict__ out, int H, int W, int C,
                                                            unsigned npix, unsigned nblocks, unsigned in_bytes, unsigned out_bytes) {
    fp16_saturating_stores<T>();
    constexpr int K = 9 * CSN, NK = (K + (FLIP ? 0 : 1) + 1) / 2;     // the stem's bias rides as one more k entry: weight bias[c], patch 1
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int HW = H * W;
    // per-lane k of every step: image plane offset and (dy, dx); k >= K contributes nothing
    int koff[NK], kdy[NK], kdx[NK];
    float wa[4][NK];
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        const int k = 2 * ks + h;
        const bool kin = k < K;
        const int s_ = kin ? k / 9 : 0, t = kin ? k % 9 : 4;
        const int dy = FLIP ? 1 - t / 3 : t / 3 - 1, dx = FLIP ? 1 - t % 3 : t % 3 - 1;
        kdy[ks] = kin ? dy : 2 * H;                     // out-of-range row: the value is dropped
        kdx[ks] = dx;
        koff[ks] = s_ * HW + dy * W + dx;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int c = cb * 32 + r;
            wa[cb][ks] = kin ? w[FLIP ? (s_ * C + c) * 9 + t : (c * CSN + s_) * 9 + t] : (!FLIP && k == K && bias) ? bias[c] : 0.f;
        }
    }
    constexpr unsigned kBadOff = 0xFFFFFF00u;
    const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)out_bytes, 0x00020000);
    // the image values of block `blk` for this lane; issued one block ahead so the gathers fly under the previous block's MFMAs
    auto gather = [&](unsigned blk, float (&pv)[NK]) {
        const unsigned g = blk * 32 + r;
        const bool live = blk < nblocks && g < npix;
        const unsigned gg = live ? g : 0;
        const int b = (int)(gg / (unsigned)HW), pp = (int)(gg - (unsigned)b * (unsigned)HW);
        const int y = pp / W, x = pp - y * W;
        const int base = b * CSN * HW + pp;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {     // out-of-range offsets read as 0 (buffer semantics): no branches around the loads
            const int ok = (int)live & (int)((unsigned)(y + kdy[ks]) < (unsigned)H) & (int)((unsigned)(x + kdx[ks]) < (unsigned)W);
            pv[ks] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsi, ok ? (unsigned)(base + koff[ks]) * 4u : kBadOff, 0, 0));
            if (!FLIP && (2 * ks == K || 2 * ks + 1 == K) && 2 * ks + h == K) pv[ks] = 1.f;
        }
    };
    float pn[NK];
    gather(blockIdx.x * 4 + wave, pn);
    for (unsigned blk = blockIdx.x * 4 + wave; blk < nblocks; blk += gridDim.x * 4) {
        const unsigned g = blk * 32 + r;
        const bool live = g < npix;
        float pv[NK];
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) pv[ks] = pn[ks];
        gather(blk + gridDim.x * 4, pn);
        f32x16 acc[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[cb][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[cb][ks], pv[ks], acc[cb], 0, 0, 0);
        }
        // lane (pixel r, half h) holds channels 32 cb + 8 q4 + 4 h + (0..3), q4 = 0..3
        const unsigned row_b = g * (unsigned)C * (unsigned)sizeof(T);
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4s;
        if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 v = {acc[cb][4 * q4], acc[cb][4 * q4 + 1], acc[cb][4 * q4 + 2], acc[cb][4 * q4 + 3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, v), rso,
                                                           live ? row_b + (unsigned)(cb * 32 + 8 * q4 + 4 * h) * 4u : kBadOff, 0, 0);
                }
        } else {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                unsigned pk[4][2];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    pk[q4][0] = pack_pair<T>(sat16<T>(acc[cb][4 * q4]), sat16<T>(acc[cb][4 * q4 + 1]));
                    pk[q4][1] = pack_pair<T>(sat16<T>(acc[cb][4 * q4 + 2]), sat16<T>(acc[cb][4 * q4 + 3]));
                }
#pragma unroll
                for (int q4 = 0; q4 < 4; q4 += 2) {       // after the swap: lanes < 32 hold channels 8 q4 .. 8 q4 + 7, lanes >= 32 the next 8
                    const auto s0 = __builtin_amdgcn_permlane32_swap(pk[q4][0], pk[q4 + 1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(pk[q4][1], pk[q4 + 1][1], false, false);
                    const u32x4s o = {s0[0], s1[0], s0[1], s1[1]};
                    __builtin_amdgcn_raw_buffer_store_b128(o, rso, live ? row_b + (unsigned)(cb * 32 + 8 * (q4 + h)) * 2u : kBadOff, 0, 0);
                }
            }
        }
    }
}

// ---- the "expand" convolution for 16-bit outputs, round 4: tiles of 128 pixels, image window in LDS, bf16 hi + lo operands ----------------
// What binds the exact-fp32 kernel above is not one thing (timing-only builds, docs/EXPERIMENTS.md section 7b.4): with a ninth of the MFMA cycles it is
// 9 % faster, with half of its gathers no faster, with dense stores no faster, with neither gathers nor MFMAs it writes at 5.2 TB/s - the
// per-lane image gathers (16 four-byte loads per 32 pixels through the texture path) and the fp32 MFMA chain (56 x 64 cycles per block) each
// cap it near 2.7 / 4.8 TB/s.  This form removes both: a workgroup owns 128 consecutive pixels of ONE sample, the window of image values they
// touch (128 + 2 W + 2 per image channel) is loaded with coalesced loads one tile ahead and parked in LDS, the patch values are LDS reads;
// both operands are split x = hi + lo into bf16 (2^-17) and multiplied as hi hi + hi lo + lo hi on v_mfma_f32_32x32x16_bf16 (results equal to
// the exact chain to the last 16-bit ulp); the results leave through a per-wave LDS tile so that a store instruction writes 1 KiB of contiguous
// memory.  k = 16 ks + 8 h + e as in the MFMA's operand layout.
constexpr int kTileWin = 128 + 2 * 128 + 2;            // window floats per image channel at W <= 128
template <typename T, int CSN, bool FLIP>
__global__ __launch_bounds__(256, 2) void expand3x3_tile16_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                                 const float* __restrict__ bias, T* __restrict__ out, int H, int W, int C,
                                                                 int tiles_per_sample, unsigned ntiles, unsigned in_bytes, unsigned out_bytes) {
    static_assert(sizeof(T) == 2, "16-bit outputs only");
    fp16_saturating_stores<T>();
    constexpr int K = 9 * CSN, KB = K + (FLIP ? 0 : 1), NKS = (KB + 15) / 16, NE = NKS * 8;
    constexpr int kTrRow = 272;
    __shared__ float win[2][CSN][kTileWin];
    __shared__ __attribute__((aligned(16))) char trans[4 * 32 * kTrRow];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int HW = H * W, WIN = 128 + 2 * W + 2;
    bf16x8 whi[4][NKS], wlo[4][NKS];
    int kinfo[NE];              // per k entry of this lane: window offset << 6 | (dy + 1) | (dx + 1) << 2 | inside K << 4 | the bias entry << 5
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 16 * ks + 8 * h + e;
            const bool kin = k < K;
            const int s_ = kin ? k / 9 : 0, t = kin ? k % 9 : 4;
            const int dy = FLIP ? 1 - t / 3 : t / 3 - 1, dx = FLIP ? 1 - t % 3 : t % 3 - 1;
            kinfo[ks * 8 + e] = ((s_ * kTileWin + dy * W + dx) * 64) | (dy + 1) | ((dx + 1) << 2) | ((int)kin << 4) | ((int)(!FLIP && k == K) << 5);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const int c = cb * 32 + r;
                const float wv = kin ? w[FLIP ? (s_ * C + c) * 9 + t : (c * CSN + s_) * 9 + t] : (!FLIP && k == K && bias) ? bias[c] : 0.f;
                const bf16_t hi = (bf16_t)wv;
                whi[cb][ks][e] = hi;
                wlo[cb][ks][e] = (bf16_t)(wv - (float)hi);
            }
        }
    constexpr unsigned kBadOff = 0xFFFFFF00u;
    const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)out_bytes, 0x00020000);
    // window element i of tile (b, pp0): plane pixel pp0 - (W + 1) + i, zero outside the plane (those taps are masked anyway)
    auto load_window = [&](unsigned tile, float (&wr)[CSN][2]) {
        const int b = (int)(tile / (unsigned)tiles_per_sample), pp0 = ((int)tile - b * tiles_per_sample) * 128;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + 256 * j, flat = pp0 - (W + 1) + i;
            const bool ok = tile < ntiles && i < WIN && flat >= 0 && flat < HW;
#pragma unroll
            for (int s_ = 0; s_ < CSN; ++s_)
                wr[s_][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsi, ok ? (unsigned)((b * CSN + s_) * HW + flat) * 4u : kBadOff, 0, 0));
        }
    };
    auto park_window = [&](int buf, const float (&wr)[CSN][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + 256 * j;
            if (i < WIN) {
#pragma unroll
                for (int s_ = 0; s_ < CSN; ++s_) win[buf][s_][i] = wr[s_][j];
            }
        }
    };
    float wr[CSN][2];
    load_window(blockIdx.x, wr);
    park_window(0, wr);
    __syncthreads();
    int buf = 0;
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4s;
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        load_window(tile + gridDim.x, wr);               // in flight under this tile's work
        const int b = (int)(tile / (unsigned)tiles_per_sample), pp0 = ((int)tile - b * tiles_per_sample) * 128;
        const int pl = wave * 32 + r, pp = pp0 + pl;
        const bool live = pp < HW;
        const int y = pp / W, x = pp - y * W;
        const float* wb = &win[buf][0][0] + pl + (W + 1);
        bf16x8 phi[NKS], plo[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int m = kinfo[ks * 8 + e];
                const int ok = (int)live & (m >> 4) & (int)((unsigned)(y + (m & 3) - 1) < (unsigned)H) & (int)((unsigned)(x + ((m >> 2) & 3) - 1) < (unsigned)W);
                float v = wb[m >> 6];
                v = (ok & 1) ? v : 0.f;
                v = (m & 32) ? 1.f : v;                  // the bias entry multiplies a patch value of 1
                const bf16_t hi = (bf16_t)v;
                phi[ks][e] = hi;
                plo[ks][e] = (bf16_t)(v - (float)hi);
            }
        char* tr = trans + wave * (32 * kTrRow);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {           // small terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[cb][ks], phi[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[cb][ks], plo[ks], acc, 0, 0, 0);
            }
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[cb][ks], phi[ks], acc, 0, 0, 0);
            // lane (pixel r, half h) holds channels 32 cb + 8 q4 + 4 h + (0..3), q4 = 0..3
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
                *reinterpret_cast<u32x4s*>(tr + r * kTrRow + (cb * 4 + q4 + h) * 16) = o;
            }
        }
        // the wave's own tile, LDS operations of one wave complete in order: no barrier.  Lane -> (row 4 i + lane / 16, 16-byte chunk lane % 16)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 4 * i + (lane >> 4);
            const u32x4s v = *reinterpret_cast<const u32x4s*>(tr + row * kTrRow + (lane & 15) * 16);
            const int ppr = pp0 + wave * 32 + row;
            __builtin_amdgcn_raw_buffer_store_b128(v, rso, ppr < HW ? ((unsigned)(b * HW + ppr) * 256u + (unsigned)(lane & 15) * 16u) : kBadOff, 0, 0);
        }
        park_window(buf ^ 1, wr);
        __syncthreads();
    }
}

// ---- weight gradients of both: the C-channel tensor is read ONCE (all image channels accumulate together), the image
// through L1 -----------------------------------------------------------------------------------------------------------
//   stem (FLIP = false): dw[c][s][t] = sum_{b,p} x[b,s,p + off(t)]    * dy[b,p,c]
//   head (FLIP = true):  dw[s][c][t] = sum_{b,p} dout[b,s,p - off(t)] * a[b,p,c];  db[s] = sum dout[b,s,p]
// part: [nblk][cs*C*9 (+ cs bias sums when FLIP)] per-block partials in the reference's weight layout.
template <typename T, int CSN, int VW, bool FLIP>
__global__ __launch_bounds__(256) void wgrad3x3_kernel(const float* __restrict__ small, const T* __restrict__ big,
                                                      float* __restrict__ part, int H, int W, int C, float inv_w, unsigned npix,
                                                      int ppb, int cs, int s0) {
    __shared__ float red[256 * VW];
    const int tid = threadIdx.x;
    const int nvec = C / VW, planes = 256 / nvec;
    const int vec = tid % nvec, pl = tid / nvec;
    const int HW = H * W;
    const unsigned g0 = blockIdx.x * (unsigned)ppb;
    const unsigned g1 = min(g0 + (unsigned)ppb, npix);
    // this launch covers image channels s0 .. s0+CSN-1 of cs
    float* out = part + (size_t)blockIdx.x * ((size_t)cs * C * 9 + (FLIP ? cs : 0));
    float acc[CSN][9][VW];
    float bsum[CSN];
#pragma unroll
    for (int s = 0; s < CSN; ++s) {
        bsum[s] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < VW; ++i) acc[s][t][i] = 0.f;
    }
    if (g0 + pl < g1) {
        PixWalk pw(g0 + pl, HW);
        for (unsigned g = g0 + pl; g < g1; g += planes) {
            const int oy = div_small(pw.p, inv_w), ox = pw.p - oy * W;
            float v[VW];
            VecIO<VW>::load(big + (size_t)g * C + vec * VW, v);
#pragma unroll
            for (int s = 0; s < CSN; ++s) {
                const float* sp = small + (size_t)(pw.b * cs + s0 + s) * HW;
                if (FLIP) bsum[s] += sp[pw.p];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int iy = FLIP ? oy + 1 - ky : oy + ky - 1;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int ix = FLIP ? ox + 1 - kx : ox + kx - 1;
                        const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                        const float sv = ok ? sp[iy * W + ix] : 0.f;
#pragma unroll
                        for (int i = 0; i < VW; ++i) acc[s][ky * 3 + kx][i] = fmaf(sv, v[i], acc[s][ky * 3 + kx][i]);
                    }
                }
            }
            pw.advance(planes, HW);
        }
    }
#pragma unroll
    for (int s = 0; s < CSN; ++s) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < VW; ++i) red[tid * VW + i] = acc[s][t][i];
            __syncthreads();
            if (tid < C) {
                const int vv = tid / VW, i = tid % VW;
                float sum = 0.f;
                for (int q = 0; q < planes; ++q) sum += red[(q * nvec + vv) * VW + i];
                out[FLIP ? ((size_t)(s0 + s) * C + tid) * 9 + t : ((size_t)tid * cs + s0 + s) * 9 + t] = sum;
            }
        }
        if (FLIP) {     // bias partial: only vec == 0 lanes hold distinct pixels' dout
            __syncthreads();
            red[tid] = vec == 0 ? bsum[s] : 0.f;
            __syncthreads();
            if (tid == 0) {
                float sum = 0.f;
                for (int i = 0; i < 256; ++i) sum += red[i];
                out[(size_t)cs * C * 9 + s0 + s] = sum;
            }
        }
    }
}


// ---- the same weight gradients on the fp32 matrix cores ---------------------------------------------------------------------
// dW^T[c][j] = sum_px big[px][c] * patch[px][j],  j = s * 9 + t < 32 (up to 3 image channels): the PIXEL is the contraction index of
// v_mfma_f32_32x32x2_f32 (exact fp32 products and sums), two pixels per instruction.  A operand: lane (r, h) holds channel
// 32 cb + r of pixel 2 q + h - a 2- or 4-byte load whose 32 lanes cover 64 / 128 contiguous bytes of the pixel's row; B operand: lane
// (r, h) holds the image value of tap entry j = r at that pixel (zero outside the image / for j >= K).  A wave walks a contiguous
// range of pixel pairs with 8 pairs (40 loads) in flight ahead of the 32 MFMAs that consume them and keeps the whole 128 x 32
// gradient in 64 accumulator registers; the four waves of a workgroup are summed through LDS into ONE partial row, and the grid is
// exactly the resident set, so `part` has <= 512 rows instead of one per 512-1024 pixels.  The VALU kernel above ran at 0.8-1.0 TB/s
// (one launch per image channel, 9 LDS reduction rounds each); this one is bound by the MFMA issue rate (4 per pixel pair).
template <typename T, int CSN, bool FLIP>
__global__ __launch_bounds__(256) void wgrad3x3_mfma_kernel(const float* __restrict__ small, const T* __restrict__ big,
                                                           float* __restrict__ part, int H, int W, int C, unsigned npix,
                                                           unsigned small_bytes, unsigned big_bytes, unsigned pairs_per_wave) {
    constexpr int K = 9 * CSN, U = 8;
    constexpr unsigned kBadOff = 0xFFFFFF00u;
    __shared__ float red[4][128 * 32];
    __shared__ float bred[4][4][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int HW = H * W;
    const __amdgpu_buffer_rsrc_t rss = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(small), 0, (int)small_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(big), 0, (int)big_bytes, 0x00020000);
    const bool kin = r < K;
    const int s_ = kin ? r / 9 : 0, t = kin ? r % 9 : 4;
    const int dy = FLIP ? 1 - t / 3 : t / 3 - 1, dx = FLIP ? 1 - t % 3 : t % 3 - 1;
    const int koff = s_ * HW + dy * W + dx;
    const unsigned npairs = (npix + 1) / 2;
    const unsigned q0 = (blockIdx.x * 4 + wave) * pairs_per_wave;
    const unsigned q1 = q0 + pairs_per_wave < npairs ? q0 + pairs_per_wave : npairs;
    // walking state of THIS lane's pixel g = 2 q + h (W is even, so a pair never straddles a row)
    unsigned g = 2 * q0 + h;
    int b0 = (int)(g / (unsigned)HW), pp = (int)(g - (unsigned)b0 * (unsigned)HW);
    int y = pp / W, x = pp - y * W;
    int base = b0 * CSN * HW + pp;
    float bsum = 0.f;
    f32x16 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[cb][e] = 0.f;

    auto fetch = [&](unsigned q, float (&av)[4], float& bv) {
        const int live = (int)(q < q1) & (int)(g < npix);
        const int ok = live & (int)kin & (int)((unsigned)(y + dy) < (unsigned)H) & (int)((unsigned)(x + dx) < (unsigned)W);
        bv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rss, ok ? (unsigned)(base + koff) * 4u : kBadOff, 0, 0));
        const unsigned ao = live ? (g * (unsigned)C + (unsigned)r) * (unsigned)sizeof(T) : kBadOff;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            if constexpr (__is_same(T, f16_t))
                av[cb] = (float)__builtin_bit_cast(f16_t, (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsb, ao, cb * 64, 0));
            else if constexpr (sizeof(T) == 2)
                av[cb] = __builtin_bit_cast(float, (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsb, ao, cb * 64, 0) << 16);
            else
                av[cb] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsb, ao, cb * 128, 0));
        }
        // next pair of this lane
        g += 2; base += 2; x += 2;
        const int wrapx = x >= W;
        x -= wrapx ? W : 0; y += wrapx;
        const int wrapy = y >= H;
        y = wrapy ? 0 : y; base += wrapy ? (CSN - 1) * HW : 0;
    };

    float an[U][4], bn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) fetch(q0 + u, an[u], bn[u]);
    for (unsigned q = q0; q < q1; q += U) {
        float ac[U][4], bc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bc[u] = bn[u];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) ac[u][cb] = an[u][cb];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) fetch(q + U + u, an[u], bn[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (FLIP) bsum += t == 4 ? bc[u] : 0.f;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u][cb], bc[u], acc[cb], 0, 0, 0);
        }
    }
    // lane (j = r, h) holds dW^T[32 cb + (e & 3) + 8 (e >> 2) + 4 h][j]
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wave][(cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = acc[cb][e];
    if (FLIP && kin && t == 4) bred[wave][s_][h] = bsum;
    __syncthreads();
    const size_t rowlen = (size_t)CSN * C * 9 + (FLIP ? CSN : 0);
    float* out = part + (size_t)blockIdx.x * rowlen;
    for (int idx = tid; idx < CSN * 128 * 9; idx += 256) {
        int c, j;
        if (FLIP) { const int s = idx / (128 * 9), rem = idx - s * 128 * 9; c = rem / 9; j = s * 9 + (rem - c * 9); }
        else { c = idx / K; j = idx - c * K; }
        out[idx] = (red[0][c * 32 + j] + red[1][c * 32 + j]) + (red[2][c * 32 + j] + red[3][c * 32 + j]);
    }
    if (FLIP && tid < CSN) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) sum += bred[w][tid][0] + bred[w][tid][1];
        out[(size_t)CSN * C * 9 + tid] = sum;
    }
}

// ---- the same weight gradients for 16-bit tensors, round 4: 128-pixel tiles through LDS, 16-bit matrix cores -------------------------------
// The fp32 form above walks pixel PAIRS with one 2-byte load per lane and channel block (80 load instructions per 32 pixels) and four fp32 MFMAs
// per pair: 2.4 TB/s of the one tensor it reads.  Here a workgroup stages 128 consecutive pixels of one sample - the 32 KiB of the 16-bit tensor
// with dense 16-byte loads, the image window as in expand3x3_tile16_kernel - one tile ahead in registers, parks them in LDS, and every wave
// contracts its 32 pixels with v_mfma_f32_32x32x16_bf16 (A: the tensor transposed by 2-byte LDS reads - rows = channels, k = pixels; B: the image
// patch of tap entry j = lane % 32).  The fp32 patch is split hi + lo into bf16 (2^-17, and bf16's range: gradients of any scale); a bf16 tensor is
// exact as it is, an fp16 one (the saved activations of the 16-bit mode) is split into bf16 hi + lo too - 8 + 3 significant bits: exact - so
// the products are hi hi + hi lo + lo hi as in the expand kernel.  Accumulators, the reduction over the four waves and the partial-row layout
// are those of the kernel above.
template <typename T, int CSN, bool FLIP>
__global__ __launch_bounds__(256, 2) void wgrad3x3_tile16_kernel(const float* __restrict__ small, const T* __restrict__ big,
                                                                float* __restrict__ part, int H, int W, int C, int tiles_per_sample,
                                                                unsigned ntiles, unsigned small_bytes, unsigned big_bytes) {
    static_assert(sizeof(T) == 2, "16-bit tensors only");
    constexpr int K = 9 * CSN, kRow = 272;                 // LDS row of a pixel: 256 B + 16 B of padding
    constexpr unsigned kBadOff = 0xFFFFFF00u;
    typedef bf16x8 frag_t;
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4s;
    __shared__ __attribute__((aligned(16))) char smem[4 * 128 * 32 * 4];     // tiles while streaming (128 x 272 B + window), then red[4][128 * 32]
    __shared__ float bred[4][4][2];
    char* bigt = smem;                                       // [128 pixels][272 B]
    float* win = reinterpret_cast<float*>(smem + 128 * kRow);     // [CSN][kTileWin]
    static_assert(128 * kRow + 3 * kTileWin * 4 <= (int)sizeof(smem), "tile + window fit the reduction buffer");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int HW = H * W, WIN = 128 + 2 * W + 2;
    const __amdgpu_buffer_rsrc_t rss = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(small), 0, (int)small_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(big), 0, (int)big_bytes, 0x00020000);
    const bool kin = r < K;
    const int s_ = kin ? r / 9 : 0, t = kin ? r % 9 : 4;
    const int dy = FLIP ? 1 - t / 3 : t / 3 - 1, dx = FLIP ? 1 - t % 3 : t % 3 - 1;
    const int woff = s_ * kTileWin + dy * W + dx + (W + 1);       // window index of this lane's tap relative to the pixel's tile position
    float bsum = 0.f;
    f32x16 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[cb][e] = 0.f;

    auto load_tile = [&](unsigned tile, u32x4s (&br)[8], float (&wr)[CSN][2]) {
        const bool on = tile < ntiles;
        const int b = on ? (int)(tile / (unsigned)tiles_per_sample) : 0, pp0 = on ? ((int)tile - b * tiles_per_sample) * 128 : 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {          // chunk c = tid + 256 i of the tile's 128 x 16 sixteen-byte chunks: pixel row c / 16
            const int c = tid + 256 * i, row = c >> 4;
            br[i] = __builtin_amdgcn_raw_buffer_load_b128(rsb, on && pp0 + row < HW ? (unsigned)(b * HW + pp0) * 256u + (unsigned)c * 16u : kBadOff, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + 256 * j, flat = pp0 - (W + 1) + i;
            const bool ok = on && i < WIN && flat >= 0 && flat < HW;
#pragma unroll
            for (int sc = 0; sc < CSN; ++sc)
                wr[sc][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rss, ok ? (unsigned)((b * CSN + sc) * HW + flat) * 4u : kBadOff, 0, 0));
        }
    };
    auto park_tile = [&](const u32x4s (&br)[8], const float (&wr)[CSN][2]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = tid + 256 * i;
            *reinterpret_cast<u32x4s*>(bigt + (c >> 4) * kRow + (c & 15) * 16) = br[i];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + 256 * j;
            if (i < WIN) {
#pragma unroll
                for (int sc = 0; sc < CSN; ++sc) win[sc * kTileWin + i] = wr[sc][j];
            }
        }
    };
    u32x4s br[8];
    float wr[CSN][2];
    load_tile(blockIdx.x, br, wr);
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                                  // every wave is done with the previous tile
        park_tile(br, wr);
        __syncthreads();
        load_tile(tile + gridDim.x, br, wr);              // in flight under this tile's work
        const int b = (int)(tile / (unsigned)tiles_per_sample), pp0 = ((int)tile - b * tiles_per_sample) * 128;
        (void)b;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // B operand: tap entry j = r at pixels pl .. pl + 7
            const int pl = wave * 32 + 16 * ks + 8 * h;
            int pp = pp0 + pl;
            int y = pp / W, x = pp - y * W;
            frag_t bhi, blo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ok = (int)kin & (int)(pp < HW) & (int)((unsigned)(y + dy) < (unsigned)H) & (int)((unsigned)(x + dx) < (unsigned)W);
                float v = win[woff + pl + e];
                v = ok ? v : 0.f;
                if (FLIP) bsum += t == 4 ? v : 0.f;
                const bf16_t hi = (bf16_t)v;
                bhi[e] = hi;
                blo[e] = (bf16_t)(v - (float)hi);
                ++pp; ++x;
                const int wrap = x >= W;
                x = wrap ? 0 : x; y += wrap;
            }
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                // A operand: channel 32 cb + r at the same 8 pixels, transposed out of the tile by 2-byte reads
                frag_t a, alo;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const T av = *reinterpret_cast<const T*>(bigt + (pl + e) * kRow + (cb * 32 + r) * 2);
                    if constexpr (__is_same(T, bf16_t)) a[e] = av;
                    else {
                        const float af = (float)av;
                        const bf16_t hi = (bf16_t)af;
                        a[e] = hi;
                        alo[e] = (bf16_t)(af - (float)hi);
                    }
                }
                if constexpr (!__is_same(T, bf16_t)) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, blo, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bhi, acc[cb], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                      // the tile buffers become the reduction buffer
    float* red = reinterpret_cast<float*>(smem);
    // lane (j = r, h) holds dW^T[32 cb + (e & 3) + 8 (e >> 2) + 4 h][j]
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wave * (128 * 32) + (cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = acc[cb][e];
    if (FLIP && kin && t == 4) bred[wave][s_][h] = bsum;
    __syncthreads();
    const size_t rowlen = (size_t)CSN * C * 9 + (FLIP ? CSN : 0);
    float* out = part + (size_t)blockIdx.x * rowlen;
    for (int idx = tid; idx < CSN * 128 * 9; idx += 256) {
        int c, j;
        if (FLIP) { const int sc = idx / (128 * 9), rem = idx - sc * 128 * 9; c = rem / 9; j = sc * 9 + (rem - c * 9); }
        else { c = idx / K; j = idx - c * K; }
        out[idx] = (red[c * 32 + j] + red[128 * 32 + c * 32 + j]) + (red[2 * 128 * 32 + c * 32 + j] + red[3 * 128 * 32 + c * 32 + j]);
    }
    if (FLIP && tid < CSN) {
        float sum = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) sum += bred[wv][tid][0] + bred[wv][tid][1];
        out[(size_t)CSN * C * 9 + tid] = sum;
    }
}

// ---- head forward (C -> cs): out[b,s,p] = bias[s] + sum_{t,c} a[b,p + off(t),c] * w[s][c][t] ----------------------------
// One workgroup = one band of image rows.  Pass 1 reads every activation vector of the band (+1 row above/below) once and
// leaves its 9 per-tap channel sums in LDS (tap[t][q] = sum_c a[q,c] w[s][c][t], reduced over the pixel's lanes with DPP);
// pass 2 adds the 9 shifted planes per output pixel.  (A gather would read each activation 9 times through L1.)
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ a, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ out, int cs,
                                                      int H, int W, int C, float inv_w, int band, int nbands) {
    extern __shared__ __attribute__((aligned(16))) float tapl[];   // [9][(band+2)*W]
    const int tid = threadIdx.x;
    const int nvec = C >> 3, planes = 256 / nvec;
    const int vec = tid % nvec, pl = tid / nvec;
    const int b = blockIdx.x / nbands, r0 = (blockIdx.x % nbands) * band;
    const int rows = min(band, H - r0);
    const int e0 = max(r0 - 1, 0), e1 = min(r0 + rows + 1, H);       // rows whose activations feed this band
    const int next = (e1 - e0) * W, plane = (band + 2) * W;
    const T* ab = a + ((size_t)b * H + e0) * W * C + vec * 8;
    for (int s = 0; s < cs; ++s) {
        float wr[9][8];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i) wr[t][i] = w[((size_t)s * C + vec * 8 + i) * 9 + t];
        if (s) __syncthreads();
        for (int q0 = 0; q0 < next; q0 += planes) {      // every lane of a wave runs the same trip count (DPP below)
            const int q = q0 + pl;
            float v[8];
            if (q < next) load8(ab + (size_t)q * C, v);
            else {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = 0.f;
            }
            float mine = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float d = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) d = fmaf(v[i], wr[t][i], d);
                d = row16_sum(d);
                if (nvec == 32) { float d2 = d; rows_swap16(d, d2); d += d2; }      // lane ^ 16, off the LDS crossbar
                if (vec == t) mine = d;
            }
            if (vec < 9 && q < next) tapl[vec * plane + q] = mine;
        }
        __syncthreads();
        const float bv = bias[s];
        for (int o = tid; o < rows * W; o += 256) {
            const int oyl = div_small(o, inv_w), ox = o - oyl * W;
            const int oy = r0 + oyl;
            float sum = bv;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy + ky - 1;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = ox + kx - 1;
                    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                        sum += tapl[(ky * 3 + kx) * plane + (iy - e0) * W + ix];
                }
            }
            out[((size_t)(b * cs + s) * H + oy) * W + ox] = sum;
        }
    }
}


// ---- head forward on the 16-bit matrix cores (bf16 or fp16 activations) ------------------------------------------------------
// Pass 1 of the kernel above is a GEMM: Tap[j][q] = sum_c w[j][c] a[q][c], j = (image channel, tap) <= 27 rows, 128-deep.  Here it IS
// one: v_mfma_f32_32x32x16_bf16 with the fp32 weights split into bf16 hi + lo parts (two MFMAs per k-step, the weight error drops to
// 2^-17; the activations are bf16 already), the weights resident in registers as A operands, a lane's B operand = the 16 bytes of
// its pixel's k-slice straight from HBM.  All image channels come out of ONE read of the activations (the VALU kernel re-read the band
// once per image channel), D[j][q] lands a lane's 16 tap sums for its own pixel, which go to the LDS planes; pass 2 is unchanged.
// One workgroup of 8 waves = one band of rows (+1 row above / below) of one image.
template <typename T>
__global__ __launch_bounds__(512) void head_fwd_mfma_kernel(const T* __restrict__ a, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ out, int cs, int H,
                                                           int W, int band, int nbands, unsigned a_bytes) {
    constexpr int C = 128;
    constexpr unsigned kBadOff = 0xFFFFFF00u;
    extern __shared__ __attribute__((aligned(16))) float tapl[];   // [9 cs][(band+2)*W]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / nbands, r0 = (blockIdx.x % nbands) * band;
    const int rows = min(band, H - r0);
    const int e0 = max(r0 - 1, 0), e1 = min(r0 + rows + 1, H);       // rows whose activations feed this band
    const int next = (e1 - e0) * W, plane = (band + 2) * W;
    const int K = 9 * cs;
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(a), 0, (int)a_bytes, 0x00020000);
    // A operands: row j = r, k = 16 ks + 8 h + e
    typedef typename Frag16<T>::type frag_t;
    frag_t whi[8], wlo[8];
    {
        const int sj = r / 9, tj = r - sj * 9;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float wv = r < K ? w[((size_t)sj * C + 16 * ks + 8 * h + e) * 9 + tj] : 0.f;
                const T hi = (T)wv;
                whi[ks][e] = hi;
                wlo[ks][e] = (T)(wv - (float)hi);
            }
    }
    const unsigned pix0 = ((unsigned)b * H + e0) * W;               // first pixel of the region
    const int nblk = (next + 31) >> 5;
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4s;
    auto fetch = [&](int blk, u32x4s (&bf)[8]) {
        const int q = blk * 32 + r;
        const unsigned off = (blk < nblk && q < next) ? ((pix0 + (unsigned)q) * C + 8u * h) * 2u : kBadOff;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) bf[ks] = __builtin_amdgcn_raw_buffer_load_b128(rsa, off, ks * 32, 0);
    };
    u32x4s bn[8];
    fetch(wave, bn);
    for (int blk = wave; blk < nblk; blk += 8) {
        u32x4s bc[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) bc[ks] = bn[ks];
        fetch(blk + 8, bn);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const frag_t bv = __builtin_bit_cast(frag_t, bc[ks]);
            acc = mfma_32x32x16<T>(whi[ks], bv, acc);
            acc = mfma_32x32x16<T>(wlo[ks], bv, acc);
        }
        const int q = blk * 32 + r;
        if (q < next) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int j = (e & 3) + 8 * (e >> 2) + 4 * h;
                if (j < K) tapl[j * plane + q] = acc[e];
            }
        }
    }
    __syncthreads();
    const int per = rows * W;
    for (int o = tid; o < cs * per; o += 512) {
        const int sc = o / per, rem = o - sc * per;
        const int oyl = rem / W, ox = rem - oyl * W, oy = r0 + oyl;
        float sum = bias[sc];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy + ky - 1;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox + kx - 1;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                    sum += tapl[(sc * 9 + ky * 3 + kx) * plane + (iy - e0) * W + ix];
            }
        }
        out[((size_t)(b * cs + sc) * H + oy) * W + ox] = sum;
    }
}

int small_ppb(int64_t npix) {          // pixels per workgroup of the streaming kernels
    return npix > 4096 * 256 ? 1024 : 512;
}

int small_blocks(int64_t npix) {
    const int ppb = small_ppb(npix);
    return (int)((npix + ppb - 1) / ppb);
}

// rows of the weight-gradient partials: one per workgroup of the matrix-core kernel, whose grid is the resident set (2 per CU);
// the VALU kernels accept any row count (their pixels per workgroup follow from it)
int wgrad_rows(int64_t npix) {
    const int nb = small_blocks(npix), resident = 2 * gmk_cu_limit();
    return nb < resident ? nb : resident;
}

bool small_shape_ok(int B, int cs, int H, int W, int C) {
    return B > 0 && cs >= 1 && cs <= kMaxSmall && H > 0 && W > 0 && (C == 128 || C == 256) && (int64_t)B * H * W < (1ll << 24) &&
           H * W >= 32;
}

}  // namespace

extern "C" int gmk_stem_wgrad_blocks(int64_t n_pixels) { return wgrad_rows(n_pixels); }
extern "C" int gmk_head_wgrad_blocks(int64_t n_pixels) { return wgrad_rows(n_pixels); }

template <typename T, bool FLIP>
static void launch_expand(const float* in, const float* w, const float* bias, T* out, int B, int cs, int H, int W, int C,
                          hipStream_t stream) {
    const int64_t npix = (int64_t)B * H * W;
    const int ppb = small_ppb(npix), nb = small_blocks(npix);
    const float inv_w = 1.0f / (float)W;
    // the matrix-core form (exact fp32, bit-identical sums) for 128-channel nets; GMK_DEV_VARIANT 41 keeps the VALU kernel (A/B)
    const size_t in_bytes = (size_t)npix * cs * 4, out_bytes = (size_t)npix * C * sizeof(T);
    if (C == 128 && out_bytes < 0xFFFFFF00ull && gmk_kernel_choice(3, "GMK_DEV_VARIANT") != 41) {
        const unsigned nblocks = (unsigned)((npix + 31) / 32);
        // persistent: exactly the resident set (2 workgroups per CU at <= 256 registers), so the per-wave operand setup runs once
        const unsigned resident = 2u * (unsigned)gmk_cu_limit();
        const unsigned grid = (nblocks + 3) / 4 < resident ? (nblocks + 3) / 4 : resident;
        if constexpr (sizeof(T) == 2) {          // 16-bit results: tiles with the image window in LDS and bf16 hi / lo operands (GMK_DEV_VARIANT 42 keeps the exact-fp32 chain: A/B)
            if (W <= 128 && cs <= 3 && gmk_kernel_choice(3, "GMK_DEV_VARIANT") != 42) {
                const int tps = (H * W + 127) / 128;
                const unsigned ntiles = (unsigned)B * (unsigned)tps;
                const unsigned grid16 = ntiles < resident ? ntiles : resident;
#define GMK_EXPAND_T16(CSN) expand3x3_tile16_kernel<T, CSN, FLIP><<<grid16, 256, 0, stream>>>(in, w, bias, out, H, W, C, tps, ntiles, (unsigned)in_bytes, (unsigned)out_bytes)
                if (cs == 1) GMK_EXPAND_T16(1);
                else if (cs == 2) GMK_EXPAND_T16(2);
                else GMK_EXPAND_T16(3);
#undef GMK_EXPAND_T16
                return;
            }
        }
#define GMK_EXPAND_M(CSN) expand3x3_mfma_kernel<T, CSN, FLIP><<<grid, 256, 0, stream>>>(in, w, bias, out, H, W, C, (unsigned)npix, nblocks, (unsigned)in_bytes, (unsigned)out_bytes)
        if (cs == 1) GMK_EXPAND_M(1);
        else if (cs == 2) GMK_EXPAND_M(2);
        else if (cs == 3) GMK_EXPAND_M(3);
        else GMK_EXPAND_M(4);
#undef GMK_EXPAND_M
        return;
    }
#define GMK_EXPAND(CSN, VW) expand3x3_kernel<T, CSN, VW, FLIP><<<nb, 256, 0, stream>>>(in, w, bias, out, H, W, C, inv_w, (unsigned)npix, ppb)
    if (cs == 1) GMK_EXPAND(1, 8);
    else if (cs == 2) GMK_EXPAND(2, 8);
    else if (cs == 3) GMK_EXPAND(3, 4);
    else GMK_EXPAND(4, 4);
#undef GMK_EXPAND
}

extern "C" int gmk_stem_fwd(const float* x, const float* w, const float* bias, void* y, int B, int cin, int H, int W, int C,
                            int dtype, void* stream) {
    GMK_REQUIRE(x && w && bias && y, "gmk_stem_fwd: null pointer");
    GMK_REQUIRE(small_shape_ok(B, cin, H, W, C), "gmk_stem_fwd: unsupported shape B=%d cin=%d %dx%d C=%d", B, cin, H, W, C);
    if (dtype == GMK_BF16) launch_expand<bf16_t, false>(x, w, bias, (bf16_t*)y, B, cin, H, W, C, gmk_stream(stream));
    else if (dtype == GMK_F16) launch_expand<f16_t, false>(x, w, bias, (f16_t*)y, B, cin, H, W, C, gmk_stream(stream));
    else if (dtype == GMK_F32) launch_expand<float, false>(x, w, bias, (float*)y, B, cin, H, W, C, gmk_stream(stream));
    else GMK_REQUIRE(false, "gmk_stem_fwd: bad dtype %d", dtype);
    return gmk_check_launch("gmk_stem_fwd");
}

extern "C" int gmk_head_dgrad(const float* dout, const float* w, void* da, int B, int cout, int H, int W, int C, int dtype,
                              void* stream) {
    GMK_REQUIRE(dout && w && da, "gmk_head_dgrad: null pointer");
    GMK_REQUIRE(small_shape_ok(B, cout, H, W, C), "gmk_head_dgrad: unsupported shape");
    if (dtype == GMK_BF16) launch_expand<bf16_t, true>(dout, w, nullptr, (bf16_t*)da, B, cout, H, W, C, gmk_stream(stream));
    else if (dtype == GMK_F32) launch_expand<float, true>(dout, w, nullptr, (float*)da, B, cout, H, W, C, gmk_stream(stream));
    else GMK_REQUIRE(false, "gmk_head_dgrad: bad dtype %d", dtype);
    return gmk_check_launch("gmk_head_dgrad");
}

template <typename T, bool FLIP>
static void launch_wgrad(const float* small, const T* big, float* part, int B, int cs, int H, int W, int C, hipStream_t stream) {
    const int64_t npix = (int64_t)B * H * W;
    const int nb = wgrad_rows(npix);
    int ppb = (int)((npix + nb - 1) / nb);
    ppb = (ppb + 31) & ~31;
    const float inv_w = 1.0f / (float)W;
    const size_t small_bytes = (size_t)npix * cs * 4, big_bytes = (size_t)npix * C * sizeof(T);
    if (C == 128 && cs <= 3 && W % 2 == 0 && big_bytes < 0xFFFFFF00ull && gmk_kernel_choice(3, "GMK_DEV_VARIANT") != 41) {
        const unsigned npairs = (unsigned)((npix + 1) / 2);
        unsigned ppw = (npairs + (unsigned)nb * 4 - 1) / ((unsigned)nb * 4);
        ppw = (ppw + 7) & ~7u;                                         // whole groups of 8 pairs
        if constexpr (sizeof(T) == 2) {          // 16-bit tensors: tiles through LDS, 16-bit matrix cores (GMK_DEV_VARIANT 42 keeps the fp32 chain: A/B)
            if (W <= 128 && small_bytes < 0xFFFFFF00ull && gmk_kernel_choice(3, "GMK_DEV_VARIANT") != 42) {
                const int tps = (H * W + 127) / 128;
                const unsigned ntiles = (unsigned)B * (unsigned)tps;
#define GMK_WGRAD_T16(CSN) wgrad3x3_tile16_kernel<T, CSN, FLIP><<<nb, 256, 0, stream>>>(small, big, part, H, W, C, tps, ntiles, (unsigned)small_bytes, (unsigned)big_bytes)
                if (cs == 1) GMK_WGRAD_T16(1);
                else if (cs == 2) GMK_WGRAD_T16(2);
                else GMK_WGRAD_T16(3);
#undef GMK_WGRAD_T16
                return;
            }
        }
#define GMK_WGRAD_M(CSN) wgrad3x3_mfma_kernel<T, CSN, FLIP><<<nb, 256, 0, stream>>>(small, big, part, H, W, C, (unsigned)npix, (unsigned)small_bytes, (unsigned)big_bytes, ppw)
        if (cs == 1) GMK_WGRAD_M(1);
        else if (cs == 2) GMK_WGRAD_M(2);
        else GMK_WGRAD_M(3);
#undef GMK_WGRAD_M
        return;
    }
    // VALU fallback.  Measured (B=1024, 28x28, 3 image channels): one launch per image channel with 8-channel vectors (the C-channel
    // tensor is read cs times, 204 us) beats all channels at once with 4-channel vectors (one read, but 27 reduction rounds per block: 279 us)
    for (int s0 = 0; s0 < cs; ++s0)
        wgrad3x3_kernel<T, 1, 8, FLIP><<<nb, 256, 0, stream>>>(small, big, part, H, W, C, inv_w, (unsigned)npix, ppb, cs, s0);
}

extern "C" int gmk_stem_wgrad(const float* x, const void* dy, float* dw_part, int B, int cin, int H, int W, int C, int dtype,
                              void* stream) {
    GMK_REQUIRE(x && dy && dw_part, "gmk_stem_wgrad: null pointer");
    GMK_REQUIRE(small_shape_ok(B, cin, H, W, C), "gmk_stem_wgrad: unsupported shape");
    if (dtype == GMK_BF16) launch_wgrad<bf16_t, false>(x, (const bf16_t*)dy, dw_part, B, cin, H, W, C, gmk_stream(stream));
    else if (dtype == GMK_F32) launch_wgrad<float, false>(x, (const float*)dy, dw_part, B, cin, H, W, C, gmk_stream(stream));
    else GMK_REQUIRE(false, "gmk_stem_wgrad: bad dtype %d", dtype);
    return gmk_check_launch("gmk_stem_wgrad");
}

extern "C" int gmk_head_wgrad(const float* dout, const void* a, float* dw_part, int B, int cout, int H, int W, int C,
                              int dtype, void* stream) {
    GMK_REQUIRE(dout && a && dw_part, "gmk_head_wgrad: null pointer");
    GMK_REQUIRE(small_shape_ok(B, cout, H, W, C), "gmk_head_wgrad: unsupported shape");
    if (dtype == GMK_BF16) launch_wgrad<bf16_t, true>(dout, (const bf16_t*)a, dw_part, B, cout, H, W, C, gmk_stream(stream));
    else if (dtype == GMK_F16) launch_wgrad<f16_t, true>(dout, (const f16_t*)a, dw_part, B, cout, H, W, C, gmk_stream(stream));
    else if (dtype == GMK_F32) launch_wgrad<float, true>(dout, (const float*)a, dw_part, B, cout, H, W, C, gmk_stream(stream));
    else GMK_REQUIRE(false, "gmk_head_wgrad: bad dtype %d", dtype);
    return gmk_check_launch("gmk_head_wgrad");
}

extern "C" int gmk_head_fwd(const void* a, const float* w, const float* bias, float* out, int B, int cout, int H, int W,
                            int C, int dtype, void* stream) {
    GMK_REQUIRE(a && w && bias && out, "gmk_head_fwd: null pointer");
    GMK_REQUIRE(small_shape_ok(B, cout, H, W, C), "gmk_head_fwd: unsupported shape");
    const size_t a_bytes = (size_t)B * H * W * C * 2;
    if (gmk_is16(dtype) && C == 128 && cout <= 3 && a_bytes < 0xFFFFFF00ull && gmk_kernel_choice(3, "GMK_DEV_VARIANT") != 41) {
        // matrix-core kernel: all image channels from one read; 9 * cout fp32 planes of (band + 2) rows in <= 124 KiB of LDS
        int mb = (int)(126976 / ((size_t)36 * cout * W)) - 2;
        if (mb >= 1) {
            if (mb >= H) mb = H;
            else {                       // equal bands
                const int n = (H + mb - 1) / mb;
                mb = (H + n - 1) / n;
            }
            const int nb2 = (H + mb - 1) / mb;
            const size_t lds2 = (size_t)9 * cout * (mb + 2) * W * 4;
            static const hipError_t attr = hipFuncSetAttribute((const void*)head_fwd_mfma_kernel<bf16_t>,
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            static const hipError_t attr16 = hipFuncSetAttribute((const void*)head_fwd_mfma_kernel<f16_t>,
                                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            (void)attr; (void)attr16;
            if (dtype == GMK_F16)
                head_fwd_mfma_kernel<f16_t><<<B * nb2, 512, lds2, gmk_stream(stream)>>>((const f16_t*)a, w, bias, out, cout, H, W, mb, nb2,
                                                                                       (unsigned)a_bytes);
            else
                head_fwd_mfma_kernel<bf16_t><<<B * nb2, 512, lds2, gmk_stream(stream)>>>((const bf16_t*)a, w, bias, out, cout, H, W, mb, nb2,
                                                                                        (unsigned)a_bytes);
            return gmk_check_launch("gmk_head_fwd");
        }
    }
    // band of rows per workgroup: 9 fp32 planes of (band + 2) rows within 48 KiB of LDS, at least 4 workgroups per CU's worth of bands
    int band = 49152 / (36 * W) - 2;
    GMK_REQUIRE(band >= 1, "gmk_head_fwd: image too wide (W=%d)", W);
    if (band > H) band = H;
    const int nbands = (H + band - 1) / band;
    const size_t lds = (size_t)9 * (band + 2) * W * 4;
    const float inv_w = 1.0f / (float)W;
    if (dtype == GMK_BF16)
        head_fwd_kernel<bf16_t><<<B * nbands, 256, lds, gmk_stream(stream)>>>((const bf16_t*)a, w, bias, out, cout, H, W, C, inv_w,
                                                                              band, nbands);
    else if (dtype == GMK_F16)
        head_fwd_kernel<f16_t><<<B * nbands, 256, lds, gmk_stream(stream)>>>((const f16_t*)a, w, bias, out, cout, H, W, C, inv_w,
                                                                             band, nbands);
    else if (dtype == GMK_F32)
        head_fwd_kernel<float><<<B * nbands, 256, lds, gmk_stream(stream)>>>((const float*)a, w, bias, out, cout, H, W, C, inv_w,
                                                                             band, nbands);
    else
        GMK_REQUIRE(false, "gmk_head_fwd: bad dtype %d", dtype);
    return gmk_check_launch("gmk_head_fwd");
}
