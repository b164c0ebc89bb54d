// Implicit-GEMM convolution on the MFMA units: forward, data gradient and weight gradient of every
// 128-channel-wide Conv2d of the U-Net (reference gms/diffusion/simple_unet.py:81,117,163,172,177 and the
// autograd backward of each).  NHWC activations, K-contiguous packed weights.
//
// Forward / data-gradient kernel (conv_igemm_kernel):
//   GEMM view  M = B*Ho*Wo output pixels, N = 128 output channels per workgroup, K = taps * (C0 + C1).
//   Workgroup = 256 threads = 4 waves, 128x128 output tile, each wave a 64x64 sub-tile = 2x2 MFMA 32x32 tiles.
//   K-step = 128 bytes of channels of one filter tap (64 bf16 / 32 fp32).  Per step every thread gathers
//   4x16 B of the A tile (source pixel of (output pixel, tap); zero outside the image) and 4x16 B of the B
//   tile (weights), register-staged one step ahead, then written to a 2-deep LDS ring with a 16-B-chunk XOR
//   swizzle chunk ^= (row>>1)&7 so that every ds_read_b128 lane group hits 16 distinct bank slots.
//   bf16: v_mfma_f32_32x32x16_bf16; fp32: 4 x v_mfma_f32_32x32x2_f32 per 16-B fragment (exact fp32 FMA chain).
//   The concatenated input of the up path (torch.cat([x, skip], 1), simple_unet.py:150), nearest x2 upsampling
//   (:120), stride 2 (:81) and the transposed form (dgrad of stride 2) are address arithmetic in the gather.
//   Epilogue: accumulators -> LDS (fp32 128x128) -> coalesced rows with bias + per-sample embedding
//   broadcast (simple_unet.py:183-184) + residual (:186) added in fp32.
//
// Weight-gradient kernel (conv_wgrad_kernel):
//   D[co][ci] = sum_pixels dY[px][co] * A_tap[px][ci]; both operands are pixel-major in memory while the MFMA
//   contracts over pixels, so tiles are staged [pixel][channel] and read back transposed: bf16 through
//   ds_read_b64_tr_b16 (window swizzle win ^= px&3, conflict-free), fp32 through plain ds_read_b32 (one k per
//   lane).  Split-K over pixel ranges into fp32 slabs + a deterministic reduce that also converts to the
//   reference's [Cout][Cin][k][k] layout.
#include <stdlib.h>

#include <type_traits>

#include "gmk_common.h"

namespace {

constexpr int kBM = 128, kBN = 128;

struct GatherParams {
    int hs, ws;          // source spatial size
    int ho, wo;          // output spatial size
    int ksize, pad;
    int mul, shift, mask, lim_h, lim_w;
    // LDS-DMA kernel: tap slot t gathers source pixel (base + ty[t], base + tx[t]) against weight tap tapw[t]; ntaps slots are live.
    // Plain convolutions: slot = tap (ty = t / 3, tx = t % 3).  The four output-parity phases of the stride-2 data gradient
    // (make_phase_gather) list only the taps that meet a non-zero of the zero-stuffed gradient: 1, 2, 2 and 4 of the 9.
    signed char ty[9], tx[9];
    unsigned char tapw[9];
    int ntaps;
    int oscale, ooy, oox, out_w, out_hw;      // oscale = 2: row m = (b, i, j) of the phase grid is written to pixel (2 i + ooy, 2 j + oox) of out
};

__host__ bool make_gather(int mode, int ksize, int hs, int ws, int ho, int wo, GatherParams* g) {
    g->hs = hs; g->ws = ws; g->ho = ho; g->wo = wo;
    g->ksize = ksize; g->pad = ksize == 3 ? 1 : 0;
    g->mul = 1; g->shift = 0; g->mask = 0; g->lim_h = hs; g->lim_w = ws;
    for (int t = 0; t < 9; ++t) { g->ty[t] = (signed char)(ksize == 3 ? t / 3 : 0); g->tx[t] = (signed char)(ksize == 3 ? t % 3 : 0); g->tapw[t] = (unsigned char)t; }
    g->ntaps = ksize * ksize;
    g->oscale = 1; g->ooy = 0; g->oox = 0; g->out_w = wo; g->out_hw = ho * wo;
    switch (mode) {
        case GMK_CONV_NORMAL:
            return ho == hs && wo == ws;
        case GMK_CONV_STRIDE2:
            g->mul = 2;
            return ksize == 3 && ho == (hs - 1) / 2 + 1 && wo == (ws - 1) / 2 + 1;
        case GMK_CONV_UPSAMPLE2:
            g->shift = 1; g->lim_h = 2 * hs; g->lim_w = 2 * ws;
            return ksize == 3 && ho == 2 * hs && wo == 2 * ws;
        case GMK_CONV_TRANSPOSED2:   // output = input of a stride-2 conv (ho x wo), source = its output gradient (hs x ws)
            g->shift = 1; g->mask = 1; g->lim_h = 2 * hs; g->lim_w = 2 * ws;
            return ksize == 3 && hs == (ho - 1) / 2 + 1 && ws == (wo - 1) / 2 + 1;
        default:
            return false;
    }
}

// Data gradient of the stride-2 3x3 convolution (GMK_CONV_TRANSPOSED2), output pixels of parity (a, b) only: dx[2i + a][2j + b] =
// sum over the taps (ky, kx) of the flipped kernel with (a + ky - 1), (b + kx - 1) even of dy[i + (ky == 2)][j + (kx == 2)] . Wd[ky][kx] -
// a 1-, 2- or 4-tap convolution ON THE GRADIENT'S OWN GRID (hs x ws) with a stride-2 scatter of its rows; the four phases together do
// 9 / 4 tap-images of MFMA work where the zero-stuffed form does 9.
__host__ void make_phase_gather(int a, int b, int hs, int ws, int ho, int wo, GatherParams* g) {
    g->hs = hs; g->ws = ws; g->ho = hs; g->wo = ws;       // the phase's rows run over the gradient's grid
    g->ksize = 3; g->pad = 0; g->mul = 1; g->shift = 0; g->mask = 0; g->lim_h = hs; g->lim_w = ws;
    int n = 0;
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {
            if (((a + ky - 1) & 1) || ((b + kx - 1) & 1)) continue;
            g->ty[n] = (signed char)(ky == 2); g->tx[n] = (signed char)(kx == 2); g->tapw[n] = (unsigned char)(ky * 3 + kx);
            ++n;
        }
    g->ntaps = n;
    for (int t = n; t < 9; ++t) { g->ty[t] = 0; g->tx[t] = 0; g->tapw[t] = 0; }
    g->oscale = 2; g->ooy = a; g->oox = b; g->out_w = wo; g->out_hw = ho * wo;
}

struct ConvParams {
    const void* src0; const void* src1;
    int c0, c1, ktot;
    GatherParams g;
    const void* w; int64_t w_tap_stride;   // elements
    int n0;
    const float* bias; const float* emb; int emb_stride;
    const void* residual; void* out; int out_cstride;
    void* out2;   // LDS-DMA kernel only: a second 128-channel output block (weight rows n0 + 128 ...) from the same pass over the source
    int M;   // B*ho*wo
    unsigned nb0, nb1, nbw, nbo;   // byte sizes of src0 / src1 / w / out for the buffer descriptors of the LDS-DMA kernel
};

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

// ---- fp32 storage on the bf16 matrix cores (round 6): x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (both round-to-nearest: |x - hi - lo| <=
// 2^-17 |x|), a product a . b as hi hi + hi lo + lo hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (the dropped lo lo term is <= 2^-16
// of the product).  Three bf16 MFMAs of 16 k each replace eight exact-fp32 MFMAs of 2 k: 96 instead of 512 matrix-core cycles per 16 k and
// 32 x 32 tile: 1e-5-class results at a fifth of the exact chain's matrix time.  Opt-in (GMK_FP32_SPLIT=1 / gmk_set_fp32_exact(0)): the default fp32
// mode keeps the exact chains (v_mfma_f32_32x32x2_f32), which the ill-conditioned closed-form reference vectors need for the 1e-3 bar.
__device__ __forceinline__ void split_bf16(const f32x4 a, const f32x4 b, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bf16_t h0 = (bf16_t)a[e], h1 = (bf16_t)b[e];
        hi[e] = h0; hi[4 + e] = h1;
        lo[e] = (bf16_t)(a[e] - (float)h0); lo[4 + e] = (bf16_t)(b[e] - (float)h1);
    }
}
__device__ __forceinline__ f32x16 mfma_split(const bf16x8 ahi, const bf16x8 alo, const bf16x8 bhi, const bf16x8 blo, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi, c, 0, 0, 0);      // the small terms first
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi, c, 0, 0, 0);
}

template <typename T, bool kSplit = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvParams p) {
    static_assert(!kSplit || sizeof(T) == 4, "the hi / lo split is the fp32 mode's");
    constexpr int ES = sizeof(T);
    fp16_saturating_stores<T>();
    constexpr int KCH = 128 / ES;          // elements per 128-byte K chunk
    constexpr int EPC = 16 / ES;           // elements per 16-byte staging chunk
    __shared__ __attribute__((aligned(16))) char smem[65536];
    char* As = smem;                       // [2][128 rows][128 B]
    char* Bs = smem + 32768;               // [2][128 rows][128 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * kBM;
    const int nblk = blockIdx.y * kBN;
    const GatherParams g = p.g;
    const int hw_o = g.ho * g.wo;

    // ---- per-thread staging rows: 4 rows (tid>>3) + 32*i, 16-byte chunk (tid&7)
    const int srow = tid >> 3, sc = tid & 7;
    int rb_[4], ry_[4], rx_[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + srow + 32 * i;
        if (m < p.M) {
            const int b = m / hw_o;
            const int rem = m - b * hw_o;
            const int oy = rem / g.wo;
            rb_[i] = b; ry_[i] = oy * g.mul - g.pad; rx_[i] = (rem - oy * g.wo) * g.mul - g.pad;
        } else {
            rb_[i] = -1; ry_[i] = 0; rx_[i] = 0;
        }
    }
    int lds_w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = srow + 32 * i;
        lds_w[i] = row * 128 + ((sc ^ ((row >> 1) & 7)) << 4);
    }
    const T* wrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        wrow[i] = (const T*)p.w + (int64_t)(p.n0 + nblk + srow + 32 * i) * p.ktot + sc * EPC;

    const int kpt = p.ktot / KCH;            // K chunks per tap
    const int nk = kpt * g.ksize * g.ksize;

    u32x4 ra[4], rbv[4];
    int tap = 0, ky = 0, kx = 0, kc = 0;     // position of the NEXT step to load

    auto load_step = [&]() {
        const int kelem = kc * KCH;
        const T* src; int cs, koff;
        if (kelem < p.c0) { src = (const T*)p.src0; cs = p.c0; koff = kelem; }
        else { src = (const T*)p.src1; cs = p.c1; koff = kelem - p.c0; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ty = ry_[i] + ky, tx = rx_[i] + kx;
            const bool ok = rb_[i] >= 0 && ty >= 0 && ty < g.lim_h && tx >= 0 && tx < g.lim_w && !((ty | tx) & g.mask);
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) {
                const int sy = ty >> g.shift, sx = tx >> g.shift;
                const int64_t pix = ((int64_t)rb_[i] * g.hs + sy) * g.ws + sx;
                v = *reinterpret_cast<const u32x4*>(src + pix * cs + koff + sc * EPC);
            }
            ra[i] = v;
        }
        const int64_t woff = (int64_t)tap * p.w_tap_stride + kelem;
#pragma unroll
        for (int i = 0; i < 4; ++i) rbv[i] = *reinterpret_cast<const u32x4*>(wrow[i] + woff);
        // advance
        if (++kc == kpt) {
            kc = 0; ++tap;
            if (++kx == g.ksize) { kx = 0; ++ky; }
        }
    };
    auto write_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(As + buf * 16384 + lds_w[i]) = ra[i];
            *reinterpret_cast<u32x4*>(Bs + buf * 16384 + lds_w[i]) = rbv[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int swz = (r >> 1) & 7;
    const int a_off = (wm * 64 + r) * 128;
    const int b_off = (wn * 64 + r) * 128;

    load_step();
    write_step(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nk) load_step();
        const char* Ab = As + buf * 16384;
        const char* Bb = Bs + buf * 16384;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            const int coff = ((kg * 2 + h) ^ swz) << 4;
            if constexpr (ES == 2) {
                typedef typename Frag16<T>::type frag_t;
                frag_t a[2], b[2];
                a[0] = *reinterpret_cast<const frag_t*>(Ab + a_off + coff);
                a[1] = *reinterpret_cast<const frag_t*>(Ab + a_off + 4096 + coff);
                b[0] = *reinterpret_cast<const frag_t*>(Bb + b_off + coff);
                b[1] = *reinterpret_cast<const frag_t*>(Bb + b_off + 4096 + coff);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = mfma_32x32x16<T>(a[i], b[j], acc[i][j]);
            } else if constexpr (kSplit) {
                // two 16-k sub-steps per pass (kg = 0, 2 of the four); lane (r, h) takes the chunks 4 (kg / 2) + 2 h, + 1 of its row: 8 consecutive k
                if (kg & 1) continue;
                const int c0 = (((kg * 2) + 2 * h) ^ swz) << 4, c1 = (((kg * 2) + 2 * h + 1) ^ swz) << 4;
                bf16x8 ahi[2], alo[2], bhi[2], blo[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    split_bf16(*reinterpret_cast<const f32x4*>(Ab + a_off + i * 4096 + c0), *reinterpret_cast<const f32x4*>(Ab + a_off + i * 4096 + c1), ahi[i], alo[i]);
                    bhi[i] = *reinterpret_cast<const bf16x8*>(Bb + b_off + i * 4096 + c0);          // the weight pack is stored split (store_pack)
                    blo[i] = *reinterpret_cast<const bf16x8*>(Bb + b_off + i * 4096 + c1);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = mfma_split(ahi[i], alo[i], bhi[j], blo[j], acc[i][j]);
            } else {
                f32x4 a[2], b[2];
                a[0] = *reinterpret_cast<const f32x4*>(Ab + a_off + coff);
                a[1] = *reinterpret_cast<const f32x4*>(Ab + a_off + 4096 + coff);
                b[0] = *reinterpret_cast<const f32x4*>(Bb + b_off + coff);
                b[1] = *reinterpret_cast<const f32x4*>(Bb + b_off + 4096 + coff);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
            }
        }
        if (ks + 1 < nk) write_step(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: accumulators -> LDS fp32 [128][128] -> coalesced rows
    float* cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ml = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int nl = wn * 64 + j * 32 + r;
                cs[ml * 128 + nl] = acc[i][j][e];
            }
    __syncthreads();
    const int c4 = (tid & 31) * 4;
    const int och = nblk + c4;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) load4(p.bias + och, bv);
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int row = it * 8 + (tid >> 5);
        const int m = m0 + row;
        if (m >= p.M) continue;
        float v[4];
        load4(cs + row * 128 + c4, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bv[e];
        if (p.emb) {
            float t[4];
            load4(p.emb + (int64_t)(m / hw_o) * p.emb_stride + och, t);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += t[e];
        }
        const int64_t o = (int64_t)m * p.out_cstride + och;
        if (p.residual) {
            float t[4];
            load4((const T*)p.residual + o, t);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += t[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = sat16<T>(v[e]);
        store4((T*)p.out + o, v);
    }
}

// ------------------------------------------------------------------------------------------------------
// LDS-DMA variant for large problems — the kernel the step time is dominated by.
//   * 256(pixels) x 128(channels) output tile per workgroup, 8 waves (4 along pixels x 2 along channels, 64x64 each).
//   * 3-slot LDS ring of {pixel tile 256x128 B, weight tile 128x128 B} filled by `buffer_load_dwordx4 ... lds`
//     (no VGPR staging, no ds_write); two K-steps in flight behind a counted `s_waitcnt vmcnt`, one raw
//     `s_barrier` per K-step.  Zero padding / masked taps come from the buffer descriptor's range check (an
//     out-of-range offset feeds zeros), so the gather has no branches.  One DMA instruction writes 1 KiB
//     lane-linearly (8 rows x 128 B), so the bank swizzle is applied to each lane's SOURCE chunk:
//     physical chunk (lane&7) holds logical chunk (lane&7) ^ ((row>>1)&7); fragments are read with the same XOR.
//   * the source pixel of every (tile row, tap) is resolved once per tile (36 registers); the K loop is unrolled
//     over the taps so those registers are indexed statically, and a DMA issue is one 24-bit multiply-add.
//   * operands are fed to the MFMA as D[channel][pixel] = W . P^T, so a lane ends up owning 4 consecutive
//     channels of one pixel: after one v_permlane32_swap per dword every lane stores 16 contiguous bytes straight
//     from its accumulators (bias / embedding / residual added in fp32 before the rounding) — no LDS round trip,
//     no barrier in the epilogue.
//   * persistent workgroups (one per CU) walk the tiles; the first two K-steps of the next tile are issued
//     before the current tile's epilogue, so the epilogue runs under the next tile's DMA latency.
// vmcnt bookkeeping (CDNA4 counts loads, stores and LDS-DMA together, in issue order): every K-step is 6 DMA
// instructions per wave and the epilogue issues exactly kEpiStores buffer stores per wave (masked lanes use an
// out-of-range offset so the instruction is always issued), hence the first two K-steps after an epilogue wait with
// vmcnt(6 + kEpiStores) and every other one with vmcnt(6).
// ------------------------------------------------------------------------------------------------------
template <typename T, bool kSplit = false>
__global__ __launch_bounds__(512, 2) void conv_igemm_dma_kernel(const ConvParams p) {
    static_assert(!kSplit || sizeof(T) == 4, "the hi / lo split is the fp32 mode's");
    constexpr int ES = sizeof(T);
    fp16_saturating_stores<T>();
    constexpr int KCH = 128 / ES;
    constexpr int NS = 3, A_BYTES = 32768, STAGE = 49152;
    constexpr int kEpiStores = ES == 2 ? 8 : 16;
    static_assert(6 + kEpiStores == (ES == 2 ? 14 : 22), "wait_step's vmcnt literals");
    __shared__ __attribute__((aligned(16))) char smem[NS * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int nblk = blockIdx.y * kBN;
    const GatherParams g = p.g;
    const int hw_o = g.ho * g.wo;
    const int ntiles = (p.M + 255) >> 8;

    constexpr unsigned kBadPix = 0x00FFFFFFu;   // kBadPix * bytes-per-pixel lies beyond any buffer (host-checked)
    constexpr unsigned kBadOff = 0xFFFFFF00u;
    const int lrow = lane >> 3, lch = lane & 7;
    unsigned pixi[4][9];
    unsigned a_ch[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a_ch[i] = (unsigned)((lch ^ (((32 * wave + 8 * i + lrow) >> 1) & 7)) << 4);
    unsigned w_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 16 * wave + 8 * i + lrow;
        w_off[i] = (unsigned)(p.n0 + nblk + row) * (unsigned)p.ktot * ES + (unsigned)((lch ^ ((row >> 1) & 7)) << 4);
    }
    auto resolve_tile = [&](int tile) {      // fills pixi for the rows this lane stages
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = tile * 256 + 32 * wave + 8 * i + lrow;
            const bool live = m < p.M;
            const int mm = live ? m : 0;
            const int b = mm / hw_o;
            const int rem = mm - b * hw_o;
            const int oy = rem / g.wo;
            const int by = oy * g.mul - g.pad, bx = (rem - oy * g.wo) * g.mul - g.pad;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ty = by + g.ty[t], tx = bx + g.tx[t];
                const bool ok = live && ty >= 0 && ty < g.lim_h && tx >= 0 && tx < g.lim_w && !((ty | tx) & g.mask);
                pixi[i][t] = ok ? (unsigned)((b * g.hs + (ty >> g.shift)) * g.ws + (tx >> g.shift)) : kBadPix;
            }
        }
    };

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src0), 0, (int)p.nb0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.c1 ? p.src1 : p.src0), 0, (int)(p.c1 ? p.nb1 : p.nb0), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.nbw, 0x00020000);
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.nbo, 0x00020000);
    const __amdgpu_buffer_rsrc_t rso2 = __builtin_amdgcn_make_buffer_rsrc(p.out2 ? p.out2 : p.out, 0, (int)p.nbo, 0x00020000);
    const int NH = p.out2 ? 2 : 1;          // output blocks per pixel tile (virtual tile = tile * NH + block)
    const __amdgpu_buffer_rsrc_t rsr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.residual ? p.residual : p.out), 0, (int)p.nbo, 0x00020000);

    const int kpt = p.ktot / KCH;            // K chunks per tap (>= 2)
    const int ntaps = g.ntaps;               // 1 or 9 (a 1x1 conv uses tap slot 0: pad 0, offsets (0,0)); 1 / 2 / 4 for a stride-2 dgrad phase

    // DMA of K-step (tap, kc) into ring slot `stage`; `tap` is a compile-time constant at every call site
    auto issue = [&](int stage, const unsigned (&pix)[4], int tap, int kc, unsigned wblock = 0u) {
        const int kelem = kc * KCH;
        const bool second = kelem >= p.c0;
        const unsigned cs_b = (unsigned)(second ? p.c1 : p.c0) * ES;         // bytes per source pixel
        const unsigned koff_b = (unsigned)(second ? kelem - p.c0 : kelem) * ES;
        GMK_LDS char* lds_a = (GMK_LDS char*)(smem + stage * STAGE + wave * 4096);
        if (second) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (GMK_LDS void*)(lds_a + i * 1024), 16,
                                                         __umul24(pix[i], cs_b) + koff_b + a_ch[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (GMK_LDS void*)(lds_a + i * 1024), 16,
                                                         __umul24(pix[i], cs_b) + koff_b + a_ch[i], 0, 0, 0);
        }
        GMK_LDS char* lds_b = (GMK_LDS char*)(smem + stage * STAGE + A_BYTES + wave * 2048);
        const unsigned wk = ((unsigned)g.tapw[tap] * (unsigned)p.w_tap_stride + (unsigned)kelem) * ES + wblock;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (GMK_LDS void*)(lds_b + i * 1024), 16, w_off[i] + wk, 0, 0, 0);
    };

    f32x16 acc[2][2];     // [j: channel tile][i: pixel tile]

    const int swz = (r >> 1) & 7;
    const int a_off = (wm * 64 + r) * 128;
    const int b_off = A_BYTES + (wn * 64 + r) * 128;

    auto compute = [&](int st) {
        const char* Sb = smem + st * STAGE;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            const int coff = ((kg * 2 + h) ^ swz) << 4;
            if constexpr (ES == 2) {
                typedef typename Frag16<T>::type frag_t;
                frag_t px[2], wt[2];
                px[0] = *reinterpret_cast<const frag_t*>(Sb + a_off + coff);
                px[1] = *reinterpret_cast<const frag_t*>(Sb + a_off + 4096 + coff);
                wt[0] = *reinterpret_cast<const frag_t*>(Sb + b_off + coff);
                wt[1] = *reinterpret_cast<const frag_t*>(Sb + b_off + 4096 + coff);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[j][i] = mfma_32x32x16<T>(wt[j], px[i], acc[j][i]);
            } else if constexpr (kSplit) {
                // (see split_bf16) two 16-k sub-steps per K-step; lane (r, h) takes the chunks 4 (kg / 2) + 2 h, + 1 of its row: 8 consecutive k
                if (kg & 1) continue;
                const int c0 = (((kg * 2) + 2 * h) ^ swz) << 4, c1 = (((kg * 2) + 2 * h + 1) ^ swz) << 4;
                bf16x8 phi[2], plo[2], whi[2], wlo[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    split_bf16(*reinterpret_cast<const f32x4*>(Sb + a_off + i * 4096 + c0), *reinterpret_cast<const f32x4*>(Sb + a_off + i * 4096 + c1), phi[i], plo[i]);
                    whi[i] = *reinterpret_cast<const bf16x8*>(Sb + b_off + i * 4096 + c0);          // the weight pack is stored split (store_pack)
                    wlo[i] = *reinterpret_cast<const bf16x8*>(Sb + b_off + i * 4096 + c1);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[j][i] = mfma_split(whi[j], wlo[j], phi[i], plo[i], acc[j][i]);
            } else {
                f32x4 px[2], wt[2];
                px[0] = *reinterpret_cast<const f32x4*>(Sb + a_off + coff);
                px[1] = *reinterpret_cast<const f32x4*>(Sb + a_off + 4096 + coff);
                wt[0] = *reinterpret_cast<const f32x4*>(Sb + b_off + coff);
                wt[1] = *reinterpret_cast<const f32x4*>(Sb + b_off + 4096 + coff);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[j][s], px[i][s], acc[j][i], 0, 0, 0);
            }
        }
    };

    // per-lane bias for its 16 channels of each channel tile: channel = nblk + wn*64 + j*32 + (e&3) + 8*(e>>2) + 4*h
    float bias_r[2][16];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e)
            bias_r[j][e] = p.bias ? p.bias[nblk + wn * 64 + j * 32 + (e & 3) + 8 * (e >> 2) + 4 * h] : 0.f;

    // ---- epilogue straight from the accumulators
    auto epilogue = [&](int tile, int block = 0) {
        const __amdgpu_buffer_rsrc_t rso_b = block ? rso2 : rso;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = tile * 256 + wm * 64 + i * 32 + r;
            const bool live = m < p.M;
            int orow = m;
            if (g.oscale == 2) {       // stride-2 scatter of a data-gradient phase
                const int mm = live ? m : 0;
                const int b = mm / hw_o, rem = mm - b * hw_o, oy = rem / g.wo, ox = rem - oy * g.wo;
                orow = b * g.out_hw + (2 * oy + g.ooy) * g.out_w + 2 * ox + g.oox;
            }
            const unsigned row_b = (unsigned)orow * (unsigned)p.out_cstride * ES;
            const float* embp = p.emb ? p.emb + (int64_t)((live ? m : 0) / hw_o) * p.emb_stride : nullptr;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cb = nblk + wn * 64 + j * 32;       // first channel of this 32-channel tile
                float v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = acc[j][i][e] + bias_r[j][e];
                if (embp) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        float t[4];
                        load4(embp + cb + 8 * q4 + 4 * h, t);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * q4 + e] += t[e];
                    }
                }
                if (p.residual) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const unsigned off = live ? row_b + (unsigned)(cb + 8 * q4 + 4 * h) * ES : kBadOff;
                        float t[4];
                        if constexpr (ES == 2) {
                            const auto raw = __builtin_amdgcn_raw_buffer_load_b64(rsr, off, 0, 0);
                            const auto rb = __builtin_bit_cast(typename Frag16<T>::half_type, raw);
#pragma unroll
                            for (int e = 0; e < 4; ++e) t[e] = (float)rb[e];
                        } else {
                            const auto raw = __builtin_amdgcn_raw_buffer_load_b128(rsr, off, 0, 0);
                            const f32x4 rf = __builtin_bit_cast(f32x4, raw);
#pragma unroll
                            for (int e = 0; e < 4; ++e) t[e] = rf[e];
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * q4 + e] += t[e];
                    }
                }
                if constexpr (ES == 2) {
                    // pack to 16 bits: group q4 = channels cb + 8*q4 + 4*h + {0..3}  ->  2 dwords
                    unsigned pk[4][2];
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        pk[q4][0] = pack_pair<T>(sat16<T>(v[4 * q4]), sat16<T>(v[4 * q4 + 1]));
                        pk[q4][1] = pack_pair<T>(sat16<T>(v[4 * q4 + 2]), sat16<T>(v[4 * q4 + 3]));
                    }
                    // exchange between lane l (h=0) and l+32 (h=1): afterwards h=0 holds channels cb+8q..cb+8q+7 of
                    // group pair (q, q+1) and h=1 holds cb+8(q+1)..cb+8(q+1)+7: 16 contiguous bytes per lane
#pragma unroll
                    for (int q4 = 0; q4 < 4; q4 += 2) {
                        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[q4][0], pk[q4 + 1][0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[q4][1], pk[q4 + 1][1], false, false);
                        u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                        const unsigned off = live ? row_b + (unsigned)(cb + 8 * (q4 + h)) * ES : kBadOff;
                        __builtin_amdgcn_raw_buffer_store_b128(o, rso_b, off, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        f32x4 t = {v[4 * q4], v[4 * q4 + 1], v[4 * q4 + 2], v[4 * q4 + 3]};
                        const unsigned off = live ? row_b + (unsigned)(cb + 8 * q4 + 4 * h) * ES : kBadOff;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), rso_b, off, 0, 0);
                    }
                }
            }
        }
    };

    int st = 0;          // ring slot of the K-step computed next
    int sq = 2;          // ring slot the next issue goes to
    int fresh = 0;       // K-steps still to run with the epilogue's stores younger than their data
    // virtual tiles: (pixel tile, output block); a workgroup runs the NH blocks of its tile back to back, so the second block's
    // source DMA hits the lines the first one just pulled through this XCD's L2
    const unsigned wblock_b = (unsigned)kBN * (unsigned)p.ktot * ES;
    const int nvt = ntiles * NH;
    int vt = blockIdx.x * NH;
    if (vt < nvt) {
        resolve_tile(vt / NH);
        unsigned pix0[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pix0[i] = pixi[i][0];
        issue(0, pix0, 0, 0);
        issue(1, pix0, 0, 1);
    }
    auto wait_step = [&]() {       // the oldest K-step in flight has landed for this wave, then for everyone
        if (fresh > 0) {
            --fresh;
            if constexpr (ES == 2) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    };
    while (vt < nvt) {
        const int tile = vt / NH, block = vt - tile * NH;
        const unsigned wb = block ? wblock_b : 0u;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap < ntaps) {
                unsigned pix[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) pix[i] = pixi[i][tap];
                for (int kc = (tap == 0 ? 2 : 0); kc < kpt; ++kc) {
                    wait_step();
                    issue(sq, pix, tap, kc, wb);
                    compute(st);
                    st = st == 2 ? 0 : st + 1;
                    sq = sq == 2 ? 0 : sq + 1;
                }
            }
        }
        // the last two K-steps of this virtual tile run while the first two of the next one are issued
        const int nvt_next = block + 1 < NH ? vt + 1 : (tile + (int)gridDim.x) * NH;
        const bool more = nvt_next < nvt;
        const int ntile = nvt_next / NH;
        const unsigned nwb = (nvt_next - ntile * NH) ? wblock_b : 0u;
        if (more && ntile != tile) resolve_tile(ntile);
        unsigned pix0[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pix0[i] = pixi[i][0];
        wait_step();
        if (more) issue(sq, pix0, 0, 0, nwb);
        compute(st);
        st = st == 2 ? 0 : st + 1;
        sq = sq == 2 ? 0 : sq + 1;
        if (more) {
            wait_step();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (more) issue(sq, pix0, 0, 1, nwb);
        compute(st);
        st = st == 2 ? 0 : st + 1;
        sq = sq == 2 ? 0 : sq + 1;
        asm volatile("" ::: "memory");
        epilogue(tile, block);
        asm volatile("" ::: "memory");
        fresh = 2;
        vt = nvt_next;
    }
}

// ------------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------------
struct WgradParams {
    const void* dy; int dy_cstride;
    const void* src0; const void* src1;
    int c0, c1, ktot, cout;
    GatherParams g;
    float* slab;        // [nsplit][taps][cout][ktot]
    int M, chunk;       // pixels, pixels per split (multiple of 64)
    int ns, taps;       // splits, filter taps (the grid is 1-D: see the block decode)
    int order;          // 1: the (split, tap, block) order of rounds 1 - 3 (GMK_DEV_VARIANT=61, A/B)
};

// T: type of dy (and of the MFMA operands); TX: storage type of the activation operand.  TX = f16_t with T = bf16_t is the train
// step's case (fp16 forward activations, bf16 gradients): the gathered activation vectors are re-rounded to bf16 in the staging
// registers (the value set of bf16 x bf16 products the MFMA needs; 3 VALU per dword, next to 16-byte loads), nothing else changes.
template <typename T, typename TX = T, bool kSplit = false>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradParams p) {
    static_assert(sizeof(T) == sizeof(TX), "mixed operand widths");
    static_assert(!kSplit || sizeof(T) == 4, "the hi / lo split is the fp32 mode's");
    constexpr int ES = sizeof(T);
    constexpr int KP = ES == 2 ? 64 : 32;  // pixels per K-step (a 16 KB tile of 128 channels)
    constexpr int ROWB = 128 * ES;         // bytes per pixel row of a tile (128 channels)
    constexpr int CPR = ROWB / 16;         // 16-byte chunks per row (16 / 32)
    constexpr int EPC = 16 / ES;
    __shared__ __attribute__((aligned(16))) char smem[65536];
    char* Ys = smem;                       // [2][KP px][128 ch]  dY tile
    char* Xs = smem + 32768;               // [2][KP px][128 ch]  gathered activation tile

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const GatherParams g = p.g;
    const int hw_o = g.ho * g.wo;
    // 1-D grid, decoded so that the workgroups of ONE pixel range - its taps, its ci blocks, its co blocks: they all read the same dY rows,
    // the taps overlapping X rows too - sit 8 block ids apart: on one XCD (block id mod 8) and next to each other in its dispatch order.  Then
    // the first of them pulls a line into that XCD's L2 and the others find it there, and because a workgroup that hits runs ahead until it
    // misses, they fall into step.  (As a (split, tap, block) grid they were `nsplit` ids apart: same XCD, but out of step by more than the
    // L2 holds - the K = 256 skip convolution fetched dY twice, the stride-2 im2col form most lines several times: FETCH_SIZE of
    // round 4's im2col weight-gradient mix (docs/EXPERIMENTS.md 7b.14) 1,997 -> 1,348 MB per launch, 3 - 8 % of the time at 64 x 64 / 28 x 28 and of the stride-2 launches.)
    // (One workgroup for BOTH input blocks of the skip convolution, dY staged once - 96 KiB of LDS, one workgroup per CU - was built and
    // measured 3 - 5 % slower than two workgroups that meet in L2.)
    const int ncib = p.ktot / 128;
    const int ntile = p.taps * ncib * (p.cout / 128);
    const int bj = blockIdx.x >> 3;
    const int split = p.order ? (int)(blockIdx.x % p.ns) : (bj / ntile) * 8 + (blockIdx.x & 7);
    const int tl = p.order ? (int)(blockIdx.x / p.ns) : bj % ntile;
    if (split >= p.ns || tl >= ntile) return;
    const int tap = tl % p.taps, bz = tl / p.taps;
    const int ky = tap / g.ksize, kx = tap - ky * g.ksize;
    const int cib = bz % ncib, cob = bz / ncib;
    const int kelem0 = cib * 128;
    const TX* src; int cs, ci_off;
    if (kelem0 < p.c0) { src = (const TX*)p.src0; cs = p.c0; ci_off = kelem0; }
    else { src = (const TX*)p.src1; cs = p.c1; ci_off = kelem0 - p.c0; }
    const T* dy = (const T*)p.dy + cob * 128;

    const int pix_begin = split * p.chunk;
    const int pix_end = min(pix_begin + p.chunk, p.M);
    const int nk = pix_end > pix_begin ? (pix_end - pix_begin + KP - 1) / KP : 0;

    // staging: 4 rows per thread
    const int srow = tid / CPR, sc = tid % CPR;
    constexpr int RSTEP = 256 / CPR;       // 16 (bf16) / 8 (fp32)
    int lds_w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = srow + RSTEP * i;
        if constexpr (ES == 2) lds_w[i] = row * 256 + ((((sc >> 2) ^ (row & 3)) << 2 | (sc & 3)) << 4);
        else if constexpr (kSplit) {
            // split form: the tile is staged as TWO bf16 tiles (hi at +0, lo at +8 KiB) in the 16-bit path's layout - 256-byte pixel rows, window
            // swizzle - so the K loop is that path's transposed reads with no arithmetic; this thread's 4 channels are half of a 16-byte chunk
            const int c16 = sc >> 1;
            lds_w[i] = row * 256 + ((((c16 >> 2) ^ (row & 3)) << 2 | (c16 & 3)) << 4) + (sc & 1) * 8;
        } else lds_w[i] = row * 512 + sc * 16;
    }
    // two K-steps of operands in flight per thread (register sets ks & 1; one step ahead: 2 - 3 % slower on the 1x1 launches)
    u32x4 ryy[2][4], rxx[2][4];
    auto load_step = [&](int ks, u32x4 (&ry)[4], u32x4 (&rx)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = pix_begin + ks * KP + srow + RSTEP * i;
            u32x4 vy = {0u, 0u, 0u, 0u}, vx = {0u, 0u, 0u, 0u};
            if (m < pix_end) {
                vy = *reinterpret_cast<const u32x4*>(dy + (int64_t)m * p.dy_cstride + sc * EPC);
                const int b = m / hw_o;
                const int rem = m - b * hw_o;
                const int oy = rem / g.wo;
                const int ox = rem - oy * g.wo;
                const int ty = oy * g.mul - g.pad + ky, tx = ox * g.mul - g.pad + kx;
                if (ty >= 0 && ty < g.lim_h && tx >= 0 && tx < g.lim_w && !((ty | tx) & g.mask)) {
                    const int64_t pix = ((int64_t)b * g.hs + (ty >> g.shift)) * g.ws + (tx >> g.shift);
                    vx = *reinterpret_cast<const u32x4*>(src + pix * cs + ci_off + sc * EPC);
                }
            }
            ry[i] = vy; rx[i] = vx;
        }
    };
    auto write_step = [&](int buf, u32x4 (&ry)[4], u32x4 (&rx)[4]) {
        if constexpr (kSplit) {
            // x = hi + lo, hi = bf16(x), lo = bf16(x - hi): converted ONCE per element here (every element is read by two waves in the K loop)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 vy = __builtin_bit_cast(f32x4, ry[i]), vx = __builtin_bit_cast(f32x4, rx[i]);
                bf16x4 yh, yl, xh, xl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    yh[e] = (bf16_t)vy[e]; yl[e] = (bf16_t)(vy[e] - (float)yh[e]);
                    xh[e] = (bf16_t)vx[e]; xl[e] = (bf16_t)(vx[e] - (float)xh[e]);
                }
                *reinterpret_cast<bf16x4*>(Ys + buf * 16384 + lds_w[i]) = yh;
                *reinterpret_cast<bf16x4*>(Ys + buf * 16384 + 8192 + lds_w[i]) = yl;
                *reinterpret_cast<bf16x4*>(Xs + buf * 16384 + lds_w[i]) = xh;
                *reinterpret_cast<bf16x4*>(Xs + buf * 16384 + 8192 + lds_w[i]) = xl;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(Ys + buf * 16384 + lds_w[i]) = ry[i];
            u32x4 vx = rx[i];
            if constexpr (!__is_same(T, TX)) {      // re-round here, behind the MFMAs the loads were issued under - not where they are issued
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    float lo, hi;
                    unpack_pair<TX>(vx[d], lo, hi);
                    vx[d] = pack_pair<T>(lo, hi);
                }
            }
            *reinterpret_cast<u32x4*>(Xs + buf * 16384 + lds_w[i]) = vx;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // transposed-read lane geometry (bf16): 16-lane group gg, lane-in-group 4q+pp
    const int gg = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
    const int hh = gg >> 1, cblk = gg & 1;
    const int r = lane & 31, h = lane >> 5;

    if (nk > 0) {
        load_step(0, ryy[0], rxx[0]);
        if (nk > 1) load_step(1, ryy[1], rxx[1]);
        write_step(0, ryy[0], rxx[0]);
    }
    __syncthreads();
    auto k_step = [&](int ks, auto par) {
        constexpr int PAR = decltype(par)::value;          // ks & 1: LDS buffer of this step, register set of step ks + 2
        const int buf = PAR;
        if (ks + 2 < nk) load_step(ks + 2, ryy[PAR], rxx[PAR]);
        const char* Yb = Ys + buf * 16384;
        const char* Xb = Xs + buf * 16384;
        if constexpr (ES == 2) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                bf16x8 a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    s16x4 lo, hi;
                    const int pxl = kk * 16 + 8 * hh + q;
                    const int wa = (((wm * 2 + i) ^ q) << 6) + 32 * cblk + 8 * pp;
                    const int wb = (((wn * 2 + i) ^ q) << 6) + 32 * cblk + 8 * pp;
                    lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)(Yb + pxl * 256 + wa));
                    hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)(Yb + (pxl + 4) * 256 + wa));
                    typedef __attribute__((ext_vector_type(8))) short s16x8;
                    s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    a[i] = __builtin_bit_cast(bf16x8, t);
                    lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)(Xb + pxl * 256 + wb));
                    hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)(Xb + (pxl + 4) * 256 + wb));
                    s16x8 u = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    b[i] = __builtin_bit_cast(bf16x8, u);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else if constexpr (kSplit) {
            // (see split_bf16) the tile is staged as bf16 hi / lo tiles in the 16-bit path's layout (write_step): two 16-pixel sub-steps of that
            // path's transposed reads, three bf16 MFMAs per tile pair
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            auto tr_read = [&](const char* base, int off) -> bf16x8 {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)(base + off));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)(base + off + 4 * 256));
                const s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                return __builtin_bit_cast(bf16x8, t);
            };
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 ahi[2], alo[2], bhi[2], blo[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int pxl = kk * 16 + 8 * hh + q;
                    const int wa = (((wm * 2 + i) ^ q) << 6) + 32 * cblk + 8 * pp;
                    const int wb = (((wn * 2 + i) ^ q) << 6) + 32 * cblk + 8 * pp;
                    ahi[i] = tr_read(Yb, pxl * 256 + wa); alo[i] = tr_read(Yb + 8192, pxl * 256 + wa);
                    bhi[i] = tr_read(Xb, pxl * 256 + wb); blo[i] = tr_read(Xb + 8192, pxl * 256 + wb);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = mfma_split(ahi[i], alo[i], bhi[j], blo[j], acc[i][j]);
            }
        } else {
#pragma unroll 4
            for (int s = 0; s < 16; ++s) {
                const int pxl = 2 * s + h;
                float a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[i] = *reinterpret_cast<const float*>(Yb + pxl * 512 + (wm * 64 + i * 32 + r) * 4);
                    b[i] = *reinterpret_cast<const float*>(Xb + pxl * 512 + (wn * 64 + i * 32 + r) * 4);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        if (ks + 1 < nk) write_step(buf ^ 1, ryy[PAR ^ 1], rxx[PAR ^ 1]);
        __syncthreads();
    };
    for (int ks = 0; ks < nk; ks += 2) {
        k_step(ks, std::integral_constant<int, 0>{});
        if (ks + 1 < nk) k_step(ks + 1, std::integral_constant<int, 1>{});
    }

    // slab[split][tap][co][ci]
    float* slab = p.slab + (((int64_t)split * p.taps + tap) * p.cout + cob * 128) * p.ktot + kelem0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int ci = wn * 64 + j * 32 + r;
                slab[(int64_t)co * p.ktot + ci] = acc[i][j][e];
            }
}

// dw[(co*ktot + ci)*taps + tap] = sum_s slab[((s*taps + tap)*cout + co)*ktot + ci]
// Workgroup = 64 float4 columns x 4 slab groups (group g sums slabs g, g+4, ... with four independent 16-byte loads in flight per
// thread; ~9 MB in flight over the grid), the four partial sums meet in LDS in a fixed order: deterministic, and at HBM speed
// where one-float-per-thread chains were latency-bound (30 -> 17 us for the 128 x 590 KB slabs of one 3x3 convolution).
template <int NG>       // slab groups per workgroup (64 threads each): 4 for the long 3x3 rows, 16 where the row is short (1x1: 8192 float4
                        // columns = 128 workgroups only, which 4 groups left latency-bound: 60 us for 64 MB)
__global__ __launch_bounds__(64 * NG) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                               int nsplit, int taps, int cout, int ktot) {
    __shared__ f32x4 part[NG - 1][64];
    const int64_t per = (int64_t)taps * cout * ktot;          // multiple of 4 (ktot is a multiple of 64)
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t idx = ((int64_t)blockIdx.x * 64 + lane) * 4;
    const bool live = idx < per;
    f32x4 a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (live) {
        for (int s = grp; s < nsplit; s += 4 * NG) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (s + NG * u < nsplit) a[u] += *reinterpret_cast<const f32x4*>(slab + (int64_t)(s + NG * u) * per + idx);
        }
    }
    const f32x4 mine = (a[0] + a[1]) + (a[2] + a[3]);
    if (grp > 0) part[grp - 1][lane] = mine;
    __syncthreads();
    if (grp == 0 && live) {
        f32x4 tot = mine;
#pragma unroll
        for (int g2 = 0; g2 < NG - 1; ++g2) tot += part[g2][lane];
        const int ci = (int)(idx % ktot);
        const int64_t t2 = idx / ktot;
        const int co = (int)(t2 % cout);
        const int tap = (int)(t2 / cout);
#pragma unroll
        for (int e = 0; e < 4; ++e) dw[((int64_t)co * ktot + ci + e) * taps + tap] = tot[e];
    }
}

static void launch_wgrad_reduce(const float* slab, float* dw, int nsplit, int taps, int cout, int ktot, hipStream_t stream) {
    const int64_t per = (int64_t)taps * cout * ktot;
    const int blocks = (int)((per / 4 + 63) / 64);
    if (blocks < 512 && nsplit >= 64) wgrad_reduce_kernel<16><<<blocks, 1024, 0, stream>>>(slab, dw, nsplit, taps, cout, ktot);
    else wgrad_reduce_kernel<4><<<blocks, 256, 0, stream>>>(slab, dw, nsplit, taps, cout, ktot);
}

// One element of an fp32 pack.  Plain: the fp32 value at position `pos`.  Split (the fp32 mode's default since round 6, see split_bf16): every
// group of 8 consecutive k (32 bytes) holds its eight bf16 hi parts in the first 16 bytes and its eight bf16 lo parts in the second - exactly
// the two 16-byte chunks a lane of the split kernels reads for one MFMA operand, so the weight fragments need no arithmetic at all.
template <typename T>
__device__ __forceinline__ void store_pack(T* base, int64_t pos, float v, bool split) {
    if constexpr (sizeof(T) == 4) {
        if (split) {
            const bf16_t hi = (bf16_t)v, lo = (bf16_t)(v - (float)hi);
            bf16_t* g = reinterpret_cast<bf16_t*>(base + (pos & ~(int64_t)7));
            g[pos & 7] = hi; g[8 + (pos & 7)] = lo;
            return;
        }
    }
    base[pos] = (T)v;
}

// w [Cout][Cin][k][k] fp32 -> w_fwd [tap][Cout][Cin], w_dgrad [taps-1-tap][Cin][Cout]
template <typename T, typename TF = T>
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, TF* __restrict__ wf,
                                                         T* __restrict__ wd, int cout, int cin, int taps, bool split = false) {
    const int64_t n = (int64_t)cout * cin * taps;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int tap = (int)(idx % taps);
    const int64_t t2 = idx / taps;
    const int ci = (int)(t2 % cin);
    const int co = (int)(t2 / cin);
    if (wf) store_pack(wf, ((int64_t)tap * cout + co) * cin + ci, w[idx], split);
    if (wd) store_pack(wd, ((int64_t)(taps - 1 - tap) * cin + ci) * cout + co, w[idx], split);
}

// every convolution of the net in one launch: entry e packs arena[w_off ..] into packs[f_off ..] / packs[d_off ..]
struct PackTable {
    int n;
    int w_off[40], f_off[40], cout[40], cin[40], taps[40];
    unsigned long long fwd_f16;          // bit e: entry e's forward pack is fp16 (16-bit packs only; the data-gradient pack is always T)
};
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const float* __restrict__ arena, T* __restrict__ packs, const PackTable t, bool split = false) {
    const int e = blockIdx.y;
    const int cout = t.cout[e], cin = t.cin[e], taps = t.taps[e];
    const int n = cout * cin * taps;
    const float* w = arena + t.w_off[e];
    T* wf = packs + t.f_off[e];
    T* wd = wf + n;
    const bool f16 = sizeof(T) == 2 && ((t.fwd_f16 >> e) & 1ull);
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n; idx += gridDim.x * 256) {
        const int tap = idx % taps;
        const int t2 = idx / taps;
        const int ci = t2 % cin, co = t2 / cin;
        if (f16) reinterpret_cast<f16_t*>(wf)[(tap * cout + co) * cin + ci] = (f16_t)w[idx];
        else store_pack(wf, (int64_t)(tap * cout + co) * cin + ci, w[idx], split);
        store_pack(wd, (int64_t)((taps - 1 - tap) * cin + ci) * cout + co, w[idx], split);
    }
}

int wgrad_nsplit(int64_t n_pixels, int taps, int cout, int ktot, int* chunk) {
    const int64_t tiles = (int64_t)taps * (cout / 128) * (ktot / 128);
    // 1x1 (HBM-bound): 512 workgroups = one resident round (2 per CU), more slabs only feed the reduce (375 + 60 -> 335 + 11 us at
    // 32 x 32, B = 2048); 3x3 stride-2 (MFMA-bound): two rounds balance better (432 vs 496 us)
    int64_t ns = ((taps == 1 ? 512 : 1024) + tiles - 1) / tiles;
    const int64_t max_ns = (n_pixels + 63) / 64;
    if (ns > max_ns) ns = max_ns;
    if (ns < 1) ns = 1;
    int64_t ch = (n_pixels + ns - 1) / ns;
    ch = (ch + 63) / 64 * 64;
    ns = (n_pixels + ch - 1) / ch;
    *chunk = (int)ch;
    return (int)ns;
}

}  // namespace

extern "C" int64_t gmk_conv_wgrad_workspace_bytes(int64_t n_pixels, int taps, int cout, int ktot) {
    if (n_pixels <= 0 || taps <= 0 || cout % 128 || ktot % 128) return -1;
    int chunk;
    int64_t ns = wgrad_nsplit(n_pixels, taps, cout, ktot, &chunk);
    if (taps == 9) {
        const int64_t ns2 = gmk_wgrad_slots_nsplit(cout, ktot);
        if (ns2 > ns) ns = ns2;
    }
    return ns * taps * cout * ktot * 4;
}

extern "C" int gmk_pack_conv_weight(const float* w, void* w_fwd, void* w_dgrad, int cout, int cin, int ksize, int dtype,
                                    void* stream) {
    GMK_REQUIRE(w && (w_fwd || w_dgrad), "gmk_pack_conv_weight: null pointer");
    if (dtype == GMK_F16) {      // an fp16 pack: the forward operand of the 16-bit mode
        GMK_REQUIRE(cout > 0 && cin > 0 && (ksize == 1 || ksize == 3), "gmk_pack_conv_weight: bad shape");
        const int64_t n16 = (int64_t)cout * cin * ksize * ksize;
        pack_weight_kernel<f16_t><<<(int)((n16 + 255) / 256), 256, 0, gmk_stream(stream)>>>(w, (f16_t*)w_fwd, (f16_t*)w_dgrad, cout, cin, ksize * ksize);
        return gmk_check_launch("gmk_pack_conv_weight");
    }
    GMK_REQUIRE(cout > 0 && cin > 0 && (ksize == 1 || ksize == 3), "gmk_pack_conv_weight: bad shape");
    const int taps = ksize * ksize;
    const int64_t n = (int64_t)cout * cin * taps;
    const int blocks = (int)((n + 255) / 256);
    if (dtype == GMK_BF16)
        pack_weight_kernel<bf16_t><<<blocks, 256, 0, gmk_stream(stream)>>>(w, (bf16_t*)w_fwd, (bf16_t*)w_dgrad, cout, cin,
                                                                           taps);
    else if (dtype == GMK_F32)
        pack_weight_kernel<float><<<blocks, 256, 0, gmk_stream(stream)>>>(w, (float*)w_fwd, (float*)w_dgrad, cout, cin,
                                                                         taps, fp32_split() && cin % 8 == 0 && cout % 8 == 0);
    else
        GMK_REQUIRE(false, "gmk_pack_conv_weight: bad dtype %d", dtype);
    return gmk_check_launch("gmk_pack_conv_weight");
}

extern "C" int gmk_pack_conv_weights_multi(const float* arena, void* packs, int count, const int* w_off, const int* pack_off,
                                           const int* cout, const int* cin, const int* ksize, const int* fwd_f16, int dtype,
                                           void* stream) {
    GMK_REQUIRE(arena && packs && w_off && pack_off && cout && cin && ksize, "gmk_pack_conv_weights_multi: null pointer");
    GMK_REQUIRE(count > 0 && count <= 40, "gmk_pack_conv_weights_multi: 1..40 tensors per call, got %d", count);
    PackTable t;
    t.n = count;
    t.fwd_f16 = 0;
    int nmax = 0;
    for (int e = 0; e < count; ++e) {
        if (fwd_f16 && fwd_f16[e]) {
            GMK_REQUIRE(dtype == GMK_BF16, "gmk_pack_conv_weights_multi: fp16 forward packs go with bf16 data-gradient packs");
            t.fwd_f16 |= 1ull << e;
        }
        GMK_REQUIRE(cout[e] > 0 && cin[e] > 0 && (ksize[e] == 1 || ksize[e] == 3) && w_off[e] >= 0 && pack_off[e] >= 0,
                    "gmk_pack_conv_weights_multi: bad entry %d", e);
        t.w_off[e] = w_off[e]; t.f_off[e] = pack_off[e]; t.cout[e] = cout[e]; t.cin[e] = cin[e]; t.taps[e] = ksize[e] * ksize[e];
        const int n = cout[e] * cin[e] * t.taps[e];
        nmax = n > nmax ? n : nmax;
    }
    int bx = (nmax + 255) / 256;
    if (bx > 256) bx = 256;
    const dim3 grid(bx, count);
    if (dtype == GMK_BF16) pack_weights_multi_kernel<bf16_t><<<grid, 256, 0, gmk_stream(stream)>>>(arena, (bf16_t*)packs, t);
    else if (dtype == GMK_F32) pack_weights_multi_kernel<float><<<grid, 256, 0, gmk_stream(stream)>>>(arena, (float*)packs, t, fp32_split());
    else GMK_REQUIRE(false, "gmk_pack_conv_weights_multi: bad dtype %d", dtype);
    return gmk_check_launch("gmk_pack_conv_weights_multi");
}

static int conv_igemm_impl(const void* src0, const void* src1, int c0, int c1, int B, int hs, int ws, int ho, int wo,
                           int ksize, int mode, const void* w, int w_rows, int n0, int cout, const float* bias,
                           const float* emb, int emb_stride, const void* residual, void* out, void* out2, int out_cstride,
                           float* gn_stats, int64_t gn_stats_bytes, const float* gn_scale, const float* gn_shift, int gn_stride,
                           int dtype, void* stream) {
    GMK_REQUIRE(src0 && w && out, "gmk_conv_igemm: null pointer");
    GMK_REQUIRE(dtype == GMK_BF16 || dtype == GMK_F16 || dtype == GMK_F32, "gmk_conv_igemm: bad dtype %d", dtype);
    GMK_REQUIRE(ksize == 1 || ksize == 3, "gmk_conv_igemm: ksize %d", ksize);
    GMK_REQUIRE(c0 > 0 && c0 % 128 == 0 && c1 >= 0 && c1 % 128 == 0 && (c1 == 0 || src1),
                "gmk_conv_igemm: source channels must be multiples of 128 (c0=%d c1=%d)", c0, c1);
    GMK_REQUIRE(cout > 0 && cout % 128 == 0 && n0 >= 0 && n0 + cout <= w_rows && out_cstride >= cout,
                "gmk_conv_igemm: bad output channels n0=%d cout=%d w_rows=%d cstride=%d", n0, cout, w_rows, out_cstride);
    GMK_REQUIRE(B > 0 && (int64_t)B * ho * wo < (1ll << 31) - 256 && (int64_t)B * hs * ws < (1ll << 31),
                "gmk_conv_igemm: problem too large for 32-bit pixel indices");
    ConvParams p;
    GMK_REQUIRE(make_gather(mode, ksize, hs, ws, ho, wo, &p.g), "gmk_conv_igemm: mode %d inconsistent with %dx%d -> %dx%d",
                mode, hs, ws, ho, wo);
    GMK_REQUIRE(!emb || emb_stride >= cout, "gmk_conv_igemm: emb_stride");
    p.src0 = src0; p.src1 = src1; p.c0 = c0; p.c1 = c1; p.ktot = c0 + c1;
    p.w = w; p.w_tap_stride = (int64_t)w_rows * (c0 + c1); p.n0 = n0;
    p.bias = bias; p.emb = emb; p.emb_stride = emb_stride; p.residual = residual; p.out = out; p.out2 = nullptr;
    p.out_cstride = out_cstride; p.M = B * ho * wo;
    // kernel choice: 0 = automatic, 1 = register-staged 128x128, 2 = LDS-DMA im2col 256x128, 3 = LDS halo (3x3 s1 bf16)
    const int force = gmk_kernel_choice(0, "GMK_CONV_KERNEL");
    const int es = gmk_esize(dtype);
    // kBadOff and (kBadPix * bytes-per-pixel) mod 2^32 (>= 0xFFFFF000 for pixels of up to 4 KiB) must lie beyond every buffer
    const int64_t lim = 0xFFFF0000ll;
    const int64_t nb0 = (int64_t)B * hs * ws * c0 * es, nb1 = (int64_t)B * hs * ws * c1 * es;
    const int64_t nbw = (int64_t)ksize * ksize * w_rows * (c0 + c1) * es;
    // Tile-count floor of the halo kernel: NONE for the plain and the nearest-x2 forms since round 4 - the reference's own sampler sizes
    // (25 images in `evaluate`, diffusion_model.py:98-104) put the 14 x 14 / 7 x 7 levels at 20 / 5 tiles, where the register-staged im2col
    // kernel took 32 us a launch against the halo kernel's 19 (as half jobs on twice the CUs): B = 25 DDIM step 1.25 -> 1.15 ms, B = 8
    // 1.46 -> 1.16 ms.  The zero-stuffed transposed form keeps 32 (below it the four parity phases win).
    // the transposed form (data gradient of the stride-2 conv) is the plain 3x3 conv of the zero-stuffed gradient: on the halo
    // kernel 3/4 of the resident pixels are zeros, but it still beats the im2col gather by 1.7x (215 vs 360 us at 28x28)
    const bool stuffed = mode == GMK_CONV_TRANSPOSED2 && !((ho | wo) & 1);
    // The transposed form in its sub-pixel shape (conv_subpixel.hip, round 5): output parities meet 1 / 2 / 2 / 4 of the 9 taps over the gradient's
    // own grid, the halo of a low-resolution tile stays in LDS for all four parities - the zero-stuffed form below multiplies 27 of 36 tap-images by zero.
    if (stuffed && force == 0 && gmk_is16(dtype) && ksize == 3 && c0 == 128 && c1 == 0 && cout == 128 && !emb && !gn_scale && !gn_stats && !out2 &&
        ho == 2 * hs && wo == 2 * ws && n0 >= 0 && n0 + cout <= w_rows && gmk_conv_subpixel_takes(B, hs, ws, c0, cout, w_rows, 9, out_cstride, dtype))
        return gmk_conv_subpixel(src0, B, hs, ws, c0, w, w_rows, n0, cout, GMK_SUBPIXEL_TRANSPOSED, bias, residual, out, out_cstride, dtype, stream);
    if ((force == 0 || force == 3) && gmk_is16(dtype) && (mode == GMK_CONV_NORMAL || mode == GMK_CONV_UPSAMPLE2 || stuffed) && ksize == 3) {
        const int rc = gmk_conv3x3_halo_try(src0, src1, c0, c1, B, ho, wo, w, w_rows, n0, cout, bias, emb, emb_stride, residual,
                                            out, out_cstride, (force == 3 || !stuffed) ? 1 : 32, stuffed ? 2 : mode == GMK_CONV_UPSAMPLE2, gn_stats,
                                            gn_stats_bytes, gn_scale, gn_shift, gn_stride, dtype, gmk_stream(stream));
        if (rc == 1 || rc == 2) {
            gmk_note_kernel(stuffed ? 5 : rc == 2 ? 4 : 3);       // 5: a halo kernel on the zero-stuffed source (transposed conv)
            return gmk_check_launch("gmk_conv_igemm(halo)");
        }
    }
    // Where the halo kernel does not take the transposed form (fp32, fewer than 32 tiles, GMK_CONV_KERNEL=2): four output-parity phases on
    // the LDS-DMA kernel, each a 1- / 2- / 2- / 4-tap convolution over the gradient's own grid with a stride-2 scatter of its rows
    // (make_phase_gather): 9 / 4 tap-images of MFMA work where the masked im2col gather does 9.  At the train step's sizes the zero-stuffed
    // halo form stays in front (536 vs 565 us at 16x16 -> 32x32, B = 2048, with the residual: each phase launch walks every DRAM page
    // of the output and of the residual for a quarter of their bytes; without a residual the phases win, 393 vs 462 us; docs/EXPERIMENTS.md 7b.9).
    if (stuffed && force != 1 && !out2 && !gn_scale && !gn_stats) {
        const int64_t nbo = (int64_t)B * ho * wo * out_cstride * es;
        if (nb0 < lim && nb1 < lim && nbw < lim && nbo < lim && (int64_t)B * hs * ws < 0x00FFFFFF && (int64_t)c0 * es <= 4096 &&
            (int64_t)c1 * es <= 4096) {
            gmk_note_kernel(6);
            const int ncu = gmk_cu_limit();
            for (int ph = 0; ph < 4; ++ph) {
                ConvParams q = p;
                make_phase_gather(ph >> 1, ph & 1, hs, ws, ho, wo, &q.g);
                q.M = B * hs * ws;
                q.nb0 = (unsigned)nb0; q.nb1 = (unsigned)nb1; q.nbw = (unsigned)nbw; q.nbo = (unsigned)nbo;
                const int ntiles = (q.M + 255) / 256;
                dim3 grid(ntiles < ncu ? ntiles : ncu, cout / kBN);
                if (dtype == GMK_BF16) conv_igemm_dma_kernel<bf16_t><<<grid, 512, 0, gmk_stream(stream)>>>(q);
                else if (dtype == GMK_F16) conv_igemm_dma_kernel<f16_t><<<grid, 512, 0, gmk_stream(stream)>>>(q);
                else if (fp32_split()) conv_igemm_dma_kernel<float, true><<<grid, 512, 0, gmk_stream(stream)>>>(q);
                else conv_igemm_dma_kernel<float><<<grid, 512, 0, gmk_stream(stream)>>>(q);
            }
            return gmk_check_launch("gmk_conv_igemm(stride-2 dgrad phases)");
        }
    }
    GMK_REQUIRE(!gn_scale, "gmk_conv_igemm: the fused GroupNorm-apply needs the 3x3 halo kernel (16-bit, plain 3x3, a tile within two samples): "
                           "ask gmk_conv_gn_fusable first");
    // LDS-DMA kernel: problems with at least ~2 tiles of 256 pixels per CU, buffers addressable with 32-bit offsets
    const int64_t nbo = (int64_t)p.M * out_cstride * es;
    const bool fits = nb0 < lim && nb1 < lim && nbw < lim && nbo < lim && (int64_t)B * hs * ws < 0x00FFFFFF &&
                      (int64_t)c0 * es <= 4096 && (int64_t)c1 * es <= 4096;
    const bool dma = fits && (force == 2 || (force != 1 && p.M >= 256 * 512));
    if (out2 && !dma) {        // the pair form lives in the LDS-DMA kernel only: small problems run as two plain launches
        const int rc = conv_igemm_impl(src0, src1, c0, c1, B, hs, ws, ho, wo, ksize, mode, w, w_rows, n0, cout, bias, emb, emb_stride,
                                       residual, out, nullptr, out_cstride, nullptr, 0, nullptr, nullptr, 0, dtype, stream);
        if (rc) return rc;
        return conv_igemm_impl(src0, src1, c0, c1, B, hs, ws, ho, wo, ksize, mode, w, w_rows, n0 + cout, cout, bias, emb, emb_stride,
                               residual, out2, nullptr, out_cstride, nullptr, 0, nullptr, nullptr, 0, dtype, stream);
    }
    p.out2 = out2;
    gmk_note_kernel(dma ? 2 : 1);
    if (dma) {
        p.nb0 = (unsigned)nb0; p.nb1 = (unsigned)nb1; p.nbw = (unsigned)nbw; p.nbo = (unsigned)nbo;
        const int ncu = gmk_cu_limit();
        const int ntiles = (p.M + 255) / 256;
        dim3 grid(ntiles < ncu ? ntiles : ncu, cout / kBN);
        if (dtype == GMK_BF16) conv_igemm_dma_kernel<bf16_t><<<grid, 512, 0, gmk_stream(stream)>>>(p);
        else if (dtype == GMK_F16) conv_igemm_dma_kernel<f16_t><<<grid, 512, 0, gmk_stream(stream)>>>(p);
        else if (fp32_split()) conv_igemm_dma_kernel<float, true><<<grid, 512, 0, gmk_stream(stream)>>>(p);
        else conv_igemm_dma_kernel<float><<<grid, 512, 0, gmk_stream(stream)>>>(p);
    } else {
        dim3 grid((p.M + kBM - 1) / kBM, cout / kBN);
        if (dtype == GMK_BF16) conv_igemm_kernel<bf16_t><<<grid, 256, 0, gmk_stream(stream)>>>(p);
        else if (dtype == GMK_F16) conv_igemm_kernel<f16_t><<<grid, 256, 0, gmk_stream(stream)>>>(p);
        else if (fp32_split()) conv_igemm_kernel<float, true><<<grid, 256, 0, gmk_stream(stream)>>>(p);
        else conv_igemm_kernel<float><<<grid, 256, 0, gmk_stream(stream)>>>(p);
    }
    return gmk_check_launch("gmk_conv_igemm");
}

extern "C" int gmk_conv_igemm(const void* src0, const void* src1, int c0, int c1, int B, int hs, int ws, int ho, int wo,
                              int ksize, int mode, const void* w, int w_rows, int n0, int cout, const float* bias,
                              const float* emb, int emb_stride, const void* residual, void* out, int out_cstride,
                              float* gn_stats, int64_t gn_stats_bytes, const float* gn_scale, const float* gn_shift, int gn_stride,
                              int dtype, void* stream) {
    return conv_igemm_impl(src0, src1, c0, c1, B, hs, ws, ho, wo, ksize, mode, w, w_rows, n0, cout, bias, emb, emb_stride, residual,
                           out, nullptr, out_cstride, gn_stats, gn_stats_bytes, gn_scale, gn_shift, gn_stride, dtype, stream);
}

extern "C" int gmk_conv1x1_pair(const void* src, int c, int B, int H, int W, const void* w, int w_rows, int n0, void* out_a,
                                void* out_b, int dtype, void* stream) {
    GMK_REQUIRE(src && w && out_a && out_b, "gmk_conv1x1_pair: null pointer");
    GMK_REQUIRE(n0 >= 0 && n0 + 256 <= w_rows, "gmk_conv1x1_pair: rows n0 .. n0 + 255 outside the %d packed rows", w_rows);
    if (c == 128 && B > 0 && H > 0 && W > 0) {             // the streaming form for the train step's sizes (conv1x1_stream.hip)
        const size_t es = 2;
        if (gmk_conv1x1_pair_stream_try(src, (int64_t)B * H * W, (const char*)w + (size_t)n0 * 128 * es, out_a, out_b, dtype, gmk_stream(stream))) {
            gmk_note_kernel(14);
            return gmk_check_launch("gmk_conv1x1_pair(stream)");
        }
    }
    return conv_igemm_impl(src, nullptr, c, 0, B, H, W, H, W, 1, GMK_CONV_NORMAL, w, w_rows, n0, 128, nullptr, nullptr, 0, nullptr,
                           out_a, out_b, 128, nullptr, 0, nullptr, nullptr, 0, dtype, stream);
}

// 1 if gmk_conv_igemm(ksize 3, GMK_CONV_NORMAL, bf16) of this shape runs on the halo kernel AND can apply a GroupNorm to its source
extern "C" int gmk_conv_gn_fusable(int B, int H, int W, int c0, int c1, int cout) {
    const int force = gmk_kernel_choice(0, "GMK_CONV_KERNEL");
    if (force != 0 && force != 3) return 0;
    // the same rule gmk_conv_igemm applies when it hands the launch to the halo kernel (weights [9][cout][c0 + c1], dense output)
    return gmk_halo_geometry(B, H, W, c0, c1, cout, cout, cout, 1, 0, 1, nullptr);
}

extern "C" int gmk_conv_wgrad(const void* dy, int dy_cstride, const void* src0, const void* src1, int c0, int c1, int B,
                              int hs, int ws, int ho, int wo, int ksize, int mode, float* dw, int cout, void* workspace,
                              int64_t workspace_bytes, int dtype, int x_dtype, void* stream) {
    GMK_REQUIRE(dy && src0 && dw && workspace, "gmk_conv_wgrad: null pointer");
    GMK_REQUIRE(dtype == GMK_BF16 || dtype == GMK_F32, "gmk_conv_wgrad: bad dtype %d", dtype);
    GMK_REQUIRE(x_dtype == dtype || (dtype == GMK_BF16 && x_dtype == GMK_F16),
                "gmk_conv_wgrad: activations of type %d with gradients of type %d (same type, or fp16 activations with bf16 gradients)", x_dtype, dtype);
    const bool xf16 = x_dtype == GMK_F16;
    GMK_REQUIRE(ksize == 1 || ksize == 3, "gmk_conv_wgrad: ksize %d", ksize);
    GMK_REQUIRE(c0 > 0 && c0 % 128 == 0 && c1 >= 0 && c1 % 128 == 0 && (c1 == 0 || src1),
                "gmk_conv_wgrad: source channels must be multiples of 128 (c0=%d c1=%d)", c0, c1);
    GMK_REQUIRE(cout > 0 && cout % 128 == 0 && dy_cstride >= cout, "gmk_conv_wgrad: bad cout=%d", cout);
    GMK_REQUIRE(B > 0 && (int64_t)B * ho * wo < (1ll << 31) - 256 && (int64_t)B * hs * ws < (1ll << 31),
                "gmk_conv_wgrad: problem too large for 32-bit pixel indices");
    GMK_REQUIRE(mode != GMK_CONV_TRANSPOSED2, "gmk_conv_wgrad: no weight gradient for the transposed gather");
    WgradParams p;
    GMK_REQUIRE(make_gather(mode, ksize, hs, ws, ho, wo, &p.g), "gmk_conv_wgrad: mode %d inconsistent with %dx%d -> %dx%d",
                mode, hs, ws, ho, wo);
    const int taps = ksize * ksize;
    // kernel choice: 0 = automatic, 1 = im2col split-K, 2 = padded-slot correlation (3x3 s1 bf16)
    const int wforce = gmk_kernel_choice(1, "GMK_WGRAD_KERNEL");
    // stride 2 (round 6): the four-plane form of the slot kernel where the source is exactly twice the output (GMK_WGRAD_KERNEL=1 keeps the im2col kernel)
    const bool s2 = mode == GMK_CONV_STRIDE2 && hs == 2 * ho && ws == 2 * wo;
    if (wforce != 1 && dtype == GMK_BF16 && (mode == GMK_CONV_NORMAL || mode == GMK_CONV_UPSAMPLE2 || s2) && ksize == 3) {
        const int ns2 = gmk_conv_wgrad_slots_try(dy, dy_cstride, src0, src1, c0, c1, B, ho, wo, cout, (float*)workspace,
                                                 workspace_bytes, wforce, mode == GMK_CONV_UPSAMPLE2, xf16, gmk_stream(stream), s2);
        if (ns2 > 0) {
            int rc2 = gmk_check_launch("gmk_conv_wgrad(slots)");
            if (rc2) return rc2;
            launch_wgrad_reduce((const float*)workspace, dw, ns2, taps, cout, c0 + c1, gmk_stream(stream));
            return gmk_check_launch("gmk_conv_wgrad(reduce)");
        }
    }
    if (wforce != 1 && ksize == 1 && mode == GMK_CONV_NORMAL && dtype == GMK_BF16 && c0 == 128 && c1 == 128 && cout == 128) {
        // the skip convolution over the concatenated input at the train step's sizes: one workgroup per pixel range owns both input blocks
        const int ns1 = gmk_conv1x1_wgrad_stream_try(dy, dy_cstride, src0, src1, (int64_t)B * ho * wo, (float*)workspace, workspace_bytes, xf16,
                                                     gmk_stream(stream));
        if (ns1 > 0) {
            gmk_note_kernel(15);
            int rc1 = gmk_check_launch("gmk_conv_wgrad(1x1 stream)");
            if (rc1) return rc1;
            launch_wgrad_reduce((const float*)workspace, dw, ns1, 1, cout, 256, gmk_stream(stream));
            return gmk_check_launch("gmk_conv_wgrad(reduce)");
        }
    }
    p.dy = dy; p.dy_cstride = dy_cstride; p.src0 = src0; p.src1 = src1; p.c0 = c0; p.c1 = c1; p.ktot = c0 + c1;
    p.cout = cout; p.M = B * ho * wo;
    const int ns = wgrad_nsplit(p.M, taps, cout, p.ktot, &p.chunk);
    const int64_t need = (int64_t)ns * taps * cout * p.ktot * 4;
    GMK_REQUIRE(workspace_bytes >= need, "gmk_conv_wgrad: workspace %lld < %lld bytes", (long long)workspace_bytes,
                (long long)need);
    p.slab = (float*)workspace;
    gmk_note_kernel(11);
    p.ns = ns; p.taps = taps; p.order = (gmk_kernel_choice(3, "GMK_DEV_VARIANT") & 0xFF) == 61;
    dim3 grid((ns + 7) / 8 * 8 * taps * (cout / 128) * (p.ktot / 128));
    if (dtype == GMK_BF16 && xf16) conv_wgrad_kernel<bf16_t, f16_t><<<grid, 256, 0, gmk_stream(stream)>>>(p);
    else if (dtype == GMK_BF16) conv_wgrad_kernel<bf16_t><<<grid, 256, 0, gmk_stream(stream)>>>(p);
    else if (fp32_split()) conv_wgrad_kernel<float, float, true><<<grid, 256, 0, gmk_stream(stream)>>>(p);
    else conv_wgrad_kernel<float><<<grid, 256, 0, gmk_stream(stream)>>>(p);
    int rc = gmk_check_launch("gmk_conv_wgrad");
    if (rc) return rc;
    launch_wgrad_reduce(p.slab, dw, ns, taps, cout, p.ktot, gmk_stream(stream));
    return gmk_check_launch("gmk_conv_wgrad(reduce)");
}
