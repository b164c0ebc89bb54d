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
#include "gmk_common.h"

namespace {

constexpr int kBM = 128, kBN = 128;

struct GatherParams {
    int hs, ws;          // source spatial size
    int ho, wo;          // output spatial size
    int ksize, pad;
    int mul, shift, mask, lim_h, lim_w;
};

__host__ bool make_gather(int mode, int ksize, int hs, int ws, int ho, int wo, GatherParams* g) {
    g->hs = hs; g->ws = ws; g->ho = ho; g->wo = wo;
    g->ksize = ksize; g->pad = ksize == 3 ? 1 : 0;
    g->mul = 1; g->shift = 0; g->mask = 0; g->lim_h = hs; g->lim_w = ws;
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

struct ConvParams {
    const void* src0; const void* src1;
    int c0, c1, ktot;
    GatherParams g;
    const void* w; int64_t w_tap_stride;   // elements
    int n0;
    const float* bias; const float* emb; int emb_stride;
    const void* residual; void* out; int out_cstride;
    int M;   // B*ho*wo
};

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <typename T>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvParams p) {
    constexpr int ES = sizeof(T);
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
                bf16x8 a[2], b[2];
                a[0] = *reinterpret_cast<const bf16x8*>(Ab + a_off + coff);
                a[1] = *reinterpret_cast<const bf16x8*>(Ab + a_off + 4096 + coff);
                b[0] = *reinterpret_cast<const bf16x8*>(Bb + b_off + coff);
                b[1] = *reinterpret_cast<const bf16x8*>(Bb + b_off + 4096 + coff);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
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
        store4((T*)p.out + o, v);
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
};

template <typename T>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradParams p) {
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
    const int tap = blockIdx.y;
    const int ky = tap / g.ksize, kx = tap - ky * g.ksize;
    const int ncib = p.ktot / 128;
    const int cib = blockIdx.z % ncib, cob = blockIdx.z / ncib;
    const int kelem0 = cib * 128;
    const T* src; int cs, ci_off;
    if (kelem0 < p.c0) { src = (const T*)p.src0; cs = p.c0; ci_off = kelem0; }
    else { src = (const T*)p.src1; cs = p.c1; ci_off = kelem0 - p.c0; }
    const T* dy = (const T*)p.dy + cob * 128;

    const int pix_begin = blockIdx.x * p.chunk;
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
        else lds_w[i] = row * 512 + sc * 16;
    }
    u32x4 ry[4], rx[4];
    auto load_step = [&](int ks) {
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
    auto write_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(Ys + buf * 16384 + lds_w[i]) = ry[i];
            *reinterpret_cast<u32x4*>(Xs + buf * 16384 + lds_w[i]) = rx[i];
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
        load_step(0);
        write_step(0);
    }
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nk) load_step(ks + 1);
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
        if (ks + 1 < nk) write_step(buf ^ 1);
        __syncthreads();
    }

    // slab[split][tap][co][ci]
    float* slab = p.slab + (((int64_t)blockIdx.x * gridDim.y + tap) * p.cout + cob * 128) * p.ktot + kelem0;
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
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                          int nsplit, int taps, int cout, int ktot) {
    const int64_t per = (int64_t)taps * cout * ktot;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per) return;
    const int ci = (int)(idx % ktot);
    const int64_t t2 = idx / ktot;
    const int co = (int)(t2 % cout);
    const int tap = (int)(t2 / cout);
    float s0 = 0.f, s1 = 0.f;
    int s = 0;
    for (; s + 1 < nsplit; s += 2) {
        s0 += slab[(int64_t)s * per + idx];
        s1 += slab[(int64_t)(s + 1) * per + idx];
    }
    if (s < nsplit) s0 += slab[(int64_t)s * per + idx];
    dw[((int64_t)co * ktot + ci) * taps + tap] = s0 + s1;
}

// w [Cout][Cin][k][k] fp32 -> w_fwd [tap][Cout][Cin], w_dgrad [taps-1-tap][Cin][Cout]
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ wf,
                                                         T* __restrict__ wd, int cout, int cin, int taps) {
    const int64_t n = (int64_t)cout * cin * taps;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int tap = (int)(idx % taps);
    const int64_t t2 = idx / taps;
    const int ci = (int)(t2 % cin);
    const int co = (int)(t2 / cin);
    const T v = (T)w[idx];
    if (wf) wf[((int64_t)tap * cout + co) * cin + ci] = v;
    if (wd) wd[((int64_t)(taps - 1 - tap) * cin + ci) * cout + co] = v;
}

int wgrad_nsplit(int64_t n_pixels, int taps, int cout, int ktot, int* chunk) {
    const int64_t tiles = (int64_t)taps * (cout / 128) * (ktot / 128);
    int64_t ns = (1024 + tiles - 1) / tiles;
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
    const int ns = wgrad_nsplit(n_pixels, taps, cout, ktot, &chunk);
    return (int64_t)ns * taps * cout * ktot * 4;
}

extern "C" int gmk_pack_conv_weight(const float* w, void* w_fwd, void* w_dgrad, int cout, int cin, int ksize, int dtype,
                                    void* stream) {
    GMK_REQUIRE(w && (w_fwd || w_dgrad), "gmk_pack_conv_weight: null pointer");
    GMK_REQUIRE(cout > 0 && cin > 0 && (ksize == 1 || ksize == 3), "gmk_pack_conv_weight: bad shape");
    const int taps = ksize * ksize;
    const int64_t n = (int64_t)cout * cin * taps;
    const int blocks = (int)((n + 255) / 256);
    if (dtype == GMK_BF16)
        pack_weight_kernel<bf16_t><<<blocks, 256, 0, gmk_stream(stream)>>>(w, (bf16_t*)w_fwd, (bf16_t*)w_dgrad, cout, cin,
                                                                           taps);
    else if (dtype == GMK_F32)
        pack_weight_kernel<float><<<blocks, 256, 0, gmk_stream(stream)>>>(w, (float*)w_fwd, (float*)w_dgrad, cout, cin,
                                                                         taps);
    else
        GMK_REQUIRE(false, "gmk_pack_conv_weight: bad dtype %d", dtype);
    return gmk_check_launch("gmk_pack_conv_weight");
}

extern "C" int gmk_conv_igemm(const void* src0, const void* src1, int c0, int c1, int B, int hs, int ws, int ho, int wo,
                              int ksize, int mode, const void* w, int w_rows, int n0, int cout, const float* bias,
                              const float* emb, int emb_stride, const void* residual, void* out, int out_cstride,
                              int dtype, void* stream) {
    GMK_REQUIRE(src0 && w && out, "gmk_conv_igemm: null pointer");
    GMK_REQUIRE(dtype == GMK_BF16 || dtype == GMK_F32, "gmk_conv_igemm: bad dtype %d", dtype);
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
    p.bias = bias; p.emb = emb; p.emb_stride = emb_stride; p.residual = residual; p.out = out;
    p.out_cstride = out_cstride; p.M = B * ho * wo;
    dim3 grid((p.M + kBM - 1) / kBM, cout / kBN);
    if (dtype == GMK_BF16) conv_igemm_kernel<bf16_t><<<grid, 256, 0, gmk_stream(stream)>>>(p);
    else conv_igemm_kernel<float><<<grid, 256, 0, gmk_stream(stream)>>>(p);
    return gmk_check_launch("gmk_conv_igemm");
}

extern "C" int gmk_conv_wgrad(const void* dy, int dy_cstride, const void* src0, const void* src1, int c0, int c1, int B,
                              int hs, int ws, int ho, int wo, int ksize, int mode, float* dw, int cout, void* workspace,
                              int64_t workspace_bytes, int dtype, void* stream) {
    GMK_REQUIRE(dy && src0 && dw && workspace, "gmk_conv_wgrad: null pointer");
    GMK_REQUIRE(dtype == GMK_BF16 || dtype == GMK_F32, "gmk_conv_wgrad: bad dtype %d", dtype);
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
    p.dy = dy; p.dy_cstride = dy_cstride; p.src0 = src0; p.src1 = src1; p.c0 = c0; p.c1 = c1; p.ktot = c0 + c1;
    p.cout = cout; p.M = B * ho * wo;
    const int ns = wgrad_nsplit(p.M, taps, cout, p.ktot, &p.chunk);
    const int64_t need = (int64_t)ns * taps * cout * p.ktot * 4;
    GMK_REQUIRE(workspace_bytes >= need, "gmk_conv_wgrad: workspace %lld < %lld bytes", (long long)workspace_bytes,
                (long long)need);
    p.slab = (float*)workspace;
    dim3 grid(ns, taps, (cout / 128) * (p.ktot / 128));
    if (dtype == GMK_BF16) conv_wgrad_kernel<bf16_t><<<grid, 256, 0, gmk_stream(stream)>>>(p);
    else conv_wgrad_kernel<float><<<grid, 256, 0, gmk_stream(stream)>>>(p);
    int rc = gmk_check_launch("gmk_conv_wgrad");
    if (rc) return rc;
    const int64_t per = (int64_t)taps * cout * p.ktot;
    wgrad_reduce_kernel<<<(int)((per + 255) / 256), 256, 0, gmk_stream(stream)>>>(p.slab, dw, ns, taps, cout, p.ktot);
    return gmk_check_launch("gmk_conv_wgrad(reduce)");
}
