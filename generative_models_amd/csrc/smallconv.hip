// Stem (C_in <= 4 -> C) and head (C -> C_out <= 4) 3x3 convolutions with their backward passes.
// Reference: gms/diffusion/simple_unet.py:92-94 (Downsample(1, channels, 1)) and :41 (Conv2d(channels, 1, 3, padding=1)).
// Degenerate GEMM shapes (K = 9 or N = 1): HBM-bound streaming kernels, no MFMA.  Image-side tensors are NCHW
// fp32 as the reference passes them; the network side is NHWC in `dtype`.
#include "gmk_common.h"

namespace {

constexpr int kMaxSmall = 4;   // max image channels handled (1 in the reference, 3 for the CIFAR-shape configs)

// ---- stem forward: thread = (pixel, 8-channel vector) -----------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, T* __restrict__ y, int B, int cin,
                                                      int H, int W, int C) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [cin*9][C] then bias[C]
    const int nw = cin * 9;
    for (int i = threadIdx.x; i < nw * C; i += blockDim.x) {
        const int co = i % C, k = i / C;                          // k = ci*9 + tap
        wl[i] = w[(int64_t)co * nw + k];
    }
    for (int i = threadIdx.x; i < C; i += blockDim.x) wl[nw * C + i] = bias[i];
    __syncthreads();
    const int nvec = C >> 3;
    const int64_t total = (int64_t)B * H * W * nvec;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int vec = (int)(idx % nvec);
        int64_t pix = idx / nvec;
        const int ox = (int)(pix % W);
        const int64_t t2 = pix / W;
        const int oy = (int)(t2 % H);
        const int64_t b = t2 / H;
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = wl[nw * C + vec * 8 + i];
        for (int ci = 0; ci < cin; ++ci) {
            const float* xp = x + (b * cin + ci) * (int64_t)H * W;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy + ky - 1;
                if (iy < 0 || iy >= H) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = ox + kx - 1;
                    if (ix < 0 || ix >= W) continue;
                    const float xv = xp[iy * W + ix];
                    const float* wp = wl + (ci * 9 + ky * 3 + kx) * C + vec * 8;
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = fmaf(xv, wp[i], acc[i]);
                }
            }
        }
        store8(y + pix * C + vec * 8, acc);
    }
}

// ---- stem weight gradient: block = pixel range; thread = (pixel lane, 8-channel vector) ------------------
template <typename T>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, const T* __restrict__ dy,
                                                        float* __restrict__ part, int64_t npix, int cin, int H, int W,
                                                        int C, int64_t pix_per_blk) {
    __shared__ float red[256 * 8];
    const int tid = threadIdx.x;
    const int nvec = C >> 3, planes = 256 / nvec;
    const int vec = tid % nvec, pl = tid / nvec;
    const int64_t p0 = blockIdx.x * pix_per_blk;
    const int64_t p1 = min(p0 + pix_per_blk, npix);
    float* out = part + (int64_t)blockIdx.x * C * cin * 9;
    for (int ci = 0; ci < cin; ++ci) {
        float acc[9][8];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[t][i] = 0.f;
        for (int64_t pix = p0 + pl; pix < p1; pix += planes) {
            const int ox = (int)(pix % W);
            const int64_t t2 = pix / W;
            const int oy = (int)(t2 % H);
            const int64_t b = t2 / H;
            float d[8];
            load8(dy + pix * C + vec * 8, d);
            const float* xp = x + (b * cin + ci) * (int64_t)H * W;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy + ky - 1;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = ox + kx - 1;
                    const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
                    const float xv = ok ? xp[iy * W + ix] : 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[ky * 3 + kx][i] = fmaf(xv, d[i], acc[ky * 3 + kx][i]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) red[tid * 8 + i] = acc[t][i];
            __syncthreads();
            if (tid < C) {
                const int vv = tid >> 3, i = tid & 7;
                float s = 0.f;
                for (int p = 0; p < planes; ++p) s += red[(p * nvec + vv) * 8 + i];
                out[((int64_t)tid * cin + ci) * 9 + t] = s;     // reference layout [C][cin][3][3]
            }
        }
    }
}

// ---- head forward: nvec threads per pixel, each 8 channels x 9 taps, reduced with wave shuffles ----------
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ a, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ out, int B,
                                                      int cout, int H, int W, int C) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [cout][9][C]
    for (int i = threadIdx.x; i < cout * 9 * C; i += blockDim.x) {
        const int ci = i % C, t = (i / C) % 9, co = i / (9 * C);
        wl[i] = w[((int64_t)co * C + ci) * 9 + t];
    }
    __syncthreads();
    const int nvec = C >> 3;                 // 16 or 32: a power of two inside one wave
    const int64_t npix = (int64_t)B * H * W;
    const int64_t total = npix * nvec;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // every lane of a wave iterates the same number of times (total is a multiple of 64 when padded)
    const int64_t total_pad = (total + 63) / 64 * 64;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total_pad; idx += stride) {
        const bool live = idx < total;
        const int vec = (int)(idx % nvec);
        const int64_t pix = live ? idx / nvec : 0;
        const int ox = (int)(pix % W);
        const int64_t t2 = pix / W;
        const int oy = (int)(t2 % H);
        const int64_t b = t2 / H;
        float acc[kMaxSmall] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy + ky - 1;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox + kx - 1;
                if (!live || iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                float v[8];
                load8(a + (((b * H + iy) * W) + ix) * (int64_t)C + vec * 8, v);
                for (int co = 0; co < cout; ++co) {
                    const float* wp = wl + (co * 9 + ky * 3 + kx) * C + vec * 8;
                    float s = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) s = fmaf(v[i], wp[i], s);
                    acc[co] += s;
                }
            }
        }
        for (int co = 0; co < cout; ++co) {
            float s = acc[co];
            for (int off = nvec >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            if (live && vec == 0) out[((b * cout + co) * H + oy) * (int64_t)W + ox] = s + bias[co];
        }
    }
}

// ---- head data gradient: thread = (pixel, 8-channel vector) ----------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void head_dgrad_kernel(const float* __restrict__ dout, const float* __restrict__ w,
                                                        T* __restrict__ da, int B, int cout, int H, int W, int C) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [cout][9][C]
    for (int i = threadIdx.x; i < cout * 9 * C; i += blockDim.x) {
        const int ci = i % C, t = (i / C) % 9, co = i / (9 * C);
        wl[i] = w[((int64_t)co * C + ci) * 9 + t];
    }
    __syncthreads();
    const int nvec = C >> 3;
    const int64_t total = (int64_t)B * H * W * nvec;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int vec = (int)(idx % nvec);
        const int64_t pix = idx / nvec;
        const int x0 = (int)(pix % W);
        const int64_t t2 = pix / W;
        const int y0 = (int)(t2 % H);
        const int64_t b = t2 / H;
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        // out[oy][ox] used a[oy+ky-1][ox+kx-1]  =>  a[y0][x0] feeds out[y0-ky+1][x0-kx+1]
        for (int co = 0; co < cout; ++co) {
            const float* dp = dout + (b * cout + co) * (int64_t)H * W;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int oy = y0 - ky + 1;
                if (oy < 0 || oy >= H) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ox = x0 - kx + 1;
                    if (ox < 0 || ox >= W) continue;
                    const float d = dp[oy * W + ox];
                    const float* wp = wl + (co * 9 + ky * 3 + kx) * C + vec * 8;
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = fmaf(d, wp[i], acc[i]);
                }
            }
        }
        store8(da + pix * C + vec * 8, acc);
    }
}

// ---- head weight (+bias) gradient: block = pixel range; thread = (pixel lane, 8-channel vector) ----------
template <typename T>
__global__ __launch_bounds__(256) void head_wgrad_kernel(const float* __restrict__ dout, const T* __restrict__ a,
                                                        float* __restrict__ part, int64_t npix, int cout, int H, int W,
                                                        int C, int64_t pix_per_blk) {
    __shared__ float red[256 * 8];
    const int tid = threadIdx.x;
    const int nvec = C >> 3, planes = 256 / nvec;
    const int vec = tid % nvec, pl = tid / nvec;
    const int64_t p0 = blockIdx.x * pix_per_blk;
    const int64_t p1 = min(p0 + pix_per_blk, npix);
    float* out = part + (int64_t)blockIdx.x * ((int64_t)cout * C * 9 + cout);
    for (int co = 0; co < cout; ++co) {
        float acc[9][8];
        float bsum = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[t][i] = 0.f;
        for (int64_t pix = p0 + pl; pix < p1; pix += planes) {
            const int ox = (int)(pix % W);
            const int64_t t2 = pix / W;
            const int oy = (int)(t2 % H);
            const int64_t b = t2 / H;
            const float d = dout[((b * cout + co) * H + oy) * (int64_t)W + ox];
            bsum += d;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy + ky - 1;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = ox + kx - 1;
                    if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                    float v[8];
                    load8(a + (((b * H + iy) * W) + ix) * (int64_t)C + vec * 8, v);
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[ky * 3 + kx][i] = fmaf(d, v[i], acc[ky * 3 + kx][i]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) red[tid * 8 + i] = acc[t][i];
            __syncthreads();
            if (tid < C) {
                const int vv = tid >> 3, i = tid & 7;
                float s = 0.f;
                for (int p = 0; p < planes; ++p) s += red[(p * nvec + vv) * 8 + i];
                out[((int64_t)co * C + tid) * 9 + t] = s;        // reference layout [cout][C][3][3]
            }
        }
        // bias partial: only vec == 0 lanes hold distinct pixels' d
        __syncthreads();
        red[tid] = vec == 0 ? bsum : 0.f;
        __syncthreads();
        if (tid == 0) {
            float s = 0.f;
            for (int i = 0; i < 256; ++i) s += red[i];
            out[(int64_t)cout * C * 9 + co] = s;
        }
    }
}

int small_blocks(int64_t npix) {
    int64_t nb = (npix + 1023) / 1024;
    if (nb < 1) nb = 1;
    if (nb > 1024) nb = 1024;
    return (int)nb;
}

bool small_shape_ok(int B, int cs, int H, int W, int C) {
    return B > 0 && cs >= 1 && cs <= kMaxSmall && H > 0 && W > 0 && (C == 128 || C == 256);
}

int grid_for(int64_t total) {
    int64_t g = (total + 255) / 256;
    return (int)(g < 8192 ? g : 8192);
}

}  // namespace

extern "C" int gmk_stem_wgrad_blocks(int64_t n_pixels) { return small_blocks(n_pixels); }
extern "C" int gmk_head_wgrad_blocks(int64_t n_pixels) { return small_blocks(n_pixels); }

extern "C" int gmk_stem_fwd(const float* x, const float* w, const float* bias, void* y, int B, int cin, int H, int W, int C,
                            int dtype, void* stream) {
    GMK_REQUIRE(x && w && bias && y, "gmk_stem_fwd: null pointer");
    GMK_REQUIRE(small_shape_ok(B, cin, H, W, C), "gmk_stem_fwd: unsupported shape B=%d cin=%d %dx%d C=%d", B, cin, H, W, C);
    const int64_t total = (int64_t)B * H * W * (C >> 3);
    const size_t lds = (size_t)(cin * 9 * C + C) * 4;
    if (dtype == GMK_BF16)
        stem_fwd_kernel<bf16_t><<<grid_for(total), 256, lds, gmk_stream(stream)>>>(x, w, bias, (bf16_t*)y, B, cin, H, W, C);
    else if (dtype == GMK_F32)
        stem_fwd_kernel<float><<<grid_for(total), 256, lds, gmk_stream(stream)>>>(x, w, bias, (float*)y, B, cin, H, W, C);
    else
        GMK_REQUIRE(false, "gmk_stem_fwd: bad dtype %d", dtype);
    return gmk_check_launch("gmk_stem_fwd");
}

extern "C" int gmk_stem_wgrad(const float* x, const void* dy, float* dw_part, int B, int cin, int H, int W, int C, int dtype,
                              void* stream) {
    GMK_REQUIRE(x && dy && dw_part, "gmk_stem_wgrad: null pointer");
    GMK_REQUIRE(small_shape_ok(B, cin, H, W, C), "gmk_stem_wgrad: unsupported shape");
    const int64_t npix = (int64_t)B * H * W;
    const int nb = small_blocks(npix);
    const int64_t ppb = (npix + nb - 1) / nb;
    if (dtype == GMK_BF16)
        stem_wgrad_kernel<bf16_t><<<nb, 256, 0, gmk_stream(stream)>>>(x, (const bf16_t*)dy, dw_part, npix, cin, H, W, C, ppb);
    else if (dtype == GMK_F32)
        stem_wgrad_kernel<float><<<nb, 256, 0, gmk_stream(stream)>>>(x, (const float*)dy, dw_part, npix, cin, H, W, C, ppb);
    else
        GMK_REQUIRE(false, "gmk_stem_wgrad: bad dtype %d", dtype);
    return gmk_check_launch("gmk_stem_wgrad");
}

extern "C" int gmk_head_fwd(const void* a, const float* w, const float* bias, float* out, int B, int cout, int H, int W,
                            int C, int dtype, void* stream) {
    GMK_REQUIRE(a && w && bias && out, "gmk_head_fwd: null pointer");
    GMK_REQUIRE(small_shape_ok(B, cout, H, W, C), "gmk_head_fwd: unsupported shape");
    const int64_t total = (int64_t)B * H * W * (C >> 3);
    const size_t lds = (size_t)cout * 9 * C * 4;
    if (dtype == GMK_BF16)
        head_fwd_kernel<bf16_t><<<grid_for(total), 256, lds, gmk_stream(stream)>>>((const bf16_t*)a, w, bias, out, B, cout, H,
                                                                                    W, C);
    else if (dtype == GMK_F32)
        head_fwd_kernel<float><<<grid_for(total), 256, lds, gmk_stream(stream)>>>((const float*)a, w, bias, out, B, cout, H, W,
                                                                                  C);
    else
        GMK_REQUIRE(false, "gmk_head_fwd: bad dtype %d", dtype);
    return gmk_check_launch("gmk_head_fwd");
}

extern "C" int gmk_head_dgrad(const float* dout, const float* w, void* da, int B, int cout, int H, int W, int C, int dtype,
                              void* stream) {
    GMK_REQUIRE(dout && w && da, "gmk_head_dgrad: null pointer");
    GMK_REQUIRE(small_shape_ok(B, cout, H, W, C), "gmk_head_dgrad: unsupported shape");
    const int64_t total = (int64_t)B * H * W * (C >> 3);
    const size_t lds = (size_t)cout * 9 * C * 4;
    if (dtype == GMK_BF16)
        head_dgrad_kernel<bf16_t><<<grid_for(total), 256, lds, gmk_stream(stream)>>>(dout, w, (bf16_t*)da, B, cout, H, W, C);
    else if (dtype == GMK_F32)
        head_dgrad_kernel<float><<<grid_for(total), 256, lds, gmk_stream(stream)>>>(dout, w, (float*)da, B, cout, H, W, C);
    else
        GMK_REQUIRE(false, "gmk_head_dgrad: bad dtype %d", dtype);
    return gmk_check_launch("gmk_head_dgrad");
}

extern "C" int gmk_head_wgrad(const float* dout, const void* a, float* dw_part, int B, int cout, int H, int W, int C,
                              int dtype, void* stream) {
    GMK_REQUIRE(dout && a && dw_part, "gmk_head_wgrad: null pointer");
    GMK_REQUIRE(small_shape_ok(B, cout, H, W, C), "gmk_head_wgrad: unsupported shape");
    const int64_t npix = (int64_t)B * H * W;
    const int nb = small_blocks(npix);
    const int64_t ppb = (npix + nb - 1) / nb;
    if (dtype == GMK_BF16)
        head_wgrad_kernel<bf16_t><<<nb, 256, 0, gmk_stream(stream)>>>(dout, (const bf16_t*)a, dw_part, npix, cout, H, W, C, ppb);
    else if (dtype == GMK_F32)
        head_wgrad_kernel<float><<<nb, 256, 0, gmk_stream(stream)>>>(dout, (const float*)a, dw_part, npix, cout, H, W, C, ppb);
    else
        GMK_REQUIRE(false, "gmk_head_wgrad: bad dtype %d", dtype);
    return gmk_check_launch("gmk_head_wgrad");
}
