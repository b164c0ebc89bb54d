// GroupNorm + SiLU forward/backward on NHWC activations, plus the small reductions that travel with them.
// Replaces nn.GroupNorm(32, C) -> nn.SiLU() pairs (reference gms/diffusion/simple_unet.py:39-40,161-162,169-170)
// and their autograd backward.  HBM-bound: one 256-thread workgroup per sample streams the sample twice
// (statistics, then apply); the second sweep is served from L2 (a 28x28x128 bf16 sample is 200 KB).
// Every thread owns one 8-channel vector (16 B of bf16) of a pixel, so loads are 16 B/lane and a wave covers
// whole 256-B pixel rows; reductions go per-thread -> LDS -> per-group.
#include <stdlib.h>

#include "gmk_common.h"

namespace {

constexpr int kThreads = 256;

// nn.Dropout(p) behind the SiLU (simple_unet.py:171): element e of the NHWC tensor is kept iff its Philox uniform
// u01(philox4x32(offset + e/4, seed)[e%4]) >= p — exactly gmk_rng_uniform(seed, offset) of the same shape, so the mask can be
// regenerated anywhere (the backward kernel does); kept values are scaled by 1/(1-p).  v: 8 consecutive channels from e0.
__device__ __forceinline__ void drop8(float (&v)[8], size_t e0, float p, uint64_t seed, uint64_t off) {
    const float scale = 1.0f / (1.0f - p);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        uint32_t r[4];
        philox4x32(off + (uint64_t)(e0 >> 2) + q, seed, r);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[4 * q + k] = u01(r[k]) >= p ? v[4 * q + k] * scale : 0.f;
    }
}

// A workgroup handles one (sample, channel slab).  Slabs narrower than the sample keep the set of lines the resident
// workgroups of an XCD are sweeping (32 CUs x a few workgroups x HW*CS*2 B) inside that XCD's 4 MiB L2, so the second
// sweep of the two-sweep kernels below is an L2 hit instead of a second trip to HBM.  Workgroup ids round-robin over the 8
// XCDs, so the slabs of one sample are given ids 8 apart: they run on the same XCD at about the same time and share
// the 128-B lines that straddle two slabs.
__device__ __forceinline__ void slab_of_block(int idx, int nslab, int B, int& b, int& slab) {
    if ((B & 7) == 0) {
        const int xcd = idx & 7, j = idx >> 3;
        slab = j % nslab;
        b = (j / nslab) * 8 + xcd;
    } else {
        b = idx / nslab;
        slab = idx % nslab;
    }
}

// NARROW: groups of 1 or 2 channels (hidden_size 32 / 64 zero-padded to the 128-channel tiles: GroupNorm(32, C) over the real channels is
// GroupNorm(128 / cpg) over the padded ones, the all-zero groups normalise to zero): per-channel instead of per-half-vector sums.
template <typename T, bool NARROW = false>
__global__ __launch_bounds__(kThreads) void gn_silu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              float* __restrict__ mean, float* __restrict__ rstd,
                                                              int HW, int C, int G, float eps,
                                                              const float* __restrict__ part, int TP, int ntiles,
                                                              int CS, int B, float drop_p, uint64_t drop_seed,
                                                              uint64_t drop_off, const float* __restrict__ xadd,
                                                              int xadd_stride, float* __restrict__ tab_sc = nullptr,
                                                              float* __restrict__ tab_sh = nullptr, int tab_stride = 0) {
    __shared__ float red[kThreads * (NARROW ? 16 : 4)];
    __shared__ float smean[NARROW ? 256 : 64], srstd[NARROW ? 256 : 64];
    const int tid = threadIdx.x;
    int b, slab;
    slab_of_block(blockIdx.x, C / CS, B, b, slab);
    const int nvec = CS >> 3, planes = kThreads / nvec;
    const int vec = tid % nvec, pl = tid / nvec;
    // Gq < 0 (whole-sample launches only): -Gq channels per group, any size up to 16 (widths that are not a power of two times 32: GroupNorm(32, 96)
    // has 3 channels per group), ceil(C / cpg) groups - the last one may be partial and lies in the zero padding
    const int Gq = G;
    const int cpg = Gq < 0 ? -Gq : C / Gq;
    G = Gq < 0 ? (C + cpg - 1) / cpg : Gq;
    const int c0 = slab * CS, g0 = c0 / cpg, gps = (CS + cpg - 1) / cpg;
    const T* xb = x + (size_t)b * HW * C + c0 + vec * 8;
    T* yb = y + (size_t)b * HW * C + c0 + vec * 8;
    float ea[8];           // per-(sample, channel) addend applied to x on load (conv bias + embedding broadcast of the producer)
#pragma unroll
    for (int i = 0; i < 8; ++i) ea[i] = xadd ? xadd[(size_t)b * xadd_stride + c0 + vec * 8 + i] : 0.f;

    if (part) {
        // statistics from the producer's partial sums: [tile][8 pixel groups][2 sample slots][C/4 units][sum, sumsq],
        // added in a fixed order (deterministic)
        if (tid < G) {
            const int units = cpg >> 2, G4 = C >> 2;
            const int m_beg = b * HW, m_end = m_beg + HW;
            const int t0 = m_beg / TP, t1 = min((m_end - 1) / TP, ntiles - 1);
            float s = 0.f, q = 0.f;
            for (int t = t0; t <= t1; ++t)
                for (int pg = 0; pg < 8; ++pg) {
                    const int g_beg = t * TP + pg * 32;
                    const int g_end = min(g_beg + 32, t * TP + TP);
                    if (g_beg >= g_end || g_end <= m_beg || g_beg >= m_end) continue;
                    const int slot = b - g_beg / HW;              // 0, or 1 when the group started in the previous sample
                    const float* pp = part + ((size_t)((t * 8 + pg) * 2 + slot) * G4 + tid * units) * 2;
                    for (int u = 0; u < units; ++u) { s += pp[2 * u]; q += pp[2 * u + 1]; }
                }
            const float n = (float)cpg * (float)HW;
            const float m = s / n;
            const float var = fmaxf(q / n - m * m, 0.f);
            const float r = 1.0f / sqrtf(var + eps);
            smean[tid] = m; srstd[tid] = r;
            mean[b * G + tid] = m; rstd[b * G + tid] = r;
        }
    } else if constexpr (NARROW) {
        float s[8], q[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { s[i] = 0.f; q[i] = 0.f; }
#pragma unroll 4
        for (int p = pl; p < HW; p += planes) {
            float v[8];
            load8(xb + (size_t)p * C, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[i] += ea[i]; s[i] += v[i]; q[i] = fmaf(v[i], v[i], q[i]); }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { red[tid * 16 + i] = s[i]; red[tid * 16 + 8 + i] = q[i]; }
        __syncthreads();
        float cs = 0.f, cq = 0.f;
        if (tid < CS) {
            const int vv = tid >> 3, i = tid & 7;
            for (int p = 0; p < planes; ++p) { cs += red[(p * nvec + vv) * 16 + i]; cq += red[(p * nvec + vv) * 16 + 8 + i]; }
        }
        __syncthreads();
        if (tid < CS) { red[tid] = cs; red[256 + tid] = cq; }
        __syncthreads();
        if (tid < gps) {
            float ss = 0.f, qq = 0.f;
            for (int j = 0; j < cpg && tid * cpg + j < CS; ++j) { ss += red[tid * cpg + j]; qq += red[256 + tid * cpg + j]; }
            const float n = (float)cpg * (float)HW;
            const float m = ss / n;
            const float var = fmaxf(qq / n - m * m, 0.f);
            const float r = 1.0f / sqrtf(var + eps);
            smean[tid] = m; srstd[tid] = r;
            mean[b * G + g0 + tid] = m; rstd[b * G + g0 + tid] = r;
        }
    } else {
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll 4
    for (int p = pl; p < HW; p += planes) {
        float v[8];
        load8(xb + (size_t)p * C, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += ea[i];
        s0 += (v[0] + v[1]) + (v[2] + v[3]);
        q0 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        s1 += (v[4] + v[5]) + (v[6] + v[7]);
        q1 += (v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]);
    }
    red[tid * 4 + 0] = s0; red[tid * 4 + 1] = q0; red[tid * 4 + 2] = s1; red[tid * 4 + 3] = q1;
    __syncthreads();
    if (tid < gps) {
        const int hv_per_g = cpg >> 2;   // 4-channel half-vectors per group
        float s = 0.f, q = 0.f;
        for (int j = 0; j < hv_per_g; ++j) {
            const int hv = tid * hv_per_g + j, vv = hv >> 1, hf = hv & 1;
            for (int p = 0; p < planes; ++p) {
                s += red[(p * nvec + vv) * 4 + hf * 2];
                q += red[(p * nvec + vv) * 4 + hf * 2 + 1];
            }
        }
        const float n = (float)cpg * (float)HW;
        const float m = s / n;
        const float var = fmaxf(q / n - m * m, 0.f);
        const float r = 1.0f / sqrtf(var + eps);
        smean[tid] = m; srstd[tid] = r;
        mean[b * G + g0 + tid] = m; rstd[b * G + g0 + tid] = r;
    }
    }
    __syncthreads();
    float sc[8], sh[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int cl = vec * 8 + i, g = cl / cpg, c = c0 + cl;
        sc[i] = srstd[g] * gamma[c];
        sh[i] = fmaf(ea[i], sc[i], beta[c] - smean[g] * sc[i]);      // the addend folded into the shift: (x + ea - mean) * sc + beta
    }
    if (tab_sc && pl == 0) {        // y = silu(x * sc + sh) for whoever applies the normalisation itself (gmk_gn_stats)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            tab_sc[(size_t)b * tab_stride + c0 + vec * 8 + i] = sc[i];
            tab_sh[(size_t)b * tab_stride + c0 + vec * 8 + i] = sh[i];
        }
    }
    if (!y) return;
#pragma unroll 4
    for (int p = pl; p < HW; p += planes) {
        float v[8];
        load8(xb + (size_t)p * C, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = siluf_(fmaf(v[i], sc[i], sh[i]));
        if (drop_p > 0.f) drop8(v, ((size_t)b * HW + p) * C + c0 + vec * 8, drop_p, drop_seed, drop_off);
        store8(yb + (size_t)p * C, v);
    }
}

template <typename T, typename TX = T>      // T: gradient tensors (dy, addends, dx); TX: the saved forward activation x
__global__ __launch_bounds__(kThreads) void gn_silu_bwd_kernel(
    const T* __restrict__ dy, const TX* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ mean, const float* __restrict__ rstd, const T* __restrict__ dadd1,
    const T* __restrict__ dadd2, T* __restrict__ dx, float* __restrict__ dgp, float* __restrict__ dbp,
    float* __restrict__ dxsum, int dxsum_stride, int HW, int C, int G, int CS, int B, float drop_p, uint64_t drop_seed,
    uint64_t drop_off, const float* __restrict__ xadd, int xadd_stride) {
    __shared__ float red[kThreads * 16];
    __shared__ float chg[256], chb[256];
    __shared__ float sA[256], sB[256];          // (up to 256 one-channel groups: the zero-padded narrow widths)
    const int tid = threadIdx.x;
    int b, slab;
    slab_of_block(blockIdx.x, C / CS, B, b, slab);
    const int nvec = CS >> 3, planes = kThreads / nvec;
    const int vec = tid % nvec, pl = tid / nvec;
    const int Gq = G;                              // < 0: -Gq channels per group (see gn_silu_fwd_kernel)
    const int cpg = Gq < 0 ? -Gq : C / Gq;
    G = Gq < 0 ? (C + cpg - 1) / cpg : Gq;
    const int c0 = slab * CS, g0 = c0 / cpg, gps = (CS + cpg - 1) / cpg;
    const size_t base = (size_t)b * HW * C + c0 + vec * 8;

    float gam[8], bet[8], mu[8], rs[8];
    int grp[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int cl = vec * 8 + i, c = c0 + cl;
        grp[i] = cl / cpg;
        gam[i] = gamma[c]; bet[i] = beta[c];
        // x enters as x + xadd[b][c]: fold the addend into the mean that is subtracted
        mu[i] = mean[b * G + g0 + grp[i]] - (xadd ? xadd[(size_t)b * xadd_stride + c] : 0.f);
        rs[i] = rstd[b * G + g0 + grp[i]];
    }
    float ag[8], ab[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { ag[i] = 0.f; ab[i] = 0.f; }
#pragma unroll 2
    for (int p = pl; p < HW; p += planes) {
        float xv[8], dv[8];
        load8(x + base + (size_t)p * C, xv);
        load8(dy + base + (size_t)p * C, dv);
        if (drop_p > 0.f) drop8(dv, base + (size_t)p * C, drop_p, drop_seed, drop_off);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float xh = (xv[i] - mu[i]) * rs[i];
            const float g = fmaf(xh, gam[i], bet[i]);
            const float s = sigmoidf_(g);
            const float dg = dv[i] * s * (1.f + g * (1.f - s));
            ag[i] = fmaf(dg, xh, ag[i]);
            ab[i] += dg;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[tid * 16 + i] = ag[i]; red[tid * 16 + 8 + i] = ab[i]; }
    __syncthreads();
    if (tid < CS) {
        const int vv = tid >> 3, i = tid & 7;
        float a = 0.f, bb = 0.f;
        for (int p = 0; p < planes; ++p) {
            a += red[(p * nvec + vv) * 16 + i];
            bb += red[(p * nvec + vv) * 16 + 8 + i];
        }
        chg[tid] = a; chb[tid] = bb;
        dgp[(size_t)b * C + c0 + tid] = a;
        dbp[(size_t)b * C + c0 + tid] = bb;
    }
    __syncthreads();
    if (tid < gps) {
        float A = 0.f, Bq = 0.f;
        for (int j = 0; j < cpg && tid * cpg + j < CS; ++j) {
            const int cl = tid * cpg + j;
            A = fmaf(gamma[c0 + cl], chb[cl], A);
            Bq = fmaf(gamma[c0 + cl], chg[cl], Bq);
        }
        const float inv_n = 1.f / ((float)cpg * (float)HW);
        sA[tid] = A * inv_n; sB[tid] = Bq * inv_n;
    }
    __syncthreads();
    float cA[8], cB[8], xs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { cA[i] = sA[grp[i]]; cB[i] = sB[grp[i]]; xs[i] = 0.f; }
#pragma unroll 2
    for (int p = pl; p < HW; p += planes) {
        float xv[8], dv[8], o[8];
        load8(x + base + (size_t)p * C, xv);
        load8(dy + base + (size_t)p * C, dv);
        if (drop_p > 0.f) drop8(dv, base + (size_t)p * C, drop_p, drop_seed, drop_off);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float xh = (xv[i] - mu[i]) * rs[i];
            const float g = fmaf(xh, gam[i], bet[i]);
            const float s = sigmoidf_(g);
            const float dg = dv[i] * s * (1.f + g * (1.f - s));
            o[i] = rs[i] * (dg * gam[i] - cA[i] - xh * cB[i]);
        }
        if (dadd1) {
            float t[8];
            load8(dadd1 + base + (size_t)p * C, t);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] += t[i];
        }
        if (dadd2) {
            float t[8];
            load8(dadd2 + base + (size_t)p * C, t);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] += t[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) xs[i] += o[i];
        store8(dx + base + (size_t)p * C, o);
    }
    if (dxsum) {
        __syncthreads();   // red is re-used
#pragma unroll
        for (int i = 0; i < 8; ++i) red[tid * 16 + i] = xs[i];
        __syncthreads();
        if (tid < CS) {
            const int vv = tid >> 3, i = tid & 7;
            float a = 0.f;
            for (int p = 0; p < planes; ++p) a += red[(p * nvec + vv) * 16 + i];
            dxsum[(size_t)b * dxsum_stride + c0 + tid] = a;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Register-resident variants (bf16): one workgroup = one (sample, 32-channel slab); every thread loads ALL its pixels'
// 16-byte vectors up front (ITER independent loads in flight per thread), the statistics are reduced over the workgroup,
// and the apply sweep runs from registers: the tensor is read from HBM exactly once (the streaming kernels read it twice;
// PMC shows their second sweep is NOT served by L2).  Thread = (vec = tid & 3, plane = tid >> 2); pixel p = plane + i*planes.
// ------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;


template <typename T>
__device__ __forceinline__ void unpack8(const u32x4_t r, float (&v)[8]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) unpack_pair<T>(r[k], v[2 * k], v[2 * k + 1]);
}

// sum over the lanes of a wave that share (lane & (NVEC-1)); every such lane ends with the total
template <int NVEC>
__device__ __forceinline__ float vec_lane_sum(float v) { return lanes_sum_from<NVEC>(v); }

template <typename T, int ITER, int NVEC>       // T: bf16_t or f16_t (input and output)
__global__ __launch_bounds__(1024) void gn_silu_fwd_reg_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ mean, float* __restrict__ rstd, int HW, int C,
                                                             int G, float eps, int B, int planes, float drop_p,
                                                             uint64_t drop_seed, uint64_t drop_off, const float* __restrict__ xadd,
                                                             int xadd_stride, float* __restrict__ tab_sc = nullptr,
                                                             float* __restrict__ tab_sh = nullptr, int tab_stride = 0) {
    __shared__ float red[16][NVEC][4];      // [wave][vec][s0, q0, s1, q1]
    __shared__ float smean[16], srstd[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    constexpr int CS = NVEC * 8, LOGV = NVEC == 8 ? 3 : 2;
    const int vec = tid & (NVEC - 1), pl = tid >> LOGV;
    int b, slab;
    slab_of_block(blockIdx.x, C / CS, B, b, slab);
    const int cpg = C / G, c0 = slab * CS, g0 = c0 / cpg, gps = CS / cpg;
    const size_t base = (size_t)b * HW * C + c0 + vec * 8;
    u32x4_t raw[ITER];
    bool ok[ITER];
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        const int p = pl + i * planes;
        ok[i] = pl < planes && p < HW;
        raw[i] = ok[i] ? *reinterpret_cast<const u32x4_t*>(x + base + (size_t)p * C) : u32x4_t{0u, 0u, 0u, 0u};
    }
    float ea[8];           // per-(sample, channel) addend applied to x on load (conv bias + embedding broadcast of the producer)
#pragma unroll
    for (int k = 0; k < 8; ++k) ea[k] = xadd ? xadd[(size_t)b * xadd_stride + c0 + vec * 8 + k] : 0.f;
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        float v[8];
        unpack8<T>(raw[i], v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = ok[i] ? v[k] + ea[k] : 0.f;
        s0 += (v[0] + v[1]) + (v[2] + v[3]);
        q0 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        s1 += (v[4] + v[5]) + (v[6] + v[7]);
        q1 += (v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]);
    }
    // keep the pixels PACKED across the reduction (the apply sweep unpacks again): otherwise the compiler carries all 8*ITER
    // floats through the barrier and the register count halves the occupancy
#pragma unroll
    for (int i = 0; i < ITER; ++i) asm volatile("" : "+v"(raw[i]));
    s0 = vec_lane_sum<NVEC>(s0); q0 = vec_lane_sum<NVEC>(q0); s1 = vec_lane_sum<NVEC>(s1); q1 = vec_lane_sum<NVEC>(q1);
    if (lane < NVEC) { red[wave][lane][0] = s0; red[wave][lane][1] = q0; red[wave][lane][2] = s1; red[wave][lane][3] = q1; }
    __syncthreads();
    if (tid < gps) {
        const int hv_per_g = cpg >> 2;          // 4-channel half-vectors per group
        float s = 0.f, q = 0.f;
        for (int j = 0; j < hv_per_g; ++j) {
            const int hv = tid * hv_per_g + j, vv = hv >> 1, hf = hv & 1;
            for (int w = 0; w < nwaves; ++w) { s += red[w][vv][hf * 2]; q += red[w][vv][hf * 2 + 1]; }
        }
        const float n = (float)cpg * (float)HW;
        const float m = s / n;
        const float var = fmaxf(q / n - m * m, 0.f);
        const float r = 1.0f / sqrtf(var + eps);
        smean[tid] = m; srstd[tid] = r;
        mean[b * G + g0 + tid] = m; rstd[b * G + g0 + tid] = r;
    }
    __syncthreads();
    float sc[8], sh[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int cl = vec * 8 + i, g = cl / cpg, c = c0 + cl;
        sc[i] = srstd[g] * gamma[c];
        sh[i] = beta[c] - (smean[g] - ea[i]) * sc[i];          // (x + ea - mean) * sc + beta
    }
    if (tab_sc && pl == 0) {        // y = silu(x * sc + sh) for whoever applies the normalisation itself (gmk_gn_stats)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            tab_sc[(size_t)b * tab_stride + c0 + vec * 8 + i] = sc[i];
            tab_sh[(size_t)b * tab_stride + c0 + vec * 8 + i] = sh[i];
        }
    }
    if (!y) return;
    // one running element offset (a 64-bit add per pixel) instead of ITER pixel indices kept alive through the whole kernel: at ITER = 16
    // those cost 16 VGPRs, which put the fp16 instantiation at 138 registers - one workgroup per CU instead of two
    size_t eoff = base + (size_t)pl * C;
    const size_t estride = (size_t)planes * C;
#pragma unroll
    for (int i = 0; i < ITER; ++i, eoff += estride) {
        asm volatile("" : "+v"(eoff));
        if (!ok[i]) continue;
        float v[8];
        unpack8<T>(raw[i], v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = siluf_(fmaf(v[k], sc[k], sh[k]));
        if (drop_p > 0.f) drop8(v, eoff, drop_p, drop_seed, drop_off);
        store8(y + eoff, v);
    }
}

// (A backward kernel with the same residency — x and dy packed in registers, both sweeps from them — was built and measured
// 30-60 % SLOWER than the streaming backward: it needs 188 VGPRs at 8 pixels per thread (one workgroup per CU) or spills at
// 128, and it recomputes the sigmoid in both sweeps.  Removed; the streaming kernel with 32-channel slabs stays.)

typedef float f32x2_t __attribute__((ext_vector_type(2)));
// the two 16-bit values of one dword as a float pair (low half first)
template <typename T>
__device__ __forceinline__ f32x2_t unpack2(unsigned r) {
    f32x2_t v;
    float lo, hi;
    unpack_pair<T>(r, lo, hi);
    v[0] = lo; v[1] = hi;
    return v;
}
// d/dg [g * sigmoid(g)] * dy for a channel pair, g = fma(x, s, t): 6 packed ops + 2 x (v_exp_f32, v_rcp_f32)
__device__ __forceinline__ f32x2_t silu_grad2(f32x2_t x, f32x2_t dy, f32x2_t s, f32x2_t t, float neg_log2e) {
    const f32x2_t g = __builtin_elementwise_fma(x, s, t);
    const f32x2_t g2 = g * neg_log2e;
    const f32x2_t den = f32x2_t{__builtin_amdgcn_exp2f(g2[0]), __builtin_amdgcn_exp2f(g2[1])} + 1.0f;
    const f32x2_t sg = f32x2_t{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    const f32x2_t u = dy * sg;
    return __builtin_elementwise_fma(u, __builtin_elementwise_fma(-g, sg, g), u);
}

// Backward with the tensor read once: one workgroup (256 threads) = one (sample, 32-channel slab); dy stays PACKED in
// registers (ITER x 16 B per thread), x is parked in LDS (HW x 64 B) on its way through the first sweep, and the second sweep
// reads x from LDS and dy from registers.  HBM: x, dy, addends once + dx once (the streaming kernel reads x and dy twice).
// 3 workgroups per CU (LDS), thread = (vec = tid & 3, plane = tid >> 2), pixel p = plane + 64 i.
// NVEC = 2 (16-channel slabs, 1024 threads, x in 128 KiB of LDS) carries the same scheme to 64 x 64 images, where a 32-channel slab
// fits neither the registers nor the LDS and the two-sweep streaming kernel used to run (5 passes over HBM instead of 3).
// KEEP: iterations whose dy vector stays in registers between the sweeps (default: all; the others are read again in the second sweep).
// XREG: the last XREG iterations keep their x in registers instead of LDS.  EXACT: HW == ITER * PL, no pixel masks, and both sweeps are
// software-pipelined (the loads of the next pixel chunk are in flight under the arithmetic of the current one).
// The 64 x 64 form on 32-channel slabs is <32, 512, 4, 28, 13>: 512 KiB of x + dy per slab against 160 KiB of LDS, so ONE 512-thread
// workgroup per CU at 256 registers per lane holds 19 / 32 of x in LDS, 13 / 32 of x and 28 / 32 of dy in registers and reads an eighth
// of dy twice: 3.125 passes on 64-byte segments per pixel row (the 16-channel form: 3 passes on 32-byte segments, 11 - 25 % slower).
// With two waves per SIMD this form is VALU-bound in its second sweep (exp + rcp are quarter rate: 8 of ~ 17 issue slots per element),
// which is why the gradient addends are almost free there.
template <typename TX, int ITER, int THREADS, int NVEC = 4, int KEEP = ITER, int XREG = 0, bool EXACT = (XREG > 0)>      // TX: storage type of the saved activation x (bf16_t / f16_t); gradients are bf16
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(XREG > 0 ? 2 : THREADS == 256 ? 3 : 4, XREG > 0 ? 2 : 4))) void gn_silu_bwd_hybrid_kernel(
    const bf16_t* __restrict__ dy, const TX* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ mean, const float* __restrict__ rstd, const bf16_t* __restrict__ dadd1,
    const bf16_t* __restrict__ dadd2, bf16_t* __restrict__ dx, float* __restrict__ dgp, float* __restrict__ dbp,
    float* __restrict__ dxsum, int dxsum_stride, int HW, int C, int G, int B, float drop_p, uint64_t drop_seed,
    uint64_t drop_off, const float* __restrict__ xadd, int xadd_stride) {
    constexpr int CS = NVEC * 8;
    constexpr int XLDS = ITER - XREG;                                    // iterations whose x is parked in LDS
    constexpr int CH = ITER >= 32 ? 4 : 2;                               // pixels per load chunk of the first sweep
    constexpr bool kExact = EXACT;                                       // HW == ITER * PL (the host checks): no pixel masks, pipelined sweeps
    extern __shared__ __attribute__((aligned(16))) char xs_lds[];        // [min(HW, XLDS * PL)][NVEC] x 16 B
    constexpr int NW = THREADS / 64, PL = THREADS / NVEC;      // waves, pixel planes
    __shared__ float red[NW][NVEC][16];
    __shared__ float chg[CS], chb[CS];
    __shared__ float sA[8], sB[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int vec = tid & (NVEC - 1), pl = tid / NVEC;
    int b, slab;
    slab_of_block(blockIdx.x, C / CS, B, b, slab);
    const int cpg = C / G, c0 = slab * CS, g0 = c0 / cpg, gps = CS / cpg;
    const size_t base = (size_t)b * HW * C + c0 + vec * 8;
    const int glo = (vec * 8) / cpg, ghi = (vec * 8 + 4) / cpg;
    const float rs0 = rstd[b * G + g0 + glo], rs1 = rstd[b * G + g0 + ghi];
    // Per-channel constants that turn both sweeps into chains of (packed) FMAs:
    //   g  = gamma * (x - mu) * rstd + beta        = fma(x, s, t)          s = rstd * gamma, t = beta - mu * s
    //   dg = dy * sig(g) * (1 + g * (1 - sig(g)))  = fma(u, fma(-g, sg, g), u),  u = dy * sg
    //   sum_x dg * xhat                            = rstd * (sum dg * x - mu * sum dg)   (accumulated as sum dg * x)
    //   dx = rstd * (dg * gamma - cA - xhat * cB)  = fma(dg, s, fma(x, nP, q)),   nP = -rstd^2 cB, q = mu rstd^2 cB - rstd cA
    // the channel pairs (2j, 2j+1) of a thread's 16-byte vector are carried as float2 so the compiler emits v_pk_*_f32.
    f32x2_t sv[4], tv[4], muv[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int k = 2 * jj + e, c = c0 + vec * 8 + k;
            // x enters as x + xadd[b][c]: fold the addend into the mean that is subtracted
            const float m = mean[b * G + g0 + (k < 4 ? glo : ghi)] - (xadd ? xadd[(size_t)b * xadd_stride + c] : 0.f);
            const float sc = (k < 4 ? rs0 : rs1) * gamma[c];
            muv[jj][e] = m; sv[jj][e] = sc; tv[jj][e] = beta[c] - m * sc;
        }
    }
    constexpr float kNegLog2e = -1.4426950408889634f;
    u32x4_t dr[KEEP];
    u32x4_t xk[XREG > 0 ? XREG : 1];
    f32x2_t ax[4], ab[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) { ax[jj] = f32x2_t{0.f, 0.f}; ab[jj] = f32x2_t{0.f, 0.f}; }
    // chunks of CH pixels: 2 CH independent 16-B loads in flight per thread, then their arithmetic; the fence keeps the
    // scheduler from hoisting every load of the sweep to the top (26 x 4 VGPRs)
    // (running offsets, opaque to the optimiser: with the sweep unrolled it otherwise materialises every iteration's 64-bit address and
    // LDS address up front - 100+ spilled registers in the 16-iteration form)
    size_t eoff = base + (size_t)pl * C;
    const size_t estride = (size_t)PL * C;
    uint32_t loff = ((uint32_t)pl * NVEC + vec) * 16;
    // kPipe: the loads of chunk c + 1 are issued before the arithmetic of chunk c (the 512-thread form has one workgroup per CU and
    // nothing else to hide the load latency behind)
    constexpr int NCH = (ITER + CH - 1) / CH, kPipe = kExact ? 1 : 0;
    u32x4_t xr[2][CH], dq[2][CH];
#pragma unroll
    for (int c = 0; c < NCH + kPipe; ++c) {
        if (c < NCH) {
            if (ITER > 8) asm volatile("" : "+v"(eoff));
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int i = c * CH + u;
                if (i >= ITER) continue;
                const int p = pl + PL * i;
                const bool ok = kExact || p < HW;
                xr[c & 1][u] = ok ? *reinterpret_cast<const u32x4_t*>(x + eoff + u * estride) : u32x4_t{0u, 0u, 0u, 0u};
                dq[c & 1][u] = ok ? *reinterpret_cast<const u32x4_t*>(dy + eoff + u * estride) : u32x4_t{0u, 0u, 0u, 0u};     // dy = 0 masks the pixel
            }
            eoff += CH * estride;
            if (kPipe) __builtin_amdgcn_sched_barrier(0);
        }
        if (c >= kPipe) {
            const int cc = c - kPipe;
            if (ITER > 8) asm volatile("" : "+v"(loff));
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int i = cc * CH + u;
                if (i >= ITER) continue;
                const int p = pl + PL * i;
                const u32x4_t xq = xr[cc & 1][u], dv4 = dq[cc & 1][u];
                if (i < KEEP) dr[i < KEEP ? i : 0] = dv4;
                if (i >= XLDS) xk[i >= XLDS ? i - XLDS : 0] = xq;
                else if (kExact || p < HW) *reinterpret_cast<u32x4_t*>(xs_lds + loff + u * (PL * NVEC * 16)) = xq;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const f32x2_t xv = unpack2<TX>(xq[jj]), dv = unpack2<bf16_t>(dv4[jj]);
                    const f32x2_t dg = silu_grad2(xv, dv, sv[jj], tv[jj], kNegLog2e);
                    ax[jj] = __builtin_elementwise_fma(dg, xv, ax[jj]);
                    ab[jj] += dg;
                }
            }
            loff += CH * PL * NVEC * 16;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int i = 0; i < KEEP; ++i) asm volatile("" : "+v"(dr[i]));          // stay packed across the reduction
#pragma unroll
    for (int i = 0; i < XREG; ++i) asm volatile("" : "+v"(xk[i]));
    float ag[8], abk[8];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int k = 2 * jj + e;
            abk[k] = ab[jj][e];
            ag[k] = (ax[jj][e] - muv[jj][e] * ab[jj][e]) * (k < 4 ? rs0 : rs1);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { ag[k] = vec_lane_sum<NVEC>(ag[k]); abk[k] = vec_lane_sum<NVEC>(abk[k]); }
    if (lane < NVEC) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { red[wave][lane][k] = ag[k]; red[wave][lane][8 + k] = abk[k]; }
    }
    __syncthreads();
    if (tid < CS) {
        const int vv = tid >> 3, k = tid & 7;
        float a = 0.f, bb = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { a += red[w][vv][k]; bb += red[w][vv][8 + k]; }
        chg[tid] = a; chb[tid] = bb;
        dgp[(size_t)b * C + c0 + tid] = a;
        dbp[(size_t)b * C + c0 + tid] = bb;
    }
    __syncthreads();
    if (tid < gps) {
        float A = 0.f, Bq = 0.f;
        for (int j = 0; j < cpg; ++j) {
            const int cl = tid * cpg + j;
            A = fmaf(gamma[c0 + cl], chb[cl], A);
            Bq = fmaf(gamma[c0 + cl], chg[cl], Bq);
        }
        const float inv_n = 1.f / ((float)cpg * (float)HW);
        sA[tid] = A * inv_n; sB[tid] = Bq * inv_n;
    }
    __syncthreads();
    f32x2_t npv[4], qv[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const float rsk = jj < 2 ? rs0 : rs1, cA = jj < 2 ? sA[glo] : sA[ghi], cB = jj < 2 ? sB[glo] : sB[ghi];
        const float r2b = rsk * rsk * cB;
        npv[jj] = f32x2_t{-r2b, -r2b};
        qv[jj] = muv[jj] * r2b - rsk * cA;
    }
    float xs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) xs[k] = 0.f;
    eoff = base + (size_t)pl * C;
    loff = ((uint32_t)pl * NVEC + vec) * 16;
    if constexpr (kExact) {
        // pipelined like the first sweep: the global loads of pixel pair c + 1 (gradient addends, the part of dy that was not kept) are
        // issued before the arithmetic and the stores of pair c
        constexpr int N2 = ITER / 2;
        u32x4_t a1[2][2], a2[2][2], dq2[2][2];
        size_t lo = eoff;
#pragma unroll
        for (int c = 0; c < N2 + 1; ++c) {
            if (c < N2) {
                asm volatile("" : "+v"(lo));
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = 2 * c + u;
                    if (i >= KEEP) dq2[c & 1][u] = *reinterpret_cast<const u32x4_t*>(dy + lo + u * estride);
                    if (dadd1) a1[c & 1][u] = *reinterpret_cast<const u32x4_t*>(dadd1 + lo + u * estride);
                    if (dadd2) a2[c & 1][u] = *reinterpret_cast<const u32x4_t*>(dadd2 + lo + u * estride);
                }
                lo += 2 * estride;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (c >= 1) {
                const int cc = c - 1;
                asm volatile("" : "+v"(eoff), "+v"(loff));
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = 2 * cc + u;
                    const u32x4_t xq = i >= XLDS ? xk[i >= XLDS ? i - XLDS : 0]
                                                 : *reinterpret_cast<const u32x4_t*>(xs_lds + loff + u * (PL * NVEC * 16));
                    const u32x4_t dv4 = i < KEEP ? dr[i < KEEP ? i : 0] : dq2[cc & 1][u];
                    u32x4_t ow;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const f32x2_t xv = unpack2<TX>(xq[jj]), dv = unpack2<bf16_t>(dv4[jj]);
                        const f32x2_t dg = silu_grad2(xv, dv, sv[jj], tv[jj], kNegLog2e);
                        f32x2_t ov = __builtin_elementwise_fma(dg, sv[jj], __builtin_elementwise_fma(xv, npv[jj], qv[jj]));
                        if (dadd1) ov += unpack2<bf16_t>(a1[cc & 1][u][jj]);
                        if (dadd2) ov += unpack2<bf16_t>(a2[cc & 1][u][jj]);
                        xs[2 * jj] += ov[0]; xs[2 * jj + 1] += ov[1];
                        ow[jj] = pack_pair<bf16_t>(ov[0], ov[1]);
                    }
                    *reinterpret_cast<u32x4_t*>(dx + eoff + u * estride) = ow;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(xs[k]));    // (or the adds are sunk into the `if (dxsum)` below)
                eoff += 2 * estride; loff += 2 * PL * NVEC * 16;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else
#pragma unroll
    for (int i = 0; i < ITER; ++i, eoff += estride, loff += PL * NVEC * 16) {
        if (ITER > 8) asm volatile("" : "+v"(eoff), "+v"(loff));
        const int p = pl + PL * i;
        if (!kExact && p >= HW) continue;
        const size_t off = eoff;
        const u32x4_t xr = i >= XLDS ? xk[i >= XLDS ? i - XLDS : 0] : *reinterpret_cast<const u32x4_t*>(xs_lds + loff);
        const u32x4_t dq2 = i < KEEP ? dr[i < KEEP ? i : 0] : *reinterpret_cast<const u32x4_t*>(dy + off);
        float o[8];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const f32x2_t xv = unpack2<TX>(xr[jj]), dv = unpack2<bf16_t>(dq2[jj]);
            // (recomputing the sigmoid here instead of caching dy * silu'(g) costs 3-4 % of the kernel: a timing-only build without it)
            const f32x2_t dg = silu_grad2(xv, dv, sv[jj], tv[jj], kNegLog2e);
            const f32x2_t ov = __builtin_elementwise_fma(dg, sv[jj], __builtin_elementwise_fma(xv, npv[jj], qv[jj]));
            o[2 * jj] = ov[0]; o[2 * jj + 1] = ov[1];
        }
        if (dadd1) {
            float t[8];
            load8(dadd1 + off, t);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] += t[k];
        }
        if (dadd2) {
            float t[8];
            load8(dadd2 + off, t);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] += t[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) xs[k] += o[k];
        store8(dx + off, o);
        if (i & 1) __builtin_amdgcn_sched_barrier(0);
    }
    if (dxsum) {
#pragma unroll
        for (int k = 0; k < 8; ++k) xs[k] = vec_lane_sum<NVEC>(xs[k]);
        __syncthreads();   // red is re-used
        if (lane < NVEC) {
#pragma unroll
            for (int k = 0; k < 8; ++k) red[wave][lane][k] = xs[k];
        }
        __syncthreads();
        if (tid < CS) {
            const int vv = tid >> 3, k = tid & 7;
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) a += red[w][vv][k];
            dxsum[(size_t)b * dxsum_stride + c0 + tid] = a;
        }
    }
}

// pixels per thread of the register-resident kernels: the smallest of 1, 2, 4, 8 that fits the slab into <= 512 threads
int gn_reg_iter(int HW, int nvec) {
    // 28 x 28 (784 pixels; round 4): 16 pixels per thread leave 49 planes x 8 = 392 of 448 threads busy (4.05 TB/s against 4.65 at 32 x 32, same batch);
    // 14 per thread make it 56 planes = 448 threads, all busy, on 8 fewer data registers (GMK_GN_KERNEL=6 keeps the power-of-two choice: A/B)
    if (nvec == 8 && HW == 784 && gmk_kernel_choice(2, "GMK_GN_KERNEL") != 6) return 14;
    for (int it = 1; it <= 16; it <<= 1)
        if (((HW + it - 1) / it) * nvec <= 512) return it;
    return 0;
}

// (LDS-resident single-read variants of both kernels - the sample slice pulled into LDS by LDS-DMA - were built and measured 0-50 %
// slower than the streaming kernels at 28x28 / 14x14 / 7x7: one or few resident workgroups per CU serialise their load / compute /
// store phases.  Removed in round 2; the register / hybrid kernels above are the single-read forms that pay.)

template <typename T>
__global__ __launch_bounds__(kThreads) void chansum_kernel(const T* __restrict__ x, float* __restrict__ out,
                                                          int out_stride, int HW, int C) {
    __shared__ float red[kThreads * 8];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nvec = C >> 3, planes = kThreads / nvec;
    const int vec = tid % nvec, pl = tid / nvec;
    const T* xb = x + (size_t)b * HW * C + vec * 8;
    float s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = 0.f;
    for (int p = pl; p < HW; p += planes) {
        float v[8];
        load8(xb + (size_t)p * C, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] += v[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[tid * 8 + i] = s[i];
    __syncthreads();
    if (tid < C) {
        const int vv = tid >> 3, i = tid & 7;
        float a = 0.f;
        for (int p = 0; p < planes; ++p) a += red[(p * nvec + vv) * 8 + i];
        out[(size_t)b * out_stride + tid] = a;
    }
}

// out[c] (+)= sum_r part[r*stride + c]; block = 32 columns x 8 row lanes
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ part, int64_t stride,
                                                    float* __restrict__ out, int R, int C, int accumulate) {
    __shared__ float red[8][33];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s = 0.f;
    if (c < C) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int r = rl;
        for (; r + 24 < R; r += 32) {
            s0 += part[(int64_t)r * stride + c];
            s1 += part[(int64_t)(r + 8) * stride + c];
            s2 += part[(int64_t)(r + 16) * stride + c];
            s3 += part[(int64_t)(r + 24) * stride + c];
        }
        for (; r < R; r += 8) s0 += part[(int64_t)r * stride + c];
        s = (s0 + s1) + (s2 + s3);
    }
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i][cl];
        out[c] = accumulate ? out[c] + t : t;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void sumpool2x2_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t total,
                                                        int H, int W, int C) {
    const int nvec = C >> 3;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int vec = (int)(idx % nvec);
        int64_t pix = idx / nvec;
        const int ox = (int)(pix % W);
        pix /= W;
        const int oy = (int)(pix % H);
        const int64_t b = pix / H;
        const T* p00 = x + (((b * 2 * H + 2 * oy) * 2 * W) + 2 * ox) * (int64_t)C + vec * 8;
        float a[8], t[8];
        load8(p00, a);
        load8(p00 + C, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] += t[i];
        load8(p00 + (int64_t)2 * W * C, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] += t[i];
        load8(p00 + (int64_t)2 * W * C + C, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] += t[i];
        store8(y + ((b * H + oy) * W + ox) * (int64_t)C + vec * 8, a);
    }
}

// 16-bit storage conversion (fp16 <-> bf16), 8 elements per thread: the attention extension keeps its internals in bf16 and
// converts the fp16 forward stream at its boundary
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast16_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int64_t nvec) {
    fp16_saturating_stores<TD>();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        float v[8];
        load8(src + i * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = sat16<TD>(v[k]);
        store8(dst + i * 8, v);
    }
}

bool gn_shape_ok(int C, int G) {
    if (C <= 0 || C > 256 || (C & 7) || 256 % (C >> 3)) return false;
    if (G < 0) return -G >= 1 && -G <= 16;        // -G channels per group, ceil(C / -G) groups (the last one may be partial: zero padding)
    if (G <= 0 || G > 256 || C % G) return false;
    const int cpg = C / G;
    return cpg == 1 || cpg == 2 || cpg == 4 || cpg == 8 || cpg == 16;
}
// groups of 1 or 2 channels, and group sizes given explicitly (G < 0: 3, 5, 6, 7 ... channels - the zero-padded widths 96, 160, 192, 224): the
// whole-sample streaming kernels with per-channel sums only (no register-resident / hybrid / fused forms)
static bool gn_narrow(int C, int G) { return G < 0 || C / G < 4; }

// Channel slab width of one workgroup: GMK_GN_KERNEL / gmk_set_kernel_choice(gn) 1 = whole sample, 3 = 32 channels,
// 4 = 64 channels, otherwise automatic.
int gn_slab_channels(int mode, int C, int G, int HW, int elem_bytes, bool backward) {
    if (G < 0) return C;                           // explicit group sizes: whole sample
    const int cpg = C / G;
    int CS = C;
    if (mode == 3) CS = 32;
    else if (mode == 4) CS = 64;
    else if (mode != 1) CS = backward && (int64_t)HW * C * elem_bytes > 65536 ? 32 : C;   // measured: docs/EXPERIMENTS.md 7b.5
    if (CS > C || C % CS || CS % cpg || (CS & 7) || kThreads % (CS >> 3)) CS = C;
    return CS;
}

}  // namespace

extern "C" int gmk_gn_silu_fwd(const void* x, void* y, const float* gamma, const float* beta, float* mean, float* rstd,
                               int B, int HW, int C, int groups, float eps, const float* stats_part, int tile_pixels,
                               int ntiles, float drop_p, uint64_t drop_seed, uint64_t drop_offset, const float* xadd,
                               int xadd_stride, int dtype, void* stream) {
    GMK_REQUIRE(!xadd || xadd_stride >= C, "gmk_gn_silu_fwd: xadd_stride %d < C %d", xadd_stride, C);
    if (xadd) stats_part = nullptr;     // the producer's statistics are those of x without the addend
    GMK_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "gmk_gn_silu_fwd: dropout probability %g outside [0, 1)", (double)drop_p);
    GMK_REQUIRE(!stats_part || (tile_pixels >= 32 && ntiles > 0 && HW >= 32 && C % 4 == 0),
                "gmk_gn_silu_fwd: bad statistics geometry");
    GMK_REQUIRE(x && y && gamma && beta && mean && rstd, "gmk_gn_silu_fwd: null pointer");
    GMK_REQUIRE(B > 0 && HW > 0 && gn_shape_ok(C, groups), "gmk_gn_silu_fwd: unsupported shape B=%d HW=%d C=%d G=%d", B,
                HW, C, groups);
    const int gn_mode = gmk_kernel_choice(2, "GMK_GN_KERNEL");
    // (at 7x7 / 8x8 the whole-sample streaming kernel wins: 19 vs 26 us at 8x8, B = 2048 - the slabs are too small to pay for a workgroup each)
    GMK_REQUIRE(dtype != GMK_F16 || !stats_part, "gmk_gn_silu_fwd: producer statistics go with bf16 tensors only");
    // 64 x 64 (HW up to 4096): a 32-channel slab of a sample is 256 KiB - the registers of ONE 1024-thread workgroup (16 pixels x 16 B per
    // thread); single read instead of the streaming kernel's two sweeps
    if (gn_narrow(C, groups)) {
        GMK_REQUIRE(!stats_part, "gmk_gn_silu_fwd: producer statistics come in 4-channel units; groups of %d channels", groups < 0 ? -groups : C / groups);
        gmk_note_kernel(22);
        if (dtype == GMK_BF16)
            gn_silu_fwd_kernel<bf16_t, true><<<B, kThreads, 0, gmk_stream(stream)>>>((const bf16_t*)x, (bf16_t*)y, gamma, beta, mean, rstd, HW, C, groups,
                                                                                   eps, nullptr, 0, 0, C, B, drop_p, drop_seed, drop_offset, xadd, xadd_stride);
        else if (dtype == GMK_F16)
            gn_silu_fwd_kernel<f16_t, true><<<B, kThreads, 0, gmk_stream(stream)>>>((const f16_t*)x, (f16_t*)y, gamma, beta, mean, rstd, HW, C, groups,
                                                                                  eps, nullptr, 0, 0, C, B, drop_p, drop_seed, drop_offset, xadd, xadd_stride);
        else if (dtype == GMK_F32)
            gn_silu_fwd_kernel<float, true><<<B, kThreads, 0, gmk_stream(stream)>>>((const float*)x, (float*)y, gamma, beta, mean, rstd, HW, C, groups,
                                                                                  eps, nullptr, 0, 0, C, B, drop_p, drop_seed, drop_offset, xadd, xadd_stride);
        else
            GMK_REQUIRE(false, "gmk_gn_silu_fwd: bad dtype %d", dtype);
        return gmk_check_launch("gmk_gn_silu_fwd");
    }
    const bool big = gmk_is16(dtype) && !stats_part && (gn_mode == 8 || gn_mode == 0) && HW > 1024 && HW <= 4096 && HW % 16 == 0 &&
                     C % 32 == 0 && 32 % (C / groups) == 0;
    if (big || (gmk_is16(dtype) && !stats_part && (gn_mode == 5 || gn_mode == 6 || (gn_mode == 0 && HW > 64)) && C % 64 == 0 &&
               32 % (C / groups) == 0 && gn_reg_iter(HW, 8) > 0)) {
        const int nvec = (big || gn_mode == 5) ? 4 : 8;        // 32- or 64-channel slabs (64 = whole 128-B lines)
        const int it = big ? 16 : gn_reg_iter(HW, nvec), planes = (HW + it - 1) / it, threads = (planes * nvec + 63) / 64 * 64;
        const int nblk = B * (C / (nvec * 8));
        gmk_note_kernel(21);
#define GMK_GN_FWD_REG(IT, NV)                                                                                                           \
    do {                                                                                                                                 \
        if (dtype == GMK_F16)                                                                                                            \
            gn_silu_fwd_reg_kernel<f16_t, IT, NV><<<nblk, threads, 0, gmk_stream(stream)>>>((const f16_t*)x, (f16_t*)y, gamma, beta, mean, rstd, \
                                                                                            HW, C, groups, eps, B, planes, drop_p, drop_seed,  \
                                                                                            drop_offset, xadd, xadd_stride);                   \
        else                                                                                                                             \
            gn_silu_fwd_reg_kernel<bf16_t, IT, NV><<<nblk, threads, 0, gmk_stream(stream)>>>((const bf16_t*)x, (bf16_t*)y, gamma, beta, mean,  \
                                                                                             rstd, HW, C, groups, eps, B, planes, drop_p,      \
                                                                                             drop_seed, drop_offset, xadd, xadd_stride);       \
    } while (0)
        if (nvec == 4) {
            if (it == 1) GMK_GN_FWD_REG(1, 4);
            else if (it == 2) GMK_GN_FWD_REG(2, 4);
            else if (it == 4) GMK_GN_FWD_REG(4, 4);
            else if (it == 8) GMK_GN_FWD_REG(8, 4);
            else GMK_GN_FWD_REG(16, 4);
        } else {
            if (it == 1) GMK_GN_FWD_REG(1, 8);
            else if (it == 2) GMK_GN_FWD_REG(2, 8);
            else if (it == 4) GMK_GN_FWD_REG(4, 8);
            else if (it == 8) GMK_GN_FWD_REG(8, 8);
            else if (it == 14) GMK_GN_FWD_REG(14, 8);
            else GMK_GN_FWD_REG(16, 8);
        }
#undef GMK_GN_FWD_REG
    } else if (dtype == GMK_BF16) {
        gmk_note_kernel(22);
        const int CS = stats_part ? C : gn_slab_channels(gn_mode, C, groups, HW, 2, false);
        gn_silu_fwd_kernel<bf16_t><<<B * (C / CS), kThreads, 0, gmk_stream(stream)>>>(
            (const bf16_t*)x, (bf16_t*)y, gamma, beta, mean, rstd, HW, C, groups, eps, stats_part, tile_pixels, ntiles, CS, B, drop_p,
            drop_seed, drop_offset, xadd, xadd_stride);
    } else if (dtype == GMK_F16) {
        gmk_note_kernel(22);
        const int CS = gn_slab_channels(gn_mode, C, groups, HW, 2, false);
        gn_silu_fwd_kernel<f16_t><<<B * (C / CS), kThreads, 0, gmk_stream(stream)>>>(
            (const f16_t*)x, (f16_t*)y, gamma, beta, mean, rstd, HW, C, groups, eps, nullptr, 0, 0, CS, B, drop_p,
            drop_seed, drop_offset, xadd, xadd_stride);
    } else if (dtype == GMK_F32) {
        gmk_note_kernel(22);
        const int CS = stats_part ? C : gn_slab_channels(gn_mode, C, groups, HW, 4, false);
        gn_silu_fwd_kernel<float><<<B * (C / CS), kThreads, 0, gmk_stream(stream)>>>(
            (const float*)x, (float*)y, gamma, beta, mean, rstd, HW, C, groups, eps, stats_part, tile_pixels, ntiles, CS, B, drop_p,
            drop_seed, drop_offset, xadd, xadd_stride);
    }
    else
        GMK_REQUIRE(false, "gmk_gn_silu_fwd: bad dtype %d", dtype);
    return gmk_check_launch("gmk_gn_silu_fwd");
}

// Statistics-only GroupNorm: one read of x, mean / rstd and the per-(sample, channel) affine tables of the normalisation,
// y = silu(x * scale + shift) with scale = rstd * gamma, shift = beta + (xadd - mean) * scale.  The convolution that consumes the
// normalised tensor applies them in its producer waves (gmk_conv_igemm gn_scale / gn_shift), so the normalised tensor is never
// written to or read back from HBM (reference sites simple_unet.py:161-163,169-172).
extern "C" int gmk_gn_stats(const void* x, const float* gamma, const float* beta, float* mean, float* rstd, float* tab_scale,
                            float* tab_shift, int tab_stride, int B, int HW, int C, int groups, float eps, const float* xadd,
                            int xadd_stride, int dtype, void* stream) {
    GMK_REQUIRE(x && gamma && beta && mean && rstd && tab_scale && tab_shift, "gmk_gn_stats: null pointer");
    GMK_REQUIRE(!xadd || xadd_stride >= C, "gmk_gn_stats: xadd_stride %d < C %d", xadd_stride, C);
    GMK_REQUIRE(tab_stride >= C, "gmk_gn_stats: tab_stride %d < C %d", tab_stride, C);
    GMK_REQUIRE(B > 0 && HW > 0 && gn_shape_ok(C, groups), "gmk_gn_stats: unsupported shape B=%d HW=%d C=%d G=%d", B, HW, C, groups);
    GMK_REQUIRE(gmk_is16(dtype), "gmk_gn_stats: 16-bit tensors only (the fused apply lives in the halo convolution)");
    GMK_REQUIRE(!gn_narrow(C, groups), "gmk_gn_stats: groups of %d channels run the unfused GroupNorm only", groups < 0 ? -groups : C / groups);
    if (C % 64 == 0 && 32 % (C / groups) == 0 && HW > 64 && gn_reg_iter(HW, 8) > 0) {      // same choice as gmk_gn_silu_fwd: same statistics bits
        const int nvec = 8;
        const int it = gn_reg_iter(HW, nvec), planes = (HW + it - 1) / it, threads = (planes * nvec + 63) / 64 * 64;
        const int nblk = B * (C / (nvec * 8));
        gmk_note_kernel(21);
#define GMK_GN_STATS_REG(IT)                                                                                                              \
    do {                                                                                                                                  \
        if (dtype == GMK_F16)                                                                                                             \
            gn_silu_fwd_reg_kernel<f16_t, IT, 8><<<nblk, threads, 0, gmk_stream(stream)>>>((const f16_t*)x, (f16_t*)nullptr, gamma, beta, mean,  \
                                                                                           rstd, HW, C, groups, eps, B, planes, 0.f, 0, 0, xadd, \
                                                                                           xadd_stride, tab_scale, tab_shift, tab_stride);        \
        else                                                                                                                              \
            gn_silu_fwd_reg_kernel<bf16_t, IT, 8><<<nblk, threads, 0, gmk_stream(stream)>>>((const bf16_t*)x, (bf16_t*)nullptr, gamma, beta,     \
                                                                                            mean, rstd, HW, C, groups, eps, B, planes, 0.f, 0, 0, \
                                                                                            xadd, xadd_stride, tab_scale, tab_shift, tab_stride); \
    } while (0)
        if (it == 1) GMK_GN_STATS_REG(1);
        else if (it == 2) GMK_GN_STATS_REG(2);
        else if (it == 4) GMK_GN_STATS_REG(4);
        else if (it == 8) GMK_GN_STATS_REG(8);
        else if (it == 14) GMK_GN_STATS_REG(14);
        else GMK_GN_STATS_REG(16);
#undef GMK_GN_STATS_REG
    } else if (HW > 1024 && HW <= 4096 && HW % 16 == 0 && C % 32 == 0 && 32 % (C / groups) == 0) {
        // 64 x 64: the 1024-thread register kernel on 32-channel slabs, as gmk_gn_silu_fwd chooses (same statistics bits)
        const int planes = HW / 16, nblk = B * (C / 32);
        gmk_note_kernel(21);
        if (dtype == GMK_F16)
            gn_silu_fwd_reg_kernel<f16_t, 16, 4><<<nblk, planes * 4, 0, gmk_stream(stream)>>>((const f16_t*)x, (f16_t*)nullptr, gamma, beta, mean, rstd,
                                                                                             HW, C, groups, eps, B, planes, 0.f, 0, 0, xadd, xadd_stride,
                                                                                             tab_scale, tab_shift, tab_stride);
        else
            gn_silu_fwd_reg_kernel<bf16_t, 16, 4><<<nblk, planes * 4, 0, gmk_stream(stream)>>>((const bf16_t*)x, (bf16_t*)nullptr, gamma, beta, mean, rstd,
                                                                                              HW, C, groups, eps, B, planes, 0.f, 0, 0, xadd, xadd_stride,
                                                                                              tab_scale, tab_shift, tab_stride);
    } else {
        const int CS = gn_slab_channels(0, C, groups, HW, 2, false);
        gmk_note_kernel(22);
        if (dtype == GMK_F16)
            gn_silu_fwd_kernel<f16_t><<<B * (C / CS), kThreads, 0, gmk_stream(stream)>>>(
                (const f16_t*)x, (f16_t*)nullptr, gamma, beta, mean, rstd, HW, C, groups, eps, nullptr, 0, 0, CS, B, 0.f, 0, 0, xadd,
                xadd_stride, tab_scale, tab_shift, tab_stride);
        else
            gn_silu_fwd_kernel<bf16_t><<<B * (C / CS), kThreads, 0, gmk_stream(stream)>>>(
                (const bf16_t*)x, (bf16_t*)nullptr, gamma, beta, mean, rstd, HW, C, groups, eps, nullptr, 0, 0, CS, B, 0.f, 0, 0, xadd,
                xadd_stride, tab_scale, tab_shift, tab_stride);
    }
    return gmk_check_launch("gmk_gn_stats");
}

extern "C" int gmk_gn_silu_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                               const float* rstd, const void* dadd1, const void* dadd2, void* dx, float* dgamma_part,
                               float* dbeta_part, float* dxsum, int dxsum_stride, int B, int HW, int C, int groups,
                               float drop_p, uint64_t drop_seed, uint64_t drop_offset, const float* xadd, int xadd_stride,
                               int dtype, int x_dtype, void* stream) {
    GMK_REQUIRE(!xadd || xadd_stride >= C, "gmk_gn_silu_bwd: xadd_stride %d < C %d", xadd_stride, C);
    GMK_REQUIRE(x_dtype == dtype || (dtype == GMK_BF16 && x_dtype == GMK_F16),
                "gmk_gn_silu_bwd: x of type %d with gradients of type %d (same type, or fp16 activations with bf16 gradients)", x_dtype, dtype);
    const bool xf16 = x_dtype == GMK_F16;
    GMK_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "gmk_gn_silu_bwd: dropout probability %g outside [0, 1)", (double)drop_p);
    GMK_REQUIRE(dy && x && gamma && beta && mean && rstd && dx && dgamma_part && dbeta_part,
                "gmk_gn_silu_bwd: null pointer");
    GMK_REQUIRE(B > 0 && HW > 0 && gn_shape_ok(C, groups), "gmk_gn_silu_bwd: unsupported shape B=%d HW=%d C=%d G=%d", B,
                HW, C, groups);
    GMK_REQUIRE(!dxsum || dxsum_stride >= C, "gmk_gn_silu_bwd: dxsum_stride %d < C %d", dxsum_stride, C);
    const int gn_mode = gn_narrow(C, groups) ? 1 : gmk_kernel_choice(2, "GMK_GN_KERNEL");      // narrow groups: whole-sample streaming kernel
    if (dtype == GMK_BF16 && (gn_mode == 0 || gn_mode == 7) && C % 32 == 0 && 32 % (C / groups) == 0 && HW > (gn_mode == 7 ? 511 : 64) &&
               HW <= 1024 && drop_p == 0.f) {
        const size_t lds = (size_t)HW * 64;
        gmk_note_kernel(23);
#define GMK_GN_BWD_HYB(IT, TH)                                                                                                          \
    do {                                                                                                                                \
        if (xf16)                                                                                                                       \
            gn_silu_bwd_hybrid_kernel<f16_t, IT, TH><<<B * (C / 32), TH, lds, gmk_stream(stream)>>>(                                     \
                (const bf16_t*)dy, (const f16_t*)x, gamma, beta, mean, rstd, (const bf16_t*)dadd1, (const bf16_t*)dadd2, (bf16_t*)dx,     \
                dgamma_part, dbeta_part, dxsum, dxsum_stride, HW, C, groups, B, drop_p, drop_seed, drop_offset, xadd, xadd_stride);   \
        else                                                                                                                            \
            gn_silu_bwd_hybrid_kernel<bf16_t, IT, TH><<<B * (C / 32), TH, lds, gmk_stream(stream)>>>(                                    \
                (const bf16_t*)dy, (const bf16_t*)x, gamma, beta, mean, rstd, (const bf16_t*)dadd1, (const bf16_t*)dadd2, (bf16_t*)dx,    \
                dgamma_part, dbeta_part, dxsum, dxsum_stride, HW, C, groups, B, drop_p, drop_seed, drop_offset, xadd, xadd_stride);   \
    } while (0)
        if (HW == 1024) {                            // 32x32 exactly: the pipelined form (-0.3 % of the headline step against the masked one)
            if (xf16)
                gn_silu_bwd_hybrid_kernel<f16_t, 8, 512, 4, 8, 0, true><<<B * (C / 32), 512, lds, gmk_stream(stream)>>>(
                    (const bf16_t*)dy, (const f16_t*)x, gamma, beta, mean, rstd, (const bf16_t*)dadd1, (const bf16_t*)dadd2, (bf16_t*)dx,
                    dgamma_part, dbeta_part, dxsum, dxsum_stride, HW, C, groups, B, drop_p, drop_seed, drop_offset, xadd, xadd_stride);
            else
                gn_silu_bwd_hybrid_kernel<bf16_t, 8, 512, 4, 8, 0, true><<<B * (C / 32), 512, lds, gmk_stream(stream)>>>(
                    (const bf16_t*)dy, (const bf16_t*)x, gamma, beta, mean, rstd, (const bf16_t*)dadd1, (const bf16_t*)dadd2, (bf16_t*)dx,
                    dgamma_part, dbeta_part, dxsum, dxsum_stride, HW, C, groups, B, drop_p, drop_seed, drop_offset, xadd, xadd_stride);
        } else
        if (HW <= 256) GMK_GN_BWD_HYB(4, 256);       // 14x14 (-16 % against the streaming kernel), 16x16 (-10 %); at 7x7 / 8x8 the whole-sample
                                                     // streaming kernel is as fast or faster (42 vs 46 us at 8x8, B = 2048)
        else if (HW <= 832) GMK_GN_BWD_HYB(13, 256); // 28x28: 3 workgroups of 4 waves per CU
        else GMK_GN_BWD_HYB(8, 512);                 // 32x32: 2 workgroups of 8 waves (-13 % against the streaming kernel)
#undef GMK_GN_BWD_HYB
    } else if (dtype == GMK_BF16 && (gn_mode == 0 || gn_mode == 9) && C % 32 == 0 && 32 % (C / groups) == 0 && HW == 4096 && drop_p == 0.f) {
        // 64 x 64 on 32-channel slabs (64-byte segments per pixel row instead of the 16-channel form's 32): one 512-thread workgroup
        // per CU at 256 registers per lane - 19 / 32 of x in 152 KiB of LDS, the rest of x and 7 / 8 of dy in registers
        gmk_note_kernel(23);
        constexpr int kParkBytes = 19 * 128 * 64;
        static const hipError_t attr = hipFuncSetAttribute((const void*)gn_silu_bwd_hybrid_kernel<bf16_t, 32, 512, 4, 28, 13>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, kParkBytes);
        static const hipError_t attr16 = hipFuncSetAttribute((const void*)gn_silu_bwd_hybrid_kernel<f16_t, 32, 512, 4, 28, 13>,
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, kParkBytes);
        (void)attr; (void)attr16;
        if (xf16)
            gn_silu_bwd_hybrid_kernel<f16_t, 32, 512, 4, 28, 13><<<B * (C / 32), 512, kParkBytes, gmk_stream(stream)>>>(
                (const bf16_t*)dy, (const f16_t*)x, gamma, beta, mean, rstd, (const bf16_t*)dadd1, (const bf16_t*)dadd2, (bf16_t*)dx,
                dgamma_part, dbeta_part, dxsum, dxsum_stride, HW, C, groups, B, drop_p, drop_seed, drop_offset, xadd, xadd_stride);
        else
            gn_silu_bwd_hybrid_kernel<bf16_t, 32, 512, 4, 28, 13><<<B * (C / 32), 512, kParkBytes, gmk_stream(stream)>>>(
                (const bf16_t*)dy, (const bf16_t*)x, gamma, beta, mean, rstd, (const bf16_t*)dadd1, (const bf16_t*)dadd2, (bf16_t*)dx,
                dgamma_part, dbeta_part, dxsum, dxsum_stride, HW, C, groups, B, drop_p, drop_seed, drop_offset, xadd, xadd_stride);
    } else if (dtype == GMK_BF16 && (gn_mode == 0 || gn_mode == 7) && C % 16 == 0 && 16 % (C / groups) == 0 && HW > 1024 &&
               HW <= 4096 && drop_p == 0.f) {
        // 64 x 64: 16-channel slabs, one workgroup of 16 waves per CU
        gmk_note_kernel(23);
        static const hipError_t attr = hipFuncSetAttribute((const void*)gn_silu_bwd_hybrid_kernel<bf16_t, 8, 1024, 2>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 4096 * 32);
        static const hipError_t attr16 = hipFuncSetAttribute((const void*)gn_silu_bwd_hybrid_kernel<f16_t, 8, 1024, 2>,
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, 4096 * 32);
        (void)attr; (void)attr16;
        if (xf16)
            gn_silu_bwd_hybrid_kernel<f16_t, 8, 1024, 2><<<B * (C / 16), 1024, (size_t)HW * 32, gmk_stream(stream)>>>(
                (const bf16_t*)dy, (const f16_t*)x, gamma, beta, mean, rstd, (const bf16_t*)dadd1, (const bf16_t*)dadd2, (bf16_t*)dx,
                dgamma_part, dbeta_part, dxsum, dxsum_stride, HW, C, groups, B, drop_p, drop_seed, drop_offset, xadd, xadd_stride);
        else
            gn_silu_bwd_hybrid_kernel<bf16_t, 8, 1024, 2><<<B * (C / 16), 1024, (size_t)HW * 32, gmk_stream(stream)>>>(
                (const bf16_t*)dy, (const bf16_t*)x, gamma, beta, mean, rstd, (const bf16_t*)dadd1, (const bf16_t*)dadd2, (bf16_t*)dx,
                dgamma_part, dbeta_part, dxsum, dxsum_stride, HW, C, groups, B, drop_p, drop_seed, drop_offset, xadd, xadd_stride);
    } else if (dtype == GMK_BF16 && xf16) {
        gmk_note_kernel(24);
        const int CS = gn_slab_channels(gn_mode, C, groups, HW, 2, true);
        gn_silu_bwd_kernel<bf16_t, f16_t><<<B * (C / CS), kThreads, 0, gmk_stream(stream)>>>(
            (const bf16_t*)dy, (const f16_t*)x, gamma, beta, mean, rstd, (const bf16_t*)dadd1, (const bf16_t*)dadd2,
            (bf16_t*)dx, dgamma_part, dbeta_part, dxsum, dxsum_stride, HW, C, groups, CS, B, drop_p, drop_seed, drop_offset, xadd,
            xadd_stride);
    } else if (dtype == GMK_BF16) {
        gmk_note_kernel(24);
        const int CS = gn_slab_channels(gn_mode, C, groups, HW, 2, true);
        gn_silu_bwd_kernel<bf16_t><<<B * (C / CS), kThreads, 0, gmk_stream(stream)>>>(
            (const bf16_t*)dy, (const bf16_t*)x, gamma, beta, mean, rstd, (const bf16_t*)dadd1, (const bf16_t*)dadd2,
            (bf16_t*)dx, dgamma_part, dbeta_part, dxsum, dxsum_stride, HW, C, groups, CS, B, drop_p, drop_seed, drop_offset, xadd,
            xadd_stride);
    } else if (dtype == GMK_F32) {
        gmk_note_kernel(24);
        const int CS = gn_slab_channels(gn_mode, C, groups, HW, 4, true);
        gn_silu_bwd_kernel<float><<<B * (C / CS), kThreads, 0, gmk_stream(stream)>>>(
            (const float*)dy, (const float*)x, gamma, beta, mean, rstd, (const float*)dadd1, (const float*)dadd2,
            (float*)dx, dgamma_part, dbeta_part, dxsum, dxsum_stride, HW, C, groups, CS, B, drop_p, drop_seed, drop_offset, xadd,
            xadd_stride);
    }
    else
        GMK_REQUIRE(false, "gmk_gn_silu_bwd: bad dtype %d", dtype);
    return gmk_check_launch("gmk_gn_silu_bwd");
}

extern "C" int gmk_cast16(const void* src, void* dst, int64_t n, int src_dtype, int dst_dtype, void* stream) {
    GMK_REQUIRE(src && dst && n > 0 && n % 8 == 0, "gmk_cast16: null pointer or n %lld not a multiple of 8", (long long)n);
    const int64_t nvec = n / 8;
    const int blocks = (int)((nvec + 255) / 256 < 8192 ? (nvec + 255) / 256 : 8192);
    if (src_dtype == GMK_F16 && dst_dtype == GMK_BF16)
        cast16_kernel<f16_t, bf16_t><<<blocks, 256, 0, gmk_stream(stream)>>>((const f16_t*)src, (bf16_t*)dst, nvec);
    else if (src_dtype == GMK_BF16 && dst_dtype == GMK_F16)
        cast16_kernel<bf16_t, f16_t><<<blocks, 256, 0, gmk_stream(stream)>>>((const bf16_t*)src, (f16_t*)dst, nvec);
    else
        GMK_REQUIRE(false, "gmk_cast16: %d -> %d (fp16 <-> bf16 only)", src_dtype, dst_dtype);
    return gmk_check_launch("gmk_cast16");
}

extern "C" int gmk_chansum(const void* x, float* out, int out_stride, int B, int HW, int C, int dtype, void* stream) {
    GMK_REQUIRE(x && out, "gmk_chansum: null pointer");
    GMK_REQUIRE(B > 0 && HW > 0 && C > 0 && C <= 256 && !(C & 7) && 256 % (C >> 3) == 0 && out_stride >= C,
                "gmk_chansum: unsupported shape B=%d HW=%d C=%d stride=%d", B, HW, C, out_stride);
    if (dtype == GMK_BF16)
        chansum_kernel<bf16_t><<<B, kThreads, 0, gmk_stream(stream)>>>((const bf16_t*)x, out, out_stride, HW, C);
    else if (dtype == GMK_F32)
        chansum_kernel<float><<<B, kThreads, 0, gmk_stream(stream)>>>((const float*)x, out, out_stride, HW, C);
    else
        GMK_REQUIRE(false, "gmk_chansum: bad dtype %d", dtype);
    return gmk_check_launch("gmk_chansum");
}

extern "C" int gmk_colsum(const float* part, int64_t stride, float* out, int R, int C, int accumulate, void* stream) {
    GMK_REQUIRE(part && out, "gmk_colsum: null pointer");
    GMK_REQUIRE(R > 0 && C > 0 && stride >= C, "gmk_colsum: bad shape R=%d C=%d stride=%lld", R, C, (long long)stride);
    colsum_kernel<<<(C + 31) / 32, 256, 0, gmk_stream(stream)>>>(part, stride, out, R, C, accumulate);
    return gmk_check_launch("gmk_colsum");
}

extern "C" int gmk_sumpool2x2(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream) {
    GMK_REQUIRE(x && y, "gmk_sumpool2x2: null pointer");
    GMK_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && !(C & 7), "gmk_sumpool2x2: bad shape");
    const int64_t total = (int64_t)B * H * W * (C >> 3);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (dtype == GMK_BF16)
        sumpool2x2_kernel<bf16_t><<<blocks, 256, 0, gmk_stream(stream)>>>((const bf16_t*)x, (bf16_t*)y, total, H, W, C);
    else if (dtype == GMK_F32)
        sumpool2x2_kernel<float><<<blocks, 256, 0, gmk_stream(stream)>>>((const float*)x, (float*)y, total, H, W, C);
    else
        GMK_REQUIRE(false, "gmk_sumpool2x2: bad dtype %d", dtype);
    return gmk_check_launch("gmk_sumpool2x2");
}

