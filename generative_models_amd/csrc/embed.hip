// Embedding path of the U-Net (reference gms/diffusion/simple_unet.py:20-34,45-64,166,205-224), fp32:
// sinusoidal features, the one-hot label encoding with its -1 mask, and a small strided fp32 GEMM used for every
// nn.Linear forward / backward of the path ([B,64..256] x [256,..]: ~1 MFLOP per image; fp32 MFMA, exact fp32 products).
#include "gmk_common.h"

namespace {

__global__ __launch_bounds__(256) void temb_kernel(const float* __restrict__ t, const float* __restrict__ freqs,
                                                  float* __restrict__ out, int B) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * 32) return;
    const int b = idx >> 5, k = idx & 31;
    const float arg = t[b] * freqs[k];                 // simple_unet.py:220
    out[b * 64 + k] = cosf(arg);                       // :221  [cos | sin]
    out[b * 64 + 32 + k] = sinf(arg);
}

__global__ __launch_bounds__(256) void guide_onehot_kernel(const int64_t* __restrict__ guide, float* __restrict__ onehot,
                                                          float* __restrict__ keep, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int64_t g = guide[b];
    const bool masked = g == -1;                       // simple_unet.py:54 (integer compare)
    const int64_t cls = masked ? 0 : g;                // :55
    for (int k = 0; k < 10; ++k) onehot[b * 10 + k] = (k == cls) ? 1.f : 0.f;   // :56
    keep[b] = masked ? 0.f : 1.f;                      // :57 rows zeroed after the MLP
}

// classifier-free label drop, in place: y[b] = -1 where element b of the uniform stream (seed, offset) is below p
__global__ __launch_bounds__(256) void label_drop_kernel(int64_t* __restrict__ y, int B, float p, uint64_t seed, uint64_t offset) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    uint32_t rnd[4];
    philox4x32(offset + (uint64_t)(b >> 2), seed, rnd);      // same element mapping as gmk_rng_uniform
    if (u01(rnd[b & 3]) < p) y[b] = -1;
}

// out[0] = scale * sum(x[0..n)): one workgroup, fixed summation order (deterministic)
__global__ __launch_bounds__(256) void sum_scale_kernel(const float* __restrict__ x, int n, float scale, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + (red[2] + red[3])) * scale;
}

constexpr int TM = 64, TN = 64, TK = 16;

// C[i][j] = (acc ? C : 0) + rowscale[i] * (bias[j] + sum_k fa(A[i*sa0 + k*sa1]) * fb(B[k*sb0 + j*sb1]))
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int64_t sa0, int64_t sa1,
                                                      const float* __restrict__ Bm, int64_t sb0, int64_t sb1,
                                                      float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                      const float* __restrict__ bias, const float* __restrict__ bias2,
                                                      const float* __restrict__ rowscale,
                                                      int silu, int accumulate, float* __restrict__ ws, int kchunk) {
    // Two LDS stages: the next k-slab's 8 global values per thread are loaded into registers BEFORE the current slab's MFMAs and stored
    // to the other stage behind them - one barrier per slab, the global-load latency (the whole cost of the single-stage loop: ~ 2 us per
    // 16-k slab, 60 us for K = 256) hides under the matrix instructions.  Same MFMA sequence per accumulator: bit-identical results.
    __shared__ float As[2][TK][TM + 4];
    __shared__ float Bs[2][TK][TN + 4];
    const int tid = threadIdx.x;
    // wave (wm, wn) owns the 32 x 32 quarter of the 64 x 64 tile: one fp32 MFMA accumulator (v_mfma_f32_32x32x2_f32 - true fp32
    // multiply-adds, K = 2 per instruction: lane = (r, h) feeds A[row r][k + h] and B[k + h][col r])
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    const int i0 = blockIdx.y * TM, j0 = blockIdx.x * TN;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    // split-K: slice blockIdx.z covers k in [z*kchunk, (z+1)*kchunk) and, when there is more than one slice, writes its
    // raw partial sums to ws[z][M][N]; gemm_splitk_reduce_kernel then applies bias / rowscale / accumulate
    const int kbeg = blockIdx.z * kchunk;
    if (gridDim.z > 1) K = min(K, kbeg + kchunk);
    int ia[4], ka[4], jb[4], kb[4];              // this thread's 4 + 4 elements of a slab: A tile k fastest when sa1 == 1 (else i fastest), B alike
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int idx = tid + e * 256;
        if (sa1 == 1) { ka[e] = idx % TK; ia[e] = idx / TK; } else { ia[e] = idx % TM; ka[e] = idx / TM; }
        if (sb0 == 1) { kb[e] = idx % TK; jb[e] = idx / TK; } else { jb[e] = idx % TN; kb[e] = idx / TN; }
    }
    float ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = 0.f;
            if (i0 + ia[e] < M && k0 + ka[e] < K) {
                v = A[(int64_t)(i0 + ia[e]) * sa0 + (int64_t)(k0 + ka[e]) * sa1];
                if (silu & 1) v = v / (1.0f + expf(-v));
            }
            ra[e] = v;
            float w = 0.f;
            if (j0 + jb[e] < N && k0 + kb[e] < K) {
                w = Bm[(int64_t)(k0 + kb[e]) * sb0 + (int64_t)(j0 + jb[e]) * sb1];
                if (silu & 2) w = w / (1.0f + expf(-w));
            }
            rb[e] = w;
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { As[buf][ka[e]][ia[e]] = ra[e]; Bs[buf][kb[e]][jb[e]] = rb[e]; }
    };
    gload(kbeg);
    sstore(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = kbeg; k0 < K; k0 += TK) {
        const bool more = k0 + TK < K;
        if (more) gload(k0 + TK);
#pragma unroll
        for (int k = 0; k < TK; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[buf][k + h][wm * 32 + r], Bs[buf][k + h][wn * 32 + r], acc, 0, 0, 0);
        if (more) sstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // accumulator element e of lane (r, h): row (e & 3) + 8 (e >> 2) + 4 h, column r of the wave's quarter
    const int j = j0 + wn * 32 + r;
    if (gridDim.z > 1) {
        float* w = ws + (int64_t)blockIdx.z * M * N;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = i0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (i < M && j < N) w[(int64_t)i * N + j] = acc[e];
        }
        return;
    }
    if (j >= N) return;
    const float bj = (bias ? bias[j] : 0.f) + (bias2 ? bias2[j] : 0.f);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = i0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (i >= M) continue;
        const float v = (acc[e] + bj) * (rowscale ? rowscale[i] : 1.f);
        float* c = C + (int64_t)i * ldc + j;
        *c = accumulate ? *c + v : v;
    }
}

__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, int64_t ldc,
                                                               int M, int N, int nz, const float* __restrict__ bias,
                                                               const float* __restrict__ bias2,
                                                               const float* __restrict__ rowscale, int accumulate) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)M * N) return;
    const int i = (int)(idx / N), j = (int)(idx - (int64_t)i * N);
    float s = 0.f;
    for (int z = 0; z < nz; ++z) s += ws[(int64_t)z * M * N + idx];
    float v = (s + ((bias ? bias[j] : 0.f) + (bias2 ? bias2[j] : 0.f))) * (rowscale ? rowscale[i] : 1.f);
    float* c = C + (int64_t)i * ldc + j;
    *c = accumulate ? *c + v : v;
}

// up to 16 independent column sums per launch: out_k[c] = sum_r part_k[r*stride_k + c]
struct ColsumTable {
    const float* part[16]; long long stride[16]; float* out[16]; int R[16]; int C[16];
};
// 32 columns x 32 row lanes per workgroup: a [2048 x 128] partial is 64 rows per thread, four loads in flight (8 row lanes made it 256
// rows per thread on 64 workgroups: 35 us per launch, latency-bound)
__global__ __launch_bounds__(1024) void colsum_multi_kernel(const ColsumTable t) {
    __shared__ float red[32][33];
    const int k = blockIdx.y;
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const int R = t.R[k], C = t.C[k];
    if (blockIdx.x * 32 >= C) return;
    const float* part = t.part[k];
    const long long stride = t.stride[k];
    float s = 0.f;
    if (c < C) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int r = rl;
        for (; r + 96 < R; r += 128) {
            s0 += part[(long long)r * stride + c];
            s1 += part[(long long)(r + 32) * stride + c];
            s2 += part[(long long)(r + 64) * stride + c];
            s3 += part[(long long)(r + 96) * stride + c];
        }
        for (; r < R; r += 32) s0 += part[(long long)r * stride + c];
        s = (s0 + s1) + (s2 + s3);
    }
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) v += red[i][cl];
        t.out[k][c] = v;
    }
}

__global__ __launch_bounds__(256) void silu_bwd_kernel(const float* __restrict__ dpost, const float* __restrict__ pre,
                                                      const float* __restrict__ rowscale, float* __restrict__ dpre,
                                                      int64_t n, int ncols) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = pre[i];
    const float s = 1.0f / (1.0f + expf(-x));
    float d = dpost[i] * s * (1.f + x * (1.f - s));
    if (rowscale) d *= rowscale[i / ncols];
    dpre[i] = d;
}

__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ in, const float* __restrict__ rowscale,
                                                        float* __restrict__ out, int64_t n, int ncols) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = in[i] * rowscale[i / ncols];
}

}  // namespace

extern "C" int gmk_timestep_embedding(const float* t, const float* freqs, float* out, int B, void* stream) {
    GMK_REQUIRE(t && freqs && out && B > 0, "gmk_timestep_embedding: bad arguments");
    temb_kernel<<<(B * 32 + 255) / 256, 256, 0, gmk_stream(stream)>>>(t, freqs, out, B);
    return gmk_check_launch("gmk_timestep_embedding");
}

extern "C" int gmk_guide_onehot(const int64_t* guide, float* onehot, float* keep, int B, void* stream) {
    GMK_REQUIRE(guide && onehot && keep && B > 0, "gmk_guide_onehot: bad arguments");
    guide_onehot_kernel<<<(B + 255) / 256, 256, 0, gmk_stream(stream)>>>(guide, onehot, keep, B);
    return gmk_check_launch("gmk_guide_onehot");
}

extern "C" int gmk_label_drop(int64_t* y, int B, float p, uint64_t seed, uint64_t offset, void* stream) {
    GMK_REQUIRE(y && B > 0 && p >= 0.f && p <= 1.f, "gmk_label_drop: bad arguments");
    label_drop_kernel<<<(B + 255) / 256, 256, 0, gmk_stream(stream)>>>(y, B, p, seed, offset);
    return gmk_check_launch("gmk_label_drop");
}

extern "C" int gmk_mean(const float* x, int n, float* out, void* stream) {
    GMK_REQUIRE(x && out && n > 0, "gmk_mean: bad arguments");
    sum_scale_kernel<<<1, 256, 0, gmk_stream(stream)>>>(x, n, 1.0f / (float)n, out);
    return gmk_check_launch("gmk_mean");
}

static int gemm_ksplit(int M, int N, int K, int* kchunk) {
    // Short contractions (K <= 256: the Linears of the embedding MLPs and the 12 emb_layers, one ROW per sample) decide their split as if M were 512,
    // whatever it is: a sample's embedding then does not depend on how many other samples share its launch (round 5: the sampler runs large batches
    // as two halves on two streams and has to produce the same bits; before, M = 2048 ran unsplit and M = 1024 in four K-chunks - one ulp apart).
    if (K <= 256) M = 512;
    const int tiles = ((N + TN - 1) / TN) * ((M + TM - 1) / TM);
    int nz = 1;
    if (tiles < 128 && K >= 256) {
        nz = (256 + tiles - 1) / tiles;
        if (nz > K / 64) nz = K / 64;
        if (nz > 32) nz = 32;
        if (nz < 1) nz = 1;
    } else if (tiles < 256 && K >= 1024) {
        nz = 2;                                   // half a chip of tiles and a long contraction (the 12 emb_layers' data gradient: 128 tiles, K = 1536)
    }
    int kc = (K + nz - 1) / nz;
    kc = (kc + TK - 1) / TK * TK;
    nz = (K + kc - 1) / kc;
    *kchunk = kc;
    return nz;
}

extern "C" int64_t gmk_gemm_f32_workspace_bytes(int M, int N, int K) {
    int kc;
    const int nz = gemm_ksplit(M, N, K, &kc);
    return nz > 1 ? (int64_t)nz * M * N * 4 : 0;
}

extern "C" int gmk_gemm_f32(const float* A, int64_t sa0, int64_t sa1, const float* B, int64_t sb0, int64_t sb1, float* C,
                            int64_t ldc, int M, int N, int K, const float* bias, const float* bias2, const float* rowscale,
                            int silu, int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
    GMK_REQUIRE(A && B && C, "gmk_gemm_f32: null pointer");
    GMK_REQUIRE(M > 0 && N > 0 && K > 0 && ldc >= N, "gmk_gemm_f32: bad shape M=%d N=%d K=%d ldc=%lld", M, N, K,
                (long long)ldc);
    int kc;
    int nz = gemm_ksplit(M, N, K, &kc);
    if (nz > 1 && (!workspace || workspace_bytes < (int64_t)nz * M * N * 4)) { nz = 1; kc = K; }
    dim3 grid((N + TN - 1) / TN, (M + TM - 1) / TM, nz);
    gemm_f32_kernel<<<grid, 256, 0, gmk_stream(stream)>>>(A, sa0, sa1, B, sb0, sb1, C, ldc, M, N, K, bias, bias2, rowscale,
                                                          silu, accumulate, (float*)workspace, kc);
    int rc = gmk_check_launch("gmk_gemm_f32");
    if (rc || nz == 1) return rc;
    const int64_t n = (int64_t)M * N;
    gemm_splitk_reduce_kernel<<<(int)((n + 255) / 256), 256, 0, gmk_stream(stream)>>>((const float*)workspace, C, ldc, M, N, nz,
                                                                                       bias, bias2, rowscale, accumulate);
    return gmk_check_launch("gmk_gemm_f32(reduce)");
}

extern "C" int gmk_colsum_multi(int n, const float* const* part, const int64_t* stride, float* const* out, const int* R,
                                const int* C, void* stream) {
    GMK_REQUIRE(n >= 1 && n <= 16 && part && stride && out && R && C, "gmk_colsum_multi: bad arguments (1 <= n <= 16)");
    ColsumTable t;
    int cmax = 0;
    for (int k = 0; k < n; ++k) {
        GMK_REQUIRE(part[k] && out[k] && R[k] > 0 && C[k] > 0 && stride[k] >= C[k], "gmk_colsum_multi: bad problem %d", k);
        t.part[k] = part[k]; t.stride[k] = stride[k]; t.out[k] = out[k]; t.R[k] = R[k]; t.C[k] = C[k];
        if (C[k] > cmax) cmax = C[k];
    }
    colsum_multi_kernel<<<dim3((cmax + 31) / 32, n), 1024, 0, gmk_stream(stream)>>>(t);
    return gmk_check_launch("gmk_colsum_multi");
}

extern "C" int gmk_silu_bwd(const float* dpost, const float* pre, const float* rowscale, float* dpre, int64_t n, int ncols,
                            void* stream) {
    GMK_REQUIRE(dpost && pre && dpre && n > 0 && ncols > 0, "gmk_silu_bwd: bad arguments");
    silu_bwd_kernel<<<(int)((n + 255) / 256), 256, 0, gmk_stream(stream)>>>(dpost, pre, rowscale, dpre, n, ncols);
    return gmk_check_launch("gmk_silu_bwd");
}

extern "C" int gmk_scale_rows(const float* in, const float* rowscale, float* out, int64_t n, int ncols, void* stream) {
    GMK_REQUIRE(in && rowscale && out && n > 0 && ncols > 0, "gmk_scale_rows: bad arguments");
    scale_rows_kernel<<<(int)((n + 255) / 256), 256, 0, gmk_stream(stream)>>>(in, rowscale, out, n, ncols);
    return gmk_check_launch("gmk_scale_rows");
}
