// Embedding path of the U-Net (reference gms/diffusion/simple_unet.py:20-34,45-64,166,205-224), fp32:
// sinusoidal features, the one-hot label encoding with its -1 mask, and a small strided fp32 GEMM used for every
// nn.Linear forward / backward of the path ([B,64..256] x [256,..]: ~1 MFLOP per image, launch-bound, not MFMA work).
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

constexpr int TM = 64, TN = 64, TK = 16;

// C[i][j] = (acc ? C : 0) + rowscale[i] * (bias[j] + sum_k fa(A[i*sa0 + k*sa1]) * fb(B[k*sb0 + j*sb1]))
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int64_t sa0, int64_t sa1,
                                                      const float* __restrict__ Bm, int64_t sb0, int64_t sb1,
                                                      float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                      const float* __restrict__ bias, const float* __restrict__ rowscale,
                                                      int silu, int accumulate) {
    __shared__ float As[TK][TM + 4];
    __shared__ float Bs[TK][TN + 4];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int i0 = blockIdx.y * TM, j0 = blockIdx.x * TN;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    for (int k0 = 0; k0 < K; k0 += TK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256;              // 1024 elements per tile
            {   // A tile: k fastest when sa1 == 1, else i fastest
                int i, k;
                if (sa1 == 1) { k = idx % TK; i = idx / TK; } else { i = idx % TM; k = idx / TM; }
                float v = 0.f;
                if (i0 + i < M && k0 + k < K) {
                    v = A[(int64_t)(i0 + i) * sa0 + (int64_t)(k0 + k) * sa1];
                    if (silu & 1) v = v / (1.0f + expf(-v));
                }
                As[k][i] = v;
            }
            {
                int j, k;
                if (sb0 == 1) { k = idx % TK; j = idx / TK; } else { j = idx % TN; k = idx / TN; }
                float v = 0.f;
                if (j0 + j < N && k0 + k < K) {
                    v = Bm[(int64_t)(k0 + k) * sb0 + (int64_t)(j0 + j) * sb1];
                    if (silu & 2) v = v / (1.0f + expf(-v));
                }
                Bs[k][j] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TK; ++k) {
            float a[4], b[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = As[k][ty * 4 + e]; b[e] = Bs[k][tx * 4 + e]; }
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = fmaf(a[x], b[y], acc[x][y]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const int i = i0 + ty * 4 + x;
        if (i >= M) continue;
        const float rsc = rowscale ? rowscale[i] : 1.f;
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const int j = j0 + tx * 4 + y;
            if (j >= N) continue;
            float v = acc[x][y] + (bias ? bias[j] : 0.f);
            v *= rsc;
            float* c = C + (int64_t)i * ldc + j;
            *c = accumulate ? *c + v : v;
        }
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

extern "C" int gmk_gemm_f32(const float* A, int64_t sa0, int64_t sa1, const float* B, int64_t sb0, int64_t sb1, float* C,
                            int64_t ldc, int M, int N, int K, const float* bias, const float* rowscale, int silu,
                            int accumulate, void* stream) {
    GMK_REQUIRE(A && B && C, "gmk_gemm_f32: null pointer");
    GMK_REQUIRE(M > 0 && N > 0 && K > 0 && ldc >= N, "gmk_gemm_f32: bad shape M=%d N=%d K=%d ldc=%lld", M, N, K,
                (long long)ldc);
    dim3 grid((N + TN - 1) / TN, (M + TM - 1) / TM);
    gemm_f32_kernel<<<grid, 256, 0, gmk_stream(stream)>>>(A, sa0, sa1, B, sb0, sb1, C, ldc, M, N, K, bias, rowscale, silu,
                                                          accumulate);
    return gmk_check_launch("gmk_gemm_f32");
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
