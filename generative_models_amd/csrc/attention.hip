// Self-attention core (BASELINE config 5): batched QK^T / softmax / PV and their backward as four small
// building blocks.  The reference `SimpleUnet` has NO attention block — this is the north_star's "optional
// self-attention block", an extension without a reference call site; its definition is the CPU restatement `attention_block` the tests check it against
// ("parity unpinned").  At 256 tokens x 128 channels per sample the whole block is < 1 % of the step's FLOPs, so the design
// goal is simplicity: one batched K-contiguous GEMM on the matrix cores (v_mfma_f32_32x32x16_bf16; a plain fp32 FMA path
// for the fp32 parity mode), a batched transpose, and row softmax forward / backward.
//   C[b][m][n] = alpha * sum_k A[b][m][k] * B[b][n][k]        ("NT": both operands K-contiguous)
#include "gmk_common.h"

namespace {

constexpr int kT = 64;            // output tile 64 x 64, K-step 64

// ---- bf16 inputs: MFMA ---------------------------------------------------------------------------------------
template <typename TOUT>
__global__ __launch_bounds__(256) void bgemm_nt_bf16_kernel(const bf16_t* __restrict__ A, int64_t a_batch, int64_t lda,
                                                           const bf16_t* __restrict__ Bm, int64_t b_batch, int64_t ldb,
                                                           TOUT* __restrict__ C, int64_t c_batch, int64_t ldc, int M, int N, int K,
                                                           float alpha) {
    __shared__ __attribute__((aligned(16))) bf16_t As[kT][kT + 8];
    __shared__ __attribute__((aligned(16))) bf16_t Bs[kT][kT + 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int ntn = (N + kT - 1) / kT;
    const int m0 = (blockIdx.x / ntn) * kT, n0 = (blockIdx.x % ntn) * kT;
    const bf16_t* Ab = A + (int64_t)blockIdx.y * a_batch;
    const bf16_t* Bb = Bm + (int64_t)blockIdx.y * b_batch;
    const int lrow = tid >> 2, lcol = (tid & 3) * 16;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += kT) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = k0 + lcol + 8 * u;
            const bool kin = k < K;                           // K is a multiple of 8: a 16-B chunk is all in or all out
            bf16x8 va = zero, vb = zero;
            if (kin && m0 + lrow < M) va = *reinterpret_cast<const bf16x8*>(Ab + (int64_t)(m0 + lrow) * lda + k);
            if (kin && n0 + lrow < N) vb = *reinterpret_cast<const bf16x8*>(Bb + (int64_t)(n0 + lrow) * ldb + k);
            *reinterpret_cast<bf16x8*>(&As[lrow][lcol + 8 * u]) = va;
            *reinterpret_cast<bf16x8*>(&Bs[lrow][lcol + 8 * u]) = vb;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(&As[wm * 32 + r][ks * 16 + 8 * h]);
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(&Bs[wn * 32 + r][ks * 16 + 8 * h]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // D: column n = lane & 31, rows m = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
    TOUT* Cb = C + (int64_t)blockIdx.y * c_batch;
    const int n = n0 + wn * 32 + r;
    if (n < N) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (m < M) Cb[(int64_t)m * ldc + n] = (TOUT)(alpha * acc[e]);
        }
    }
}

// ---- fp32 inputs (parity mode): plain FMA, 4 x 4 outputs per thread -----------------------------------------------
template <typename TOUT>
__global__ __launch_bounds__(256) void bgemm_nt_f32_kernel(const float* __restrict__ A, int64_t a_batch, int64_t lda,
                                                          const float* __restrict__ Bm, int64_t b_batch, int64_t ldb,
                                                          TOUT* __restrict__ C, int64_t c_batch, int64_t ldc, int M, int N, int K,
                                                          float alpha) {
    constexpr int TK = 16;
    __shared__ float As[TK][kT + 4];
    __shared__ float Bs[TK][kT + 4];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int ntn = (N + kT - 1) / kT;
    const int m0 = (blockIdx.x / ntn) * kT, n0 = (blockIdx.x % ntn) * kT;
    const float* Ab = A + (int64_t)blockIdx.y * a_batch;
    const float* Bb = Bm + (int64_t)blockIdx.y * b_batch;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    for (int k0 = 0; k0 < K; k0 += TK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256, k = idx % TK, i = idx / TK;
            As[k][i] = (m0 + i < M && k0 + k < K) ? Ab[(int64_t)(m0 + i) * lda + k0 + k] : 0.f;
            Bs[k][i] = (n0 + i < N && k0 + k < K) ? Bb[(int64_t)(n0 + i) * ldb + k0 + k] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TK; ++k) {
            float av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { av[a] = As[k][ty * 4 + a]; bv[a] = Bs[k][tx * 4 + a]; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = fmaf(av[a], bv[b], acc[a][b]);
        }
        __syncthreads();
    }
    TOUT* Cb = C + (int64_t)blockIdx.y * c_batch;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int m = m0 + ty * 4 + a, n = n0 + tx * 4 + b;
            if (m < M && n < N) Cb[(int64_t)m * ldc + n] = (TOUT)(alpha * acc[a][b]);
        }
}

// ---- out[b][c][r] = in[b][r][c] -----------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ in, int64_t in_batch, int64_t ld_in, T* __restrict__ out,
                                                       int64_t out_batch, int64_t ld_out, int R, int Cc) {
    __shared__ T tile[32][33];
    const int tc = (Cc + 31) / 32;
    const int r0 = (blockIdx.x / tc) * 32, c0 = (blockIdx.x % tc) * 32;
    const T* ib = in + (int64_t)blockIdx.y * in_batch;
    T* ob = out + (int64_t)blockIdx.y * out_batch;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < R && c0 + tx < Cc) tile[i][tx] = ib[(int64_t)(r0 + i) * ld_in + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < Cc && r0 + tx < R) ob[(int64_t)(c0 + i) * ld_out + r0 + tx] = tile[tx][i];
}

// ---- row softmax of scale * S (one wave per row, N <= 1024) and its backward -------------------------------------------
template <typename TOUT>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ S, TOUT* __restrict__ P, int64_t rows, int N,
                                                         float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* s = S + row * N;
    float v[16];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int j = lane + 64 * i;
        v[i] = j < N ? s[j] * scale : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        v[i] = lane + 64 * i < N ? expf(v[i] - mx) : 0.f;
        sum += v[i];
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int j = lane + 64 * i;
        if (j < N) P[row * N + j] = (TOUT)(v[i] * inv);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* __restrict__ P, const float* __restrict__ dP, T* __restrict__ dS,
                                                         int64_t rows, int N, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float p[16], g[16];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int j = lane + 64 * i;
        p[i] = j < N ? (float)P[row * N + j] : 0.f;
        g[i] = j < N ? dP[row * N + j] : 0.f;
        dot = fmaf(p[i], g[i], dot);
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int j = lane + 64 * i;
        if (j < N) dS[row * N + j] = (T)(scale * p[i] * (g[i] - dot));
    }
}

}  // namespace

extern "C" int gmk_bgemm_nt(const void* A, int64_t a_batch, int64_t lda, const void* B, int64_t b_batch, int64_t ldb, void* C,
                            int64_t c_batch, int64_t ldc, int batch, int M, int N, int K, float alpha, int in_dtype, int out_dtype,
                            void* stream) {
    GMK_REQUIRE(A && B && C, "gmk_bgemm_nt: null pointer");
    GMK_REQUIRE(batch > 0 && batch < 65536 && M > 0 && N > 0 && K > 0 && lda >= K && ldb >= K && ldc >= N, "gmk_bgemm_nt: bad shape");
    const dim3 grid(((M + kT - 1) / kT) * ((N + kT - 1) / kT), batch);
    hipStream_t st = gmk_stream(stream);
    if (in_dtype == GMK_BF16) {
        GMK_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && a_batch % 8 == 0 && b_batch % 8 == 0,
                    "gmk_bgemm_nt: bf16 operands need K and the strides to be multiples of 8 elements (16-byte rows)");
        if (out_dtype == GMK_BF16)
            bgemm_nt_bf16_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)A, a_batch, lda, (const bf16_t*)B, b_batch, ldb, (bf16_t*)C,
                                                               c_batch, ldc, M, N, K, alpha);
        else if (out_dtype == GMK_F32)
            bgemm_nt_bf16_kernel<float><<<grid, 256, 0, st>>>((const bf16_t*)A, a_batch, lda, (const bf16_t*)B, b_batch, ldb, (float*)C,
                                                              c_batch, ldc, M, N, K, alpha);
        else GMK_REQUIRE(false, "gmk_bgemm_nt: bad out_dtype %d", out_dtype);
    } else if (in_dtype == GMK_F32) {
        GMK_REQUIRE(out_dtype == GMK_F32, "gmk_bgemm_nt: fp32 operands give fp32 results");
        bgemm_nt_f32_kernel<float><<<grid, 256, 0, st>>>((const float*)A, a_batch, lda, (const float*)B, b_batch, ldb, (float*)C, c_batch,
                                                         ldc, M, N, K, alpha);
    } else GMK_REQUIRE(false, "gmk_bgemm_nt: bad in_dtype %d", in_dtype);
    return gmk_check_launch("gmk_bgemm_nt");
}

extern "C" int gmk_transpose(const void* in, int64_t in_batch, int64_t ld_in, void* out, int64_t out_batch, int64_t ld_out,
                             int batch, int R, int Cc, int dtype, void* stream) {
    GMK_REQUIRE(in && out, "gmk_transpose: null pointer");
    GMK_REQUIRE(batch > 0 && batch < 65536 && R > 0 && Cc > 0 && ld_in >= Cc && ld_out >= R, "gmk_transpose: bad shape");
    const dim3 grid(((R + 31) / 32) * ((Cc + 31) / 32), batch);
    if (dtype == GMK_BF16)
        transpose_kernel<bf16_t><<<grid, 256, 0, gmk_stream(stream)>>>((const bf16_t*)in, in_batch, ld_in, (bf16_t*)out, out_batch, ld_out, R, Cc);
    else if (dtype == GMK_F32)
        transpose_kernel<float><<<grid, 256, 0, gmk_stream(stream)>>>((const float*)in, in_batch, ld_in, (float*)out, out_batch, ld_out, R, Cc);
    else GMK_REQUIRE(false, "gmk_transpose: bad dtype %d", dtype);
    return gmk_check_launch("gmk_transpose");
}

extern "C" int gmk_softmax_fwd(const float* S, void* P, int64_t rows, int N, float scale, int out_dtype, void* stream) {
    GMK_REQUIRE(S && P, "gmk_softmax_fwd: null pointer");
    GMK_REQUIRE(rows > 0 && N > 0 && N <= 1024, "gmk_softmax_fwd: rows of 1..1024 elements");
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    if (out_dtype == GMK_BF16) softmax_fwd_kernel<bf16_t><<<blocks, 256, 0, gmk_stream(stream)>>>(S, (bf16_t*)P, rows, N, scale);
    else if (out_dtype == GMK_F32) softmax_fwd_kernel<float><<<blocks, 256, 0, gmk_stream(stream)>>>(S, (float*)P, rows, N, scale);
    else GMK_REQUIRE(false, "gmk_softmax_fwd: bad dtype %d", out_dtype);
    return gmk_check_launch("gmk_softmax_fwd");
}

extern "C" int gmk_softmax_bwd(const void* P, const float* dP, void* dS, int64_t rows, int N, float scale, int dtype, void* stream) {
    GMK_REQUIRE(P && dP && dS, "gmk_softmax_bwd: null pointer");
    GMK_REQUIRE(rows > 0 && N > 0 && N <= 1024, "gmk_softmax_bwd: rows of 1..1024 elements");
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    if (dtype == GMK_BF16) softmax_bwd_kernel<bf16_t><<<blocks, 256, 0, gmk_stream(stream)>>>((const bf16_t*)P, dP, (bf16_t*)dS, rows, N, scale);
    else if (dtype == GMK_F32) softmax_bwd_kernel<float><<<blocks, 256, 0, gmk_stream(stream)>>>((const float*)P, dP, (float*)dS, rows, N, scale);
    else GMK_REQUIRE(false, "gmk_softmax_bwd: bad dtype %d", dtype);
    return gmk_check_launch("gmk_softmax_bwd");
}
