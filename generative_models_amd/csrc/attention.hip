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

// ---------------------------------------------------------------------------------------------------------------
// Fused forward: O = softmax(Q K^T / sqrt(C)) V for one sample per workgroup, C = 128 channels, N <= 256 tokens (BASELINE config 5:
// 16 x 16 tokens at the lowest level of a 64 x 64 net), no N x N matrix in HBM (training passes `pout` to keep P for the backward).
// K and V of the sample are staged once into LDS (128 KiB at N = 256); wave w owns queries 32 w .. 32 w + 31.
//   S^T = K Q^T   one v_mfma_f32_32x32x16 chain per 32-key block with the KEY on the accumulator row and the QUERY on the lane,
//                 so a lane holds 128 of its query's scores: the row maximum / sum are register reductions plus one
//                 v_permlane32_swap with the lane that holds the other 128;
//   O^T = V^T P^T the probabilities never leave the registers: registers 8 s .. 8 s + 7 of a score block, rounded pairwise, ARE
//                 the B operand of k-step s of the next product (k order 16 s + 8 (j >> 2) + 4 h + (j & 3)); the A operand
//                 V^T[d][key] comes from LDS in that same key order - ds_read_b64_tr_b16 on the bf16 image, plain 8-byte reads
//                 on a transposed fp8 image laid out in that order.
// kFp8: both contractions on v_mfma_f32_32x32x16_fp8_fp8 (OCP e4m3; K, V, Q and P rounded to fp8, fp32 accumulation) - the
// north_star's "fp8 MFMA" for config 5.  Parity: vs the bf16 three-kernel path, tests/test_gpu_ops.py.
// ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2a;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4a;

__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
    int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (unsigned)v;
}

template <bool kFp8>
__global__ __launch_bounds__(512, 2) void attn_fused_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                               bf16_t* __restrict__ pout, int N, float c_log2e) {
    constexpr int C = 128;
    // bf16: Ks [N][256 B] (16-B chunk ^ (row & 15)), Vs [N][256 B] (T10 image (b): chunk ^ (((row & 3) << 2) | ((row >> 2) & 3)))
    // fp8 : Ks [N][128 B] (8-B chunk ^ ((row >> 1) & 15)), Vt [128 d][N B] in the permuted key order (8-B chunk ^ (d & 31))
    __shared__ __attribute__((aligned(16))) char smem[2 * 256 * 256];
    char* Ks = smem;
    char* Vs = smem + (kFp8 ? N * 128 : N * 256);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.x;
    const bf16_t* base = qkv + (size_t)b * N * (3 * C);

    // ---- stage K and V: N rows x 16 chunks of 8 channels each
    for (int idx = tid; idx < N * 16; idx += 512) {
        const int row = idx >> 4, ch = idx & 15;
        const bf16x8 kv = *reinterpret_cast<const bf16x8*>(base + (size_t)row * (3 * C) + C + ch * 8);
        const bf16x8 vv = *reinterpret_cast<const bf16x8*>(base + (size_t)row * (3 * C) + 2 * C + ch * 8);
        if (!kFp8) {
            *reinterpret_cast<bf16x8*>(Ks + row * 256 + ((ch ^ (row & 15)) << 4)) = kv;
            *reinterpret_cast<bf16x8*>(Vs + row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4)) = vv;
        } else {
            u32x2a k8 = {pack4_fp8((float)kv[0], (float)kv[1], (float)kv[2], (float)kv[3]),
                         pack4_fp8((float)kv[4], (float)kv[5], (float)kv[6], (float)kv[7])};
            *reinterpret_cast<u32x2a*>(Ks + row * 128 + ((ch ^ ((row >> 1) & 15)) << 3)) = k8;
            // V^T image: byte (d, pos), pos = the key's place in the k order of the P^T operand: key = 32 kb + 16 s + 8 g + 4 hh + jj
            // sits at pos = 32 kb + 16 s + 8 hh + 4 g + jj
            const int kin = row & 31;
            const int pos = (row & ~31) | (kin & 16) | (((kin >> 2) & 1) << 3) | (((kin >> 3) & 1) << 2) | (kin & 3);
            const unsigned lo = pack4_fp8((float)vv[0], (float)vv[1], (float)vv[2], (float)vv[3]);
            const unsigned hi = pack4_fp8((float)vv[4], (float)vv[5], (float)vv[6], (float)vv[7]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int d = ch * 8 + e;
                const unsigned byte = ((e < 4 ? lo : hi) >> (8 * (e & 3))) & 0xFFu;
                Vs[d * N + ((((pos >> 3) ^ (d & 31)) & ((N >> 3) - 1)) << 3) + (pos & 7)] = (char)byte;
            }
        }
    }
    __syncthreads();
    if (wave * 32 >= N) return;                      // N = 64: two waves compute (no barrier follows)

    // ---- Q fragments of this wave's 32 queries: B operand, lane (query r, half h) holds d = 16 kg + 8 h .. + 7
    const bf16_t* qrow = base + (size_t)(wave * 32 + r) * (3 * C);
    bf16x8 qf[8];
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) qf[kg] = *reinterpret_cast<const bf16x8*>(qrow + kg * 16 + h * 8);
    long q8[8];
    if (kFp8) {
#pragma unroll
        for (int kg = 0; kg < 8; ++kg) {
            const u32x2a v = {pack4_fp8((float)qf[kg][0], (float)qf[kg][1], (float)qf[kg][2], (float)qf[kg][3]),
                              pack4_fp8((float)qf[kg][4], (float)qf[kg][5], (float)qf[kg][6], (float)qf[kg][7])};
            q8[kg] = __builtin_bit_cast(long, v);
        }
    }

    // ---- S^T = K Q^T, one 32-key block at a time
    const int nkb = N >> 5;
    f32x16 sc[8];
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
#pragma unroll
        for (int e = 0; e < 16; ++e) sc[kb][e] = 0.f;
        if (kb < nkb) {
            const int key = kb * 32 + r;
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) {
                if (!kFp8) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + key * 256 + (((kg * 2 + h) ^ (key & 15)) << 4));
                    sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kg], sc[kb], 0, 0, 0);
                } else {
                    const long kf = *reinterpret_cast<const long*>(Ks + key * 128 + (((kg * 2 + h) ^ ((key >> 1) & 15)) << 3));
                    sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(kf, q8[kg], sc[kb], 0, 0, 0);
                }
            }
        }
    }
    // ---- softmax over the query's N keys: this lane holds N / 2 of them, lane ^ 32 the others
    float mx = -3.0e38f;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
        if (kb < nkb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sc[kb][e]);
        }
    { float a = mx, bq = mx; halves_swap32(a, bq); mx = fmaxf(a, bq); }
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
        if (kb < nkb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) { sc[kb][e] = __builtin_amdgcn_exp2f((sc[kb][e] - mx) * c_log2e); sum += sc[kb][e]; }
        }
    { float a = sum, bq = sum; halves_swap32(a, bq); sum = a + bq; }
    const float inv = 1.0f / sum;
    const int qi = wave * 32 + r;
    if (pout) {       // P[b][q][key] for the backward pass: key = 32 kb + 8 g + 4 h + (0..3) -> 8-byte stores
        bf16_t* prow = pout + ((size_t)b * N + qi) * N;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb)
            if (kb < nkb) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bf16x4 t = {(bf16_t)(sc[kb][4 * g] * inv), (bf16_t)(sc[kb][4 * g + 1] * inv), (bf16_t)(sc[kb][4 * g + 2] * inv),
                                      (bf16_t)(sc[kb][4 * g + 3] * inv)};
                    *reinterpret_cast<bf16x4*>(prow + kb * 32 + 8 * g + 4 * h) = t;
                }
            }
    }
    // ---- O^T = V^T P^T (un-normalised probabilities; 1 / sum applied to the result)
    f32x16 oc[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) oc[db][e] = 0.f;
    const int gg = lane >> 4, i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        if (kb < nkb) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (!kFp8) {
                    const bf16x8 pf = {(bf16_t)sc[kb][8 * s], (bf16_t)sc[kb][8 * s + 1], (bf16_t)sc[kb][8 * s + 2], (bf16_t)sc[kb][8 * s + 3],
                                       (bf16_t)sc[kb][8 * s + 4], (bf16_t)sc[kb][8 * s + 5], (bf16_t)sc[kb][8 * s + 6], (bf16_t)sc[kb][8 * s + 7]};
#pragma unroll
                    for (int db = 0; db < 4; ++db) {
                        // transposed read of V: rows = keys 32 kb + 16 s + 4 hh (+ 8) + 0..3, columns d = 32 db + 16 (gg & 1) + 0..15
                        const int c0 = 4 * db + 2 * (gg & 1) + (pp >> 1);
                        typedef __attribute__((ext_vector_type(8))) short s16x8;
                        s16x4 lo, hi;
                        {
                            const int row = kb * 32 + 16 * s + 4 * (gg >> 1) + qq;
                            lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)(Vs + row * 256 + ((c0 ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4) + 8 * (pp & 1)));
                        }
                        {
                            const int row = kb * 32 + 16 * s + 8 + 4 * (gg >> 1) + qq;
                            hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)(Vs + row * 256 + ((c0 ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4) + 8 * (pp & 1)));
                        }
                        const s16x8 av = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        oc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), pf, oc[db], 0, 0, 0);
                    }
                } else {
                    const u32x2a pv = {pack4_fp8(sc[kb][8 * s], sc[kb][8 * s + 1], sc[kb][8 * s + 2], sc[kb][8 * s + 3]),
                                       pack4_fp8(sc[kb][8 * s + 4], sc[kb][8 * s + 5], sc[kb][8 * s + 6], sc[kb][8 * s + 7])};
                    const long pf = __builtin_bit_cast(long, pv);
#pragma unroll
                    for (int db = 0; db < 4; ++db) {
                        const int d = 32 * db + r;
                        const int chunk = (kb * 32 + 16 * s + 8 * h) >> 3;
                        const long vf = *reinterpret_cast<const long*>(Vs + d * N + (((chunk ^ (d & 31)) & ((N >> 3) - 1)) << 3));
                        oc[db] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(vf, pf, oc[db], 0, 0, 0);
                    }
                }
            }
        }
    }
    // ---- O[b][q][d], d = 32 db + 8 g + 4 h + (0..3)
    bf16_t* orow = o + ((size_t)b * N + qi) * C;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const bf16x4 t = {(bf16_t)(oc[db][4 * g] * inv), (bf16_t)(oc[db][4 * g + 1] * inv), (bf16_t)(oc[db][4 * g + 2] * inv),
                              (bf16_t)(oc[db][4 * g + 3] * inv)};
            *reinterpret_cast<bf16x4*>(orow + db * 32 + 8 * g + 4 * h) = t;
        }
}


// ---------------------------------------------------------------------------------------------------------------
// Fused backward of the same core (round 5): given dO, recompute P = softmax(scale Q K^T) block by block and produce dQ, dK, dV without any
// N x N matrix in HBM (the three-kernel path stored P [B][N][N] in bf16 and a fp32 dP of the same shape).  Two kernels on the forward's
// skeleton, one workgroup per sample, 32 rows per wave, bf16 operands, fp32 accumulation, everything deterministic (no atomics):
//   attn_bwd_dq_kernel    K and V staged in LDS; wave w owns QUERIES 32 w ..: pass 1 = S^T = K Q^T for the row statistics (maximum, sum - the
//                         forward's own arithmetic) and D_i = sum_c dO_ic O_ic; pass 2 per 32-key block: S^T again, dP^T = V dO^T, dS^T = scale P^T
//                         (dP^T - D) in registers = the B operand of dQ^T += K^T dS^T (K^T by transposed LDS reads, as V^T in the forward).  Leaves
//                         (log2-domain LSE, D) per query in a small fp32 table for the second kernel.
//   attn_bwd_dkv_kernel   Q and dO staged in LDS; wave w owns KEYS 32 w ..: per 32-query block S = Q K^T and dP = dO V^T with the query on the
//                         accumulator row, P = exp2(s c - LSE), dS = scale P (dP - D); P and dS, rounded pairwise, are the B operands of
//                         dV^T += dO^T P and dK^T += Q^T dS.
// LDS image of a staged [N][128] bf16 matrix: row * 256 B, 16-byte chunk ch at ((ch ^ (row & 15)) << 4) (conflict-free row reads; the transposed
// reads take some bank conflicts on it - the block is < 1 % of a step).  The fp8 forward's backward runs here too: the gradient of the bf16 attention
// at the same q, k, v (straight-through for the e4m3 rounding).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void stage_rows_bf16(char* dst, const bf16_t* src, int ld, int N, int tid) {
    for (int idx = tid; idx < N * 16; idx += 512) {
        const int row = idx >> 4, ch = idx & 15;
        *reinterpret_cast<bf16x8*>(dst + row * 256 + ((ch ^ (row & 15)) << 4)) = *reinterpret_cast<const bf16x8*>(src + (size_t)row * ld + ch * 8);
    }
}
// A operand [m = 32 channels d of block db][k = 16 rows of the staged matrix, in the k order of a score block's registers 8 s .. 8 s + 7]:
// rows 32 blk + 16 s + 4 (gg >> 1) + qq (+ 8), columns d = 32 db + 16 (gg & 1) + ... (the forward's V^T read, on the row image)
__device__ __forceinline__ bf16x8 read_transposed(const char* img, int blk, int s, int db, int lane) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const int gg = lane >> 4, i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;
    const int c0 = 4 * db + 2 * (gg & 1) + (pp >> 1);
    const int row0 = blk * 32 + 16 * s + 4 * (gg >> 1) + qq, row1 = row0 + 8;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)(img + row0 * 256 + ((c0 ^ (row0 & 15)) << 4) + 8 * (pp & 1)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)(img + row1 * 256 + ((c0 ^ (row1 & 15)) << 4) + 8 * (pp & 1)));
    const s16x8 av = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, av);
}
__device__ __forceinline__ bf16x8 round8(const f32x16& v, int s) {
    const bf16x8 t = {(bf16_t)v[8 * s], (bf16_t)v[8 * s + 1], (bf16_t)v[8 * s + 2], (bf16_t)v[8 * s + 3],
                      (bf16_t)v[8 * s + 4], (bf16_t)v[8 * s + 5], (bf16_t)v[8 * s + 6], (bf16_t)v[8 * s + 7]};
    return t;
}

__global__ __launch_bounds__(512, 2) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                            const bf16_t* __restrict__ dout, bf16_t* __restrict__ dqkv,
                                                            float* __restrict__ stats, int N, float scale, float c_log2e) {
    constexpr int C = 128;
    __shared__ __attribute__((aligned(16))) char smem[2 * 256 * 256];
    char* Ks = smem;
    char* Vs = smem + N * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.x;
    const bf16_t* base = qkv + (size_t)b * N * (3 * C);
    stage_rows_bf16(Ks, base + C, 3 * C, N, tid);
    stage_rows_bf16(Vs, base + 2 * C, 3 * C, N, tid);
    __syncthreads();
    if (wave * 32 >= N) return;
    const int nkb = N >> 5;
    const int qi = wave * 32 + r;
    // B operands of this lane: query qi, channels d = 16 kg + 8 h .. + 7
    bf16x8 qf[8], dof[8];
    const bf16_t* qrow = base + (size_t)qi * (3 * C);
    const bf16_t* dorow = dout + ((size_t)b * N + qi) * C;
    const bf16_t* orow = o + ((size_t)b * N + qi) * C;
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {
        qf[kg] = *reinterpret_cast<const bf16x8*>(qrow + kg * 16 + h * 8);
        dof[kg] = *reinterpret_cast<const bf16x8*>(dorow + kg * 16 + h * 8);
    }
    // ---- D = sum_c dO O over the query's 128 channels (this lane holds 64 of them, lane ^ 32 the others)
    float D = 0.f;
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {
        const bf16x8 of = *reinterpret_cast<const bf16x8*>(orow + kg * 16 + h * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) D = fmaf((float)dof[kg][e], (float)of[e], D);
    }
    { float a = D, bq = D; halves_swap32(a, bq); D = a + bq; }
    // ---- pass 1: the row statistics of S^T = K Q^T (the forward's arithmetic)
    float mx = -3.0e38f, sum = 0.f;
    {
        f32x16 sc[8];
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) sc[kb][e] = 0.f;
            if (kb < nkb) {
                const int key = kb * 32 + r;
#pragma unroll
                for (int kg = 0; kg < 8; ++kg) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + key * 256 + (((kg * 2 + h) ^ (key & 15)) << 4));
                    sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kg], sc[kb], 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sc[kb][e]);
            }
        }
        { float a = mx, bq = mx; halves_swap32(a, bq); mx = fmaxf(a, bq); }
#pragma unroll
        for (int kb = 0; kb < 8; ++kb)
            if (kb < nkb) {
#pragma unroll
                for (int e = 0; e < 16; ++e) sum += __builtin_amdgcn_exp2f((sc[kb][e] - mx) * c_log2e);
            }
        { float a = sum, bq = sum; halves_swap32(a, bq); sum = a + bq; }
    }
    const float inv = 1.0f / sum;
    if (h == 0) {
        float* st = stats + ((size_t)b * N + qi) * 2;
        st[0] = mx * c_log2e + __builtin_amdgcn_logf(sum);        // log2-domain LSE: P = exp2(s c - st[0])   (v_log_f32 is log2)
        st[1] = D;
    }
    // ---- pass 2: per key block S^T again, dP^T = V dO^T, dS^T, dQ^T += K^T dS^T
    f32x16 oc[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) oc[db][e] = 0.f;
#pragma unroll 1
    for (int kb = 0; kb < nkb; ++kb) {
        const int key = kb * 32 + r;
        f32x16 sv, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) { sv[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
        for (int kg = 0; kg < 8; ++kg) {
            const int off = key * 256 + (((kg * 2 + h) ^ (key & 15)) << 4);
            sv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(Ks + off), qf[kg], sv, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(Vs + off), dof[kg], dp, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float pe = __builtin_amdgcn_exp2f((sv[e] - mx) * c_log2e) * inv;
            sv[e] = scale * pe * (dp[e] - D);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8 dsf = round8(sv, s);
#pragma unroll
            for (int db = 0; db < 4; ++db)
                oc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_transposed(Ks, kb, s, db, lane), dsf, oc[db], 0, 0, 0);
        }
    }
    // ---- dQ[b][q][d] into columns [0, C) of dqkv, d = 32 db + 8 g + 4 h + (0..3)
    bf16_t* drow = dqkv + ((size_t)b * N + qi) * (3 * C);
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const bf16x4 t = {(bf16_t)oc[db][4 * g], (bf16_t)oc[db][4 * g + 1], (bf16_t)oc[db][4 * g + 2], (bf16_t)oc[db][4 * g + 3]};
            *reinterpret_cast<bf16x4*>(drow + db * 32 + 8 * g + 4 * h) = t;
        }
}

__global__ __launch_bounds__(512, 2) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                             const float* __restrict__ stats, bf16_t* __restrict__ dqkv, int N, float scale,
                                                             float c_log2e) {
    constexpr int C = 128;
    __shared__ __attribute__((aligned(16))) char smem[2 * 256 * 256];
    char* Qs = smem;
    char* Gs = smem + N * 256;                                 // dO
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.x;
    const bf16_t* base = qkv + (size_t)b * N * (3 * C);
    stage_rows_bf16(Qs, base, 3 * C, N, tid);
    stage_rows_bf16(Gs, dout + (size_t)b * N * C, C, N, tid);
    __syncthreads();
    if (wave * 32 >= N) return;
    const int nqb = N >> 5;
    const int ki = wave * 32 + r;
    const bf16_t* krow = base + (size_t)ki * (3 * C) + C;
    const bf16_t* vrow = krow + C;
    const float* st = stats + (size_t)b * N * 2;
    f32x16 dvc[4], dkc[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dvc[db][e] = 0.f; dkc[db][e] = 0.f; }
#pragma unroll 1
    for (int qb = 0; qb < nqb; ++qb) {
        const int qrow = qb * 32 + r;
        f32x16 sv, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) { sv[e] = 0.f; dp[e] = 0.f; }
#pragma unroll 2
        for (int kg = 0; kg < 8; ++kg) {      // B operands: this lane's key, channels 16 kg + 8 h .. (re-read per block: L1 / L2 hits; 64 registers otherwise)
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(krow + kg * 16 + h * 8);
            const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vrow + kg * 16 + h * 8);
            const int off = qrow * 256 + (((kg * 2 + h) ^ (qrow & 15)) << 4);
            sv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(Qs + off), kf, sv, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(Gs + off), vf, dp, 0, 0, 0);
        }
        // accumulator element e = 4 g + j belongs to query 32 qb + 8 g + 4 h + j: its LSE and D from the table of the first kernel
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float* sp = st + (size_t)(qb * 32 + 8 * g + 4 * h) * 2;
            const f32x4 s01 = *reinterpret_cast<const f32x4*>(sp), s23 = *reinterpret_cast<const f32x4*>(sp + 4);
            const float lse[4] = {s01[0], s01[2], s23[0], s23[2]}, Dq[4] = {s01[1], s01[3], s23[1], s23[3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float pe = __builtin_amdgcn_exp2f(sv[4 * g + j] * c_log2e - lse[j]);
                sv[4 * g + j] = pe;
                dp[4 * g + j] = scale * pe * (dp[4 * g + j] - Dq[j]);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8 pf = round8(sv, s), dsf = round8(dp, s);
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                dvc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_transposed(Gs, qb, s, db, lane), pf, dvc[db], 0, 0, 0);
                dkc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_transposed(Qs, qb, s, db, lane), dsf, dkc[db], 0, 0, 0);
            }
        }
    }
    bf16_t* drow = dqkv + ((size_t)b * N + ki) * (3 * C);
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const bf16x4 tk = {(bf16_t)dkc[db][4 * g], (bf16_t)dkc[db][4 * g + 1], (bf16_t)dkc[db][4 * g + 2], (bf16_t)dkc[db][4 * g + 3]};
            const bf16x4 tv = {(bf16_t)dvc[db][4 * g], (bf16_t)dvc[db][4 * g + 1], (bf16_t)dvc[db][4 * g + 2], (bf16_t)dvc[db][4 * g + 3]};
            *reinterpret_cast<bf16x4*>(drow + C + db * 32 + 8 * g + 4 * h) = tk;
            *reinterpret_cast<bf16x4*>(drow + 2 * C + db * 32 + 8 * g + 4 * h) = tv;
        }
}

}  // namespace

extern "C" int gmk_attention_fwd(const void* qkv, void* o, void* p_out, int B, int N, int C, float scale, int fp8, void* stream) {
    GMK_REQUIRE(qkv && o && B > 0, "gmk_attention_fwd: bad arguments");
    GMK_REQUIRE(C == 128 && (N == 64 || N == 128 || N == 256), "gmk_attention_fwd: needs C = 128 and N in {64, 128, 256} tokens (got C=%d N=%d)", C, N);
    const float c_log2e = scale * 1.4426950408889634f;
    if (fp8) attn_fused_fwd_kernel<true><<<B, 512, 0, gmk_stream(stream)>>>((const bf16_t*)qkv, (bf16_t*)o, (bf16_t*)p_out, N, c_log2e);
    else attn_fused_fwd_kernel<false><<<B, 512, 0, gmk_stream(stream)>>>((const bf16_t*)qkv, (bf16_t*)o, (bf16_t*)p_out, N, c_log2e);
    return gmk_check_launch("gmk_attention_fwd");
}

extern "C" int gmk_attention_bwd(const void* qkv, const void* o, const void* d_o, void* dqkv, float* stats, int B, int N, int C, float scale,
                                 void* stream) {
    GMK_REQUIRE(qkv && o && d_o && dqkv && stats && B > 0, "gmk_attention_bwd: bad arguments");
    GMK_REQUIRE(C == 128 && (N == 64 || N == 128 || N == 256), "gmk_attention_bwd: needs C = 128 and N in {64, 128, 256} tokens (got C=%d N=%d)", C, N);
    const float c_log2e = scale * 1.4426950408889634f;
    hipStream_t st = gmk_stream(stream);
    attn_bwd_dq_kernel<<<B, 512, 0, st>>>((const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)d_o, (bf16_t*)dqkv, stats, N, scale, c_log2e);
    attn_bwd_dkv_kernel<<<B, 512, 0, st>>>((const bf16_t*)qkv, (const bf16_t*)d_o, stats, (bf16_t*)dqkv, N, scale, c_log2e);
    return gmk_check_launch("gmk_attention_bwd");
}

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
