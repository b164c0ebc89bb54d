// gmk — gfx950 (MI355X) kernels for the diffusion hot path.  Internal helpers shared by the .hip files.
// Device code is written for CDNA4 only: 64-lane waves, MFMA 32x32 tiles, 160 KiB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gmk.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define GMK_LDS __attribute__((address_space(3)))

// ---- host-side error plumbing (no exceptions cross the C ABI) -------------------------------
void gmk_set_error(const char* fmt, ...);
int gmk_check_launch(const char* what);   // returns 0 or the positive hipError_t of the launch
int gmk_kernel_choice(int which, const char* env);   // 0 conv (GMK_CONV_KERNEL), 1 wgrad (GMK_WGRAD_KERNEL), 2 GN
int gmk_conv1x1_pair_stream_try(const void* src, int64_t npix, const void* w_rows256, void* out_a, void* out_b, int dtype, hipStream_t stream);   // conv1x1_stream.hip
int gmk_conv1x1_wgrad_stream_try(const void* dy, int dy_cstride, const void* src0, const void* src1, int64_t npix, float* slab, int64_t slab_bytes,
                                 bool x_f16, hipStream_t stream);        // conv1x1_stream.hip: > 0 = slabs written
int gmk_conv_subpixel_takes(int B, int H, int W, int cin, int cout, int w_rows, int ntaps, int out_cstride, int dtype);      // conv_subpixel.hip
bool fp32_split(void);                // fp32 mode: exact fp32 MFMA chains (default) or operands as bf16 hi + lo on the bf16 matrix cores (GMK_FP32_SPLIT=1 / gmk_set_fp32_exact(0))
int gmk_cu_limit(void);                   // workgroups a persistent kernel may occupy (gmk_set_cu_limit / GMK_CU_LIMIT, default 256)
void gmk_note_kernel(int id);             // 1 conv_igemm_kernel, 2 conv_igemm_dma_kernel, 3 conv3x3_halo_kernel,
                                          // 4 conv3x3_halo_ws_kernel, 5 halo kernel on a zero-stuffed source, 6 stride-2 dgrad as four phase launches of the LDS-DMA kernel, 7 halo kernel with the folded 1x1 skip
                                          // convolution, 8 / 9 / 10 conv_subpixel_ws_kernel (upsample / transposed / upsample data gradient), 11 conv_wgrad_kernel, 12 conv_wgrad_slots_kernel, 13 conv_wgrad_slots_ws_kernel,
                                          // 14 / 15 conv1x1_pair_stream_kernel / conv1x1_wgrad_stream_kernel, 16 conv_wgrad_subpixel_ws_kernel,
                                          // 21 gn_silu_fwd_reg_kernel, 22 gn_silu_fwd_kernel, 23 gn_silu_bwd_hybrid_kernel, 24 gn_silu_bwd_kernel

#define GMK_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            gmk_set_error(__VA_ARGS__);        \
            return GMK_ERR_ARG;                \
        }                                      \
    } while (0)

// conv_halo.hip: 3x3 stride-1 convolution of 16-bit operands (dtype GMK_BF16 / GMK_F16) with an LDS-resident halo; returns 1 / 2 if launched, 0 if not eligible
int gmk_conv3x3_halo_try(const void* src0, const void* src1, int c0, int c1, int B, int H, int W, const void* w, int w_rows,
                         int n0, int cout, const float* bias, const float* emb, int emb_stride, const void* residual,
                         void* out, int out_cstride, int min_tiles, int upsample, float* stats, int64_t stats_bytes,
                         const float* gn_scale, const float* gn_shift, int gn_stride, int dtype, hipStream_t stream);

struct HaloGeometry { int R, TP, slots; int64_t rows_total, M, ntiles, nb0, nb1, nbw, nbo; };      // slots: halo slots a tile uses (<= 448)
int gmk_halo_geometry(int B, int H, int W, int c0, int c1, int w_rows, int cout, int out_cstride, int min_tiles, int upsample,
                      int fused_gn, HaloGeometry* g);      // conv_halo.hip: 1 if the halo kernels take this problem (g filled), else 0

// conv_wgrad_slots.hip: 3x3 stride-1 bf16 weight gradient over padded slots; returns the number of slabs written or 0
int gmk_wgrad_slots_nsplit(int cout, int ktot);
int gmk_conv_wgrad_slots_try(const void* dy, int dy_cstride, const void* src0, const void* src1, int c0, int c1, int B, int H,
                             int W, int cout, float* slab, int64_t slab_bytes, int forced, int upsample, bool x_f16,
                             hipStream_t stream, int stride2 = 0);


static inline hipStream_t gmk_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int gmk_esize(int dtype) { return dtype == GMK_F32 ? 4 : 2; }
static inline bool gmk_is16(int dtype) { return dtype == GMK_BF16 || dtype == GMK_F16; }

// ---- device helpers ---------------------------------------------------------------------------
template <typename T> struct Vec8;   // 8 activation elements as the natural 16/32-byte vector

__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
__device__ __forceinline__ void load8(const f16_t* p, float (&v)[8]) {
    const f16x8 a = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = a;
}
__device__ __forceinline__ void store8(f16_t* p, const float (&v)[8]) {
    f16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (f16_t)v[i];
    *reinterpret_cast<f16x8*>(p) = a;
}
__device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
}
__device__ __forceinline__ void load4(const bf16_t* p, float (&v)[4]) {
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)a[i];
}
__device__ __forceinline__ void load4(const f16_t* p, float (&v)[4]) {
    const f16x4 a = *reinterpret_cast<const f16x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)a[i];
}
__device__ __forceinline__ void store4(float* p, const float (&v)[4]) {
    f32x4 a = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = a;
}
__device__ __forceinline__ void store4(bf16_t* p, const float (&v)[4]) {
    bf16x4 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x4*>(p) = a;
}

__device__ __forceinline__ void store4(f16_t* p, const float (&v)[4]) {
    f16x4 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = (f16_t)v[i];
    *reinterpret_cast<f16x4*>(p) = a;
}

// ---- 16-bit storage types.  Forward activations and forward weight packs are fp16 (f16_t: 11 significant bits - the precision of the
// reference's own fp16 autocast, gms/diffusion/diffusion_model.py:68), gradients are bf16 (fp32's exponent range: no loss scaling).
// Both are 2 bytes, so every layout is shared; what differs is the pack / unpack arithmetic and the MFMA instruction.
template <typename T> struct Frag16;                      // 8 elements = one 16-byte MFMA operand fragment
template <> struct Frag16<bf16_t> { typedef bf16x8 type; typedef bf16x4 half_type; };
template <> struct Frag16<f16_t> { typedef f16x8 type; typedef f16x4 half_type; };
template <typename T> __device__ __forceinline__ f32x16 mfma_32x32x16(typename Frag16<T>::type a, typename Frag16<T>::type b, f32x16 c);
template <> __device__ __forceinline__ f32x16 mfma_32x32x16<bf16_t>(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
template <> __device__ __forceinline__ f32x16 mfma_32x32x16<f16_t>(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
template <typename T> __device__ __forceinline__ f32x4 mfma_16x16x32(typename Frag16<T>::type a, typename Frag16<T>::type b, f32x4 c);
template <> __device__ __forceinline__ f32x4 mfma_16x16x32<bf16_t>(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
template <> __device__ __forceinline__ f32x4 mfma_16x16x32<f16_t>(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
// the two 16-bit values of one dword as a float pair (low half first), and back (round to nearest even)
template <typename T> __device__ __forceinline__ void unpack_pair(unsigned r, float& lo, float& hi);
template <> __device__ __forceinline__ void unpack_pair<bf16_t>(unsigned r, float& lo, float& hi) {
    lo = __builtin_bit_cast(float, r << 16); hi = __builtin_bit_cast(float, r & 0xFFFF0000u);
}
template <> __device__ __forceinline__ void unpack_pair<f16_t>(unsigned r, float& lo, float& hi) {
    const f16x2 h = __builtin_bit_cast(f16x2, r);
    lo = (float)h[0]; hi = (float)h[1];
}
template <typename T> __device__ __forceinline__ unsigned pack_pair(float lo, float hi);
template <> __device__ __forceinline__ unsigned pack_pair<bf16_t>(float lo, float hi) { const bf16x2 t = {(bf16_t)lo, (bf16_t)hi}; return __builtin_bit_cast(unsigned, t); }
template <> __device__ __forceinline__ unsigned pack_pair<f16_t>(float lo, float hi) { const f16x2 t = {(f16_t)lo, (f16_t)hi}; return __builtin_bit_cast(unsigned, t); }
// Convolution outputs are unbounded: an fp16 store has to saturate at the largest finite value instead of producing inf.  The hardware
// does it for nothing: with MODE.FP16_OVFL set (bit 23 of the MODE register) every instruction that produces an fp16 result clamps an
// overflow to +-65504 (true infinities pass).  Kernels that store fp16 activations call fp16_saturating_stores<T>() once at their top;
// sat16 stays as the (now empty) marker of the places that rely on it.
template <typename T> __device__ __forceinline__ void fp16_saturating_stores() {}
template <> __device__ __forceinline__ void fp16_saturating_stores<f16_t>() { __builtin_amdgcn_s_setreg((23 << 6) | 1, 1); }      // hwreg(HW_REG_MODE, 23, 1)
template <typename T> __device__ __forceinline__ float sat16(float v) { return v; }

// v_exp_f32 + v_rcp_f32 (1 ulp each): an IEEE `1.0f / d` costs ~12 VALU issue slots more per element (div_scale / fma chain /
// div_fmas / div_fixup plus hazard nops), which made the GroupNorm kernels VALU-bound instead of HBM-bound
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float siluf_(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// ---- Philox4x32-10 counter RNG (gmk_rng_*, dropout masks): 4 x 32 bits per (counter, seed) ----------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4x32(uint64_t ctr, uint64_t seed, uint32_t (&out)[4]) {
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }   // [0, 1)


// Cross-lane reductions WITHOUT ds_bpermute (`__shfl_xor` lowers to ds_bpermute_b32, which runs through the LDS crossbar): DPP inside a
// 16-lane row and v_permlane{16,32}_swap across rows never touch the LDS pipeline, pair the same lanes as the xor butterfly
// (bit-identical sums) and are faster.  docs/EXPERIMENTS.md section 5 records the round-1/2 investigation of a GroupNorm miscompare that only a
// ds_bpermute build showed, including what the ISA of both builds looks like; no kernel in csrc/ uses ds_bpermute.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// a <- [r0 r0 r2 r2], b <- [r1 r1 r3 r3] for a == b == [r0 r1 r2 r3] (16-lane rows): (a, b) is the lane's {own, lane^16} pair in
// some order.  Inline asm: given the same value for both operands the builtin's two results are folded into one register.
__device__ __forceinline__ void rows_swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
// a <- [lo lo], b <- [hi hi] for a == b == [lo hi] (32-lane halves)
__device__ __forceinline__ void halves_swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }

// sum over the lanes whose index differs in bits >= log2(FROM) (FROM = 1: whole wave; FROM = 8: lanes sharing lane & 7); every
// lane ends with its total.  Same pairing as the butterfly `for off = 32 .. FROM: v += shfl_xor(v, off)`.
template <int FROM>
__device__ __forceinline__ float lanes_sum_from(float v) {
    float a = v, b = v;
    halves_swap32(a, b); v = a + b;
    a = v; b = v;
    rows_swap16(a, b); v = a + b;
    if (FROM <= 8) v += dpp_f32<0x128>(v);      // row_ror:8   (lane ^ 8)
    if (FROM <= 4) v += dpp_f32<0x124>(v);      // row_ror:4   (lane ^ 4 up to the halves made equal by the step before)
    if (FROM <= 2) v += dpp_f32<0x4E>(v);       // quad_perm [2,3,0,1]
    if (FROM <= 1) v += dpp_f32<0xB1>(v);       // quad_perm [1,0,3,2]
    return v;
}
// 64-lane wave sum / max (every lane ends with the result)
__device__ __forceinline__ float wave_sum(float v) { return lanes_sum_from<1>(v); }
__device__ __forceinline__ float wave_max(float v) {
    float a = v, b = v;
    halves_swap32(a, b); v = fmaxf(a, b);
    a = v; b = v;
    rows_swap16(a, b); v = fmaxf(a, b);
    v = fmaxf(v, dpp_f32<0x128>(v)); v = fmaxf(v, dpp_f32<0x124>(v));
    v = fmaxf(v, dpp_f32<0x4E>(v)); v = fmaxf(v, dpp_f32<0xB1>(v));
    return v;
}
