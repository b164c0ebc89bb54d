// Weight gradient of `Upsample` (reference gms/diffusion/simple_unet.py:112-122: F.interpolate(nearest, x2), then Conv2d(C, C, 3, padding=1)) in
// its sub-pixel form, on the slot correlation of conv_wgrad_slots.hip.
//
// The forward is Y[2i + a][2j + b] = sum_{ty, tx} W'_{ab, ty tx} . X[i + a - 1 + ty][j + b - 1 + tx] with the pre-summed 2x2-tap matrices W' of
// gmk_pack_upsample_weight, so the 16 tap gradients are
//     G_{ab, ty tx}[co][ci] = sum over low-resolution pixels (i, j) of dY[2i + a][2j + b][co] . X[i + a - 1 + ty][j + b - 1 + tx][ci]
// and dW[ky][kx] = the sum of the four G whose pre-sums contain 3x3 tap (ky, kx) (row 0 <- (a, ty) = (0, 0), (1, 0); row 1 <- (0, 1), (1, 0);
// row 2 <- (0, 1), (1, 1); columns alike): 16 tap-products per low-resolution pixel where the nearest-x2 slot kernel (`xshift`) multiplies 36.
//
// Slots are those of the LOW-resolution images ((H + 1) x (W + 1), shared zero border); X is the saved low-resolution activation, the dY
// operand of parity (a, b) is a stride-2 VIEW of the high-resolution gradient (pixel index 4 pix - 2 x + 2 a W + b for low-resolution pixel
// index pix, column x).  A workgroup (64 co x 64 ci, wave-specialised like conv_wgrad_slots_ws_kernel) runs ONE row parity a with both column
// parities: two dY streams and eight accumulators (b, ty, tx) per consumer wave, X offsets (a - 1 + ty)(W + 1) + (b - 1 + tx); the two row parities
// are two sets of workgroups of one launch.  Producers: everything through registers as in the kXF16 / kShare form of the slot kernel (X chunk
// re-rounded fp16 -> bf16 on its way into LDS when the activation is fp16; each chunk's slots decoded once, the dY issue LOOK steps later reuses
// the indices from a register FIFO) - six 16-byte loads per lane and step, five blocks ahead, counted `vmcnt`.  LDS: two dY rings of 4 x 8 KiB,
// the X ring (512 slots + 128 mirrored) = 144 KiB.  Slabs [row parity][split][8][cout][cin] fp32; a two-stage deterministic reduce (over the
// splits, then 16 -> 9 taps) writes the reference's [Cout][Cin][3][3] layout.
#include <type_traits>

#include "gmk_common.h"

namespace {

constexpr int kDyBase = 0;                  // 2 streams x 4 x 8 KiB
constexpr int kXBase = 65536;               // 512-slot ring + mirror of its first 128 slots, 128 B per slot
constexpr int kLdsBytes = kXBase + (512 + 128) * 128;
constexpr int LOOK = 1;                     // taps reach at most 64 slots: W + 2 <= 64

struct SubWgradParams {
    const void* dy; int dy_cstride;         // high resolution [B][2H][2W][dy_cstride]
    const void* x;                          // low resolution [B][H][W][cin]
    int cin, cout;
    int B, H, W, WE, RE;                    // low resolution; WE = W + 1, RE = H + 1
    float* slab;                            // [2][nsplit][8][cout][cin]
    int nchunks, chunks_per_split, nsplit;
    unsigned nbdy, nbx;
};

__device__ __forceinline__ unsigned mad24(unsigned a, unsigned b, unsigned c) {      // a * b + c on the low 24 bits of a and b (full rate)
    unsigned d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

template <bool kXF16>
__global__ __launch_bounds__(512, 2) void conv_wgrad_subpixel_ws_kernel(const SubWgradParams p) {
    __shared__ __attribute__((aligned(16))) char smem[kLdsBytes];
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    constexpr unsigned kBadPix = 0x00FFFFFFu;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int H = p.H, W = p.W, WE = p.WE, RE = p.RE;
    const int ncob = p.cout >> 6;
    const int cis = blockIdx.y, cob = (int)blockIdx.z % ncob, pa = (int)blockIdx.z / ncob;      // 64-ci tile, 64-co tile, row parity
    const int c_begin = blockIdx.x * p.chunks_per_split;
    const int c_end = min(c_begin + p.chunks_per_split, p.nchunks);
    if (c_begin >= c_end) return;
    const int nsteps = c_end - c_begin;

    if (wave >= 4) {
        // =========================================== producer waves ===========================================
        const int pw = wave - 4;
        const unsigned xs_b = (unsigned)p.cin * 2, xoff_b = (unsigned)cis * 128;
        const unsigned ys_b = (unsigned)p.dy_cstride * 2, yoff_b = (unsigned)cob * 128;
        // one instruction = 8 slots x 128 B; lane -> slot 8 (2 pw + u) + (lane >> 3) of the chunk, physical 16-B chunk lane & 7 holding
        // logical chunk (lane & 7) ^ (bit1(S) << 2), bit1(S) = bit 4 of the lane index (same swizzle for X and dY)
        const unsigned lc = (unsigned)(((lane & 7) ^ (((lane >> 4) & 1) << 2)) << 4);
        const float inv_re = 1.0f / (float)RE;
        const int d64r = 64 / WE, d64x = 64 % WE, d8r = 8 / WE, d8x = 8 % WE;
        auto advance = [&](int& row, int& xe, int dr, int dx) { xe += dx; row += dr; if (xe >= WE) { xe -= WE; ++row; } };
        // low-resolution pixel index of a slot (X operand) and the index of its parity-(0, 0) pixel in the high-resolution tensor (dY operand)
        auto pixel = [&](int row, int xe, unsigned& pix, unsigned& yb) {
            const int b = (int)(((float)row + 0.5f) * inv_re);           // exact for row < 2^22
            const int ye = row - __mul24(b, RE);
            const bool ok = b < p.B && ye >= 1 && xe >= 1;               // ye <= H and xe <= W hold by construction
            const unsigned rowpix = mad24((unsigned)b, (unsigned)H, (unsigned)(ye - 1));
            const unsigned px = mad24(rowpix, (unsigned)W, (unsigned)(xe - 1));
            pix = ok ? px : kBadPix;
            yb = ok ? 4u * px - 2u * (unsigned)(xe - 1) : kBadPix;
        };
        int xc = c_begin - LOOK, yc = c_begin;
        int xrow, xxe;
        {
            const int S = 64 * max(xc, 0) + 16 * pw + (lane >> 3);
            xrow = S / WE; xxe = S - xrow * WE;
        }
        unsigned hist[LOOK + 1][2];          // dY base indices of this lane's two slots of X chunks xc - 1 - LOOK .. xc - 1 (oldest first)
#pragma unroll
        for (int i = 0; i <= LOOK; ++i) { hist[i][0] = kBadPix; hist[i][1] = kBadPix; }
        auto x_pixels = [&](unsigned& pix0, unsigned& pix1) {      // decode the next X chunk's two slots of this lane (and remember them for dY)
            unsigned y0 = kBadPix, y1 = kBadPix;
            pix0 = kBadPix; pix1 = kBadPix;
            if (xc >= 0) {
                int r1 = xrow, x1 = xxe;
                advance(r1, x1, d8r, d8x);
                pixel(xrow, xxe, pix0, y0); pixel(r1, x1, pix1, y1);
                advance(xrow, xxe, d64r, d64x);
            }
#pragma unroll
            for (int i = 0; i < LOOK; ++i) { hist[i][0] = hist[i + 1][0]; hist[i][1] = hist[i + 1][1]; }
            hist[LOOK][0] = y0; hist[LOOK][1] = y1;
        };
        // descriptors as four scalars for the inline-asm loads (base, size, raw 32-bit format): a border slot's out-of-range offset reads zeros
        const unsigned long long xbase = (unsigned long long)p.x, ybase = (unsigned long long)p.dy;
        const i32x4 xdesc = {(int)(unsigned)xbase, (int)((unsigned)(xbase >> 32) & 0xFFFFu), (int)p.nbx, 0x00020000};
        const i32x4 ydesc = {(int)(unsigned)ybase, (int)((unsigned)(ybase >> 32) & 0xFFFFu), (int)p.nbdy, 0x00020000};
        // byte offset of parity (pa, b) inside the high-resolution tensor: scalar, rides in the loads' soffset
        const unsigned ypar0 = (unsigned)(2 * pa * W) * ys_b, ypar1 = ypar0 + ys_b;
        constexpr int FA = 5;                            // blocks issued ahead
        constexpr int NSET = FA - 1;                     // register sets: blocks s + 2 .. s + FA are live during step s
        u32x4 xr[NSET][2], y0r[NSET][2], y1r[NSET][2];
        int xrp[NSET], yrp[NSET];                        // ring positions the sets go to (wave-uniform)
        auto load_x = [&](u32x4 (&r)[2], int& rp) {      // 2 buffer loads to registers (inline asm: they stay in flight across barriers)
            unsigned pix0, pix1;
            x_pixels(pix0, pix1);
            const unsigned v0 = __umul24(pix0, xs_b) + xoff_b + lc, v1 = __umul24(pix1, xs_b) + xoff_b + lc;
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r[0]) : "v"(v0), "s"(xdesc) : "memory");
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r[1]) : "v"(v1), "s"(xdesc) : "memory");
            rp = xc & 7;
            ++xc;
        };
        auto load_y = [&](u32x4 (&r0)[2], u32x4 (&r1)[2], int& rp) {      // dY chunk yc of both column parities = the X chunk decoded LOOK issues ago
            const unsigned v0 = __umul24(hist[0][0], ys_b) + yoff_b + lc, v1 = __umul24(hist[0][1], ys_b) + yoff_b + lc;
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r0[0]) : "v"(v0), "s"(ydesc), "s"(ypar0) : "memory");
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r0[1]) : "v"(v1), "s"(ydesc), "s"(ypar0) : "memory");
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r1[0]) : "v"(v0), "s"(ydesc), "s"(ypar1) : "memory");
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r1[1]) : "v"(v1), "s"(ydesc), "s"(ypar1) : "memory");
            rp = yc & 3;
            ++yc;
        };
        auto store_y = [&](u32x4 (&r0)[2], u32x4 (&r1)[2], int rp) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                asm volatile("" : "+v"(r0[u]), "+v"(r1[u]));
                char* dst = smem + kDyBase + rp * 8192 + pw * 2048 + u * 1024 + lane * 16;
                *reinterpret_cast<u32x4*>(dst) = r0[u];
                *reinterpret_cast<u32x4*>(dst + 32768) = r1[u];
            }
        };
        auto store_x = [&](u32x4 (&r)[2], int rp) {       // (fp16 -> bf16,) ds_write_b128 (+ mirror)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                asm volatile("" : "+v"(r[u]));
                u32x4 o = r[u];
                if constexpr (kXF16) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        float lo, hi;
                        unpack_pair<f16_t>(r[u][d], lo, hi);
                        o[d] = pack_pair<bf16_t>(lo, hi);
                    }
                }
                char* dst = smem + kXBase + rp * 8192 + pw * 2048 + u * 1024 + lane * 16;
                *reinterpret_cast<u32x4*>(dst) = o;
                if (rp < 2 * LOOK) *reinterpret_cast<u32x4*>(dst + 65536) = o;
            }
        };
        // prologue: blocks 0 and 1 (X chunks c-LOOK .. c+LOOK+1, dY chunks c and c+1) through temporary registers, awaited and written: barrier 0
        // then sees every chunk up to c + 1; blocks 2 .. FA-1 follow and stay in flight.  (dY chunk k is issued right behind X chunk k + LOOK:
        // hist[0] is then X chunk k's decode.)
        {
            u32x4 t[2 * LOOK + 2][2], ta[2][2], tb[2][2]; int tp[2 * LOOK + 2], tyq[2];
#pragma unroll
            for (int k = 0; k < 2 * LOOK; ++k) load_x(t[k], tp[k]);
            load_x(t[2 * LOOK], tp[2 * LOOK]); load_y(ta[0], tb[0], tyq[0]);
            load_x(t[2 * LOOK + 1], tp[2 * LOOK + 1]); load_y(ta[1], tb[1], tyq[1]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 2 * LOOK + 2; ++k) store_x(t[k], tp[k]);
            store_y(ta[0], tb[0], tyq[0]);
            store_y(ta[1], tb[1], tyq[1]);
        }
#pragma unroll
        for (int k = 2; k < FA; ++k) {
            load_x(xr[k % NSET], xrp[k % NSET]);
            load_y(y0r[k % NSET], y1r[k % NSET], yrp[k % NSET]);
        }
        // Step s.  Block k (X chunk c + k + LOOK, dY chunks c + k) has to be in LDS at barrier k - 1.  Its ds_writes are issued during step k - 2
        // and only awaited at the top of step k - 1.
        //   lgkmcnt(0): block s + 1 is in LDS | barrier | issue block s + FA (set freed by the write of step s - 1) |
        //   vmcnt: everything but blocks s + 3 .. s + FA has landed | convert + write block s + 2
        auto step = [&](auto load_tag, auto store_tag) {
            constexpr int LSET = decltype(load_tag)::value, SSET = decltype(store_tag)::value;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            load_x(xr[LSET], xrp[LSET]);
            load_y(y0r[LSET], y1r[LSET], yrp[LSET]);
            static_assert(FA == 5, "the literal below is 6 x (FA - 2)");
            asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
            store_x(xr[SSET], xrp[SSET]);
            store_y(y0r[SSET], y1r[SSET], yrp[SSET]);
        };
        for (int s = 0; s < nsteps;) {        // load set (s + FA) % NSET = (s + 1) % 4, store set (s + 2) % 4: statically indexed
            step(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}); ++s;
            if (s < nsteps) { step(std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{}); ++s; }
            if (s < nsteps) { step(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{}); ++s; }
            if (s < nsteps) { step(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}); ++s; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        return;
    }

    // =============================================== consumer waves ===============================================
    const int wco = wave >> 1, wci = wave & 1;
    const int gg = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
    const int hh = gg >> 1, cblk = gg & 1;
    // transposed fragment reads, 16-lane group gg, lane-in-group 4q + pp: rows = slots 16 kk + 8 hh + 4 t + q of the chunk, the
    // 64-B half of a row is the wave's 32-channel block flipped by bit 1 of the slot index
    const int dy_lane = kDyBase + (8 * hh + q) * 128 + ((wco ^ ((q >> 1) & 1)) << 6) + cblk * 32 + pp * 8;
    int x_tap[8];                 // tap t = 4 b + 2 ty + tx: X at slot offset (pa - 1 + ty) WE + (b - 1 + tx)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int off = (pa - 1 + ((t >> 1) & 1)) * WE + ((t >> 2) - 1 + (t & 1));
        const int cls = (off + 64) & 3;
        x_tap[t] = kXBase + (8 * hh + q + 64 * LOOK + off) * 128 + ((wci ^ (((q + cls) >> 1) & 1)) << 6) + cblk * 32 + pp * 8;
    }
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    typedef __attribute__((ext_vector_type(8))) short s16x8;
    auto ld_a = [&](int c, int kk, int stream) -> bf16x8 {
        const char* yb = smem + stream * 32768 + (c & 3) * 8192 + dy_lane + (16 * kk) * 128;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)yb);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)(yb + 4 * 128));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    auto ld_b = [&](int c, int kk, int t) -> bf16x8 {
        const char* xb = smem + (((c - LOOK) & 7) << 13) + x_tap[t] + (16 * kk) * 128;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)xb);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)(xb + 4 * 128));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };

    bf16x8 a[2][2], b[4];         // a[column parity][kk & 1]
    __builtin_amdgcn_s_barrier();                 // step 0's barrier: chunk c_begin (and c_begin + 1) are in LDS
    a[0][0] = ld_a(c_begin, 0, 0); a[1][0] = ld_a(c_begin, 0, 1);
    b[0] = ld_b(c_begin, 0, 0); b[1] = ld_b(c_begin, 0, 1); b[2] = ld_b(c_begin, 0, 2);
    for (int c = c_begin; c < c_end; ++c) {
        // 32 units (kk, tap) per step; the X fragment of unit u + 3 and, at taps 2 / 5, the two dY fragments of the next kk are read under
        // the MFMA of unit u; the last three units read the first fragments of the NEXT step (valid already, see the producers)
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int kk = u >> 3, t = u & 7;
            if (u + 3 < 32) b[(u + 3) & 3] = ld_b(c, (u + 3) >> 3, (u + 3) & 7);
            else b[(u + 3) & 3] = ld_b(c + 1, 0, u + 3 - 32);
            if (t == 2) a[0][(kk + 1) & 1] = kk < 3 ? ld_a(c, kk + 1, 0) : ld_a(c + 1, 0, 0);
            if (t == 5) a[1][(kk + 1) & 1] = kk < 3 ? ld_a(c, kk + 1, 1) : ld_a(c + 1, 0, 1);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t >> 2][kk & 1], b[u & 3], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (t == 2 || t == 5) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            else __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        if (c + 1 < c_end) __builtin_amdgcn_s_barrier();      // the next step's barrier (one per producer iteration)
    }

    // ---- slab[row parity][split][tap][co][ci]
    const int r = lane & 31, h = lane >> 5;
    float* slab = p.slab + ((((int64_t)pa * p.nsplit + blockIdx.x) * 8) * p.cout + cob * 64 + wco * 32) * p.cin + cis * 64 + wci * 32 + r;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = (e & 3) + 8 * (e >> 2) + 4 * h;
            slab[((int64_t)t * p.cout + co) * p.cin] = acc[t][e];
        }
}

// stage 1: G[pa][t][co][ci] = sum over splits of slab[pa][split][t][co][ci]  (float4 lanes, fixed order: deterministic)
__global__ __launch_bounds__(256) void subpixel_split_reduce_kernel(const float* __restrict__ slab, float* __restrict__ G, int nsplit, int per) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // float4 index inside one [8][cout][cin] block
    const int pa = blockIdx.y;
    if (i * 4 >= per) return;
    const f32x4* s = reinterpret_cast<const f32x4*>(slab + (int64_t)pa * nsplit * per) + i;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < nsplit; ++k) acc += s[(int64_t)k * (per / 4)];
    reinterpret_cast<f32x4*>(G + (int64_t)pa * per)[i] = acc;
}

// stage 2: dW[co][ci][ky][kx] = sum of the four tap gradients whose pre-summed matrices contain 3x3 tap (ky, kx)
__global__ __launch_bounds__(256) void subpixel_taps_kernel(const float* __restrict__ G, float* __restrict__ dw, int n /* cout * cin */) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    float g[2][2][2][2];          // [a][b][ty][tx]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < 8; ++t) g[a][t >> 2][(t >> 1) & 1][t & 1] = G[((int64_t)a * 8 + t) * n + idx];
    float o[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            // (parity, tap) pairs along one axis whose pre-sum contains 3x3 index k: 0 <- (0, 0), (1, 0); 1 <- (0, 1), (1, 0); 2 <- (0, 1), (1, 1)
            const int ya0 = 0, yt0 = ky == 0 ? 0 : 1, ya1 = 1, yt1 = ky == 2 ? 1 : 0;
            const int xa0 = 0, xt0 = kx == 0 ? 0 : 1, xa1 = 1, xt1 = kx == 2 ? 1 : 0;
            o[3 * ky + kx] = (g[ya0][xa0][yt0][xt0] + g[ya0][xa1][yt0][xt1]) + (g[ya1][xa0][yt1][xt0] + g[ya1][xa1][yt1][xt1]);
        }
#pragma unroll
    for (int k = 0; k < 9; ++k) dw[(int64_t)idx * 9 + k] = o[k];
}

}  // namespace

static int wgrad_subpixel_plan(int B, int H, int W, int cin, int cout, int dy_cstride, int* nsplit, int* cps, int* nchunks) {
    if (cin != 128 || cout != 128 || dy_cstride < cout || dy_cstride > 2048) return 0;
    const int WE = W + 1, RE = H + 1;
    if (WE + 1 > 64 || W < 4 || H < 2) return 0;
    const int64_t M = (int64_t)B * H * W, total = (int64_t)B * RE * WE;
    if (4 * M >= 0x00FFFFFF || total >= (1ll << 22) * WE) return 0;          // 24-bit pixel indices of the high-resolution tensor; row < 2^22
    const int64_t lim = 0xFFFF0000ll;
    if (4 * M * dy_cstride * 2 >= lim || M * cin * 2 >= lim) return 0;
    const int nch = (int)((total + 63) / 64);
    int ns = gmk_cu_limit() / ((cout / 64) * (cin / 64) * 2);                  // both row parities in one launch
    if (ns >= 8) {                                                             // the 8 workgroups of a slot range sit on one XCD (a multiple of 8 splits) -
        const int all8 = ns & ~7, two = ns & ~3;                               // or on two (a multiple of 4) when the CU limit of a data-parallel run (248)
        ns = all8 * 16 < ns * 15 ? two : all8;                                 // would otherwise leave a fifth of the CUs without a workgroup
    }
    if (ns < 1) ns = 1;
    if (nch < 8 * ns) return 0;                                                // too little work per split: the nearest-x2 forms
    const int c = (nch + ns - 1) / ns;
    *cps = c; *nsplit = (nch + c - 1) / c; *nchunks = nch;
    return 1;
}

extern "C" int64_t gmk_conv_wgrad_subpixel_workspace_bytes(int B, int H, int W, int cin, int cout) {
    int ns, cps, nch;
    if (!wgrad_subpixel_plan(B, H, W, cin, cout, cout, &ns, &cps, &nch)) return 0;
    return ((int64_t)2 * ns * 8 + 16) * cout * cin * 4;
}

extern "C" int gmk_conv_wgrad_subpixel_ok(int B, int H, int W, int cin, int cout) {
    const int force = gmk_kernel_choice(1, "GMK_WGRAD_KERNEL");
    const char* e = getenv("GMK_SUBPIXEL");
    if (force != 0 || (e && e[0] == '0')) return 0;
    return gmk_conv_wgrad_subpixel_workspace_bytes(B, H, W, cin, cout) > 0;
}

extern "C" int gmk_conv_wgrad_subpixel(const void* dy, int dy_cstride, const void* x, int B, int H, int W, int cin, int cout, float* dw,
                                       void* workspace, int64_t workspace_bytes, int dtype, int x_dtype, void* stream) {
    GMK_REQUIRE(dy && x && dw && workspace, "gmk_conv_wgrad_subpixel: null pointer");
    GMK_REQUIRE(dtype == GMK_BF16 && (x_dtype == GMK_BF16 || x_dtype == GMK_F16),
                "gmk_conv_wgrad_subpixel: bf16 gradients with bf16 or fp16 activations (dtypes %d, %d)", dtype, x_dtype);
    int ns, cps, nch;
    GMK_REQUIRE(wgrad_subpixel_plan(B, H, W, cin, cout, dy_cstride, &ns, &cps, &nch),
                "gmk_conv_wgrad_subpixel: shape B=%d %dx%d cin=%d cout=%d is not eligible (ask gmk_conv_wgrad_subpixel_ok first)", B, H, W, cin, cout);
    const int64_t per = (int64_t)8 * cout * cin;
    GMK_REQUIRE(workspace_bytes >= ((int64_t)2 * ns * 8 + 16) * cout * cin * 4, "gmk_conv_wgrad_subpixel: workspace too small");
    SubWgradParams p;
    p.dy = dy; p.dy_cstride = dy_cstride; p.x = x; p.cin = cin; p.cout = cout;
    p.B = B; p.H = H; p.W = W; p.WE = W + 1; p.RE = H + 1;
    p.slab = (float*)workspace; p.nchunks = nch; p.chunks_per_split = cps; p.nsplit = ns;
    p.nbdy = (unsigned)((int64_t)4 * B * H * W * dy_cstride * 2); p.nbx = (unsigned)((int64_t)B * H * W * cin * 2);
    hipStream_t st = gmk_stream(stream);
    const dim3 grid(ns, cin / 64, (cout / 64) * 2);
    if (x_dtype == GMK_F16) conv_wgrad_subpixel_ws_kernel<true><<<grid, 512, 0, st>>>(p);
    else conv_wgrad_subpixel_ws_kernel<false><<<grid, 512, 0, st>>>(p);
    float* G = p.slab + (int64_t)2 * ns * per;
    subpixel_split_reduce_kernel<<<dim3((unsigned)((per / 4 + 255) / 256), 2), 256, 0, st>>>(p.slab, G, ns, (int)per);
    subpixel_taps_kernel<<<(cout * cin + 255) / 256, 256, 0, st>>>(G, dw, cout * cin);
    gmk_note_kernel(16);
    return gmk_check_launch("gmk_conv_wgrad_subpixel");
}
