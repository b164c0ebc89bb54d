// Weight gradient of the 3x3 stride-1 convolutions as a 1-D correlation over padded "slots" (bf16, MFMA).
// Reference: the autograd backward of nn.Conv2d(C, C, 3, padding=1) at gms/diffusion/simple_unet.py:163,172,117.
//
// Every image is viewed with a zero border above and to its left: (H+1) x (W+1) slots, slot S = (b*(H+1) + ye)*(W+1) + xe,
// pixel (y, x) at (ye, xe) = (y+1, x+1).  The border is SHARED: the slot right of a row's last pixel is the next row's
// border slot, the row below an image is the next image's border row (beyond the last image: out of range = zero) - 841
// instead of 900 slots per 28x28 image, 225 instead of 256 at 14x14, 64 instead of 81 at 7x7.  With that padding a filter
// tap is a CONSTANT slot offset off = (ky-1)*(W+1) + (kx-1), and
//     dW[tap][co][ci] = sum_S dY[S][co] * X[S + off][ci]          (dY and X read 0 in border slots)
// i.e. one long GEMM-like reduction over S in which the 9 taps reuse the same streamed data: each dY / X slot
// enters LDS exactly once per workgroup (the im2col formulation re-gathers X once per tap and is bound by the
// vector-memory -> LDS path).  Workgroup = 8 waves, output tile = 9 taps x 128 co x 64 ci: wave (wc, wi) owns
// co tile wc (32) x ci tile wi (32) for all 9 taps = 9 MFMA 32x32 accumulators (144 VGPRs); per 16 slots it reads
// one dY^T fragment and nine shifted X^T fragments with ds_read_b64_tr_b16 and issues 9 v_mfma_f32_32x32x16_bf16.
//
// LDS (135 KiB): dY ring 3 x 64 slots x 256 B; X ring 512 slots x 128 B whose first 128 slots are mirrored behind
// the end, so the window [S-64, S+128) of a 64-slot chunk never wraps and every fragment address is
// lane_base + chunk_base + IMMEDIATE.  Swizzles (applied to the DMA source chunk and to the reads): X slot rows flip
// their 64-B halves when bit 1 of S is set, dY rows rotate their 64-B windows by S&3 — both make the 4-row
// transposed reads conflict-free.  All loads are `buffer_load_dwordx4 ... lds`; border slots use an out-of-range
// offset (the descriptor's range check feeds zeros).  Per 64-slot step every wave issues exactly 4 DMA
// instructions (X chunk c+3 + its mirror-or-dummy, dY chunk c+2 x2), so `s_waitcnt vmcnt(4)` retires exactly what
// the step needs; one raw s_barrier per step.  Split-K over slot ranges into fp32 slabs; the deterministic reduce of
// conv_igemm.hip converts to the reference's [Cout][Cin][3][3] layout.
#include <type_traits>

#include "gmk_common.h"

namespace {

constexpr int kDyBase = 0;                  // 3 x 16 KiB
constexpr int kXBase = 49152;               // 512-slot ring + mirror of its first LOOK*128 slots, 128 B per slot
// LOOK = 1: taps reach at most 64 slots (W <= 62): X chunks c-1..c+1 are live, 128 mirrored slots (136 KiB of LDS).
// LOOK = 2: taps reach up to 128 slots (W <= 126, e.g. 64x64 images): chunks c-2..c+2, 256 mirrored slots (152 KiB).
template <int LOOK> struct SlotLds {
    static constexpr int kMirrorSlots = LOOK * 128;
    static constexpr int kScratch = kXBase + (512 + kMirrorSlots) * 128;    // 8 x 1 KiB sink for the dummy DMA of steps without a mirror copy
    static constexpr int kBytes = kScratch + 8192;
};

struct SlotParams {
    const void* dy; int dy_cstride;
    const void* src0; const void* src1;
    int c0, c1, ktot, cout;
    int B, H, W, WE, RE;         // WE = W + 1 slots per row, RE = H + 1 rows per image (borders shared, see the header)
    int xshift;                  // 1: X is the half-resolution tensor (nearest x2 upsample folded into the gather)
    int HS, WS, ntile;           // kS2 (stride-2 convolution): X is the (HS, WS) = (2 H, 2 W) tensor; ntile = 64-channel tiles of X (blockIdx.y = plane * ntile + tile)
    float* slab;                 // [nsplit][9][cout][ktot]
    int nchunks, chunks_per_split;
    unsigned nbdy, nb0, nb1;
    int variant;                 // GMK_DEV_VARIANT (experiments)
};

struct SlotPos { int b, ye, xe; };

__device__ __forceinline__ unsigned mad24(unsigned a, unsigned b, unsigned c) {      // a * b + c on the low 24 bits of a and b (full rate)
    unsigned d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

__device__ __forceinline__ SlotPos slot_decode(int S, int RE, int WE) {
    SlotPos p;
    const int rowi = S / WE;
    p.xe = S - rowi * WE;
    p.b = rowi / RE;
    p.ye = rowi - p.b * RE;
    return p;
}
__device__ __forceinline__ void slot_advance(SlotPos& p, int dxe, int dye, int RE, int WE) {   // by a fixed slot count, dxe < WE
    p.xe += dxe; p.ye += dye;
    if (p.xe >= WE) { p.xe -= WE; ++p.ye; }
    while (p.ye >= RE) { p.ye -= RE; ++p.b; }      // small images: 64 slots can span more than one image
}

template <int LOOK>
__global__ __launch_bounds__(512, 2) void conv_wgrad_slots_kernel(const SlotParams p) {
    __shared__ __attribute__((aligned(16))) char smem[SlotLds<LOOK>::kBytes];
    constexpr int kScratch = SlotLds<LOOK>::kScratch;
    constexpr unsigned kBadPix = 0x00FFFFFFu;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wc = wave >> 1, wi = wave & 1;
    const int H = p.H, W = p.W, WE = p.WE, RE = p.RE;
    const int cis = blockIdx.y, cob = blockIdx.z;
    const int c_begin = blockIdx.x * p.chunks_per_split;
    const int c_end = min(c_begin + p.chunks_per_split, p.nchunks);
    if (c_begin >= c_end) return;

    const int kelem0 = cis * 64;
    const bool second = kelem0 >= p.c0;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(second ? p.src1 : p.src0), 0, (int)(second ? p.nb1 : p.nb0), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, (int)p.nbdy, 0x00020000);
    const unsigned xs_b = (unsigned)(second ? p.c1 : p.c0) * 2;                     // bytes per X pixel
    const unsigned xoff_b = (unsigned)(second ? kelem0 - p.c0 : kelem0) * 2;
    const unsigned ys_b = (unsigned)p.dy_cstride * 2;
    const unsigned yoff_b = (unsigned)cob * 256;

    // ---- DMA lane geometry ------------------------------------------------------------------------------------
    // X: one instruction = 8 slots x 128 B; lane -> slot 8*wave + (lane>>3) of the chunk, physical chunk lane&7,
    //    which holds logical chunk (lane&7) ^ (bit1(S) << 2);  bit1(S) = bit 4 of the lane index
    const unsigned x_lc = (unsigned)(((lane & 7) ^ (((lane >> 4) & 1) << 2)) << 4);
    // dY: one instruction = 4 slots x 256 B; lane -> slot 8*wave + 4*i + (lane>>4), physical chunk lane&15 holding
    //    logical chunk (((pc>>2) ^ (S&3)) << 2) | (pc&3),  S&3 = (lane>>4)&3
    const unsigned y_lc = (unsigned)((((((lane & 15) >> 2) ^ ((lane >> 4) & 3)) << 2) | (lane & 3)) << 4);
    const int dxe64 = 64 % WE, dye64 = 64 / WE;

    auto pix_of = [&](const SlotPos& s) -> unsigned {
        const bool ok = s.b >= 0 && s.b < p.B && s.ye >= 1 && s.ye <= H && s.xe >= 1 && s.xe <= W;
        return ok ? (unsigned)((s.b * H + s.ye - 1) * W + s.xe - 1) : kBadPix;
    };
    auto xpix_of = [&](const SlotPos& s) -> unsigned {
        const bool ok = s.b >= 0 && s.b < p.B && s.ye >= 1 && s.ye <= H && s.xe >= 1 && s.xe <= W;
        const int sh = p.xshift;
        return ok ? (unsigned)((s.b * (H >> sh) + ((s.ye - 1) >> sh)) * (W >> sh) + ((s.xe - 1) >> sh)) : kBadPix;
    };

    // trackers: slot of this lane in the next X chunk / dY chunk to be issued
    int xc = c_begin - LOOK;                      // next X chunk index to issue
    int yc = c_begin;                             // next dY chunk index to issue
    SlotPos xpos = slot_decode(64 * max(xc, 0) + 8 * wave + (lane >> 3), RE, WE);
    SlotPos ypos = slot_decode(64 * yc + 8 * wave + (lane >> 4), RE, WE);

    auto issue_x = [&]() {
        unsigned pix = kBadPix;
        if (xc >= 0) {
            pix = xpix_of(xpos);
            slot_advance(xpos, dxe64, dye64, RE, WE);
        }
        const unsigned voff = __umul24(pix, xs_b) + xoff_b + x_lc;
        const int rp = xc & 7;
        GMK_LDS char* dst = (GMK_LDS char*)(smem + kXBase + rp * 8192 + wave * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (GMK_LDS void*)dst, 16, voff, 0, 0, 0);
        // mirror of the first ring chunks behind the end; other steps send a dummy (zero-fill) to the scratch sink so
        // that every step issues the same number of DMA instructions
        GMK_LDS char* dst2 = rp < 2 * LOOK ? (GMK_LDS char*)(smem + kXBase + 65536 + rp * 8192 + wave * 1024)
                                           : (GMK_LDS char*)(smem + kScratch + wave * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (GMK_LDS void*)dst2, 16, rp < 2 * LOOK ? voff : 0xFFFFFF00u, 0, 0, 0);
        ++xc;
    };
    int yslot3 = 0;                               // yc % 3 of the next dY chunk to issue
    auto issue_y = [&]() {
        SlotPos s1 = ypos;                        // the lane's slot of instruction i = 1 is 4 slots further
        slot_advance(s1, 4, 0, RE, WE);
        const unsigned p0 = pix_of(ypos), p1 = pix_of(s1);
        slot_advance(ypos, dxe64, dye64, RE, WE);
        GMK_LDS char* dst = (GMK_LDS char*)(smem + kDyBase + yslot3 * 16384 + wave * 2048);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsy, (GMK_LDS void*)dst, 16, __umul24(p0, ys_b) + yoff_b + y_lc, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsy, (GMK_LDS void*)(dst + 1024), 16, __umul24(p1, ys_b) + yoff_b + y_lc, 0, 0,
                                                 0);
        ++yc;
        yslot3 = yslot3 == 2 ? 0 : yslot3 + 1;
    };

    // ---- fragment read geometry (transposed reads: 16-lane group gg, lane-in-group 4q + pp) --------------------
    const int gg = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
    const int hh = gg >> 1, cblk = gg & 1;
    // dY^T fragment (A operand, rows = co): row 16kk + 8hh + 4t + q of the chunk, window wc rotated by q
    const int dy_lane = kDyBase + (8 * hh + q) * 256 + ((wc ^ q) << 6) + cblk * 32 + pp * 8;
    // X^T fragments (B operand, cols = ci): slot 64 + 16kk + 8hh + 4t + q + off relative to chunk c-1's ring position;
    // the 64-B half is wi ^ bit1(q + off): one lane base per tap (9 VGPRs), everything else is an immediate
    int x_tap[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int off = (t / 3 - 1) * WE + (t % 3 - 1);
        const int cls = (off + 64) & 3;
        x_tap[t] = kXBase + (8 * hh + q + 64 * LOOK + off) * 128 + ((wi ^ (((q + cls) >> 1) & 1)) << 6) + cblk * 32 + pp * 8;
    }

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // ---- prologue: X chunks c-LOOK .. c+LOOK, dY chunk c, then the "previous step" issue (X c+LOOK+1, dY c+1)
#pragma unroll
    for (int k = 0; k < 2 * LOOK + 1; ++k) issue_x();
    issue_y();
    issue_x();
    issue_y();

    int ycons3 = 0;                               // c % 3 relative to c_begin for the dY chunk consumed
    for (int c = c_begin; c < c_end; ++c) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue_x();                                // X chunk c+LOOK+2 (+ mirror / dummy)
        issue_y();                                // dY chunk c+2
        const char* ybase = smem + ycons3 * 16384 + dy_lane;
        const int xrot = ((c - LOOK) & 7) << 13;  // ring position of chunk c-LOOK
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            const s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)(ybase + (16 * kk) * 256));
            const s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)(ybase + (16 * kk + 4) * 256));
            const s16x8 av = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
            const bf16x8 a = __builtin_bit_cast(bf16x8, av);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const char* xb = smem + xrot + x_tap[t];
                const s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)(xb + (16 * kk) * 128));
                const s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)(xb + (16 * kk + 4) * 128));
                const s16x8 bv = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, bv), acc[t], 0, 0, 0);
            }
        }
        ycons3 = ycons3 == 2 ? 0 : ycons3 + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- slab[split][tap][co][ci]
    const int r = lane & 31, h = lane >> 5;
    float* slab = p.slab + (((int64_t)blockIdx.x * 9) * p.cout + cob * 128 + wc * 32) * p.ktot + kelem0 + wi * 32 + r;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = (e & 3) + 8 * (e >> 2) + 4 * h;
            slab[((int64_t)t * p.cout + co) * p.ktot] = acc[t][e];
        }
}

// ---------------------------------------------------------------------------------------------------------------
// Wave-specialised form of the same slot correlation (the shape that took the forward 3x3 kernel from 810 to 970 TFLOP/s):
// waves 4..7 only move data - every LDS-DMA instruction of a step and ALL the slot arithmetic behind it - waves 0..3 only
// compute, one per SIMD (w and w + 4 share a SIMD).  Why: in the kernel above a SIMD hosts two waves that each issue 36 MFMAs,
// 80 transposed fragment reads, 4 LDS-DMA instructions (60 - 185 issue cycles apiece) and ~150 VALU of slot arithmetic per
// 64-slot step - about 2,900 - 3,900 issue cycles per SIMD against 2,304 cycles of MFMA time, in lockstep on both sides of one
// barrier (MFMA busy 38 %).  A consumer cannot hold the 128 co x 64 ci x 9 taps tile alone (18 accumulators = 288 registers),
// so the workgroup tile is 64 co x 64 ci x 9 taps: consumer (wco, wci) owns one 32 x 32 block of all 9 taps (9 accumulators) and
// issues per step 36 MFMAs + 80 fragment reads and nothing else; twice as many workgroups split the slot range half as often,
// so the slab bytes stay what they were, and the X / dY streams of the workgroups that share them meet in L2.
// Pipeline: chunk c + AHEAD is issued during step c (AHEAD = 4, or 3 where 64-pixel rows need the wider X window) and the wait
// in front of barrier c leaves the last AHEAD - 2 issue blocks in flight, so every chunk up to c + 1 has landed at barrier c: all
// data of step c + 1 is already valid during step c, and the consumers read the next step's first fragments before its barrier.
// LDS: dY ring 8 x 64 slots x 128 B (the workgroup's 64-co half of every dY row, swizzled like X), X ring as above.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kWsDyBase = 0;                 // 8 x 8 KiB
constexpr int kWsXBase = 65536;              // 512-slot ring + mirror of its first LOOK*128 slots
template <int LOOK> struct SlotWsLds {
    static constexpr int kBytes = kWsXBase + (512 + LOOK * 128) * 128;
    static constexpr int kAhead = LOOK == 1 ? 4 : 3;          // X ring: chunks c-LOOK .. c+LOOK+AHEAD must fit 8
};

// kXF16: the activation operand X is stored as fp16 (forward activations of the 16-bit mode) while dY is bf16.  The producers then
// bring the X chunk through registers instead of LDS-DMA: two 16-byte global loads per lane and step, issued AHEAD steps early like the
// DMA they replace and counted by the same `s_waitcnt vmcnt`; when they have landed each dword is re-rounded fp16 -> bf16 (3 VALU) and
// written with ds_write_b128 to the place the DMA would have filled (and to the mirror copy - one load serves both).  The consumers,
// the dY path and the LDS layout are unchanged: the MFMA sees bf16 x bf16 as before.
// kShare (every launch without the nearest-x2 source): slot S of dY and slot S of X are the same pixel, and the X stream runs LOOK chunks
// ahead of the dY stream - so the producers decode each chunk's slots ONCE (for X) and the dY issue LOOK steps later reuses the pixel
// indices from a (LOOK + 1)-deep register FIFO.  Halves the producers' address arithmetic, which sits on the kernel's critical path
// (producers = DMA issue + ~90 VALU per step against the consumers' 36 MFMAs: adding 30 VALU cost 5 - 8 %, round 3).
// kS2 (round 6): the weight gradient of the STRIDE-2 convolution (Downsample, simple_unet.py:81,97,100) on the same machinery.  The slots are those of
// the LOW-resolution gradient dY (H x W = the convolution's output); the input X (2H x 2W) is seen as its four parity planes
//     P_ab[slot (ye, xe)] = X[2 (ye - 1) + a][2 (xe - 1) + b]          (zero in border slots),
// and filter row ky reads plane row-parity a = (ky != 1) at slot-row offset (ky == 0 ? -1 : 0), likewise the columns: every tap is again a CONSTANT
// slot offset into ONE plane - (1,1) in P_00; (1,0), (1,2) in P_01; (0,1), (2,1) in P_10; (0,0), (0,2), (2,0), (2,2) in P_11.  blockIdx.y carries
// (plane, 64-channel tile): a workgroup streams dY and its plane of X once (the producers gather the plane's pixels, nothing else changes for them)
// and its consumers form the plane's 1, 2 or 4 taps - 9 tap-products per output pixel in total, none of them with a zero (the im2col kernel it
// replaces re-gathered X once per tap: 2.25 reads of every input pixel through the vector-memory -> LDS path).
template <int LOOK, bool kXF16 = false, bool kShare = false, bool kS2 = false>
__global__ __launch_bounds__(512, kS2 ? 1 : 2) void conv_wgrad_slots_ws_kernel(const SlotParams p) {
    static_assert(!(kS2 && !kShare) && !(kS2 && LOOK != 1), "stride-2 planes: the slot is decoded once (kShare); offsets reach one row back");
    __shared__ __attribute__((aligned(16))) char smem[SlotWsLds<LOOK>::kBytes];
    constexpr unsigned kBadPix = 0x00FFFFFFu;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int H = p.H, W = p.W, WE = p.WE, RE = p.RE;
    const int plane = kS2 ? (int)blockIdx.y / p.ntile : 0;   // kS2: (row parity, column parity) of the X plane this workgroup reads
    const int cis = kS2 ? (int)blockIdx.y - plane * p.ntile : (int)blockIdx.y, cob = blockIdx.z;            // 64-ci tile, 64-co tile
    const int c_begin = blockIdx.x * p.chunks_per_split;
    const int c_end = min(c_begin + p.chunks_per_split, p.nchunks);
    if (c_begin >= c_end) return;
    const int nsteps = c_end - c_begin;
    constexpr int AHEAD = SlotWsLds<LOOK>::kAhead;

    if (wave >= 4) {
        // =========================================== producer waves ===========================================
        const int pw = wave - 4;
        const int kelem0 = cis * 64;
        const bool second = kelem0 >= p.c0;
        const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(second ? p.src1 : p.src0), 0, (int)(second ? p.nb1 : p.nb0), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, (int)p.nbdy, 0x00020000);
        const unsigned xs_b = (unsigned)(second ? p.c1 : p.c0) * 2;
        const unsigned xoff_b = (unsigned)(second ? kelem0 - p.c0 : kelem0) * 2;
        const unsigned ys_b = (unsigned)p.dy_cstride * 2;
        const unsigned yoff_b = (unsigned)cob * 128;
        // one instruction = 8 slots x 128 B; lane -> slot 8 (2 pw + u) + (lane >> 3) of the chunk, physical 16-B chunk lane & 7 holding
        // logical chunk (lane & 7) ^ (bit1(S) << 2), bit1(S) = bit 4 of the lane index (same swizzle for X and dY)
        const unsigned lc = (unsigned)(((lane & 7) ^ (((lane >> 4) & 1) << 2)) << 4);
        const float inv_we = 1.0f / (float)WE, inv_re = 1.0f / (float)RE;
        // position of this lane's slot u = 0 in the next chunk to issue, as (row index over all images, xe); branch-free advance
        auto decode = [&](int S, int& row, int& xe) { row = S / WE; xe = S - row * WE; };
        const int d64r = 64 / WE, d64x = 64 % WE, d8r = 8 / WE, d8x = 8 % WE;
        auto advance = [&](int& row, int& xe, int dr, int dx) { xe += dx; row += dr; if (xe >= WE) { xe -= WE; ++row; } };
        auto pixel = [&](int row, int xe, int shift) -> unsigned {
            const int b = (int)(((float)row + 0.5f) * inv_re);           // exact for row < 2^22
            const int ye = row - __mul24(b, RE);                         // (24-bit multiplies: full rate; every product is a pixel / row index < 2^24)
            const bool ok = b < p.B && ye >= 1 && xe >= 1;               // ye <= H and xe <= W hold by construction
            // v_mad_u32_u24 by hand: left to itself the compiler turns these into v_mul_lo_u32 / v_mad_u64_u32 (quarter rate), and every
            // producer VALU cycle is taken from the consumer wave on the same SIMD
            const unsigned rowpix = mad24((unsigned)b, (unsigned)(H >> shift), (unsigned)((ye - 1) >> shift));
            return ok ? mad24(rowpix, (unsigned)(W >> shift), (unsigned)((xe - 1) >> shift)) : kBadPix;
        };
        const unsigned plane_add = (unsigned)(p.WS * (plane >> 1) + (plane & 1));      // kS2: 2W a + b' (WS = 2W)
        int xc = c_begin - LOOK, yc = c_begin;
        int xrow, xxe, yrow, yxe;
        decode(64 * max(xc, 0) + 16 * pw + (lane >> 3), xrow, xxe);
        decode(64 * yc + 16 * pw + (lane >> 3), yrow, yxe);
        (void)inv_we;

        unsigned hist[LOOK + 1][2];          // kShare: pixel indices of this lane's two slots of X chunks xc - 1 - LOOK .. xc - 1 (oldest first)
#pragma unroll
        for (int i = 0; i <= LOOK; ++i) { hist[i][0] = kBadPix; hist[i][1] = kBadPix; }
        auto x_pixels = [&](unsigned& pix0, unsigned& pix1) {      // decode the next X chunk's two slots of this lane (and remember them)
            pix0 = kBadPix; pix1 = kBadPix;
            unsigned lo0 = kBadPix, lo1 = kBadPix;
            if (xc >= 0) {
                int r1 = xrow, x1 = xxe;
                advance(r1, x1, d8r, d8x);
                pix0 = pixel(xrow, xxe, kShare ? 0 : p.xshift); pix1 = pixel(r1, x1, kShare ? 0 : p.xshift);
                if constexpr (kS2) {       // the slot's own (low-resolution) pixel goes to the dY FIFO; X reads the plane's pixel behind it:
                    lo0 = pix0; lo1 = pix1;      // (b 2H + 2 (ye-1) + a) 2W + 2 (xe-1) + b' = 4 pixel - 2 (xe-1) + (2W a + b')
                    pix0 = lo0 == kBadPix ? kBadPix : 4u * lo0 - 2u * (unsigned)(xxe - 1) + plane_add;
                    pix1 = lo1 == kBadPix ? kBadPix : 4u * lo1 - 2u * (unsigned)(x1 - 1) + plane_add;
                }
                advance(xrow, xxe, d64r, d64x);
            }
            if constexpr (kShare) {
#pragma unroll
                for (int i = 0; i < LOOK; ++i) { hist[i][0] = hist[i + 1][0]; hist[i][1] = hist[i + 1][1]; }
                hist[LOOK][0] = kS2 ? lo0 : pix0; hist[LOOK][1] = kS2 ? lo1 : pix1;
            }
        };
        auto issue_x = [&]() -> int {      // -> number of DMA instructions (2, or 4 with the mirror copy)
            unsigned pix0, pix1;
            x_pixels(pix0, pix1);
            const unsigned v0 = __umul24(pix0, xs_b) + xoff_b + lc, v1 = __umul24(pix1, xs_b) + xoff_b + lc;
            const int rp = xc & 7;
            GMK_LDS char* dst = (GMK_LDS char*)(smem + kWsXBase + rp * 8192 + pw * 2048);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (GMK_LDS void*)dst, 16, v0, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (GMK_LDS void*)(dst + 1024), 16, v1, 0, 0, 0);
            ++xc;
            if (rp < 2 * LOOK) {           // wave-uniform: mirror of the first ring chunks behind the ring's end
                GMK_LDS char* dst2 = (GMK_LDS char*)(smem + kWsXBase + 65536 + rp * 8192 + pw * 2048);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (GMK_LDS void*)dst2, 16, v0, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (GMK_LDS void*)(dst2 + 1024), 16, v1, 0, 0, 0);
                return 4;
            }
            return 2;
        };
        auto y_pixels = [&](unsigned& p0, unsigned& p1) {
            if constexpr (kShare) {        // dY chunk yc = the X chunk decoded LOOK issues ago (every dY issue follows an X issue: hist[0])
                p0 = hist[0][0]; p1 = hist[0][1];
            } else {
                int r1 = yrow, x1 = yxe;
                advance(r1, x1, d8r, d8x);
                p0 = pixel(yrow, yxe, 0); p1 = pixel(r1, x1, 0);
                advance(yrow, yxe, d64r, d64x);
            }
        };
        auto issue_y = [&]() {
            unsigned p0, p1;
            y_pixels(p0, p1);
            GMK_LDS char* dst = (GMK_LDS char*)(smem + kWsDyBase + (yc & 7) * 8192 + pw * 2048);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsy, (GMK_LDS void*)dst, 16, __umul24(p0, ys_b) + yoff_b + lc, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsy, (GMK_LDS void*)(dst + 1024), 16, __umul24(p1, ys_b) + yoff_b + lc, 0, 0, 0);
            ++yc;
        };
        if constexpr (kXF16) {
            typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
            // blocks issued ahead (the X ring only holds WRITTEN chunks here, the dY ring has 8 slots).  kS2: its consumers have 4 - 16 MFMAs per
            // step, the step is the producers' - and with 4 blocks (64 KiB per CU) in flight it was the memory latency's (1,300 cycles per 16 KiB);
            // one workgroup per CU leaves the producers 256 registers: 8 blocks in flight
            constexpr int FA = kS2 ? 9 : 5;
            constexpr int NSET = FA - 1;                     // register sets: blocks s + 2 .. s + FA are live during step s
            // the X descriptor as four scalars for the inline-asm loads (same words as make_buffer_rsrc: base, size, raw 32-bit format): a
            // border slot's out-of-range offset reads zeros, as in the DMA form - no validity masks, 32-bit address arithmetic only
            typedef __attribute__((ext_vector_type(4))) int i32x4;
            const unsigned long long xbase = (unsigned long long)(second ? p.src1 : p.src0);
            const i32x4 xdesc = {(int)(unsigned)xbase, (int)((unsigned)(xbase >> 32) & 0xFFFFu), (int)(second ? p.nb1 : p.nb0), 0x00020000};
            u32x4 xr[NSET][2];
            int xrp[NSET];                                   // ring position the set goes to (wave-uniform)
            auto load_x = [&](u32x4 (&r)[2], int& rp) {      // 2 buffer loads to registers (inline asm: they stay in flight across barriers)
                unsigned pix0, pix1;
                x_pixels(pix0, pix1);
                const unsigned v0 = __umul24(pix0, xs_b) + xoff_b + lc, v1 = __umul24(pix1, xs_b) + xoff_b + lc;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r[0]) : "v"(v0), "s"(xdesc) : "memory");
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r[1]) : "v"(v1), "s"(xdesc) : "memory");
                rp = xc & 7;
                ++xc;
            };
            // dY through registers too (round 4).  The two ci-tiles of a split read the same dY half rows in step, on one XCD; LDS-DMA reads of one
            // line by two workgroups both reach the memory side, register loads are merged by the L2 (tools/fetch_probe.hip: counter / bytes 1.0
            // against 0.5).  With dY by `buffer_load ... lds` the kernel fetched X + 2 dY = 1.5 x its algorithmic bytes; through the same
            // register pipeline as X (loaded five blocks ahead, written to the dY ring two blocks ahead, no conversion) the corrected FETCH_SIZE of
            // the large launches drops by 37 % (2,092 -> 1,323 MB on round 4's four-shape mix, docs/EXPERIMENTS.md 7b.14) and the kernel runs 0.5 - 2.8 % faster, the
            // same bits in LDS (the DMA form was A/B'd behind a runtime switch and then removed: one more compare in the step loop for nothing).
            const unsigned long long ybase = (unsigned long long)p.dy;
            const i32x4 ydesc = {(int)(unsigned)ybase, (int)((unsigned)(ybase >> 32) & 0xFFFFu), (int)p.nbdy, 0x00020000};
            u32x4 yr[NSET][2];
            int yrp[NSET];
            auto load_y = [&](u32x4 (&r)[2], int& rp) {
                unsigned p0, p1;
                y_pixels(p0, p1);
                const unsigned v0 = __umul24(p0, ys_b) + yoff_b + lc, v1 = __umul24(p1, ys_b) + yoff_b + lc;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r[0]) : "v"(v0), "s"(ydesc) : "memory");
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r[1]) : "v"(v1), "s"(ydesc) : "memory");
                rp = yc & 7;
                ++yc;
            };
            auto store_y = [&](u32x4 (&r)[2], int rp) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    asm volatile("" : "+v"(r[u]));
                    *reinterpret_cast<u32x4*>(smem + kWsDyBase + rp * 8192 + pw * 2048 + u * 1024 + lane * 16) = r[u];
                }
            };
            auto store_x = [&](u32x4 (&r)[2], int rp) {       // fp16 -> bf16, ds_write_b128 (+ mirror)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    asm volatile("" : "+v"(r[u]));
                    u32x4 o;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        float lo, hi;
                        unpack_pair<f16_t>(r[u][d], lo, hi);
                        o[d] = pack_pair<bf16_t>(lo, hi);
                    }
                    char* dst = smem + kWsXBase + rp * 8192 + pw * 2048 + u * 1024 + lane * 16;
                    *reinterpret_cast<u32x4*>(dst) = o;
                    if (rp < 2 * LOOK) *reinterpret_cast<u32x4*>(dst + 65536) = o;
                }
            };
            // prologue: X chunks c-LOOK .. c+LOOK+1 through a temporary register set each, dY chunks c and c+1, all awaited and written:
            // barrier 0 then sees what the DMA form guarantees (every chunk up to c + 1); blocks 2 .. FA-1 follow and stay in flight
            {
                u32x4 t[2 * LOOK + 2][2]; int tp[2 * LOOK + 2];
#pragma unroll
                for (int k = 0; k < 2 * LOOK + 1; ++k) load_x(t[k], tp[k]);
                issue_y();
                load_x(t[2 * LOOK + 1], tp[2 * LOOK + 1]); issue_y();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < 2 * LOOK + 2; ++k) store_x(t[k], tp[k]);
            }
#pragma unroll
            for (int k = 2; k < FA; ++k) {
                load_x(xr[k % NSET], xrp[k % NSET]);
                load_y(yr[k % NSET], yrp[k % NSET]);
            }

            // Step s.  Block k (X chunk c + k + LOOK, dY chunk c + k) has to be in LDS at barrier k - 1.  Its ds_writes are issued during step
            // k - 2 and only awaited at the top of step k - 1: a whole step for them to drain through an LDS queue the consumers keep full
            // (waiting for them in front of the same step's barrier put that latency on the barrier: +7 ... 10 %).
            //   lgkmcnt(0): block s + 1 is in LDS | barrier | issue block s + FA (set freed by the write of step s - 1) |
            //   vmcnt: everything but blocks s + 3 .. s + FA has landed | convert + write block s + 2
            auto step = [&](auto load_tag, auto store_tag) {
                constexpr int LSET = decltype(load_tag)::value, SSET = decltype(store_tag)::value;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                load_x(xr[LSET], xrp[LSET]);
                load_y(yr[LSET], yrp[LSET]);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (FA - 2)) : "memory");
                store_x(xr[SSET], xrp[SSET]);
                store_y(yr[SSET], yrp[SSET]);
            };
            for (int s = 0; s < nsteps;) {        // load set (s + FA) % NSET = (s + 1) % NSET, store set (s + 2) % NSET: statically indexed
                step(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}); ++s;
                if (s < nsteps) { step(std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{}); ++s; }
                if (s < nsteps) { step(std::integral_constant<int, 3>{}, std::integral_constant<int, 4 % NSET>{}); ++s; }
                if constexpr (NSET == 4) {
                    if (s < nsteps) { step(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}); ++s; }
                } else {
                    static_assert(NSET == 4 || NSET == 8, "the unrolled set rotation below");
                    if (s < nsteps) { step(std::integral_constant<int, 4 % NSET>{}, std::integral_constant<int, 5 % NSET>{}); ++s; }
                    if (s < nsteps) { step(std::integral_constant<int, 5 % NSET>{}, std::integral_constant<int, 6 % NSET>{}); ++s; }
                    if (s < nsteps) { step(std::integral_constant<int, 6 % NSET>{}, std::integral_constant<int, 7 % NSET>{}); ++s; }
                    if (s < nsteps) { step(std::integral_constant<int, 7 % NSET>{}, std::integral_constant<int, 0>{}); ++s; }
                    if (s < nsteps) { step(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}); ++s; }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
        // prologue: X chunks c-LOOK .. c+LOOK + dY chunk c (step 0's data), one block per further chunk up to c + AHEAD - 1
#pragma unroll
        for (int k = 0; k < 2 * LOOK + 1; ++k) issue_x();
        issue_y();
        issue_x(); issue_y();                    // chunk c + 1 (step 1's data)
        int n2 = 0, n1 = 0;                      // instruction counts of the last two issue blocks (4, or 6 with a mirror copy)
        if (AHEAD == 4) { n2 = issue_x() + 2; issue_y(); }
        n1 = issue_x() + 2; issue_y();
        for (int s = 0; s < nsteps; ++s) {
            // everything but the last AHEAD - 2 issue blocks has to have landed
            const int n = n1 + n2;
            if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (n == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (AHEAD == 4) n2 = n1;
            n1 = issue_x() + 2;                    // chunk c + AHEAD: X (+ mirror), dY
            issue_y();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the run-ahead DMA before the LDS is released
        return;
    }

    // =============================================== consumer waves ===============================================
    const int wco = wave >> 1, wci = wave & 1;
    const int gg = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
    const int hh = gg >> 1, cblk = gg & 1;
    // transposed fragment reads, 16-lane group gg, lane-in-group 4q + pp: rows = slots 16 kk + 8 hh + 4 t + q of the chunk, the
    // 64-B half of a row is the wave's 32-channel block flipped by bit 1 of the slot index
    const int dy_lane = kWsDyBase + (8 * hh + q) * 128 + ((wco ^ ((q >> 1) & 1)) << 6) + cblk * 32 + pp * 8;
    if constexpr (kS2) {
        // the plane's taps: ky in (a ? {0, 2} : {1}), kx in (b ? {0, 2} : {1}); slot offset (ky == 0 ? -WE : 0) + (kx == 0 ? -1 : 0)
        const int pa = plane >> 1, pb = plane & 1;
        int x_off[4], tap_id[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ky = pa ? 2 * (t >> (pb ? 1 : 0) & 1) : 1, kx = pb ? 2 * (t & 1) : 1;
            const int off = (ky == 0 ? -WE : 0) + (kx == 0 ? -1 : 0);
            const int cls = (off + 64) & 3;
            x_off[t] = kWsXBase + (8 * hh + q + 64 * LOOK + off) * 128 + ((wci ^ (((q + cls) >> 1) & 1)) << 6) + cblk * 32 + pp * 8;
            tap_id[t] = ky * 3 + kx;
        }
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        auto ld_a = [&](int c, int kk) -> bf16x8 {
            const char* yb = smem + (c & 7) * 8192 + dy_lane + (16 * kk) * 128;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)yb);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)(yb + 4 * 128));
            const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(bf16x8, v);
        };
        auto ld_b = [&](int c, int kk, int xo) -> bf16x8 {
            const char* xb = smem + (((c - LOOK) & 7) << 13) + xo + (16 * kk) * 128;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)xb);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GMK_LDS s16x4*)(xb + 4 * 128));
            const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(bf16x8, v);
        };
        // NT taps per 16-slot k-block; the workgroup is bound by its data stream (16 KiB per step for 4 NT MFMAs), so the fragment reads are
        // simply issued one k-block ahead of their MFMAs (the next step's first block is valid already: see the header)
        auto run = [&](auto nt_tag) {
            constexpr int NT = decltype(nt_tag)::value;
            f32x16 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
            bf16x8 a[2], b[2][NT];
            __builtin_amdgcn_s_barrier();             // step 0's barrier
            a[0] = ld_a(c_begin, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) b[0][t] = ld_b(c_begin, 0, x_off[t]);
            for (int c = c_begin; c < c_end; ++c) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int cn = kk < 3 ? c : c + 1, kn = (kk + 1) & 3;
                    a[(kk + 1) & 1] = ld_a(cn, kn);
#pragma unroll
                    for (int t = 0; t < NT; ++t) b[(kk + 1) & 1][t] = ld_b(cn, kn, x_off[t]);
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk & 1], b[kk & 1][t], acc[t], 0, 0, 0);
                }
                if (c + 1 < c_end) __builtin_amdgcn_s_barrier();
            }
            const int r = lane & 31, h = lane >> 5;
            float* slab = p.slab + (((int64_t)blockIdx.x * 9) * p.cout + cob * 64 + wco * 32) * p.ktot + cis * 64 + wci * 32 + r;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int co = (e & 3) + 8 * (e >> 2) + 4 * h;
                    slab[((int64_t)tap_id[t] * p.cout + co) * p.ktot] = acc[t][e];
                }
        };
        if (plane == 3) run(std::integral_constant<int, 4>{});
        else if (plane == 0) run(std::integral_constant<int, 1>{});
        else run(std::integral_constant<int, 2>{});
        return;
    }
    int x_tap[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int off = (t / 3 - 1) * WE + (t % 3 - 1);
        const int cls = (off + 64) & 3;
        x_tap[t] = kWsXBase + (8 * hh + q + 64 * LOOK + off) * 128 + ((wci ^ (((q + cls) >> 1) & 1)) << 6) + cblk * 32 + pp * 8;
    }
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    typedef __attribute__((ext_vector_type(8))) short s16x8;
    auto ld_a = [&](int c, int kk) -> bf16x8 {
        const char* yb = smem + (c & 7) * 8192 + dy_lane + (16 * kk) * 128;
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

    bf16x8 a[2], b[4];
    __builtin_amdgcn_s_barrier();                 // step 0's barrier: chunk c_begin (and c_begin + 1) are in LDS
    a[0] = ld_a(c_begin, 0);
    b[0] = ld_b(c_begin, 0, 0); b[1] = ld_b(c_begin, 0, 1); b[2] = ld_b(c_begin, 0, 2);
    for (int c = c_begin; c < c_end; ++c) {
        // 36 units (kk, tap) per step; the X fragment of unit u + 3 (three MFMAs = 96 cycles ahead of its use) and, at tap 5, the dY
        // fragment of the next kk are read under the MFMA of unit u; the last three units read the first fragments of the NEXT
        // step (valid already, see the header)
#pragma unroll
        for (int u = 0; u < 36; ++u) {
            const int kk = u / 9, t = u % 9;
            if (u + 3 < 36) b[(u + 3) & 3] = ld_b(c, (u + 3) / 9, (u + 3) % 9);
            else b[(u + 3) & 3] = ld_b(c + 1, 0, u + 3 - 36);
            if (t == 5) a[(kk + 1) & 1] = kk < 3 ? ld_a(c, kk + 1) : ld_a(c + 1, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk & 1], b[u & 3], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (t == 5) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            else __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        if (c + 1 < c_end) __builtin_amdgcn_s_barrier();      // the next step's barrier (one per producer iteration)
    }

    // ---- slab[split][tap][co][ci]
    const int r = lane & 31, h = lane >> 5;
    float* slab = p.slab + (((int64_t)blockIdx.x * 9) * p.cout + cob * 64 + wco * 32) * p.ktot + cis * 64 + wci * 32 + r;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = (e & 3) + 8 * (e >> 2) + 4 * h;
            slab[((int64_t)t * p.cout + co) * p.ktot] = acc[t][e];
        }
}

}  // namespace

int gmk_wgrad_slots_nsplit(int cout, int ktot) {
    const int tiles = (cout / 128) * (ktot / 64);
    int ns = gmk_cu_limit() / tiles;
    return ns < 1 ? 1 : ns;
}

// Returns the number of splits written (>= 1) if the slot kernel was launched, 0 if the problem is not eligible.
// stride2: (H, W) is the convolution's OUTPUT (= dY) size, the source is (2H, 2W): the four-plane form of the wave-specialised kernel (kS2)
int gmk_conv_wgrad_slots_try(const void* dy, int dy_cstride, const void* src0, const void* src1, int c0, int c1, int B, int H,
                             int W, int cout, float* slab, int64_t slab_bytes, int forced, int upsample, bool x_f16,
                             hipStream_t stream, int stride2) {
    if (c0 % 64 || c1 % 64 || cout % 128) return 0;
    if (x_f16 && forced == 2) return 0;                      // the 8-compute-wave A/B kernel has no fp16-activation form
    if (stride2 && (upsample || c1 || forced == 2 || W + 2 > 64)) return 0;
    const int WE = W + 1, RE = H + 1;
    if (WE + 1 > 128 || W < 4 || H < 2) return 0;
    const bool wide = WE + 1 > 64;
    const int64_t M = (int64_t)B * H * W;
    const int64_t total = (int64_t)B * RE * WE;              // the last image's lower border lies beyond: b == B reads as zero
    if (M >= 0x00FFFFFF || total >= (1ll << 30)) return 0;
    if (upsample && ((H | W) & 1)) return 0;
    const int64_t Msrc = upsample ? M / 4 : stride2 ? M * 4 : M;
    if (Msrc >= 0x00FFFFFF) return 0;
    const int64_t nbdy = M * dy_cstride * 2, nb0 = Msrc * c0 * 2, nb1 = Msrc * c1 * 2;
    const int64_t lim = 0xFFFF0000ll;     // below (kBadPix * bytes-per-pixel) mod 2^32 for pixels of up to 4 KiB
    if (nbdy >= lim || nb0 >= lim || nb1 >= lim || c0 > 2048 || c1 > 2048 || dy_cstride > 2048) return 0;
    const int ktot = c0 + c1;
    const int nchunks = (int)((total + 63) / 64);
    int ns = gmk_wgrad_slots_nsplit(cout, ktot);
    if (nchunks < 8 * ns) {                                 // too little work per split: the im2col kernel
        if (!forced) return 0;
        ns = nchunks / 4 > 0 ? nchunks / 4 : 1;
    }
    const int cps = (nchunks + ns - 1) / ns;
    ns = (nchunks + cps - 1) / cps;
    if ((int64_t)ns * 9 * cout * ktot * 4 > slab_bytes) return 0;
    SlotParams p;
    p.dy = dy; p.dy_cstride = dy_cstride; p.src0 = src0; p.src1 = src1; p.c0 = c0; p.c1 = c1; p.ktot = ktot; p.cout = cout;
    p.xshift = upsample ? 1 : 0;
    p.HS = 2 * H; p.WS = 2 * W; p.ntile = ktot / 64;
    p.B = B; p.H = H; p.W = W; p.WE = WE; p.RE = RE; p.slab = slab; p.nchunks = nchunks; p.chunks_per_split = cps;
    p.nbdy = (unsigned)nbdy; p.nb0 = (unsigned)nb0; p.nb1 = (unsigned)nb1;
    p.variant = gmk_kernel_choice(3, "GMK_DEV_VARIANT") & 0xFF;
    dim3 grid(ns, ktot / 64, cout / 128);
    // forced: 0 automatic = 3 the wave-specialised kernel (1.37 - 1.58 x the 8-compute-wave kernel at every shape of the train step:
    // docs/EXPERIMENTS.md 7b.11); 2 the 8-compute-wave kernel (A/B)
    const bool ws = forced != 2;
    if (ws) {
        const int ytiles = (stride2 ? 4 : 1) * (ktot / 64);
        int ns3 = gmk_cu_limit() / ((cout / 64) * ytiles);
        if (ns3 >= 8) {                   // workgroups that stream the same dY / X are ns3 block ids apart: a multiple of 8 keeps them on one XCD
            const int all8 = ns3 & ~7, two = ns3 & ~3;       // (a multiple of 4: on two XCDs) - taken when the CU limit of a data-parallel
            ns3 = all8 * 16 < ns3 * 15 ? two : all8;      // run (248) would otherwise leave 10 % of the CUs without a workgroup (1.18 -> 1.25 PFLOP/s at 64 x 64)
        }
        if (ns3 < 1) ns3 = 1;
        if (nchunks < 8 * ns3) ns3 = nchunks / 8 > 0 ? nchunks / 8 : 1;
        const int cps3 = (nchunks + ns3 - 1) / ns3;
        ns3 = (nchunks + cps3 - 1) / cps3;
        if ((int64_t)ns3 * 9 * cout * ktot * 4 > slab_bytes) return 0;
        p.chunks_per_split = cps3;
        dim3 grid3(ns3, ytiles, cout / 64);
        if (stride2) {
            if (x_f16) conv_wgrad_slots_ws_kernel<1, true, true, true><<<grid3, 512, 0, stream>>>(p);
            else conv_wgrad_slots_ws_kernel<1, false, true, true><<<grid3, 512, 0, stream>>>(p);
            gmk_note_kernel(17);
            return ns3;
        }
        const bool share = !upsample;          // dY slot S and X slot S are the same pixel: decode once (kShare)
#define GMK_SLOT_WS(LK)                                                                                        \
    do {                                                                                                       \
        if (x_f16 && share) conv_wgrad_slots_ws_kernel<LK, true, true><<<grid3, 512, 0, stream>>>(p);         \
        else if (x_f16) conv_wgrad_slots_ws_kernel<LK, true, false><<<grid3, 512, 0, stream>>>(p);            \
        else if (share) conv_wgrad_slots_ws_kernel<LK, false, true><<<grid3, 512, 0, stream>>>(p);            \
        else conv_wgrad_slots_ws_kernel<LK, false, false><<<grid3, 512, 0, stream>>>(p);                      \
    } while (0)
        if (wide) GMK_SLOT_WS(2);
        else GMK_SLOT_WS(1);
#undef GMK_SLOT_WS
        gmk_note_kernel(13);
        return ns3;
    }
    if (wide) conv_wgrad_slots_kernel<2><<<grid, 512, 0, stream>>>(p);
    else conv_wgrad_slots_kernel<1><<<grid, 512, 0, stream>>>(p);
    gmk_note_kernel(12);
    return ns;
}
