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
    float* slab;                 // [nsplit][9][cout][ktot]
    int nchunks, chunks_per_split;
    unsigned nbdy, nb0, nb1;
};

struct SlotPos { int b, ye, xe; };

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

}  // namespace

int gmk_wgrad_slots_nsplit(int cout, int ktot) {
    const int tiles = (cout / 128) * (ktot / 64);
    int ns = gmk_cu_limit() / tiles;
    return ns < 1 ? 1 : ns;
}

// Returns the number of splits written (>= 1) if the slot kernel was launched, 0 if the problem is not eligible.
int gmk_conv_wgrad_slots_try(const void* dy, int dy_cstride, const void* src0, const void* src1, int c0, int c1, int B, int H,
                             int W, int cout, float* slab, int64_t slab_bytes, int forced, int upsample, hipStream_t stream) {
    if (c0 % 64 || c1 % 64 || cout % 128) return 0;
    const int WE = W + 1, RE = H + 1;
    if (WE + 1 > 128 || W < 4 || H < 2) return 0;
    const bool wide = WE + 1 > 64;
    const int64_t M = (int64_t)B * H * W;
    const int64_t total = (int64_t)B * RE * WE;              // the last image's lower border lies beyond: b == B reads as zero
    if (M >= 0x00FFFFFF || total >= (1ll << 30)) return 0;
    if (upsample && ((H | W) & 1)) return 0;
    const int64_t Msrc = upsample ? M / 4 : M;
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
    p.B = B; p.H = H; p.W = W; p.WE = WE; p.RE = RE; p.slab = slab; p.nchunks = nchunks; p.chunks_per_split = cps;
    p.nbdy = (unsigned)nbdy; p.nb0 = (unsigned)nb0; p.nb1 = (unsigned)nb1;
    dim3 grid(ns, ktot / 64, cout / 128);
    if (wide) conv_wgrad_slots_kernel<2><<<grid, 512, 0, stream>>>(p);
    else conv_wgrad_slots_kernel<1><<<grid, 512, 0, stream>>>(p);
    return ns;
}
