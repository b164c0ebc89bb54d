// 3x3 stride-1 convolution with an LDS-resident input halo — the kernel the train step is dominated by
// (24 ResBlock convolutions forward + their data gradients; reference gms/diffusion/simple_unet.py:163,172).
//
// Why: the im2col-style kernels (conv_igemm.hip) re-fetch every input pixel 9 times (once per tap) through the
// vector-memory -> LDS path, which tops out near 32 B/clk/CU and bounds them at ~900 TFLOP/s.  Here each input pixel
// enters LDS once per tile: a tile is R whole image rows (R*W <= 256 output pixels, rows taken from the global row
// list b*H + y so tiles may span images), and its input rows plus a zero/neighbour halo live in LDS as "slots" of
// 64 channels (128 B).  The 9 taps are 9 different constant slot offsets of the same LDS image, so the MFMA pixel
// operand is read straight from the halo with per-lane addresses; only the weight tiles (16 KB per K-step) stream.
//
// LDS map (160 KiB, one workgroup per CU, 8 waves):
//   2 x 448 slots x 128 B   halo half-buffers: channels [64*kh, 64*kh+64) of one source; while buffer hb is
//                           consumed (9 taps = 9 K-steps) the next 64-channel half streams into hb^1
//   3 x 16 KiB              ring of weight tiles [128 channels][64 k], two K-steps ahead
// Slot s of a tile = ext-row * (W+1) + x + 1, ext rows = the tile's rows with a pad row above/below every image
// segment; with E = ext-row + y0 (y0 = first row's position in its image) the layout is (H+2)-periodic:
// image k = E / (H+2), y = E % (H+2) - 1.  ONE pad column per row (round 6; two before): the slot in front of a row's first pixel is
// also the slot behind the previous row's last pixel - both are zero - so a tile needs ext-rows * (W+1) + 1 slots and 28 x 28
// (13 x 29 + 1 = 378) and 8 x 8 (361) fit the 384 slots of the 6-piece forms (kRing4, kMerge).  Pad rows / columns and rows of
// non-existent images read a zero through the buffer descriptor's range check.  Swizzle: physical chunk c of slot s holds logical chunk
// c ^ ((n>>1)&7), n = s - ext-row - 1 (the slot index with the pad column of every row taken out; applied to the DMA source address and to
// the fragment reads; a pad slot is read under two keys, as the left pad of its row and the right pad of the row above: it holds zeros in every chunk).
// For any tap the 16 pixels a ds_read_b128 lane group touches have CONSECUTIVE n even where they wrap to the next image
// row (their slots jump by 2 there), so every group is conflict-free; keying the swizzle on s itself cost 29 % extra LDS
// cycles at W = 28 (two 2-way conflicts in every group that contains a wrap).
//
// Schedule: K-step q = (phase, tap); each step every wave issues the 2 DMA instructions of weight tile q+2 and, for
// taps 0..6, one 1-KiB piece (8 slots) of the next phase's halo.  The counts are static, so `s_waitcnt vmcnt(N)`
// (N = ops issued in the previous step, +8 while the epilogue's 8 stores are younger) retires exactly the data the
// step needs; one raw s_barrier per step.  Persistent workgroups walk the tiles; the next tile's first halo half and
// weight tiles are already in flight during the epilogue, which runs straight from the accumulators (same
// D[channel][pixel] orientation + v_permlane32_swap widening as conv_igemm_dma_kernel).
#include "gmk_common.h"


namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

constexpr int kHaloSlots = 448;
constexpr int kHB = kHaloSlots * 128;          // 57344 bytes per halo half-buffer
constexpr int kWOFF = 2 * kHB;                 // weight ring offset
constexpr int kWST = 16384;

struct HaloParams {
    const void* src0; const void* src1;
    int c0, c1, ktot;
    int B, H, W, WE, R, TP, ntiles, rows_total;
    int nfull, nhalf;                          // wave-specialised kernel: tiles [0, nfull) are whole jobs, the rest run as nhalf half-channel jobs (see there)
    int shift;                                 // 1: the source is the half-resolution tensor (nearest x2 upsample folded in)
    int pmask;                                 // 1 (with shift = 1): only even (y, x) exist - the zero-stuffed source of a transposed conv
    const void* w; unsigned w_tap_stride_b; int n0;
    const float* bias; const float* emb; int emb_stride;
    const void* residual; void* out; int out_cstride; int M;
    unsigned nb0, nb1, nbw, nbo;
    const float* gn_scale; const float* gn_shift; int gn_stride;      // fused GroupNorm-apply + SiLU on the SOURCE: y = silu(x * scale[b][k] + shift[b][k])
    float* stats;                              // optional GroupNorm partial sums [ntiles][8][2][Cout/4][2] (see gmk.h)
    int stats_groups;                          // Cout/4
    int variant;                               // GMK_DEV_VARIANT (experiments)
    float inv_hp2, inv_h, inv_we, inv_w;       // reciprocals for exact small-integer division
    // folded 1x1 skip convolution (kSkip kernels): out += wsk[n][:] . cat(sk0, sk1)[pixel][:] + bias2[n]
    const void* sk0; const void* sk1; const void* wsk; const float* bias2;
    int cs, sk_ktot, nsk0;                     // channels per skip source (128), total skip K (2 cs), first weight row
    unsigned nbs, nbws;                        // bytes of one skip source / of the skip weight pack
    unsigned* stamps;                          // development (kStamp instantiations, gmk_dev_set_stamp_buffer): per-workgroup cycle sums per K-step
};

// first of the two 8-KiB LDS slots (pixels 0..127 / 128..255) that hold dense sub-phase m (0..3) of a halo phase's four: slots of the
// half-buffer that is being FILLED during that phase (see the kSkip kernel's header)
__host__ __device__ constexpr int skip_slot(int m) { return m == 3 ? 0 : 2 * m; }

// sum over the 32 lanes of a half-wave with DPP; lanes 16..31 (and 48..63) end up holding the half's total
__device__ __forceinline__ float half_wave_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, true));   // row_bcast15 into rows 1 and 3
    return v;
}

__device__ __forceinline__ int div_small(int a, float inv) { return (int)(((float)a + 0.5f) * inv); }

template <typename T>
__global__ __launch_bounds__(512, 2) void conv3x3_halo_kernel(const HaloParams p) {
    typedef typename Frag16<T>::type frag_t;
    typedef typename Frag16<T>::half_type half_t;
    constexpr int ES = 2;
    fp16_saturating_stores<T>();
    __shared__ __attribute__((aligned(16))) char smem[kWOFF + 3 * kWST];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int nblk = blockIdx.y * 128;
    const int H = p.H, W = p.W, WE = p.WE;
    const int nph = p.ktot >> 6;                 // 64-channel phases per tile
    constexpr unsigned kBadPix = 0x00FFFFFFu;
    constexpr unsigned kBadOff = 0xFFFFFF00u;

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src0), 0, (int)p.nb0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.c1 ? p.src1 : p.src0), 0, (int)(p.c1 ? p.nb1 : p.nb0), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.nbw, 0x00020000);
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.nbo, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.residual ? p.residual : p.out), 0, (int)p.nbo, 0x00020000);
    const __amdgpu_buffer_rsrc_t rss = __builtin_amdgcn_make_buffer_rsrc(
        p.stats ? (void*)p.stats : p.out, 0, p.stats ? p.ntiles * 8 * 2 * p.stats_groups * 2 * 4 : 0, 0x00020000);

    // ---- per-lane constants -------------------------------------------------------------------------------
    const int lrow = lane >> 3, lch = lane & 7;
    // halo fill: piece j covers slots j*64 + wave*8 + lrow; swizzle term is independent of j
    const int fs0 = wave * 8 + lrow;
    const int f_er0 = div_small(fs0, p.inv_we), f_xe0 = fs0 - f_er0 * WE;
    const int f_der = div_small(64, p.inv_we), f_dxe = 64 - f_der * WE;
    unsigned f_swz = 0;      // (n>>1)&7 of this lane's slot in each of the 7 fill pieces, 3 bits per piece (tile independent)
    {
        int er = f_er0, xe = f_xe0;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int n = p.variant == 1 ? er * WE + xe : er * WE + xe - er - 1;
            f_swz |= (unsigned)((n >> 1) & 7) << (3 * j);
            er += f_der; xe += f_dxe;
            if (xe >= WE) { xe -= WE; ++er; }
        }
    }
    // weight tile rows: 16*wave + 8*i + lrow
    unsigned w_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 16 * wave + 8 * i + lrow;
        w_off[i] = (unsigned)(p.n0 + nblk + row) * (unsigned)p.ktot * ES + (unsigned)((lch ^ ((row >> 1) & 7)) << 4);
    }
    // MFMA pixel rows of this lane: m_local = wm*64 + i*32 + r  ->  (row in tile, x)
    int rit[2], px_x[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ml = wm * 64 + i * 32 + r;
        rit[i] = div_small(ml, p.inv_w);
        px_x[i] = ml - rit[i] * W;
    }
    unsigned hpix[7];       // source pixel of this lane's slot in each fill piece (tile whose fills are being issued)
    int cslot[2];           // centre slot of this lane's two MFMA pixel rows (tile being computed)
    int cn[2];              // ... and its pad-free index n = slot - 2*ext-row - 1 (the swizzle key)

    auto resolve_fill = [&](int tile) {
        const bool exists = tile < p.ntiles;
        const int gr0 = tile * p.R;
        const int b0 = gr0 / H;
        const int y0 = gr0 - b0 * H;
        int er = f_er0, xe = f_xe0;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int E = er + y0;
            const int k = div_small(E, p.inv_hp2);
            const int y = E - k * (H + 2) - 1;
            const int x = xe - 1;
            const int b = b0 + k;
            const bool ok = exists && y >= 0 && y < H && x >= 0 && x < W && b < p.B;
            hpix[j] = ok && !((y | x) & p.pmask) ? (unsigned)((b * (H >> p.shift) + (y >> p.shift)) * (W >> p.shift) + (x >> p.shift)) : kBadPix;
            er += f_der; xe += f_dxe;
            if (xe >= WE) { xe -= WE; ++er; }
        }
    };
    auto resolve_centres = [&](int tile) {
        const int gr0 = tile * p.R;
        const int b0 = gr0 / H;
        const int y0 = gr0 - b0 * H;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = y0 + rit[i];
            const int k = div_small(t, p.inv_h);
            const int y = t - k * H;
            const int er = k * (H + 2) + y + 1 - y0;
            int s = er * WE + px_x[i] + 1;
            int n = p.variant == 1 ? s : s - er - 1;
            if (s < WE + 1 || s >= kHaloSlots - WE - 1) { s = WE + 1; n = W; }   // dead rows (m_local >= TP): any in-range slot
            cslot[i] = s;
            cn[i] = n;
        }
    };

    // fill piece j of phase `ph` (of the tile hpix describes) into halo buffer hbuf
    auto issue_fill = [&](int hbuf, int ph, int j) {
        const int kelem = ph << 6;
        const bool second = kelem >= p.c0;
        const unsigned cs_b = (unsigned)(second ? p.c1 : p.c0) * ES;
        const unsigned koff_b = (unsigned)(second ? kelem - p.c0 : kelem) * ES;
        GMK_LDS char* dst = (GMK_LDS char*)(smem + hbuf * kHB + j * 8192 + wave * 1024);
        const unsigned f_ch = (unsigned)((lch ^ ((f_swz >> (3 * j)) & 7)) << 4);
        const unsigned voff = __umul24(hpix[j], cs_b) + koff_b + f_ch;
        if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (GMK_LDS void*)dst, 16, voff, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (GMK_LDS void*)dst, 16, voff, 0, 0, 0);
    };
    // weight tile (tap, phase) into ring slot `stage`
    auto issue_w = [&](int stage, int tap, int ph) {
        GMK_LDS char* dst = (GMK_LDS char*)(smem + kWOFF + stage * kWST + wave * 2048);
        const unsigned wk = (unsigned)tap * p.w_tap_stride_b + ((unsigned)ph << 7);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (GMK_LDS void*)(dst + i * 1024), 16, w_off[i] + wk, 0, 0, 0);
    };

    f32x16 acc[2][2];     // [j: channel tile][i: pixel tile]
    // a tile's accumulators START from the bias (as in the wave-specialised consumers, whose first MFMA takes it as its C operand: the two
    // kernels stay bit-identical, tests/test_gpu_ops.py::test_halo_tail_runs_as_half_jobs); 32 values per lane, kept for the whole launch
    float bzv[2][16];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            float t[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) load4(p.bias + nblk + wn * 64 + j * 32 + 8 * q4 + 4 * h, t);
#pragma unroll
            for (int e = 0; e < 4; ++e) bzv[j][4 * q4 + e] = t[e];
        }
    const int swz = (r >> 1) & 7;
    const int b_off = kWOFF + (wn * 64 + r) * 128;

    auto compute = [&](int hbuf, int st, int tapoff, int tapoff_n) {
        const char* Hb = smem + hbuf * kHB;
        const char* Wb = smem + st * kWST + b_off;
        int rowb[2], sw[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int c = cslot[i], n = cn[i];
            asm volatile("" : "+v"(c), "+v"(n));      // keep the per-tap addresses out of loop-invariant hoisting (72 VGPRs)
            rowb[i] = (c + tapoff) << 7;
            sw[i] = ((n + tapoff_n) >> 1) & 7;
        }
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            frag_t px[2], wt[2];
            px[0] = *reinterpret_cast<const frag_t*>(Hb + rowb[0] + (((kg * 2 + h) ^ sw[0]) << 4));
            px[1] = *reinterpret_cast<const frag_t*>(Hb + rowb[1] + (((kg * 2 + h) ^ sw[1]) << 4));
            const int coff = ((kg * 2 + h) ^ swz) << 4;
            wt[0] = *reinterpret_cast<const frag_t*>(Wb + coff);
            wt[1] = *reinterpret_cast<const frag_t*>(Wb + 4096 + coff);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[j][i] = mfma_32x32x16<T>(wt[j], px[i], acc[j][i]);
        }
    };

    // residual tile of this lane (same layout as its accumulators), prefetched three K-steps before the epilogue so its
    // HBM latency hides under the last MFMA steps instead of stalling the epilogue (and draining the next tile's DMA)
    u32x2_t resv[2][2][4];
    auto prefetch_residual = [&](int tile) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ml = wm * 64 + i * 32 + r;
            const int m = tile * p.TP + ml;
            const bool live = ml < p.TP && m < p.M;
            const unsigned row_b = (unsigned)m * (unsigned)p.out_cstride * ES;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cb = nblk + wn * 64 + j * 32;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const unsigned off = live ? row_b + (unsigned)(cb + 8 * q4 + 4 * h) * ES : kBadOff;
                    resv[i][j][q4] = __builtin_amdgcn_raw_buffer_load_b64(rsr, off, 0, 0);
                }
            }
        }
    };

    auto epilogue = [&](int tile) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ml = wm * 64 + i * 32 + r;
            const int m = tile * p.TP + ml;
            const bool live = ml < p.TP && m < p.M;
            const unsigned row_b = (unsigned)m * (unsigned)p.out_cstride * ES;
            const float* embp = nullptr;
            int bsamp = 0;
            if (p.emb || p.stats) bsamp = (live ? m : 0) / (H * W);
            if (p.emb) embp = p.emb + (int64_t)bsamp * p.emb_stride;
            // GroupNorm statistics: this 32-pixel group may straddle two samples (never three: 32 < H*W)
            int b_first = 0;
            bool hi_seg = false, straddle = false;
            if (p.stats) {
                b_first = __builtin_amdgcn_readfirstlane(bsamp);          // lane 0 = first pixel of the group
                hi_seg = live && bsamp != b_first;
                straddle = __any(hi_seg);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cb = nblk + wn * 64 + j * 32;
                float v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = acc[j][i][e];          // (the bias is in the accumulators)
                if (p.emb) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        float t[4];
                        load4(embp + cb + 8 * q4 + 4 * h, t);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * q4 + e] += t[e];
                    }
                }
                if (p.residual) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const half_t rb = __builtin_bit_cast(half_t, resv[i][j][q4]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * q4 + e] += (float)rb[e];
                    }
                }
                unsigned pk[4][2];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    pk[q4][0] = pack_pair<T>(sat16<T>(v[4 * q4]), sat16<T>(v[4 * q4 + 1]));
                    pk[q4][1] = pack_pair<T>(sat16<T>(v[4 * q4 + 2]), sat16<T>(v[4 * q4 + 3]));
                }
                if (p.stats) {
                    // per-lane sum / sum of squares of the ROUNDED values of each 4-channel unit, reduced over the group's
                    // pixels per sample segment; lanes 16..31 / 48..63 then write 16 floats each with ONE store
                    float lo[4][2], hi[4][2];
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const half_t t = __builtin_bit_cast(half_t, (u32x2_t){pk[q4][0], pk[q4][1]});
                        const float a0 = (float)t[0], a1 = (float)t[1], a2 = (float)t[2], a3 = (float)t[3];
                        const float sm = live ? (a0 + a1) + (a2 + a3) : 0.f;
                        const float sq = live ? (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3) : 0.f;
                        lo[q4][0] = half_wave_sum(hi_seg ? 0.f : sm);
                        lo[q4][1] = half_wave_sum(hi_seg ? 0.f : sq);
                        hi[q4][0] = 0.f; hi[q4][1] = 0.f;
                        if (straddle) {
                            hi[q4][0] = half_wave_sum(hi_seg ? sm : 0.f);
                            hi[q4][1] = half_wave_sum(hi_seg ? sq : 0.f);
                        }
                    }
                    const int idx = r - 16;                       // writer lanes: slot = idx>>3, q4 = (idx>>1)&3, comp = idx&1
                    float val = 0.f;
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
                        for (int cpt = 0; cpt < 2; ++cpt) {
                            if (idx == q4 * 2 + cpt) val = lo[q4][cpt];
                            if (idx == 8 + q4 * 2 + cpt) val = hi[q4][cpt];
                        }
                    const int pg = wm * 2 + i;
                    const int slot = idx >> 3, q4w = (idx >> 1) & 3, cpt = idx & 1;
                    const int grp = (cb >> 2) + 2 * q4w + h;
                    const unsigned soff = idx >= 0
                        ? (unsigned)(((((tile * 8 + pg) * 2 + slot) * p.stats_groups) + grp) * 2 + cpt) * 4u
                        : kBadOff;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rss, soff, 0, 0);
                }
#pragma unroll
                for (int q4 = 0; q4 < 4; q4 += 2) {
                    const auto s0 = __builtin_amdgcn_permlane32_swap(pk[q4][0], pk[q4 + 1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(pk[q4][1], pk[q4 + 1][1], false, false);
                    u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                    const unsigned off = live ? row_b + (unsigned)(cb + 8 * (q4 + h)) * ES : kBadOff;
                    __builtin_amdgcn_raw_buffer_store_b128(o, rso, off, 0, 0);
                }
            }
        }
    };

    int st = 0, sq = 2;      // weight ring: slot computed next / slot issued next
    int hbuf = 0;            // halo buffer of the phase computed next
    int fresh = 0;           // K-steps left whose data is older than the epilogue's 8 stores
    int tile = blockIdx.x;
    if (tile >= p.ntiles) return;

    // ---- prologue: whole first halo half + the first two weight tiles
    resolve_fill(tile);
#pragma unroll
    for (int j = 0; j < 7; ++j) issue_fill(0, 0, j);
    issue_w(0, 0, 0);
    issue_w(1, 1, 0);

    for (; tile < p.ntiles; tile += gridDim.x) {
        resolve_centres(tile);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][i][e] = bzv[j][e];
        for (int ph = 0; ph < nph; ++ph) {
            const bool last_ph = ph + 1 == nph;
            if (last_ph) resolve_fill(tile + gridDim.x);        // the next fills belong to the next tile (or are zeros)
            const int ph_next = last_ph ? 0 : ph + 1;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                // ---- wait: retire everything but the ops issued in the previous step (+ the epilogue's stores)
                if (tap == 0) {
                    if (fresh > 0) {            // 8 output stores (+ 4 statistics stores) are younger than this step's data
                        --fresh;
                        if (p.stats) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                    } else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                } else if (tap == 1) {
                    if (fresh > 0) {
                        --fresh;
                        if (p.stats) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
                    } else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                } else if (tap == 7) {      // the 16 residual loads issued behind tap 6's DMA are younger than this step's data
                    if (last_ph && p.residual) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                } else if (tap == 8) {
                    if (last_ph && p.residual) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                // ---- issue: weight tile of step q+2, and one piece of the next phase's halo
                if (tap < 7) issue_w(sq, tap + 2, ph);
                else issue_w(sq, tap - 7, ph_next);
                if (tap < 7) issue_fill(hbuf ^ 1, ph_next, tap);
                if (tap == 6 && last_ph && p.residual) prefetch_residual(tile);
                // ---- compute
                compute(hbuf, st, (tap / 3 - 1) * WE + (tap % 3 - 1), (tap / 3 - 1) * (p.variant == 1 ? WE : W) + (tap % 3 - 1));
                st = st == 2 ? 0 : st + 1;
                sq = sq == 2 ? 0 : sq + 1;
            }
            hbuf ^= 1;
        }
        asm volatile("" ::: "memory");
        epilogue(tile);
        asm volatile("" ::: "memory");
        fresh = 2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the speculative DMA before the LDS is released
}


template <int N> struct IntTag { static constexpr int value = N; };

// ---------------------------------------------------------------------------------------------------------------
// Wave-specialised variant of the same convolution (same LDS map, same schedule): waves 4..7 only move data (every
// LDS-DMA instruction of a K-step: 4 weight pieces + 2 halo pieces each), waves 0..3 only compute, each owning a
// 128-pixel x 64-channel quarter of the tile (8 accumulators).  Why: in the kernel above every wave pays the issue cost
// of 3 LDS-DMA instructions (60-185 cycles each), 16 ds_read_b128 and ~25 address VALU per K-step next to its 16 MFMAs,
// and with two such waves per SIMD the issue port, not the MFMA pipe, sets the K-step (1430-1570 cycles against 1024 of
// MFMA time, measured with per-tile s_memtime stamps in round 1).  Here a SIMD hosts one consumer (32 MFMAs + 24 ds_read_b128 per K-step: 0.75 reads
// per MFMA instead of 1) and one producer, w and w+4 share a SIMD, and the consumer's epilogue stores no longer sit in the
// same vmcnt queue as the DMA (no `fresh` bookkeeping).  One s_barrier per K-step, crossed by all 8 waves.
//
// kSkip (round 4): `skip_connection(x) + h` of the up-path ResBlocks (reference simple_unet.py:174-186) as ONE launch.  conv2's K grows from
// 9 x 128 to 9 x 128 + 256: a tile also contracts the 256 channels of torch.cat([x, skip]) (two 128-channel sources, never concatenated)
// with the 1x1 skip weights, centre tap only.  Those K-steps have no tap reuse, so their pixel operand is laid out DENSE (no halo), in
// sub-phases of 32 channels: 256 pixels x 64 B = 16 KiB = two 8-KiB slots + an 8-KiB weight tile [128 rows][32 k] - eight short K-steps
// E0..E7 of half the MFMA work.  They are INTERLEAVED with the halo taps, four per halo phase, in the slots of the half-buffer that phase is
// filling (which stay free until the next phase's pieces are issued - later than in the plain kernel, but still two K-steps ahead):
//   K-steps of phase ph (reads buffer ph, fills buffer ph ^ 1):   t0 t1 t2 t3 e0 t4 e1 t5 e2 t6 e3 t7 t8      (e_m = sub-phase 4 ph + m)
//   pixel pieces issued in:   t0: E(4ph+0) -> slots 0,1    t1: E(4ph+1) -> 2,3    t2: E(4ph+2) -> 4,5    t3: NEXT[6] -> 6
//                             t4: E(4ph+3) -> 0,1 (e0 is done)    t5: NEXT[2,3] -> 2,3    t6: NEXT[4,5] -> 4,5    t7: NEXT[0,1] -> 0,1
//   (NEXT = the next halo phase's pieces: phase 1 of this tile, or phase 0 of the next tile), weights of the step after next in every step.
// Every dense piece is issued at least four K-steps before its step and every slot is re-used only behind the barrier that follows its last
// read.  (The first form of the round put the eight dense steps in ONE block in front of the halo phases: 17 of a tile's 31 pieces then had
// to arrive within eight 512-cycle steps, an issue distance of two short steps could not cover the HBM latency, and the block took 8.0 us
// per tile for 2.0 us of MFMA work - 979 -> 796 us per launch at 32 x 32, B = 2048; docs/EXPERIMENTS.md section 7b.2 has both measurements.)

// 64-byte rows: lane (row = lane >> 2, chunk = lane & 3) of a DMA instruction writes 16 rows of 64 B; physical chunk c of row n holds logical
// chunk c ^ ((n >> 2) & 3), so the 16 rows a ds_read_b128 lane group touches (consecutive n, one logical chunk) cover all 64 banks.
// No residual in this form (the skip convolution is the residual); half jobs as in the plain kernel (64 weight rows per step).
// kStamp (development, tools/step_stamps.py): wave 4 (a producer) and wave 0 (a consumer) of every workgroup sum, per K-step index, the shader
// cycles (s_memtime) they spend (producer) issuing / in the counted vmcnt wait / in the barrier, (consumer) working / in the barrier, and
// write the sums to p.stamps[workgroup][2][64] at the end: who waits for whom in each step.  Product launches use kStamp = false.
//
// kMerge (round 6; kSkip kernels whose tile needs at most 384 halo slots = 6 fill pieces: 32 x 32 and 16 x 16): the eight dense sub-phases run
// INSIDE the K-step of the tap in front of them - 9 barriers per phase instead of 13.  The step stamps (tools/step_stamps.py) showed a K-step
// costing ~ 1,650 cycles whatever its MFMA work (512 or 1,024 cycles): barrier jitter and the fragment-read latency exposed behind every
// barrier, not the arithmetic.  What made the extra barriers necessary was the weight ring (three 16-KiB slots: tap q, tap q + 1 prefetched,
// tap q + 2 in flight - no room for a dense tile); with 6-piece halos the SEVENTH 8-KiB piece of each half-buffer is free, and the dense weight
// tiles (8 KiB) alternate between the two of them: e0, e2 in the piece of the buffer being filled, e1, e3 in that of the buffer being read.
//   K-steps of phase ph:   t0 t1 t2 [t3 e0] [t4 e1] [t5 e2] [t6 e3] t7 t8
//   issued in            t0: W(e0), E0 -> pieces 0,1   t1: E1 -> 2,3   t2: E2 -> 4,5   t3: W(e1)   t4: W(e2), E3 -> 0,1 (e0 is done)
//                        t5: W(e3), NEXT[2,3]   t6: NEXT[4,5]   t7: NEXT[0,1]      (+ the tap weights of the step after next in every step)
// Tap weights first, dense weights second, pixel pieces last in every step: only the step's four pixel instructions may still fly at the next
// barrier (`vmcnt(4)`; `vmcnt(0)` at t0 and at t4, whose dense weights were issued one step earlier).  In the consumers the dense fragments take the
// fragment set that held the tap's first k half (free after its first four groups), so the sets swap roles behind every dense sub-phase.
// kRes: -1 the residual's presence is a run-time property of the launch (p.residual); 0 / 1 compiled in: the 16x16x32 consumers' epilogue
// then has no branch per 16-byte store (the blocks of a tile interleave freely); the launcher picks 0 / 1 for the plain form, 0 for the folded.
// kRing4 (round 6; plain form without a residual whose tile needs at most 384 halo slots): the two halo half-buffers shrink to 6 pieces (48 KiB)
// and the freed 16 KiB become a FOURTH weight-ring slot; the producers issue the weight tile of step q + 3 in step q.  With three slots
// (tap q being read, q + 1 landed for the consumers' prefetch, q + 2 in flight) the tile issued in step q - 1 had to LAND before barrier q + 1:
// the producers spent 150 - 370 cycles of every 1,650-cycle K-step in that wait and reached the barrier about as late as the consumers, so
// every step paid for the slower of the two (tools/step_stamps.py).  With four slots a weight tile has two steps to land, the producers'
// step is their issue time alone, and the barrier waits for the consumers only.
template <typename T, bool kPrefetchW, int kShape = 32, bool kFuse = false, bool kSkip = false, bool kStamp = false, bool kMerge = false, int kRes = -1, bool kRing4 = false>
__global__ __launch_bounds__(512, 2) void conv3x3_halo_ws_kernel(const HaloParams p) {
    static_assert(!(kSkip && kRes > 0), "the folded skip convolution has no residual (the skip convolution is the residual)");
    static_assert(!kSkip || (kPrefetchW && kShape == 16 && !kFuse), "the folded skip convolution is built on the plain 16x16x32 form");
    static_assert(!kMerge || kSkip, "merged dense sub-phases belong to the folded skip convolution");
    static_assert(!kRing4 || (kPrefetchW && kShape == 16 && !kFuse && !kSkip && kRes == 0), "the four-slot weight ring is built for the plain 16x16x32 form without a residual");
    constexpr int HB = kRing4 ? 6 * 8192 : kHB;          // bytes of a halo half-buffer
    constexpr int WOFF = 2 * HB;                         // the weight ring behind them
    constexpr int RING = kRing4 ? 4 : 3;                 // its slots (WOFF + RING * kWST = 160 KiB either way)
    static_assert(!kStamp || (kPrefetchW && kShape == 16 && !kFuse), "stamps exist for the shipped 16x16x32 forms only");
    typedef typename Frag16<T>::type frag_t;
    typedef typename Frag16<T>::half_type half_t;
    constexpr int ES = 2;
    fp16_saturating_stores<T>();
    __shared__ __attribute__((aligned(16))) char smem[kWOFF + 3 * kWST];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int nblk = blockIdx.y * 128;
    const int H = p.H, W = p.W, WE = p.WE;
    const int nph = p.ktot >> 6;                 // 64-channel phases per tile
    constexpr unsigned kBadPix = 0x00FFFFFFu;
    constexpr unsigned kBadOff = 0xFFFFFF00u;
    if ((int)blockIdx.x >= p.ntiles && (int)blockIdx.x >= p.nhalf) return;
    if constexpr (kStamp) {          // (experiment, GMK_DEV_VARIANT=22: every second workgroup of an XCD starts ~ half a tile late, so that the epilogues' store bursts of the two halves of the chip do not coincide)
        if (p.variant == 22 && ((blockIdx.x >> 3) & 1)) {
            const unsigned long long t0 = __builtin_readcyclecounter();
            while (__builtin_readcyclecounter() - t0 < 18000ull) __builtin_amdgcn_s_sleep(8);
        }
    }
    // Jobs of this workgroup, in order: the whole tiles g, g + G, ... below nfull, then - for the first nhalf workgroups - one
    // HALF job: channel half (g & 1) of tile nfull + g/2, computed by all four consumers as 64 pixels x 64 channels each.
    // The host sets nhalf = 2 x (ntiles mod G) when that remainder fits (<= G/2): the last, partly filled round of whole tiles
    // (0.45 of a round at 28x28, 0.11 at 14x14 for B = 1024) then costs about half a tile time on all CUs instead of a whole one.
    const int G = gridDim.x, g = blockIdx.x;
    // XCD-aware order of the whole tiles: workgroup ids round-robin over the 8 XCDs (each with its own L2), neighbouring tiles share two
    // halo rows (a quarter of a tile's input at 8 rows per tile), so the tiles of a round are dealt out in 8 contiguous runs, one per XCD:
    // the shared rows are fetched from HBM once per XCD run instead of once per tile
    const int pg = (G & 7) == 0 && p.variant != 9 ? (g & 7) * (G >> 3) + (g >> 3) : g;
    const int nk_full = (p.nfull - pg + G - 1) / G;
    const int njobs = nk_full + (g < p.nhalf ? 1 : 0);
    auto job_tile = [&](int k) { return k < nk_full ? pg + k * G : (k < njobs ? p.nfull + (g >> 1) : p.ntiles); };
    auto job_half = [&](int k) { return k >= nk_full && k < njobs ? (g & 1) : -1; };

    if (wave >= 4) {
        // =========================================== producer waves ===========================================
        // Per K-step each of the 4 producers issues 4 weight pieces (tile q+2) and, behind taps 0..6, 2 halo pieces of the next
        // phase — weights FIRST: vmcnt retires in issue order, and the step only needs the weight tile issued two steps ago, so
        // with the halo pieces queued behind the weights of their step they may stay in flight one step longer (HBM latency)
        // without holding up the barrier.
        // Every VALU instruction a producer issues is taken from the consumer wave on its SIMD (measured, round 3: 32 extra producer VALU per
        // K-step cost the kernel 7 - 10 %), so the steady-state loop is kept scalar: the wave index is made an SGPR (LDS destinations / M0 become
        // SALU), per-K-step parts of the DMA addresses (tap, phase) ride in the instructions' scalar offset, per-lane parts are precomputed.
        const int pw = __builtin_amdgcn_readfirstlane(wave) - 4;
        const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src0), 0, (int)p.nb0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs1 =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.c1 ? p.src1 : p.src0), 0, (int)(p.c1 ? p.nb1 : p.nb0), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.nbw, 0x00020000);
        const int lrow = lane >> 3, lch = lane & 7;
        // halo fill: piece j = slots [64j, 64j+64); this wave's two instructions cover slots 64j + (2pw+u)*8 + lrow
        const int f_der = div_small(64, p.inv_we), f_dxe = 64 - f_der * WE;
        int f_er0[2], f_xe0[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int fs0 = (pw * 2 + u) * 8 + lrow;
            f_er0[u] = div_small(fs0, p.inv_we);
            f_xe0[u] = fs0 - f_er0[u] * WE;
        }
        // weight tile rows: 32*pw + 8*u + lrow
        unsigned w_off[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = 32 * pw + 8 * u + lrow;
            w_off[u] = (unsigned)(p.n0 + nblk + row) * (unsigned)p.ktot * ES + (unsigned)((lch ^ ((row >> 1) & 7)) << 4);
        }
        // half jobs (channel half c of a tile) only need weight rows c*64 .. c*64+63: rows c*64 + 16*pw + 8*u + lrow, u < 2
        unsigned wh_off0[2], wh_off1[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r0 = 16 * pw + 8 * u + lrow, r1 = 64 + r0;
            wh_off0[u] = (unsigned)(p.n0 + nblk + r0) * (unsigned)p.ktot * ES + (unsigned)((lch ^ ((r0 >> 1) & 7)) << 4);
            wh_off1[u] = (unsigned)(p.n0 + nblk + r1) * (unsigned)p.ktot * ES + (unsigned)((lch ^ ((r1 >> 1) & 7)) << 4);
        }
        // source pixel (24 bits) and swizzled chunk offset (bits 24..31, = f_ch >> 4) of this lane's slot in piece j of the
        // tile being filled; resolved one piece at a time, right before the piece is first needed
        unsigned hpix[7][2];
        unsigned hvoff[7][2];           // plain path: the slot's byte offset in its source (pixel x row bytes + swizzled chunk): the DMA's whole vector offset
        const unsigned row_b = (unsigned)p.c0 * ES;      // bytes per source pixel (both sources have c0 channels: host-checked)
        auto resolve_piece = [&](int tile, int j) {
            const bool exists = tile < p.ntiles;
            const int gr0 = tile * p.R;
            const int b0 = gr0 / H;
            const int y0 = gr0 - b0 * H;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                // slot = 64j + fs0: advance (er, xe) of fs0 by j * 64 slots
                int xe = f_xe0[u] + j * f_dxe, er = f_er0[u] + j * f_der;
                const int wraps = div_small(xe, p.inv_we);
                xe -= wraps * WE; er += wraps;
                const int n = er * WE + xe - er - 1;
                const int E = er + y0;
                const int k = div_small(E, p.inv_hp2);
                const int y = E - k * (H + 2) - 1;
                const int x = xe - 1;
                const int b = b0 + k;
                const bool ok = exists && y >= 0 && y < H && x >= 0 && x < W && b < p.B;
                const unsigned pix = ok && !((y | x) & p.pmask) ? (unsigned)((b * (H >> p.shift) + (y >> p.shift)) * (W >> p.shift) + (x >> p.shift)) : kBadPix;
                if constexpr (kFuse) hpix[j][u] = pix | ((unsigned)(lch ^ ((n >> 1) & 7)) << 24) | ((unsigned)(k & 1) << 28);      // bit 28: sample b0 + k (fused GroupNorm)
                else hvoff[j][u] = __umul24(pix, row_b) + ((unsigned)(lch ^ ((n >> 1) & 7)) << 4);
            }
        };
        auto issue_fill = [&](int hbuf, int ph, int j) {          // (plain path; the fused path loads through registers)
            const int kelem = ph << 6;
            const bool second = kelem >= p.c0;
            const unsigned koff_b = (unsigned)(second ? kelem - p.c0 : kelem) * ES;      // scalar: rides in the instruction's soffset
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                GMK_LDS char* dst = (GMK_LDS char*)(smem + hbuf * HB + j * 8192 + (pw * 2 + u) * 1024);
                if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (GMK_LDS void*)dst, 16, hvoff[j][u], koff_b, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (GMK_LDS void*)dst, 16, hvoff[j][u], koff_b, 0, 0);
            }
        };
        auto issue_w = [&](int stage, int tap, int ph, int c) {      // c < 0: all 128 rows (4 instructions), else rows of channel half c (2)
            const unsigned wk = (unsigned)tap * p.w_tap_stride_b + ((unsigned)ph << 7);      // scalar: soffset
            if (c < 0) {
                GMK_LDS char* dst = (GMK_LDS char*)(smem + WOFF + stage * kWST + pw * 4096);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (GMK_LDS void*)(dst + u * 1024), 16, w_off[u], wk, 0, 0);
            } else {
                GMK_LDS char* dst = (GMK_LDS char*)(smem + WOFF + stage * kWST + (c * 64 + 16 * pw) * 128);
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (GMK_LDS void*)(dst + u * 1024), 16, c ? wh_off1[u] : wh_off0[u], wk, 0, 0);
            }
        };


        if constexpr (kFuse) {
            // ------------------------------------------------------------------------------------------------------------
            // Fused GroupNorm-apply + SiLU (reference sites simple_unet.py:161-163,169-172): the source is the RAW tensor; the halo
            // pieces come through registers instead of LDS-DMA: 16-byte loads two K-steps ahead of their use, y = silu(x * scale +
            // shift) with the (sample, channel) tables of gmk_gn_stats in fp32, rounded to bf16, ds_write_b128 into the slot's
            // swizzled chunk (pad slots stay exact zeros).  A lane always handles the SAME 8 channels of its slots (logical chunk
            // lane & 7), so a phase needs 2 x 8 scale and 2 x 8 shift values per lane: a tile of R <= H rows touches at most the
            // samples b0 and b0 + 1 (bit 28 of hpix selects).  Counted waits: the weight DMA of a step is issued first, then the
            // table loads (tap 0) and the step's two halo loads; loads of piece t are complete at the top of step t + 2 because the
            // weights issued behind them at step t + 1 have to be (vmcnt retires in order).
            // ------------------------------------------------------------------------------------------------------------
            const char* gsrc[2] = {(const char*)p.src0, (const char*)(p.c1 ? p.src1 : p.src0)};
            u32x4 L[3][2];
            f32x4 Csc[2][2], Csh[2][2];         // [sample b0 / b0 + 1][channels 0-3 / 4-7 of this lane's chunk] of the phase being filled
            // ident (experiment, GMK_DEV_VARIANT=11 without GroupNorm tables): the same register-staged fill with NO transform - the halo rows two
            // neighbouring tiles share then arrive by register loads, which the L2 merges (LDS-DMA reads of one line by two workgroups both reach memory)
            const bool ident = p.gn_scale == nullptr;
            auto load_tables = [&](int tl, int ph) {          // 8 x 16-byte loads (counted by hand)
                const int gr0 = (tl < p.ntiles ? tl : 0) * p.R;
                const int b0 = gr0 / H;
                const int kch = (ph << 6) + lch * 8;
#pragma unroll
                for (int sidx = 0; sidx < 2; ++sidx) {
                    const int bb = min(b0 + sidx, p.B - 1);
                    const float* ps = ident ? (const float*)p.src0 : p.gn_scale + (size_t)bb * p.gn_stride + kch;          // (ident: any valid 32 bytes - the counts stay)
                    const float* ph_ = ident ? (const float*)p.src0 : p.gn_shift + (size_t)bb * p.gn_stride + kch;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(Csc[sidx][0]) : "v"(ps) : "memory");
                    asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(Csc[sidx][1]) : "v"(ps) : "memory");
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(Csh[sidx][0]) : "v"(ph_) : "memory");
                    asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(Csh[sidx][1]) : "v"(ph_) : "memory");
                }
            };
            auto load_piece = [&](int ph, int j, int set) {   // 2 x 16-byte loads of this lane's logical chunk of its two slots
                const int kelem = ph << 6;
                const bool second = kelem >= p.c0;
                const unsigned cs_b = (unsigned)(second ? p.c1 : p.c0) * ES;
                const unsigned koff_b = (unsigned)(second ? kelem - p.c0 : kelem) * ES + (unsigned)lch * 16u;
                const char* sb = gsrc[second ? 1 : 0];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const unsigned pix = hpix[j][u] & 0x00FFFFFFu;
                    const char* a = sb + (pix == kBadPix ? 0u : __umul24(pix, cs_b) + koff_b);      // pad slots: any valid address, zeroed below
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(L[set][u]) : "v"(a) : "memory");
                }
            };
            // one instruction's worth (this lane's 16-byte chunk of one slot): kSel 0 / 1 = every live lane belongs to sample b0 / b0 + 1
            // (wave-uniform, the common case), 2 = mixed (a piece that straddles the image boundary): per-lane select
            auto xform = [&](const u32x4 raw, unsigned hp, auto sel_tag) -> u32x4 {
                constexpr int kSel = decltype(sel_tag)::value;
                constexpr float kNegLog2e = -1.4426950408889634f;
                const bool valid = (hp & 0x00FFFFFFu) != kBadPix;
                const bool s1 = (hp >> 28) & 1u;
                u32x4 o;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    float x0, x1;
                    unpack_pair<T>(raw[w], x0, x1);
                    const int h0 = (2 * w) >> 2, l0 = (2 * w) & 3, l1 = l0 + 1;
                    float a0, c0, a1, c1;
                    if (kSel == 0) { a0 = Csc[0][h0][l0]; c0 = Csh[0][h0][l0]; a1 = Csc[0][h0][l1]; c1 = Csh[0][h0][l1]; }
                    else if (kSel == 1) { a0 = Csc[1][h0][l0]; c0 = Csh[1][h0][l0]; a1 = Csc[1][h0][l1]; c1 = Csh[1][h0][l1]; }
                    else {
                        a0 = s1 ? Csc[1][h0][l0] : Csc[0][h0][l0]; c0 = s1 ? Csh[1][h0][l0] : Csh[0][h0][l0];
                        a1 = s1 ? Csc[1][h0][l1] : Csc[0][h0][l1]; c1 = s1 ? Csh[1][h0][l1] : Csh[0][h0][l1];
                    }
                    const float t0 = fmaf(x0, a0, c0), t1 = fmaf(x1, a1, c1);
                    const float y0 = t0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t0 * kNegLog2e));
                    const float y1 = t1 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t1 * kNegLog2e));
                    o[w] = valid ? pack_pair<T>(y0, y1) : 0u;
                }
                return o;
            };
            auto store_piece = [&](int hb, int j, int set) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    asm volatile("" : "+v"(L[set][u]));
                    const unsigned hp = hpix[j][u];
                    const bool valid = (hp & 0x00FFFFFFu) != kBadPix;
                    const unsigned long long mv = __builtin_amdgcn_ballot_w64(valid);
                    const unsigned long long m1 = __builtin_amdgcn_ballot_w64(valid && ((hp >> 28) & 1u));
                    u32x4 o = {0u, 0u, 0u, 0u};
                    if (ident) {
                        if (valid) o = L[set][u];
                    } else if (mv != 0) {                // pieces of pad slots only (a third of a 28 x 28 halo) skip the arithmetic
                        if (m1 == 0) o = xform(L[set][u], hp, IntTag<0>{});
                        else if (m1 == mv) o = xform(L[set][u], hp, IntTag<1>{});
                        else o = xform(L[set][u], hp, IntTag<2>{});
                    }
                    char* dst = smem + hb * HB + j * 8192 + (pw * 2 + u) * 1024 + lrow * 128 + (((hp >> 24) & 7u) << 4);
                    *reinterpret_cast<u32x4*>(dst) = o;
                }
            };
            const __amdgpu_buffer_rsrc_t rsr =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.residual ? p.residual : p.out), 0, (int)p.nbo, 0x00020000);
            unsigned pf0 = 0, pf1 = 0;
            int sq = 2, hbuf = 0;
            int tile = job_tile(0), ch = job_half(0);
            // prologue: the whole first halo half through registers, then the first two weight tiles
            load_tables(tile, 0);
#pragma unroll
            for (int j = 0; j < 7; ++j) resolve_piece(tile, j);
#pragma unroll
            for (int j0 = 0; j0 < 7; j0 += 3) {
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) if (j0 + jj < 7) load_piece(0, j0 + jj, jj);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
                    for (int w = 0; w < 2; ++w) asm volatile("" : "+v"(Csc[sidx][w]), "+v"(Csh[sidx][w]));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) if (j0 + jj < 7) store_piece(0, j0 + jj, jj);
            }
            issue_w(0, 0, 0, ch);
            issue_w(1, 1, 0, ch);
            if (ch < 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                      // start-up barrier: weight tile 0 and the first halo half are in LDS
            for (int k = 0; k < njobs; ++k) {
                const int ntile = job_tile(k + 1), nch = job_half(k + 1);
                for (int ph = 0; ph < nph; ++ph) {
                    const bool last_ph = ph + 1 == nph;
                    const int ph_next = last_ph ? 0 : ph + 1;
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        // behind the weights of step q - 1 there may still fly: the tables + piece 0 (issued at tap 0), one piece
                        // (taps 1 .. 6), the residual warm-up (tap 7)
                        const bool warmed = last_ph && p.residual != nullptr;
                        if (tap == 0 || (tap == 8 && !warmed)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        else if (tap == 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's ds_writes of the previous step
                        if (tap == 0) asm volatile("" :: "v"(pf0), "v"(pf1));
                        __builtin_amdgcn_s_barrier();
                        if (tap < 7) issue_w(sq, tap + 2, ph, ch);
                        else issue_w(sq, tap - 7, ph_next, last_ph ? nch : ch);
                        if (tap == 0) load_tables(last_ph ? ntile : tile, ph_next);
                        if (tap < 7) {
                            if (last_ph) resolve_piece(ntile, tap);
                            load_piece(ph_next, tap, tap % 3);
                        }
                        if (tap == 7 && warmed) {
                            const int ml = pw * 64 + lane;
                            const int m = tile * p.TP + ml;
                            const unsigned off = (ml < p.TP && m < p.M) ? (unsigned)m * (unsigned)p.out_cstride * ES + (unsigned)nblk * ES : kBadOff;
                            pf0 = __builtin_amdgcn_raw_buffer_load_b32(rsr, off, 0, 0);
                            pf1 = __builtin_amdgcn_raw_buffer_load_b32(rsr, off + 128u, 0, 0);
                        }
                        if (tap >= 2) {           // piece tap - 2 (and at tap 2 the tables) landed with the top-of-step wait
                            if (tap == 2) {
#pragma unroll
                                for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
                                    for (int w = 0; w < 2; ++w) asm volatile("" : "+v"(Csc[sidx][w]), "+v"(Csh[sidx][w]));
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            store_piece(hbuf ^ 1, tap - 2, (tap - 2) % 3);
                        }
                        sq = sq == RING - 1 ? 0 : sq + 1;
                    }
                    hbuf ^= 1;
                }
                tile = ntile; ch = nch;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
        if constexpr (kSkip) {
            // ---------------------------------------------------------------------------------------------------- folded skip convolution
            const __amdgpu_buffer_rsrc_t rsk0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.sk0), 0, (int)p.nbs, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsk1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.sk1), 0, (int)p.nbs, 0x00020000);
            const __amdgpu_buffer_rsrc_t rswk = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wsk), 0, (int)p.nbws, 0x00020000);
            const int drow = lane >> 2;
            const unsigned dchunk = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);      // logical chunk of this lane's physical chunk
            const unsigned srow_b = (unsigned)p.cs * ES;
            unsigned wdo[2];                    // dense weight tile: rows 32 pw + 16 u + drow of [cout][sk_ktot], 64 B of k per sub-phase (soffset)
#pragma unroll
            for (int u = 0; u < 2; ++u)
                wdo[u] = (unsigned)(p.nsk0 + nblk + 32 * pw + 16 * u + drow) * (unsigned)p.sk_ktot * ES + dchunk;
            // a tile's pixels are contiguous in memory (whole rows of the global row list): pixel px of tile tl is row tl * TP + px of the source
            unsigned dv[2][2];                  // [half][u]: this lane's pixel of its two instructions of a half-piece
            auto resolve_dense = [&](int tl) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int px = 128 * hf + 32 * pw + 16 * u + drow;
                        const int m = tl * p.TP + px;
                        dv[hf][u] = (tl < p.ntiles && px < p.TP && m < p.M) ? (unsigned)m * srow_b + dchunk : kBadOff;
                    }
            };
            auto issue_dense = [&](int fb, int slot, int k) {      // sub-phase k (channels 32 k .. 32 k + 31 of the concatenation): both halves, 4 instructions
                const unsigned koff_b = (unsigned)((32 * k) % p.cs) * ES;           // scalar: soffset
                const bool second = 32 * k >= p.cs;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        GMK_LDS char* dst = (GMK_LDS char*)(smem + fb * HB + (slot + hf) * 8192 + (32 * pw + 16 * u) * 64);
                        if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsk1, (GMK_LDS void*)dst, 16, dv[hf][u], koff_b, 0, 0);
                        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsk0, (GMK_LDS void*)dst, 16, dv[hf][u], koff_b, 0, 0);
                    }
            };
            unsigned wdh[2];                    // half jobs (channel half c): rows c * 64 + 16 pw + drow, one instruction
#pragma unroll
            for (int c = 0; c < 2; ++c)
                wdh[c] = (unsigned)(p.nsk0 + nblk + c * 64 + 16 * pw + drow) * (unsigned)p.sk_ktot * ES + dchunk;
            // dense weight tile k into LDS at byte offset `base` (ring slot `stage`: WOFF + stage * kWST; merged form: piece 6 of a half-buffer)
            auto issue_wd_at = [&](int base, int k, int c) {      // c < 0: all 128 rows (2 instructions), else the 64 rows of channel half c (1)
                if (c < 0) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        GMK_LDS char* dst = (GMK_LDS char*)(smem + base + (32 * pw + 16 * u) * 64);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rswk, (GMK_LDS void*)dst, 16, wdo[u], (unsigned)k * 64u, 0, 0);
                    }
                } else {
                    GMK_LDS char* dst = (GMK_LDS char*)(smem + base + (c * 64 + 16 * pw) * 64);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rswk, (GMK_LDS void*)dst, 16, c ? wdh[1] : wdh[0], (unsigned)k * 64u, 0, 0);
                }
            };
            auto issue_wd = [&](int stage, int k, int c) { issue_wd_at(WOFF + stage * kWST, k, c); };
            if constexpr (kMerge) {
                // ------------------------------------------------------------------------------ merged form: 9 K-steps per phase (see the kernel's header)
                auto wait4 = [&](bool all) __attribute__((always_inline)) {
                    if (all) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                };
                int sq = 2;
                int tile = job_tile(0), ch = job_half(0);
#pragma unroll
                for (int j = 0; j < 6; ++j) { resolve_piece(tile, j); issue_fill(0, 0, j); }
                issue_w(0, 0, 0, ch);
                issue_w(1, 1, 0, ch);
                if (ch < 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                __builtin_amdgcn_s_barrier();                      // start-up barrier: weight tile 0 and the first halo half are in LDS
                unsigned st_iss[13] = {}, st_wait[13] = {}, st_bar[13] = {};
                unsigned long long st_prev = 0, st_t0 = 0, st_t1 = 0;
                for (int k = 0; k < njobs; ++k) {
                    const int ntile = job_tile(k + 1), nch = job_half(k + 1);
                    resolve_dense(tile);
#pragma unroll
                    for (int ph = 0; ph < 2; ++ph) {
                        const int fb = ph ^ 1, hb = ph;            // the half-buffer this phase fills / reads
                        auto next_piece = [&](int j) __attribute__((always_inline)) {
                            if (ph == 1) resolve_piece(ntile, j);
                            issue_fill(fb, ph ^ 1, j);
                        };
#pragma unroll
                        for (int i = 0; i < 9; ++i) {
                            if constexpr (kStamp) { st_t0 = __builtin_readcyclecounter(); if (st_prev) st_iss[i] += (unsigned)(st_t0 - st_prev); }
                            wait4(i == 0 || i == 4);
                            if constexpr (kStamp) st_t1 = __builtin_readcyclecounter();
                            __builtin_amdgcn_s_barrier();
                            if constexpr (kStamp) { st_prev = __builtin_readcyclecounter(); st_wait[i] += (unsigned)(st_t1 - st_t0); st_bar[i] += (unsigned)(st_prev - st_t1); }
                            // tap weights of the step after next (the first two of the next phase / the next job behind taps 7 and 8)
                            if (i < 7) issue_w(sq, i + 2, ph, ch);
                            else issue_w(sq, i - 7, ph ^ 1, ph == 1 ? nch : ch);
                            // dense weight tiles: piece 6 of the filling (e0, e2) / of the reading (e1, e3) half-buffer
                            if (i == 0) issue_wd_at(fb * HB + 6 * 8192, 4 * ph + 0, ch);
                            if (i == 3) issue_wd_at(hb * HB + 6 * 8192, 4 * ph + 1, ch);
                            if (i == 4) issue_wd_at(fb * HB + 6 * 8192, 4 * ph + 2, ch);
                            if (i == 5) issue_wd_at(hb * HB + 6 * 8192, 4 * ph + 3, ch);
                            // pixel pieces
                            if (i == 0) issue_dense(fb, 0, 4 * ph + 0);
                            if (i == 1) issue_dense(fb, 2, 4 * ph + 1);
                            if (i == 2) issue_dense(fb, 4, 4 * ph + 2);
                            if (i == 4) issue_dense(fb, 0, 4 * ph + 3);
                            if (i == 5) { next_piece(2); next_piece(3); }
                            if (i == 6) { next_piece(4); next_piece(5); }
                            if (i == 7) { next_piece(0); next_piece(1); }
                            sq = sq == RING - 1 ? 0 : sq + 1;
                        }
                    }
                    tile = ntile; ch = nch;
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the speculative DMA before the LDS is released
                if constexpr (kStamp) {
                    if (pw == ((p.variant >= 30 && p.variant <= 33) ? p.variant - 30 : 0) && lane == 0 && p.stamps) {
                        unsigned* o = p.stamps + (size_t)blockIdx.x * 128;
#pragma unroll
                        for (int i = 0; i < 13; ++i) { o[i] = st_iss[i]; o[16 + i] = st_wait[i]; o[32 + i] = st_bar[i]; }
                        o[48] = (unsigned)njobs;
                    }
                }
                return;
            }
            // weight tile of step i (0..12, or 13 / 14 = steps 0 / 1 of the following phase) of phase ph: steps 4, 6, 8, 10 are the dense ones;
            // c / cn: channel half of this job / of the job whose phase 0 follows this job's phase 1 (-1: whole job)
            auto issue_weights = [&](int stage, int ph, int i, int c, int cn) __attribute__((always_inline)) {
                if (i >= 13) { i -= 13; if (ph == 1) c = cn; ph ^= 1; }          // (the next job runs the same convolution: same weights)
                if (i == 4 || i == 6 || i == 8 || i == 10) issue_wd(stage, 4 * ph + (i - 4) / 2, c);
                else issue_w(stage, i < 4 ? i : i == 5 ? 4 : i == 7 ? 5 : i == 9 ? 6 : i == 11 ? 7 : 8, ph, c);
            };
            auto wait_vm = [&](int n) __attribute__((always_inline)) {      // n is a constant after unrolling
                if (n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            };
            int sq = 2;
            int tile = job_tile(0), ch = job_half(0);
#pragma unroll
            for (int j = 0; j < 7; ++j) { resolve_piece(tile, j); issue_fill(0, 0, j); }
            issue_w(0, 0, 0, ch);
            issue_w(1, 1, 0, ch);
            if (ch < 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            __builtin_amdgcn_s_barrier();                      // start-up barrier: weight tile 0 and the first halo half are in LDS
            unsigned st_iss[13] = {}, st_wait[13] = {}, st_bar[13] = {};
            unsigned long long st_prev = 0, st_t0 = 0, st_t1 = 0;
            for (int k = 0; k < njobs; ++k) {
                const int ntile = job_tile(k + 1), nch = job_half(k + 1);
                resolve_dense(tile);
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                    const int fb = ph ^ 1;                     // the half-buffer this phase fills
                    // NEXT[j]: piece j of the next halo phase (phase 1 of this tile; in phase 1, phase 0 of the next job's tile)
                    auto next_piece = [&](int j) __attribute__((always_inline)) {
                        if (ph == 1) resolve_piece(ntile, j);
                        issue_fill(fb, ph ^ 1, j);
                    };
#pragma unroll
                    for (int i = 0; i < 13; ++i) {
                        // pixel-piece instructions of the previous step may still fly; its weights (for step i + 1) and everything older have landed
                        if constexpr (kStamp) { st_t0 = __builtin_readcyclecounter(); if (st_prev) st_iss[i] += (unsigned)(st_t0 - st_prev); }
                        wait_vm(i == 0 ? 0 : i == 4 ? 2 : (i == 1 || i == 2 || i == 3 || i == 6 || i == 8 || i == 10 || i == 12) ? 4 : 0);
                        if constexpr (kStamp) st_t1 = __builtin_readcyclecounter();
                        __builtin_amdgcn_s_barrier();
                        if constexpr (kStamp) { st_prev = __builtin_readcyclecounter(); st_wait[i] += (unsigned)(st_t1 - st_t0); st_bar[i] += (unsigned)(st_prev - st_t1); }
                        issue_weights(sq, ph, i + 2, ch, nch);
                        if (i == 0) issue_dense(fb, skip_slot(0), 4 * ph + 0);
                        if (i == 1) issue_dense(fb, skip_slot(1), 4 * ph + 1);
                        if (i == 2) issue_dense(fb, skip_slot(2), 4 * ph + 2);
                        if (i == 3) next_piece(6);
                        if (i == 5) issue_dense(fb, skip_slot(3), 4 * ph + 3);
                        if (i == 7) { next_piece(2); next_piece(3); }
                        if (i == 9) { next_piece(4); next_piece(5); }
                        if (i == 11) { next_piece(0); next_piece(1); }
                        sq = sq == RING - 1 ? 0 : sq + 1;
                    }
                }
                tile = ntile; ch = nch;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the speculative DMA before the LDS is released
            if constexpr (kStamp) {
                if (pw == ((p.variant >= 30 && p.variant <= 33) ? p.variant - 30 : 0) && lane == 0 && p.stamps) {
                    unsigned* o = p.stamps + (size_t)blockIdx.x * 128;
#pragma unroll
                    for (int i = 0; i < 13; ++i) { o[i] = st_iss[i]; o[16 + i] = st_wait[i]; o[32 + i] = st_bar[i]; }
                    o[48] = (unsigned)njobs;
                }
            }
            return;
        }
        if constexpr (kRing4) {
            // ---------------------------------------------------------------------------------------------------- four-slot weight ring (see the header)
            // In step q: the weight tile of step q + 3 (the first three of the next phase / job behind taps 6, 7, 8), then one of the next phase's SIX
            // halo pieces behind taps 0..5.  At barrier q everything but step q - 1's own instructions has landed: its weight tile (4 instructions,
            // 2 for a half job) and its halo piece (2) may still fly.
            auto wait_prev = [&](int nw, bool piece) __attribute__((always_inline)) {
                if (piece) { if (nw == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
                else { if (nw == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
            };
            int sq = 3, hbuf = 0;
            int tile = job_tile(0), ch = job_half(0);
#pragma unroll
            for (int j = 0; j < 6; ++j) { resolve_piece(tile, j); issue_fill(0, 0, j); }
            issue_w(0, 0, 0, ch);
            issue_w(1, 1, 0, ch);
            issue_w(2, 2, 0, ch);
            if (ch < 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();                      // start-up barrier: weight tile 0 and the first halo half are in LDS
            unsigned st_iss[9] = {}, st_wait[9] = {}, st_bar[9] = {};
            unsigned long long st_prev = 0, st_t0 = 0, st_t1 = 0;
            for (int k = 0; k < njobs; ++k) {
                const int ntile = job_tile(k + 1), nch = job_half(k + 1);
                for (int ph = 0; ph < nph; ++ph) {
                    const bool last_ph = ph + 1 == nph;
                    const int ph_next = last_ph ? 0 : ph + 1;
                    const int chn = last_ph ? nch : ch;         // the job the next phase belongs to
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        if constexpr (kStamp) { st_t0 = __builtin_readcyclecounter(); if (st_prev) st_iss[tap] += (unsigned)(st_t0 - st_prev); }
                        // the previous step issued: taps 0..5 a tile of THIS job + a piece; tap 6, 7 (seen at taps 7, 8) a tile of the next phase's job;
                        // tap 8 of the previous phase (seen at tap 0) the third tile of this phase: this job's
                        if (tap == 0) wait_prev(ch < 0 ? 4 : 2, false);
                        else if (tap <= 6) wait_prev(ch < 0 ? 4 : 2, true);
                        else wait_prev(chn < 0 ? 4 : 2, false);
                        if constexpr (kStamp) st_t1 = __builtin_readcyclecounter();
                        __builtin_amdgcn_s_barrier();
                        if constexpr (kStamp) { st_prev = __builtin_readcyclecounter(); st_wait[tap] += (unsigned)(st_t1 - st_t0); st_bar[tap] += (unsigned)(st_prev - st_t1); }
                        if (tap < 6) issue_w(sq, tap + 3, ph, ch);
                        else issue_w(sq, tap - 6, ph_next, chn);
                        if (tap < 6) {
                            if (last_ph) resolve_piece(ntile, tap);                  // the next fills belong to the next job's tile (or are zeros)
                            issue_fill(hbuf ^ 1, ph_next, tap);
                        }
                        sq = sq == RING - 1 ? 0 : sq + 1;
                    }
                    hbuf ^= 1;
                }
                tile = ntile; ch = nch;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the speculative DMA before the LDS is released
            if constexpr (kStamp) {
                if (pw == ((p.variant >= 30 && p.variant <= 33) ? p.variant - 30 : 0) && lane == 0 && p.stamps) {
                    unsigned* o = p.stamps + (size_t)blockIdx.x * 128;
#pragma unroll
                    for (int i = 0; i < 9; ++i) { o[i] = st_iss[i]; o[16 + i] = st_wait[i]; o[32 + i] = st_bar[i]; }
                    o[48] = (unsigned)(njobs * nph);
                }
            }
            return;
        }
        const __amdgpu_buffer_rsrc_t rsr =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.residual ? p.residual : p.out), 0, (int)p.nbo, 0x00020000);
        // Residual hand-over (whole jobs with a residual): the consumers have no registers to keep the residual tile's loads in flight under
        // the K loop (253 - 255 VGPRs), the producers have ~ 200 idle ones.  During taps 0..6 of a job's LAST phase each producer lane loads
        // 2 x 16 bytes per step of the tile's residual (14 loads: 7 / 8 of the 64-KiB tile; the counted waits below include them), all landed
        // at the top of tap 8.  After that step (barrier A: the consumers are done with the phase's halo buffer) the producers write
        // them into that buffer as R[unit 0..6][pixel 0..255][32 B] (unit = 16 channels; 7 x 8 KiB = the 56 KiB of a half-buffer), barrier
        // B, and the consumers' epilogue reads its residual from LDS (conflict-free: 64 lanes = 1 KiB contiguous) - only unit 7 still comes
        // from memory, issued ahead of everything else.  Replaces the L2 warm-up loads of round 2.
        const bool hand = kPrefetchW && p.residual != nullptr && p.variant != 7;      // (GMK_DEV_VARIANT=7: the A/B switch)
        u32x4 rr[14];
        unsigned roff[2] = {kBadOff, kBadOff};
        auto resolve_res = [&](int tl) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ml = pw * 64 + u * 32 + (lane >> 1);
                const int m = tl * p.TP + ml;
                roff[u] = (ml < p.TP && m < p.M) ? (unsigned)m * (unsigned)p.out_cstride * ES + (unsigned)nblk * ES + (unsigned)(lane & 1) * 16u : kBadOff;
            }
        };
        int sq = 2, hbuf = 0;
        int tile = job_tile(0), ch = job_half(0);
        if (hand) resolve_res(tile);
#pragma unroll
        for (int j = 0; j < 7; ++j) { resolve_piece(tile, j); issue_fill(0, 0, j); }
        issue_w(0, 0, 0, ch);
        issue_w(1, 1, 0, ch);
        if (ch < 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // start-up barrier: weight tile 0 is in LDS
        unsigned st_iss[9] = {}, st_wait[9] = {}, st_bar[9] = {};
        unsigned long long st_prev = 0, st_t0 = 0, st_t1 = 0;
        for (int k = 0; k < njobs; ++k) {
            const int ntile = job_tile(k + 1), nch = job_half(k + 1);
            for (int ph = 0; ph < nph; ++ph) {
                const bool last_ph = ph + 1 == nph;
                const int ph_next = last_ph ? 0 : ph + 1;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    if constexpr (kStamp) { st_t0 = __builtin_readcyclecounter(); if (st_prev) st_iss[tap] += (unsigned)(st_t0 - st_prev); }
                    // at barrier q the weight tile of step q+1 (the first 4 ops of step q-1) must have landed — the consumers read its
                    // first fragments before barrier q+1; only the 2 halo pieces issued behind it may still fly.  Tap 0 also needs
                    // the phase's whole halo (all older).
                    const bool handing = last_ph && hand && ch < 0;     // taps 0..6 carry 2 residual loads each, behind the halo pieces
                    if (kPrefetchW) {
                        if (tap == 0 || tap == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tap 8: every residual load has landed too
                        else if (handing) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    } else {      // only the weight tile of THIS step (issued two steps ago) has to be there
                        if (tap == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                        else if (tap == 1 || tap == 8) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    }
                    if constexpr (kStamp) st_t1 = __builtin_readcyclecounter();
                    __builtin_amdgcn_s_barrier();
                    if constexpr (kStamp) { st_prev = __builtin_readcyclecounter(); st_wait[tap] += (unsigned)(st_t1 - st_t0); st_bar[tap] += (unsigned)(st_prev - st_t1); }
                    if (tap < 7) issue_w(sq, tap + 2, ph, ch);
                    else issue_w(sq, tap - 7, ph_next, last_ph ? nch : ch);      // the first two weight tiles of the next job
                    if (tap < 7) {
                        if (last_ph) resolve_piece(ntile, tap);                  // the next fills belong to the next job's tile (or are zeros)
                        issue_fill(hbuf ^ 1, ph_next, tap);
                    }
                    if (tap < 7 && handing) {
#pragma unroll
                        for (int u = 0; u < 2; ++u)      // unit `tap` (32 bytes per pixel: the scalar offset), pixels pw * 64 + u * 32 + lane / 2
                            rr[2 * tap + u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsr, roff[u], tap * 32, 0));
                    }
                    if (tap == 8 && handing) {
                        __builtin_amdgcn_s_barrier();                              // A: the consumers have read this phase's halo for the last time
                        GMK_LDS char* dst = (GMK_LDS char*)(smem + hbuf * HB + pw * 2048 + lane * 16);
#pragma unroll
                        for (int t = 0; t < 7; ++t)
#pragma unroll
                            for (int u = 0; u < 2; ++u) *reinterpret_cast<GMK_LDS u32x4*>(dst + t * 8192 + u * 1024) = rr[2 * t + u];
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();                              // B: the residual tile is in LDS
                    }
                    sq = sq == RING - 1 ? 0 : sq + 1;
                }
                hbuf ^= 1;
            }
            tile = ntile; ch = nch;
            if (hand && ch < 0 && tile < p.ntiles) resolve_res(tile);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the speculative DMA before the LDS is released
        if constexpr (kStamp) {
            if (pw == ((p.variant >= 30 && p.variant <= 33) ? p.variant - 30 : 0) && lane == 0 && p.stamps) {
                unsigned* o = p.stamps + (size_t)blockIdx.x * 128;
#pragma unroll
                for (int i = 0; i < 9; ++i) { o[i] = st_iss[i]; o[16 + i] = st_wait[i]; o[32 + i] = st_bar[i]; }
                o[48] = (unsigned)(njobs * nph);
            }
        }
        return;
    }

    // =============================================== consumer waves ===============================================
    if constexpr (kShape == 16) {
        // v_mfma_f32_16x16x32_bf16 form.  Same operands, same LDS traffic (24 ds_read_b128 per K-step) and the same MFMA cycles
        // as the 32x32x16 form below (64 x 16 instead of 32 x 32 per K-step), but under the board's power limit the chip holds a
        // higher clock on this shape (guide: 1.12 - 1.15 x FLOP/s at equal cycles).  Wave w owns pixels 64 w .. 64 w + 63
        // (4 blocks of 16) x ALL channels of the job (whole job: 128 = 8 blocks of 16; half job: the 64 of channel half c), so the
        // geometry is the same for both job kinds.  Lane = (r16 = lane & 15, q = lane >> 4): A fragment = 16 channel rows x k chunk
        // (4 k2 + q), B fragment = 16 pixel columns x the same chunk; D: lane holds channels 4 q .. 4 q + 3 of pixel r16.
        // Per tap: 2 k2 halves x NCB / 2 groups of 8 MFMAs (2 channel blocks x 4 pixel blocks); the weight fragments of the next
        // group and the pixel fragments of the next k2 half / next tap are read under the current group's MFMAs.
        const int r16 = lane & 15, q = lane >> 4;
        const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.nbo, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsr =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.residual ? p.residual : p.out), 0, (int)p.nbo, 0x00020000);
        const int pxbase = wave * 64;
        int cslot[4], cn[4];
        auto resolve_centres = [&](int tile) {
            const int gr0 = tile * p.R;
            const int b0 = gr0 / H;
            const int y0 = gr0 - b0 * H;
            // (row / column of this lane's pixels: recomputed per tile from an opaque copy of the lane index - as loop invariants they were spilled
            // and reloaded from scratch in the middle of the epilogue's stores, see there)
            int lane_c = lane;
            asm volatile("" : "+v"(lane_c));
            int rit[4], px_x[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ml = pxbase + i * 16 + (lane_c & 15);
                rit[i] = div_small(ml, p.inv_w);
                px_x[i] = ml - rit[i] * W;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = y0 + rit[i];
                const int k = div_small(t, p.inv_h);
                const int y = t - k * H;
                const int er = k * (H + 2) + y + 1 - y0;
                int s = er * WE + px_x[i] + 1;
                int n = s - er - 1;
                if (s < WE + 1 || s >= kHaloSlots - WE - 1) { s = WE + 1; n = W; }   // dead rows (m_local >= TP): any in-range slot
                cslot[i] = s;
                cn[i] = n;
            }
        };
        f32x4 acc[8][4];          // [channel block][pixel block]
        const int swz = (r16 >> 1) & 7;          // weight rows 16 cb + r16: (row >> 1) & 7 does not depend on cb
        int b_off = 0;                           // LDS offset of this lane's weight row 0 of the job's channel range
        frag_t px[2][4], wt[2][2];
        int rowb[4], sw[4];
        int st = 0, hbuf = 0;
        unsigned sc_work[13] = {}, sc_bar[13] = {}, sc_epi[3] = {};
        unsigned long long sc_prev = 0;
        // barrier of K-step `idx` (taps 0..8, dense sub-phases 9..12): kStamp sums the cycles from the previous barrier's exit to this one's entry and in it
        auto step_barrier = [&](int idx) __attribute__((always_inline)) {
            if constexpr (kStamp) {
                const unsigned long long a = __builtin_readcyclecounter();
                if (sc_prev) sc_work[idx] += (unsigned)(a - sc_prev);
                __builtin_amdgcn_s_barrier();
                sc_prev = __builtin_readcyclecounter();
                sc_bar[idx] += (unsigned)(sc_prev - a);
            } else {
                __builtin_amdgcn_s_barrier();
            }
        };
        auto load_wt = [&](int stg, int k2, int pair, int set) {
            const char* Wb = smem + stg * kWST + b_off + pair * 4096;
            const int coff = ((k2 * 4 + q) ^ swz) << 4;
            wt[set][0] = *reinterpret_cast<const frag_t*>(Wb + coff);
            wt[set][1] = *reinterpret_cast<const frag_t*>(Wb + 2048 + coff);
        };
        // folded skip convolution: 64-byte rows (32 channels); this lane's byte offset inside a half-piece / a dense weight tile
        const int hfs = __builtin_amdgcn_readfirstlane(wave) >> 1;                 // which 128-pixel half this wave's pixels are in
        const int d_off = (((wave & 1) * 64 + r16) << 6) + ((q ^ ((r16 >> 2) & 3)) << 4);
        const int dw_off = WOFF + (r16 << 6) + ((q ^ ((r16 >> 2) & 3)) << 4);
        int dw_half = 0;                                                           // byte offset of the job's channel half in a dense weight tile
        auto load_wt_d_at = [&](int base, int pair, int set) {                    // channel blocks 2 pair, 2 pair + 1 of the dense weight tile at LDS byte `base`
            int o = dw_off - WOFF;
            asm volatile("" : "+v"(o));
            const char* Wb = smem + base + dw_half + pair * 2048 + o;
            wt[set][0] = *reinterpret_cast<const frag_t*>(Wb);
            wt[set][1] = *reinterpret_cast<const frag_t*>(Wb + 1024);
        };
        auto load_wt_d = [&](int stg, int pair, int set) { load_wt_d_at(WOFF + stg * kWST, pair, set); };
        // This block's 128 bias values (+ the folded skip convolution's own, simple_unet.py:177-179) live in TWO registers per consumer wave for
        // the whole launch: lane l holds channels nblk + l and nblk + 64 + l.  A tile's accumulator-layout copy (lane (r16, q): channels
        // 16 cb + 4 q .. + 3 of block cb) is gathered from them by 32 ds_bpermute_b32 - no memory instruction: the per-tile global loads this
        // replaces took 540 (8 loads) / 2,500 (16 loads, folded kernel) cycles of a tile's epilogue to ISSUE behind the producers' DMA and
        // the output stores (round 6, tools/step_stamps.py).
        float bl0 = 0.f, bl1 = 0.f;
        if (p.bias) { bl0 = p.bias[nblk + lane]; bl1 = p.bias[nblk + 64 + lane]; }
        if constexpr (kSkip) { bl0 += p.bias2[nblk + lane]; bl1 += p.bias2[nblk + 64 + lane]; }
        f32x4 bz[8];              // bias of the NEXT job's channel blocks in the accumulator layout (see mfma_group)
        auto load_bias = [&](int chalf) __attribute__((always_inline)) {      // chalf: channel half of the job the values are for (-1: whole job)
            int lane_b = lane;
            asm volatile("" : "+v"(lane_b));          // (recomputed per tile, not kept live or spilled across the K loop: see the epilogue)
            const int qa = (lane_b >> 4) << 4;        // byte address of source lane 4 q
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
                const float src = (chalf >= 0 ? chalf : (cb >> 2)) ? bl1 : bl0;      // (a half job has four blocks: 4..7 repeat them, unused)
                float t[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    t[e] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(qa + 4 * (16 * (cb & 3) + e), __builtin_bit_cast(int, src)));
                bz[cb] = (f32x4){t[0], t[1], t[2], t[3]};
            }
        };
        auto run_job = [&](auto ncb_tag, int tile, int chalf, int next_boff, int next_chalf) {
            constexpr int NCB = decltype(ncb_tag)::value;          // 8: whole job, 4: half job
            constexpr int NG = NCB / 2;                            // groups (channel-block pairs) per k2 half
            auto addr = [&](int tap) {
                const int tapoff = (tap / 3 - 1) * WE + (tap % 3 - 1), tapoff_n = (tap / 3 - 1) * W + (tap % 3 - 1);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int c = cslot[i], n = cn[i];
                    asm volatile("" : "+v"(c), "+v"(n));      // keep the per-tap addresses out of loop-invariant hoisting
                    rowb[i] = (c + tapoff) << 7;
                    sw[i] = ((n + tapoff_n) >> 1) & 7;
                }
            };
            auto load_px = [&](int hb, int k2, int set, int i0, int i1) {
                const char* Hb = smem + hb * HB;
#pragma unroll
                for (int i = i0; i < i1; ++i)
                    px[set][i] = *reinterpret_cast<const frag_t*>(Hb + rowb[i] + (((k2 * 4 + q) ^ sw[i]) << 4));
            };
            // A tile's first MFMA into an accumulator takes the channel block's BIAS as its C operand (lane (r16, q) holds channels 4 q .. 4 q + 3
            // of its pixels: the same four values for every pixel block) - the epilogue then has no bias loads to wait for and no adds (round 6:
            // the bias round trip was 1,250 of a 32 x 32 tile's 37,000 cycles, tools/step_stamps.py).  bz is gathered at the END of the previous
            // tile's epilogue (load_bias), or in the prologue; zeros when there is no bias.
            auto mfma_group = [&](int pair, int wset, int pset, auto fresh_tag) __attribute__((always_inline)) {
                constexpr bool kFresh = decltype(fresh_tag)::value != 0;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[2 * pair + j][i] = mfma_16x16x32<T>(wt[wset][j], px[pset][i], kFresh ? bz[2 * pair + j] : acc[2 * pair + j][i]);
            };
            auto phase = [&](auto first_tag, int ph) __attribute__((always_inline)) {
                constexpr int kFirst = decltype(first_tag)::value;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const bool dense_after = kSkip && tap >= 3 && tap <= 6;      // a dense sub-phase of the folded skip convolution follows this tap
                    // merged form: the fragment set that holds this tap's first k half (the other one holds the second); the sets swap roles behind
                    // every dense sub-phase, whose fragments take the first-half set once its four groups are done
                    const int f = (kMerge && (tap == 4 || tap == 6)) ? 1 : 0;
                    // LDS byte offset of the dense weight tile of sub-phase tap - 3 (merged form): piece 6 of the filling / reading half-buffer
                    const int dwb = (((tap - 3) & 1) ? hbuf : (hbuf ^ 1)) * HB + 6 * 8192;
                    step_barrier(tap);
                    if (tap == 0) { addr(0); load_px(hbuf, 0, 0, 0, 4); }      // the phase's halo only became valid with this barrier
                    if (!kPrefetchW) load_wt(st, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < 2 * NG; ++g) {
                        const int k2 = g / NG, pair = g % NG;
                        const bool last = g + 1 == 2 * NG;
                        // reads issued under this group's 8 MFMAs: the next group's weight fragments; the second k2 half's pixel
                        // fragments during the first half's last two groups; in the tap's last group the next tap's addresses and
                        // first pixel fragments (same halo) and the next step's first weight fragments (landed at this barrier)
                        if (!last) load_wt(st, (g + 1) / NG, (g + 1) % NG, (g + 1) & 1);
                        if (k2 == 0 && pair == NG - 2) load_px(hbuf, 1, f ^ 1, 0, 2);
                        if (k2 == 0 && pair == NG - 1) load_px(hbuf, 1, f ^ 1, 2, 4);
                        if (last) {
                            st = st == RING - 1 ? 0 : st + 1;
                            if (tap < 8 && !dense_after) { addr(tap + 1); load_px(hbuf, 0, 0, 0, 4); }
                            if (tap == 8 && ph + 1 == nph) b_off = next_boff;
                            if (dense_after && kMerge) {
                                // the dense sub-phase's pixel fragments (landed at this step's barrier) and first weight fragments
                                int o = d_off;
                                asm volatile("" : "+v"(o));          // per-step addresses stay out of loop-invariant hoisting (registers)
                                const char* Eb = smem + (hbuf ^ 1) * HB + ((tap == 3 || tap == 6 ? 0 : tap == 4 ? 2 : 4) + hfs) * 8192 + o;
#pragma unroll
                                for (int i = 0; i < 4; ++i) px[f][i] = *reinterpret_cast<const frag_t*>(Eb + i * 1024);
                                load_wt_d_at(dwb, 0, 0);
                            } else if (dense_after) load_wt_d(st, 0, 0);      // the dense step's first weight fragments (landed at this barrier)
                            else if (kPrefetchW) load_wt(st, 0, 0, 0);
                        }
                        if (kFirst && tap == 0 && k2 == 0) mfma_group(pair, g & 1, k2 ^ f, IntTag<1>{}); else mfma_group(pair, g & 1, k2 ^ f, IntTag<0>{});
                        // one read per MFMA where there are reads to hide; the address VALU of the tap's last group rides along
                        if (last && dense_after && kMerge) {
#pragma unroll
                            for (int k = 0; k < 6; ++k) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            }
                            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        } else if (last && tap < 8 && !dense_after) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                            }
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            }
                            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            }
                            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (kSkip) {
                        if (dense_after) {
                            // ---- dense sub-phase m = tap - 3 of this phase (32 channels of torch.cat([x, skip]) x the 1x1 skip weights): its pixel
                            // rows sit in the half-buffer this phase is filling.  4 pixel fragments (this wave's 64 pixels, one k chunk per lane)
                            // behind the barrier, then 4 groups of 8 MFMAs (2 channel blocks x 4 pixel blocks) with the next group's weight
                            // fragments read underneath; the last group brings the next tap's addresses, pixel and weight fragments.
                            if constexpr (!kMerge) {
                                step_barrier(9 + tap - 3);
                                int o = d_off;
                                asm volatile("" : "+v"(o));          // per-step addresses stay out of loop-invariant hoisting (registers)
                                const char* Eb = smem + (hbuf ^ 1) * HB + (skip_slot(tap - 3) + hfs) * 8192 + o;
#pragma unroll
                                for (int i = 0; i < 4; ++i) px[1][i] = *reinterpret_cast<const frag_t*>(Eb + i * 1024);
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int g = 0; g < NG; ++g) {                   // NG channel-block pairs: 4 (whole job) or 2 (half job)
                                const bool last = g == NG - 1;
                                if (!last) { if (kMerge) load_wt_d_at(dwb, g + 1, (g + 1) & 1); else load_wt_d(st, g + 1, (g + 1) & 1); }
                                else {
                                    if (!kMerge) st = st == RING - 1 ? 0 : st + 1;      // (merged form: the dense tile is not in the ring)
                                    addr(tap + 1); load_px(hbuf, 0, kMerge ? f ^ 1 : 0, 0, 4);
                                    load_wt(st, 0, 0, 0);
                                }
                                mfma_group(g, g & 1, kMerge ? f : 1, IntTag<0>{});
                                if (last) {
#pragma unroll
                                    for (int k = 0; k < 4; ++k) {
                                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                                    }
#pragma unroll
                                    for (int k = 0; k < 4; ++k) {
                                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                                    }
                                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                                } else {
#pragma unroll
                                    for (int k = 0; k < 2; ++k) {
                                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                                    }
                                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                }
                hbuf ^= 1;
            };
            if constexpr (kSkip) dw_half = chalf > 0 ? 4096 : 0;          // dense weight rows c * 64 .. of a half job (64 B per row)
            phase(IntTag<1>{}, 0);
            for (int ph = 1; ph < nph; ++ph) phase(IntTag<0>{}, ph);
            asm volatile("" ::: "memory");
            unsigned long long se0 = 0, se1 = 0;
            if constexpr (kStamp) {          // the last step's MFMAs have been issued; reading an accumulator waits for them
                float t = acc[7][3][3];
                asm volatile("v_mov_b32 %0, %0" : "+v"(t));
                acc[7][3][3] = t;
                se0 = __builtin_readcyclecounter();
                sc_epi[0] += (unsigned)(se0 - sc_prev);
            }
            // Everything the epilogue derives from the lane index is RECOMPUTED here from an opaque copy of it: left to itself the compiler hoists
            // those loop invariants (pixel offsets, LDS read addresses of the handed-over residual, channel offsets) in front of the job loop,
            // spills them (the K loop has no registers to spare) and reloads them from scratch in the middle of the epilogue - and a scratch
            // reload sits in the same in-order vmcnt queue as the output stores, so each reload waited for the stores in front of it to be
            // ACKNOWLEDGED (round 6, tools/step_stamps.py: 5,000 of a 32 x 32 tile's 37,000 cycles were conversion + store issue; 3,500 with the stores dropped).
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            const int r16e = lane_e & 15, qe = lane_e >> 4;
            // residual hand-over (see the producers): units 0..6 of the tile's residual arrive in the halo buffer this job's last phase
            // just finished with; unit 7 (channels 112..127) is loaded from memory here, ahead of the two barriers that hide its latency
            constexpr bool kHandJob = kPrefetchW && !kFuse && NCB == 8;
            bool has_res = p.residual != nullptr;
            if constexpr (kRes >= 0) has_res = kRes != 0;
            const bool hand = kHandJob && has_res && p.variant != 7;
            u32x4 r7[2];
            const char* Rl = smem + (hbuf ^ 1) * HB;
            if (hand) {
#pragma unroll
                for (int ip = 0; ip < 2; ++ip) {
                    const int ml = pxbase + (2 * ip + (qe & 1)) * 16 + r16e;
                    const int m = tile * p.TP + ml;
                    const unsigned off = (ml < p.TP && m < p.M) ? (unsigned)m * (unsigned)p.out_cstride * ES + (unsigned)(nblk + (qe >> 1) * 8 + 7 * 16) * ES : kBadOff;
                    r7[ip] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsr, off, 0, 0));
                }
                __builtin_amdgcn_s_barrier();            // A
                __builtin_amdgcn_s_barrier();            // B
            }
            // ---- epilogue: lane (r16e, qe) holds channels 16 cb + 4 qe .. + 3 of pixel 16 i + r16e.  One v_permlane16_swap per dword between
            // the packed values of pixel blocks (i, i + 1) leaves every lane with 8 consecutive channels (16 bytes) of ONE pixel: even
            // rows (qe = 0, 2) pixel block i, odd rows pixel block i + 1, channel offset 8 (qe >> 1).  The bias is already in the accumulators
            // (mfma_group); the residual comes as 16-byte loads at the store addresses, swapped back into the accumulator layout.
            const int cbase = nblk + (chalf >= 0 ? chalf * 64 : 0);
            if constexpr (kStamp) {          // (the hand-over barriers, if any)
                se1 = __builtin_readcyclecounter();
                sc_epi[1] += (unsigned)(se1 - se0);
            }
            const bool no_store = kStamp && p.variant == 21;          // (experiment: the epilogue without its HBM writes)
#pragma unroll
            for (int ip = 0; ip < 2; ++ip) {
                const int ml = pxbase + (2 * ip + (qe & 1)) * 16 + r16e;          // the pixel this lane stores after the swap
                const int m = tile * p.TP + ml;
                const bool live = ml < p.TP && m < p.M;
                const unsigned row_b = (unsigned)m * (unsigned)p.out_cstride * ES + (unsigned)(cbase + (qe >> 1) * 8) * ES;
                u32x4 rres[NCB];
                if (hand) {
#pragma unroll
                    for (int cb = 0; cb < NCB - 1; ++cb)
                        rres[cb] = *reinterpret_cast<const u32x4*>(Rl + cb * 8192 + ml * 32 + (qe >> 1) * 16);
                    rres[NCB - 1] = r7[ip];
                } else if (has_res) {
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
                        rres[cb] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsr, live ? row_b + (unsigned)(cb * 16) * ES : kBadOff, 0, 0));
                }
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    float v0[4], v1[4];         // pixel blocks 2 ip and 2 ip + 1 in the accumulator layout
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v0[e] = acc[cb][2 * ip][e]; v1[e] = acc[cb][2 * ip + 1][e]; }
                    if (has_res) {
                        const u32x4 R = rres[cb];
                        const auto s0 = __builtin_amdgcn_permlane16_swap(R[0], R[2], false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(R[1], R[3], false, false);
                        const half_t ra = __builtin_bit_cast(half_t, (u32x2_t){s0[0], s1[0]});      // pixel block 2 ip
                        const half_t rb = __builtin_bit_cast(half_t, (u32x2_t){s0[1], s1[1]});      // pixel block 2 ip + 1
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v0[e] += (float)ra[e]; v1[e] += (float)rb[e]; }
                    }
                    const u32x2_t u0 = {pack_pair<T>(sat16<T>(v0[0]), sat16<T>(v0[1])), pack_pair<T>(sat16<T>(v0[2]), sat16<T>(v0[3]))};
                    const u32x2_t u1 = {pack_pair<T>(sat16<T>(v1[0]), sat16<T>(v1[1])), pack_pair<T>(sat16<T>(v1[2]), sat16<T>(v1[3]))};
                    const auto s0 = __builtin_amdgcn_permlane16_swap(u0[0], u1[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(u0[1], u1[1], false, false);
                    const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                    __builtin_amdgcn_raw_buffer_store_b128(o, rso, live && !no_store ? row_b + (unsigned)(cb * 16) * ES : kBadOff, 0, 0);
                }
            }
            load_bias(next_chalf);          // the NEXT job's bias in the accumulator layout (first used by its first MFMAs)
            if constexpr (kStamp) sc_epi[2] += (unsigned)(__builtin_readcyclecounter() - se1);      // conversion + store issue + the bias gather
        };

        {
            const int hc0 = job_half(0);
            b_off = WOFF + ((hc0 >= 0 ? hc0 * 64 : 0) + r16) * 128;
        }
        resolve_centres(job_tile(0));
        load_bias(job_half(0));
        __builtin_amdgcn_s_barrier();                          // start-up barrier: weight tile 0 is in LDS
        if (kPrefetchW) load_wt(0, 0, 0, 0);
        for (int k = 0; k < njobs; ++k) {
            const int tile = job_tile(k), hc = job_half(k), ntile = job_tile(k + 1), nhc = job_half(k + 1);
            const int next_boff = WOFF + ((nhc >= 0 ? nhc * 64 : 0) + r16) * 128;
            if (hc < 0) run_job(IntTag<8>{}, tile, hc, next_boff, nhc);
            else run_job(IntTag<4>{}, tile, hc, next_boff, nhc);
            resolve_centres(ntile);
            asm volatile("" ::: "memory");
        }
        if constexpr (kStamp) {
            if (wave == ((p.variant >= 30 && p.variant <= 33) ? p.variant - 30 : 0) && lane == 0 && p.stamps) {
                unsigned* o = p.stamps + (size_t)blockIdx.x * 128 + 64;
#pragma unroll
                for (int i = 0; i < 13; ++i) { o[i] = sc_work[i]; o[16 + i] = sc_bar[i]; }
                o[32] = sc_epi[0]; o[33] = sc_epi[1]; o[34] = sc_epi[2]; o[35] = (unsigned)njobs;
            }
        }
        return;
    }
    // Whole job: wave = (cm, cw) owns pixel half cm (128 = 4 blocks of 32) x channel half cw (64).  Half job: all four waves share
    // channel half `ch`, wave w owns pixels 64w .. 64w+63 (NI = 2 blocks): half the MFMAs of a whole job in every wave.
    const int cm = wave >> 1, cw = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.nbo, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.residual ? p.residual : p.out), 0, (int)p.nbo, 0x00020000);
    int pxbase = cm * 128, cwe = cw;                  // first pixel / channel half of this wave in the current job
    int rit[4], px_x[4];
    auto set_geometry = [&](int half_c) {
        if (half_c >= 0) { pxbase = wave * 64; cwe = half_c; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ml = pxbase + i * 32 + r;
            rit[i] = div_small(ml, p.inv_w);
            px_x[i] = ml - rit[i] * W;
        }
    };
    int cslot[4], cn[4];
    auto resolve_centres = [&](int tile) {
        const int gr0 = tile * p.R;
        const int b0 = gr0 / H;
        const int y0 = gr0 - b0 * H;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = y0 + rit[i];
            const int k = div_small(t, p.inv_h);
            const int y = t - k * H;
            const int er = k * (H + 2) + y + 1 - y0;
            int s = er * WE + px_x[i] + 1;
            int n = s - er - 1;
            if (s < WE + 1 || s >= kHaloSlots - WE - 1) { s = WE + 1; n = W; }   // dead rows (m_local >= TP): any in-range slot
            cslot[i] = s;
            cn[i] = n;
        }
    };

    f32x16 acc[2][4];     // [j: channel tile][i: pixel tile]
    const int swz = (r >> 1) & 7;
    int b_off = 0;

    // Software pipeline of one K-step (tap): the pixel-fragment reads of group 0 and all address arithmetic of tap t+1 are
    // issued before the last MFMA group of tap t (the halo does not change inside a phase), so after the barrier only the two
    // weight-fragment reads of group 0 stand between the wave and its first MFMA.  Two fragment sets alternate per group.
    frag_t px[2][4], wt[2][2];
    int rowb[4], sw[4];
    int st = 0, hbuf = 0;
    auto load_wt = [&](int stg, int kg, int set) {
        const char* Wb = smem + stg * kWST + b_off;
        const int coff = ((kg * 2 + h) ^ swz) << 4;
        wt[set][0] = *reinterpret_cast<const frag_t*>(Wb + coff);
        wt[set][1] = *reinterpret_cast<const frag_t*>(Wb + 4096 + coff);
    };

    // one job = nph phases x 9 taps + epilogue; NI = pixel blocks per wave (4: whole job, 2: half job)
    auto run_job = [&](auto ni_tag, int tile, int next_boff) {
        constexpr int NI = decltype(ni_tag)::value;
        auto pre = [&](int hb, int tap) {             // addresses of `tap` + its group-0 pixel fragments -> set 0
            const char* Hb = smem + hb * HB;
            const int tapoff = (tap / 3 - 1) * WE + (tap % 3 - 1), tapoff_n = (tap / 3 - 1) * W + (tap % 3 - 1);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                int c = cslot[i], n = cn[i];
                asm volatile("" : "+v"(c), "+v"(n));      // keep the per-tap addresses out of loop-invariant hoisting
                rowb[i] = (c + tapoff) << 7;
                sw[i] = ((n + tapoff_n) >> 1) & 7;
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
                px[0][i] = *reinterpret_cast<const frag_t*>(Hb + rowb[i] + ((h ^ sw[i]) << 4));
        };
        auto load_px = [&](int hb, int kg, int set) {
            const char* Hb = smem + hb * HB;
#pragma unroll
            for (int i = 0; i < NI; ++i)
                px[set][i] = *reinterpret_cast<const frag_t*>(Hb + rowb[i] + (((kg * 2 + h) ^ sw[i]) << 4));
        };
        auto interleave_reads = [&]() {       // whole job: (MFMA, 2 VALU, read) x 4, (MFMA, read) x 2, MFMA x 2;  half job: half of each
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if (NI == 4) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        };
        auto interleave_pre = [&]() {         // whole job: (MFMA, 5 VALU) x 4, (MFMA, 2 VALU, read) x 4, 2 reads
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            }
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        };
        auto mfma_group = [&](int set, auto fresh_tag) __attribute__((always_inline)) {
            constexpr bool kFresh = decltype(fresh_tag)::value != 0;      // first MFMAs of a job: C = 0, no accumulator re-zeroing anywhere
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    acc[j][i] = mfma_32x32x16<T>(wt[set][j], px[set][i], kFresh ? z : acc[j][i]);
        };
        // residual hand-over (see the producers): set before the epilogue runs
        constexpr bool kHandJob = kPrefetchW && !kFuse && NI == 4;
        const bool hand = kHandJob && p.residual != nullptr && p.variant != 7;
        u32x4 r7[4];
        const char* Rl = smem;
        auto epilogue = [&]() {
            // every global load of the epilogue is issued up front, in as few and as wide instructions as possible: the bias once
            // per tile (not once per accumulator), the residual as 16-B loads at the addresses the stores use (the lane layout after
            // v_permlane32_swap) and swapped back into the accumulator layout — half the instructions of 8-B loads in that layout
            float bz[2][16];
            if (p.bias) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        float t[4];
                        load4(p.bias + nblk + cwe * 64 + j * 32 + 8 * q4 + 4 * h, t);
#pragma unroll
                        for (int e = 0; e < 4; ++e) bz[j][4 * q4 + e] = t[e];
                    }
            }
            u32x4 rres[2][2][2];            // residual of pixel block i (set i & 1), loaded one block ahead
            auto load_res = [&](int i) {
                const int ml = pxbase + i * 32 + r;
                if (hand) {                 // from the producers' hand-over: unit = 16 channels, R[unit][pixel][32 B]; unit 7 was loaded ahead
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int qq = 0; qq < 2; ++qq) {
                            if (j == 1 && qq == 1 && cwe == 1) rres[i & 1][j][qq] = r7[i];
                            else rres[i & 1][j][qq] = *reinterpret_cast<const u32x4*>(Rl + (cwe * 4 + j * 2 + qq) * 8192 + ml * 32 + h * 16);
                        }
                    return;
                }
                const int m = tile * p.TP + ml;
                const bool live = ml < p.TP && m < p.M;
                const unsigned row_b = (unsigned)m * (unsigned)p.out_cstride * ES;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int cb = nblk + cwe * 64 + j * 32;
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        const unsigned off = live ? row_b + (unsigned)(cb + 8 * (2 * qq + h)) * ES : kBadOff;
                        rres[i & 1][j][qq] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsr, off, 0, 0));
                    }
                }
            };
            if (p.residual) load_res(0);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int ml = pxbase + i * 32 + r;
                const int m = tile * p.TP + ml;
                const bool live = ml < p.TP && m < p.M;
                const unsigned row_b = (unsigned)m * (unsigned)p.out_cstride * ES;
                const float* embp = nullptr;
                if (p.emb) embp = p.emb + (int64_t)((live ? m : 0) / (H * W)) * p.emb_stride;
                if (p.residual && i < NI - 1) load_res(i + 1);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int cb = nblk + cwe * 64 + j * 32;
                    float v[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = acc[j][i][e];
                    if (p.bias) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] += bz[j][e];
                    }
                    if (p.emb) {
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            float t[4];
                            load4(embp + cb + 8 * q4 + 4 * h, t);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[4 * q4 + e] += t[e];
                        }
                    }
                    if (p.residual) {
#pragma unroll
                        for (int qq = 0; qq < 2; ++qq) {
                            const u32x4 R = rres[i & 1][j][qq];
                            const auto s0 = __builtin_amdgcn_permlane32_swap(R[0], R[2], false, false);
                            const auto s1 = __builtin_amdgcn_permlane32_swap(R[1], R[3], false, false);
                            const half_t ra = __builtin_bit_cast(half_t, (u32x2_t){s0[0], s1[0]});      // block q4 = 2qq
                            const half_t rb = __builtin_bit_cast(half_t, (u32x2_t){s0[1], s1[1]});      // block q4 = 2qq + 1
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v[8 * qq + e] += (float)ra[e];
                                v[8 * qq + 4 + e] += (float)rb[e];
                            }
                        }
                    }
                    unsigned pk[4][2];
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        pk[q4][0] = pack_pair<T>(sat16<T>(v[4 * q4]), sat16<T>(v[4 * q4 + 1]));
                        pk[q4][1] = pack_pair<T>(sat16<T>(v[4 * q4 + 2]), sat16<T>(v[4 * q4 + 3]));
                    }
#pragma unroll
                    for (int q4 = 0; q4 < 4; q4 += 2) {
                        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[q4][0], pk[q4 + 1][0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[q4][1], pk[q4 + 1][1], false, false);
                        u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                        const unsigned off = live ? row_b + (unsigned)(cb + 8 * (q4 + h)) * ES : kBadOff;
                        __builtin_amdgcn_raw_buffer_store_b128(o, rso, off, 0, 0);
                    }
                }
            }
        };

        // phase 0 is peeled: its first MFMA group starts the accumulators from C = 0 (a compile-time property of the copy; a run-time
        // select of the C operand costs a second set of accumulator registers)
        auto phase = [&](auto first_tag, int ph) __attribute__((always_inline)) {
            constexpr int kFirst = decltype(first_tag)::value;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                __builtin_amdgcn_s_barrier();
                if (tap == 0) pre(hbuf, 0);           // the phase's halo only became valid with this barrier
                if (!kPrefetchW) load_wt(st, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // each region = the reads (+ address VALU) of group kg+1 and the MFMAs of group kg, interleaved one read per
                // MFMA so that read issue hides in the 24 free issue cycles of every MFMA; sched_barrier(0) keeps the stages apart
                // (left alone the scheduler re-serialises read -> use, and reads issued in a block let the MFMA pipe drain)
                load_px(hbuf, 1, 1); load_wt(st, 1, 1);
                if (kFirst && tap == 0) mfma_group(0, IntTag<1>{}); else mfma_group(0, IntTag<0>{});
                interleave_reads();
                __builtin_amdgcn_sched_barrier(0);
                load_px(hbuf, 2, 0); load_wt(st, 2, 0);
                mfma_group(1, IntTag<0>{});
                interleave_reads();
                __builtin_amdgcn_sched_barrier(0);
                load_px(hbuf, 3, 1); load_wt(st, 3, 1);
                mfma_group(0, IntTag<0>{});
                interleave_reads();
                __builtin_amdgcn_sched_barrier(0);
                st = st == RING - 1 ? 0 : st + 1;
                if (tap < 8) pre(hbuf, tap + 1);      // next tap's addresses + group-0 pixels, under the last MFMA group
                if (tap == 8 && ph + 1 == nph) b_off = next_boff;      // the next step belongs to the next job (other channel half?)
                if (kPrefetchW) load_wt(st, 0, 0);    // ... and the next step's first weight fragments (tile landed at this step's barrier)
                mfma_group(1, IntTag<0>{});
                if (tap < 8) interleave_pre();
                __builtin_amdgcn_sched_barrier(0);
            }
            hbuf ^= 1;
        };
        phase(IntTag<1>{}, 0);
        for (int ph = 1; ph < nph; ++ph) phase(IntTag<0>{}, ph);
        asm volatile("" ::: "memory");
        if (hand) {
            // units 0..6 of the tile's residual arrive in the halo buffer the last phase just finished with; unit 7 (channels 112..127: this
            // wave's j = 1, qq = 1 if it owns channel half 1) is loaded here, ahead of the two barriers that hide its latency
            Rl = smem + (hbuf ^ 1) * HB;
            if (cwe == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ml = pxbase + i * 32 + r;
                    const int m = tile * p.TP + ml;
                    const unsigned off = (ml < p.TP && m < p.M) ? (unsigned)m * (unsigned)p.out_cstride * ES + (unsigned)(nblk + 112 + 8 * h) * ES : kBadOff;
                    r7[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsr, off, 0, 0));
                }
            }
            __builtin_amdgcn_s_barrier();            // A
            __builtin_amdgcn_s_barrier();            // B
        }
        // the epilogue is bound by store issue: it also re-zeroes the accumulators (and the caller resolves the next tile's pixel
        // rows) in that shadow, so the next tile starts on its first barrier
        epilogue();
    };

    set_geometry(job_half(0));
    b_off = WOFF + (cwe * 64 + r) * 128;
    resolve_centres(job_tile(0));
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
    __builtin_amdgcn_s_barrier();                          // start-up barrier: weight tile 0 is in LDS
    if (kPrefetchW) load_wt(0, 0, 0);
    for (int k = 0; k < njobs; ++k) {
        const int tile = job_tile(k), hc = job_half(k), ntile = job_tile(k + 1), nhc = job_half(k + 1);
        const int next_boff = WOFF + ((nhc >= 0 ? nhc : cw) * 64 + r) * 128;
        if (hc < 0) run_job(IntTag<4>{}, tile, next_boff);
        else run_job(IntTag<2>{}, tile, next_boff);
        if (nhc != hc) set_geometry(nhc);                  // whole -> half happens at most once, before the last job
        resolve_centres(ntile);
        asm volatile("" ::: "memory");
    }
}

}  // namespace

// development: the buffer the kStamp instantiations write their per-K-step cycle sums to (tools/step_stamps.py); while it is set the
// wave-specialised 16x16x32 launches (plain and folded) run their stamped instantiation
static unsigned* g_stamps = nullptr;
static int64_t g_stamp_bytes = 0;
extern "C" int gmk_dev_set_stamp_buffer(void* buf, int64_t bytes) { g_stamps = (unsigned*)buf; g_stamp_bytes = bytes; return 0; }

// The ONE eligibility rule of the halo kernels (gmk_conv3x3_halo_try launches by it, gmk_conv_gn_fusable answers by it): geometry of
// the tiling, LDS capacity, 32-bit buffer offsets, enough tiles to fill the chip; fused_gn adds the conditions of the in-kernel
// GroupNorm-apply (plain 3x3 only, a tile of R rows of the global row list touches at most two samples).
int gmk_halo_geometry(int B, int H, int W, int c0, int c1, int w_rows, int cout, int out_cstride, int min_tiles, int upsample,
                      int fused_gn, HaloGeometry* g) {
    if (c0 % 64 || c1 % 64 || cout % 128) return 0;
    if (c1 != 0 && c1 != c0) return 0;                        // the producers precompute byte offsets with one row size for both sources
    if (W < 4 || W > 254 || H < 2) return 0;
    const int R = 256 / W;
    if (R < 1) return 0;
    const int crossings = H % R == 0 ? 0 : (R - 1 + H - 1) / H;   // image boundaries a tile of R rows can span (none if tiles align)
    const int ner = R + 2 + 2 * crossings;
    if (ner * (W + 2) > kHaloSlots) return 0;
    const int64_t rows_total = (int64_t)B * H;
    const int64_t M = rows_total * W;
    if (upsample && ((H | W) & 1)) return 0;
    const int64_t Msrc = upsample ? M / 4 : M;
    const int64_t nb0 = Msrc * c0 * 2, nb1 = Msrc * c1 * 2;
    const int64_t nbw = (int64_t)9 * w_rows * (c0 + c1) * 2, nbo = M * out_cstride * 2;
    const int64_t lim = 0xFFFF0000ll;     // below (kBadPix * bytes-per-pixel) mod 2^32 for pixels of up to 4 KiB
    if (nb0 >= lim || nb1 >= lim || nbw >= lim || nbo >= lim || M >= 0x00FFFFFF || c0 > 2048 || c1 > 2048) return 0;
    const int64_t ntiles = (rows_total + R - 1) / R;
    if (ntiles < min_tiles) return 0;                         // not enough tiles to fill the chip: im2col kernels
    if (fused_gn && (upsample || R > H)) return 0;
    if (g) {
        // slots: the halo kernels' one-pad-column layout (the capacity check above keeps the two-column count, which conv_subpixel.hip's own layout needs)
        g->R = R; g->TP = R * W; g->slots = ner * (W + 1) + 1; g->rows_total = rows_total; g->M = M; g->ntiles = ntiles;
        g->nb0 = nb0; g->nb1 = nb1; g->nbw = nbw; g->nbo = nbo;
    }
    return 1;
}

// Returns 1 (8-compute-wave kernel) or 2 (wave-specialised kernel) if a halo kernel was launched, 0 if the problem is not eligible (caller falls back), <0 / >0 on error.
int gmk_conv3x3_halo_try(const void* src0, const void* src1, int c0, int c1, int B, int H, int W, const void* w, int w_rows,
                         int n0, int cout, const float* bias, const float* emb, int emb_stride, const void* residual,
                         void* out, int out_cstride, int min_tiles, int upsample, float* stats, int64_t stats_bytes,
                         const float* gn_scale, const float* gn_shift, int gn_stride, int dtype, hipStream_t stream) {
    HaloGeometry g;
    if (!gmk_halo_geometry(B, H, W, c0, c1, w_rows, cout, out_cstride, min_tiles, upsample, gn_scale != nullptr, &g)) return 0;
    if (gn_scale && (!gn_shift || gn_stride < c0 + c1)) return 0;
    const int R = g.R, TP = g.TP;
    const int64_t rows_total = g.rows_total, M = g.M, ntiles = g.ntiles;
    const int64_t nb0 = g.nb0, nb1 = g.nb1, nbw = g.nbw, nbo = g.nbo;
    HaloParams p;
    p.src0 = src0; p.src1 = src1; p.c0 = c0; p.c1 = c1; p.ktot = c0 + c1;
    p.shift = upsample ? 1 : 0; p.pmask = upsample == 2 ? 1 : 0;      // upsample: 1 nearest x2, 2 zero-stuffed x2 (transposed conv)
    p.B = B; p.H = H; p.W = W; p.WE = W + 1; p.R = R; p.TP = TP; p.ntiles = (int)ntiles; p.rows_total = (int)rows_total;
    p.w = w; p.w_tap_stride_b = (unsigned)w_rows * (unsigned)(c0 + c1) * 2u; p.n0 = n0;
    p.bias = bias; p.emb = emb; p.emb_stride = emb_stride; p.residual = residual; p.out = out; p.out_cstride = out_cstride;
    p.M = (int)M;
    p.nb0 = (unsigned)nb0; p.nb1 = (unsigned)nb1; p.nbw = (unsigned)nbw; p.nbo = (unsigned)nbo;
    p.gn_scale = gn_scale; p.gn_shift = gn_shift; p.gn_stride = gn_stride;
    p.stats = nullptr; p.stats_groups = out_cstride / 4; p.stamps = nullptr;
    p.variant = gmk_kernel_choice(3, "GMK_DEV_VARIANT") & 0xFF;      // code variant (A/B switches, see below)
    if (stats && cout == out_cstride && stats_bytes >= ntiles * 8 * 2 * (int64_t)(out_cstride / 4) * 2 * 4 && H * W >= 32) p.stats = stats;
    p.inv_hp2 = 1.0f / (float)(H + 2); p.inv_h = 1.0f / (float)H; p.inv_we = 1.0f / (float)(W + 1); p.inv_w = 1.0f / (float)W;
    const int ncu = gmk_cu_limit();
    dim3 grid((unsigned)(ntiles < ncu ? ntiles : ncu), cout / 128);
    p.nfull = (int)ntiles; p.nhalf = 0;
    {   // tail balancing of the wave-specialised kernel: a last round of at most G/2 tiles runs as twice as many half jobs
        const int G = (int)grid.x, rem = (int)(ntiles % G);
        // not for variant 4 (its vmcnt waits count 4 weight DMAs per step); variant 6 is the A/B switch
        if (ntiles > G && rem > 0 && 2 * rem <= G && p.variant != 4 && p.variant != 6) { p.nfull = (int)ntiles - rem; p.nhalf = 2 * rem; }
        // Small problems (round 4: the reference's own sampler sizes, 25 images in `evaluate`, diffusion_model.py:98-104): with at most
        // half as many tiles as CUs EVERY tile runs as two half jobs on two CUs - the launch is one tile time long either way, and a
        // half job's K-step carries half the MFMA work.  GMK_DEV_VARIANT=8 keeps whole jobs (A/B).
        else if (2 * ntiles <= ncu && p.variant != 4 && p.variant != 6 && p.variant != 8) { p.nfull = 0; p.nhalf = 2 * (int)ntiles; grid.x = (unsigned)p.nhalf; }
    }
    const bool use16 = p.variant != 32 && p.variant != 1 && p.variant != 3;      // (variants 1 / 3: the 8-compute-wave kernel)
    // MFMA shape of the consumers: bit-identical results either way; v_mfma_f32_16x16x32 measured +2 ... +4 % at K = 2304 and at
    // 64- / 32- / 14-pixel rows in round 2.  At 28 x 28 with K = 1152 it had measured -2 % with bf16 operands and kept the 32 x 32 x 16 form
    // there; re-measured in round 3 (fp16 forward operands, whole bench, same box): +1.2 ... 1.9 % train, +3.5 % sampler at 1x28x28 with
    // the 16 x 16 x 32 form everywhere, so it is the form of every launch now.  GMK_DEV_VARIANT=32 forces the other.
    // kind: 0 the 8-compute-wave kernel (variants 1, 3, or statistics wanted), 1 fused GroupNorm-apply + SiLU in the producer waves
    // (tables from gmk_gn_stats), 2 variant 4 (no weight prefetch), 3 the wave-specialised kernel
    const bool regfill = p.variant == 11 && !gn_scale && !p.stats && !upsample;      // experiment: halo pieces through registers, no transform
    // (an `emb` addend in the epilogue - not used by the U-Net any more: conv1's bias + embedding ride with the consuming GroupNorm - is served by
    // the 8-compute-wave kernel only since round 6: the wave-specialised consumers' epilogue carries no per-sample loads)
    const int kind = emb ? 0 : (gn_scale || regfill) ? 1 : (p.variant == 4 && !p.stats) ? 2 : (!p.stats && (use16 || (p.variant != 1 && p.variant != 3))) ? 3 : 0;
    auto launch = [&](auto tag) {
        typedef decltype(tag) T;
        if (kind == 1) {
            if (use16) conv3x3_halo_ws_kernel<T, true, 16, true><<<grid, 512, 0, stream>>>(p);
            else conv3x3_halo_ws_kernel<T, true, 32, true><<<grid, 512, 0, stream>>>(p);
        } else if (kind == 2) conv3x3_halo_ws_kernel<T, false><<<grid, 512, 0, stream>>>(p);
        else if (kind == 3) {
            // four-slot weight ring (6-piece halos) where a tile fits 384 slots and there is no residual: 32 x 32, 16 x 16, 14 x 14 (GMK_DEV_VARIANT=13: the A/B switch)
            const bool ring4 = use16 && !residual && g.slots <= 384 && p.variant != 13;
            if (use16 && g_stamps && g_stamp_bytes >= (int64_t)grid.x * 512) {
                p.stamps = g_stamps;
                if (ring4) conv3x3_halo_ws_kernel<T, true, 16, false, false, true, false, 0, true><<<grid, 512, 0, stream>>>(p);
                else conv3x3_halo_ws_kernel<T, true, 16, false, false, true><<<grid, 512, 0, stream>>>(p);
            } else if (ring4) conv3x3_halo_ws_kernel<T, true, 16, false, false, false, false, 0, true><<<grid, 512, 0, stream>>>(p);
            else if (use16 && residual) conv3x3_halo_ws_kernel<T, true, 16, false, false, false, false, 1><<<grid, 512, 0, stream>>>(p);
            else if (use16) conv3x3_halo_ws_kernel<T, true, 16, false, false, false, false, 0><<<grid, 512, 0, stream>>>(p);
            else conv3x3_halo_ws_kernel<T, true, 32><<<grid, 512, 0, stream>>>(p);
        } else conv3x3_halo_kernel<T><<<grid, 512, 0, stream>>>(p);
    };
    if (dtype == GMK_F16) launch(f16_t{});
    else launch(bf16_t{});
    return kind == 0 ? 1 : 2;
}

// ---- conv2 + folded 1x1 skip convolution of an up-path ResBlock (reference simple_unet.py:172-186) --------------------------------
static int skipfold_geometry(int B, int H, int W, int c0, int cs, int cout, int w_rows, int out_cstride, int min_tiles, HaloGeometry* g) {
    if (c0 != 128 || cs != 128 || cout != 128) return 0;          // two halo phases + eight 32-channel sub-phases: the C = 128 nets
    if (!gmk_halo_geometry(B, H, W, c0, 0, w_rows, cout, out_cstride, min_tiles, 0, 0, g)) return 0;
    return (int64_t)g->M * cs * 2 < 0xFFFF0000ll;
}

extern "C" int gmk_conv3x3_skipfold_ok(int B, int H, int W, int c0, int cs, int cout) {
    const int force = gmk_kernel_choice(0, "GMK_CONV_KERNEL");
    if (force != 0 && force != 3) return 0;
    HaloGeometry g;
    return skipfold_geometry(B, H, W, c0, cs, cout, cout, cout, 1, &g);
}

extern "C" int gmk_conv3x3_skipfold(const void* src, int c0, int B, int H, int W, const void* w, int w_rows, int n0, int cout,
                                    const float* bias, const void* sk0, const void* sk1, int cs, const void* wsk, int wsk_rows, int nsk0,
                                    const float* bias_sk, void* out, int out_cstride, int dtype, void* stream) {
    GMK_REQUIRE(src && w && sk0 && sk1 && wsk && bias && bias_sk && out, "gmk_conv3x3_skipfold: null pointer");
    GMK_REQUIRE(dtype == GMK_BF16 || dtype == GMK_F16, "gmk_conv3x3_skipfold: 16-bit types only (dtype %d)", dtype);
    GMK_REQUIRE(n0 >= 0 && n0 + cout <= w_rows && nsk0 >= 0 && nsk0 + cout <= wsk_rows && out_cstride >= cout,
                "gmk_conv3x3_skipfold: bad output channels n0=%d nsk0=%d cout=%d", n0, nsk0, cout);
    HaloGeometry g;
    GMK_REQUIRE(skipfold_geometry(B, H, W, c0, cs, cout, w_rows, out_cstride, 1, &g),
                "gmk_conv3x3_skipfold: shape B=%d %dx%d c0=%d cs=%d cout=%d is not foldable (ask gmk_conv3x3_skipfold_ok first)", B, H, W, c0, cs, cout);
    HaloParams p = {};
    p.src0 = src; p.src1 = nullptr; p.c0 = c0; p.c1 = 0; p.ktot = c0;
    p.shift = 0; p.pmask = 0;
    p.B = B; p.H = H; p.W = W; p.WE = W + 1; p.R = g.R; p.TP = g.TP; p.ntiles = (int)g.ntiles; p.rows_total = (int)g.rows_total;
    p.w = w; p.w_tap_stride_b = (unsigned)w_rows * (unsigned)c0 * 2u; p.n0 = n0;
    p.bias = bias; p.emb = nullptr; p.emb_stride = 0; p.residual = nullptr; p.out = out; p.out_cstride = out_cstride;
    p.M = (int)g.M;
    p.nb0 = (unsigned)g.nb0; p.nb1 = 0; p.nbw = (unsigned)g.nbw; p.nbo = (unsigned)g.nbo;
    p.stats = nullptr; p.stats_groups = out_cstride / 4;
    p.variant = gmk_kernel_choice(3, "GMK_DEV_VARIANT") & 0xFF;
    p.inv_hp2 = 1.0f / (float)(H + 2); p.inv_h = 1.0f / (float)H; p.inv_we = 1.0f / (float)(W + 1); p.inv_w = 1.0f / (float)W;
    p.sk0 = sk0; p.sk1 = sk1; p.wsk = wsk; p.bias2 = bias_sk; p.cs = cs; p.sk_ktot = 2 * cs; p.nsk0 = nsk0;
    p.nbs = (unsigned)((int64_t)g.M * cs * 2); p.nbws = (unsigned)((int64_t)wsk_rows * 2 * cs * 2);
    const int ncu = gmk_cu_limit();
    dim3 grid((unsigned)(g.ntiles < ncu ? g.ntiles : ncu), cout / 128);
    p.nfull = (int)g.ntiles; p.nhalf = 0;
    {   // half jobs as in gmk_conv3x3_halo_try: the last partly filled round, or every tile of a small problem
        const int G = (int)grid.x, rem = (int)(g.ntiles % G);
        if (g.ntiles > G && rem > 0 && 2 * rem <= G && p.variant != 6) { p.nfull = (int)g.ntiles - rem; p.nhalf = 2 * rem; }
        else if (2 * g.ntiles <= ncu && p.variant != 6 && p.variant != 8) { p.nfull = 0; p.nhalf = 2 * (int)g.ntiles; grid.x = (unsigned)p.nhalf; }
    }
    // merged dense sub-phases (9 barriers per phase) where a tile's halo fits 6 of the 7 fill pieces: 32 x 32, 16 x 16 (GMK_DEV_VARIANT=12: the A/B switch)
    const bool merge = g.slots <= 384 && p.variant != 12;
    const bool stamp = g_stamps && g_stamp_bytes >= (int64_t)grid.x * 512;      // development: the stamped instantiations (tools/step_stamps.py)
    if (stamp) p.stamps = g_stamps;
    auto launch = [&](auto tag) {
        typedef decltype(tag) T;
        if (merge && stamp) conv3x3_halo_ws_kernel<T, true, 16, false, true, true, true><<<grid, 512, 0, gmk_stream(stream)>>>(p);
        else if (merge) conv3x3_halo_ws_kernel<T, true, 16, false, true, false, true, 0><<<grid, 512, 0, gmk_stream(stream)>>>(p);
        else if (stamp) conv3x3_halo_ws_kernel<T, true, 16, false, true, true><<<grid, 512, 0, gmk_stream(stream)>>>(p);
        else conv3x3_halo_ws_kernel<T, true, 16, false, true, false, false, 0><<<grid, 512, 0, gmk_stream(stream)>>>(p);
    };
    if (dtype == GMK_F16) launch(f16_t{});
    else launch(bf16_t{});
    gmk_note_kernel(7);
    return gmk_check_launch("gmk_conv3x3_skipfold");
}
