// Sub-pixel (output-parity) form of the two x2 resampling convolutions of the U-Net, on the LDS-halo architecture of conv_halo.hip:
//   * `Upsample` = F.interpolate(nearest, x2) then Conv2d(C, C, 3, padding=1)          (reference gms/diffusion/simple_unet.py:112-122)
//   * the data gradient of `Downsample` = Conv2d(C, C, 3, stride=2, padding=1)          (reference simple_unet.py:75-84), a transposed convolution
// Both map a LOW-resolution tensor [B][H][W][128] to a HIGH-resolution one [B][2H][2W][128], and in both an output pixel (2i + a, 2j + b)
// only ever meets the low-resolution pixels (i + dy, j + dx) of a parity-dependent tap list:
//   upsample     4 taps per parity: rows {i + a - 1, i + a}, columns {j + b - 1, j + b}; the 3x3 weights that land on one low-resolution pixel
//                are pre-summed in fp32 at pack time (gmk_pack_upsample_weight: [16 = parity x tap][Cout][Cin]) - 16 tap-products per low-resolution
//                pixel where the nearest-x2 form of conv_halo.hip (`shift` addressing) multiplies 36
//   transposed   1 / 2 / 2 / 4 taps per parity over the ordinary 9-tap data-gradient pack - 9 tap-products where the zero-stuffed form multiplies 36
//
// Kernel: a tile is R whole rows of the LOW-resolution global row list (R * W <= 256 pixels; 1024 output pixels).  Its halo - BOTH 64-channel
// halves, one per half-buffer - stays in LDS for the whole tile (the nearest-x2 form re-fetched every low-resolution pixel four times into LDS and
// re-read it from there nine times), the consumers run the four parities one after the other over the same halo, each with its own epilogue that
// scatters to the parity's output pixels.  K-steps of a tile: for parity p: for channel half h: for tap t  (32 for upsample, 18 for transposed);
// weight tiles [128 cout][64 k] stream through the 3-deep ring exactly as in conv3x3_halo_ws_kernel (two K-steps ahead).  Halo traffic: half-buffer 1
// of a tile is filled during the first segment of that tile (parity 0, half 0 reads buffer 0 only), half-buffer 0 of the NEXT tile during the
// last segment (parity 3, half 1 reads buffer 1 only): 7 pieces in a window of NT0 / NT3 K-steps, everything else carries weights only.
// Wave specialisation, LDS map, swizzles, counted vmcnt waits, one s_barrier per K-step and the 16x16x32 consumer are those of conv_halo.hip.
//
// The third form runs the other way (HIGH -> LOW): the data gradient of `Upsample`, dx_low = sumpool2x2(dgrad3x3(dy_high)) in the reference's
// autograd, is the transpose of the sub-pixel forward: dx_low[i][j] = sum over the four parity VIEWS V_ab[r][c] = dy[2r + a][2c + b] of a 2x2-tap
// convolution with the transposed pre-summed matrices - 16 tap-products per low-resolution pixel, one launch, no high-resolution intermediate and no
// pooling pass.  Here the eight (view, channel half) segments of a tile each have their OWN halo: the two half-buffers alternate per segment, the
// next segment's 7 pieces stream in during the current segment's 4 K-steps (conv_halo.hip's double buffering at 4 steps per phase instead of 9),
// one accumulator set, one dense epilogue per tile.
#include "gmk_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

constexpr int kHaloSlots = 448;
constexpr int kHB = kHaloSlots * 128;          // 57344 bytes per halo half-buffer
constexpr int kWOFF = 2 * kHB;                 // weight ring offset
constexpr int kWST = 16384;

struct SubParams {
    const void* src;                           // LOW -> HIGH: the low-resolution source [B][H][W][128]; HIGH -> LOW: the high-resolution gradient [B][2H][2W][128]
    int B, H, W, WE, R, TP, ntiles, M;         // the LOW-resolution grid in every mode; M = B * H * W
    const void* w; unsigned w_tap_stride_b; int n0;
    const float* bias; const void* residual; void* out; int out_cstride;
    unsigned nb0, nbw, nbo;
    float inv_hp2, inv_h, inv_we, inv_w;
};

template <int N> struct IntTag { static constexpr int value = N; };
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(IntTag<I>{}); static_for<I + 1, N>(f); }
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

__device__ __forceinline__ int div_small(int a, float inv) { return (int)(((float)a + 0.5f) * inv); }

// ---- the K-step plan of a tile (compile time) -----------------------------------------------------------------------------------------
struct Step { int ph, a, b, dy, dx, ptap, half, buf, first_ph, last_ph; };      // parity (order index, a, b), low-resolution offset, pack index, channel half, half-buffer
// kMode 0: upsample (parities in order (0,0) (0,1) (1,0) (1,1), 4 taps each, pack index 4 (2a + b) + 2 ty + tx, dy = a - 1 + ty, dx = b - 1 + tx)
// kMode 1: transposed (parity order (1,1) (1,0) (0,0) (0,1): the 4-tap parity first and a 2-tap one last, so the fill windows are 4 and 2 steps, and the two
//          column parities of an output row - adjacent 256-byte pixels - are written back to back;
//          per axis: parity 0 meets tap 1 (offset 0), parity 1 meets tap 0 (offset 0) and tap 2 (offset +1); pack index 3 ty + tx)
// kMode 2: `Upsample` data gradient (HIGH -> LOW): segments (view (a, b), channel half) in order, 4 taps each; tap (ty, tx) of view (a, b) reads the
//          view at low-resolution offset (1 - a - ty, 1 - b - tx) (the transpose of kMode 0's (a - 1 + ty, b - 1 + tx)); pack index as kMode 0;
//          half-buffer = segment & 1; one accumulation over all 32 steps
template <int kMode> struct Plan {
    static constexpr bool kDown = kMode == 2;
    static constexpr int pa(int i) { return kMode != 1 ? (i >> 1) : i < 2 ? 1 : 0; }
    static constexpr int pb(int i) { return kMode != 1 ? (i & 1) : (i == 0 || i == 3) ? 1 : 0; }
    static constexpr int nax(int par) { return kMode != 1 ? 2 : (par ? 2 : 1); }            // taps along one axis
    static constexpr int ntaps(int i) { return nax(pa(i)) * nax(pb(i)); }
    static constexpr int nsteps() { return 2 * (ntaps(0) + ntaps(1) + ntaps(2) + ntaps(3)); }
    static constexpr int axis_tap(int par, int k) { return kMode != 1 ? k : (par ? 2 * k : 1); }          // (transposed) 3x3 tap index along the axis
    static constexpr int axis_off(int par, int k) { return kMode == 0 ? par - 1 + k : kMode == 2 ? 1 - par - k : (par ? k : 0); }      // low-resolution offset
    static constexpr Step at(int s) {
        int base = 0;
        for (int i = 0; i < 4; ++i) {
            const int nt = ntaps(i);
            if (s < base + 2 * nt) {
                const int l = s - base, half = l / nt, t = l % nt;
                const int a = pa(i), b = pb(i), nx = nax(b);
                const int ky = t / nx, kx = t % nx;
                const int ptap = kMode != 1 ? 4 * (2 * a + b) + 2 * ky + kx : 3 * axis_tap(a, ky) + axis_tap(b, kx);
                return Step{i, a, b, axis_off(a, ky), axis_off(b, kx), ptap, half, kDown ? ((2 * i + half) & 1) : half,
                            kDown ? s == 0 : l == 0, kDown ? s == nsteps() - 1 : l == 2 * nt - 1};
            }
            base += 2 * nt;
        }
        return Step{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    }
    static constexpr int S = nsteps(), NT0 = ntaps(0), NT3 = ntaps(3);
    // halo pieces issued behind the weights of step s.  LOW -> HIGH: [lo, hi) of half-buffer 1 of THIS tile (window A: steps 0 .. NT0 - 1) or of
    // half-buffer 0 of the NEXT tile (window B: steps S - NT3 .. S - 1).  HIGH -> LOW: of the NEXT segment, spread over the current segment's 4 steps.
    static constexpr int win_lo(int k, int n) { return (7 * k + n - 1) / n; }
    static constexpr int fill_lo(int s) { return kDown ? win_lo(s & 3, 4) : s < NT0 ? win_lo(s, NT0) : s >= S - NT3 ? win_lo(s - (S - NT3), NT3) : 0; }
    static constexpr int fill_hi(int s) { return kDown ? win_lo((s & 3) + 1, 4) : s < NT0 ? win_lo(s + 1, NT0) : s >= S - NT3 ? win_lo(s - (S - NT3) + 1, NT3) : 0; }
    static constexpr int nfill(int s) { return fill_hi(s) - fill_lo(s); }
    // steps whose half-buffer only became valid with their own barrier (everything older has to have landed: vmcnt(0))
    static constexpr bool fresh_halo(int s) { return kDown ? (s & 3) == 0 : (s == 0 || s == NT0); }
    // the consumers read a step's first pixel fragments behind its barrier (instead of under the previous step's last MFMA group) there, and
    // at the start of every parity of the LOW -> HIGH forms (the epilogue needs the registers)
    static constexpr bool post_barrier_loads(int s) { return fresh_halo(s) || at(s).first_ph; }
};
static_assert(Plan<0>::S == 32 && Plan<1>::S == 18 && Plan<0>::NT0 == 4 && Plan<1>::NT0 == 4 && Plan<1>::NT3 == 2, "K-step plan");
static_assert(Plan<1>::at(0).ptap == 0 && Plan<1>::at(3).ptap == 8 && Plan<1>::at(3).dy == 1 && Plan<1>::at(8).ptap == 1 && Plan<1>::at(9).ptap == 7 &&
              Plan<1>::at(9).dy == 1 && Plan<1>::at(9).last_ph == 0 && Plan<1>::at(11).last_ph == 1 && Plan<1>::at(12).ptap == 4 && Plan<1>::at(13).last_ph == 1 &&
              Plan<1>::at(14).ptap == 3 && Plan<1>::at(15).ptap == 5 && Plan<1>::at(15).dx == 1 && Plan<1>::at(17).half == 1, "transposed plan");
static_assert(Plan<2>::S == 32 && Plan<2>::at(0).dy == 1 && Plan<2>::at(3).dy == 0 && Plan<2>::at(3).dx == 0 && Plan<2>::at(4).buf == 1 && Plan<2>::at(8).buf == 0 &&
              Plan<2>::at(8).b == 1 && Plan<2>::at(8).dx == 0 && Plan<2>::at(9).dx == -1 && Plan<2>::at(31).dy == -1 && Plan<2>::at(31).last_ph == 1 &&
              Plan<2>::at(7).last_ph == 0 && Plan<2>::at(28).ptap == 12 && Plan<2>::nfill(3) == 1 && Plan<2>::nfill(4) == 2, "upsample data-gradient plan");

template <typename T, int kMode>
__global__ __launch_bounds__(512, 2) void conv_subpixel_ws_kernel(const SubParams p) {
    typedef Plan<kMode> PL;
    typedef typename Frag16<T>::type frag_t;
    typedef typename Frag16<T>::half_type half_t;
    constexpr int ES = 2;
    constexpr int S = PL::S, NT0 = PL::NT0, NT3 = PL::NT3;
    constexpr bool kDown = PL::kDown;
    fp16_saturating_stores<T>();
    __shared__ __attribute__((aligned(16))) char smem[kWOFF + 3 * kWST];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int H = p.H, W = p.W, WE = p.WE;
    constexpr unsigned kBadPix = 0x00FFFFFFu;
    constexpr unsigned kBadOff = 0xFFFFFF00u;
    if ((int)blockIdx.x >= p.ntiles) return;
    // tiles of this workgroup: pg, pg + G, ...; XCD-aware order as in conv3x3_halo_ws_kernel (neighbouring tiles share halo rows: the tiles
    // of a round are dealt out in 8 contiguous runs, one per XCD)
    const int G = gridDim.x, g = blockIdx.x;
    const int pg = (G & 7) == 0 ? (g & 7) * (G >> 3) + (g >> 3) : g;
    const int njobs = (p.ntiles - pg + G - 1) / G;
    auto job_tile = [&](int k) { return k < njobs ? pg + k * G : p.ntiles; };

    if (wave >= 4) {
        // =========================================== producer waves ===========================================
        const int pw = __builtin_amdgcn_readfirstlane(wave) - 4;
        const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src), 0, (int)p.nb0, 0x00020000);
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
        unsigned w_off[4];                    // weight tile rows: 32*pw + 8*u + lrow of [ptap][w_rows][128]
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = 32 * pw + 8 * u + lrow;
            w_off[u] = (unsigned)(p.n0 + row) * 128u * ES + (unsigned)((lch ^ ((row >> 1) & 7)) << 4);
        }
        unsigned hvoff[7][2];                 // byte offset of this lane's slot of piece j in the source (pixel x 256 B + swizzled chunk)
        auto resolve_piece = [&](int tile, int j, int va = 0, int vb = 0) {      // HIGH -> LOW: of the parity view (va, vb) of the high-resolution source
            const bool exists = tile < p.ntiles;
            const int gr0 = tile * p.R;
            const int b0 = gr0 / H;
            const int y0 = gr0 - b0 * H;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                int xe = f_xe0[u] + j * f_dxe, er = f_er0[u] + j * f_der;
                const int wraps = div_small(xe, p.inv_we);
                xe -= wraps * WE; er += wraps;
                const int n = er * WE + xe - 2 * er - 1;
                const int E = er + y0;
                const int k = div_small(E, p.inv_hp2);
                const int y = E - k * (H + 2) - 1;
                const int x = xe - 1;
                const int b = b0 + k;
                const bool ok = exists && y >= 0 && y < H && x >= 0 && x < W && b < p.B;
                const unsigned pix = !ok ? kBadPix : kDown ? (unsigned)(((b * H + y) * 2 + va) * (2 * W) + 2 * x + vb) : (unsigned)((b * H + y) * W + x);
                hvoff[j][u] = __umul24(pix, 128u * ES) + ((unsigned)(lch ^ ((n >> 1) & 7)) << 4);
            }
        };
        auto issue_fill = [&](int hbuf, int j, int half) {          // channel half `half` of the source into half-buffer hbuf
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                GMK_LDS char* dst = (GMK_LDS char*)(smem + hbuf * kHB + j * 8192 + (pw * 2 + u) * 1024);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (GMK_LDS void*)dst, 16, hvoff[j][u], (unsigned)half << 7, 0, 0);
            }
        };
        auto issue_w = [&](int stage, int ptap, int half) {
            const unsigned wk = (unsigned)ptap * p.w_tap_stride_b + ((unsigned)half << 7);      // scalar: soffset
            GMK_LDS char* dst = (GMK_LDS char*)(smem + kWOFF + stage * kWST + pw * 4096);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (GMK_LDS void*)(dst + u * 1024), 16, w_off[u], wk, 0, 0);
        };
        int sq = 2;
        int tile = job_tile(0);
#pragma unroll
        for (int j = 0; j < 7; ++j) { resolve_piece(tile, j); issue_fill(0, j, 0); }
        issue_w(0, PL::at(0).ptap, PL::at(0).half);
        issue_w(1, PL::at(1).ptap, PL::at(1).half);
        wait_vmcnt<4>();
        __builtin_amdgcn_s_barrier();                      // start-up barrier: weight tile 0 is in LDS
        for (int k = 0; k < njobs; ++k) {
            const int ntile = job_tile(k + 1);
            asm volatile("" : "+s"(sq));      // the ring position stays a run-time value (with S a multiple of 3 the compiler would specialise every step's addresses)
            static_for<0, S>([&](auto tag) __attribute__((always_inline)) {
                constexpr int s = decltype(tag)::value;
                // at barrier s the weight tile of step s + 1 (the first 4 ops of step s - 1) must have landed - the consumers read its first
                // fragments before barrier s + 1; only the halo pieces issued behind it may still fly.  Steps 0 and NT0 are the first to read
                // half-buffer 0 / 1 of this tile: everything older has to be there.
                if constexpr (PL::fresh_halo(s)) wait_vmcnt<0>();
                else wait_vmcnt<2 * PL::nfill(s - 1)>();
                __builtin_amdgcn_s_barrier();
                constexpr Step s2 = PL::at((s + 2) % S);              // (the next tile runs the same convolution: same weights)
                issue_w(sq, s2.ptap, s2.half);
                if constexpr (kDown) {
                    // the NEXT segment's halo (the other half-buffer): the same view's second channel half, or a new view (of the next tile
                    // behind the last segment), whose pieces are resolved right before they are issued
                    constexpr Step sg = PL::at((s | 3) + 1 < S ? (s | 3) + 1 : 0);
#pragma unroll
                    for (int j = PL::fill_lo(s); j < PL::fill_hi(s); ++j) {
                        if constexpr (sg.half == 0) resolve_piece((s | 3) + 1 < S ? tile : ntile, j, sg.a, sg.b);
                        issue_fill(sg.buf, j, sg.half);
                    }
                } else if constexpr (s < NT0) {
#pragma unroll
                    for (int j = PL::fill_lo(s); j < PL::fill_hi(s); ++j) issue_fill(1, j, 1);
                } else if constexpr (s >= S - NT3) {
#pragma unroll
                    for (int j = PL::fill_lo(s); j < PL::fill_hi(s); ++j) { resolve_piece(ntile, j); issue_fill(0, j, 0); }
                }
                sq = sq == 2 ? 0 : sq + 1;
            });
            tile = ntile;
        }
        wait_vmcnt<0>();                                    // drain the speculative DMA before the LDS is released
        return;
    }

    // =============================================== consumer waves ===============================================
    // v_mfma_f32_16x16x32 form of conv3x3_halo_ws_kernel: wave w owns low-resolution pixels 64 w .. 64 w + 63 (4 blocks of 16) x all 128 output
    // channels (8 blocks of 16).  Lane = (r16 = lane & 15, q = lane >> 4): A fragment = 16 channel rows x k chunk (4 k2 + q), B fragment =
    // 16 pixel columns x the same chunk; D: lane holds channels 4 q .. 4 q + 3 of pixel r16.
    const int r16 = lane & 15, q = lane >> 4;
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.nbo, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.residual ? p.residual : p.out), 0, (int)p.nbo, 0x00020000);
    const int pxbase = wave * 64;
    int rit[4], px_x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ml = pxbase + i * 16 + r16;
        rit[i] = div_small(ml, p.inv_w);
        px_x[i] = ml - rit[i] * W;
    }
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
            int n = s - 2 * er - 1;
            if (s < WE + 1 || s >= kHaloSlots - WE - 1) { s = WE + 1; n = WE - 2; }   // dead rows (m_local >= TP): any in-range slot
            cslot[i] = s;
            cn[i] = n;
        }
    };
    f32x4 acc[8][4];          // [channel block][pixel block]
    const int swz = (r16 >> 1) & 7;          // weight rows 16 cb + r16: (row >> 1) & 7 does not depend on cb
    const int b_off = kWOFF + r16 * 128;
    frag_t px[2][4], wt[2][2];
    int rowb[4], sw[4];
    int st = 0;
    auto load_wt = [&](int stg, int k2, int pair, int set) {
        const char* Wb = smem + stg * kWST + b_off + pair * 4096;
        const int coff = ((k2 * 4 + q) ^ swz) << 4;
        wt[set][0] = *reinterpret_cast<const frag_t*>(Wb + coff);
        wt[set][1] = *reinterpret_cast<const frag_t*>(Wb + 2048 + coff);
    };
    auto addr = [&](int dy, int dx) {
        const int tapoff = dy * WE + dx, tapoff_n = dy * W + dx;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int c = cslot[i], n = cn[i];
            asm volatile("" : "+v"(c), "+v"(n));      // keep the per-tap addresses out of loop-invariant hoisting
            rowb[i] = (c + tapoff) << 7;
            sw[i] = ((n + tapoff_n) >> 1) & 7;
        }
    };
    auto load_px = [&](int hb, int k2, int set, int i0, int i1) {
        const char* Hb = smem + hb * kHB;
#pragma unroll
        for (int i = i0; i < i1; ++i)
            px[set][i] = *reinterpret_cast<const frag_t*>(Hb + rowb[i] + (((k2 * 4 + q) ^ sw[i]) << 4));
    };
    auto mfma_group = [&](int pair, int wset, int pset, auto fresh_tag) __attribute__((always_inline)) {
        constexpr bool kFresh = decltype(fresh_tag)::value != 0;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[2 * pair + j][i] = mfma_16x16x32<T>(wt[wset][j], px[pset][i], kFresh ? z : acc[2 * pair + j][i]);
    };
    // ---- epilogue of one parity (a, b): lane (r16, q) holds channels 16 cb + 4 q .. + 3 of low-resolution pixel 16 i + r16.  One
    // v_permlane16_swap per dword between the packed values of pixel blocks (i, i + 1) leaves every lane with 8 consecutive channels (16 bytes)
    // of ONE pixel, stored at output pixel (2 row + a, 2 x + b): a pixel's 256 bytes are two whole cache lines, so the stride-2 scatter
    // costs no partial-line writes, and the four parities of a tile complete each other's DRAM pages within one tile time.
    auto epilogue = [&](int tile, int a, int b) __attribute__((always_inline)) {
        // (the consumers sit at the 256-register limit: the bias is loaded per channel-block pair; addresses are recomputed per epilogue)
        unsigned row_b[2];
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
            int ml = pxbase + (2 * ip + (q & 1)) * 16 + r16;                // the pixel this lane stores after the swap
            asm volatile("" : "+v"(ml));                                    // (hoisted out of the tile loop these cost 16 registers)
            const int m = tile * p.TP + ml;
            const bool live = ml < p.TP && m < p.M;
            unsigned mo = (unsigned)m;                                       // HIGH -> LOW: the low-resolution pixel itself
            if constexpr (!kDown) {
                const int srow = div_small(ml, p.inv_w), sx = ml - srow * W;     // its row in the tile and column, low resolution
                const int orow = 2 * (tile * p.R + srow) + a;                    // row of the output's global row list
                mo = (unsigned)(orow * (2 * W) + 2 * sx + b);
            }
            row_b[ip] = live ? mo * (unsigned)p.out_cstride * ES + (unsigned)((q >> 1) * 8) * ES : kBadOff;
        }
        constexpr bool kRes = kMode == 1;          // only the stride-2 data gradient is ever called with a residual (the skip path's gradient)
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
            // the residual of this pixel: all eight 16-byte loads up front (the fragment registers are dead during the epilogue; two at a time
            // left each pair's HBM latency exposed: 6.6 us per epilogue against 2.9 without a residual; sixteen - both pixels - spill and gain nothing)
            u32x4 rres[4][2];
            if (kRes && p.residual) {
#pragma unroll
                for (int cp = 0; cp < 4; ++cp)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        rres[cp][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsr, row_b[ip], (2 * cp + j) * 16 * ES, 0));
            }
#pragma unroll
            for (int cp = 0; cp < 4; ++cp) {
                float bz[2][4];
                if (p.bias) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) load4(p.bias + (2 * cp + j) * 16 + 4 * q, bz[j]);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int cb = 2 * cp + j;
                    float v0[4], v1[4];         // pixel blocks 2 ip and 2 ip + 1 in the accumulator layout
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v0[e] = acc[cb][2 * ip][e]; v1[e] = acc[cb][2 * ip + 1][e]; }
                    if (p.bias) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v0[e] += bz[j][e]; v1[e] += bz[j][e]; }
                    }
                    if (kRes && p.residual) {
                        const u32x4 Rr = rres[cp][j];
                        const auto s0 = __builtin_amdgcn_permlane16_swap(Rr[0], Rr[2], false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(Rr[1], Rr[3], false, false);
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
                    __builtin_amdgcn_raw_buffer_store_b128(o, rso, row_b[ip], cb * 16 * ES, 0);
                }
            }
        }
    };

    resolve_centres(job_tile(0));
    __builtin_amdgcn_s_barrier();                          // start-up barrier: weight tile 0 is in LDS
    load_wt(0, 0, 0, 0);
    for (int k = 0; k < njobs; ++k) {
        const int tile = job_tile(k);
        asm volatile("" : "+s"(st));          // (see the producers)
        static_for<0, S>([&](auto tag) __attribute__((always_inline)) {
            constexpr int s = decltype(tag)::value;
            constexpr Step si = PL::at(s);
            constexpr bool post = PL::post_barrier_loads(s);
            constexpr bool pref_next = s + 1 < S && !PL::post_barrier_loads(s + 1);      // next step's addresses + first pixel fragments under this step's last group
            __builtin_amdgcn_s_barrier();
            if constexpr (post) { addr(si.dy, si.dx); load_px(si.buf, 0, 0, 0, 4); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g8 = 0; g8 < 8; ++g8) {
                const int k2 = g8 / 4, pair = g8 % 4;
                const bool last = g8 == 7;
                // reads issued under this group's 8 MFMAs: the next group's weight fragments; the second k2 half's pixel fragments during the
                // first half's last two groups; in the step's last group the next step's addresses, first pixel fragments and first weight
                // fragments (its weight tile landed at this step's barrier)
                if (!last) load_wt(st, (g8 + 1) / 4, (g8 + 1) % 4, (g8 + 1) & 1);
                if (k2 == 0 && pair == 2) load_px(si.buf, 1, 1, 0, 2);
                if (k2 == 0 && pair == 3) load_px(si.buf, 1, 1, 2, 4);
                if (last) {
                    st = st == 2 ? 0 : st + 1;
                    if constexpr (pref_next) {
                        constexpr Step sn = PL::at(s + 1 < S ? s + 1 : 0);
                        addr(sn.dy, sn.dx); load_px(sn.buf, 0, 0, 0, 4);
                    }
                    load_wt(st, 0, 0, 0);
                }
                if (si.first_ph && k2 == 0) mfma_group(pair, g8 & 1, k2, IntTag<1>{}); else mfma_group(pair, g8 & 1, k2, IntTag<0>{});
                if (last && pref_next) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                    }
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                } else {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (si.last_ph != 0) {
                asm volatile("" ::: "memory");
                epilogue(tile, si.a, si.b);
                asm volatile("" ::: "memory");
            }
        });
        resolve_centres(job_tile(k + 1));
        asm volatile("" ::: "memory");
    }
}

// nn.Conv2d weight [Cout][Cin][3][3] fp32 -> the 16 pre-summed 2x2-tap matrices of the sub-pixel form, [4 (2a + b) + 2 ty + tx][Cout][Cin] (forward) and
// their transposes [..][Cin][Cout] (data gradient): parity a = 0 meets low-resolution rows i - 1 (3x3 row 0) and i (rows 1 + 2); parity 1 rows i
// (rows 0 + 1) and i + 1 (row 2); columns alike.  Summed in fp32, rounded once.
template <typename TF, typename TD>
__global__ __launch_bounds__(256) void pack_upsample_kernel(const float* __restrict__ w, TF* __restrict__ wf, TD* __restrict__ wd, int cout, int cin) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= cout * cin) return;
    const int co = idx / cin, ci = idx - co * cin;
    const float* s = w + (int64_t)idx * 9;
    float v[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) v[i] = s[i];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int ty = 0; ty < 2; ++ty)
#pragma unroll
                for (int tx = 0; tx < 2; ++tx) {
                    const int y0 = a == 0 ? (ty == 0 ? 0 : 1) : (ty == 0 ? 0 : 2), y1 = a == 0 ? (ty == 0 ? 0 : 2) : (ty == 0 ? 1 : 2);
                    const int x0 = b == 0 ? (tx == 0 ? 0 : 1) : (tx == 0 ? 0 : 2), x1 = b == 0 ? (tx == 0 ? 0 : 2) : (tx == 0 ? 1 : 2);
                    float acc = 0.f;
                    for (int y = y0; y <= y1; ++y)
                        for (int x = x0; x <= x1; ++x) acc += v[3 * y + x];
                    const int64_t pt = 4 * (2 * a + b) + 2 * ty + tx;
                    if (wf) wf[pt * cout * cin + idx] = (TF)acc;
                    if (wd) wd[(pt * cin + ci) * cout + co] = (TD)acc;
                }
}

bool subpixel_enabled() {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("GMK_SUBPIXEL");
        on = (e && e[0] == '0') ? 0 : 1;
    }
    return on != 0;
}

}  // namespace

// geometry of the low-resolution tiling; 0 if the sub-pixel kernel does not take the problem
static int subpixel_geometry(int B, int H, int W, int cin, int cout, int w_rows, int ntaps, int out_cstride, HaloGeometry* g) {
    if (cin != 128 || cout != 128) return 0;                       // one 128-channel source, one output block: the C = 128 nets
    if (!gmk_halo_geometry(B, H, W, cin, 0, w_rows, cout, out_cstride, 1, 0, 0, g)) return 0;
    const int64_t lim = 0xFFFF0000ll;
    const int64_t nbo = 4 * g->M * out_cstride * 2, nbw = (int64_t)ntaps * w_rows * cin * 2;
    if (nbo >= lim || nbw >= lim || 4 * g->nb0 >= lim || 4 * g->M >= 0x00FFFFFF) return 0;      // (4 M: pixel indices of the high-resolution tensor, 24 bits)
    g->nbo = nbo; g->nbw = nbw;
    return 1;
}

extern "C" int gmk_conv_subpixel_ok(int B, int H, int W, int cin, int cout, int dtype) {
    const int force = gmk_kernel_choice(0, "GMK_CONV_KERNEL");
    if ((force != 0 && force != 3) || !subpixel_enabled() || !gmk_is16(dtype)) return 0;
    HaloGeometry g;
    return subpixel_geometry(B, H, W, cin, cout, cout, 16, cout, &g);
}

// the same answer for the EXACT launch a caller is about to make (its own weight-row count and output stride enter the byte limits): what
// gmk_conv_igemm's dispatch asks, so that a shape the launch would refuse falls through to the other kernels instead of failing the call
int gmk_conv_subpixel_takes(int B, int H, int W, int cin, int cout, int w_rows, int ntaps, int out_cstride, int dtype) {
    const int force = gmk_kernel_choice(0, "GMK_CONV_KERNEL");
    if ((force != 0 && force != 3) || !subpixel_enabled() || !gmk_is16(dtype)) return 0;
    if (w_rows < cout || out_cstride < cout) return 0;
    HaloGeometry g;
    return subpixel_geometry(B, H, W, cin, cout, w_rows, ntaps, out_cstride, &g);
}

extern "C" int gmk_pack_upsample_weight(const float* w, void* w_sub, void* w_sub_dgrad, int cout, int cin, int dtype, int dgrad_dtype, void* stream) {
    GMK_REQUIRE(w && (w_sub || w_sub_dgrad) && cout > 0 && cin > 0, "gmk_pack_upsample_weight: bad arguments");
    GMK_REQUIRE(gmk_is16(dtype) && gmk_is16(dgrad_dtype), "gmk_pack_upsample_weight: 16-bit packs only (dtypes %d, %d)", dtype, dgrad_dtype);
    const int blocks = (cout * cin + 255) / 256;
    hipStream_t st = gmk_stream(stream);
    if (dtype == GMK_F16 && dgrad_dtype == GMK_BF16) pack_upsample_kernel<f16_t, bf16_t><<<blocks, 256, 0, st>>>(w, (f16_t*)w_sub, (bf16_t*)w_sub_dgrad, cout, cin);
    else if (dtype == GMK_BF16 && dgrad_dtype == GMK_BF16) pack_upsample_kernel<bf16_t, bf16_t><<<blocks, 256, 0, st>>>(w, (bf16_t*)w_sub, (bf16_t*)w_sub_dgrad, cout, cin);
    else if (dtype == GMK_F16 && dgrad_dtype == GMK_F16) pack_upsample_kernel<f16_t, f16_t><<<blocks, 256, 0, st>>>(w, (f16_t*)w_sub, (f16_t*)w_sub_dgrad, cout, cin);
    else GMK_REQUIRE(false, "gmk_pack_upsample_weight: forward packs fp16 or bf16, data-gradient packs of the same type or bf16");
    return gmk_check_launch("gmk_pack_upsample_weight");
}

extern "C" int gmk_conv_subpixel(const void* src, int B, int H, int W, int cin, const void* w, int w_rows, int n0, int cout, int mode,
                                 const float* bias, const void* residual, void* out, int out_cstride, int dtype, void* stream) {
    GMK_REQUIRE(src && w && out, "gmk_conv_subpixel: null pointer");
    GMK_REQUIRE(gmk_is16(dtype), "gmk_conv_subpixel: 16-bit types only (dtype %d)", dtype);
    GMK_REQUIRE(mode == GMK_SUBPIXEL_UPSAMPLE || mode == GMK_SUBPIXEL_TRANSPOSED || mode == GMK_SUBPIXEL_UPSAMPLE_DGRAD, "gmk_conv_subpixel: bad mode %d", mode);
    GMK_REQUIRE(n0 >= 0 && n0 + cout <= w_rows && out_cstride >= cout, "gmk_conv_subpixel: bad output channels n0=%d cout=%d w_rows=%d", n0, cout, w_rows);
    GMK_REQUIRE(!residual || mode == GMK_SUBPIXEL_TRANSPOSED, "gmk_conv_subpixel: a residual is taken by GMK_SUBPIXEL_TRANSPOSED only (mode %d)", mode);
    const bool down = mode == GMK_SUBPIXEL_UPSAMPLE_DGRAD;
    const int ntaps = mode == GMK_SUBPIXEL_TRANSPOSED ? 9 : 16;
    HaloGeometry g;
    GMK_REQUIRE(subpixel_geometry(B, H, W, cin, cout, w_rows, ntaps, out_cstride, &g),
                "gmk_conv_subpixel: shape B=%d %dx%d cin=%d cout=%d is not eligible (ask gmk_conv_subpixel_ok first)", B, H, W, cin, cout);
    SubParams p = {};
    p.src = src; p.B = B; p.H = H; p.W = W; p.WE = W + 2; p.R = g.R; p.TP = g.TP; p.ntiles = (int)g.ntiles; p.M = (int)g.M;
    p.w = w; p.w_tap_stride_b = (unsigned)w_rows * (unsigned)cin * 2u; p.n0 = n0;
    p.bias = bias; p.residual = residual; p.out = out; p.out_cstride = out_cstride;
    // (H, W) is the LOW-resolution grid in every mode: the source of the LOW -> HIGH forms, the output of the HIGH -> LOW one
    p.nb0 = (unsigned)(down ? 4 * g.nb0 : g.nb0); p.nbw = (unsigned)g.nbw; p.nbo = (unsigned)(down ? g.nbo / 4 : g.nbo);
    p.inv_hp2 = 1.0f / (float)(H + 2); p.inv_h = 1.0f / (float)H; p.inv_we = 1.0f / (float)(W + 2); p.inv_w = 1.0f / (float)W;
    const int ncu = gmk_cu_limit();
    const dim3 grid((unsigned)(g.ntiles < ncu ? g.ntiles : ncu));
    hipStream_t st = gmk_stream(stream);
    if (mode == GMK_SUBPIXEL_UPSAMPLE) {
        if (dtype == GMK_F16) conv_subpixel_ws_kernel<f16_t, 0><<<grid, 512, 0, st>>>(p);
        else conv_subpixel_ws_kernel<bf16_t, 0><<<grid, 512, 0, st>>>(p);
    } else if (mode == GMK_SUBPIXEL_TRANSPOSED) {
        if (dtype == GMK_F16) conv_subpixel_ws_kernel<f16_t, 1><<<grid, 512, 0, st>>>(p);
        else conv_subpixel_ws_kernel<bf16_t, 1><<<grid, 512, 0, st>>>(p);
    } else {
        if (dtype == GMK_F16) conv_subpixel_ws_kernel<f16_t, 2><<<grid, 512, 0, st>>>(p);
        else conv_subpixel_ws_kernel<bf16_t, 2><<<grid, 512, 0, st>>>(p);
    }
    gmk_note_kernel(mode == GMK_SUBPIXEL_UPSAMPLE ? 8 : mode == GMK_SUBPIXEL_TRANSPOSED ? 9 : 10);
    return gmk_check_launch("gmk_conv_subpixel");
}
