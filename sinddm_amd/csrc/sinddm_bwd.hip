// Training path of the SinDDM hot path for gfx950: activation-saving forward, backward
// (data gradients through the same MFMA conv kernel with transposed weights, weight gradients on a
// dedicated MFMA kernel with pixels as the GEMM K dimension), L1 loss, fused Adam / EMA.
// Replaces autograd through SinDDMNet + torch.optim.Adam + EMA of the reference
// (SinDDM/models.py:578-611, trainer.py:134,194-214, models.py:18-31).
#include <utility>
#include "conv_mfma.h"
#include "conv_wino.h"
#include "conv_wino4.h"
#include "internal.h"
#include "wgrad_wino.h"
#include "wgrad_wh.h"

namespace sinddm {

// =====================================================================================
// weight-gradient MFMA kernel
//   gw[co][ci][tap] += sum_{b,y,x} dout[b][co][y][x] * in[b][ci][y+dy-1][x+dx-1]
// GEMM view: M = co (16-row tiles), N = ci (one 16-col tile per tap), K = pixels (4 per MFMA).
// A workgroup owns one (co-block of MT*16, ci-block of 16) slab of the gradient and walks a strided
// subset of the 2x32 pixel tiles; its 4 waves split the tile's 16 k-steps (K split), each holding all
// MT x TAPS accumulator tiles (180 registers at MT=5, TAPS=9) across the whole walk.  Both operand
// tiles are fetched with LDS DMA (global_load_lds: one wave instruction per dout channel row / per
// input (channel,row); out-of-image and missing-channel rows come from a page of zeros) into a
// double-buffered LDS image, ONE barrier per tile.  At the end the 4 partial slabs are combined in
// LDS and added to the global gradient with one atomic per element.
// LDS strides == 2 (mod 32) make both operand reads (lane -> channel*stride + pixel) conflict-free.
// =====================================================================================
constexpr int WG_THREADS = 256;
constexpr int WG_TH = 2, WG_TW = 32;
constexpr int WG_CI = 16;
constexpr int WG_PSO = WG_TH * WG_TW + 2;              // 66
constexpr int WG_IRS = WG_TW + 2;                      // 34
constexpr int WG_IHR = WG_TH + 2;                      // 4
constexpr int WG_PSI = ((WG_IHR * WG_IRS - 2 + 31) / 32) * 32 + 2;   // 162

struct WgradArgs {
    const float* dout;   // [B][Cout][H][W]
    const float* in;     // [B][Cin][H][W]
    const float* zero;   // >= 64 zero floats
    float* gw;           // [Cout][Cin][TAPS]   (+=)
    float* gb;           // [Cout] (+=) or nullptr
    int B, H, W, Cin, Cout;
    int coblks, ciblks, S;
    int tilesX, tilesY, ntiles;
    int abl;             // ablation bits (tuning only): 1 = skip the steady-state DMA, 2 = skip the epilogue
    int scr;             // 1: gw is a [co][tap][ci] staging slab (wgrad3x3_kernel only)
};

template <int MT, int TAPS>
__global__ __launch_bounds__(WG_THREADS, 2) void wgrad_mfma_kernel(WgradArgs p) {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BUF = MT * 16 * WG_PSO + WG_CI * WG_PSI;   // floats per buffer

    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int slot = id >> 3;
    const int pairs = p.coblks * p.ciblks;
    const int q = slot % pairs;
    const int s = (slot / pairs) * 8 + xcd;          // pixel-split index; same-split slabs share an XCD/L2
    const int cb = q / p.ciblks, cib = q - cb * p.ciblks;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, kq = lane >> 4;
    const int H = p.H, W = p.W, HW = H * W;
    const int tpi = p.tilesX * p.tilesY;
    // LDS-DMA of one pixel tile with buffer descriptors: the bounds check zero-fills everything outside the image or
    // the channel range (those lanes / rows carry an offset >= 2^30), so an instruction costs one v_add.
    constexpr unsigned OOB = 0x40000000u;
    auto issue = [&](int tile, float* buf) {
        const int b = tile / tpi;
        const int tr = tile - b * tpi;
        const int ty = tr / p.tilesX, tx = tr - ty * p.tilesX;
        const int y0 = ty * WG_TH, x0 = tx * WG_TW;
        // dout tile: one wave instruction per channel row of 64 pixels (lane -> (row, col))
        {
            const int r = lane >> 5, c = lane & 31;
            const unsigned loff = ((y0 + r < H) && (x0 + c < W)) ? (unsigned)((y0 + r) * W + x0 + c) * 4u : OOB;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.dout + (size_t)b * p.Cout * HW), 0, p.Cout * HW * 4, 0x00020000);
#pragma unroll 4
            for (int k = 0; k < MT * 4; ++k) {
                const int col = wave + 4 * k;
                const int co = cb * MT * 16 + col;
                const unsigned coff = co < p.Cout ? (unsigned)(co * HW) * 4u : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(buf + col * WG_PSO), 4, (int)(loff + coff), 0, 0, 0);
            }
        }
        // input tile with 1-pixel halo: one wave instruction per (channel, row), 34 active lanes
        if (lane < WG_IRS) {
            const int gx = x0 + lane - 1;
            const unsigned loff = (gx >= 0 && gx < W) ? (unsigned)gx * 4u : OOB;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.in + (size_t)b * p.Cin * HW), 0, p.Cin * HW * 4, 0x00020000);
            float* ibuf = buf + MT * 16 * WG_PSO;
#pragma unroll 4
            for (int k = 0; k < WG_CI * WG_IHR / 4; ++k) {
                const int qi = wave + 4 * k;
                const int cil = qi / WG_IHR, r = qi - cil * WG_IHR;
                const int ci = cib * WG_CI + cil;
                const int gy = y0 + r - 1;
                const unsigned roff = (ci < p.Cin && gy >= 0 && gy < H) ? (unsigned)(ci * HW + gy * W) * 4u : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(ibuf + cil * WG_PSI + r * WG_IRS), 4, (int)(loff + roff), 0, 0, 0);
            }
        }
    };

    f32x4 acc[MT][TAPS];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < TAPS; ++t) acc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) bsum[mt] = 0.f;

    // wave -> (tile row, half row): 4 k-steps of 4 consecutive pixels each
    const int wr = wave >> 1, wh = wave & 1;
    const int aBase = l16 * WG_PSO + wr * WG_TW + wh * 16 + kq;
    const int bBase = MT * 16 * WG_PSO + l16 * WG_PSI + wr * WG_IRS + wh * 16 + kq;

    int it = 0;
    if (s < p.ntiles) issue(s, smem);
    for (int tile = s; tile < p.ntiles; tile += p.S, ++it) {
        dma_barrier();          // DMA of `tile` landed (dma_barrier: vmcnt(0) + barrier); previous tile consumed
        float* cur = smem + (it & 1) * BUF;
        if (tile + p.S < p.ntiles) issue(tile + p.S, smem + ((it + 1) & 1) * BUF);
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            float a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                a[mt] = cur[aBase + mt * 16 * WG_PSO + 4 * j];
                bsum[mt] += a[mt];
            }
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                const int dy = (TAPS == 9) ? t / 3 : 1;
                const int dx = (TAPS == 9) ? t % 3 : 1;
                const float bv = cur[bBase + dy * WG_IRS + 4 * j + dx];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], bv, acc[mt][t], 0, 0, 0);
            }
        }
    }

    // ---- combine the 4 waves' partial slabs in LDS, then one global atomic per element ----
    __syncthreads();
    constexpr int SLAB = MT * 16 * WG_CI * TAPS;
    static_assert(SLAB + MT * 16 <= 2 * BUF, "reduction slab must fit in the staging buffers");
    float* sR = smem;            // SLAB floats
    float* sB = smem + SLAB;     // MT*16 floats
    for (int i = tid; i < SLAB + MT * 16; i += WG_THREADS) sR[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // C layout: col = lane&15 -> ci (N), row = (lane>>4)*4 + r -> co (M)
                const int col = mt * 16 + kq * 4 + r;
                atomicAdd(&sR[(col * WG_CI + l16) * TAPS + t], acc[mt][t][r]);
            }
        }
        float v = bsum[mt];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (kq == 0) atomicAdd(&sB[mt * 16 + l16], v);
    }
    __syncthreads();
    for (int i = tid; i < SLAB; i += WG_THREADS) {
        const int col = i / (WG_CI * TAPS);
        const int rem = i - col * (WG_CI * TAPS);
        const int cil = rem / TAPS, t = rem - cil * TAPS;
        const int co = cb * MT * 16 + col, ci = cib * WG_CI + cil;
        if (co < p.Cout && ci < p.Cin) atomicAdd(&p.gw[((size_t)co * p.Cin + ci) * TAPS + t], sR[i]);
    }
    if (p.gb && cib == 0) {
        for (int i = tid; i < MT * 16; i += WG_THREADS) {
            const int co = cb * MT * 16 + i;
            if (co < p.Cout) atomicAdd(&p.gb[co], sB[i]);
        }
    }
}

// -------------------------------------------------------------------------------------
// Direct-form 3x3 weight gradient for Cout % 80 == 0 (used for C_in < 16 and as SINDDM_WGRAD_WINO=0 fallback; the
// default is the Winograd-domain kernel of wgrad_wino.h): one 16-wave workgroup per CU owns an 80(co) x 80(ci) x 9 slab.
// Wave (tap-row, ci-tile) accumulates 5 co-tiles x 3 taps (60 registers) over every pixel of the tile, so no
// cross-wave reduction is needed and a 64-pixel tile (73 KB of LDS) feeds 3600 MFMAs: 2.4x fewer L2->LDS bytes
// per FLOP than the K-split kernel above, whose DMA latency cost 38% of its time (ablation in DESIGN.md).
// -------------------------------------------------------------------------------------
constexpr int W3_WAVES = 16;                              // 15 compute waves + 1 that only moves data (4 waves per SIMD)
constexpr int W3_THREADS = W3_WAVES * 64;
constexpr int W3_C = 80;                                  // co and ci per workgroup
constexpr int W3_BUF = W3_C * WG_PSO + W3_C * WG_PSI;     // floats per stage

__global__ __launch_bounds__(W3_THREADS) void wgrad3x3_kernel(WgradArgs p) {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int slot = id >> 3;
    const int pairs = p.coblks * p.ciblks;
    const int q = slot % pairs;
    const int s = (slot / pairs) * 8 + xcd;          // pixel-split index; same-split slabs share an XCD/L2
    const int cb = q / p.ciblks, cib = q - cb * p.ciblks;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int trw = wave / 5, cit = wave - trw * 5;  // tap row, ci tile of this wave
    const int l16 = lane & 15, kq = lane >> 4;
    const int H = p.H, W = p.W, HW = H * W;
    const int tpi = p.tilesX * p.tilesY;
    const int ci0 = cib * W3_C;
    const int nci = min(W3_C, p.Cin - ci0);

    // LDS-DMA of one pixel tile, cut into 5 uniform per-wave "groups" (1 dout row + 4 input rows, one wave
    // instruction each) so the main loop can spread them between its MFMA groups: 16 waves issuing 25 loads back
    // to back after the barrier stalled the matrix pipe for ~20% of a tile.  Buffer bounds checking zero-fills
    // halo / channel padding (offset >= 2^30).  16 waves x 5 = 80 dout rows, 16 x 20 = 320 (channel, row) pairs.
    constexpr unsigned OOB = 0x40000000u;
    constexpr int NG = 5;
    static_assert(W3_WAVES * NG == W3_C && W3_WAVES * NG * 4 == W3_C * WG_IHR, "slot split");
    struct TileAddr {
        __amdgpu_buffer_rsrc_t rd, ri;
        unsigned loff_d, loff_i;
        int gyW;        // (gy * W) of this wave's input row, or -1 if the row is outside the image
    };
    const int irow = wave & 3;          // input-tile row this wave loads (of 4), channels (wave>>2) + 4*k
    auto tile_addr = [&](int tile) {
        TileAddr ta;
        const int b = tile / tpi;
        const int tr = tile - b * tpi;
        const int ty = tr / p.tilesX, tx = tr - ty * p.tilesX;
        const int y0 = ty * WG_TH, x0 = tx * WG_TW;
        const int r = lane >> 5, c = lane & 31;
        ta.loff_d = ((y0 + r < H) && (x0 + c < W)) ? (unsigned)((y0 + r) * W + x0 + c) * 4u : OOB;
        const int gx = x0 + lane - 1;
        ta.loff_i = (gx >= 0 && gx < W) ? (unsigned)gx * 4u : OOB;
        ta.rd = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.dout + ((size_t)b * p.Cout + (size_t)cb * W3_C) * HW), 0, W3_C * HW * 4, 0x00020000);
        ta.ri = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.in + ((size_t)b * p.Cin + ci0) * HW), 0, nci * HW * 4, 0x00020000);
        const int gy = y0 + irow - 1;
        ta.gyW = (gy >= 0 && gy < H) ? gy * W : -1;
        return ta;
    };
    auto issue_group = [&](const TileAddr& ta, float* buf, int g) {
        {   // dout: one instruction per channel row of 64 pixels (lane -> (row, col))
            const int col = wave + W3_WAVES * g;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ta.rd, (lds_ptr)(buf + col * WG_PSO), 4,
                                                     (int)(ta.loff_d + (unsigned)(col * HW) * 4u), 0, 0, 0);
        }
        if (lane < WG_IRS) {   // input with 1-pixel halo: one instruction per (channel, row), 34 lanes
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cil = (wave >> 2) + 16 * g + 4 * i;
                const unsigned roff = (cil < nci && ta.gyW >= 0) ? (unsigned)(cil * HW + ta.gyW) * 4u : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    ta.ri, (lds_ptr)(buf + W3_C * WG_PSO + cil * WG_PSI + irow * WG_IRS), 4, (int)(ta.loff_i + roff), 0, 0, 0);
            }
        }
    };

    f32x4 acc[5][3];
#pragma unroll
    for (int mt = 0; mt < 5; ++mt)
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[5] = {0.f, 0.f, 0.f, 0.f, 0.f};

    const bool active = wave < 15 && ci0 + cit * 16 < p.Cin;   // wave-uniform; idle waves still move data
    const bool dobias = p.gb != nullptr && cib == 0 && wave == 15;   // the data-moving wave also sums dout for the bias
    const int aBase = l16 * WG_PSO + kq;
    const int bBase = W3_C * WG_PSO + (cit * 16 + l16) * WG_PSI + trw * WG_IRS + kq;

    int it = 0;
    if (s < p.ntiles) {
        const TileAddr ta = tile_addr(s);
#pragma unroll
        for (int g = 0; g < NG; ++g) issue_group(ta, smem, g);
    }
    for (int tile = s; tile < p.ntiles; tile += p.S, ++it) {
        dma_barrier();          // DMA of `tile` landed (dma_barrier: vmcnt(0) + barrier); previous tile consumed
        const float* cur = smem + (it & 1) * W3_BUF;
        float* nxt = smem + ((it + 1) & 1) * W3_BUF;
        const bool pf = tile + p.S < p.ntiles && !(p.abl & 1);
        TileAddr ta{};
        if (pf) ta = tile_addr(tile + p.S);
        if (!active) {
            if (pf) {
#pragma unroll
                for (int g = 0; g < NG; ++g) issue_group(ta, nxt, g);
            }
            if (dobias) {
#pragma unroll 4
                for (int j = 0; j < 16; ++j) {
#pragma unroll
                    for (int mt = 0; mt < 5; ++mt) bsum[mt] += cur[aBase + mt * 16 * WG_PSO + (j >> 3) * WG_TW + (j & 7) * 4];
                }
            }
            continue;
        }
        // 16 k-steps (4 consecutive pixels each): the LDS reads of k-step n+1 are in flight while the 15 MFMAs of
        // k-step n issue; DMA group g goes out at the start of k-step 2g
        float A0[5], B0[3], A1[5], B1[3];
        auto ld = [&](int j, float (&A)[5], float (&Bv)[3]) {
            const float* ca = cur + aBase + (j >> 3) * WG_TW + (j & 7) * 4;
            const float* cbv = cur + bBase + (j >> 3) * WG_IRS + (j & 7) * 4;
#pragma unroll
            for (int mt = 0; mt < 5; ++mt) A[mt] = ca[mt * 16 * WG_PSO];
#pragma unroll
            for (int t = 0; t < 3; ++t) Bv[t] = cbv[t];
        };
        auto mm = [&](const float (&A)[5], const float (&Bv)[3]) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int mt = 0; mt < 5; ++mt)
                    acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[mt], Bv[t], acc[mt][t], 0, 0, 0);
        };
        ld(0, A0, B0);
#pragma unroll 1
        for (int j2 = 0; j2 < 8; ++j2) {
            if (pf && j2 < NG) issue_group(ta, nxt, j2);
            ld(2 * j2 + 1, A1, B1);
            mm(A0, B0);
            if (j2 < 7) ld(2 * j2 + 2, A0, B0);
            mm(A1, B1);
        }
    }

    // ---- every wave owns its slab: one atomic per element; the pixel splits meet in L2 / memory ----
    // scr != 0: [co][tap][ci] staging slab, 16 lanes of a wave hit one 64-byte line (coalesced atomics);
    // scr == 0: straight into the [co][ci][tap] gradient (36-byte lane stride).
    if (active && !(p.abl & 2)) {
        const int ci = ci0 + cit * 16 + l16;
        if (ci < p.Cin) {
#pragma unroll
            for (int mt = 0; mt < 5; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // C layout: col = lane&15 -> ci (N), row = (lane>>4)*4 + r -> co (M)
                    const int co = cb * W3_C + mt * 16 + kq * 4 + r;
                    if (p.scr) {
                        float* g = p.gw + ((size_t)co * 9 + trw * 3) * p.Cin + ci;
#pragma unroll
                        for (int t = 0; t < 3; ++t) atomicAdd(g + (size_t)t * p.Cin, acc[mt][t][r]);
                    } else {
                        float* g = p.gw + ((size_t)co * p.Cin + ci) * 9 + trw * 3;
#pragma unroll
                        for (int t = 0; t < 3; ++t) atomicAdd(g + t, acc[mt][t][r]);
                    }
                }
        }
    }
    if (dobias) {
#pragma unroll
        for (int mt = 0; mt < 5; ++mt) {
            float v = bsum[mt];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (kq == 0) atomicAdd(&p.gb[cb * W3_C + mt * 16 + l16], v);
        }
    }
}

// -------------------------------------------------------------------------------------
// 1x1 weight gradient (res_conv, reference SinDDM/models.py:67) for Cout % 80 == 0:  gw[co][ci] += sum_px dout[co][px] in[ci][px]
// HBM-bound (4*(Cin + Cout) bytes per pixel for 2*Cin*Cout FLOP): what matters is reading dout / in once.  A 16-wave
// workgroup owns an 80(co) x 80(ci) slab (the K-split kernel above re-read dout once per 16 input channels);
// wave (k-group, ci-tile) of the 15 compute waves holds 5 co-tiles and takes every third k-step of a flat 64-pixel
// tile, wave 15 only moves data and sums dout for the bias.  Pixels are flattened (no halo): a tile is 64 consecutive
// pixels of one image, one whole-wave DMA instruction per channel.
// -------------------------------------------------------------------------------------
constexpr int W1_C = 80;
constexpr int W1_PIX = 64;
constexpr int W1_PS = W1_PIX + 2;                          // plane stride == 2 (mod 32)
constexpr int W1_BUF = 2 * W1_C * W1_PS;                   // dout planes + input planes per stage (10560 floats)

__global__ __launch_bounds__(1024) void wgrad1x1_kernel(WgradArgs p) {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int slot = id >> 3;
    const int pairs = p.coblks * p.ciblks;
    const int q = slot % pairs;
    const int s = (slot / pairs) * 8 + xcd;
    const int cb = q / p.ciblks, cib = q - cb * p.ciblks;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave / 5, cit = wave - kg * 5;      // k-group (0..2; 3 = the data-moving wave), ci tile
    const int l16 = lane & 15, kq = lane >> 4;
    const int HW = p.H * p.W;
    const int tpi = p.tilesX;                            // flat 64-pixel tiles per image
    const int ci0 = cib * W1_C;
    const int nci = min(W1_C, p.Cin - ci0);
    constexpr unsigned OOB = 0x40000000u;

    // pixel split s owns a contiguous tile range; (image, tile-in-image) advance incrementally
    const int per = (p.ntiles + p.S - 1) / p.S;
    const int t_begin = s * per, t_end = min(p.ntiles, t_begin + per);
    int nb = t_begin / tpi, nt_ = t_begin - nb * tpi;
    // this wave stages dout planes wave + 16 g and input planes wave + 16 g (g < 5): 10 instructions per tile
    auto issue = [&](float* buf) {
        const int px = nt_ * W1_PIX + lane;
        const unsigned loff = px < HW ? (unsigned)px * 4u : OOB;
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.dout + ((size_t)nb * p.Cout + (size_t)cb * W1_C) * HW), 0, W1_C * HW * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.in + ((size_t)nb * p.Cin + ci0) * HW), 0, nci * HW * 4, 0x00020000);
#pragma unroll
        for (int g = 0; g < 5; ++g) {
            const int ch = wave + 16 * g;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_ptr)(buf + ch * W1_PS), 4, (int)(loff + (unsigned)(ch * HW) * 4u), 0, 0, 0);
            const unsigned coff = ch < nci ? (unsigned)(ch * HW) * 4u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ri, (lds_ptr)(buf + (W1_C + ch) * W1_PS), 4, (int)(loff + coff), 0, 0, 0);
        }
        if (++nt_ == tpi) { nt_ = 0; ++nb; }
    };

    f32x4 acc[5];
#pragma unroll
    for (int mt = 0; mt < 5; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const bool active = wave < 15 && ci0 + cit * 16 < p.Cin;
    const bool dobias = p.gb != nullptr && cib == 0 && wave == 15;
    const int aBase = l16 * W1_PS + kq;
    const int bBase = (W1_C + cit * 16 + l16) * W1_PS + kq;

    int it = 0;
    if (t_begin < t_end) issue(smem);
    for (int tile = t_begin; tile < t_end; ++tile, ++it) {
        dma_barrier();
        const float* cur = smem + (it & 1) * W1_BUF;
        if (tile + 1 < t_end) issue(smem + ((it + 1) & 1) * W1_BUF);
        if (active) {
            // k-steps j = kg, kg + 3, ... of the 16 (4 consecutive pixels each)
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) {
                const int j = kg + 3 * jj;
                if (j < 16) {
                    const float bv = cur[bBase + 4 * j];
#pragma unroll
                    for (int mt = 0; mt < 5; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[aBase + mt * 16 * W1_PS + 4 * j], bv, acc[mt], 0, 0, 0);
                }
            }
        } else if (dobias) {
#pragma unroll 4
            for (int j = 0; j < 16; ++j)
#pragma unroll
                for (int mt = 0; mt < 5; ++mt) bsum[mt] += cur[aBase + mt * 16 * W1_PS + 4 * j];
        }
    }

    if (active) {
        const int ci = ci0 + cit * 16 + l16;
        if (ci < p.Cin) {
#pragma unroll
            for (int mt = 0; mt < 5; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // C layout: col = lane&15 -> ci (N), row = (lane>>4)*4 + r -> co (M); 16 lanes hit one 64-byte line
                    const int co = cb * W1_C + mt * 16 + kq * 4 + r;
                    atomicAdd(p.gw + (size_t)co * p.Cin + ci, acc[mt][r]);
                }
        }
    }
    if (dobias) {
#pragma unroll
        for (int mt = 0; mt < 5; ++mt) {
            float v = bsum[mt];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (kq == 0) atomicAdd(&p.gb[cb * W1_C + mt * 16 + l16], v);
        }
    }
}

// gw[co][ci][tap] += scr[co][tap][ci]   (unpacks the staging slab of wgrad3x3_kernel)
__global__ void wgrad_unstage_kernel(const float* __restrict__ scr, float* __restrict__ gw, int Cin, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;    // index into gw
    if (i >= n) return;
    const int t = i % 9;
    const int rest = i / 9;
    const int ci = rest % Cin, co = rest / Cin;
    gw[i] += scr[((size_t)co * 9 + t) * Cin + ci];
}

template <int MT, int TAPS>
static void wgrad_launch_t(const WgradArgs& a, unsigned grid, hipStream_t st) {
    constexpr size_t lds = (size_t)2 * (MT * 16 * WG_PSO + WG_CI * WG_PSI) * sizeof(float);
    hipLaunchKernelGGL((wgrad_mfma_kernel<MT, TAPS>), dim3(grid), dim3(WG_THREADS), lds, st, a);
}

// amax_d / amax_i: the per-sample running maxima of dout / in when their producers maintain them (training on the binary16
// kernels): the Winograd-domain weight gradient then runs on the binary16 matrix pipe too (wgrad_wh.h)
static int wgrad_launch(const float* zero, const float* dout, const float* in, float* gw, float* gb, int B, int H, int W,
                        int Cin, int Cout, int taps, hipStream_t st, float* scr = nullptr, const float* amax_d = nullptr,
                        const float* amax_i = nullptr) {
    WgradArgs a{};
    a.zero = zero;
    a.dout = dout; a.in = in; a.gw = gw; a.gb = gb;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    constexpr int abl = SINDDM_WGRAD_ABL;
    constexpr int w3 = SINDDM_WGRAD_W3;
    a.abl = abl;
    a.tilesX = (W + WG_TW - 1) / WG_TW;
    a.tilesY = (H + WG_TH - 1) / WG_TH;
    a.ntiles = B * a.tilesX * a.tilesY;
    constexpr int ww = SINDDM_WGRAD_WINO;
    WwArgs w{};
    int nwg = 0;
    if (taps == 9 && Cout % WW_CO == 0 && Cin >= 16 && ww && scr) {
        w.dout = dout; w.in = in; w.gw = scr; w.gb = gb;
        w.B = B; w.H = H; w.W = W; w.Cin = Cin; w.Cout = Cout;
        w.coblks = Cout / WW_CO;
        w.ciblks = (Cin + WW_CI - 1) / WW_CI;
        w.tilesX = (W + WW_TW - 1) / WW_TW;
        w.tilesY = (H + WW_TH - 1) / WW_TH;
        w.ntiles = B * w.tilesX * w.tilesY;
        nwg = ww_build_map(w, device_cu_count());      // 0: more slabs than the launch table holds -> the direct kernels below
    }
    if (nwg > 0) {
        // Winograd-domain weight gradient (2.25x fewer MFMAs); always through the [co][tap][ci] staging slab
        if ((size_t)WW_CO * H * W * 4 >= 0x40000000ull) return SINDDM_E_BADSHAPE;
        const int n = Cout * Cin * 9;
        hipError_t e = hipMemsetAsync(scr, 0, (size_t)n * sizeof(float), st);
        if (e != hipSuccess) return (int)e;
        // rows of 16-byte groups (W % 4 == 0): the variant with 16-byte DMA and ds_read_b64 operands
        const bool wide = SINDDM_WGRAD_WIDE && W % 4 == 0;
        const bool h16 = SINDDM_WGRAD_WH && wide && amax_d && amax_i && wh_enabled();
        w.amax_d = amax_d; w.amax_i = amax_i;
        const size_t lds = (size_t)WW_STAGES * (wide ? WX_BUF : WW_BUF) * sizeof(float);
        // (more than 64 KB of dynamic LDS needs the per-function opt-in: once per kernel and process, its result checked)
        static const hipError_t attr_wide = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_wino_wide_kernel),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)WW_STAGES * WX_BUF * sizeof(float)));
        static const hipError_t attr_dword = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_wino_kernel),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)WW_STAGES * WW_BUF * sizeof(float)));
        static const hipError_t attr_h16 = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_wh_kernel),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)WW_STAGES * WX_BUF * sizeof(float)));
        if ((wide ? attr_wide : attr_dword) != hipSuccess) return (int)(wide ? attr_wide : attr_dword);
        if (h16 && attr_h16 != hipSuccess) return (int)attr_h16;
        ConvProfiler& prof = conv_profiler();
        const bool rec = prof.on && prof.used < ConvProfiler::MAXREC;
        if (rec) {
            while (prof.created <= prof.used) {
                (void)hipEventCreate(&prof.ev[2 * prof.created]);
                (void)hipEventCreate(&prof.ev[2 * prof.created + 1]);
                ++prof.created;
            }
            (void)hipEventRecord(prof.ev[2 * prof.used], st);
        }
        if (h16) hipLaunchKernelGGL(wgrad_wh_kernel, dim3((unsigned)nwg), dim3(WW_THREADS), lds, st, w);
        else if (wide) hipLaunchKernelGGL(wgrad_wino_wide_kernel, dim3((unsigned)nwg), dim3(WW_THREADS), lds, st, w);
        else hipLaunchKernelGGL(wgrad_wino_kernel, dim3((unsigned)nwg), dim3(WW_THREADS), lds, st, w);
        SINDDM_LAUNCH_CHECK();
        if (rec) {
            (void)hipEventRecord(prof.ev[2 * prof.used + 1], st);
            const double fl = 2.0 * B * H * W * (double)Cout * Cin * 9.0;
            // kind 4 = Winograd-domain weight gradient, F(2x2): 16/36 executed; generation 8 = binary16 pieces, four terms
            if (h16) prof.note(4, fl, fl * 16.0 / 36.0 * 4.0, 8);
            else prof.note(4, fl, fl * 16.0 / 36.0);
        }
        hipLaunchKernelGGL(wgrad_unstage_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, scr, gw, Cin, n);
        SINDDM_LAUNCH_CHECK();
        return 0;
    }
    constexpr int w1 = SINDDM_WGRAD_W1;
    if (taps == 1 && Cout % W1_C == 0 && w1) {
        if ((size_t)W1_C * H * W * 4 >= 0x40000000ull) return SINDDM_E_BADSHAPE;
        a.coblks = Cout / W1_C;
        a.ciblks = (Cin + W1_C - 1) / W1_C;
        a.tilesX = (H * W + W1_PIX - 1) / W1_PIX;      // flat tiles per image
        a.tilesY = 1;
        a.ntiles = B * a.tilesX;
        const int pairs = a.coblks * a.ciblks;
        int S = (device_cu_count() / pairs) / 8 * 8;
        if (S < 8) S = 8;
        const int cap = (a.ntiles + 7) / 8 * 8;
        if (S > cap) S = cap;
        a.S = S;
        constexpr size_t lds = (size_t)2 * W1_BUF * sizeof(float);
        hipLaunchKernelGGL(wgrad1x1_kernel, dim3((unsigned)(pairs * S)), dim3(1024), lds, st, a);
        SINDDM_LAUNCH_CHECK();
        return 0;
    }
    if (taps == 9 && Cout % W3_C == 0 && w3) {
        // one workgroup per CU; the largest per-sample channel slab must stay below the buffer OOB marker
        if ((size_t)W3_C * H * W * 4 >= 0x40000000ull) return SINDDM_E_BADSHAPE;
        a.coblks = Cout / W3_C;
        a.ciblks = (Cin + W3_C - 1) / W3_C;
        const int pairs = a.coblks * a.ciblks;
        int S = (device_cu_count() / pairs) / 8 * 8;
        if (S < 8) S = 8;
        const int cap = (a.ntiles + 7) / 8 * 8;
        if (S > cap) S = cap;
        a.S = S;
        constexpr size_t lds = (size_t)2 * W3_BUF * sizeof(float);
        constexpr int stage = SINDDM_WGRAD_STAGE;
        const int n = Cout * Cin * 9;
        if (scr && stage) {
            hipError_t e = hipMemsetAsync(scr, 0, (size_t)n * sizeof(float), st);
            if (e != hipSuccess) return (int)e;
            a.gw = scr;
            a.scr = 1;
        }
        hipLaunchKernelGGL(wgrad3x3_kernel, dim3((unsigned)(pairs * S)), dim3(W3_THREADS), lds, st, a);
        SINDDM_LAUNCH_CHECK();
        if (a.scr) {
            hipLaunchKernelGGL(wgrad_unstage_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, scr, gw, Cin, n);
            SINDDM_LAUNCH_CHECK();
        }
        return 0;
    }
    const int mt = mt_for(Cout);
    a.coblks = (Cout + mt * 16 - 1) / (mt * 16);
    a.ciblks = (Cin + WG_CI - 1) / WG_CI;
    a.tilesX = (W + WG_TW - 1) / WG_TW;
    a.tilesY = (H + WG_TH - 1) / WG_TH;
    a.ntiles = B * a.tilesX * a.tilesY;
    const int pairs = a.coblks * a.ciblks;
    int S = (1024 / pairs) / 8 * 8;
    if (S < 8) S = 8;
    const int cap = (a.ntiles + 7) / 8 * 8;
    if (S > cap) S = cap;
    a.S = S;
    const unsigned grid = (unsigned)(pairs * S);
    if (taps == 9) {
        if (mt == 5) wgrad_launch_t<5, 9>(a, grid, st);
        else if (mt == 2) wgrad_launch_t<2, 9>(a, grid, st);
        else wgrad_launch_t<1, 9>(a, grid, st);
    } else {
        if (mt == 5) wgrad_launch_t<5, 1>(a, grid, st);
        else if (mt == 2) wgrad_launch_t<2, 1>(a, grid, st);
        else wgrad_launch_t<1, 1>(a, grid, st);
    }
    SINDDM_LAUNCH_CHECK();
    return 0;
}

// =====================================================================================
// depthwise 5x5 weight / bias / condition gradients (HBM-bound: reads dh and x once)
//   gw[c][tap] += sum_{b,p} dh[b][c][p] * x[b][c][p+tap-2];  gb[c] += sum dh;  dcond[b][c] = sum_p dh
// =====================================================================================
constexpr int DWB_TH = 16, DWB_TW = 64, DWB_RS = DWB_TW + 4, DWB_HR = DWB_TH + 4;

__global__ __launch_bounds__(256) void dwconv5_wgrad_kernel(const float* __restrict__ dh, const float* __restrict__ x,
                                                            float* __restrict__ gw, float* __restrict__ gb,
                                                            float* __restrict__ dcond, int cond_stride, int C, int H,
                                                            int W) {
    __shared__ float tile[DWB_HR * DWB_RS];
    __shared__ float red[4][26];
    const int c = blockIdx.x, b = blockIdx.y;
    const size_t plane = ((size_t)b * C + c) * H * W;
    const float* xs = x + plane;
    const float* ds = dh + plane;
    const int tilesX = (W + DWB_TW - 1) / DWB_TW, tilesY = (H + DWB_TH - 1) / DWB_TH;
    const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;    // wave wv: tile rows 4wv..4wv+3, lane: column
    float acc[26];
#pragma unroll
    for (int k = 0; k < 26; ++k) acc[k] = 0.f;
    for (int t = 0; t < tilesX * tilesY; ++t) {
        const int ty = t / tilesX, tx = t - ty * tilesX;
        const int y0 = ty * DWB_TH, x0 = tx * DWB_TW;
        // issue every global load of this tile (halo tile of x + this thread's 4 dh values) before the first wait
        constexpr int DWB_LD = (DWB_HR * DWB_RS + 255) / 256;
        float stg[DWB_LD];
#pragma unroll
        for (int k = 0; k < DWB_LD; ++k) {
            const int i = threadIdx.x + k * 256;
            const int rr = i / DWB_RS, cc = i - rr * DWB_RS;
            const int gy = y0 + rr - 2, gx = x0 + cc - 2;
            const bool ok = i < DWB_HR * DWB_RS && gy >= 0 && gy < H && gx >= 0 && gx < W;
            stg[k] = ok ? xs[(size_t)gy * W + gx] : 0.0f;
        }
        const int gx = x0 + ln;
        float d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gy = y0 + wv * 4 + i;
            d[i] = (gy < H && gx < W) ? ds[(size_t)gy * W + gx] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < DWB_LD; ++k) {
            const int i = threadIdx.x + k * 256;
            if (i < DWB_HR * DWB_RS) tile[i] = stg[k];
        }
        __syncthreads();
        acc[25] += (d[0] + d[1]) + (d[2] + d[3]);
#pragma unroll
        for (int dy = 0; dy < 8; ++dy) {
            float v[5];
#pragma unroll
            for (int dx = 0; dx < 5; ++dx) v[dx] = tile[(wv * 4 + dy) * DWB_RS + ln + dx];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ky = dy - i;
                if (ky >= 0 && ky < 5) {
#pragma unroll
                    for (int dx = 0; dx < 5; ++dx) acc[ky * 5 + dx] = fmaf(d[i], v[dx], acc[ky * 5 + dx]);
                }
            }
        }
    }
    // block reduction of the 26 partial sums
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 26; ++k) {
        float v = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 26) {
        const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (threadIdx.x < 25) atomicAdd(&gw[c * 25 + threadIdx.x], v);
        else {
            atomicAdd(&gb[c], v);
            dcond[(size_t)b * cond_stride + c] = v;
        }
    }
}

// Register-window variant for rows that are a multiple of 4 pixels and at least 192 wide (round 5; the LDS-tile kernel above
// ran at 0.55 of the HBM rate: scalar loads, two barriers per tile).  Same block per (channel, sample) plane, same reduction;
// wave w walks DOWN its quarter of the rows, lane = 4 consecutive columns: per row one 16-byte load of dh and three aligned
// 16-byte loads of x (columns -4 .. +7 around the lane's four; the neighbours' overlap is served by L1), a window of six x rows
// in registers (the newest is in flight while the five above it are used), 100 FMAs per row into the lane's 25 tap sums.
// No LDS, no barrier until the reduction.
template <class F, int... I>
__device__ __forceinline__ void dwg_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
__global__ __launch_bounds__(256) void dwconv5_wgrad_rows_kernel(const float* __restrict__ dh, const float* __restrict__ x,
                                                                 float* __restrict__ gw, float* __restrict__ gb,
                                                                 float* __restrict__ dcond, int cond_stride, int C, int H,
                                                                 int W) {
    __shared__ float red[4][26];
    const int c = blockIdx.x, b = blockIdx.y;
    const size_t plane = ((size_t)b * C + c) * H * W;
    const float* xs = x + plane;
    const float* ds = dh + plane;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[26];
#pragma unroll
    for (int k = 0; k < 26; ++k) acc[k] = 0.f;
    const int rows_per = (H + 3) / 4;
    const int r0 = wave * rows_per, r1 = min(H, r0 + rows_per);
    const f32x4 zero4{0.f, 0.f, 0.f, 0.f};
    for (int xb = 0; xb < W; xb += 256) {
        const int gx = xb + 4 * lane;
        const bool act = gx < W;
        const bool okl = act && gx >= 4, okr = act && gx + 4 < W;
        f32x4 xw[6][3];
        auto load_row = [&](int y, f32x4 (&r)[3]) {
            const bool oky = y >= 0 && y < H;
            const float* q = xs + (size_t)y * W + gx;
            r[0] = (oky && okl) ? *reinterpret_cast<const f32x4*>(q - 4) : zero4;
            r[1] = (oky && act) ? *reinterpret_cast<const f32x4*>(q) : zero4;
            r[2] = (oky && okr) ? *reinterpret_cast<const f32x4*>(q + 4) : zero4;
        };
        // rows r0 - 2 .. r0 + 2 -> slots 0 .. 4; row y + 3 is requested while row y is processed
#pragma unroll
        for (int k = 0; k < 5; ++k) load_row(r0 - 2 + k, xw[k]);
        for (int y0 = r0; y0 < r1; y0 += 6) {
            dwg_static_for_impl([&](auto U) {
                constexpr int u = decltype(U)::value;
                const int y = y0 + u;
                if (y < r1) {                                   // (wave-uniform)
                    load_row(y + 3, xw[(u + 5) % 6]);
                    const f32x4 d = act ? *reinterpret_cast<const f32x4*>(ds + (size_t)y * W + gx) : zero4;
                    acc[25] += (d.x + d.y) + (d.z + d.w);
#pragma unroll
                    for (int ky = 0; ky < 5; ++ky) {
                        const f32x4(&r)[3] = xw[(u + ky) % 6];  // x row y + ky - 2
                        const float w12[12] = {r[0].x, r[0].y, r[0].z, r[0].w, r[1].x, r[1].y, r[1].z, r[1].w, r[2].x, r[2].y, r[2].z, r[2].w};
#pragma unroll
                        for (int kx = 0; kx < 5; ++kx) {
                            // tap (ky, kx): x[y + ky - 2][gx + j + kx - 2] = window column j + kx + 2
                            float a = acc[ky * 5 + kx];
                            a = fmaf(d.x, w12[kx + 2], a);
                            a = fmaf(d.y, w12[kx + 3], a);
                            a = fmaf(d.z, w12[kx + 4], a);
                            a = fmaf(d.w, w12[kx + 5], a);
                            acc[ky * 5 + kx] = a;
                        }
                    }
                }
            }, std::make_integer_sequence<int, 6>{});
        }
    }
#pragma unroll
    for (int k = 0; k < 26; ++k) {
        float v = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 26) {
        const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (threadIdx.x < 25) atomicAdd(&gw[c * 25 + threadIdx.x], v);
        else {
            atomicAdd(&gb[c], v);
            dcond[(size_t)b * cond_stride + c] = v;
        }
    }
}

// =====================================================================================
// conditioning-path backward (tiny; ONE workgroup, phases separated by __syncthreads)
// =====================================================================================
struct CondBwdArgs {
    const float* params;
    float* grads;
    const float* dcond;   // [B][cs]
    const float* emb;     // [B][64]
    const float* hpre;    // [B][128]
    const float* cvec;    // [B][32]
    const float* mvec;    // [B][4][32]
    float* dm;            // [B][4][32]  scratch
    float* dcv;           // [B][32]     scratch
    float* dh1;           // [B][128]    scratch
    int B, cs;
    long long tm0_w, tm0_b, tm2_w, tm2_b;
    long long mlp_w[4], mlp_b[4], tr_w[4], tr_b[4];
    int cin[4], coff[4];
};

// Four dependent phases; each is launched on its own with a grid-stride loop (one 512-thread block running all four
// took 0.57 ms of pure latency per training step).
template <int PHASE>
__global__ __launch_bounds__(256) void cond_backward_kernel(CondBwdArgs a) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    const float* P = a.params;
    float* G = a.grads;
    const int B = a.B;
    if (PHASE == 1) {
    // P1: dm = dcond . Wtr ; grads of time_reshape
    for (int it = tid; it < B * 128; it += nt) {
        const int b = it >> 7, l = (it >> 5) & 3, k = it & 31;
        const float* dc = a.dcond + (size_t)b * a.cs + a.coff[l];
        const float* w = P + a.tr_w[l] + k;
        float s = 0.f;
        for (int c = 0; c < a.cin[l]; ++c) s = fmaf(dc[c], w[(size_t)c * 32], s);
        a.dm[it] = s;
    }
    for (int l = 0; l < 4; ++l) {
        for (int it = tid; it < a.cin[l] * 32; it += nt) {
            const int c = it >> 5, k = it & 31;
            float s = 0.f;
            for (int b = 0; b < B; ++b)
                s = fmaf(a.dcond[(size_t)b * a.cs + a.coff[l] + c], a.mvec[((size_t)b * 4 + l) * 32 + k], s);
            G[a.tr_w[l] + it] += s;
        }
        for (int c = tid; c < a.cin[l]; c += nt) {
            float s = 0.f;
            for (int b = 0; b < B; ++b) s += a.dcond[(size_t)b * a.cs + a.coff[l] + c];
            G[a.tr_b[l] + c] += s;
        }
    }
    }
    if (PHASE == 2) {
    // P2: through the per-block Linear(32,32) and GELU(cond)
    for (int it = tid; it < B * 32; it += nt) {
        const int b = it >> 5, k = it & 31;
        float s = 0.f;
        for (int l = 0; l < 4; ++l) {
            const float* w = P + a.mlp_w[l] + k;
            const float* d = a.dm + ((size_t)b * 4 + l) * 32;
            for (int j = 0; j < 32; ++j) s = fmaf(d[j], w[j * 32], s);
        }
        a.dcv[it] = s * gelu_erf_grad(a.cvec[it]);
    }
    for (int it = tid; it < 4 * 1024; it += nt) {
        const int l = it >> 10, j = (it >> 5) & 31, k = it & 31;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s = fmaf(a.dm[((size_t)b * 4 + l) * 32 + j], gelu_erf(a.cvec[b * 32 + k]), s);
        G[a.mlp_w[l] + j * 32 + k] += s;
    }
    for (int it = tid; it < 128; it += nt) {
        const int l = it >> 5, j = it & 31;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += a.dm[((size_t)b * 4 + l) * 32 + j];
        G[a.mlp_b[l] + j] += s;
    }
    }
    if (PHASE == 3) {
    // P3: time_mlp.2 (32 x 128) and the GELU before it
    for (int it = tid; it < B * 128; it += nt) {
        const int b = it >> 7, j = it & 127;
        float s = 0.f;
        for (int k = 0; k < 32; ++k) s = fmaf(a.dcv[b * 32 + k], P[a.tm2_w + k * 128 + j], s);
        a.dh1[it] = s * gelu_erf_grad(a.hpre[it]);
    }
    for (int it = tid; it < 32 * 128; it += nt) {
        const int k = it >> 7, j = it & 127;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s = fmaf(a.dcv[b * 32 + k], gelu_erf(a.hpre[b * 128 + j]), s);
        G[a.tm2_w + it] += s;
    }
    for (int k = tid; k < 32; k += nt) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += a.dcv[b * 32 + k];
        G[a.tm2_b + k] += s;
    }
    }
    if (PHASE == 4) {
    // P4: time_mlp.0 (128 x 64)
    for (int it = tid; it < 128 * 64; it += nt) {
        const int j = it >> 6, i = it & 63;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s = fmaf(a.dh1[b * 128 + j], a.emb[b * 64 + i], s);
        G[a.tm0_w + it] += s;
    }
    for (int j = tid; j < 128; j += nt) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += a.dh1[b * 128 + j];
        G[a.tm0_b + j] += s;
    }
    }
}

// =====================================================================================
// L1 loss forward+backward, fused Adam / EMA
// =====================================================================================
__global__ __launch_bounds__(256) void l1_loss_kernel(const float* __restrict__ noise, const float* __restrict__ eps,
                                                      float* __restrict__ loss, float* __restrict__ grad, long long n,
                                                      float inv_n, float gscale) {
    __shared__ float red[4];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float d = noise[i] - eps[i];
        s += fabsf(d);
        // d|noise-eps|/d eps = -sign(noise-eps), sign(0) = 0   (reference SinDDM/models.py:594)
        if (grad) grad[i] = (d > 0.f ? -1.f : (d < 0.f ? 1.f : 0.f)) * inv_n * gscale;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, ((red[0] + red[1]) + (red[2] + red[3])) * inv_n);
}

__global__ __launch_bounds__(256) void adam_ema_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v,
                                                       float* __restrict__ ema, float step_size, float b1, float b2,
                                                       float eps, float bc2_sqrt, float ema_decay, int mode,
                                                       long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float pv = p[i];
        if (mode & 1) {
            // torch.optim.Adam (no wd, no amsgrad): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
            // p -= step_size * m / (sqrt(v)/sqrt(bc2) + eps),  step_size = lr / bc1
            const float gv = g[i];
            const float mv = b1 * m[i] + (1.f - b1) * gv;
            const float vv = b2 * v[i] + (1.f - b2) * gv * gv;
            m[i] = mv;
            v[i] = vv;
            pv = pv - step_size * (mv / (sqrtf(vv) / bc2_sqrt + eps));
            p[i] = pv;
        }
        if (mode & 2) ema[i] = pv;                                             // copy phase (step < step_start_ema)
        else if (mode & 4) ema[i] = ema[i] * ema_decay + (1.f - ema_decay) * pv;   // models.py:28-31
    }
}

// =====================================================================================
// packed images for the data-gradient convolutions
// =====================================================================================
struct BwdPack {
    // per block: dgrad of conv2 (cout->cout), dgrad of conv1 (cout->cin), dgrad of res 1x1 (cout->cin)
    long long dg2[4], dg1[4], dres[4], dfin, zero;
    long long wdg2[4], wdg1[4];     // Winograd images of the 3x3 data-gradient convs (wdg1 = -1: stays direct)
    long long wdg2f[4], wdg1f[4];   // their F(2x4) images (conv_wino3.h), -1 = shape not supported
    long long qdg2[4], qdg1[4];     // their binary16 hi/lo F(2x4) images (conv_wh.h), -1 = shape not supported
    long long qs2[4], qs1[4];       // ... and the per-output-channel scales of those
    long long total;
    int mt2[4], mt1[4], cb2[4], cb1[4];
    int mtf, cbf;
};

static BwdPack make_bwd_pack(const NetPlan& P) {
    BwdPack k{};
    long long q = 0;
    for (int l = 0; l < 4; ++l) {
        const BlockPlan& b = P.blk[l];
        k.mt2[l] = mt_for(b.cout); k.cb2[l] = (b.cout + k.mt2[l] * 16 - 1) / (k.mt2[l] * 16);
        k.mt1[l] = mt_for(b.cin);  k.cb1[l] = (b.cin + k.mt1[l] * 16 - 1) / (k.mt1[l] * 16);
        const int nchK = (b.cout + KC - 1) / KC;      // K channels of every dgrad = forward cout
        k.dg2[l] = q; q += (long long)k.cb2[l] * nchK * 9 * KC * co_lds_for(k.mt2[l]);
        k.dg1[l] = q; q += (long long)k.cb1[l] * nchK * 9 * KC * co_lds_for(k.mt1[l]);
        if (b.res_w >= 0) { k.dres[l] = q; q += (long long)k.cb1[l] * nchK * KC * co_lds_for(k.mt1[l]); }
        else k.dres[l] = -1;
    }
    k.mtf = mt_for(P.half); k.cbf = (P.half + k.mtf * 16 - 1) / (k.mtf * 16);
    k.dfin = q; q += (long long)k.cbf * 1 * KC * co_lds_for(k.mtf);     // K = 3 channels -> one chunk
    for (int l = 0; l < 4; ++l) {
        const BlockPlan& b = P.blk[l];
        const int nchW = (b.cout + 15) / 16;
        k.wdg2[l] = q; q += (long long)k.cb2[l] * nchW * 16 * 4 * k.mt2[l] * 64;
        if (b.cin >= 8) { k.wdg1[l] = q; q += (long long)k.cb1[l] * nchW * 16 * 4 * k.mt1[l] * 64; }
        else k.wdg1[l] = -1;
    }
    // F(2x4) Winograd images of the data-gradient convs (conv_wino3.h): 80-channel row blocks, K = forward cout % 16 == 0
    for (int l = 0; l < 4; ++l) {
        const BlockPlan& b = P.blk[l];
        const int nchW = (b.cout + 15) / 16;
        const bool kok = b.cout % 16 == 0;
        if (kok && k.mt2[l] == 5 && b.cout % 80 == 0) { k.wdg2f[l] = q; q += (long long)k.cb2[l] * nchW * 32768; } else k.wdg2f[l] = -1;
        if (kok && k.mt1[l] == 5 && b.cin % 80 == 0) { k.wdg1f[l] = q; q += (long long)k.cb1[l] * nchW * 32768; } else k.wdg1f[l] = -1;
    }
    // binary16 hi/lo images of the same convs (conv_wh.h): K = forward cout in 16-channel chunks, 80-channel row blocks
    for (int l = 0; l < 4; ++l) {
        const BlockPlan& b = P.blk[l];
        q = (q + 63) / 64 * 64;
        if (wh_plan_ok(b.cout, b.cout)) {
            k.qdg2[l] = q; q += wh_plan_halfs(b.cout, b.cout) / 2;
            k.qs2[l] = q; q += (b.cout + 63) / 64 * 64;
        } else k.qdg2[l] = k.qs2[l] = -1;
        if (wh_plan_ok(b.cout, b.cin)) {
            k.qdg1[l] = q; q += wh_plan_halfs(b.cout, b.cin) / 2;
            k.qs1[l] = q; q += (b.cin + 63) / 64 * 64;
        } else k.qdg1[l] = k.qs1[l] = -1;
    }
    k.zero = q; q += 64;
    k.total = q;
    return k;
}

static int pack_backward(const NetPlan& P, const float* params, float* packed, hipStream_t st) {
    const BwdPack k = make_bwd_pack(P);
    PackArgs a{};
    int n = 0;
    long long total = 0;
    auto add = [&](long long dst, long long w, int fcin, int fcout, int taps, int mt, int coblks) {
        PackSeg s{};
        s.kind = 0; s.transpose = 1; s.w2 = -1;
        s.dst = dst; s.w = w; s.cin = fcin; s.cout = fcout; s.taps = taps; s.mt = mt; s.co_lds = co_lds_for(mt);
        s.nch = (fcout + KC - 1) / KC;
        s.count = (long long)coblks * s.nch * taps * KC * s.co_lds;
        a.seg[n++] = s;
        total += s.count;
    };
    for (int l = 0; l < 4; ++l) {
        const BlockPlan& b = P.blk[l];
        add(k.dg2[l], b.c2_w, b.cout, b.cout, 9, k.mt2[l], k.cb2[l]);
        add(k.dg1[l], b.c1_w, b.cin, b.cout, 9, k.mt1[l], k.cb1[l]);
        if (b.res_w >= 0) add(k.dres[l], b.res_w, b.cin, b.cout, 1, k.mt1[l], k.cb1[l]);
    }
    add(k.dfin, P.fin_w, P.half, CHANNELS, 1, k.mtf, k.cbf);
    auto addw = [&](long long dst, long long w, int fcin, int fcout, int mt, int coblks) {
        PackSeg s{};
        s.kind = 3; s.transpose = 1; s.w2 = -1; s.taps = 9;
        s.dst = dst; s.w = w; s.cin = fcin; s.cout = fcout; s.mt = mt;
        s.nch = (fcout + 15) / 16;
        s.count = (long long)coblks * s.nch * 16 * 4 * mt * 64;
        a.seg[n++] = s;
        total += s.count;
    };
    for (int l = 0; l < 4; ++l) {
        const BlockPlan& b = P.blk[l];
        addw(k.wdg2[l], b.c2_w, b.cout, b.cout, k.mt2[l], k.cb2[l]);
        if (k.wdg1[l] >= 0) addw(k.wdg1[l], b.c1_w, b.cin, b.cout, k.mt1[l], k.cb1[l]);
    }
    {
        PackSeg z{};
        z.kind = 2; z.dst = k.zero; z.count = 64;
        a.seg[n++] = z;
        total += z.count;
    }
    a.nseg = n;
    a.total = total;
    int rc = pack_launch(params, packed, a, st);
    if (rc) return rc;
    PackArgs f{};
    n = 0;
    total = 0;
    auto addf = [&](long long dst, long long w, int fcin, int fcout, int coblks) {
        PackSeg s{};
        s.kind = 4; s.transpose = 1; s.w2 = -1; s.taps = 9; s.mt = 5;
        s.dst = dst; s.w = w; s.cin = fcin; s.cout = fcout;
        s.nch = (fcout + 15) / 16;
        s.count = (long long)coblks * s.nch * 32768;
        f.seg[n++] = s;
        total += s.count;
    };
    for (int l = 0; l < 4; ++l) {
        const BlockPlan& b = P.blk[l];
        if (k.wdg2f[l] >= 0) addf(k.wdg2f[l], b.c2_w, b.cout, b.cout, k.cb2[l]);
        if (k.wdg1f[l] >= 0) addf(k.wdg1f[l], b.c1_w, b.cin, b.cout, k.cb1[l]);
    }
    if (n > 0) {
        f.nseg = n;
        f.total = total;
        rc = pack_launch(params, packed, f, st);
        if (rc) return rc;
    }
    for (int l = 0; l < 4; ++l) {
        const BlockPlan& b = P.blk[l];
        if (k.qdg2[l] >= 0) {
            rc = wh_pack(params + b.c2_w, packed + k.qs2[l], packed + k.qdg2[l], b.cout, b.cout, 1, st);
            if (rc) return rc;
        }
        if (k.qdg1[l] >= 0) {
            rc = wh_pack(params + b.c1_w, packed + k.qs1[l], packed + k.qdg1[l], b.cin, b.cout, 1, st);
            if (rc) return rc;
        }
    }
    return 0;
}

// =====================================================================================
// training workspace
// =====================================================================================
static size_t au(size_t v) { return (v + 255) / 256 * 256; }

static size_t carve_train(const NetPlan& P, int B, int H, int W, char* base, TrainBufs* tb) {
    size_t off = 0;
    auto take = [&](size_t nfloats) {
        float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
        off += au(nfloats * sizeof(float));
        return p;
    };
    const size_t HW = (size_t)H * W;
    TrainBufs t{};
    t.cond = take((size_t)B * P.cond_stride);
    t.emb = take((size_t)B * 64);
    t.hpre = take((size_t)B * 128);
    t.cvec = take((size_t)B * 32);
    t.mvec = take((size_t)B * 128);
    for (int l = 0; l < 4; ++l) {
        t.h[l] = take((size_t)B * P.blk[l].cin * HW);
        t.u[l] = take((size_t)B * P.blk[l].cout * HW);
        t.g[l] = take((size_t)B * P.blk[l].cout * HW);
        t.o[l] = take((size_t)B * P.blk[l].cout * HW);
    }
    for (int i = 0; i < 4; ++i) t.s[i] = take((size_t)B * P.dim * HW);
    t.dcond = take((size_t)B * P.cond_stride);
    t.small = take((size_t)B * (128 + 32 + 128));
    t.wscr = take((size_t)P.dim * P.dim * 9);
    t.amax = take((size_t)2 * B * AMAX_STRIDE);
    if (tb) *tb = t;
    return off;
}

static int conv1x1_or_3x3(const float* zero, const float* in3, int cin3, const float* w3, int nch3, const float* in1,
                          int cin1, const float* w1, int nch1, const float* aux, int act, float* out, int Cout, int mt,
                          int coblks, int B, int H, int W, hipStream_t st) {
    ConvArgs c{};
    c.zero = zero;
    c.in = in3; c.Cin = cin3; c.w3 = w3; c.nch3 = nch3;
    c.in2 = in1; c.Cin2 = cin1; c.w1 = w1; c.nch1 = nch1;
    c.aux = aux; c.act = act; c.out = out; c.Cout = Cout; c.coblks = coblks;
    c.B = B; c.H = H; c.W = W;
    if (nch3 == 0) return conv1x1_launch(c, mt, st);       // pure channel mixing: the HBM-bound 1x1 kernel
    return conv_launch(c, mt, st);
}

// `wf`: the F(2x4) image of the same conv (or nullptr): big launches take conv_wino3.h
static int conv3x3_wino(const float* zero, const float* in3, int cin3, const float* ww, const float* wf, const float* aux,
                        int act, float* out, int Cout, int mt, int coblks, int B, int H, int W, hipStream_t st) {
    ConvArgs c{};
    c.zero = zero;
    c.in = in3; c.Cin = cin3; c.w3 = ww; c.nch3 = (cin3 + 15) / 16; c.nch1 = 0;
    c.aux = aux; c.act = act; c.out = out; c.Cout = Cout; c.coblks = coblks;
    c.B = B; c.H = H; c.W = W;
    if (SINDDM_WINO_V3 && wf && mt == 5 &&
        (long long)B * ((W + 31) / 32) * ((H + 3) / 4) * coblks >= SINDDM_V3_MIN_ITEMS_PER_CU * wino2_cu_count()) {
        c.w3 = wf;
        if (SINDDM_WINO_V4 && conv_wino4_applies(B, H, W, coblks)) return conv_wino4_launch(c, st);
        return conv_wino3_launch(c, st);
    }
    return conv_wino_launch(c, mt, st);
}

// Backward of ONE SinDDMConvBlock (autograd of reference SinDDM/models.py:69-80): dO = gradient of the block output (scratch,
// overwritten), xin = the block input, saved tensors of block l from `tb`; weight / bias gradients are ADDED into `grads`,
// the per-sample condition gradient goes to tb.dcond, the input gradient to `dst` (skipped when null).  dU / dH: scratch.
static int block_backward(const NetPlan& P, const BwdPack& k, int l, const float* params, const float* packed_bwd,
                          const float* xin, float* dO, float* dU, float* dH, float* dst, float* grads, const TrainBufs& tb,
                          int B, int H, int W, hipStream_t st, float* amax_b = nullptr, bool dO_published = false) {
    const BlockPlan& b = P.blk[l];
    const float* zp = packed_bwd + k.zero;
    int rc;
    const int nchK = (b.cout + KC - 1) / KC;
    // the two 3x3 data-gradient convs on the binary16 hi/lo Winograd kernel (conv_wh.h) where its rule takes the launch;
    // `amax_b` = the backward half of tb.amax (zeroed by the caller): slot 2l = max |dO| per sample (maintained by the kernel
    // that wrote dO when `dO_published`), slot 2l + 1 = max |dU| (by the first conv's epilogue)
    const bool wh2 = amax_b && wino_enabled() && k.qdg2[l] >= 0 && wh_applies(P, B, H, W, b.cout, b.cout);
    const bool wh1 = wh2 && k.qdg1[l] >= 0 && wh_applies(P, B, H, W, b.cout, b.cin);
    if (wh2 && !dO_published) {
        rc = amax_tensor_launch(dO, amax_b + 2 * l, B, (long long)b.cout * H * W, st);
        if (rc) return rc;
    }
    // conv2 + residual projection weight grads
    // (the running maxima of g = conv2's forward input and of h = conv1's are the forward half of tb.amax, slots 2l + 1 / 2l)
    const bool wha = wh1 && wh_plan_ok(b.cin, b.cout) && wh_applies(P, B, H, W, b.cin, b.cout);
    rc = wgrad_launch(zp, dO, tb.g[l], grads + b.c2_w, grads + b.c2_b, B, H, W, b.cout, b.cout, 9, st, tb.wscr,
                      wh2 ? amax_b + 2 * l : nullptr, wh2 ? tb.amax + 2 * l + 1 : nullptr);
    if (rc) return rc;
    if (b.res_w >= 0) {
        rc = wgrad_launch(zp, dO, xin, grads + b.res_w, grads + b.res_b, B, H, W, b.cin, b.cout, 1, st);
        if (rc) return rc;
    }
    // dU = dgrad_conv2(dO) * GELU'(u)
    if (wh2) {
        ConvArgs c{};
        c.zero = zp; c.in = dO; c.Cin = b.cout; c.w3 = packed_bwd + k.qdg2[l]; c.wsinv = packed_bwd + k.qs2[l];
        c.amax_in = amax_b + 2 * l; c.amax_out = wh1 ? amax_b + 2 * l + 1 : nullptr;
        c.aux = tb.u[l]; c.act = 2; c.out = dU; c.Cout = b.cout; c.B = B; c.H = H; c.W = W;
        rc = wh_conv(c, st);
    } else if (wino_enabled() && b.cout % 4 == 0)        // (K of both data-gradient convs = cout; % 4: see conv_wino_launch)
        rc = conv3x3_wino(zp, dO, b.cout, packed_bwd + k.wdg2[l], k.wdg2f[l] >= 0 ? packed_bwd + k.wdg2f[l] : nullptr, tb.u[l], 2,
                          dU, b.cout, k.mt2[l], k.cb2[l], B, H, W, st);
    else
        rc = conv1x1_or_3x3(zp, dO, b.cout, packed_bwd + k.dg2[l], nchK, nullptr, 0, nullptr, 0, tb.u[l], 2, dU,
                            b.cout, k.mt2[l], k.cb2[l], B, H, W, st);
    if (rc) return rc;
    // conv1 weight grads, dH = dgrad_conv1(dU)
    rc = wgrad_launch(zp, dU, tb.h[l], grads + b.c1_w, grads + b.c1_b, B, H, W, b.cin, b.cout, 9, st, tb.wscr,
                      wha ? amax_b + 2 * l + 1 : nullptr, wha ? tb.amax + 2 * l : nullptr);
    if (rc) return rc;
    if (wh1) {
        ConvArgs c{};
        c.zero = zp; c.in = dU; c.Cin = b.cout; c.w3 = packed_bwd + k.qdg1[l]; c.wsinv = packed_bwd + k.qs1[l];
        c.amax_in = amax_b + 2 * l + 1;
        c.act = 0; c.out = dH; c.Cout = b.cin; c.B = B; c.H = H; c.W = W;
        rc = wh_conv(c, st);
    } else if (wino_enabled() && k.wdg1[l] >= 0 && b.cout % 4 == 0)
        rc = conv3x3_wino(zp, dU, b.cout, packed_bwd + k.wdg1[l], k.wdg1f[l] >= 0 ? packed_bwd + k.wdg1f[l] : nullptr, nullptr, 0,
                          dH, b.cin, k.mt1[l], k.cb1[l], B, H, W, st);
    else
        rc = conv1x1_or_3x3(zp, dU, b.cout, packed_bwd + k.dg1[l], nchK, nullptr, 0, nullptr, 0, nullptr, 0, dH,
                            b.cin, k.mt1[l], k.cb1[l], B, H, W, st);
    if (rc) return rc;
    // depthwise weight/bias grads and the per-sample condition grads
    if (SINDDM_DWG_ROWS && W % 4 == 0 && W >= 192)
        hipLaunchKernelGGL(dwconv5_wgrad_rows_kernel, dim3(b.cin, B), dim3(256), 0, st, dH, xin, grads + b.dw_w,
                           grads + b.dw_b, tb.dcond + b.cond_off, P.cond_stride, b.cin, H, W);
    else
        hipLaunchKernelGGL(dwconv5_wgrad_kernel, dim3(b.cin, B), dim3(256), 0, st, dH, xin, grads + b.dw_w,
                           grads + b.dw_b, tb.dcond + b.cond_off, P.cond_stride, b.cin, H, W);
    SINDDM_LAUNCH_CHECK();
    // data grad w.r.t. the block input: dw^T(dH) + residual path
    if (dst) {
        const float* radd = dO;                      // identity residual
        if (b.res_w >= 0) {
            rc = conv1x1_or_3x3(zp, nullptr, 0, nullptr, 0, dO, b.cout, packed_bwd + k.dres[l], nchK, nullptr, 0, dU,
                                b.cin, k.mt1[l], k.cb1[l], B, H, W, st);     // dU is free again
            if (rc) return rc;
            radd = dU;
        }
        // (dst is the next block's dO: its running max comes out of this launch when that block's convs will want it)
        float* pub = nullptr;
        if (amax_b && l > 0 && wino_enabled() && k.qdg2[l - 1] >= 0 && wh_applies(P, B, H, W, b.cin, b.cin)) pub = amax_b + 2 * (l - 1);
        rc = dwconv_launch(dH, params + b.dw_w, nullptr, nullptr, 0, radd, 1, dst, B, b.cin, H, W, st, 0, 0, pub);
        if (rc) return rc;
    }
    return 0;
}

static int net_backward_impl(const NetPlan& P, const float* params, const float* packed_bwd, const float* x,
                             const float* grad_out, float* grads, float* grad_x, int B, int H, int W,
                             const TrainBufs& tb, hipStream_t st) {
    const BwdPack k = make_bwd_pack(P);
    const float* zp = packed_bwd + k.zero;
    int rc;
    // ---- final 1x1 conv: weight/bias grads, then data grad into s[0] ----
    rc = wgrad_launch(zp, grad_out, tb.o[3], grads + P.fin_w, grads + P.fin_b, B, H, W, P.half, CHANNELS, 1, st);
    if (rc) return rc;
    int di = 0;   // index of the scratch buffer holding dOut of the current block
    float* amax_b = tb.amax + (size_t)B * AMAX_STRIDE;
    if (hipMemsetAsync(amax_b, 0, (size_t)B * AMAX_STRIDE * sizeof(float), st) != hipSuccess) return SINDDM_E_BADARG;
    rc = conv1x1_or_3x3(zp, nullptr, 0, nullptr, 0, grad_out, CHANNELS, packed_bwd + k.dfin, 1, nullptr, 0, tb.s[di],
                        P.half, k.mtf, k.cbf, B, H, W, st);
    if (rc) return rc;
    for (int l = 3; l >= 0; --l) {
        const float* xin = (l == 0) ? x : tb.o[l - 1];
        float* dO = tb.s[di];
        float* dU = tb.s[(di + 1) & 3];
        float* dH = tb.s[(di + 2) & 3];
        float* dX = tb.s[(di + 3) & 3];
        float* dst = (l == 0) ? grad_x : dX;
        rc = block_backward(P, k, l, params, packed_bwd, xin, dO, dU, dH, (l > 0 || grad_x) ? dst : nullptr, grads, tb, B, H, W, st,
                            amax_b, l < 3);
        if (rc) return rc;
        if (l > 0 || grad_x) di = (di + 3) & 3;
    }
    // ---- conditioning path ----
    CondBwdArgs ca{};
    ca.params = params; ca.grads = grads; ca.dcond = tb.dcond; ca.emb = tb.emb; ca.hpre = tb.hpre; ca.cvec = tb.cvec;
    ca.mvec = tb.mvec; ca.dm = tb.small; ca.dcv = tb.small + (size_t)B * 128; ca.dh1 = tb.small + (size_t)B * 160;
    ca.B = B; ca.cs = P.cond_stride;
    ca.tm0_w = P.tm0_w; ca.tm0_b = P.tm0_b; ca.tm2_w = P.tm2_w; ca.tm2_b = P.tm2_b;
    for (int l = 0; l < 4; ++l) {
        ca.mlp_w[l] = P.blk[l].mlp_w; ca.mlp_b[l] = P.blk[l].mlp_b; ca.tr_w[l] = P.blk[l].tr_w; ca.tr_b[l] = P.blk[l].tr_b;
        ca.cin[l] = P.blk[l].cin; ca.coff[l] = P.blk[l].cond_off;
    }
    hipLaunchKernelGGL(cond_backward_kernel<1>, dim3(64), dim3(256), 0, st, ca);
    hipLaunchKernelGGL(cond_backward_kernel<2>, dim3(16), dim3(256), 0, st, ca);
    hipLaunchKernelGGL(cond_backward_kernel<3>, dim3(16), dim3(256), 0, st, ca);
    hipLaunchKernelGGL(cond_backward_kernel<4>, dim3(32), dim3(256), 0, st, ca);
    SINDDM_LAUNCH_CHECK();
    return 0;
}

}  // namespace sinddm

using namespace sinddm;

extern "C" {

size_t sinddm_train_workspace_bytes(int dim, int B, int H, int W) {
    NetPlan p = make_plan(dim);
    if (!p.ok || B <= 0 || H <= 0 || W <= 0) return 0;
    return carve_train(p, B, H, W, nullptr, nullptr);
}

int64_t sinddm_packed_bwd_count(int dim) {
    NetPlan p = make_plan(dim);
    if (!p.ok) return -1;
    return make_bwd_pack(p).total;
}

int sinddm_pack_weights_bwd(const float* params, float* packed_bwd, int dim, void* stream) {
    if (!params || !packed_bwd) return SINDDM_E_BADARG;
    NetPlan p = make_plan(dim);
    if (!p.ok) return SINDDM_E_BADSHAPE;
    return pack_backward(p, params, packed_bwd, static_cast<hipStream_t>(stream));
}

int sinddm_net_forward_train(const float* params, const float* packed, const float* x, const int64_t* t_dev,
                             int t_host, float scale, float* out, int dim, int B, int H, int W, void* ws,
                             size_t ws_bytes, void* stream) {
    if (!params || !packed || !x || !out || !ws || B <= 0 || H <= 0 || W <= 0) return SINDDM_E_BADARG;
    NetPlan p = make_plan(dim);
    if (!p.ok) return SINDDM_E_BADSHAPE;
    TrainBufs tb;
    if (carve_train(p, B, H, W, static_cast<char*>(ws), &tb) > ws_bytes) return SINDDM_E_WORKSPACE;
    return net_forward_impl(p, params, packed, x, t_dev, t_host, scale, out, B, H, W, nullptr, 0,
                            static_cast<hipStream_t>(stream), &tb);
}

int sinddm_net_backward(const float* params, const float* packed, const float* packed_bwd, const float* x,
                        const float* grad_out, float* grad_params, float* grad_x, int dim, int B, int H, int W,
                        void* ws, size_t ws_bytes, void* stream) {
    (void)packed;
    if (!params || !packed_bwd || !x || !grad_out || !grad_params || !ws || B <= 0 || H <= 0 || W <= 0)
        return SINDDM_E_BADARG;
    NetPlan p = make_plan(dim);
    if (!p.ok) return SINDDM_E_BADSHAPE;
    TrainBufs tb;
    if (carve_train(p, B, H, W, static_cast<char*>(ws), &tb) > ws_bytes) return SINDDM_E_WORKSPACE;
    return net_backward_impl(p, params, packed_bwd, x, grad_out, grad_params, grad_x, B, H, W, tb,
                             static_cast<hipStream_t>(stream));
}

int sinddm_l1_loss_fwd_bwd(const float* noise, const float* eps, float* loss_out, float* grad_out, int64_t n,
                           float grad_scale, void* stream) {
    if (!noise || !eps || !loss_out || n <= 0) return SINDDM_E_BADARG;
    long long bx = (n + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(l1_loss_kernel, dim3((unsigned)bx), dim3(256), 0, static_cast<hipStream_t>(stream), noise, eps,
                       loss_out, grad_out, (long long)n, 1.0f / (float)n, grad_scale);
    SINDDM_LAUNCH_CHECK();
    return 0;
}

int sinddm_adam_ema_step(float* p, const float* g, float* m, float* v, float* ema, float step_size, float beta1,
                         float beta2, float eps, float bc2_sqrt, float ema_decay, float reserved, int mode, int64_t n,
                         void* stream) {
    (void)reserved;
    if (!p || n <= 0) return SINDDM_E_BADARG;
    if ((mode & 1) && (!g || !m || !v)) return SINDDM_E_BADARG;
    if ((mode & 6) && !ema) return SINDDM_E_BADARG;
    long long bx = (n + 255) / 256;
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(adam_ema_kernel, dim3((unsigned)bx), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m, v,
                       ema, step_size, beta1, beta2, eps, bc2_sqrt, ema_decay, mode, (long long)n);
    SINDDM_LAUNCH_CHECK();
    return 0;
}

int sinddm_debug_conv_path(int dim, int B, int H, int W) {
    NetPlan p = make_plan(dim);
    if (!p.ok || B <= 0 || H <= 0 || W <= 0) return SINDDM_E_BADARG;
    const BlockPlan& b = p.blk[2];                      // the dim -> dim block
    return conv3x3_path(b.cout, b.cout, b.coblks, B, H, W);
}

int sinddm_debug_train_path(int dim, int B, int H, int W) {
    NetPlan p = make_plan(dim);
    if (!p.ok || B <= 0 || H <= 0 || W <= 0) return SINDDM_E_BADARG;
    const BlockPlan& b = p.blk[2];
    if (wino_enabled() && b.pk_q2 >= 0 && wh_applies(p, B, H, W, b.cout, b.cout)) return 8;
    return conv3x3_path(b.cout, b.cout, b.coblks, B, H, W);
}

int sinddm_debug_block_train(const float* params, const float* packed, const float* packed_bwd, int dim, int l,
                             const float* x, const float* cond_bias, const float* grad_y, float* y, float* grad_x,
                             float* grad_params, float* dcond, int B, int H, int W, void* ws, size_t ws_bytes,
                             void* stream) {
    if (!params || !packed || !packed_bwd || !x || !cond_bias || !grad_y || !y || !grad_params || !dcond || !ws ||
        l < 0 || l > 3 || B <= 0 || H <= 0 || W <= 0)
        return SINDDM_E_BADARG;
    NetPlan p = make_plan(dim);
    if (!p.ok) return SINDDM_E_BADSHAPE;
    TrainBufs tb;
    if (carve_train(p, B, H, W, static_cast<char*>(ws), &tb) > ws_bytes) return SINDDM_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const BlockPlan& b = p.blk[l];
    const size_t HW = (size_t)H * W;
    if (hipMemsetAsync(tb.amax, 0, (size_t)2 * B * AMAX_STRIDE * sizeof(float), st) != hipSuccess) return SINDDM_E_BADARG;
    int rc = block_forward(p, l, params, packed, x, cond_bias, b.cin, tb.h[l], tb.g[l], tb.o[l], tb.u[l], B, H, W, st, 0,
                           tb.amax + 2 * l);
    if (rc) return rc;
    if (hipMemcpyAsync(y, tb.o[l], B * b.cout * HW * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return SINDDM_E_BADARG;
    if (hipMemcpyAsync(tb.s[0], grad_y, B * b.cout * HW * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return SINDDM_E_BADARG;
    const BwdPack k = make_bwd_pack(p);
    rc = block_backward(p, k, l, params, packed_bwd, x, tb.s[0], tb.s[1], tb.s[2], grad_x, grad_params, tb, B, H, W, st,
                        tb.amax + (size_t)B * AMAX_STRIDE, false);
    if (rc) return rc;
    if (hipMemcpy2DAsync(dcond, b.cin * sizeof(float), tb.dcond + b.cond_off, p.cond_stride * sizeof(float),
                         b.cin * sizeof(float), B, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return SINDDM_E_BADARG;
    return 0;
}

int sinddm_debug_wgrad_map(int Cin, int Cout, int64_t ntiles, int ncu, uint32_t* wg_out, int wg_cap, int32_t* splits_out,
                           int splits_cap) {
    if (Cin < 1 || Cout < WW_CO || Cout % WW_CO || ntiles < 1 || !wg_out || !splits_out) return SINDDM_E_BADARG;
    WwArgs w{};
    w.Cin = Cin; w.Cout = Cout;
    w.coblks = Cout / WW_CO;
    w.ciblks = (Cin + WW_CI - 1) / WW_CI;
    w.ntiles = (int)(ntiles > 0x7fffffff ? 0x7fffffff : ntiles);
    const int n = ww_build_map(w, ncu);
    if (n <= 0) return SINDDM_E_BADSHAPE;
    if (n > wg_cap || w.coblks * w.ciblks > splits_cap) return SINDDM_E_BADARG;     // caller's buffers are too small
    for (int i = 0; i < n; ++i) wg_out[i] = w.map.wg[i];
    for (int q = 0; q < w.coblks * w.ciblks; ++q) splits_out[q] = w.map.S[q];
    return n;
}

}  // extern "C"
