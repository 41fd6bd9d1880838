// Implicit-GEMM 3x3 (+ fused 1x1 residual) convolution on the gfx950 fp32 matrix cores.
//
//   out[b][co][y][x] = epi( sum_{ci,tap} W3[co][ci][tap] * in [b][ci][y+dy][x+dx]
//                         + sum_{ci}     W1[co][ci]      * in2[b][ci][y][x]      + bias[co] )
//
// GEMM view: M = C_out (16-row tiles), N = pixels (16 consecutive pixels of one image row per
// tile), K = (ci, tap).  Instruction: v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain, 157 TF/s
// chip peak).  A workgroup = 4 waves owns a (TH x 32) pixel tile for MT*16 output channels; each
// wave owns MT x NT accumulator tiles (MT*16 channels x NT*16 pixels).  K is walked in chunks of
// KC=8 input channels: the chunk's weights ([tap][ci][co], pre-packed so the copy is linear
// float4) and the chunk's input halo tile ((TH+2) x 34 per channel) are staged in LDS; the loads of
// chunk c+1 are issued (LDS DMA, double buffered) before the MFMAs of chunk c so their latency hides under
// compute.  LDS strides are chosen == 16 (mod 32) so both operand reads are bank-conflict-free.
// Inside a chunk the A/B fragments of k-step i+1 are read from LDS before the MFMAs of k-step i
// (register double buffering), so the matrix pipe never waits on an LDS round trip.
//
// This replaces nn.Conv2d(dim, dim_out, 3, padding=1) [+ GELU] and
// nn.Conv2d(dim_out, dim_out, 3, padding=1) + res_conv(x) of SinDDMConvBlock
// (reference SinDDM/models.py:63-67,79-80); with transposed/flipped packed weights the same
// kernel is the data-gradient of those convolutions.
#pragma once
#include <stdlib.h>
#include "common.h"

namespace sinddm {

struct ConvArgs {
    const float* in;     // [B][Cin][H][W]   3x3 operand
    const float* in2;    // [B][Cin2][H][W]  1x1 operand (residual projection) or nullptr
    const float* resid;  // [B][Cout][H][W]  identity residual added in the epilogue, or nullptr
    const float* aux;    // [B][Cout][H][W]  pre-activation for act==2 (multiply by GELU'(aux))
    const float* w3;     // packed [coblk][chunk][9][KC][CO_LDS]
    const float* w1;     // packed [coblk][chunk][KC][CO_LDS]
    const float* bias;   // packed [coblk][MT*16] or nullptr
    const float* zero;   // >= 64 zero floats (tail of the packed image): source of LDS-DMA zero fill
    float* out;          // [B][Cout][H][W]
    float* out_pre;      // optional: pre-activation (value before GELU) for training, or nullptr
    int B, H, W, Cin, Cin2, Cout;
    int nch3, nch1;
    int tilesX, tilesY, ntiles, tiles_per_xcd;
    int coblks;
    int mtp;             // (Winograd kernel) m-tiles per PACKED output-channel block; 0 = same as the kernel's MT
    int act;             // 0 none, 1 GELU, 2 multiply by GELU'(aux)
    const float* wsinv;  // (conv_wh.h) per output channel 2^-e of the packed binary16 weight image
    const float* amax_in;   // (conv_wh.h) per-sample device scalars [b * AMAX_STRIDE]: max |in[b]| (its producer maintains them); nullptr = unit scale
    float* amax_out;     // optional per-sample device scalars [b * AMAX_STRIDE]: running max |out[b]| (guarded atomicMax), for the conv that reads `out` next
    int Wt;              // 0, or the TRUE image width when rows are padded to W (a multiple of 4) inside the library's own
                         // workspace: columns Wt .. W-1 of every input row hold zeros and are written as zeros
};

// geometry for NT 16-pixel tiles per wave (4 waves along N, tile width 32)
template <int NT, int WV = 4>
struct ConvGeom {
    static constexpr int TW = 32;
    static constexpr int TPR = TW / 16;
    static constexpr int RPW = NT / TPR;                 // tile rows per wave
    static constexpr int TH = WV * RPW;                  // tile height
    static constexpr int RS = TW + 2;
    static constexpr int HR = TH + 2;
    static constexpr int PS = ((HR * RS - 16 + 31) / 32) * 32 + 16;   // plane stride == 16 mod 32
    static constexpr int IN_ELEMS = KC * HR * RS;
    static constexpr int IREGS = (IN_ELEMS + CONV_THREADS - 1) / CONV_THREADS;
};

template <int MT, int NT, int WV = 4>
struct ConvCfg {
    using G = ConvGeom<NT, WV>;
    static constexpr int CO_LDS = (MT * 16) % 32 == 16 ? MT * 16 : MT * 16 + 16;
    static constexpr int W3_F4 = 9 * KC * CO_LDS / 4;   // float4 per 3x3 chunk
    static constexpr int W1_F4 = KC * CO_LDS / 4;       // float4 per 1x1 chunk
    static constexpr int WREGS = (W3_F4 + CONV_THREADS - 1) / CONV_THREADS;
    static constexpr int LDS_FLOATS = 9 * KC * CO_LDS + KC * G::PS;
};

template <int MT, int NT>
struct Frag {
    float a[MT];
    float b[NT];
};

template <int MT, int NT, int TAPS, int WV = 4>
__device__ __forceinline__ void conv_load_frag(Frag<MT, NT>& f, const float* __restrict__ sW,
                                               const float* __restrict__ sIn, int aBase, int bBase, int step) {
    using G = ConvGeom<NT, WV>;
    constexpr int CO_LDS = ConvCfg<MT, NT, WV>::CO_LDS;
    const int tap = step / (KC / 4), ks = step % (KC / 4);
    const int dy = (TAPS == 9) ? tap / 3 : 1;
    const int dx = (TAPS == 9) ? tap % 3 : 1;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) f.a[mt] = sW[aBase + (tap * KC + ks * 4) * CO_LDS + mt * 16];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
        f.b[nt] = sIn[bBase + ks * 4 * G::PS + ((nt / G::TPR) + dy) * G::RS + (nt % G::TPR) * 16 + dx];
}

template <int MT, int NT>
__device__ __forceinline__ void conv_mfma_frag(f32x4 (&acc)[MT][NT], const Frag<MT, NT>& f) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[mt], f.b[nt], acc[mt][nt], 0, 0, 0);
}

// VAR 0: let the compiler schedule LDS reads.  VAR 1: explicit register double buffering of the
// fragments, pinned with sched_barrier so that the reads of step i+1 are issued before the MFMAs of
// step i.
template <int MT, int NT, int TAPS, int VAR, int WV = 4>
__device__ __forceinline__ void conv_compute_chunk(f32x4 (&acc)[MT][NT], const float* __restrict__ sW,
                                                   const float* __restrict__ sIn, int aBase, int bBase) {
    constexpr int NSTEP = TAPS * (KC / 4);
    if constexpr (VAR == 0) {
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            Frag<MT, NT> f;
            conv_load_frag<MT, NT, TAPS, WV>(f, sW, sIn, aBase, bBase, st);
            conv_mfma_frag<MT, NT>(acc, f);
        }
    } else {
        Frag<MT, NT> f0, f1;
        conv_load_frag<MT, NT, TAPS, WV>(f0, sW, sIn, aBase, bBase, 0);
#pragma unroll
        for (int st = 0; st < NSTEP; st += 2) {
            if (st + 1 < NSTEP) conv_load_frag<MT, NT, TAPS, WV>(f1, sW, sIn, aBase, bBase, st + 1);
            __builtin_amdgcn_sched_barrier(0);
            conv_mfma_frag<MT, NT>(acc, f0);
            __builtin_amdgcn_sched_barrier(0);
            if (st + 1 < NSTEP) {
                if (st + 2 < NSTEP) conv_load_frag<MT, NT, TAPS, WV>(f0, sW, sIn, aBase, bBase, st + 2);
                __builtin_amdgcn_sched_barrier(0);
                conv_mfma_frag<MT, NT>(acc, f1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// =====================================================================================
// Both operands go global -> LDS with global_load_lds (no VGPR round trip, no ds_write pass), LDS is
// double buffered and there is ONE barrier per K chunk:
//     issue DMA(chunk c+1 -> buf[(c+1)&1]) ; MFMAs(chunk c from buf[c&1]) ; vmcnt(0) ; barrier
// The DMA destination is lane-linear (wave-uniform base + lane*size), so the LDS images are walked
// linearly (including the pad floats of every plane); halo / out-of-image / missing-channel
// elements are fetched from a page of zeros instead of being predicated off.
// =====================================================================================
template <int MT, int NT, int PIN, int WV>
__global__ __launch_bounds__(WV * 64, (WV == 16 ? 4 : (WV == 8 ? 4 : 1))) void conv_mfma_dma_kernel(ConvArgs p) {
    using Cfg = ConvCfg<MT, NT, WV>;
    using G = ConvGeom<NT, WV>;
    constexpr int THREADS = WV * 64;
    constexpr int CO_LDS = Cfg::CO_LDS;
    constexpr int WREGS = (Cfg::W3_F4 + THREADS - 1) / THREADS;
    constexpr int PLANE = G::HR * G::RS;
    constexpr int IN_LIN = KC * G::PS;                                   // multiple of 64
    constexpr int IREGS = (IN_LIN + THREADS - 1) / THREADS;
    constexpr int BUF = Cfg::LDS_FLOATS;                                 // floats per buffer
    static_assert(IN_LIN % 64 == 0, "input image must be whole wave-instructions");
    typedef __attribute__((address_space(3))) void* lds_ptr;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int slot = id >> 3;
    const int cb = slot % p.coblks;
    const int tl = slot / p.coblks;
    const int tile = xcd * p.tiles_per_xcd + tl;
    if (tile >= p.ntiles) return;
    const int tpi = p.tilesX * p.tilesY;
    const int b = tile / tpi;
    const int tr = tile - b * tpi;
    const int ty = tr / p.tilesX;
    const int tx = tr - ty * p.tilesX;
    const int y0 = ty * G::TH, x0 = tx * G::TW;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: no waterfall loops around the DMA
    const int l16 = lane & 15, kq = lane >> 4;
    const int H = p.H, W = p.W;
    const int HW = H * W;

    // per-thread source map of the input image (same for every chunk): element idx of the linear LDS
    // image [kc][PS] -> offset inside the chunk's 8 channel planes, or -1 (zero fill)
    int goff[IREGS];
#pragma unroll
    for (int i = 0; i < IREGS; ++i) {
        const int idx = tid + i * THREADS;
        const int kc = idx / G::PS;
        const int e = idx - kc * G::PS;
        const int r = e / G::RS;
        const int c = e - r * G::RS;
        const int gy = y0 + r - 1, gx = x0 + c - 1;
        const bool ok = idx < IN_LIN && e < PLANE && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goff[i] = ok ? kc * HW + gy * W + gx : -1;
    }
    const float* zsrc = p.zero + lane;

    const int nch = p.nch3 + p.nch1;
    auto issue = [&](int c, float* buf) {
        const bool is3 = c < p.nch3;
        const int cc = is3 ? c : c - p.nch3;
        const float* src = is3 ? p.in : p.in2;
        const int C = is3 ? p.Cin : p.Cin2;
        const int ch0 = cc * KC;
        const float* sbase = src + ((size_t)b * C + ch0) * HW;
        const int nvalid = C - ch0;
        const float4* wsrc;
        int n4;
        if (is3) {
            wsrc = reinterpret_cast<const float4*>(p.w3) + ((size_t)cb * p.nch3 + cc) * Cfg::W3_F4;
            n4 = Cfg::W3_F4;
        } else {
            wsrc = reinterpret_cast<const float4*>(p.w1) + ((size_t)cb * p.nch1 + cc) * Cfg::W1_F4;
            n4 = Cfg::W1_F4;
        }
#pragma unroll
        for (int j = 0; j < WREGS; ++j) {
            const int i4 = tid + j * THREADS;
            if (i4 < n4)
                __builtin_amdgcn_global_load_lds(wsrc + i4, (lds_ptr)(buf + (j * THREADS + wave * 64) * 4), 16, 0, 0);
        }
        float* ibuf = buf + 9 * KC * CO_LDS;
#pragma unroll
        for (int i = 0; i < IREGS; ++i) {
            const int idx = tid + i * THREADS;
            if (idx < IN_LIN) {
                const int kc = idx / G::PS;
                const float* g = (goff[i] >= 0 && kc < nvalid) ? sbase + goff[i] : zsrc;
                __builtin_amdgcn_global_load_lds(g, (lds_ptr)(ibuf + i * THREADS + wave * 64), 4, 0, 0);
            }
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int aBase = kq * CO_LDS + l16;
    const int bBase = 9 * KC * CO_LDS + kq * G::PS + (wave * G::RPW) * G::RS + l16;

    issue(0, smem);
    dma_barrier();                       // (vmcnt(0) + barrier: the pending DMA is published)
    int c = 0;
    for (; c < p.nch3; ++c) {              // 3x3 chunks
        float* cur = smem + (c & 1) * BUF;
        if (c + 1 < nch) issue(c + 1, smem + ((c + 1) & 1) * BUF);
        conv_compute_chunk<MT, NT, 9, PIN, WV>(acc, cur, cur, aBase, bBase);
        dma_barrier();                   // everyone done with `cur`; DMA of chunk c+1 has landed
    }
    for (; c < nch; ++c) {                 // fused 1x1 (residual projection) chunks
        float* cur = smem + (c & 1) * BUF;
        if (c + 1 < nch) issue(c + 1, smem + ((c + 1) & 1) * BUF);
        conv_compute_chunk<MT, NT, 1, PIN, WV>(acc, cur, cur, aBase, bBase);
        dma_barrier();
    }

#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = mt * 16 + kq * 4 + r;
            const int co = cb * (MT * 16) + col;
            if (co >= p.Cout) continue;
            const float bv = p.bias ? p.bias[cb * (MT * 16) + col] : 0.0f;
            const size_t cbase = ((size_t)b * p.Cout + co) * HW;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int y = y0 + wave * G::RPW + nt / G::TPR;
                const int x = x0 + (nt % G::TPR) * 16 + l16;
                if (y < H && x < W) {
                    const size_t o = cbase + (size_t)y * W + x;
                    float v = acc[mt][nt][r] + bv;
                    if (p.out_pre) p.out_pre[o] = v;
                    if (p.act == 1) v = gelu_erf(v);
                    else if (p.act == 2) v *= gelu_erf_grad(p.aux[o]);
                    if (p.resid) v += p.resid[o];
                    p.out[o] = v;
                }
            }
        }
    }
}

// ---- optional launch profiler (bench.py's roofline leg): HIP events around every conv launch ----
struct ConvProfiler {
    bool on = false;
    int used = 0;
    double flops = 0.0;        // algorithmic (direct-convolution) FLOPs
    double exec_flops = 0.0;   // FLOPs the matrix cores actually executed (Winograd: 16/36 of the above)
    static constexpr int MAXREC = 8192;
    hipEvent_t ev[2 * MAXREC];
    int created = 0;
    // per record: which kernel family (1 = Winograd 3x3, 2 = 1x1, 3 = direct 3x3) and its FLOP counts
    unsigned char kind[MAXREC];
    unsigned char gen[MAXREC];   // Winograd launches: kernel generation (4 = conv_wino4, 3 = conv_wino3, 2 = conv_wino2, 1 = conv_wino)
    double rec_flops[MAXREC], rec_exec[MAXREC];
    void note(int k, double fl, double ex, int g = 0) {
        kind[used] = (unsigned char)k; gen[used] = (unsigned char)g; rec_flops[used] = fl; rec_exec[used] = ex;
        flops += fl; exec_flops += ex; ++used;
    }
};
ConvProfiler& conv_profiler();

// SINDDM_CONV_VAR (compile time, common.h) = 4 forces 4-wave workgroups on 4x32 tiles, 8 forces 8-wave workgroups on
// 8x32 tiles; default (-1) picks per launch
inline int conv_tuning_var() { return (SINDDM_CONV_VAR == 4 || SINDDM_CONV_VAR == 8) ? SINDDM_CONV_VAR : -1; }

template <int MT, int NT, int PIN, int WV>
inline void conv_launch_dma_t(const ConvArgs& a, unsigned grid, hipStream_t st) {
    constexpr size_t lds = 2 * ConvCfg<MT, NT, WV>::LDS_FLOATS * sizeof(float);
    hipLaunchKernelGGL((conv_mfma_dma_kernel<MT, NT, PIN, WV>), dim3(grid), dim3(WV * 64), lds, st, a);
}

template <int MT>
inline void conv_launch_mt(const ConvArgs& a, unsigned grid, int var, hipStream_t st) {
    if (var == 8) conv_launch_dma_t<MT, 2, 0, 8>(a, grid, st);   // 8 waves x (MT x 2) tiles: 8x32 pixel tile
    else conv_launch_dma_t<MT, 2, 0, 4>(a, grid, st);            // 4 waves x (MT x 2) tiles: 4x32 pixel tile
}

inline int conv_launch(const ConvArgs& a_in, int mt, hipStream_t st) {
    ConvArgs a = a_in;
    int var = conv_tuning_var();
    if (var < 0) {
        // default: 8-wave workgroups on 8x32 pixel tiles (4 waves/SIMD at 2 workgroups per CU); when that would
        // leave CUs without two workgroups (coarse pyramid scales, tiny batches) 4-wave workgroups on 4x32 tiles
        const long long blocks8 = (long long)a.B * ((a.W + 31) / 32) * ((a.H + 7) / 8) * a.coblks;
        var = blocks8 >= 512 ? 8 : 4;
    }
    const int TH = var == 8 ? ConvGeom<2, 8>::TH : ConvGeom<2, 4>::TH;
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
    a.tilesX = (a.W + 31) / 32;
    a.tilesY = (a.H + TH - 1) / TH;
    a.ntiles = a.B * a.tilesX * a.tilesY;
    a.tiles_per_xcd = (a.ntiles + 7) / 8;
    const unsigned grid = (unsigned)(a.tiles_per_xcd * 8 * a.coblks);
    switch (mt) {
        case 5: conv_launch_mt<5>(a, grid, var, st); break;
        case 2: conv_launch_mt<2>(a, grid, var, st); break;
        case 1: conv_launch_mt<1>(a, grid, var, st); break;
        default: return SINDDM_E_BADSHAPE;
    }
    if (rec) {
        (void)hipEventRecord(prof.ev[2 * prof.used + 1], st);
        const double fl = 2.0 * a.B * a.H * a.W * (double)a.Cout * (9.0 * a.Cin * (a.nch3 > 0) + (double)a.Cin2 * (a.nch1 > 0));
        prof.note(3, fl, fl);
    }
    SINDDM_LAUNCH_CHECK();
    return 0;
}

}  // namespace sinddm
