// Implicit-GEMM 3x3 (+ fused 1x1 residual) convolution on the gfx950 fp32 matrix cores.
//
//   out[b][co][y][x] = epi( sum_{ci,tap} W3[co][ci][tap] * in [b][ci][y+dy][x+dx]
//                         + sum_{ci}     W1[co][ci]      * in2[b][ci][y][x]      + bias[co] )
//
// GEMM view: M = C_out (16-row tiles), N = pixels (16 consecutive pixels of one image row per
// tile), K = (ci, tap).  Instruction: v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain, 157 TF/s
// chip peak).  A workgroup = 4 waves owns an 8x32 pixel tile for MT*16 output channels; each
// wave owns MT x 4 accumulator tiles (MT*16 channels x 64 pixels).  K is walked in chunks of
// KC=8 input channels: the chunk's weights ([tap][ci][co], pre-packed so the copy is linear
// float4) and the chunk's input halo tile (10x34 per channel) are staged in LDS; the loads of
// chunk c+1 are issued into registers before the MFMAs of chunk c so their latency hides under
// compute.  LDS strides are chosen == 16 (mod 32) so both operand reads are bank-conflict-free.
//
// This replaces nn.Conv2d(dim, dim_out, 3, padding=1) [+ GELU] and
// nn.Conv2d(dim_out, dim_out, 3, padding=1) + res_conv(x) of SinDDMConvBlock
// (reference SinDDM/models.py:63-67,79-80); with transposed/flipped packed weights the same
// kernel is the data-gradient of those convolutions.
#pragma once
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
    float* out;          // [B][Cout][H][W]
    float* out_pre;      // optional: pre-activation (value before GELU) for training, or nullptr
    int B, H, W, Cin, Cin2, Cout;
    int nch3, nch1;
    int tilesX, tilesY, ntiles, tiles_per_xcd;
    int coblks;
    int act;             // 0 none, 1 GELU, 2 multiply by GELU'(aux)
};

template <int MT>
struct ConvCfg {
    static constexpr int CO_LDS = (MT * 16) % 32 == 16 ? MT * 16 : MT * 16 + 16;
    static constexpr int W3_F4 = 9 * KC * CO_LDS / 4;   // float4 per 3x3 chunk
    static constexpr int W1_F4 = KC * CO_LDS / 4;       // float4 per 1x1 chunk
    static constexpr int WREGS = (W3_F4 + CONV_THREADS - 1) / CONV_THREADS;
    static constexpr int LDS_FLOATS = 9 * KC * CO_LDS + KC * CONV_PS;
};

template <int MT, int TAPS>
__device__ __forceinline__ void conv_compute_chunk(f32x4 (&acc)[MT][CONV_NT], const float* __restrict__ sW,
                                                   const float* __restrict__ sIn, int aBase, int bBase) {
    constexpr int CO_LDS = ConvCfg<MT>::CO_LDS;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
        const int dy = (TAPS == 9) ? tap / 3 : 1;
        const int dx = (TAPS == 9) ? tap % 3 : 1;
#pragma unroll
        for (int ks = 0; ks < KC / 4; ++ks) {
            float a[MT], b[CONV_NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = sW[aBase + (tap * KC + ks * 4) * CO_LDS + mt * 16];
#pragma unroll
            for (int nt = 0; nt < CONV_NT; ++nt)
                b[nt] = sIn[bBase + ks * 4 * CONV_PS + ((nt / CONV_TPR) + dy) * CONV_RS + (nt % CONV_TPR) * 16 + dx];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < CONV_NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
        }
    }
}

template <int MT>
__global__ __launch_bounds__(CONV_THREADS) void conv_mfma_kernel(ConvArgs p) {
    using Cfg = ConvCfg<MT>;
    constexpr int CO_LDS = Cfg::CO_LDS;
    constexpr int WREGS = Cfg::WREGS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sW = smem;
    float* sIn = smem + 9 * KC * CO_LDS;

    // XCD-aware decode: each XCD (private L2) gets a contiguous range of tiles; the co-blocks of
    // one tile run back to back on the same XCD so the input tile is served from that L2.
    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int slot = id >> 3;
    const int cb = slot % p.coblks;
    const int tl = slot / p.coblks;
    if (tl >= p.tiles_per_xcd) return;
    const int tile = xcd * p.tiles_per_xcd + tl;
    if (tile >= p.ntiles) return;
    const int tpi = p.tilesX * p.tilesY;
    const int b = tile / tpi;
    const int tr = tile - b * tpi;
    const int ty = tr / p.tilesX;
    const int tx = tr - ty * p.tilesX;
    const int y0 = ty * CONV_TH, x0 = tx * CONV_TW;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l16 = lane & 15, kq = lane >> 4;
    const int H = p.H, W = p.W;
    const int HW = H * W;

    // ---- per-thread staging map (same for every chunk) ----
    int goff[CONV_IREGS];
#pragma unroll
    for (int i = 0; i < CONV_IREGS; ++i) {
        const int idx = tid + i * CONV_THREADS;
        const int kc = idx / (CONV_HR * CONV_RS);
        const int e = idx - kc * (CONV_HR * CONV_RS);
        const int r = e / CONV_RS;
        const int c = e - r * CONV_RS;
        const int gy = y0 + r - 1, gx = x0 + c - 1;
        const bool ok = idx < CONV_IN_ELEMS && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goff[i] = ok ? kc * HW + gy * W + gx : -1;
    }

    float4 wreg[WREGS];
    float ireg[CONV_IREGS];

    const int nch = p.nch3 + p.nch1;
    auto load_chunk = [&](int c) {
        const bool is3 = c < p.nch3;
        const int cc = is3 ? c : c - p.nch3;
        const float* src = is3 ? p.in : p.in2;
        const int C = is3 ? p.Cin : p.Cin2;
        const int ch0 = cc * KC;
        const float* sbase = src + ((size_t)b * C + ch0) * HW;
        const int nvalid = C - ch0;  // channels of this chunk that exist
#pragma unroll
        for (int i = 0; i < CONV_IREGS; ++i) {
            const int idx = tid + i * CONV_THREADS;
            const int kc = idx / (CONV_HR * CONV_RS);
            ireg[i] = (goff[i] >= 0 && kc < nvalid) ? sbase[goff[i]] : 0.0f;
        }
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
            const int i4 = tid + j * CONV_THREADS;
            if (i4 < n4) wreg[j] = wsrc[i4];
        }
    };
    auto store_chunk = [&](int c) {
        const int n4 = (c < p.nch3) ? Cfg::W3_F4 : Cfg::W1_F4;
#pragma unroll
        for (int j = 0; j < WREGS; ++j) {
            const int i4 = tid + j * CONV_THREADS;
            if (i4 < n4) reinterpret_cast<float4*>(sW)[i4] = wreg[j];
        }
#pragma unroll
        for (int i = 0; i < CONV_IREGS; ++i) {
            const int idx = tid + i * CONV_THREADS;
            if (idx < CONV_IN_ELEMS) {
                const int kc = idx / (CONV_HR * CONV_RS);
                const int e = idx - kc * (CONV_HR * CONV_RS);
                sIn[kc * CONV_PS + e] = ireg[i];
            }
        }
    };

    f32x4 acc[MT][CONV_NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < CONV_NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int aBase = kq * CO_LDS + l16;
    const int bBase = kq * CONV_PS + (wave * (CONV_NT / CONV_TPR)) * CONV_RS + l16;

    load_chunk(0);
    for (int c = 0; c < nch; ++c) {
        __syncthreads();            // every wave finished reading the previous chunk
        store_chunk(c);
        __syncthreads();
        if (c + 1 < nch) load_chunk(c + 1);   // in flight during the MFMAs below
        if (c < p.nch3)
            conv_compute_chunk<MT, 9>(acc, sW, sIn, aBase, bBase);
        else
            conv_compute_chunk<MT, 1>(acc, sW, sIn, aBase, bBase);
    }

    // ---- epilogue: bias, activation, residual, store (C/D layout: col = lane&15 -> pixel,
    //      row = (lane>>4)*4 + r -> output channel) ----
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
            for (int nt = 0; nt < CONV_NT; ++nt) {
                const int y = y0 + wave * (CONV_NT / CONV_TPR) + nt / CONV_TPR;
                const int x = x0 + (nt % CONV_TPR) * 16 + l16;
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
    double flops = 0.0;
    static constexpr int MAXREC = 8192;
    hipEvent_t ev[2 * MAXREC];
    int created = 0;
};
ConvProfiler& conv_profiler();

inline int conv_launch(const ConvArgs& a_in, int mt, hipStream_t st) {
    ConvArgs a = a_in;
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
    a.tilesX = (a.W + CONV_TW - 1) / CONV_TW;
    a.tilesY = (a.H + CONV_TH - 1) / CONV_TH;
    a.ntiles = a.B * a.tilesX * a.tilesY;
    a.tiles_per_xcd = (a.ntiles + 7) / 8;
    const unsigned grid = (unsigned)(a.tiles_per_xcd * 8 * a.coblks);
    switch (mt) {
        case 5:
            hipLaunchKernelGGL(conv_mfma_kernel<5>, dim3(grid), dim3(CONV_THREADS),
                               ConvCfg<5>::LDS_FLOATS * sizeof(float), st, a);
            break;
        case 2:
            hipLaunchKernelGGL(conv_mfma_kernel<2>, dim3(grid), dim3(CONV_THREADS),
                               ConvCfg<2>::LDS_FLOATS * sizeof(float), st, a);
            break;
        case 1:
            hipLaunchKernelGGL(conv_mfma_kernel<1>, dim3(grid), dim3(CONV_THREADS),
                               ConvCfg<1>::LDS_FLOATS * sizeof(float), st, a);
            break;
        default:
            return SINDDM_E_BADSHAPE;
    }
    if (rec) {
        (void)hipEventRecord(prof.ev[2 * prof.used + 1], st);
        prof.flops += 2.0 * a.B * a.H * a.W * (double)a.Cout * (9.0 * a.Cin * (a.nch3 > 0) + (double)a.Cin2 * (a.nch1 > 0));
        ++prof.used;
    }
    SINDDM_LAUNCH_CHECK();
    return 0;
}

}  // namespace sinddm
