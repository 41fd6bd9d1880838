// Host-side description of SinDDMNet(dim, channels=3, multiscale=True): where every tensor
// lives in the flat parameter buffer (nn.Module registration order, reference
// SinDDM/models.py:54-67,100-132) and in the MFMA-ready packed weight image.
#pragma once
#include <stdint.h>

namespace sinddm {

constexpr int KC = 8;          // input channels per LDS chunk (two k-steps of the 16x16x4 MFMA)
constexpr int TIME_DIM = 32;   // models.py:101
constexpr int CHANNELS = 3;

// conv_wh.h (Winograd F(2x4) with binary16 hi/lo frequency GEMMs): packed image [co block of 80][chunk of 16 ci][f 24][n 5][piece][k half][co 16][8 x f16]
inline bool wh_plan_ok(int cin, int cout) { return cin >= 32 && cin % 16 == 0 && cout % 80 == 0; }
inline long long wh_plan_halfs(int cin, int cout) { return (long long)(cout / 80) * (cin / 16) * 24 * 5 * 512; }

struct BlockPlan {
    int cin, cout;
    // flat parameter offsets (floats)
    int64_t mlp_w, mlp_b, tr_w, tr_b, dw_w, dw_b, c1_w, c1_b, c2_w, c2_b, res_w, res_b;  // res_* = -1 if Identity
    // MFMA tiling of the C_out dimension
    int mt;        // 16-row M tiles per workgroup (5, 2 or 1)
    int coblks;    // workgroups along C_out = ceil(cout / (16*mt))
    int co_lds;    // LDS/packed stride of the co axis (== 16 mod 32 -> conflict-free A reads)
    int nch1;      // conv1 3x3 chunks  = ceil(cin / KC)
    int nch2;      // conv2 3x3 chunks  = ceil(cout / KC)
    int nchr;      // residual 1x1 chunks = ceil(cin / KC) or 0
    // packed image offsets (floats)
    int64_t pk_c1, pk_c2, pk_res, pk_b1, pk_b2;
    // Winograd F(2x2,3x3) images (register layout [coblk][chunk of 16 ci][xi][ks][mt][lane])
    int nchw1, nchw2;          // 16-channel chunks of conv1 / conv2
    int64_t pk_wc1, pk_wc2;    // pk_wc1 = -1 when conv1 stays on the direct kernel (C_in < 8)
    // Winograd F(2x4,3x3) images of conv_wino3.h ([coblk][chunk][i][ks][q 0..7][lane][4]); -1 = shape not supported
    int64_t pk_w1f, pk_w2f;
    // conv_wh.h images and their per-output-channel 2^-e arrays; -1 = shape not supported
    int64_t pk_q1, pk_q2, pk_qs1, pk_qs2;
    int cond_off;  // offset of this block's per-sample bias inside the cond vector
};

struct NetPlan {
    int dim, half;
    BlockPlan blk[4];
    int64_t tm0_w, tm0_b, tm2_w, tm2_b, fin_w, fin_b;
    int64_t nparams, npacked;
    int64_t pk_zero;   // 64 zero floats at the end of the packed image (LDS-DMA zero-fill source)
    int ntensors;
    int64_t tensor_off[64];
    int cond_stride;   // floats per sample of the cond-bias vector (sum of cin, padded to 4)
    bool fp32_convs;   // per-call option SINDDM_DIM_FP32_CONVS: no launch takes the binary16 hi/lo kernels
    bool ok;
};

inline int mt_for(int cout) { return (cout % 80 == 0) ? 5 : ((cout % 32 == 0) ? 2 : 1); }
inline int co_lds_for(int mt) { int m = mt * 16; return (m % 32 == 16) ? m : m + 16; }

// dim_arg: the `dim` argument of the C ABI = SinDDMNet's width in the low 16 bits + option bits above (sinddm_hip.h);
// the layouts (parameters, packed images, workspaces) do not depend on the options
inline NetPlan make_plan(int dim_arg) {
    NetPlan p{};
    const int dim = dim_arg & 0xFFFF;
    p.fp32_convs = (dim_arg & 0x10000) != 0;
    if (dim_arg < 0 || (dim_arg >> 17) != 0) { p.ok = false; return p; }
    p.ok = dim >= 2 && dim % 2 == 0 && dim <= 1024;
    p.dim = dim;
    p.half = dim / 2;
    int64_t o = 0;
    int nt = 0;
    auto take = [&](int64_t n) { int64_t r = o; p.tensor_off[nt++] = r; o += n; return r; };
    p.tm0_w = take(TIME_DIM * 4 * TIME_DIM * 2);
    p.tm0_b = take(TIME_DIM * 4);
    p.tm2_w = take(TIME_DIM * TIME_DIM * 4);
    p.tm2_b = take(TIME_DIM);
    const int cins[4] = {CHANNELS, p.half, dim, dim};
    const int couts[4] = {p.half, dim, dim, p.half};
    int64_t q = 0;
    int coff = 0;
    for (int l = 0; l < 4; ++l) {
        BlockPlan& b = p.blk[l];
        b.cin = cins[l];
        b.cout = couts[l];
        b.mlp_w = take(TIME_DIM * TIME_DIM);
        b.mlp_b = take(TIME_DIM);
        b.tr_w = take((int64_t)b.cin * TIME_DIM);
        b.tr_b = take(b.cin);
        b.dw_w = take((int64_t)b.cin * 25);
        b.dw_b = take(b.cin);
        b.c1_w = take((int64_t)b.cout * b.cin * 9);
        b.c1_b = take(b.cout);
        b.c2_w = take((int64_t)b.cout * b.cout * 9);
        b.c2_b = take(b.cout);
        if (b.cin != b.cout) {
            b.res_w = take((int64_t)b.cout * b.cin);
            b.res_b = take(b.cout);
        } else {
            b.res_w = b.res_b = -1;
        }
        b.mt = mt_for(b.cout);
        b.coblks = (b.cout + b.mt * 16 - 1) / (b.mt * 16);
        b.co_lds = co_lds_for(b.mt);
        b.nch1 = (b.cin + KC - 1) / KC;
        b.nch2 = (b.cout + KC - 1) / KC;
        b.nchr = (b.res_w >= 0) ? (b.cin + KC - 1) / KC : 0;
        b.pk_c1 = q; q += (int64_t)b.coblks * b.nch1 * 9 * KC * b.co_lds;
        b.pk_c2 = q; q += (int64_t)b.coblks * b.nch2 * 9 * KC * b.co_lds;
        b.pk_res = q; q += (int64_t)b.coblks * b.nchr * KC * b.co_lds;
        b.pk_b1 = q; q += (int64_t)b.coblks * b.mt * 16;
        b.pk_b2 = q; q += (int64_t)b.coblks * b.mt * 16;
        b.nchw1 = (b.cin + 15) / 16;
        b.nchw2 = (b.cout + 15) / 16;
        if (b.cin >= 8) { b.pk_wc1 = q; q += (int64_t)b.coblks * b.nchw1 * 16 * 4 * b.mt * 64; } else b.pk_wc1 = -1;
        b.pk_wc2 = q; q += (int64_t)b.coblks * b.nchw2 * 16 * 4 * b.mt * 64;
        // (32768 floats per (co-block, chunk): 4 waves x 4 k-steps x 8 groups x 64 lanes x 4)
        const bool f24 = b.mt == 5 && b.cout % 80 == 0;
        if (f24 && b.cin >= 16 && b.cin % 16 == 0) { b.pk_w1f = q; q += (int64_t)b.coblks * b.nchw1 * 32768; } else b.pk_w1f = -1;
        if (f24 && b.cout % 16 == 0) { b.pk_w2f = q; q += (int64_t)b.coblks * b.nchw2 * 32768; } else b.pk_w2f = -1;
        q = (q + 63) / 64 * 64;
        if (wh_plan_ok(b.cin, b.cout)) {
            b.pk_q1 = q; q += wh_plan_halfs(b.cin, b.cout) / 2;
            b.pk_qs1 = q; q += (b.cout + 63) / 64 * 64;
        } else b.pk_q1 = b.pk_qs1 = -1;
        if (wh_plan_ok(b.cout, b.cout)) {
            b.pk_q2 = q; q += wh_plan_halfs(b.cout, b.cout) / 2;
            b.pk_qs2 = q; q += (b.cout + 63) / 64 * 64;
        } else b.pk_q2 = b.pk_qs2 = -1;
        b.cond_off = coff;
        coff += b.cin;
    }
    p.fin_w = take((int64_t)CHANNELS * p.half);
    p.fin_b = take(CHANNELS);
    p.nparams = o;
    p.pk_zero = q;
    q += 64;
    p.npacked = q;
    p.ntensors = nt;
    p.cond_stride = (coff + 3) / 4 * 4;
    return p;
}

}  // namespace sinddm
