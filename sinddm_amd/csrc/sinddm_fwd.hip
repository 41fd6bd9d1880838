// Forward path of the SinDDM hot path for gfx950: weight packing, conditioning MLP, depthwise
// 5x5, MFMA 3x3 convs, final 1x1, and the HBM-bound diffusion elementwise kernels.
#include "conv_mfma.h"
#include "conv_wino.h"
#include "conv_wino4.h"
#include "conv_wh.h"
#include "internal.h"
namespace sinddm {

ConvProfiler& conv_profiler() {
    static ConvProfiler p;
    return p;
}

// =====================================================================================
// weight packing: flat nn.Module-order parameters -> MFMA chunk images
// =====================================================================================
__global__ void pack_kernel(const float* __restrict__ params, float* __restrict__ packed, PackArgs a) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.total) return;
    int s = 0;
    long long base = 0;
    while (s < a.nseg - 1 && i >= base + a.seg[s].count) { base += a.seg[s].count; ++s; }
    const PackSeg& g = a.seg[s];
    const long long j = i - base;
    float v = 0.0f;
    if (g.kind == 0) {
        // [coblk][chunk][tap][kc][co_lds]
        const int co_l = (int)(j % g.co_lds);
        long long r = j / g.co_lds;
        const int kc = (int)(r % KC); r /= KC;
        const int tap = (int)(r % g.taps); r /= g.taps;
        const int ch = (int)(r % g.nch); r /= g.nch;
        const int cb = (int)r;
        const int m = cb * g.mt * 16 + co_l;      // GEMM row (output channel of THIS conv)
        const int k = ch * KC + kc;               // GEMM k channel (input channel of THIS conv)
        if (!g.transpose) {
            if (co_l < g.mt * 16 && m < g.cout && k < g.cin)
                v = params[g.w + ((long long)m * g.cin + k) * g.taps + tap];
        } else {
            // data-gradient image: this conv maps forward-cout channels (k) to forward-cin channels (m)
            // with spatially flipped taps:  W'[m][k][tap] = W[k][m][taps-1-tap]
            if (co_l < g.mt * 16 && m < g.cin && k < g.cout)
                v = params[g.w + ((long long)k * g.cin + m) * g.taps + (g.taps - 1 - tap)];
        }
    } else if (g.kind == 3) {
        // Winograd F(2x2,3x3): U_xi = G g G^T in the MFMA A-operand register image
        // [coblk][chunk of 16 ci][xi][ks][mt][lane]; lane -> (co = lane&15, ci = lane>>4)
#if SINDDM_WINO_V2
        // second-generation kernel (conv_wino2.h): [coblk][chunk][i][ks][mt][lane][j] -- a lane's four frequencies
        // (i, 0..3) are one 16-byte load; frequency row 2 is stored negated (the kernel evaluates d1 - d2 for it)
        long long r = j;
        const int fj_ = (int)(r % 4); r /= 4;
        const int lane = (int)(r % 64); r /= 64;
        const int mt = (int)(r % g.mt); r /= g.mt;
        const int ks = (int)(r % 4); r /= 4;
        const int fi_ = (int)(r % 4); r /= 4;
        const int xi = fi_ * 4 + fj_;
        const int ch = (int)(r % g.nch); r /= g.nch;
        const int cb = (int)r;
#else
        const int lane = (int)(j % 64);
        long long r = j / 64;
        const int mt = (int)(r % g.mt); r /= g.mt;
        const int ks = (int)(r % 4); r /= 4;
        const int xi = (int)(r % 16); r /= 16;
        const int ch = (int)(r % g.nch); r /= g.nch;
        const int cb = (int)r;
#endif
        const int m = cb * g.mt * 16 + mt * 16 + (lane & 15);
        const int k = ch * 16 + ks * 4 + (lane >> 4);
        const int M = g.transpose ? g.cin : g.cout;      // rows of THIS conv
        const int K = g.transpose ? g.cout : g.cin;      // reduction channels of THIS conv
        if (m < M && k < K) {
            float gt[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b2 = 0; b2 < 3; ++b2) {
                    const int tap = a * 3 + b2;
                    gt[a][b2] = g.transpose ? params[g.w + ((long long)k * g.cin + m) * 9 + (8 - tap)]
                                            : params[g.w + ((long long)m * g.cin + k) * 9 + tap];
                }
            const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
            const int fi = xi >> 2, fj = xi & 3;
            float acc = 0.f;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float rowv = 0.f;
#pragma unroll
                for (int b2 = 0; b2 < 3; ++b2) rowv += G[fj][b2] * gt[a][b2];
                acc += G[fi][a] * rowv;
            }
            v = (SINDDM_WINO_V2 && fi == 2) ? -acc : acc;
        }
    } else if (g.kind == 4) {
        // Winograd F(2x4,3x3) of conv_wino3.h: U = G2 g G4^T, register image [coblk][chunk][i][ks][q][lane][slot];
        // slot 4 q + slot -> pair e = mt * 6 + j in the order of w3_pos_e (two halves of 15 pairs + a padding slot);
        // vertical row 2 stored negated
        long long r = j;
        const int slot = (int)(r % 4); r /= 4;
        const int lane = (int)(r % 64); r /= 64;
        const int q8 = (int)(r % 8); r /= 8;
        const int ks = (int)(r % 4); r /= 4;
        const int fi = (int)(r % 4); r /= 4;
        const int ch = (int)(r % g.nch); r /= g.nch;
        const int cb = (int)r;
        const int e = w3_pos_e(q8 * 4 + slot);
        const int mt = e < 0 ? 0 : e / 6, fj = e < 0 ? 0 : e - mt * 6;
        const int m = cb * 80 + mt * 16 + (lane & 15);
        const int k = ch * 16 + ks * 4 + (lane >> 4);
        const int M4 = g.transpose ? g.cin : g.cout, K4 = g.transpose ? g.cout : g.cin;
        if (e >= 0 && m < M4 && k < K4) {
            const double G2[4][3] = {{1., 0., 0.}, {.5, .5, .5}, {.5, -.5, .5}, {0., 0., 1.}};
            const double G4[6][3] = {{1. / 4, 0., 0.}, {-1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6},
                                     {1. / 24, 1. / 12, 1. / 6}, {1. / 24, -1. / 12, 1. / 6}, {0., 0., 1.}};
            double acc = 0.;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                double rowv = 0.;
#pragma unroll
                for (int b2 = 0; b2 < 3; ++b2)
                    rowv += G4[fj][b2] * (double)(g.transpose ? params[g.w + ((long long)k * g.cin + m) * 9 + (2 - a) * 3 + (2 - b2)]
                                                              : params[g.w + ((long long)m * g.cin + k) * 9 + a * 3 + b2]);
                acc += G2[fi][a] * rowv;
            }
            v = (float)(fi == 2 ? -acc : acc);
        }
    } else if (g.kind == 1) {
        if (j < g.cout) {
            v = params[g.w + j];
            if (g.w2 >= 0) v += params[g.w2 + j];
        }
    }                       // kind 2: zero page
    packed[g.dst + j] = v;
}

int pack_launch(const float* params, float* packed, const PackArgs& a, hipStream_t st) {
    const int threads = 256;
    const unsigned grid = (unsigned)((a.total + threads - 1) / threads);
    hipLaunchKernelGGL(pack_kernel, dim3(grid), dim3(threads), 0, st, params, packed, a);
    SINDDM_LAUNCH_CHECK();
    return 0;
}

static int pack_forward(const NetPlan& P, const float* params, float* packed, hipStream_t st) {
    PackArgs a{};
    int n = 0;
    long long total = 0;
    auto add = [&](PackSeg s) { a.seg[n++] = s; total += s.count; };
    for (int l = 0; l < 4; ++l) {
        const BlockPlan& b = P.blk[l];
        PackSeg s{};
        s.kind = 0; s.mt = b.mt; s.co_lds = b.co_lds; s.transpose = 0; s.w2 = -1;
        // conv1: cin -> cout, 3x3
        s.dst = b.pk_c1; s.count = (long long)b.coblks * b.nch1 * 9 * KC * b.co_lds;
        s.w = b.c1_w; s.cin = b.cin; s.cout = b.cout; s.taps = 9; s.nch = b.nch1;
        add(s);
        // conv2: cout -> cout, 3x3
        s.dst = b.pk_c2; s.count = (long long)b.coblks * b.nch2 * 9 * KC * b.co_lds;
        s.w = b.c2_w; s.cin = b.cout; s.cout = b.cout; s.taps = 9; s.nch = b.nch2;
        add(s);
        if (b.nchr > 0) {
            s.dst = b.pk_res; s.count = (long long)b.coblks * b.nchr * KC * b.co_lds;
            s.w = b.res_w; s.cin = b.cin; s.cout = b.cout; s.taps = 1; s.nch = b.nchr;
            add(s);
        }
        PackSeg wz{};
        wz.kind = 3; wz.mt = b.mt; wz.transpose = 0; wz.w2 = -1; wz.taps = 9;
        if (b.pk_wc1 >= 0) {
            wz.dst = b.pk_wc1; wz.nch = b.nchw1; wz.count = (long long)b.coblks * b.nchw1 * 16 * 4 * b.mt * 64;
            wz.w = b.c1_w; wz.cin = b.cin; wz.cout = b.cout;
            add(wz);
        }
        wz.dst = b.pk_wc2; wz.nch = b.nchw2; wz.count = (long long)b.coblks * b.nchw2 * 16 * 4 * b.mt * 64;
        wz.w = b.c2_w; wz.cin = b.cout; wz.cout = b.cout;
        add(wz);
        PackSeg t{};
        t.kind = 1; t.cout = b.cout; t.count = (long long)b.coblks * b.mt * 16;
        t.dst = b.pk_b1; t.w = b.c1_b; t.w2 = -1;
        add(t);
        t.dst = b.pk_b2; t.w = b.c2_b; t.w2 = b.res_b;
        add(t);
    }
    PackSeg z{};
    z.kind = 2; z.dst = P.pk_zero; z.count = 64;
    add(z);
    a.nseg = n;
    a.total = total;
    int rc = pack_launch(params, packed, a, st);
    if (rc) return rc;
    // second launch: the F(2x4) Winograd images of conv_wino3.h
    PackArgs f{};
    n = 0;
    total = 0;
    auto addf = [&](PackSeg sg) { f.seg[n++] = sg; total += sg.count; };
    for (int l = 0; l < 4; ++l) {
        const BlockPlan& b = P.blk[l];
        PackSeg wz{};
        wz.kind = 4; wz.mt = b.mt; wz.transpose = 0; wz.w2 = -1; wz.taps = 9;
        if (b.pk_w1f >= 0) {
            wz.dst = b.pk_w1f; wz.nch = b.nchw1; wz.count = (long long)b.coblks * b.nchw1 * 32768;
            wz.w = b.c1_w; wz.cin = b.cin; wz.cout = b.cout;
            addf(wz);
        }
        if (b.pk_w2f >= 0) {
            wz.dst = b.pk_w2f; wz.nch = b.nchw2; wz.count = (long long)b.coblks * b.nchw2 * 32768;
            wz.w = b.c2_w; wz.cin = b.cout; wz.cout = b.cout;
            addf(wz);
        }
    }
    if (n > 0) {
        f.nseg = n;
        f.total = total;
        rc = pack_launch(params, packed, f, st);
        if (rc) return rc;
    }
    // the binary16 hi/lo images of conv_wh.h and their per-channel scales
    for (int l = 0; l < 4; ++l) {
        const BlockPlan& b = P.blk[l];
        if (b.pk_q1 >= 0) {
            rc = wh_pack_launch(params + b.c1_w, packed + b.pk_qs1, packed + b.pk_q1, b.cin, b.cout, 0, st);
            if (rc) return rc;
        }
        if (b.pk_q2 >= 0) {
            rc = wh_pack_launch(params + b.c2_w, packed + b.pk_qs2, packed + b.pk_q2, b.cout, b.cout, 0, st);
            if (rc) return rc;
        }
    }
    return 0;
}

// =====================================================================================
// conditioning: sinusoidal embeddings -> time_mlp -> per-block (mlp, time_reshape)
// reference SinDDM/models.py:39-46,106-110,136-141 and :54-60,74-76
// =====================================================================================
struct CondArgs {
    const float* params;
    const long long* t_dev;
    int t_host;
    int t_step;          // t of block b = t_host + b * t_step when t_dev is null (a run of sampler steps in one launch)
    float scale;
    float* out;          // [B][cond_stride]
    int cond_stride;
    long long tm0_w, tm0_b, tm2_w, tm2_b;
    long long mlp_w[4], mlp_b[4], tr_w[4], tr_b[4];
    int cin[4], coff[4];
    float* cond_vec;     // optional [B][32] raw cond vector (saved for backward), may be null
    float* hidden;       // optional [B][128] pre-GELU hidden of time_mlp (saved for backward)
    float* emb_out;      // optional [B][64] sinusoidal embedding
    float* mvec_out;     // optional [B][4][32] per-block mlp outputs
};

__global__ __launch_bounds__(128) void cond_kernel(CondArgs a) {
    __shared__ float emb[64];
    __shared__ float h1[128];
    __shared__ float cv[32];
    __shared__ float gv[32];
    __shared__ float mv[32];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const float* P = a.params;
    const float tval = a.t_dev ? (float)a.t_dev[b] : (float)(a.t_host + b * a.t_step);
    if (tid < 32) {
        // f_i = exp(i * -(ln 1e4 / 15)): torch.exp(arange(16) * -emb) on CPU returns the correctly
        // rounded fp32 exp of the fp32 argument; a double-precision exp rounded to fp32 reproduces it.
        const int i = tid & 15;
        const float f = (float)exp((double)((float)i * -0.6140226914650789f));  // ln(10000)/15
        const float x = (tid < 16) ? tval : a.scale;
        const float arg = x * f;
        const int o = (tid < 16) ? 0 : 32;
        emb[o + i] = sinf(arg);
        emb[o + 16 + i] = cosf(arg);
    }
    __syncthreads();
    if (a.emb_out && tid < 64) a.emb_out[(long long)b * 64 + tid] = emb[tid];
    {
        float s = P[a.tm0_b + tid];
        const float* w = P + a.tm0_w + (long long)tid * 64;
#pragma unroll 8
        for (int k = 0; k < 64; ++k) s = fmaf(w[k], emb[k], s);
        if (a.hidden) a.hidden[(long long)b * 128 + tid] = s;
        h1[tid] = gelu_erf(s);
    }
    __syncthreads();
    if (tid < 32) {
        float s = P[a.tm2_b + tid];
        const float* w = P + a.tm2_w + (long long)tid * 128;
#pragma unroll 8
        for (int k = 0; k < 128; ++k) s = fmaf(w[k], h1[k], s);
        cv[tid] = s;
        gv[tid] = gelu_erf(s);
        if (a.cond_vec) a.cond_vec[(long long)b * 32 + tid] = s;
    }
    __syncthreads();
    for (int l = 0; l < 4; ++l) {
        if (tid < 32) {
            float s = P[a.mlp_b[l] + tid];
            const float* w = P + a.mlp_w[l] + (long long)tid * 32;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) s = fmaf(w[k], gv[k], s);
            mv[tid] = s;
            if (a.mvec_out) a.mvec_out[((long long)b * 4 + l) * 32 + tid] = s;
        }
        __syncthreads();
        for (int c = tid; c < a.cin[l]; c += blockDim.x) {
            float s = P[a.tr_b[l] + c];
            const float* w = P + a.tr_w[l] + (long long)c * 32;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) s = fmaf(w[k], mv[k], s);
            a.out[(long long)b * a.cond_stride + a.coff[l] + c] = s;
        }
        __syncthreads();
    }
}

// =====================================================================================
// depthwise 5x5 + bias + per-sample condition      reference SinDDM/models.py:61,70,77
// HBM-bound: 8 B per (channel,pixel).  A 16x128 tile is staged with its 2-pixel halo in LDS; wave w owns tile rows
// 4w..4w+3 and lane l owns columns l and l+64: one ds_read2_b32 fetches both (every wave access is 2 x 64
// consecutive floats) and the 25 taps become packed FMAs (v_pk_fma_f32: weight broadcast x column pair) -- the
// kernel is as much VALU- as bandwidth-limited (25 FMA per 8 bytes).  With flip=1 the taps are mirrored (transposed
// conv = data gradient) and `addt` is added to the result (residual-path gradient).
// =====================================================================================
#ifndef DW_RW_
#define DW_RW_ 4
#endif
constexpr int DW_RW = DW_RW_;             // output rows per wave
constexpr int DW_TH = 4 * DW_RW, DW_TW = 128, DW_RS = DW_TW + 4, DW_HR = DW_TH + 4;
#ifndef DW_NX_
#define DW_NX_ 2
#endif
constexpr int DW_NX = DW_NX_;   // x-tiles per workgroup: all their loads are in flight together (latency-bound otherwise)

__global__ __launch_bounds__(256) void dwconv5_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, const float* __restrict__ cond,
                                                       int cond_stride, const float* __restrict__ addt, int flip,
                                                       float* __restrict__ out, int C, int H, int W, int groupsX, int pi,
                                                       int po, float* __restrict__ amax) {
    // pi / po: row pitch of the input / output planes (floats; = W for plain tensors).  po > W: the library's padded
    // workspace layout -- columns W .. po-1 are written as zeros (the 3x3 convs that read them rely on it)
    __shared__ __attribute__((aligned(16))) float tile[DW_NX][DW_HR * DW_RS];
    const int c = blockIdx.y, b = blockIdx.z;
    const int ty = blockIdx.x / groupsX, gxi = blockIdx.x - ty * groupsX;
    const int y0 = ty * DW_TH, xg0 = gxi * (DW_NX * DW_TW);
    const size_t plane = ((size_t)b * C + c) * H * po;
    const float* src = x + ((size_t)b * C + c) * H * pi;
    // stage all DW_NX halo tiles: every load is issued before the first LDS write (the loads are independent; a rolled
    // loop would wait for each one in turn and make the kernel latency-bound).  16-byte groups: a halo row is 33 groups
    // of 4 floats (columns x0-2 .. x0+129), 660 groups per tile, 3 per thread; the loads are BUFFER loads on the plane
    // (rows outside the image get an out-of-range offset: hardware zero fill), columns past the row end are masked per
    // element (they would read the next row).
    constexpr int DW_GR = DW_RS / 4;                       // groups per halo row (33)
    constexpr int DW_NG = DW_HR * DW_GR;                   // groups per tile (660)
    constexpr int DW_LD = (DW_NG + 255) / 256;             // per thread (3)
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, H * pi * 4, 0x00020000);
    f32x4 stg[DW_NX][DW_LD];
#pragma unroll
    for (int t = 0; t < DW_NX; ++t) {
        const int x0 = xg0 + t * DW_TW;
#pragma unroll
        for (int k = 0; k < DW_LD; ++k) {
            const int i = threadIdx.x + k * 256;
            const int r = i / DW_GR, g = i - r * DW_GR;
            const int gy = y0 + r - 2, gx = x0 + 4 * g - 2;
            const bool rowok = i < DW_NG && x0 < W && gy >= 0 && gy < H;
            // the leftmost group of an image (gx = -2) is loaded from column 0 and shifted by two elements
            const bool left = gx < 0;
            const int off = rowok ? (gy * pi + (left ? 0 : gx)) * 4 : 0x40000000;    // (out of range -> zero fill)
            const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, off, 0, 0));
            f32x4 v = left ? f32x4{0.f, 0.f, q[0], q[1]} : q;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (rowok && gx + e < W) ? v[e] : 0.0f;
            stg[t][k] = v;
        }
    }
#pragma unroll
    for (int t = 0; t < DW_NX; ++t)
#pragma unroll
        for (int k = 0; k < DW_LD; ++k) {
            const int i = threadIdx.x + k * 256;
            if (i < DW_NG) *reinterpret_cast<f32x4*>(&tile[t][i * 4]) = stg[t][k];
        }
    float wk[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) wk[k] = w[c * 25 + (flip ? 24 - k : k)];   // flip -> transposed conv (data grad)
    const float add = (bias ? bias[c] : 0.0f) + (cond ? cond[(size_t)b * cond_stride + c] : 0.0f);
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float mx = 0.f;
#pragma unroll
    for (int t = 0; t < DW_NX; ++t) {
        const int x0 = xg0 + t * DW_TW;
        if (x0 >= W) break;
        f32x2 o[DW_RW];
#pragma unroll
        for (int r = 0; r < DW_RW; ++r) o[r] = f32x2{add, add};
#pragma unroll
        for (int dy = 0; dy < DW_RW + 4; ++dy) {
            f32x2 v[5];
            const float* row = &tile[t][(wv * DW_RW + dy) * DW_RS + lane];
#pragma unroll
            for (int dx = 0; dx < 5; ++dx) v[dx] = f32x2{row[dx], row[dx + 64]};      // columns lane+dx and lane+64+dx
#pragma unroll
            for (int r = 0; r < DW_RW; ++r) {
                const int ky = dy - r;
                if (ky >= 0 && ky < 5) {
#pragma unroll
                    for (int dx = 0; dx < 5; ++dx) o[r] += wk[ky * 5 + dx] * v[dx];
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int gx = x0 + lane + 64 * h;
            if (gx < po) {
#pragma unroll
                for (int r = 0; r < DW_RW; ++r) {
                    const int gy = y0 + wv * DW_RW + r;
                    if (gy < H) {
                        const size_t oidx = plane + (size_t)gy * po + gx;
                        const float ov = gx < W ? (addt ? o[r][h] + addt[oidx] : o[r][h]) : 0.0f;
                        mx = fmaxf(mx, fabsf(ov));
                        out[oidx] = ov;
                    }
                }
            }
        }
    }
    if (amax) amax_publish(mx, amax + (size_t)b * AMAX_STRIDE);
}

// Register-window variant for rows that are a multiple of 4 pixels (16-byte aligned): no LDS, no barrier.  A lane owns 4
// consecutive output columns of DWR_ROWS output rows and slides over the DWR_ROWS + 4 input rows: per input row three
// aligned 16-byte loads (columns x-4 .. x+7, eight of them used) feed the five output rows it touches (100 FMAs); the 25
// taps are wave-uniform (scalar registers).  Every load of a lane is independent of its FMAs, so many rows are in flight
// per wave and the only dependence on other lanes is the L1 (the halo columns of the neighbours).
#ifndef DWR_ROWS_
#define DWR_ROWS_ 12
#endif
constexpr int DWR_ROWS = DWR_ROWS_;

__global__ __launch_bounds__(256) void dwconv5_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, const float* __restrict__ cond,
                                                            int cond_stride, const float* __restrict__ addt, int flip,
                                                            float* __restrict__ out, int C, int H, int W, int bandsX, int Wt,
                                                            float* __restrict__ amax) {
    // Wt: true width when the rows are padded to W (input pads hold zeros, output pads are written as zeros); else = W
    const int c = blockIdx.y, b = blockIdx.z;
    const int by = blockIdx.x / bandsX, bx = blockIdx.x - by * bandsX;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x4 = (bx * 64 + lane) * 4;                     // first output column of this lane
    const int y0 = (by * 4 + wv) * DWR_ROWS;                 // first output row of this wave
    const size_t plane = ((size_t)b * C + c) * H * W;
    float wk[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) wk[k] = w[c * 25 + (flip ? 24 - k : k)];      // (wave-uniform: scalar loads)
    const float add = (bias ? bias[c] : 0.0f) + (cond ? cond[(size_t)b * cond_stride + c] : 0.0f);
    if (y0 >= H) return;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + plane), 0, H * W * 4, 0x00020000);
    constexpr int OOBI = 0x40000000;
    const bool colok = x4 < W;
    // quads left / right of the lane's own: outside the row -> zero (W % 4 == 0: a quad is entirely in or out)
    const bool lok = colok && x4 >= 4, rok = colok && x4 + 4 < W;
    f32x4 acc[DWR_ROWS];
#pragma unroll
    for (int r = 0; r < DWR_ROWS; ++r) acc[r] = f32x4{add, add, add, add};
#pragma unroll
    for (int ir = 0; ir < DWR_ROWS + 4; ++ir) {
        const int gy = y0 + ir - 2;
        const bool rowok = gy >= 0 && gy < H;
        const int base = (gy * W + x4) * 4;
        const f32x4 ql = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, (rowok && lok) ? base - 16 : OOBI, 0, 0));
        const f32x4 qm = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, (rowok && colok) ? base : OOBI, 0, 0));
        const f32x4 qr = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, (rowok && rok) ? base + 16 : OOBI, 0, 0));
        const float in[8] = {ql[2], ql[3], qm[0], qm[1], qm[2], qm[3], qr[0], qr[1]};     // columns x4-2 .. x4+5
#pragma unroll
        for (int r = 0; r < DWR_ROWS; ++r) {
            const int ky = ir - r;                               // input row ir contributes tap row ky to output row r
            if (ky >= 0 && ky < 5) {
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) {
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) acc[r][cc] = fmaf(wk[ky * 5 + kx], in[cc + kx], acc[r][cc]);
                }
            }
        }
    }
    float mx = 0.f;
#pragma unroll
    for (int r = 0; r < DWR_ROWS; ++r) {
        const int gy = y0 + r;
        if (colok && gy < H) {
            const size_t o = plane + (size_t)gy * W + x4;
            f32x4 v = acc[r];
            if (addt) v += *reinterpret_cast<const f32x4*>(addt + o);
            if (x4 + 4 > Wt) {
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) v[cc] = x4 + cc < Wt ? v[cc] : 0.0f;
            }
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
            *reinterpret_cast<f32x4*>(out + o) = v;
        }
    }
    if (amax) amax_publish(mx, amax + (size_t)b * AMAX_STRIDE);          // (every lane of a live wave gets here: no divergent exit above)
}

#ifndef SINDDM_DW_ROWS
#define SINDDM_DW_ROWS 1
#endif

// pi / po: row pitch of the input / output planes (0 = W).  po > W: padded workspace rows, pads written as zeros.
int dwconv_launch(const float* x, const float* w, const float* bias, const float* cond, int cond_stride,
                  const float* addt, int flip, float* out, int B, int C, int H, int W, hipStream_t st, int pi, int po,
                  float* amax) {
    if (pi <= 0) pi = W;
    if (po <= 0) po = W;
    if (SINDDM_DW_ROWS && pi % 4 == 0 && po == pi && pi >= 192) {   // (a lane owns 4 columns: narrow images leave most of a wave idle)
        const int bandsX = (pi + 255) / 256, bandsY = (H + 4 * DWR_ROWS - 1) / (4 * DWR_ROWS);
        hipLaunchKernelGGL(dwconv5_rows_kernel, dim3(bandsX * bandsY, C, B), dim3(256), 0, st, x, w, bias, cond, cond_stride,
                           addt, flip, out, C, H, pi, bandsX, W, amax);
        SINDDM_LAUNCH_CHECK();
        return 0;
    }
    const int tilesX = (W + DW_TW - 1) / DW_TW, tilesY = (H + DW_TH - 1) / DW_TH;
    const int groupsX = (tilesX + DW_NX - 1) / DW_NX;
    hipLaunchKernelGGL(dwconv5_kernel, dim3(groupsX * tilesY, C, B), dim3(256), 0, st, x, w, bias, cond, cond_stride,
                       addt, flip, out, C, H, W, groupsX, pi, po, amax);
    SINDDM_LAUNCH_CHECK();
    return 0;
}

// =====================================================================================
// first 3x3 conv of the network (C_in = 3 -> C_out, + bias + GELU)      reference SinDDM/models.py:63-64 for l1
// 27 MACs per output: nothing for the matrix cores to amortise (the implicit-GEMM kernel padded K from 27 to 72 and
// was store-issue bound at 1.6 TB/s).  VALU kernel: a thread owns one pixel, keeps its 3x3x3 patch in registers
// and walks the output channels with the weights as SCALAR operands (the channel index is wave-uniform, so the
// compiler fetches them with s_load through the scalar cache); every store is 64 consecutive floats of one channel.
// Write-bound: 4*C_out bytes per pixel.  (Two pixels per thread with packed FMAs measured 19 % slower.)
// =====================================================================================
__global__ __launch_bounds__(256) void conv3x3_c3_gelu_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              float* __restrict__ out_pre, int H, int W, int Cout,
                                                              int co_per_block, int Wt, float* __restrict__ amax) {
    // Wt: true width when the rows are padded to W (pads: zeros in, zeros out); else = W
    const int HW = H * W;
    const int b = blockIdx.y;
    // blockIdx.z owns output channels [co0, co1): small images split the channel walk over several workgroups (one
    // thread's serial 80 x (27 FMA + GELU) is 39 us of pure latency, whatever the image size)
    const int co0 = blockIdx.z * co_per_block;
    const int co1 = min(Cout, co0 + co_per_block);
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool live = p < HW;
    const int pc = live ? p : HW - 1;
    const int y = pc / W, x = pc - y * W;
    const float* src = in + (size_t)b * 3 * HW;
    float v[27];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int gy = y + ky - 1, gx = x + kx - 1;
                const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
                v[ci * 9 + ky * 3 + kx] = ok ? src[(size_t)ci * HW + (size_t)gy * W + gx] : 0.0f;
            }
    float* dst = out + (size_t)b * Cout * HW + pc;
    float* dpre = out_pre ? out_pre + (size_t)b * Cout * HW + pc : nullptr;
    // taps in pairs: (v[2j], v[2j+1]) * (w[2j], w[2j+1]) is ONE v_pk_fma_f32 with the weight pair as a scalar operand --
    // 13 packed + 1 plain FMA + 1 add instead of 27 FMAs per output; the kernel is VALU-bound (27 taps + GELU per output)
    f32x2 vp[13];
#pragma unroll
    for (int j = 0; j < 13; ++j) vp[j] = f32x2{v[2 * j], v[2 * j + 1]};
    auto dot27 = [&](int co) __attribute__((always_inline)) -> float {
        const float* wc = w + co * 27;          // wave-uniform -> scalar loads
        f32x2 acc2{bias[co], 0.f};
#pragma unroll
        for (int j = 0; j < 13; ++j) {
            const f32x2 wp{wc[2 * j], wc[2 * j + 1]};
            acc2 = __builtin_elementwise_fma(vp[j], wp, acc2);
        }
        return fmaf(v[26], wc[26], acc2.x + acc2.y);
    };
    // (two output channels per iteration: their GELUs run as one packed sequence)
    int co = co0;
    float mx = 0.f;
#pragma unroll 2
    for (; co + 1 < co1; co += 2) {
        const f32x2 acc{dot27(co), dot27(co + 1)};
        if (live) {
            if (dpre) {
                dpre[(size_t)co * HW] = acc.x;
                dpre[(size_t)(co + 1) * HW] = acc.y;
            }
            const f32x2 g = gelu_erf2(acc);
            const float g0 = x < Wt ? g.x : 0.0f, g1 = x < Wt ? g.y : 0.0f;
            mx = fmaxf(mx, fmaxf(fabsf(g0), fabsf(g1)));
            dst[(size_t)co * HW] = g0;
            dst[(size_t)(co + 1) * HW] = g1;
        }
    }
    if (co < co1) {
        const float acc = dot27(co);
        if (live) {
            if (dpre) dpre[(size_t)co * HW] = acc;
            const float g0 = x < Wt ? gelu_erf(acc) : 0.0f;
            mx = fmaxf(mx, fabsf(g0));
            dst[(size_t)co * HW] = g0;
        }
    }
    if (amax) amax_publish(mx, amax + (size_t)b * AMAX_STRIDE);
}

// =====================================================================================
// final 1x1 conv (half -> 3)                      reference SinDDM/models.py:130-132,151
// =====================================================================================
// Padded workspace rows (inference, W % 4 != 0): the library keeps its own activations with a row pitch Wp = W rounded up
// to 4 floats, pad columns zero, so that every scale runs the aligned kernels (16-byte accesses, no edge masks in the
// matrix loops).  The boundary tensors stay plain: the network input is copied into a padded 3-channel buffer ...
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                       int Wp, long long nrows) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;       // one padded quad per thread
    const int qpr = Wp >> 2;
    if (q >= nrows * qpr) return;
    const long long row = q / qpr;
    const int x = (int)(q - row * qpr) * 4;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = x + e < W ? in[row * W + x + e] : 0.0f;
    *reinterpret_cast<f32x4*>(out + row * Wp + x) = v;
}
// ... and the final 1x1 conv reads padded rows and writes the plain (B, 3, H, W) result
__global__ __launch_bounds__(256) void final_conv1x1_pitch_kernel(const float* __restrict__ a, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, float* __restrict__ out,
                                                                  int C, int H, int W, int Wp) {
    const int b = blockIdx.y;
    const int qpr = Wp >> 2;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= H * qpr) return;
    const int y = q / qpr, x = (q - y * qpr) * 4;
    const size_t HWp = (size_t)H * Wp;
    const float* src = a + (size_t)b * C * HWp + (size_t)y * Wp + x;
    f32x4 o0{bias[0], bias[0], bias[0], bias[0]}, o1{bias[1], bias[1], bias[1], bias[1]}, o2{bias[2], bias[2], bias[2], bias[2]};
#pragma unroll 8
    for (int c = 0; c < C; ++c) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)c * HWp);
        o0 += w[c] * v;
        o1 += w[C + c] * v;
        o2 += w[2 * C + c] * v;
    }
    const size_t HW = (size_t)H * W;
    float* dst = out + (size_t)b * 3 * HW + (size_t)y * W + x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (x + e < W) {
            dst[e] = o0[e];
            dst[HW + e] = o1[e];
            dst[2 * HW + e] = o2[e];
        }
    }
}
__global__ __launch_bounds__(256) void final_conv1x1_kernel(const float* __restrict__ a, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int C, int HW) {
    // 4 consecutive pixels per thread: 16-byte loads (4-byte aligned is enough for global memory), 8 channels in flight
    const int b = blockIdx.y;
    const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (p >= HW) return;
    const float* src = a + (size_t)b * C * HW + p;
    float* dst = out + (size_t)b * 3 * HW + p;
    if (p + 3 < HW) {
        f32x4 o0{bias[0], bias[0], bias[0], bias[0]}, o1{bias[1], bias[1], bias[1], bias[1]},
            o2{bias[2], bias[2], bias[2], bias[2]};
#pragma unroll 8
        for (int c = 0; c < C; ++c) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)c * HW);
            o0 += w[c] * v;
            o1 += w[C + c] * v;
            o2 += w[2 * C + c] * v;
        }
        *reinterpret_cast<f32x4*>(dst) = o0;
        *reinterpret_cast<f32x4*>(dst + HW) = o1;
        *reinterpret_cast<f32x4*>(dst + 2 * (size_t)HW) = o2;
    } else {
        for (int j = 0; p + j < HW; ++j) {
            float o0 = bias[0], o1 = bias[1], o2 = bias[2];
            for (int c = 0; c < C; ++c) {
                const float v = src[(size_t)c * HW + j];
                o0 = fmaf(w[c], v, o0);
                o1 = fmaf(w[C + c], v, o1);
                o2 = fmaf(w[2 * C + c], v, o2);
            }
            dst[j] = o0;
            dst[HW + j] = o1;
            dst[2 * (size_t)HW + j] = o2;
        }
    }
}

// =====================================================================================
// diffusion elementwise kernels (HBM-bound)
// =====================================================================================
// VEC = 4: 16-byte accesses (n % 4 == 0 keeps every sample's base 16-byte aligned), UN quads per thread requested before
// the first one is used (4 for big tensors, 1 when that would leave CUs without workgroups); VEC = 1: any n.  The per-sample coefficients are three scalar loads per workgroup.
template <int VEC, int UN>
__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ xo,
                                                       const float* __restrict__ nz, float* __restrict__ out,
                                                       const float* __restrict__ tabA, const float* __restrict__ tabB,
                                                       const float* __restrict__ gam, const long long* __restrict__ t_dev,
                                                       int t_host, long long n) {
    const int b = blockIdx.y;
    const long long t = t_dev ? t_dev[b] : (long long)t_host;
    const float ca = tabA[t], cb = tabB[t];
    const float g = (xo != nullptr) ? gam[t] : 0.0f;
    const size_t base = (size_t)b * n;
    if constexpr (VEC == 4) {
        const long long nq = n >> 2;
        const f32x4* x4 = reinterpret_cast<const f32x4*>(x0 + base);
        const f32x4* o4 = xo ? reinterpret_cast<const f32x4*>(xo + base) : nullptr;
        const f32x4* z4 = reinterpret_cast<const f32x4*>(nz + base);
        f32x4* y4 = reinterpret_cast<f32x4*>(out + base);
        for (long long q0 = (long long)blockIdx.x * (256 * UN) + threadIdx.x; q0 < nq; q0 += (long long)gridDim.x * (256 * UN)) {
            f32x4 v[UN], z[UN], o[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const long long q = q0 + u * 256;
                const bool ok = q < nq;
                v[u] = ok ? __builtin_nontemporal_load(x4 + q) : f32x4{0.f, 0.f, 0.f, 0.f};
                z[u] = ok ? __builtin_nontemporal_load(z4 + q) : f32x4{0.f, 0.f, 0.f, 0.f};
                if (o4) o[u] = ok ? __builtin_nontemporal_load(o4 + q) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const long long q = q0 + u * 256;
                if (q >= nq) continue;
                f32x4 w = v[u];
                if (o4) w = g * w + (1.0f - g) * o[u];       // models.py:584-585
                y4[q] = ca * w + cb * z[u];                   // models.py:574-575
            }
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
            float v = x0[base + i];
            if (xo) v = g * v + (1.0f - g) * xo[base + i];   // models.py:584-585
            out[base + i] = ca * v + cb * nz[base + i];      // models.py:574-575
        }
    }
}

// EDIT: the predicted clean image is replaced by  w(p) * x_recon + c(ch, p)  before the re-blur mix / clamps --
// the ROI-guided sampling of the reference (models.py:291-298,430-431) written as a per-pixel affine map
// (sequential `eta*patch + (1-eta)*x` blends over possibly overlapping boxes compose into one such map).
// x_{t-1} mean of one element: predict_start_from_noise + p_mean_variance (normal branch) + q_posterior
// (reference SinDDM/models.py:306-352,433-447); `w`, `c` = ROI edit map (1, 0 without ROI guidance)
__device__ __forceinline__ float reverse_step_mean(const sinddm_step_coefs& k, float x, float e, float xb, float w, float c,
                                                   bool edit) {
    float x0 = k.sqrt_recip_ac_t * x - k.sqrt_recipm1_ac_t * e;                   // models.py:308-309
    if (k.mode == 0) {
        if (edit) x0 = w * x0 + c;              // x_recon and x_t_mix are the same tensor here (models.py:311-312)
        const float x0c = k.clip ? fminf(fmaxf(x0, -1.0f), 1.0f) : x0;
        return k.coef1_t * x0c + k.coef2_t * x;                                   // models.py:324-327
    }
    float xp = (x0 - k.gamma_t * xb) / (1.0f - k.gamma_t);                        // models.py:315-316
    if (edit) xp = w * xp + c;
    if (k.mode == 1) {
        float mix = k.gamma_tm1 * xb + (1.0f - k.gamma_tm1) * xp;                 // models.py:435-436
        float x0c = x0;
        if (k.clip) {
            mix = fminf(fmaxf(mix, -1.0f), 1.0f);
            x0c = fminf(fmaxf(x0, -1.0f), 1.0f);
        }
        return k.sqrt_ac_tm1 * mix + k.sqrt_1m_ac_tm1_mvar * (x - k.sqrt_ac_t * x0c) / k.sqrt_1m_ac_t;  // :342-345
    }
    return k.clip ? fminf(fmaxf(xp, -1.0f), 1.0f) : xp;                           // models.py:347-348
}

// EDIT: the predicted clean image is replaced by  w(p) * x_recon + c(ch, p)  before the re-blur mix / clamps --
// the ROI-guided sampling of the reference (models.py:291-298,430-431) written as a per-pixel affine map
// (sequential `eta*patch + (1-eta)*x` blends over possibly overlapping boxes compose into one such map).
template <bool EDIT>
__global__ __launch_bounds__(256) void reverse_step_kernel(const float* __restrict__ xt, const float* __restrict__ eps,
                                                           const float* __restrict__ xtil, const float* __restrict__ z,
                                                           float* __restrict__ out, sinddm_step_coefs k, long long n,
                                                           const float* __restrict__ ew, const float* __restrict__ ec,
                                                           int chw, int hw) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float w = 1.0f, c = 0.0f;
        if (EDIT) {
            const int q = (int)(i % chw);
            w = ew[q % hw];
            c = ec[q];
        }
        const float mean = reverse_step_mean(k, xt[i], eps[i], k.mode != 0 ? xtil[i] : 0.0f, w, c, EDIT);
        out[i] = mean + k.sigma * z[i];                                           // models.py:459
    }
}

// ---- the same step with the Gaussian noise of models.py:455 drawn INSIDE the kernel (counter-based Philox4x32-10 +
// Box-Muller): no randn launch, no noise tensor (12 B/px less traffic).  Stream = (seed, step id, element index); the
// reference never seeds its generator, so there is no bit-level noise contract -- parity tests keep injecting noise
// through sinddm_reverse_step.
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        c[0] = hi1 ^ c[1] ^ k0; c[1] = lo1; c[2] = hi0 ^ c[3] ^ k1; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__device__ __forceinline__ void philox_normal4(unsigned long long seed, unsigned long long step, unsigned long long idx4,
                                               float (&z)[4]) {
    unsigned c[4] = {(unsigned)idx4, (unsigned)(idx4 >> 32), (unsigned)step, (unsigned)(step >> 32)};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
    // uniforms in (0, 1]: (x + 1) * 2^-32 evaluated so that 0 is never produced
    const float u0 = ((float)(c[0] >> 8) + 1.0f) * (1.0f / 16777216.0f), u1 = (float)(c[1] >> 8) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c[2] >> 8) + 1.0f) * (1.0f / 16777216.0f), u3 = (float)(c[3] >> 8) * (1.0f / 16777216.0f);
    const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
    float s0, c0, s1, c1;
    sincospif(2.0f * u1, &s0, &c0);
    sincospif(2.0f * u3, &s1, &c1);
    z[0] = r0 * c0; z[1] = r0 * s0; z[2] = r1 * c1; z[3] = r1 * s1;
}

__global__ __launch_bounds__(256) void reverse_step_rng_kernel(const float* __restrict__ xt, const float* __restrict__ eps,
                                                               const float* __restrict__ xtil, float* __restrict__ out,
                                                               sinddm_step_coefs k, long long n, unsigned long long seed,
                                                               unsigned long long step) {
    const long long n4 = (n + 3) >> 2;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long long)gridDim.x * 256) {
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        if (k.sigma != 0.0f) philox_normal4(seed, step, (unsigned long long)q, z);
        const long long i0 = q << 2;
        if (i0 + 3 < n) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(xt + i0);
            const f32x4 e = *reinterpret_cast<const f32x4*>(eps + i0);
            f32x4 xb{0.f, 0.f, 0.f, 0.f};
            if (k.mode != 0) xb = *reinterpret_cast<const f32x4*>(xtil + i0);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = reverse_step_mean(k, x[j], e[j], xb[j], 1.f, 0.f, false) + k.sigma * z[j];
            *reinterpret_cast<f32x4*>(out + i0) = o;
        } else {
            for (int j = 0; i0 + j < n; ++j)
                out[i0 + j] = reverse_step_mean(k, xt[i0 + j], eps[i0 + j], k.mode != 0 ? xtil[i0 + j] : 0.f, 1.f, 0.f, false) +
                              k.sigma * z[j];
        }
    }
}

// final 1x1 conv (-> eps) + reverse step + in-kernel noise in one pass (sampler runs; H*W % 4 == 0 so that a thread's
// four pixels are one quad of the flat [B][3][H][W] index the generator is keyed on -- same numbers as the two-kernel
// path): eps never goes to memory.
__global__ __launch_bounds__(256) void final_conv_reverse_step_kernel(const float* __restrict__ a, const float* __restrict__ w,
                                                                      const float* __restrict__ bias,
                                                                      const float* __restrict__ xt,
                                                                      const float* __restrict__ xtil, float* __restrict__ out,
                                                                      sinddm_step_coefs k, int C, int HW,
                                                                      unsigned long long seed, unsigned long long step, int b0) {
    const int b = blockIdx.y;
    const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (p >= HW) return;
    const float* src = a + (size_t)b * C * HW + p;
    f32x4 e[3] = {{bias[0], bias[0], bias[0], bias[0]}, {bias[1], bias[1], bias[1], bias[1]}, {bias[2], bias[2], bias[2], bias[2]}};
#pragma unroll 8
    for (int c = 0; c < C; ++c) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)c * HW);
        e[0] += w[c] * v;
        e[1] += w[C + c] * v;
        e[2] += w[2 * C + c] * v;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const long long i0 = ((long long)b * 3 + c) * HW + p;
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        if (k.sigma != 0.0f) philox_normal4(seed, step, (unsigned long long)((i0 + (long long)b0 * 3 * HW) >> 2), z);
        const f32x4 x = *reinterpret_cast<const f32x4*>(xt + i0);
        f32x4 xb{0.f, 0.f, 0.f, 0.f};
        if (k.mode != 0) xb = *reinterpret_cast<const f32x4*>(xtil + i0);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = reverse_step_mean(k, x[j], e[c][j], xb[j], 1.f, 0.f, false) + k.sigma * z[j];
        *reinterpret_cast<f32x4*>(out + i0) = o;
    }
}

// standalone N(0,1) fill from the same generator (tests; initial / re-noise draws of the sampler)
// the same for padded workspace rows (pitch Wp, true width W): a thread owns a padded quad of a row; the boundary tensors
// (x_t, x-tilde, x_{t-1}) are plain, so its up to four pixels sit at an unaligned flat index and their N(0,1) draws -- keyed
// on the FLAT quad index like everywhere else -- come from up to two Philox calls
__global__ __launch_bounds__(256) void final_conv_reverse_step_pitch_kernel(
    const float* __restrict__ a, const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ xt,
    const float* __restrict__ xtil, float* __restrict__ out, sinddm_step_coefs k, int C, int H, int W, int Wp,
    unsigned long long seed, unsigned long long step, int b0) {
    const int b = blockIdx.y;
    const int qpr = Wp >> 2;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= H * qpr) return;
    const int y = q / qpr, x = (q - y * qpr) * 4;
    const size_t HWp = (size_t)H * Wp;
    const long long HW = (long long)H * W;
    const float* src = a + (size_t)b * C * HWp + (size_t)y * Wp + x;
    f32x4 e[3] = {{bias[0], bias[0], bias[0], bias[0]}, {bias[1], bias[1], bias[1], bias[1]}, {bias[2], bias[2], bias[2], bias[2]}};
#pragma unroll 8
    for (int c = 0; c < C; ++c) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)c * HWp);
        e[0] += w[c] * v;
        e[1] += w[C + c] * v;
        e[2] += w[2 * C + c] * v;
    }
    const int nv = W - x;                                   // valid pixels of the quad (>= 1)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const long long i0 = ((long long)b * 3 + c) * HW + (long long)y * W + x;
        const long long ig = i0 + (long long)b0 * 3 * HW;          // flat index inside the whole batch: the noise key
        const int r0 = (int)(ig & 3);
        float za[4] = {0.f, 0.f, 0.f, 0.f}, zb[4] = {0.f, 0.f, 0.f, 0.f};
        if (k.sigma != 0.0f) {
            philox_normal4(seed, step, (unsigned long long)(ig >> 2), za);
            if (r0 != 0) philox_normal4(seed, step, (unsigned long long)(ig >> 2) + 1ull, zb);
        }
        const float z8[8] = {za[0], za[1], za[2], za[3], zb[0], zb[1], zb[2], zb[3]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < nv) {
                float z = z8[j];                                                   // z8[r0 + j], r0 in 0..3
                z = r0 == 1 ? z8[j + 1] : z;
                z = r0 == 2 ? z8[j + 2] : z;
                z = r0 == 3 ? z8[j + 3] : z;
                const float xb = k.mode != 0 ? xtil[i0 + j] : 0.f;
                out[i0 + j] = reverse_step_mean(k, xt[i0 + j], e[c][j], xb, 1.f, 0.f, false) + k.sigma * z;
            }
        }
    }
}
__global__ __launch_bounds__(256) void philox_normal_kernel(float* __restrict__ out, long long n, unsigned long long seed,
                                                            unsigned long long step) {
    const long long n4 = (n + 3) >> 2;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long long)gridDim.x * 256) {
        float z[4];
        philox_normal4(seed, step, (unsigned long long)q, z);
        for (int j = 0; j < 4 && (q << 2) + j < n; ++j) out[(q << 2) + j] = z[j];
    }
}

__global__ __launch_bounds__(256) void upsample_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                int h, int w, int H, int W, float sy, float sx) {
    const int bc = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const int oy = p / W, ox = p - oy * W;
    // ATen area_pixel_compute_source_index(align_corners=False): max(scale*(dst+0.5)-0.5, 0), fp32
    // explicitly ONE rounding (fma), which is what ATen's compiled CPU/GPU kernels do: at source coordinates of
    // several hundred a two-rounding evaluation moves the interpolation weight by up to an ulp of the coordinate
    // (3e-5 abs on the 364x1092 scale of config C5; fixture g8:hash_8x776_to_11x1092)
    float fy = fmaf(sy, (float)oy + 0.5f, -0.5f);
    fy = fy < 0.0f ? 0.0f : fy;
    float fx = fmaf(sx, (float)ox + 0.5f, -0.5f);
    fx = fx < 0.0f ? 0.0f : fx;
    int iy0 = (int)fy; if (iy0 > h - 1) iy0 = h - 1;
    int ix0 = (int)fx; if (ix0 > w - 1) ix0 = w - 1;
    const int iy1 = iy0 + 1 < h ? iy0 + 1 : h - 1;
    const int ix1 = ix0 + 1 < w ? ix0 + 1 : w - 1;
    const float ly = fy - (float)iy0, lx = fx - (float)ix0;
    const float* s = in + (size_t)bc * h * w;
    const float tl = s[iy0 * w + ix0], tr = s[iy0 * w + ix1], bl = s[iy1 * w + ix0], br = s[iy1 * w + ix1];
    out[(size_t)bc * H * W + p] = (1.0f - ly) * ((1.0f - lx) * tl + lx * tr) + ly * ((1.0f - lx) * bl + lx * br);
}

// =====================================================================================
// whole-network forward orchestration
// =====================================================================================
struct FwdBuffers {
    float* cond;    // [B][cond_stride]
    float* buf[4];  // each B*dim*H*W
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// conditioning rows: one per sample (sinddm_net_forward) or one per sampler step of a run (sinddm_sample_chain: every
// sample of the batch shares the step's t, so a whole run of steps is embedded by ONE cond_kernel launch)
constexpr int CHAIN_COND_ROWS = 1024;
// ... followed by the running-max scalars of one network evaluation (split16.h: [sample][AMAX_STRIDE], slot 2 l + i = max
// |input| of conv i of block l, maintained by that tensor's producer kernel, zeroed at the start of the evaluation)
static size_t amax_region_bytes(int B) { return ((size_t)B * AMAX_STRIDE * sizeof(float) + 255) / 256 * 256; }
static size_t cond_region_bytes(const NetPlan& P, int B) {
    const int rows = B > CHAIN_COND_ROWS ? B : CHAIN_COND_ROWS;
    return align_up((size_t)rows * P.cond_stride * sizeof(float), 256) + amax_region_bytes(B);
}
#ifndef SINDDM_PITCH
#define SINDDM_PITCH 1        // 1: inference keeps its activations with rows padded to a multiple of 4 floats (W % 4 != 0 scales)
#endif
// row pitch of the library's own activation buffers in inference: W rounded up to 4 floats
// Padded rows need every producer to write the pad columns as zeros (ConvArgs::Wt): the depthwise kernels, the C_in = 3
// conv, the Winograd kernels of conv_wino2/3/4.h and conv_wh do; the direct implicit-GEMM kernel (conv_mfma.h) and the
// first-generation Winograd kernel do not.  So the pitch is a property of the PLAN: rows are padded only when
// block_forward routes every 3x3 conv of the network to a Wt-aware kernel (all channel counts multiples of 4, C_in = 3 or
// >= 8) -- other widths (--dim 10, 20, 28 ...) keep plain rows and the kernels' edge variants.
static bool plan_pads_rows(const NetPlan& P) {
    if (!SINDDM_PITCH || !wino_enabled() || !SINDDM_WINO_V2 || !SINDDM_CONV_C3) return false;
    for (int l = 0; l < 4; ++l) {
        const BlockPlan& b = P.blk[l];
        if (b.cout % 4 != 0) return false;
        if (b.cin != 3 && !(b.pk_wc1 >= 0 && b.cin % 4 == 0)) return false;
    }
    return true;
}
static int fwd_pitch(const NetPlan& P, int W) { return (W % 4 != 0 && plan_pads_rows(P)) ? (W + 3) / 4 * 4 : W; }
static size_t fwd_xpad_bytes(const NetPlan& P, int B, int H, int W) {        // padded copy of the network input (only when the pitch differs)
    return fwd_pitch(P, W) != W ? align_up((size_t)B * CHANNELS * H * fwd_pitch(P, W) * sizeof(float), 256) : 0;
}
// what one network evaluation of batch B carves from its workspace: conditioning rows, 4 activation buffers, padded input
static size_t fwd_workspace_core(const NetPlan& P, int B, int H, int W) {
    const size_t act = align_up((size_t)B * P.dim * H * fwd_pitch(P, W) * sizeof(float), 256);
    return cond_region_bytes(P, B) + 4 * act + fwd_xpad_bytes(P, B, H, W);
}
// ... and what sinddm_workspace_bytes reports: room for a sampler run that splits the batch into two halves on two
// streams (sinddm_sample_chain2): the shared conditioning table + two cores of half the batch
static size_t fwd_workspace_bytes(const NetPlan& P, int B, int H, int W) {
    return fwd_workspace_core(P, B, H, W) + 2 * cond_region_bytes(P, (B + 1) / 2) + 4096;
}

// sampler-run extras of net_forward_impl: the step's conditioning row (already computed, shared by the batch) and the
// fused tail (final conv + reverse step)
struct ChainStep {
    const float* cond_row;
    const float* x_tilde;
    float* x_next;
    sinddm_step_coefs coefs;
    unsigned long long seed, stream_id;
    int b0;            // index of this call's first sample inside the whole batch (the noise is keyed on the whole batch's flat index)
};

int conv3x3_path(int cout, int cin, int coblks, int B, int H, int W) {
    if (!wino_enabled() || cout % 4 != 0) return 0;
    const bool f24 = mt_for(cout) == 5 && cout % 80 == 0 && cin >= 16 && cin % 16 == 0;
    const bool v3 = SINDDM_WINO_V3 && f24 &&
                    (long long)B * ((W + 31) / 32) * ((H + 3) / 4) * coblks >= SINDDM_V3_MIN_ITEMS_PER_CU * wino2_cu_count();
    if (v3 && SINDDM_WINO_V4 && conv_wino4_applies(B, H, W, coblks)) return 4;
    return v3 ? 3 : 2;
}

// One SinDDMConvBlock forward (reference SinDDM/models.py:69-80): depthwise 5x5 + per-sample condition -> hbuf, 3x3 conv +
// GELU -> gbuf (pre-activation -> upre when the training forward saves it), 3x3 conv + residual (1x1 projection or
// identity of `cur`) -> obuf.  `cond` = the block's per-sample bias rows ([B][cond_stride]; stride 0: one row for the batch).
// Wt > 0: padded workspace rows -- every tensor here (cur included) has row pitch W (a multiple of 4) and true width Wt.
int block_forward(const NetPlan& P, int l, const float* params, const float* packed, const float* cur, const float* cond,
                  int cond_stride, float* hbuf, float* gbuf, float* obuf, float* upre, int B, int H, int W, hipStream_t st,
                  int Wt, float* amax) {
    const BlockPlan& b = P.blk[l];
    // the Winograd F(2x4) kernel with binary16 hi/lo frequency GEMMs (conv_wh.h) where its rule takes the launch; `amax` = this
    // block's two running-max scalars (input of conv1, input of conv2), maintained by the kernels that produce those tensors
    const bool wha = amax && b.pk_q1 >= 0 && conv_wh_applies(P.fp32_convs, B, H, W, b.cin, b.cout);
    const bool whb = amax && b.pk_q2 >= 0 && conv_wh_applies(P.fp32_convs, B, H, W, b.cout, b.cout);
    int rc = Wt > 0 ? dwconv_launch(cur, params + b.dw_w, params + b.dw_b, cond, cond_stride, nullptr, 0, hbuf, B, b.cin, H,
                                    Wt, st, W, W, wha ? amax : nullptr)
                    : dwconv_launch(cur, params + b.dw_w, params + b.dw_b, cond, cond_stride, nullptr, 0, hbuf, B, b.cin, H,
                                    W, st, 0, 0, wha ? amax : nullptr);
    if (rc) return rc;
    const bool wino = wino_enabled();
    ConvArgs c1{};
    c1.in = hbuf; c1.bias = packed + b.pk_b1; c1.out = gbuf;
    c1.B = B; c1.H = H; c1.W = W; c1.Cin = b.cin; c1.Cout = b.cout; c1.nch1 = 0;
    c1.coblks = b.coblks; c1.act = 1; c1.zero = packed + P.pk_zero; c1.Wt = Wt;
    c1.out_pre = upre;
    constexpr int c3 = SINDDM_CONV_C3;
    // launches with enough work for every workgroup slot take the F(2x4) kernel (25 % fewer MFMAs)
    const bool v3 = SINDDM_WINO_V3 && wino &&
                    (long long)B * ((W + 31) / 32) * ((H + 3) / 4) * b.coblks >= SINDDM_V3_MIN_ITEMS_PER_CU * wino2_cu_count();
    // ... and the ones with several 8x32 items per CU its one-wave-per-SIMD form (weights shared by two n-tiles)
    const bool v4 = SINDDM_WINO_V4 && v3 && conv_wino4_applies(B, H, W, b.coblks);
    if (wha) {
        c1.w3 = packed + b.pk_q1; c1.wsinv = packed + b.pk_qs1; c1.amax_in = amax; c1.amax_out = whb ? amax + 1 : nullptr;
        c1.bias = params + b.c1_b;
        rc = conv_wh_launch(c1, st);
    } else if (v3 && b.pk_w1f >= 0) {
        c1.w3 = packed + b.pk_w1f; c1.nch3 = b.nchw1;
        rc = v4 ? conv_wino4_launch(c1, st) : conv_wino3_launch(c1, st);
    } else if (wino && b.pk_wc1 >= 0 && b.cin % 4 == 0) {
        c1.w3 = packed + b.pk_wc1; c1.nch3 = b.nchw1;
        rc = conv_wino_launch(c1, b.mt, st);
    } else if (b.cin == 3 && c3) {
        // C_in = 3: dedicated VALU kernel straight from the PyTorch-layout weights
        const int nwg = ((H * W + 255) / 256) * B;
        int split = 1;                                       // channel groups: aim at >= ~2048 workgroups
        while (split < 8 && nwg * split < 2048 && b.cout % (split * 2) == 0) split *= 2;
        hipLaunchKernelGGL(conv3x3_c3_gelu_kernel, dim3((H * W + 255) / 256, B, split), dim3(256), 0, st, hbuf,
                           params + b.c1_w, params + b.c1_b, gbuf, c1.out_pre, H, W, b.cout, b.cout / split, Wt > 0 ? Wt : W,
                           whb ? amax + 1 : nullptr);
        SINDDM_LAUNCH_CHECK();
        rc = 0;
    } else {
        c1.w3 = packed + b.pk_c1; c1.nch3 = b.nch1;
        rc = conv_launch(c1, b.mt, st);
    }
    if (rc) return rc;
    // conv2 on the binary16 kernel reads the running max of g (slot amax + 1): conv_wh / the C_in = 3 kernel publish it
    // from their epilogues, the fp32 kernels do not (dim = 80 / 240: block 1 has C_in = dim / 2, not a multiple of 16, so conv1
    // stays on an fp32 kernel while conv2 qualifies) -- one pass over g then
    if (whb && !(wha || (b.cin == 3 && c3 && !(v3 && b.pk_w1f >= 0) && !(wino && b.pk_wc1 >= 0 && b.cin % 4 == 0)))) {
        rc = amax_tensor_launch(gbuf, amax + 1, B, (long long)b.cout * H * W, st);
        if (rc) return rc;
    }
    ConvArgs c2{};
    c2.in = gbuf; c2.out = obuf;
    c2.B = B; c2.H = H; c2.W = W; c2.Cin = b.cout; c2.Cout = b.cout;
    c2.coblks = b.coblks; c2.act = 0; c2.zero = packed + P.pk_zero; c2.Wt = Wt;
    if (wino && b.cout % 4 == 0) {       // (C_in of conv2 = cout; % 4: see conv_wino_launch)
        // Winograd 3x3; a 1x1 residual projection runs first on the direct kernel and is added as `resid`
        if (b.nchr > 0) {
            ConvArgs r1{};
            r1.in2 = cur; r1.Cin2 = b.cin; r1.w1 = packed + b.pk_res; r1.nch1 = b.nchr; r1.nch3 = 0;
            r1.bias = packed + b.pk_b2; r1.out = obuf; r1.Cout = b.cout; r1.coblks = b.coblks;
            r1.B = B; r1.H = H; r1.W = W; r1.zero = packed + P.pk_zero;
            rc = conv1x1_launch(r1, b.mt, st);
            if (rc) return rc;
            c2.resid = obuf; c2.bias = nullptr;          // in place: each thread reads resid[o] before writing out[o]
        } else {
            c2.resid = cur; c2.bias = packed + b.pk_b2;
        }
        c2.nch1 = 0;
        if (whb) {
            c2.w3 = packed + b.pk_q2; c2.wsinv = packed + b.pk_qs2; c2.amax_in = amax + 1;
            rc = conv_wh_launch(c2, st);
        } else if (v3 && b.pk_w2f >= 0) {
            c2.w3 = packed + b.pk_w2f; c2.nch3 = b.nchw2;
            rc = v4 ? conv_wino4_launch(c2, st) : conv_wino3_launch(c2, st);
        } else {
            c2.w3 = packed + b.pk_wc2; c2.nch3 = b.nchw2;
            rc = conv_wino_launch(c2, b.mt, st);
        }
    } else {
        c2.w3 = packed + b.pk_c2; c2.bias = packed + b.pk_b2; c2.nch3 = b.nch2;
        if (b.nchr > 0) { c2.in2 = cur; c2.Cin2 = b.cin; c2.w1 = packed + b.pk_res; c2.nch1 = b.nchr; }
        else { c2.resid = cur; c2.nch1 = 0; }
        rc = conv_launch(c2, b.mt, st);
    }
    return rc;
}

bool wh_applies(const NetPlan& P, int B, int H, int W, int cin, int cout) { return conv_wh_applies(P.fp32_convs, B, H, W, cin, cout); }
bool wh_enabled() { return SINDDM_CONV_WH != 0; }
int wh_conv(const ConvArgs& c, hipStream_t st) { return conv_wh_launch(c, st); }
int wh_pack(const float* w, float* wsinv, void* img, int cin, int cout, int transpose, hipStream_t st) {
    return wh_pack_launch(w, wsinv, img, cin, cout, transpose, st);
}

__global__ __launch_bounds__(256) void amax_tensor_kernel(const float* __restrict__ x, float* __restrict__ amax, long long n4) {
    const int b = blockIdx.y;
    const f32x4* p = reinterpret_cast<const f32x4*>(x) + (size_t)b * n4;
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 v = p[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    amax_publish(m, amax + (size_t)b * AMAX_STRIDE);
}
int amax_tensor_launch(const float* x, float* amax, int B, long long per_sample, hipStream_t st) {
    if (per_sample % 4 != 0) return SINDDM_E_BADSHAPE;
    const long long n4 = per_sample / 4;
    long long bx = (n4 + 2047) / 2048;                    // eight 16-byte loads per lane
    const long long cap = (8LL * wino2_cu_count() + B - 1) / B;
    if (bx > cap) bx = cap;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(amax_tensor_kernel, dim3((unsigned)bx, B), dim3(256), 0, st, x, amax, n4);
    SINDDM_LAUNCH_CHECK();
    return 0;
}

int net_forward_impl(const NetPlan& P, const float* params, const float* packed, const float* x, const int64_t* t_dev,
                     int t_host, float scale, float* out, int B, int H, int W, void* ws, size_t ws_bytes,
                     hipStream_t st, const TrainBufs* tb, const ChainStep* cs) {
    FwdBuffers fb{};
    float* xpad = nullptr;
    float* amax = nullptr;       // eight running-max scalars per sample (split16.h)
    if (tb) {
        fb.cond = tb->cond;
        amax = tb->amax;
    } else {
        if (ws_bytes < fwd_workspace_core(P, B, H, W)) return SINDDM_E_WORKSPACE;
        char* base = static_cast<char*>(ws);
        fb.cond = reinterpret_cast<float*>(base);
        base += cond_region_bytes(P, B);
        amax = reinterpret_cast<float*>(base - amax_region_bytes(B));
        const size_t act = align_up((size_t)B * P.dim * H * fwd_pitch(P, W) * sizeof(float), 256);
        for (int i = 0; i < 4; ++i) fb.buf[i] = reinterpret_cast<float*>(base + i * act);
        xpad = reinterpret_cast<float*>(base + 4 * act);
    }
    // padded rows: inference only (the training workspace keeps plain tensors: backward reads them with the plain kernels)
    const int Wp = tb ? W : fwd_pitch(P, W);
    const bool padded = Wp != W;
    int cond_stride = P.cond_stride;
    if (cs) { fb.cond = const_cast<float*>(cs->cond_row); cond_stride = 0; }

    CondArgs ca{};
    ca.params = params; ca.t_dev = reinterpret_cast<const long long*>(t_dev); ca.t_host = t_host; ca.scale = scale;
    ca.out = fb.cond; ca.cond_stride = P.cond_stride;
    ca.tm0_w = P.tm0_w; ca.tm0_b = P.tm0_b; ca.tm2_w = P.tm2_w; ca.tm2_b = P.tm2_b;
    for (int l = 0; l < 4; ++l) {
        ca.mlp_w[l] = P.blk[l].mlp_w; ca.mlp_b[l] = P.blk[l].mlp_b;
        ca.tr_w[l] = P.blk[l].tr_w; ca.tr_b[l] = P.blk[l].tr_b;
        ca.cin[l] = P.blk[l].cin; ca.coff[l] = P.blk[l].cond_off;
    }
    ca.cond_vec = tb ? tb->cvec : nullptr; ca.hidden = tb ? tb->hpre : nullptr;
    ca.emb_out = tb ? tb->emb : nullptr; ca.mvec_out = tb ? tb->mvec : nullptr;
    if (!cs) {
        hipLaunchKernelGGL(cond_kernel, dim3(B), dim3(128), 0, st, ca);
        SINDDM_LAUNCH_CHECK();
    }

    if (amax) {
        // the running-max scalars exist for the binary16 kernels: a launch shape none of them takes (coarse pyramid scales)
        // neither maintains nor zeroes them
        bool any = false;
        for (int l = 0; l < 4; ++l) {
            const BlockPlan& b = P.blk[l];
            any = any || (b.pk_q1 >= 0 && conv_wh_applies(P.fp32_convs, B, H, Wp, b.cin, b.cout)) ||
                  (b.pk_q2 >= 0 && conv_wh_applies(P.fp32_convs, B, H, Wp, b.cout, b.cout));
        }
        if (!any) amax = nullptr;
        else if (hipMemsetAsync(amax, 0, (size_t)B * AMAX_STRIDE * sizeof(float), st) != hipSuccess) return SINDDM_E_BADARG;
    }
    const float* cur = x;
    if (padded) {
        const long long nrows = (long long)B * CHANNELS * H;
        const long long nq = nrows * (Wp / 4);
        hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, x, xpad, H, W, Wp, nrows);
        SINDDM_LAUNCH_CHECK();
        cur = xpad;
    }
    int freeb[4] = {0, 1, 2, 3};
    int curb = -1;
    for (int l = 0; l < 4; ++l) {
        const BlockPlan& b = P.blk[l];
        // pick three scratch buffers different from the one holding `cur`
        int sel[3], n = 0;
        for (int i = 0; i < 4 && n < 3; ++i)
            if (freeb[i] != curb) sel[n++] = freeb[i];
        float* hbuf = tb ? tb->h[l] : fb.buf[sel[0]];
        float* gbuf = tb ? tb->g[l] : fb.buf[sel[1]];
        float* obuf = tb ? tb->o[l] : fb.buf[sel[2]];
        const int rc = block_forward(P, l, params, packed, cur, fb.cond + b.cond_off, cond_stride, hbuf, gbuf, obuf,
                                     tb ? tb->u[l] : nullptr, B, H, Wp, st, padded ? W : 0, amax ? amax + 2 * l : nullptr);
        if (rc) return rc;
        cur = obuf;
        curb = sel[2];
    }
    const int HW = H * W;
    if (padded) {
        const unsigned gx = (unsigned)((H * (Wp / 4) + 255) / 256);
        if (cs && cs->x_next)
            hipLaunchKernelGGL(final_conv_reverse_step_pitch_kernel, dim3(gx, B), dim3(256), 0, st, cur, params + P.fin_w,
                               params + P.fin_b, x, cs->x_tilde, cs->x_next, cs->coefs, P.half, H, W, Wp, cs->seed, cs->stream_id, cs->b0);
        else
            hipLaunchKernelGGL(final_conv1x1_pitch_kernel, dim3(gx, B), dim3(256), 0, st, cur, params + P.fin_w,
                               params + P.fin_b, out, P.half, H, W, Wp);
        SINDDM_LAUNCH_CHECK();
        return 0;
    }
    if (cs && cs->x_next && HW % 4 == 0) {
        hipLaunchKernelGGL(final_conv_reverse_step_kernel, dim3((HW / 4 + 255) / 256, B), dim3(256), 0, st, cur,
                           params + P.fin_w, params + P.fin_b, x, cs->x_tilde, cs->x_next, cs->coefs, P.half, HW, cs->seed,
                           cs->stream_id, cs->b0);
        SINDDM_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(final_conv1x1_kernel, dim3(((HW + 3) / 4 + 255) / 256, B), dim3(256), 0, st, cur, params + P.fin_w,
                       params + P.fin_b, out, P.half, HW);
    SINDDM_LAUNCH_CHECK();
    return 0;
}

}  // namespace sinddm

// =====================================================================================
// C ABI
// =====================================================================================
using namespace sinddm;

extern "C" {

int sinddm_abi_version(void) { return SINDDM_ABI_VERSION; }

int64_t sinddm_param_count(int dim) { NetPlan p = make_plan(dim); return p.ok ? p.nparams : -1; }
int sinddm_param_tensors(int dim) { NetPlan p = make_plan(dim); return p.ok ? p.ntensors : -1; }
int64_t sinddm_param_offset(int dim, int idx) {
    NetPlan p = make_plan(dim);
    if (!p.ok || idx < 0 || idx >= p.ntensors) return -1;
    return p.tensor_off[idx];
}
int64_t sinddm_packed_count(int dim) { NetPlan p = make_plan(dim); return p.ok ? p.npacked : -1; }

size_t sinddm_workspace_bytes(int dim, int B, int H, int W) {
    NetPlan p = make_plan(dim);
    if (!p.ok || B <= 0 || H <= 0 || W <= 0) return 0;
    return fwd_workspace_bytes(p, B, H, W);
}

int sinddm_pack_weights(const float* params, float* packed, int dim, void* stream) {
    if (!params || !packed) return SINDDM_E_BADARG;
    NetPlan p = make_plan(dim);
    if (!p.ok) return SINDDM_E_BADSHAPE;
    return pack_forward(p, params, packed, static_cast<hipStream_t>(stream));
}

int sinddm_net_forward(const float* params, const float* packed, const float* x, const int64_t* t_dev, int t_host,
                       float scale, float* out, int dim, int B, int H, int W, void* ws, size_t ws_bytes,
                       void* stream) {
    if (!params || !packed || !x || !out || !ws || B <= 0 || H <= 0 || W <= 0) return SINDDM_E_BADARG;
    NetPlan p = make_plan(dim);
    if (!p.ok) return SINDDM_E_BADSHAPE;
    return net_forward_impl(p, params, packed, x, t_dev, t_host, scale, out, B, H, W, ws, ws_bytes,
                            static_cast<hipStream_t>(stream), nullptr);
}

int sinddm_cond_embed(const float* params, const int64_t* t_dev, int t_host, float scale, int dim, int B,
                      float* emb_out, float* cond_vec_out, float* block_bias_out, void* stream) {
    if (!params || !block_bias_out || B <= 0) return SINDDM_E_BADARG;
    NetPlan P = make_plan(dim);
    if (!P.ok) return SINDDM_E_BADSHAPE;
    CondArgs ca{};
    ca.params = params; ca.t_dev = reinterpret_cast<const long long*>(t_dev); ca.t_host = t_host; ca.scale = scale;
    ca.out = block_bias_out; ca.cond_stride = P.cond_stride;
    ca.tm0_w = P.tm0_w; ca.tm0_b = P.tm0_b; ca.tm2_w = P.tm2_w; ca.tm2_b = P.tm2_b;
    for (int l = 0; l < 4; ++l) {
        ca.mlp_w[l] = P.blk[l].mlp_w; ca.mlp_b[l] = P.blk[l].mlp_b;
        ca.tr_w[l] = P.blk[l].tr_w; ca.tr_b[l] = P.blk[l].tr_b;
        ca.cin[l] = P.blk[l].cin; ca.coff[l] = P.blk[l].cond_off;
    }
    ca.cond_vec = cond_vec_out; ca.emb_out = emb_out;
    hipLaunchKernelGGL(cond_kernel, dim3(B), dim3(128), 0, static_cast<hipStream_t>(stream), ca);
    SINDDM_LAUNCH_CHECK();
    return 0;
}

int sinddm_cond_stride(int dim) { NetPlan p = make_plan(dim); return p.ok ? p.cond_stride : -1; }

int sinddm_q_sample(const float* x0, const float* x_orig, const float* noise, float* out, const float* tab_sqrt_ac,
                    const float* tab_sqrt_1m_ac, const float* gamma_row, const int64_t* t_dev, int t_host, int B,
                    int64_t n, void* stream) {
    if (!x0 || !noise || !out || !tab_sqrt_ac || !tab_sqrt_1m_ac || B <= 0 || n <= 0) return SINDDM_E_BADARG;
    if (x_orig && !gamma_row) return SINDDM_E_BADARG;
    const bool al = ((reinterpret_cast<uintptr_t>(x0) | reinterpret_cast<uintptr_t>(noise) | reinterpret_cast<uintptr_t>(out) |
                      reinterpret_cast<uintptr_t>(x_orig)) & 15) == 0;
    if (n % 4 == 0 && al) {
        const bool big = (long long)B * (n / 4) >= 4LL * 1024 * 2048;       // >= 8 workgroups of 1024 quads per CU
        long long bx = big ? (n / 4 + 1023) / 1024 : (n / 4 + 255) / 256;
        if (bx > 4096) bx = 4096;
        if (big)
            hipLaunchKernelGGL((q_sample_kernel<4, 4>), dim3((unsigned)bx, B), dim3(256), 0, static_cast<hipStream_t>(stream), x0,
                               x_orig, noise, out, tab_sqrt_ac, tab_sqrt_1m_ac, gamma_row,
                               reinterpret_cast<const long long*>(t_dev), t_host, (long long)n);
        else
            hipLaunchKernelGGL((q_sample_kernel<4, 1>), dim3((unsigned)bx, B), dim3(256), 0, static_cast<hipStream_t>(stream), x0,
                               x_orig, noise, out, tab_sqrt_ac, tab_sqrt_1m_ac, gamma_row,
                               reinterpret_cast<const long long*>(t_dev), t_host, (long long)n);
    } else {
        long long bx = (n + 255) / 256;
        if (bx > 4096) bx = 4096;
        hipLaunchKernelGGL((q_sample_kernel<1, 1>), dim3((unsigned)bx, B), dim3(256), 0, static_cast<hipStream_t>(stream), x0,
                           x_orig, noise, out, tab_sqrt_ac, tab_sqrt_1m_ac, gamma_row,
                           reinterpret_cast<const long long*>(t_dev), t_host, (long long)n);
    }
    SINDDM_LAUNCH_CHECK();
    return 0;
}

int sinddm_reverse_step(const float* x_t, const float* eps, const float* x_tilde, const float* noise, float* out,
                        const sinddm_step_coefs* coefs, int64_t n, void* stream) {
    if (!x_t || !eps || !noise || !out || !coefs || n <= 0) return SINDDM_E_BADARG;
    if (coefs->mode != 0 && !x_tilde) return SINDDM_E_BADARG;
    long long bx = (n + 255) / 256;
    if (bx > 8192) bx = 8192;
    hipLaunchKernelGGL(reverse_step_kernel<false>, dim3((unsigned)bx), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x_t, eps, x_tilde, noise, out, *coefs, (long long)n, nullptr, nullptr, 1, 1);
    SINDDM_LAUNCH_CHECK();
    return 0;
}

int sinddm_normal_fill(float* out, int64_t n, uint64_t seed, uint64_t stream_id, void* stream) {
    if (!out || n <= 0) return SINDDM_E_BADARG;
    long long bx = ((n + 3) / 4 + 255) / 256;
    if (bx > 8192) bx = 8192;
    hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)bx), dim3(256), 0, static_cast<hipStream_t>(stream), out,
                       (long long)n, (unsigned long long)seed, (unsigned long long)stream_id);
    SINDDM_LAUNCH_CHECK();
    return 0;
}

// sampler runs whose dim -> dim conv launches carry between LO and HI (8x32 tile, 80-channel block) items per CU AND leave
// at least 4 % of their last round of items empty are run as two half-batches on two streams (sinddm_sample_chain2).
// Below LO a half-batch falls onto the one-m-tile kernels (C2 48x64 at batch 16: -3.5 %); launches that fill their rounds
// gain nothing (C2 94x126: -2 %); C2 67x90 +10 %, C4 65x82 +11 %, 86x109 +5 %, 113x144 +4.5 % (profiles/NOTES_r04.md).
#ifndef SINDDM_SPLIT_ITEMS_HI
#define SINDDM_SPLIT_ITEMS_HI 16
#endif
#ifndef SINDDM_SPLIT_ITEMS_LO
#define SINDDM_SPLIT_ITEMS_LO 3
#endif

int sinddm_sample_chain2(const float* params, const float* packed, float* x, float* x_alt, float* eps, const float* x_tilde,
                         const sinddm_step_coefs* coefs, const int* t_list, int n_steps, float scale, uint64_t seed,
                         uint64_t stream_id0, int dim, int B, int H, int W, void* ws, size_t ws_bytes, void* stream,
                         void* aux_stream, int* result_in_alt) {
    if (!params || !packed || !x || !x_alt || !eps || !coefs || !t_list || !ws || n_steps < 0 || B <= 0 || H <= 0 || W <= 0)
        return SINDDM_E_BADARG;
    NetPlan p = make_plan(dim);
    if (!p.ok) return SINDDM_E_BADSHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipStream_t sx = static_cast<hipStream_t>(aux_stream);
    const long long n = (long long)B * CHANNELS * H * W;
    long long bx = ((n + 3) / 4 + 255) / 256;
    if (bx > 8192) bx = 8192;
    if (ws_bytes < fwd_workspace_bytes(p, B, H, W)) return SINDDM_E_WORKSPACE;
    float* cond_tab = static_cast<float*>(ws);                 // the conditioning region: one row per step of a run
    const bool fuse_tail = (H * W) % 4 == 0 || fwd_pitch(p, W) != W;     // (padded rows: their own fused tail kernel)
    // Coarse pyramid scales: a launch carries a handful of work items per CU (C2 48x64 at batch 16: 1.5), every kernel ends
    // in a partly filled round and pays its fixed ramp / drain, and each step is a chain of 16 dependent launches.  The
    // chains of the batch are independent, so with a second stream the batch runs as TWO half-batches whose launches
    // overlap: the tail round of one fills with the other's items.  Same numbers either way (the noise is keyed on the
    // whole batch's flat index: ChainStep::b0).
    const long long items = (long long)B * ((fwd_pitch(p, W) + 31) / 32) * ((H + 7) / 8) * 2;
    const long long ncu = wino2_cu_count();
    const long long rounds = (items + ncu - 1) / ncu;
    bool split = sx != nullptr && sx != st && B >= 2 && fuse_tail && n_steps > 0 && items < SINDDM_SPLIT_ITEMS_HI * ncu &&
                 items >= SINDDM_SPLIT_ITEMS_LO * ncu && (rounds * ncu - items) * 25 >= rounds * ncu;
    // (a batch whose 3x3 convs take the binary16 kernels stays whole: its halves could fall below their items-per-CU
    // threshold and run on the fp32 kernels -- the same chain would then round differently with and without a second stream)
    if (split && conv_wh_applies(p.fp32_convs, B, H, fwd_pitch(p, W), p.dim, p.dim))
        split = false;
    // (the two halves carve their own cores behind the shared conditioning table: only if that really fits -- ADVICE r4)
    if (split && cond_region_bytes(p, B) + fwd_workspace_core(p, (B + 1) / 2, H, W) + fwd_workspace_core(p, B - (B + 1) / 2, H, W) > ws_bytes)
        split = false;
    const int Bh[2] = {split ? (B + 1) / 2 : B, split ? B - (B + 1) / 2 : 0};
    char* wsh[2] = {static_cast<char*>(ws), nullptr};
    size_t wsz[2] = {ws_bytes, 0};
    hipEvent_t ev_go = nullptr, ev_done = nullptr;
    if (split) {
        wsh[0] = static_cast<char*>(ws) + cond_region_bytes(p, B);
        wsz[0] = fwd_workspace_core(p, Bh[0], H, W);
        wsh[1] = wsh[0] + wsz[0];
        wsz[1] = fwd_workspace_core(p, Bh[1], H, W);
        if (hipEventCreateWithFlags(&ev_go, hipEventDisableTiming) != hipSuccess) return SINDDM_E_BADARG;
        if (hipEventCreateWithFlags(&ev_done, hipEventDisableTiming) != hipSuccess) {
            (void)hipEventDestroy(ev_go);
            return SINDDM_E_BADARG;
        }
    }
    const size_t hoff = (size_t)Bh[0] * CHANNELS * H * W;      // the second half's offset into x / x_alt / eps / x_tilde
    float* cur = x;
    float* nxt = x_alt;
    int rc = 0;
    for (int i0 = 0; i0 < n_steps && rc == 0;) {
        // a run: up to CHAIN_COND_ROWS steps whose t is an arithmetic progression (the sampler's always is)
        int len = 1, dt = 0;
        if (i0 + 1 < n_steps) {
            dt = t_list[i0 + 1] - t_list[i0];
            len = 2;
            while (i0 + len < n_steps && len < CHAIN_COND_ROWS && t_list[i0 + len] - t_list[i0 + len - 1] == dt) ++len;
        }
        CondArgs ca{};
        ca.params = params; ca.t_dev = nullptr; ca.t_host = t_list[i0]; ca.t_step = dt; ca.scale = scale;
        ca.out = cond_tab; ca.cond_stride = p.cond_stride;
        ca.tm0_w = p.tm0_w; ca.tm0_b = p.tm0_b; ca.tm2_w = p.tm2_w; ca.tm2_b = p.tm2_b;
        for (int l = 0; l < 4; ++l) {
            ca.mlp_w[l] = p.blk[l].mlp_w; ca.mlp_b[l] = p.blk[l].mlp_b;
            ca.tr_w[l] = p.blk[l].tr_w; ca.tr_b[l] = p.blk[l].tr_b;
            ca.cin[l] = p.blk[l].cin; ca.coff[l] = p.blk[l].cond_off;
        }
        hipLaunchKernelGGL(cond_kernel, dim3(len), dim3(128), 0, st, ca);
        if (hipGetLastError() != hipSuccess) { rc = SINDDM_E_BADARG; break; }      // (no early return: the events below are ours)
        if (split) {                                           // the second stream starts behind the table (and behind
            // everything the caller enqueued before this call); a failed dependency must not become a silent race
            if (hipEventRecord(ev_go, st) != hipSuccess || hipStreamWaitEvent(sx, ev_go, 0) != hipSuccess) { rc = SINDDM_E_BADARG; break; }
        }
        for (int i = i0; i < i0 + len && rc == 0; ++i) {
            if (coefs[i].mode != 0 && !x_tilde) { rc = SINDDM_E_BADARG; break; }
            for (int h = 0; h < (split ? 2 : 1) && rc == 0; ++h) {
                const size_t o = h ? hoff : 0;
                ChainStep cs{};
                cs.cond_row = cond_tab + (size_t)(i - i0) * p.cond_stride;
                cs.x_tilde = x_tilde ? x_tilde + o : nullptr; cs.x_next = fuse_tail ? nxt + o : nullptr; cs.coefs = coefs[i];
                cs.seed = (unsigned long long)seed; cs.stream_id = (unsigned long long)(stream_id0 + (uint64_t)i);
                cs.b0 = h ? Bh[0] : 0;
                rc = net_forward_impl(p, params, packed, cur + o, nullptr, t_list[i], scale, eps + o, Bh[h], H, W, wsh[h], wsz[h],
                                      h ? sx : st, nullptr, &cs);
            }
            if (rc) break;
            if (!fuse_tail) {
                hipLaunchKernelGGL(reverse_step_rng_kernel, dim3((unsigned)bx), dim3(256), 0, st, cur, eps, x_tilde, nxt, coefs[i],
                                   n, (unsigned long long)seed, (unsigned long long)(stream_id0 + (uint64_t)i));
                if (hipGetLastError() != hipSuccess) { rc = SINDDM_E_BADARG; break; }
            }
            float* t_ = cur; cur = nxt; nxt = t_;
        }
        if (split) {                                           // the caller's stream continues behind both halves (and the
            // next run's table is not written under the second half) -- ALSO when a launch failed mid-run: the caller
            // frees x / x_alt / eps on the error path, the work already queued on the second stream must be behind `st`
            if (hipEventRecord(ev_done, sx) != hipSuccess || hipStreamWaitEvent(st, ev_done, 0) != hipSuccess) {
                (void)hipStreamSynchronize(sx);
                if (!rc) rc = SINDDM_E_BADARG;
            }
        }
        i0 += len;
    }
    if (split) {
        (void)hipEventDestroy(ev_go);                          // (destruction is deferred until the recorded work is done)
        (void)hipEventDestroy(ev_done);
    }
    if (rc) return rc;
    if (result_in_alt) *result_in_alt = (cur == x_alt) ? 1 : 0;
    return 0;
}

int sinddm_sample_chain(const float* params, const float* packed, float* x, float* x_alt, float* eps, const float* x_tilde,
                        const sinddm_step_coefs* coefs, const int* t_list, int n_steps, float scale, uint64_t seed,
                        uint64_t stream_id0, int dim, int B, int H, int W, void* ws, size_t ws_bytes, void* stream,
                        int* result_in_alt) {
    return sinddm_sample_chain2(params, packed, x, x_alt, eps, x_tilde, coefs, t_list, n_steps, scale, seed, stream_id0, dim, B,
                                H, W, ws, ws_bytes, stream, nullptr, result_in_alt);
}

int sinddm_reverse_step_edit(const float* x_t, const float* eps, const float* x_tilde, const float* noise, float* out,
                             const sinddm_step_coefs* coefs, const float* edit_w, const float* edit_c, int B, int C,
                             int HW, void* stream) {
    if (!x_t || !eps || !noise || !out || !coefs || !edit_w || !edit_c || B <= 0 || C <= 0 || HW <= 0)
        return SINDDM_E_BADARG;
    if (coefs->mode != 0 && !x_tilde) return SINDDM_E_BADARG;
    if ((long long)C * HW > 0x7fffffffLL) return SINDDM_E_BADSHAPE;
    const long long n = (long long)B * C * HW;
    long long bx = (n + 255) / 256;
    if (bx > 8192) bx = 8192;
    hipLaunchKernelGGL(reverse_step_kernel<true>, dim3((unsigned)bx), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x_t, eps, x_tilde, noise, out, *coefs, n, edit_w, edit_c, C * HW, HW);
    SINDDM_LAUNCH_CHECK();
    return 0;
}

#ifdef W2_PHASE
int sinddm_debug_w2_seg(unsigned long long* host_dst, int n) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_w2_seg), sizeof(unsigned long long) * n);
}
int sinddm_debug_w2_phase(unsigned long long* host_dst, int n) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_w2_phase), sizeof(unsigned long long) * n);
}
#endif
#ifdef WH_TIMING
int sinddm_debug_wh_seg(unsigned long long* host_dst, int n) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_wh_seg), sizeof(unsigned long long) * n);
}
#endif
#ifdef W4_TIMING
int sinddm_debug_w4_seg(unsigned long long* host_dst, int n) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_w4_seg), sizeof(unsigned long long) * n);
}
#endif
#ifdef W4_KSTAMP
int sinddm_debug_w4_ks(unsigned long long* host_dst, int n) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_w4_ks), sizeof(unsigned long long) * n);
}
#endif
#ifdef W2_TIMING
int sinddm_debug_w2_timing(unsigned long long* host_dst, int n) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_w2_dbg), sizeof(unsigned long long) * n);
}
#endif
#ifdef SINDDM_WINO_TIMING
int sinddm_debug_wino_timing(unsigned long long* host_dst, int n) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_wino_dbg), sizeof(unsigned long long) * n);
}
#endif

int sinddm_prof_begin(void) {
    ConvProfiler& p = conv_profiler();
    p.on = true;
    p.used = 0;
    p.flops = 0.0;
    p.exec_flops = 0.0;
    return 0;
}

int sinddm_prof_end(double* conv_ms_total, int64_t* conv_launches, double* conv_flops_total) {
    return sinddm_prof_end2(conv_ms_total, conv_launches, conv_flops_total, nullptr);
}

int sinddm_prof_end2(double* conv_ms_total, int64_t* conv_launches, double* conv_flops_total,
                     double* conv_exec_flops_total) {
    return sinddm_prof_end3(0, conv_ms_total, conv_launches, conv_flops_total, conv_exec_flops_total, 1);
}

int sinddm_debug_infer_path(int dim, int B, int H, int W) {
    NetPlan p = make_plan(dim);
    if (!p.ok || B <= 0 || H <= 0 || W <= 0) return SINDDM_E_BADARG;
    const BlockPlan& b = p.blk[2];                      // the dim -> dim block
    const int Wp = fwd_pitch(p, W);
    if (b.pk_q2 >= 0 && conv_wh_applies(p.fp32_convs, B, H, Wp, b.cout, b.cout)) return 8;
    return conv3x3_path(b.cout, b.cout, b.coblks, B, H, Wp);
}

int sinddm_prof_end3(int kind, double* ms_total, int64_t* launches, double* flops_total, double* exec_flops_total,
                     int reset) {
    ConvProfiler& p = conv_profiler();
    p.on = false;
    double ms = 0.0, fl = 0.0, ex = 0.0;
    int64_t n = 0;
    for (int i = 0; i < p.used; ++i) {
        if (kind >= 10 ? (p.kind[i] != kind / 10 || p.gen[i] != kind % 10) : (kind != 0 && p.kind[i] != kind)) continue;
        hipError_t e = hipEventSynchronize(p.ev[2 * i + 1]);
        if (e != hipSuccess) return (int)e;
        float t = 0.f;
        e = hipEventElapsedTime(&t, p.ev[2 * i], p.ev[2 * i + 1]);
        if (e != hipSuccess) return (int)e;
        ms += t; fl += p.rec_flops[i]; ex += p.rec_exec[i]; ++n;
    }
    if (ms_total) *ms_total = ms;
    if (launches) *launches = n;
    if (flops_total) *flops_total = fl;
    if (exec_flops_total) *exec_flops_total = ex;
    if (reset) { p.used = 0; p.flops = 0.0; p.exec_flops = 0.0; }
    return 0;
}

int sinddm_upsample_bilinear(const float* in, float* out, int BC, int h, int w, int H, int W, void* stream) {
    if (!in || !out || BC <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return SINDDM_E_BADARG;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    hipLaunchKernelGGL(upsample_bilinear_kernel, dim3((H * W + 255) / 256, BC), dim3(256), 0,
                       static_cast<hipStream_t>(stream), in, out, h, w, H, W, sy, sx);
    SINDDM_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
