// Declarations shared between the forward and backward translation units of libsinddm_hip.so.
#pragma once
#include "common.h"

namespace sinddm {

struct PackSeg {
    long long dst;     // offset in packed
    long long count;   // elements in this segment
    long long w;       // source weight offset (conv: [cout][cin][taps]), or bias offset
    long long w2;      // second bias offset to add (-1 none)
    int kind;          // 0: conv chunks, 1: bias, 2: zero page, 3: Winograd F(2x2) 3x3 weights, 4: Winograd F(2x4) 3x3 weights, 5: Winograd F(4x4) 3x3 weights
    int cin, cout, taps, nch, mt, co_lds;
    int transpose;     // 1: data-gradient image (M = cin of the forward conv, taps flipped)
};
struct PackArgs {
    PackSeg seg[32];
    int nseg;
    long long total;
};
int pack_launch(const float* params, float* packed, const PackArgs& a, hipStream_t st);

// Saved activations (training forward) + backward scratch, carved from the caller's workspace.
struct TrainBufs {
    float* cond;    // [B][cond_stride]   per-sample conv-block biases
    float* emb;     // [B][64]            sinusoidal embedding
    float* hpre;    // [B][128]           time_mlp hidden, pre-GELU
    float* cvec;    // [B][32]            cond vector, pre-GELU
    float* mvec;    // [B][4][32]         per-block mlp outputs
    float* h[4];    // dw5x5 + cond output          (cin  channels)
    float* u[4];    // conv1 pre-activation         (cout channels)
    float* g[4];    // GELU(u)                      (cout channels)
    float* o[4];    // block output                 (cout channels)
    float* s[4];    // backward scratch, dim channels each
    float* dcond;   // [B][cond_stride]
    float* small;   // cond-path backward scratch  [B][4*32 + 32 + 128]
    float* wscr;    // [dim][9][dim] staging slab of the 3x3 weight-gradient kernel
    float* amax;    // [2][B][AMAX_STRIDE] running-max scalars of the binary16 3x3 kernels' inputs: forward (slot 2l + i, as
                    // in inference), backward (slot 2l: gradient of block l's output, 2l + 1: gradient of its conv1 output)
};

struct ChainStep;   // sampler-run extras (sinddm_fwd.hip)
int net_forward_impl(const NetPlan& P, const float* params, const float* packed, const float* x, const int64_t* t_dev,
                     int t_host, float scale, float* out, int B, int H, int W, void* ws, size_t ws_bytes,
                     hipStream_t st, const TrainBufs* tb, const ChainStep* cs = nullptr);

// one SinDDMConvBlock forward (sinddm_fwd.hip); `cond` = the block's per-sample bias rows, stride in floats
int block_forward(const NetPlan& P, int l, const float* params, const float* packed, const float* cur, const float* cond,
                  int cond_stride, float* hbuf, float* gbuf, float* obuf, float* upre, int B, int H, int W, hipStream_t st,
                  int Wt = 0, float* amax = nullptr);
// which kernel generation a dim -> dim 3x3 conv launch of this shape takes: 4 / 3 = F(2x4) conv_wino4 / conv_wino3,
// 2 = F(2x2) conv_wino2 / conv_wino, 0 = direct implicit GEMM
int conv3x3_path(int cout, int cin, int coblks, int B, int H, int W);

// conv_wh.h (the Winograd F(2x4) kernel with binary16 hi/lo frequency GEMMs) lives in sinddm_fwd.hip; the backward TU
// reaches it through these: the dispatch rule, one launch, the weight image of a conv (transpose = 1: of its data gradient)
struct ConvArgs;
bool wh_applies(const NetPlan& P, int B, int H, int W, int cin, int cout);   // (false for every launch of a plan with fp32_convs)
bool wh_enabled();       // compiled in
int wh_conv(const ConvArgs& c, hipStream_t st);
int wh_pack(const float* w, float* wsinv, void* img, int cin, int cout, int transpose, hipStream_t st);
// max |x| of every sample of x[B][per_sample] (per_sample % 4 == 0) into amax[b * AMAX_STRIDE] (zeroed by the caller)
int amax_tensor_launch(const float* x, float* amax, int B, long long per_sample, hipStream_t st);

int dwconv_launch(const float* x, const float* w, const float* bias, const float* cond, int cond_stride,
                  const float* addt, int flip, float* out, int B, int C, int H, int W, hipStream_t st, int pi = 0, int po = 0,
                  float* amax = nullptr);

}  // namespace sinddm
