// 1x1 convolution (channel mixing) on the fp32 matrix cores:  out[b][co][p] = epi(sum_ci W[co][ci] in[b][ci][p] + bias)
//
// Used for the residual projections res_conv (reference SinDDM/models.py:67,80), their data gradients and the data
// gradient of final_conv.  No halo, so pixels are flattened: a workgroup (4 waves) owns 256 consecutive pixels of
// one image for MT*16 output channels; K = C_in is walked in 16-channel chunks whose operands are LDS-DMA'd (double
// buffered, one barrier per chunk): weights [ci][co] from the same packed 1x1 image the direct kernel uses
// ([coblk][8-ch chunk][ci][CO_LDS]: two consecutive 8-channel chunks form one 16-channel chunk), inputs as one
// 256-byte wave instruction per (channel, 64-pixel segment) with bounds-check zero fill.  HBM-bound
// (4*(C_in + C_out) B per pixel); LDS strides == 16 (mod 32) keep both operand reads conflict-free.
// MFMA column l16 of n-tile nt is pixel 4*l16 + nt of the wave's 64-pixel segment, so the B operand of a k-step is
// ONE ds_read_b128 and a lane owns 4 consecutive pixels per (channel) in the epilogue: 16-byte stores / residual
// loads instead of four 4-byte ones (the epilogue is store-issue bound).
#pragma once
#include "conv_mfma.h"

namespace sinddm {

constexpr int C1_THREADS = 256;
constexpr int C1_PIX = 256;                 // pixels per workgroup
constexpr int C1_KC = 16;                   // channels per chunk
constexpr int C1_PS = C1_PIX + 16;          // input plane stride (== 16 mod 32)

template <int MT>
__global__ __launch_bounds__(C1_THREADS) void conv1x1_mfma_kernel(ConvArgs p, int tiles_per_img) {
    constexpr int CO_LDS = ConvCfg<MT, 4>::CO_LDS;
    constexpr int W_FLOATS = C1_KC * CO_LDS;
    constexpr int BUF = W_FLOATS + C1_KC * C1_PS;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int cb = blockIdx.y;
    const int tile = blockIdx.x;
    const int b = tile / tiles_per_img;
    const int p0 = (tile - b * tiles_per_img) * C1_PIX;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, kq = lane >> 4;
    const int HW = p.H * p.W;
    const int nch = (p.Cin2 + C1_KC - 1) / C1_KC;
    constexpr unsigned OOB = 0x40000000u;

    // this wave stages the 64-pixel segment [p0 + 64*wave, +64) of every channel of a chunk
    const int pl = p0 + wave * 64 + lane;
    const unsigned loff = pl < HW ? (unsigned)pl * 4u : OOB;
    // weights: [coblk][chunk8][ci8][CO_LDS] -> 16-channel chunk c = floats [c*2*8*CO_LDS, +16*CO_LDS), linear copy
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.w1 + (size_t)cb * p.nch1 * KC * CO_LDS), 0, p.nch1 * KC * CO_LDS * 4, 0x00020000);

    auto issue = [&](int c, float* buf) {
        const int ch0 = c * C1_KC;
        const int nvalid = (p.Cin2 - ch0) < C1_KC ? (p.Cin2 - ch0) : C1_KC;
        const float* sbase = p.in2 + ((size_t)b * p.Cin2 + ch0) * HW;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sbase), 0, nvalid * HW * 4, 0x00020000);
        float* ib = buf + W_FLOATS;
#pragma unroll
        for (int kc = 0; kc < C1_KC; ++kc) {
            const unsigned coff = (kc < nvalid) ? (unsigned)(kc * HW) * 4u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(ib + kc * C1_PS + wave * 64), 4, (int)(loff + coff), 0, 0, 0);
        }
        // weights of the chunk: W_FLOATS floats, 64 per wave instruction (out-of-range tail of the last chunk -> 0)
        for (int i = wave; i * 64 < W_FLOATS; i += 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(buf + i * 64), 4, (c * W_FLOATS + i * 64 + lane) * 4, 0, 0, 0);
    };

    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int aBase = kq * CO_LDS + l16;
    const int bBase = W_FLOATS + kq * C1_PS + wave * 64 + l16 * 4;
    issue(0, smem);
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        const float* cur = smem + (c & 1) * BUF;
        if (c + 1 < nch) issue(c + 1, smem + ((c + 1) & 1) * BUF);
#pragma unroll
        for (int ks = 0; ks < C1_KC / 4; ++ks) {
            float a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = cur[aBase + ks * 4 * CO_LDS + mt * 16];
            const f32x4 bv = *reinterpret_cast<const f32x4*>(cur + bBase + ks * 4 * C1_PS);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], bv[nt], acc[mt][nt], 0, 0, 0);
        }
        __syncthreads();
    }

    const int px0 = p0 + wave * 64 + l16 * 4;          // this lane's 4 consecutive pixels
    const bool full = px0 + 3 < HW;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = mt * 16 + kq * 4 + r;
            const int co = cb * (MT * 16) + col;
            if (co >= p.Cout || px0 >= HW) continue;
            const float bvs = p.bias ? p.bias[cb * (MT * 16) + col] : 0.0f;
            const size_t o = ((size_t)b * p.Cout + co) * HW + px0;
            f32x4 v{acc[mt][0][r] + bvs, acc[mt][1][r] + bvs, acc[mt][2][r] + bvs, acc[mt][3][r] + bvs};
            if (full) {
                // 16-byte accesses at 4-byte alignment (H*W may be odd): fine for global memory on gfx9
                if (p.act == 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
                } else if (p.act == 2) {
                    const f32x4 u = *reinterpret_cast<const f32x4*>(p.aux + o);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] *= gelu_erf_grad(u[j]);
                }
                if (p.resid) v += *reinterpret_cast<const f32x4*>(p.resid + o);
                *reinterpret_cast<f32x4*>(p.out + o) = v;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (px0 + j < HW) {
                        float w = v[j];
                        if (p.act == 1) w = gelu_erf(w);
                        else if (p.act == 2) w *= gelu_erf_grad(p.aux[o + j]);
                        if (p.resid) w += p.resid[o + j];
                        p.out[o + j] = w;
                    }
                }
            }
        }
    }
}

inline int conv1x1_launch(const ConvArgs& a, int mt, hipStream_t st) {
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
    const int HW = a.H * a.W;
    const int tpi = (HW + C1_PIX - 1) / C1_PIX;
    const dim3 grid((unsigned)(a.B * tpi), (unsigned)a.coblks);
    switch (mt) {
        case 5: { constexpr size_t lds = 2 * (C1_KC * ConvCfg<5, 4>::CO_LDS + C1_KC * C1_PS) * sizeof(float);
                  hipLaunchKernelGGL(conv1x1_mfma_kernel<5>, grid, dim3(C1_THREADS), lds, st, a, tpi); break; }
        case 2: { constexpr size_t lds = 2 * (C1_KC * ConvCfg<2, 4>::CO_LDS + C1_KC * C1_PS) * sizeof(float);
                  hipLaunchKernelGGL(conv1x1_mfma_kernel<2>, grid, dim3(C1_THREADS), lds, st, a, tpi); break; }
        case 1: { constexpr size_t lds = 2 * (C1_KC * ConvCfg<1, 4>::CO_LDS + C1_KC * C1_PS) * sizeof(float);
                  hipLaunchKernelGGL(conv1x1_mfma_kernel<1>, grid, dim3(C1_THREADS), lds, st, a, tpi); break; }
        default: return SINDDM_E_BADSHAPE;
    }
    if (rec) {
        (void)hipEventRecord(prof.ev[2 * prof.used + 1], st);
        const double fl = 2.0 * a.B * a.H * a.W * (double)a.Cout * (double)a.Cin2;
        prof.note(2, fl, fl);
    }
    SINDDM_LAUNCH_CHECK();
    return 0;
}

}  // namespace sinddm
