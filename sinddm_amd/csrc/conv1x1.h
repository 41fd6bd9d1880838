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
    dma_barrier();
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
        dma_barrier();
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

// -------------------------------------------------------------------------------------------------------------------
// Second generation for the 80-channel-block shapes (C_out = 80 or 160): the 80 -> 160 projection is MFMA-bound at
// fp32 (26.7 FLOP per byte against a machine balance of 19.6), so what matters is that every byte is read once and the
// matrix pipe never waits for a barrier:
//   * persistent 4-wave workgroups; the whole [C_in][C_out] weight matrix sits in LDS (staged once per workgroup in
//     MFMA A-operand order [k-step][m-tile][lane]: one conflict-free ds_read_b32 per (k-step, m-tile));
//   * a wave owns 64 consecutive pixels for ALL output channels (MTT = C_out / 16 <= 10 m-tiles x 4 n-tiles = 160
//     accumulator registers): the input is read exactly once, straight from global memory into the B registers (one
//     16-byte load per k-step and lane, 4-8 k-steps in flight) -- no LDS round trip, no barrier in the pixel loop;
//   * same epilogue contract as above (bias, GELU / GELU'(aux) *, residual, 16-byte stores).
// -------------------------------------------------------------------------------------------------------------------

#ifndef C1B_ABL
#define C1B_ABL 0        // timing ablations (wrong results): 1 no B loads in the loop, 2 no A reads in the loop, 4 no epilogue memory ops
#endif
template <int MTT, int ACT, int TAIL, int RES = 1>   // TAIL = 1 when H*W % 4 != 0: a lane's 4 pixels can straddle the end of a plane; RES = 0: no residual operand
__global__ __launch_bounds__(C1_THREADS, 2) void conv1x1_all_kernel(ConvArgs p, int tiles_per_img, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];       // [nks][MTT][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, kq = lane >> 4;
    const int HW = p.H * p.W;
    const int Cin = p.Cin2;
    const int nks = (Cin + 3) >> 2;
    constexpr int CO_LDS = ConvCfg<5, 4>::CO_LDS;                      // 80: packed image [coblk][chunk8][ci8][CO_LDS]
    constexpr unsigned OOB = 0x40000000u;
    constexpr int C1B_PF = 4;                                          // k-steps of B operands in flight per wave
    const int nks_pad = (nks + C1B_PF - 1) / C1B_PF * C1B_PF;          // the k-loop is branch-free: padded steps multiply zeros

    // ---- weights -> LDS, once: element (ci, local channel cl) lands at [(ci >> 2) * MTT + (cl >> 4)][(ci & 3) * 16 +
    // (cl & 15)]; ten independent loads per thread in flight; the zero padding of the k-steps beyond C_in is written first.
    // blockIdx.y picks the block of MTT m-tiles this workgroup computes (small launches split the output channels over
    // workgroups: a lone 256-pixel tile per CU would otherwise walk all C_out / 16 m-tiles -- and stage the whole weight
    // matrix -- alone)
    const int co0 = blockIdx.y * (MTT * 16);
    for (int e = tid; e < nks_pad * MTT * 64; e += C1_THREADS) smem[e] = 0.f;
    __syncthreads();
    {
        const int total = p.nch1 * KC * (MTT * 16);                     // (ci, local channel), the channel fastest
        for (int e0 = 0; e0 < total; e0 += C1_THREADS * 10) {
            float v[10];
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                const int e = e0 + u * C1_THREADS + tid;
                const int ci = e / (MTT * 16), cl = e - ci * (MTT * 16);
                const int co = co0 + cl, cb = co / 80, col = co - cb * 80;
                v[u] = e < total ? p.w1[((size_t)cb * p.nch1 * KC + ci) * CO_LDS + col] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                const int e = e0 + u * C1_THREADS + tid;
                const int ci = e / (MTT * 16), cl = e - ci * (MTT * 16);
                if (e < total && ci < Cin) smem[((ci >> 2) * MTT + (cl >> 4)) * 64 + (ci & 3) * 16 + (cl & 15)] = v[u];
            }
        }
    }
    // bias -> LDS too: the epilogue needs 4 * MTT per-lane values per tile, and a global load each would be 4 * MTT
    // serial memory round trips per tile
    float* sbias = smem + nks_pad * MTT * 64;
    for (int e = tid; e < MTT * 16; e += C1_THREADS) sbias[e] = p.bias ? p.bias[co0 + e] : 0.f;
    __syncthreads();

    // The B-operand stream runs ACROSS tiles: the last C1B_PF k-steps of a tile load the first k-steps of the next one,
    // BEFORE the epilogue's stores -- memory operations complete in order, so a load issued after 40 stores would wait
    // for every one of them (measured: 20 % of the kernel).
    struct TileCtx {
        __amdgpu_buffer_rsrc_t rs;
        unsigned pxo;
    };
    auto make_ctx = [&](int tile) {
        TileCtx c;
        const bool live = tile < ntiles;
        const int b = live ? tile / tiles_per_img : 0;
        const int p0 = (tile - b * tiles_per_img) * C1_PIX;
        const int px = p0 + wave * 64 + l16 * 4;
        c.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in2 + (size_t)b * Cin * HW), 0, live ? Cin * HW * 4 : 0,
                                                 0x00020000);
        c.pxo = px < HW ? (unsigned)px * 4u : OOB;
        return c;
    };
    // B operand of k-step ks: channel 4 ks + kq, the lane's 4 consecutive pixels (MFMA column l16 of n-tile nt = pixel + nt)
    auto load_b = [&](const TileCtx& c, int ks) -> f32x4 {
        const int ch = 4 * ks + kq;
        unsigned o = ch < Cin ? c.pxo + (unsigned)(ch * HW) * 4u : OOB;
        asm volatile("" : "+v"(o));
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(c.rs, (int)o, 0, 0));
    };
    f32x4 bq[C1B_PF];
    TileCtx ctx = make_ctx(blockIdx.x);
#pragma unroll
    for (int i = 0; i < C1B_PF; ++i) bq[i] = load_b(ctx, i);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_img;
        const int p0 = (tile - b * tiles_per_img) * C1_PIX;
        const int px0 = p0 + wave * 64 + l16 * 4;                       // this lane's 4 consecutive pixels
        const TileCtx nctx = make_ctx(tile + gridDim.x);
        f32x4 acc[MTT][4];
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // A operands ping-pong between two register sets: the reads of k-step ks+1 are ISSUED before the MFMAs of k-step
        // ks (pinned by the sched_barrier: left alone the compiler sinks them to the end of the k-step to save ten
        // registers, and every k-step then starts with an exposed LDS round trip)
        float aa[2][MTT];
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt) aa[0][mt] = smem[mt * 64 + lane];
        for (int k0 = 0; k0 < nks_pad; k0 += C1B_PF) {
#pragma unroll
            for (int i = 0; i < C1B_PF; ++i) {
                const int ks = k0 + i;
                const f32x4 bv = bq[i];
                if (!(C1B_ABL & 1)) {
                    const bool wrap = k0 + C1B_PF >= nks_pad;           // (uniform) last group: the next tile's first k-steps
                    TileCtx lc;
                    lc.rs = wrap ? nctx.rs : ctx.rs;
                    lc.pxo = wrap ? nctx.pxo : ctx.pxo;
                    bq[i] = load_b(lc, wrap ? i : ks + C1B_PF);
                }
                const int kn = ks + 1 < nks_pad ? ks + 1 : 0;           // (the wrap-around read is never used)
#pragma unroll
                for (int mt = 0; mt < MTT; ++mt) aa[(i + 1) & 1][mt] = (C1B_ABL & 2) ? aa[i & 1][mt] + 1.f : smem[(kn * MTT + mt) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MTT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[i & 1][mt], bv[nt], acc[mt][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- epilogue: lane holds rows 4 kq + r of every m-tile for its 4 pixels.  Without a plane tail every access is
        // a 16-byte BUFFER access on the sample's descriptor (lanes past the image carry an out-of-range offset; a null
        // residual is an empty descriptor): no branches, so the loads of an m-tile are in flight together and no
        // control-flow merge drains the stores.
        using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
        const size_t samp = (size_t)b * p.Cout * HW;
        const unsigned samp_b = (unsigned)(p.Cout * HW) * 4u;
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out + samp, 0, samp_b, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.resid ? p.resid + samp : p.zero), 0, p.resid ? samp_b : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_aux = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(ACT == 2 ? p.aux + samp : p.zero), 0, ACT == 2 ? samp_b : 0u, 0x00020000);
        // residual / pre-activation operands of an m-tile are requested BEFORE the stores of the previous one: memory
        // operations complete in order, so a load issued after a store waits for that store's round trip
        unsigned off[2][4];
        f32x4 rv[2][4], uv[2][4];
        auto fetch = [&](int mt, int slot) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                unsigned o = px0 < HW ? (unsigned)((co0 + mt * 16 + kq * 4 + r) * HW + px0) * 4u : OOB;
                asm volatile("" : "+v"(o));
                off[slot][r] = o;
                if (C1B_ABL & 4) { rv[slot][r] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
                if constexpr (RES) rv[slot][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, (int)o, 0, 0));
                if constexpr (ACT == 2) uv[slot][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_aux, (int)o, 0, 0));
            }
        };
        // TAIL builds (H*W % 4 != 0): in the last tile of an image ONE lane's 4 pixels straddle the end of a plane.  Its
        // 16-byte store is sent out of range and its 1..3 valid pixels are written through plain pointers behind a
        // per-lane branch (one wave per image diverges); operand LOADS may run into the next plane (still inside the
        // tensor, or clipped by the descriptor): the values they feed are never stored.  (Round 2's separate scalar path
        // for the whole tile spilled 80 registers; masked 4-byte BUFFER stores behind the dropped 16-byte one left pixels
        // 1 and 2 of the quad wrong on the hardware -- from builtins and from asm alike, unexplained.)
        const bool tail_tile = TAIL && p0 + C1_PIX > HW;
        const int keep = TAIL ? HW - px0 : 4;                     // >= 4: whole quad inside; 1..3: straddling; <= 0: outside
        const bool straddle = TAIL && keep > 0 && keep < 4;
        fetch(0, 0);
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt) {
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 bias4 = *reinterpret_cast<const f32x4*>(sbias + mt * 16 + kq * 4);
            if (mt + 1 < MTT) fetch(mt + 1, (mt + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x4 v{acc[mt][0][r], acc[mt][1][r], acc[mt][2][r], acc[mt][3][r]};
                v += bias4[r];
                if constexpr (ACT == 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
                } else if constexpr (ACT == 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] *= gelu_erf_grad(uv[mt & 1][r][j]);
                }
                if constexpr (RES) v += rv[mt & 1][r];
                if (C1B_ABL & 4) { if (v[0] + v[1] + v[2] + v[3] == 123.4f) p.out[tid] = v[1]; continue; }
                const unsigned o = off[mt & 1][r];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_out, (int)(straddle ? OOB : o), 0, 0);
                if (tail_tile && straddle) {             // (one lane of the image's last tile)
                    float* q = p.out + samp + (size_t)(co0 + mt * 16 + kq * 4 + r) * HW + px0;
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        if (j < keep) q[j] = v[j];
                }
            }
        }
        ctx = nctx;
    }
}

template <int MTT>
inline void conv1x1_all_launch(const ConvArgs& a, int tpi, int ntiles, hipStream_t st) {
    // (H*W % 4 != 0: the TAIL = 1 instantiation -- the one straddling lane of an image's last tile stores element-wise)
    const size_t lds = ((size_t)((a.Cin2 + 15) / 16 * 4) * MTT * 64 + MTT * 16) * sizeof(float);   // k-steps padded to 4, + bias
    const int cus = 256;
    const dim3 grid((unsigned)(ntiles < 2 * cus ? ntiles : 2 * cus), (unsigned)(a.Cout / (MTT * 16)));
#define C1B_GO(ACT, RES)                                                                                                \
    do {                                                                                                                \
        if ((a.H * a.W) % 4 == 0)                                                                                       \
            hipLaunchKernelGGL((conv1x1_all_kernel<MTT, ACT, 0, RES>), grid, dim3(C1_THREADS), lds, st, a, tpi, ntiles); \
        else                                                                                                            \
            hipLaunchKernelGGL((conv1x1_all_kernel<MTT, ACT, 1, RES>), grid, dim3(C1_THREADS), lds, st, a, tpi, ntiles); \
    } while (0)
    const int res = a.resid != nullptr;
    switch ((a.act & 0xff) * 2 + res) {
        case 0: C1B_GO(0, 0); break;
        case 1: C1B_GO(0, 1); break;
        case 2: C1B_GO(1, 0); break;
        case 3: C1B_GO(1, 1); break;
        case 4: C1B_GO(2, 0); break;
        default: C1B_GO(2, 1);
    }
#undef C1B_GO
}

#ifndef SINDDM_CONV1X1_V2
#define SINDDM_CONV1X1_V2 1
#endif
#ifndef SINDDM_CONV1X1_TAIL   // 1: H*W % 4 != 0 on the second-generation kernel too (TAIL instantiation)
#define SINDDM_CONV1X1_TAIL 1
#endif

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
    const int mtt = mt * a.coblks;
    if (SINDDM_CONV1X1_V2 && mt == 5 && (mtt == 5 || mtt == 10) && a.Cout == mtt * 16 && (SINDDM_CONV1X1_TAIL || HW % 4 == 0) && a.Cin2 <= 160) {
        const int ntiles = a.B * tpi;
        // enough pixel tiles to fill the chip: every workgroup computes all output channels (input read once);
        // fewer: the channels are split over workgroups -- 80 per workgroup, then 16 (coarse scales at small batch)
        const size_t lds_all = ((size_t)((a.Cin2 + 15) / 16 * 4) * mtt * 64 + mtt * 16) * sizeof(float);
        if (ntiles >= 512 && mtt == 10 && lds_all <= 64 * 1024) conv1x1_all_launch<10>(a, tpi, ntiles, st);
        else if (ntiles >= 128) conv1x1_all_launch<5>(a, tpi, ntiles, st);
        else conv1x1_all_launch<1>(a, tpi, ntiles, st);
    } else
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
