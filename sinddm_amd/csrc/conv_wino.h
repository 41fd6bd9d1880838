// Winograd F(2x2, 3x3) 3x3 convolution on the gfx950 fp32 matrix cores.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 2x2 output tile, d = its 4x4 input patch
//
// 16 independent "frequency" GEMMs  M_xi[co][tile] = sum_ci U_xi[co][ci] * V_xi[ci][tile]  replace the
// 9-tap implicit GEMM: 16 multiplies per 4 outputs instead of 36 (2.25x fewer MFMAs), all in fp32.
// Mapping: a workgroup = 16 waves owns an 8x32 pixel tile (4x16 = 64 output 2x2-tiles) for MT*16 output
// channels; WAVE xi owns frequency xi = (i,j): MT x 4 accumulator tiles (M = co, N = 16 tiles of one
// tile-row, K = ci, v_mfma_f32_16x16x4_f32).
//   * U (pre-transformed weights, pack kernel) is laid out per (co-block, chunk, xi) in exactly the MFMA
//     A-operand register image, so a wave streams its 20 registers per 16-channel chunk straight from
//     L2 with fully coalesced loads -- weights never touch LDS (no wave shares them).
//   * the RAW input halo tile (16 channels x 10 x 34) is LDS-DMA'd (double buffered, one barrier per chunk);
//     every wave builds its own V_xi operand on the fly: 4 LDS reads + 3 FMAs per value (B^T has two +-1
//     entries per row).  Plane stride 341 (odd) keeps the stride-2 lane pattern bank-conflict-free.
//   * the output transform A^T M A needs all 16 frequencies of a (co, tile): one 16-channel M tile at a time
//     goes through LDS ([xi][co][tile]), then thread (co, tile) produces its 2x2 pixels, applies the fused
//     epilogue (bias, GELU / GELU'(u), residual, pre-activation save) and stores.
// Same ConvArgs / epilogue contract as conv_mfma.h; 1x1 residual projections are not fused here (the
// caller runs them through the direct kernel first and passes the result as `resid`).
#pragma once
#include "conv_mfma.h"

namespace sinddm {

constexpr int WN_THREADS = 1024;
constexpr int WN_KC = 16;                      // input channels per chunk (4 k-steps)
constexpr int WN_TW = 32;
constexpr int WN_RS = WN_TW + 2;
// geometry for NTR tile-rows (2 pixel rows each) per workgroup
template <int NTR>
struct WinoGeom {
    static constexpr int TH = 2 * NTR;
    static constexpr int HR = TH + 2;
    static constexpr int PLANE = HR * WN_RS;
    static constexpr int PS = (PLANE % 2) ? PLANE : PLANE + 1;   // odd -> conflict-free stride-2 reads across the k lanes
    static constexpr int IN_LIN = WN_KC * PS;                    // floats per buffer
    static constexpr int IREGS = (IN_LIN + WN_THREADS - 1) / WN_THREADS;
    static constexpr int NTILES = NTR * 16;
    static constexpr int MSTRIDE = NTILES + 1;                   // [xi][16 co][tiles + 1]
    static constexpr int LDS_FLOATS = (2 * IN_LIN > 16 * 16 * MSTRIDE) ? 2 * IN_LIN : 16 * 16 * MSTRIDE;
};

template <int MT, int NTR>
__global__ __launch_bounds__(WN_THREADS) void conv_wino_kernel(ConvArgs p) {
    using WG = WinoGeom<NTR>;
    constexpr int WN_TH = WG::TH, WN_PLANE = WG::PLANE, WN_PS = WG::PS, WN_IN_LIN = WG::IN_LIN;
    constexpr int WN_IREGS = WG::IREGS, WN_MSTRIDE = WG::MSTRIDE;
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
    const int trm = tile - b * tpi;
    const int ty = trm / p.tilesX;
    const int tx = trm - ty * p.tilesX;
    const int y0 = ty * WN_TH, x0 = tx * WN_TW;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int xi = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = Winograd frequency (i, j)
    const int l16 = lane & 15, kq = lane >> 4;
    const int H = p.H, W = p.W;
    const int HW = H * W;

    // B^T rows: which two patch rows/cols a frequency index combines, and their signs
    //   0: +d0 -d2   1: +d1 +d2   2: -d1 +d2   3: +d1 -d3
    const int fi = xi >> 2, fj = xi & 3;
    const int pa0 = fi == 0 ? 0 : 1, pa1 = fi == 3 ? 3 : 2;
    const int pb0 = fj == 0 ? 0 : 1, pb1 = fj == 3 ? 3 : 2;
    const float sa0 = fi == 2 ? -1.f : 1.f, sa1 = (fi == 0 || fi == 3) ? -1.f : 1.f;
    const float sb0 = fj == 2 ? -1.f : 1.f, sb1 = (fj == 0 || fj == 3) ? -1.f : 1.f;
    const float s00 = sa0 * sb0, s01 = sa0 * sb1, s10 = sa1 * sb0, s11 = sa1 * sb1;
    const int lbase = kq * WN_PS + 2 * l16;
    const int o00 = lbase + pa0 * WN_RS + pb0, o01 = lbase + pa0 * WN_RS + pb1;
    const int o10 = lbase + pa1 * WN_RS + pb0, o11 = lbase + pa1 * WN_RS + pb1;

    // raw-tile staging map: element idx of the linear LDS image [kc][WN_PS] -> byte offset inside the chunk's
    // channel block, or -1 (out of range -> the buffer bounds check returns 0).  Recomputed at every issue
    // (a dozen VALU ops per element) instead of being held in registers: the kernel is register-bound.
    auto src_off = [&](int idx) -> int {
        const int kc = idx / WN_PS;
        const int e = idx - kc * WN_PS;
        const int r = e / WN_RS;
        const int c = e - r * WN_RS;
        const int gy = y0 + r - 1, gx = x0 + c - 1;
        const bool ok = e < WN_PLANE && gy >= 0 && gy < H && gx >= 0 && gx < W;
        return ok ? (kc * HW + gy * W + gx) * 4 : -1;
    };
    const int nch = p.nch3;                                        // 16-channel chunks
    auto issue = [&](int c, float* buf) {
        const int ch0 = c * WN_KC;
        const float* sbase = p.in + ((size_t)b * p.Cin + ch0) * HW;
        const int nvalid = (p.Cin - ch0) < WN_KC ? (p.Cin - ch0) : WN_KC;
        // buffer descriptor over the channels of this chunk that exist: every out-of-range offset (halo outside
        // the image = -1, channels past C_in, plane pad) is zero-filled by the bounds check, no select needed
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sbase), 0, nvalid * HW * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < WN_IREGS; ++i) {
            int idx = tid + i * WN_THREADS;
            asm volatile("" : "+v"(idx));      // keep the offset computation inside the chunk loop (no LICM -> no spill)
            if (idx < WN_IN_LIN)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(buf + i * WN_THREADS + xi * 64), 4, src_off(idx), 0, 0, 0);
        }
    };
    // weights: register image [coblk][chunk][xi][ks][mt][lane], streamed with buffer loads
    // (uniform descriptor + scalar chunk offset + lane*4: no 64-bit vector address arithmetic)
    const __amdgpu_buffer_rsrc_t rsw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w3), 0, 0x7FFFFFFC, 0x00020000);
    const int wlane = lane * 4;
    const int wblk = ((cb * nch) * 16 + xi) * (4 * MT * 64) * 4;          // bytes, wave-uniform
    constexpr int WCHUNK_B = 16 * 4 * MT * 64 * 4;
    // weight registers: a ring of two k-step slots (MT registers each); slot (ks & 1) holds k-step ks
    float w[2][MT];
    auto load_w = [&](int slot, int c, int ks) {
        const int so = wblk + c * WCHUNK_B + ks * (MT * 64 * 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            w[slot][mt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsw, wlane, so + mt * 256, 0));
    };
    load_w(0, 0, 0);
    load_w(1, 0, 1);

    f32x4 acc[MT][NTR];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTR; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue(0, smem);
    __syncthreads();
    // Schedule inside a chunk.  All 16 waves of the CU meet at the barrier at the end of every chunk, so
    // nothing young may be in flight there (the barrier carries a vmcnt(0) because of the LDS DMA):
    //   after k-step 0: load k-step 2 of this chunk          (slot 0)
    //   after k-step 1: load k-step 3 of this chunk (slot 1); issue the raw-tile DMA of the next chunk
    //   after k-step 2: load k-step 0 of the next chunk      (slot 0)
    //   k-step 3, barrier, then load k-step 1 of the next chunk (slot 1) -- needed one k-step later
    for (int c = 0; c < nch; ++c) {
        const float* cur = smem + (c & 1) * WN_IN_LIN;
        const bool more = c + 1 < nch;
        if (c > 0) load_w(1, c, 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int nt = 0; nt < NTR; ++nt) {
                const float* q = cur + ks * 4 * WN_PS + nt * 2 * WN_RS;
                const float bv = s00 * q[o00] + s01 * q[o01] + s10 * q[o10] + s11 * q[o11];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ks & 1][mt], bv, acc[mt][nt], 0, 0, 0);
            }
            if (ks == 0) load_w(0, c, 2);
            if (ks == 1) {
                load_w(1, c, 3);
                if (more) issue(c + 1, smem + ((c + 1) & 1) * WN_IN_LIN);
            }
            if (ks == 2 && more) load_w(0, c + 1, 0);
        }
        __syncthreads();
    }

    // ---- output transform + epilogue, one 16-channel M tile per pass ----
    float* sM = smem;
    const int cl = xi;                       // pass-2 role: this wave handles channel `cl` of the M tile
    const int tr = lane >> 4, tc = lane & 15;   // lane = 2x2 tile (tile-row, tile-col)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        // C layout: col = lane&15 -> tile-col, row = (lane>>4)*4 + r -> channel within the M tile
#pragma unroll
        for (int nt = 0; nt < NTR; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) sM[(xi * 16 + kq * 4 + r) * WN_MSTRIDE + nt * 16 + l16] = acc[mt][nt][r];
        __syncthreads();
        float m[16];
        const int lt = lane < WG::NTILES ? lane : 0;     // (NTR = 3: the last 16 lanes have no tile)
#pragma unroll
        for (int f = 0; f < 16; ++f) m[f] = sM[(f * 16 + cl) * WN_MSTRIDE + lt];
        __syncthreads();
        const int col = mt * 16 + cl;
        const int co = cb * (MT * 16) + col;
        if (co < p.Cout && lane < WG::NTILES) {
            // t[p][j] = sum_i A^T[p][i] m[i][j];  Y[p][q] = sum_j t[p][j] A^T[q][j]
            float t0[4], t1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t0[j] = m[0 * 4 + j] + m[1 * 4 + j] + m[2 * 4 + j];
                t1[j] = m[1 * 4 + j] - m[2 * 4 + j] - m[3 * 4 + j];
            }
            float yv[2][2];
            yv[0][0] = t0[0] + t0[1] + t0[2];
            yv[0][1] = t0[1] - t0[2] - t0[3];
            yv[1][0] = t1[0] + t1[1] + t1[2];
            yv[1][1] = t1[1] - t1[2] - t1[3];
            const float bvs = p.bias ? p.bias[cb * (MT * 16) + col] : 0.0f;
            const size_t cbase = ((size_t)b * p.Cout + co) * HW;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int y = y0 + 2 * tr + pp;
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const int x = x0 + 2 * tc + qq;
                    if (y < H && x < W) {
                        const size_t o = cbase + (size_t)y * W + x;
                        float v = yv[pp][qq] + bvs;
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
}

inline int wino_ntr() {
    static int v = [] {
        const char* e = getenv("SINDDM_WINO_NTR");
        const int n = e ? atoi(e) : 3;
        return n == 4 ? 4 : 3;
    }();
    return v;
}

template <int MT>
inline void conv_wino_launch_t(const ConvArgs& a, unsigned grid, int ntr, hipStream_t st) {
    if (ntr == 4) {
        constexpr size_t lds = WinoGeom<4>::LDS_FLOATS * sizeof(float);
        hipLaunchKernelGGL((conv_wino_kernel<MT, 4>), dim3(grid), dim3(WN_THREADS), lds, st, a);
    } else {
        constexpr size_t lds = WinoGeom<3>::LDS_FLOATS * sizeof(float);
        hipLaunchKernelGGL((conv_wino_kernel<MT, 3>), dim3(grid), dim3(WN_THREADS), lds, st, a);
    }
}

inline int conv_wino_launch(const ConvArgs& a_in, int mt, hipStream_t st) {
    ConvArgs a = a_in;
    const int ntr = wino_ntr();
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
    const int TH = 2 * ntr;
    a.tilesX = (a.W + WN_TW - 1) / WN_TW;
    a.tilesY = (a.H + TH - 1) / TH;
    a.ntiles = a.B * a.tilesX * a.tilesY;
    a.tiles_per_xcd = (a.ntiles + 7) / 8;
    const unsigned grid = (unsigned)(a.tiles_per_xcd * 8 * a.coblks);
    switch (mt) {
        case 5: conv_wino_launch_t<5>(a, grid, ntr, st); break;
        case 2: conv_wino_launch_t<2>(a, grid, ntr, st); break;
        case 1: conv_wino_launch_t<1>(a, grid, ntr, st); break;
        default: return SINDDM_E_BADSHAPE;
    }
    if (rec) {
        (void)hipEventRecord(prof.ev[2 * prof.used + 1], st);
        const double fl = 2.0 * a.B * a.H * a.W * (double)a.Cout * 9.0 * a.Cin;   // algorithmic (direct-conv) FLOPs
        prof.flops += fl;
        prof.exec_flops += fl * (16.0 / 36.0);
        ++prof.used;
    }
    SINDDM_LAUNCH_CHECK();
    return 0;
}

inline bool wino_enabled() {
    static int v = [] {
        const char* e = getenv("SINDDM_CONV_WINO");
        return e ? atoi(e) : 1;
    }();
    return v != 0;
}

}  // namespace sinddm
