// First-generation Winograd F(2x2,3x3) kernel (round 2), archived: the product library runs conv_wino2.h instead
// (SINDDM_WINO_V2 = 1) and the packed Winograd image has the second generation's layout.  Included by conv_wino.h only in
// -DSINDDM_WINO_V2=0 variant builds (tools/build_variant.sh); not part of the default library.
#pragma once
// Compile-time timing ablations of the Winograd kernel (-DSINDDM_WINO_ABL=bits; results are WRONG, never ship):
//   1 no LDS reads / V transform   2 weights loaded once   4 raw-tile DMA only for the first chunk
//   8 no chunk barrier             16 no epilogue
#ifndef SINDDM_WINO_ABL
#define SINDDM_WINO_ABL 0
#endif

#ifdef SINDDM_WINO_TIMING
// per-wave s_memtime stamps of the first work items of workgroup 0 (debug builds only; tools/wino_timing.py)
__device__ unsigned long long g_wino_dbg[2 * 16 * 16 * 4];
#endif

constexpr int WN_THREADS = 1024;
constexpr int WN_KC = 16;                      // input channels per chunk (4 k-steps)
constexpr int WN_TW = 32;
constexpr int WN_RS = WN_TW + 2;
// geometry for NTR tile-rows (2 pixel rows each) per workgroup
template <int NTR, int CPC = 1>
struct WinoGeom {
    static constexpr int TH = 2 * NTR;
    static constexpr int HR = TH + 2;
    static constexpr int PLANE = HR * WN_RS;
    static constexpr int PS = (PLANE % 2) ? PLANE : PLANE + 1;   // odd -> conflict-free stride-2 reads across the k lanes
    static constexpr int IN_LIN = CPC * WN_KC * PS;              // floats per raw-tile buffer (CPC x 16 channels)
    static constexpr int RAW_FLOATS = (2 * IN_LIN + 3) / 4 * 4;  // two buffers
    static constexpr int NTILES = NTR * 16;
    // [xi][co][tiles + pad]: 4 * MSTRIDE == 16 (mod 32) puts the two k-lane groups of a half-wave (rows kq*4 + r,
    // 4 rows apart) on disjoint banks when the accumulators are written (stride NTILES + 1 = 49 was a 2-way conflict)
    static constexpr int MSTRIDE = (NTILES + 4) / 8 * 8 + 4;
    // M tiles (16 channels each) exchanged per epilogue pass: two if the LDS budget (160 KB) allows
    static constexpr int MPP = ((RAW_FLOATS + 16 * 32 * MSTRIDE) * 4 <= 160 * 1024) ? 2 : 1;
    static constexpr int LDS_FLOATS = RAW_FLOATS + 16 * (16 * MPP) * MSTRIDE;
};

struct WinoItem {          // one unit of work: an (8|6)x32 pixel tile x one block of MT*16 output channels
    int b, y0, x0, cb;
};

template <int MT, int NTR, int ACT, int CPC>
__global__ __launch_bounds__(WN_THREADS) void conv_wino_kernel(ConvArgs p, int items_per_xcd, int wg_per_xcd) {
    using WG = WinoGeom<NTR, CPC>;
    constexpr int NKS = 4 * CPC;                                   // k-steps (of 4 channels) per chunk
    constexpr int WN_TH = WG::TH, WN_PS = WG::PS, WN_IN_LIN = WG::IN_LIN;
    constexpr int WN_MSTRIDE = WG::MSTRIDE, MPP = WG::MPP;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sM = smem + WG::RAW_FLOATS;       // epilogue exchange area, disjoint from the raw-tile buffers

    // persistent workgroup: XCD `xcd` owns a contiguous range of tiles (its L2 serves their halos and both
    // output-channel blocks of a tile); workgroup `ls` of that XCD takes items ls, ls + wg_per_xcd, ...
    const int xcd = blockIdx.x & 7;
    const int ls = blockIdx.x >> 3;
    const int tpi = p.tilesX * p.tilesY;
    auto decode = [&](int l, WinoItem& it) -> bool {
        if (l >= items_per_xcd) return false;
        const int tile = xcd * p.tiles_per_xcd + l / p.coblks;
        if (tile >= p.ntiles) return false;
        it.cb = l % p.coblks;
        it.b = tile / tpi;
        const int trm = tile - it.b * tpi;
        const int ty = trm / p.tilesX;
        it.y0 = ty * WN_TH;
        it.x0 = (trm - ty * p.tilesX) * WN_TW;
        return true;
    };

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

    const int nch16 = p.nch3;                                      // 16-channel groups of the reduction
    const int nch = (nch16 + CPC - 1) / CPC;                       // chunks (CPC groups each)
    const int nks_total = nch16 * 4;
    // raw-tile DMA of chunk c of item `it`.  Wave w stages channel w of the 16-channel chunk: one LDS-DMA wave
    // instruction per tile row (34 active lanes -> 34 consecutive LDS floats).  The buffer descriptor covers exactly
    // that channel plane (0 bytes if the channel does not exist), so everything outside the image -- halo rows and
    // columns, missing channels -- is zero-filled by the hardware bounds check: out-of-range lanes/rows simply
    // carry an offset >= 2^30.  Per instruction: one scalar row offset + one v_add.
    constexpr unsigned OOB = 0x40000000u;
    auto issue = [&](const WinoItem& it, int c, float* buf) {
        const int gx = it.x0 + lane - 1;
        const unsigned loff = (gx >= 0 && gx < W) ? (unsigned)gx * 4u : OOB;
#pragma unroll
        for (int g = 0; g < CPC; ++g) {
            const int kc = g * 16 + xi;                            // channel of the chunk staged by this wave
            const int ch = c * (CPC * WN_KC) + kc;
            const float* sbase = p.in + ((size_t)it.b * p.Cin + (ch < p.Cin ? ch : 0)) * HW;
            const __amdgpu_buffer_rsrc_t rs =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sbase), 0, ch < p.Cin ? HW * 4 : 0, 0x00020000);
            if (lane < WN_RS) {
                float* pl = buf + kc * WN_PS;
#pragma unroll
                for (int r = 0; r < WG::HR; ++r) {
                    const int gy = it.y0 + r - 1;
                    const unsigned roff = (gy >= 0 && gy < H) ? (unsigned)(gy * W) * 4u : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(pl + r * WN_RS), 4, (int)(loff + roff), 0, 0, 0);
                }
            }
        }
    };
    // weights: register image [coblk][chunk][xi][ks][mt][lane], streamed with buffer loads
    // (uniform descriptor + scalar offset + lane*4: no 64-bit vector address arithmetic)
    const __amdgpu_buffer_rsrc_t rsw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w3), 0, 0x7FFFFFFC, 0x00020000);
    const int wlane = lane * 4;
    // weight registers: a ring of two k-step slots (MT registers each); slot (ks & 1) holds k-step ks
    float w[2][MT];
    // (chunk c, k-step ks) -> global k-step index; nothing is loaded (or multiplied) past the last real one
    auto load_w = [&](int slot, int cb, int c, int ks) {
        const int gk = c * NKS + ks;
        if ((SINDDM_WINO_ABL & 2) && gk >= 2) return;
        if (gk >= nks_total) return;
        const int so = ((cb * nch16 + (gk >> 2)) * 16 + xi) * (4 * MT * 64 * 4) + (gk & 3) * (MT * 64 * 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            w[slot][mt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsw, wlane, so + mt * 256, 0));
    };

    WinoItem it;
    int l = ls;
    if (!decode(l, it)) return;
    load_w(0, it.cb, 0, 0);
    load_w(1, it.cb, 0, 1);
    issue(it, 0, smem);
    dma_barrier();

    const int tr = lane >> 4, tc = lane & 15;   // epilogue role of a lane: 2x2 tile (tile-row, tile-col)
    const int lt = lane < WG::NTILES ? lane : 0;

#ifdef SINDDM_WINO_TIMING
    int dbg_item = 0;
#endif
    for (;;) {
        f32x4 acc[MT][NTR];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTR; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

        // Schedule inside a chunk of NKS k-steps.  All 16 waves of the CU meet at the barrier at the end of every
        // chunk, so nothing young may be in flight there (the barrier carries a vmcnt(0) because of the LDS DMA):
        //   after k-step ks < NKS-2 : load k-step ks+2 of this chunk into the slot just freed
        //   after k-step 1          : also issue the raw-tile DMA of the next chunk
        //   after k-step NKS-2      : load k-step 0 of the next chunk
        //   k-step NKS-1, barrier, then load k-step 1 of the next chunk -- needed one k-step later
        for (int c = 0; c < nch; ++c) {
#ifdef SINDDM_WINO_TIMING
            const bool dbg = blockIdx.x == 8 && dbg_item < 2 && c < 16 && lane == 0;
            if (dbg) g_wino_dbg[((dbg_item * 16 + c) * 16 + xi) * 4 + 0] = __builtin_amdgcn_s_memtime();
#endif
            const float* cur = smem + (c & 1) * WN_IN_LIN;
            const bool more = c + 1 < nch;
            if (c > 0) load_w(1, it.cb, c, 1);
            // the four raw values of the next (k-step, tile-row) group are fetched from LDS before the MFMAs of
            // the current group are issued, so the LDS round trip hides under those MFMAs
            float r4[4];
            {
                const float* q = cur;
                r4[0] = q[o00]; r4[1] = q[o01]; r4[2] = q[o10]; r4[3] = q[o11];
            }
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const bool live = (CPC == 1) || (c * NKS + ks < nks_total);
#pragma unroll
                for (int nt = 0; nt < NTR; ++nt) {
                    const float bv = (SINDDM_WINO_ABL & 1) ? (float)(ks + nt) * s00
                                                           : s00 * r4[0] + s01 * r4[1] + s10 * r4[2] + s11 * r4[3];
                    // next group (same k-step next tile-row, or first tile-row of the next k-step)
                    if (!(SINDDM_WINO_ABL & 1) && (nt + 1 < NTR || ks + 1 < NKS)) {
                        const int nks = (nt + 1 < NTR) ? ks : ks + 1;
                        const int nnt = (nt + 1 < NTR) ? nt + 1 : 0;
                        const float* q = cur + nks * 4 * WN_PS + nnt * 2 * WN_RS;
                        r4[0] = q[o00]; r4[1] = q[o01]; r4[2] = q[o10]; r4[3] = q[o11];
                    }
                    if (live) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ks & 1][mt], bv, acc[mt][nt], 0, 0, 0);
                    }
#ifdef SINDDM_WINO_TIMING
                    if (ks == 0 && nt == 0 && dbg) g_wino_dbg[((dbg_item * 16 + c) * 16 + xi) * 4 + 3] = __builtin_amdgcn_s_memtime();
#endif
                }
                if (ks < NKS - 2) load_w(ks & 1, it.cb, c, ks + 2);
                if (ks == 1 && more && !(SINDDM_WINO_ABL & 4)) issue(it, c + 1, smem + ((c + 1) & 1) * WN_IN_LIN);
                if (ks == NKS - 2 && more) load_w(0, it.cb, c + 1, 0);
            }
            // pin the schedule here: left alone, the compiler sinks the last k-step's MFMAs below the barrier and
            // rotates the accumulators through spare registers (3.5% slower, measured A/B on one box)
            __builtin_amdgcn_sched_barrier(0);
#ifdef SINDDM_WINO_TIMING
            if (dbg) g_wino_dbg[((dbg_item * 16 + c) * 16 + xi) * 4 + 1] = __builtin_amdgcn_s_memtime();
#endif
            if (!(SINDDM_WINO_ABL & 8)) dma_barrier();
#ifdef SINDDM_WINO_TIMING
            if (dbg) g_wino_dbg[((dbg_item * 16 + c) * 16 + xi) * 4 + 2] = __builtin_amdgcn_s_memtime();
#endif
        }
#ifdef SINDDM_WINO_TIMING
        ++dbg_item;
#endif

        // next work item: start its first raw tile and weight registers now, so that they arrive while this
        // item's epilogue runs (raw buffer 0 is free: every wave passed the last chunk barrier)
        WinoItem nx;
        l += wg_per_xcd;
        const bool have_next = decode(l, nx);
        if (have_next) {
            load_w(0, nx.cb, 0, 0);
            load_w(1, nx.cb, 0, 1);
            issue(nx, 0, smem);
        }

        if (!(SINDDM_WINO_ABL & 16)) {
            // ---- output transform + epilogue, MPP 16-channel M tiles per pass ----
#pragma unroll
            for (int m0 = 0; m0 < MT; m0 += MPP) {
                // C layout: col = lane&15 -> tile-col, row = (lane>>4)*4 + r -> channel within the M tile
#pragma unroll
                for (int h = 0; h < MPP; ++h) {
                    if (m0 + h < MT) {
#pragma unroll
                        for (int nt = 0; nt < NTR; ++nt)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                sM[(xi * (16 * MPP) + h * 16 + kq * 4 + r) * WN_MSTRIDE + nt * 16 + l16] = acc[m0 + h][nt][r];
                    }
                }
                // residual operands of this pass are fetched before the exchange so that their latency overlaps it
                float rs_v[MPP][4];
                size_t obase[MPP];
#pragma unroll
                for (int h = 0; h < MPP; ++h) {
                    const int co = it.cb * (MT * 16) + (m0 + h) * 16 + xi;
                    obase[h] = ((size_t)it.b * p.Cout + co) * HW + (size_t)(it.y0 + 2 * tr) * W + it.x0 + 2 * tc;
#pragma unroll
                    for (int k = 0; k < 4; ++k) rs_v[h][k] = 0.f;
                    if (p.resid && m0 + h < MT && co < p.Cout && lane < WG::NTILES) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int y = it.y0 + 2 * tr + (k >> 1), x = it.x0 + 2 * tc + (k & 1);
                            if (y < H && x < W) rs_v[h][k] = p.resid[obase[h] + (size_t)(k >> 1) * W + (k & 1)];
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int h = 0; h < MPP; ++h) {
                    if (m0 + h >= MT) break;
                    const int col = (m0 + h) * 16 + xi;              // this wave's channel of the M tile
                    const int co = it.cb * (MT * 16) + col;
                    float m[16];
#pragma unroll
                    for (int f = 0; f < 16; ++f) m[f] = sM[(f * (16 * MPP) + h * 16 + xi) * WN_MSTRIDE + lt];
                    if (co < p.Cout && lane < WG::NTILES) {
                        // t[p][j] = sum_i A^T[p][i] m[i][j];  Y[p][q] = sum_j t[p][j] A^T[q][j]
                        float t0[4], t1[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            t0[j] = m[0 * 4 + j] + m[1 * 4 + j] + m[2 * 4 + j];
                            t1[j] = m[1 * 4 + j] - m[2 * 4 + j] - m[3 * 4 + j];
                        }
                        float yv[4];
                        yv[0] = t0[0] + t0[1] + t0[2];
                        yv[1] = t0[1] - t0[2] - t0[3];
                        yv[2] = t1[0] + t1[1] + t1[2];
                        yv[3] = t1[1] - t1[2] - t1[3];
                        const float bvs = p.bias ? p.bias[it.cb * (MT * 16) + col] : 0.0f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int y = it.y0 + 2 * tr + (k >> 1), x = it.x0 + 2 * tc + (k & 1);
                            if (y < H && x < W) {
                                const size_t o = obase[h] + (size_t)(k >> 1) * W + (k & 1);
                                float v = yv[k] + bvs;
                                if (ACT == 1) {
                                    if (p.out_pre) p.out_pre[o] = v;
                                    v = gelu_erf(v);
                                } else if (ACT == 2) {
                                    v *= gelu_erf_grad(p.aux[o]);
                                }
                                v += rs_v[h][k];
                                p.out[o] = v;
                            }
                        }
                    }
                }
                __syncthreads();
            }
        } else {
            float sink = 0.f;                                   // keep every accumulator chain alive
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTR; ++nt) sink += acc[mt][nt][0] + acc[mt][nt][1] + acc[mt][nt][2] + acc[mt][nt][3];
            if (sink == 123.456f) p.out[tid] = sink;
        }
        if (!have_next) break;
        it = nx;
    }
}

// tile-rows per workgroup: 3 (6x32 pixel tiles).  4 would amortise the weight stream better but needs 20 more
// accumulator registers than the 128-register budget of a 16-wave workgroup allows (measured: spills, 3x slower).
inline int wino_ntr() { return 3; }

template <int MT, int NTR, int CPC>
inline void conv_wino_launch_c(const ConvArgs& a, unsigned grid, int ipx, int wpx, hipStream_t st) {
    constexpr size_t lds = WinoGeom<NTR, CPC>::LDS_FLOATS * sizeof(float);
    switch (a.act & 0xff) {
        case 1: hipLaunchKernelGGL((conv_wino_kernel<MT, NTR, 1, CPC>), dim3(grid), dim3(WN_THREADS), lds, st, a, ipx, wpx); break;
        case 2: hipLaunchKernelGGL((conv_wino_kernel<MT, NTR, 2, CPC>), dim3(grid), dim3(WN_THREADS), lds, st, a, ipx, wpx); break;
        default: hipLaunchKernelGGL((conv_wino_kernel<MT, NTR, 0, CPC>), dim3(grid), dim3(WN_THREADS), lds, st, a, ipx, wpx);
    }
}

// channels per chunk: 16 (measured: 32-channel chunks = half the barriers, but only one M tile per epilogue pass
// fits in LDS then; 2 % slower end to end).  -DSINDDM_WINO_CPC=2 selects the 32-channel variant for A/B builds.
#ifndef SINDDM_WINO_CPC
#define SINDDM_WINO_CPC 1
#endif
inline int wino_cpc() { return SINDDM_WINO_CPC == 2 ? 2 : 1; }

template <int MT, int NTR>
inline void conv_wino_launch_a(const ConvArgs& a, unsigned grid, int ipx, int wpx, hipStream_t st) {
    if (wino_cpc() == 2 && a.nch3 >= 2) conv_wino_launch_c<MT, NTR, 2>(a, grid, ipx, wpx, st);
    else conv_wino_launch_c<MT, NTR, 1>(a, grid, ipx, wpx, st);
}

template <int MT>
inline void conv_wino_launch_t(const ConvArgs& a, unsigned grid, int ipx, int wpx, int ntr, hipStream_t st) {
    (void)ntr;
    conv_wino_launch_a<MT, 3>(a, grid, ipx, wpx, st);
}

