// 3x3 weight gradient in the Winograd F(2x2,3x3) domain on the fp32 matrix cores.
//
//   forward:   Y_tile = A^T [ sum_ci U_xi(co,ci) (.) V_xi(ci,tile) ] A,   U = G g G^T,  V = B^T d B
//   gradient:  dU_xi(co,ci) = sum_tiles dM_xi(co,tile) * V_xi(ci,tile),   dM = A dY A^T (4x4 from the 2x2 dY block)
//              dg = G^T dU G
// 16 frequency GEMMs with K = number of 2x2 tiles = pixels/4 instead of 9 GEMMs with K = pixels: 2.25x fewer MFMAs
// than the direct weight gradient (autograd of the convs of reference SinDDM/models.py:63,65 via functions.py:97-102).
//
// Mapping: a 16-wave workgroup per CU owns an (80 co x 48 ci) slab and a strided subset of the 4x16-pixel tiles
// (2 x 8 2x2-tiles = 4 k-steps of v_mfma_f32_16x16x4_f32 each; a dY plane is ONE and an input halo plane TWO
// whole-wave DMA instructions); WAVE xi owns frequency xi: 5x3 accumulator tiles.
// Both operands are built on the fly from LDS: dM_xi from the dY tile (<= 4 signed reads), V_xi from the input
// halo tile (4 signed reads), like the forward kernel's V.  The tiles are LDS-DMA'd with the pixel columns
// DE-INTERLEAVED (a row is stored [even columns | odd columns]; the gather happens in the per-lane global offsets),
// so the four pixels of tile column c sit at index c of their half-row and the k-lanes read consecutive floats:
// plane strides == 2 (mod 32) keep every operand read bank-conflict free.
// The tile loop is software-pipelined over a three-stage LDS ring (see the loop) and every slab gets a number of pixel
// splits proportional to its cost (WwMap).
// Epilogue (once per workgroup): the 16 frequencies of an M tile meet in LDS, thread (co, ci) applies G^T . G and
// adds its 9 taps into the [co][tap][ci] staging slab with coalesced atomics; the bias gradient falls out of
// frequency (1,1), whose dM is the plain sum of the 2x2 block.
#pragma once
#include "common.h"
#include <type_traits>

namespace sinddm {

// compile-time timing ablations (-DSINDDM_WW_ABL=bits; results are wrong): 1 no DMA traffic after the first tiles (instructions still issue, empty descriptor), 2 no operand reads, 4 no DMA instructions
#ifndef SINDDM_WW_ABL
#define SINDDM_WW_ABL 0
#endif

constexpr int WW_STAGES = 3;                         // LDS ring: tile T is read while T+1 is visible and T+2 is in flight
#ifndef WW_SPLIT_C0
#define WW_SPLIT_C0 1      // fixed per-tile cost of a slab (DMA issue, barrier) in n-tile units.  Training step, same box:
                             // one split count for all slabs 45.9 ms; cost = n-tiles + 8 / 4 / 2: 45.0 / 44.6 / 44.3; pipelined loop with
                             // + 4 / 2 / 1 / 0: 43.3 / 42.9 / 42.65 / 47.6 (a 16-channel slab's tile is bound by its 96 DMA instructions)
#endif
constexpr int WW_THREADS = 1024;
constexpr int WW_CO = 80, WW_CI = 48;
constexpr int WW_TW = 16, WW_TH = 4;                 // pixel tile = 2 x 8 2x2-tiles (4 k-steps of 4 tile columns)
constexpr int WW_PSO = 4 * 16 + 2;                   // dY plane: [4 rows][even 8 | odd 8] + pad        (== 2 mod 32)
constexpr int WW_XR = 18;                            // input row incl. halo: [even 9 | odd 9]
constexpr int WW_XP = 6 * WW_XR;                     // input plane: 6 rows x 18 = 108 floats = 2 wave instructions
constexpr int WW_PSI = 130;                          // plane stride                                     (== 2 mod 32)
constexpr int WW_BUF = WW_CO * WW_PSO + WW_CI * WW_PSI;   // floats per stage (11520 = 45 KB)
constexpr int WW_ESTRIDE = WW_CI + 1;                // epilogue exchange [xi][16 co][48 ci + 1]

// Workgroup -> (slab, pixel split) table, built on the host per launch (ww_build_map): a slab with fewer ci tiles
// (160 = 48+48+48+16, 80 = 48+32) gets proportionally fewer pixel splits, so every workgroup carries the same number of
// MFMAs (with one S for all slabs the 16-channel slab's workgroups idled for two thirds of the launch: 0.83 of the CUs'
// time used).  Entry = slab << 16 | split; workgroups are ordered by the start of their tile range and dealt to the XCDs
// in contiguous runs, so the workgroups that stream the same dY / input tiles at the same time share an L2.
constexpr int WW_MAXWG = 512, WW_MAXSLAB = 256;       // (2.5 KB of kernel arguments)
struct WwMap {
    unsigned wg[WW_MAXWG];
    unsigned short S[WW_MAXSLAB];
};

struct WwArgs {
    const float* dout;   // [B][Cout][H][W]
    const float* in;     // [B][Cin][H][W]
    float* gw;           // staging slab [Cout][9][Cin]  (+=, atomics)
    float* gb;           // [Cout] (+=) or nullptr
    int B, H, W, Cin, Cout;
    int coblks, ciblks;
    int tilesX, tilesY, ntiles;
    const float* amax_d;   // wgrad_wh.h only: per-sample running max |dout| / |in| ([B][AMAX_STRIDE] floats each)
    const float* amax_i;
    WwMap map;
};

// returns the number of workgroups, or 0 when the table cannot describe the launch (more slabs than it holds: the caller
// then takes the direct weight-gradient kernels -- a slower path, not an error)
inline int ww_build_map(WwArgs& w, int ncu) {
    const int slabs = w.coblks * w.ciblks;
    if (slabs > WW_MAXSLAB || ncu < 1) return 0;
    if (ncu > WW_MAXWG) ncu = WW_MAXWG;
    int nt[WW_MAXSLAB], S[WW_MAXSLAB], units = 0;
    for (int q = 0; q < slabs; ++q) {
        const int cib = q % w.ciblks;
        const int nci = w.Cin - cib * WW_CI < WW_CI ? w.Cin - cib * WW_CI : WW_CI;
        nt[q] = ((nci + 15) >> 4) + WW_SPLIT_C0;     // cost of a tile: n-tiles + fixed part
        units += nt[q];
    }
    const int cap = w.ntiles < 0xffff ? w.ntiles : 0xffff;
    int total = 0;
    for (int q = 0; q < slabs; ++q) {
        int v = (int)((long long)ncu * nt[q] / units);
        if (v < 1) v = 1;
        if (v > cap) v = cap;
        S[q] = v;
        total += v;
    }
    // leftover CUs go, one at a time, to the slab whose workgroups carry the most work
    while (total < ncu) {
        int best = -1;
        for (int q = 0; q < slabs; ++q)
            if (S[q] < cap && (best < 0 || (long long)nt[q] * S[best] > (long long)nt[best] * S[q])) best = q;
        if (best < 0) break;
        ++S[best];
        ++total;
    }
    if (total > WW_MAXWG) return 0;
    // merge the slabs' splits by the start of their tile range s / S (ties: slab order)
    unsigned order[WW_MAXWG];
    int cur[WW_MAXSLAB] = {0};
    for (int n = 0; n < total; ++n) {
        int best = -1;
        for (int q = 0; q < slabs; ++q) {
            if (cur[q] >= S[q]) continue;
            if (best < 0 || (long long)cur[q] * S[best] < (long long)cur[best] * S[q]) best = q;
        }
        order[n] = (unsigned)best << 16 | (unsigned)cur[best];
        ++cur[best];
    }
    // workgroup id -> XCD id & 7; XCD x owns the contiguous run of the order that starts where the runs of XCDs < x end
    int start[9];
    start[0] = 0;
    for (int x = 0; x < 8; ++x) start[x + 1] = start[x] + (total - x + 7) / 8;
    for (int id = 0; id < total; ++id) w.map.wg[id] = order[start[id & 7] + (id >> 3)];
    for (int q = 0; q < slabs; ++q) w.map.S[q] = (unsigned short)S[q];
    return total;
}

// ---- epilogue of both weight-gradient kernels: dg = G^T dU G per (co, ci), one 16-channel M tile at a time through LDS ----
__device__ __forceinline__ void ww_epilogue(const WwArgs& p, float* smem, f32x4 (&acc)[5][3], float (&bsum)[5], bool dobias,
                                            int xi, int lane, int cb, int ci0) {
    const int l16 = lane & 15, kq = lane >> 4;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (LDS-DMA still in flight into the ring the exchange reuses)
    // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]];   t[i][b] = sum_j dU[i][j] G[j][b];   dg[a][b] = sum_i G[i][a] t[i][b]
    float* sE = smem;              // [xi][16 co][WW_ESTRIDE]  (12.5 K floats, inside the stage buffers)
#pragma unroll
    for (int mt = 0; mt < 5; ++mt) {          // (fully unrolled: a dynamic index would push acc[] into scratch)
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                // C layout: col = lane&15 -> ci (N), row = (lane>>4)*4 + r -> co (M)
                sE[(xi * 16 + kq * 4 + r) * WW_ESTRIDE + nt * 16 + l16] = acc[mt][nt][r];
        __syncthreads();
        // thread (co = wave, ci = lane)
        const int ci = ci0 + lane;
        const int co = cb * WW_CO + mt * 16 + xi;
        if (lane < WW_CI && ci < p.Cin) {
            float u[16];
#pragma unroll
            for (int f = 0; f < 16; ++f) u[f] = sE[(f * 16 + xi) * WW_ESTRIDE + lane];
            float t[4][3];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float h1 = 0.5f * (u[i * 4 + 1] + u[i * 4 + 2]), h2 = 0.5f * (u[i * 4 + 1] - u[i * 4 + 2]);
                t[i][0] = u[i * 4 + 0] + h1;
                t[i][1] = h2;
                t[i][2] = h1 + u[i * 4 + 3];
            }
            float* g = p.gw + ((size_t)co * 9) * p.Cin + ci;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float h1 = 0.5f * (t[1][b] + t[2][b]), h2 = 0.5f * (t[1][b] - t[2][b]);
                atomicAdd(g + (size_t)(0 * 3 + b) * p.Cin, t[0][b] + h1);
                atomicAdd(g + (size_t)(1 * 3 + b) * p.Cin, h2);
                atomicAdd(g + (size_t)(2 * 3 + b) * p.Cin, h1 + t[3][b]);
            }
        }
    }
    if (dobias) {
#pragma unroll
        for (int mt = 0; mt < 5; ++mt) {
            float v = bsum[mt];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (kq == 0) atomicAdd(&p.gb[cb * WW_CO + mt * 16 + l16], v);
        }
    }
}

__global__ __launch_bounds__(WW_THREADS) void wgrad_wino_kernel(WwArgs p) {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const unsigned ent = p.map.wg[blockIdx.x];
    const int q = (int)(ent >> 16);                  // slab
    const int s = (int)(ent & 0xffffu);              // pixel-split index of this slab
    const int S = p.map.S[q];
    const int cb = q / p.ciblks, cib = q - cb * p.ciblks;

    const int tid = threadIdx.x, lane = tid & 63;
    const int xi = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave = Winograd frequency (i, j)
    const int l16 = lane & 15, kq = lane >> 4;
    const int H = p.H, W = p.W, HW = H * W;
    const int tpi = p.tilesX * p.tilesY;
    const int ci0 = cib * WW_CI;
    const int nci = min(WW_CI, p.Cin - ci0);
    const int nnt = (nci + 15) >> 4;                 // 16-channel ci tiles that exist in this slab (wave-uniform)

    // ---- LDS-DMA of one pixel tile (buffer bounds check zero-fills everything outside the image / channel range) ----
    // dY: 80 planes of 64 floats, one whole-wave instruction each (5 per wave).  Input: 48 halo planes of 108 floats,
    // two instructions each; wave w always moves half q = w & 1 of channels (w >> 1) + 8 k (6 per wave), so one
    // per-lane offset register per operand describes the gather (recomputed per tile for the image borders).
    constexpr unsigned OOB = 0x40000000u;
    struct TileAddr {
        __amdgpu_buffer_rsrc_t rd, ri;
        unsigned loff_d, loff_i;
    };
    const int xq = xi & 1;
    auto tile_addr = [&](int b, int ty, int tx) {
        TileAddr ta;
        const int y0 = ty * WW_TH, x0 = tx * WW_TW;
        {   // dY: lane -> (row, half, idx): pixel column 2*idx + half
            const int r = lane >> 4, rem = lane & 15;
            const int col = 2 * (rem & 7) + (rem >> 3);
            ta.loff_d = ((y0 + r < H) && (x0 + col < W)) ? (unsigned)((y0 + r) * W + x0 + col) * 4u : OOB;
        }
        {   // input halo plane element e = 64 q + lane -> (row, half, idx): halo column 2*idx + half
            const int e = xq * 64 + lane;
            const int r = (e * 3641) >> 16;                    // e / 18 for e < 128
            const int rem = e - r * WW_XR;
            const int hc = (rem < 9) ? 2 * rem : 2 * (rem - 9) + 1;
            const int gy = y0 - 1 + r, gx = x0 - 1 + hc;
            ta.loff_i = (e < WW_XP && gy >= 0 && gy < H && gx >= 0 && gx < W) ? (unsigned)(gy * W + gx) * 4u : OOB;
        }
        ta.rd = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.dout + ((size_t)b * p.Cout + (size_t)cb * WW_CO) * HW), 0, WW_CO * HW * 4, 0x00020000);
        ta.ri = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.in + ((size_t)b * p.Cin + ci0) * HW), 0, nci * HW * 4, 0x00020000);
        return ta;
    };
    // 11 instructions per wave in 6 groups: group g < 5 carries dY plane xi + 16 g; every group carries one input
    // half-plane (channel (xi >> 1) + 8 g)
    auto issue_group = [&](const TileAddr& ta, float* buf, int g) {
        if (g < 5) {
            const int col = xi + 16 * g;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ta.rd, (lds_ptr)(buf + col * WW_PSO), 4,
                                                     (int)(ta.loff_d + (unsigned)(col * HW) * 4u), 0, 0, 0);
        }
        const int cil = (xi >> 1) + 8 * g;
        const unsigned coff = (cil < nci) ? (unsigned)(cil * HW) * 4u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ta.ri, (lds_ptr)(buf + WW_CO * WW_PSO + cil * WW_PSI + xq * 64), 4,
                                                 (int)(ta.loff_i + coff), 0, 0, 0);
    };

    f32x4 acc[5][3];
#pragma unroll
    for (int mt = 0; mt < 5; ++mt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const bool dobias = p.gb != nullptr && cib == 0 && xi == 5;      // frequency (1,1): dM = sum of the 2x2 block

    // pixel split s owns a CONTIGUOUS tile range, walked with an incremental (image, tile row, tile column) counter:
    // two integer divisions per tile and wave were ~10 % of the loop
    const int per = (p.ntiles + S - 1) / S;
    const int t_begin = s * per, t_end = min(p.ntiles, t_begin + per);
    int nb = t_begin / tpi;                           // coordinates of the NEXT tile to prefetch
    int nty = (t_begin - nb * tpi) / p.tilesX;
    int ntx = t_begin - nb * tpi - nty * p.tilesX;
    auto advance = [&]() {
        if (++ntx == p.tilesX) {
            ntx = 0;
            if (++nty == p.tilesY) { nty = 0; ++nb; }
        }
    };
    // ---- software-pipelined tile loop over a THREE-stage LDS ring ----
    // Tile T is DMA'd during tile T-2 and made visible by the barrier at the top of tile T-1, so the operands of k-step
    // (T, 0) can be read during k-step (T-1, 3): every k-step issues the LDS reads of the NEXT k-step's operands in front
    // of its own 15 MFMAs and combines them behind -- no k-step waits for LDS, and no branch splits the loop body (the
    // n-tile count of the slab is a template parameter, the DMA of a tile that does not exist goes through an empty
    // descriptor, the bias sum is compiled into wave 5's copy only).
    float* b0 = smem;
    float* b1 = smem + WW_BUF;
    float* b2 = smem + 2 * WW_BUF;
    auto tile_addr_or_null = [&](bool exists) {
        TileAddr ta = tile_addr(nb, nty, ntx);
        if (!exists) {      // zero records: every lane out of range, nothing is fetched
            ta.rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dout), 0, 0, 0x00020000);
            ta.ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, 0, 0x00020000);
        } else {
            advance();
        }
        return ta;
    };
    {
        const TileAddr t0 = tile_addr_or_null(t_begin < t_end);
#pragma unroll
        for (int g = 0; g < 6; ++g) issue_group(t0, b0, g);
        const TileAddr t1 = tile_addr_or_null(t_begin + 1 < t_end);
#pragma unroll
        for (int g = 0; g < 6; ++g) issue_group(t1, b1, g);
    }
    auto run_tiles = [&](auto xi_c, auto nnt_c) {
        constexpr int XI = decltype(xi_c)::value;
        constexpr int NNT = decltype(nnt_c)::value;
        constexpr int fi = XI >> 2, fj = XI & 3;
        // dM = A dY A^T:  A rows  0: +y0   1: +y0 +y1   2: +y0 -y1   3: -y1   (same for the columns: even / odd pixel).
        // Only the non-zero terms are read: 1, 2 or 4 of them (2.25 on average over the 16 frequencies).
        constexpr int nr = (fi == 1 || fi == 2) ? 2 : 1, nc = (fj == 1 || fj == 2) ? 2 : 1;
        constexpr int arow = fi == 3 ? 1 : 0, acol = fj == 3 ? 1 : 0;            // first (or only) row / half read
        constexpr float sr0 = fi == 3 ? -1.f : 1.f, sr1 = fi == 2 ? -1.f : 1.f;   // sign of the first / second row term
        constexpr float sc0 = fj == 3 ? -1.f : 1.f, sc1 = fj == 2 ? -1.f : 1.f;
        constexpr int aterms = nr * nc;                                           // 1, 2 or 4
        // V = B^T d B:  rows  0: +d0 -d2   1: +d1 +d2   2: -d1 +d2   3: +d1 -d3   (patch index p = 2*idx + half)
        constexpr int pa0 = fi == 0 ? 0 : 1, pa1 = fi == 3 ? 3 : 2;
        constexpr int pb0 = fj == 0 ? 0 : 1, pb1 = fj == 3 ? 3 : 2;
        constexpr float sa0 = fi == 2 ? -1.f : 1.f, sa1 = (fi == 0 || fi == 3) ? -1.f : 1.f;
        constexpr float sb0 = fj == 2 ? -1.f : 1.f, sb1 = (fj == 0 || fj == 3) ? -1.f : 1.f;
        constexpr float b00 = sa0 * sb0, b01 = sa0 * sb1, b10 = sa1 * sb0, b11 = sa1 * sb1;
        const int abase = l16 * WW_PSO + kq;                                      // plane = [row][half][idx]
        constexpr int oa0 = arow * 16 + acol * 8, oa1 = oa0 + (nc == 2 ? 8 : 16), oa2 = oa0 + 16, oa3 = oa0 + 24;
        constexpr float as0 = sr0 * sc0, as1 = (nc == 2) ? sr0 * sc1 : sr1 * sc0, as2 = sr1 * sc0, as3 = sr1 * sc1;
        const int bbase = WW_CO * WW_PSO + l16 * WW_PSI + kq;
        constexpr int ob00 = pa0 * WW_XR + (pb0 & 1) * 9 + (pb0 >> 1), ob01 = pa0 * WW_XR + (pb1 & 1) * 9 + (pb1 >> 1);
        constexpr int ob10 = pa1 * WW_XR + (pb0 & 1) * 9 + (pb0 >> 1), ob11 = pa1 * WW_XR + (pb1 & 1) * 9 + (pb1 >> 1);
        // operands of k-step j of the tile in `buf`: tile row j>>1, tile columns 4*(j&1) + kq
        auto build = [&](const float* buf, int j, float (&a)[5], float (&bv)[NNT]) {
            const float* qd = buf + abase + (j >> 1) * 32 + (j & 1) * 4;
#pragma unroll
            for (int mt = 0; mt < 5; ++mt) {
                const float* q = qd + mt * 16 * WW_PSO;
                if (SINDDM_WW_ABL & 2) a[mt] = as0 * (float)(j + mt);
                else if constexpr (aterms == 4) a[mt] = as0 * q[oa0] + as1 * q[oa1] + as2 * q[oa2] + as3 * q[oa3];
                else if constexpr (aterms == 2) a[mt] = as0 * q[oa0] + as1 * q[oa1];
                else a[mt] = as0 * q[oa0];
            }
#pragma unroll
            for (int nt = 0; nt < NNT; ++nt) {
                const float* qx = buf + bbase + nt * 16 * WW_PSI + (j >> 1) * 36 + (j & 1) * 4;
                bv[nt] = (SINDDM_WW_ABL & 2) ? b00 * (float)(j - nt)
                                             : b00 * qx[ob00] + b01 * qx[ob01] + b10 * qx[ob10] + b11 * qx[ob11];
            }
        };
        float a[2][5], bv[2][NNT];
        dma_barrier();                // tile t_begin landed
        build(b0, 0, a[0], bv[0]);
        for (int tile = t_begin; tile < t_end; ++tile) {
            dma_barrier();            // tile + 1 landed and is visible; every wave is done with the reads of tile - 1
            const TileAddr ta = (SINDDM_WW_ABL & 1) ? tile_addr_or_null(false) : tile_addr_or_null(tile + 2 < t_end);
#pragma unroll
            for (int j = 0; j < 4; ++j) {              // k-step: tile columns 4j .. 4j+3 (lane group kq)
                if (!(SINDDM_WW_ABL & 4)) {
                    issue_group(ta, b2, j);
                    if (j < 2) issue_group(ta, b2, j + 4);
                }
                if (j < 3) build(b0, j + 1, a[(j + 1) & 1], bv[(j + 1) & 1]);
                else build(b1, 0, a[0], bv[0]);        // (behind the last tile: values nobody uses)
                if constexpr (XI == 5) {               // frequency (1,1): dM = sum of the 2x2 block = the bias gradient
#pragma unroll
                    for (int mt = 0; mt < 5; ++mt) bsum[mt] += a[j & 1][mt];
                }
#pragma unroll
                for (int nt = 0; nt < NNT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 5; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j & 1][mt], bv[j & 1][nt], acc[mt][nt], 0, 0, 0);
            }
            float* t = b0; b0 = b1; b1 = b2; b2 = t;
        }
    };
    auto run_nnt = [&](auto xi_c) {
        if (nnt == 3) run_tiles(xi_c, std::integral_constant<int, 3>{});
        else if (nnt == 2) run_tiles(xi_c, std::integral_constant<int, 2>{});
        else run_tiles(xi_c, std::integral_constant<int, 1>{});
    };
    switch (xi) {
#define WW_CASE(n) case n: run_nnt(std::integral_constant<int, n>{}); break;
        WW_CASE(0) WW_CASE(1) WW_CASE(2) WW_CASE(3) WW_CASE(4) WW_CASE(5) WW_CASE(6) WW_CASE(7)
        WW_CASE(8) WW_CASE(9) WW_CASE(10) WW_CASE(11) WW_CASE(12) WW_CASE(13) WW_CASE(14) default: run_nnt(std::integral_constant<int, 15>{});
#undef WW_CASE
    }

    ww_epilogue(p, smem, acc, bsum, dobias, xi, lane, cb, ci0);
}

// ---------------------------------------------------------------------------------------------------------------------
// Wide variant for W % 4 == 0 (every row of every plane 16-byte aligned, a 16-byte column group never straddles the right
// border): the same slabs, frequencies, tile ring and epilogue, but the tiles are DMA'd with 16-byte LDS-DMA instructions
// -- 47 per tile and workgroup instead of 176 (the dword version spends 1.3 ms of a 42.7 ms training step ISSUING them:
// ablation in DESIGN.md section 5.0) -- and the operands are read with conflict-free ds_read_b64.
//
// LDS image of a tile (floats): one DMA instruction fills one 1-KB block = 64 16-byte slots, slot = lane.
//   dY:    block (m-tile mt, row r) at (4 mt + r) * 256; slot 16 q + p holds plane 16 mt + p, row r, columns 4q .. 4q+3.
//   input: the halo plane is 6 rows x 6 column groups (columns x0-4 .. x0+19) = 36 groups e = 6 hr + k; slot 16 e + p of
//          n-tile nt's 9 blocks (at WX_DY + nt * 2304) holds plane 16 nt + p, group e.
// Every 16-byte group of plane p sits at a slot == p (mod 16), so a ds_read_b64 whose 32-lane halves are 16 planes x
// 2 column pairs (the k-lanes kq, kq+1: the two halves of one group, or the upper half of one group and the lower half of
// the next) touches each of the 64 banks once.
// A lane reads the 2x2 dY block of tile column c = 4 (j & 1) + kq as two float2 (rows), the input patch columns
// 2c+3+pb (counted from x0-4) as the column pairs c+1 .. c+3 of the two patch rows its frequency needs.
// (volatile: the load-store optimizer would fuse two ds_read_b64 of one base register into ds_read2_b64 / ds_read2st64_b64,
// whose 16-lane groups and 32-bank modulus turn the 16-byte slot layout into 2-way conflicts; waitcnt tracking is unaffected)
__device__ __forceinline__ f32x2 ww_ld2(const float* q) {
    typedef const volatile __attribute__((address_space(3))) f32x2* lds_v2;
    return *(lds_v2)q;            // (q points into the dynamic LDS array)
}
constexpr int WX_DY = 20 * 256;
constexpr int WX_IN = 27 * 256;
constexpr int WX_BUF = WX_DY + WX_IN;                // 12032 floats = 47 KB per stage, 141 KB for the ring
constexpr int WX_NDMA = 47;

__global__ __launch_bounds__(WW_THREADS) void wgrad_wino_wide_kernel(WwArgs p) {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const unsigned ent = p.map.wg[blockIdx.x];
    const int q = (int)(ent >> 16);                  // slab
    const int s = (int)(ent & 0xffffu);              // pixel-split index of this slab
    const int S = p.map.S[q];
    const int cb = q / p.ciblks, cib = q - cb * p.ciblks;

    const int tid = threadIdx.x, lane = tid & 63;
    const int xi = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave = Winograd frequency (i, j)
    const int l16 = lane & 15, kq = lane >> 4;
    const int H = p.H, W = p.W, HW = H * W;
    const int tpi = p.tilesX * p.tilesY;
    const int ci0 = cib * WW_CI;
    const int nci = min(WW_CI, p.Cin - ci0);
    const int nnt = (nci + 15) >> 4;

    // ---- the (at most) three DMA instructions of this wave: idx = xi, xi + 16, xi + 32 of the 47; per lane the plane and
    // the (row, column) of its 16-byte group relative to the tile origin ----
    constexpr unsigned OOB = 0x40000000u;
    unsigned dloc[3];      // byte offset of the lane's group from the tile origin (plane included)
    int dyx[3];            // its row + 1 (0 .. 5) | (column + 4) << 8 (0 .. 20) relative to the tile origin
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
        const int idx = min(xi + 16 * sl, WX_NDMA - 1);          // (wave 15's third slot repeats instruction 46: no branch)
        int pl, dy, dx;
        if (idx < 20) {
            pl = (idx >> 2) * 16 + l16; dy = idx & 3; dx = 4 * kq;
        } else {
            const int ii = idx - 20;
            const int nt = ii / 9, g = ii - nt * 9;
            const int e = 4 * g + kq;
            const int hr = e / 6, k = e - hr * 6;
            pl = nt * 16 + l16; dy = hr - 1; dx = 4 * k - 4;
        }
        dloc[sl] = (unsigned)(pl * HW + dy * W + dx) * 4u;
        dyx[sl] = (dy + 1) | (dx + 4) << 8;
    }
    struct TileAddr {
        __amdgpu_buffer_rsrc_t rd, ri;
        int y0, x0;
    };
    const int per = (p.ntiles + S - 1) / S;
    const int t_begin = s * per, t_end = min(p.ntiles, t_begin + per);
    int nb = t_begin / tpi;                           // coordinates of the NEXT tile to prefetch
    int nty = (t_begin - nb * tpi) / p.tilesX;
    int ntx = t_begin - nb * tpi - nty * p.tilesX;
    auto next_tile = [&](bool exists) {
        TileAddr ta;
        ta.y0 = nty * WW_TH; ta.x0 = ntx * WW_TW;
        if (exists) {
            ta.rd = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.dout + ((size_t)nb * p.Cout + (size_t)cb * WW_CO) * HW), 0, WW_CO * HW * 4, 0x00020000);
            ta.ri = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.in + ((size_t)nb * p.Cin + ci0) * HW), 0, nci * HW * 4, 0x00020000);
            if (++ntx == p.tilesX) {
                ntx = 0;
                if (++nty == p.tilesY) { nty = 0; ++nb; }
            }
        } else {            // zero records: every lane out of range, nothing is fetched
            ta.rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dout), 0, 0, 0x00020000);
            ta.ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, 0, 0x00020000);
        }
        return ta;
    };
    auto issue = [&](const TileAddr& ta, float* buf, int sl) {
        const int idx = min(xi + 16 * sl, WX_NDMA - 1);
        const int gy = ta.y0 - 1 + (dyx[sl] & 0xff), gx = ta.x0 - 4 + (dyx[sl] >> 8);
        const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        // (arithmetic instead of a select: the compiler turned `ok ? offset : OOB` into an exec-masked branch that cut the
        // tile loop's body into three scheduling regions)
        const unsigned voff = (dloc[sl] + (unsigned)(ta.y0 * W + ta.x0) * 4u) | (ok ? 0u : OOB);
        const __amdgpu_buffer_rsrc_t rs = idx < 20 ? ta.rd : ta.ri;       // (scalar select, no branch in the tile loop)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(buf + idx * 256), 16, (int)voff, 0, 0, 0);
    };

    f32x4 acc[5][3];
#pragma unroll
    for (int mt = 0; mt < 5; ++mt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const bool dobias = p.gb != nullptr && cib == 0 && xi == 5;      // frequency (1,1): dM = sum of the 2x2 block

    float* b0 = smem;
    float* b1 = smem + WX_BUF;
    float* b2 = smem + 2 * WX_BUF;
    {
        const TileAddr t0 = next_tile(t_begin < t_end);
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) issue(t0, b0, sl);
        const TileAddr t1 = next_tile(t_begin + 1 < t_end);
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) issue(t1, b1, sl);
    }
    auto run_tiles = [&](auto xi_c, auto nnt_c) {
        constexpr int XI = decltype(xi_c)::value;
        constexpr int NNT = decltype(nnt_c)::value;
        constexpr int fi = XI >> 2, fj = XI & 3;
        // dM = A dY A^T on the 2x2 block (rows r0, r1; a float2 = the two columns):
        //   columns  fj 0: +x   1: x + y   2: x - y   3: -y ;   rows  fi 0: +r0   1: r0 + r1   2: r0 - r1   3: -r1
        constexpr bool need_r0 = fi != 3, need_r1 = fi != 0;
        // V = B^T d B:  rows  0: +d0 -d2   1: +d1 +d2   2: -d1 +d2   3: +d1 -d3   (same for the columns)
        constexpr int pa0 = fi == 0 ? 0 : 1, pa1 = fi == 3 ? 3 : 2;
        constexpr float sa0 = fi == 2 ? -1.f : 1.f, sa1 = (fi == 0 || fi == 3) ? -1.f : 1.f;
        constexpr float sb0 = fj == 2 ? -1.f : 1.f, sb1 = (fj == 0 || fj == 3) ? -1.f : 1.f;
        // patch columns pb0, pb1 = columns 2c+3+pb of the staged plane: pairs c+1 (.y) and c+2 (.y) for fj = 0, the pair
        // c+2 for fj = 1, 2, pairs c+2 (.x) and c+3 (.x) for fj = 3
        constexpr int d0 = fj == 0 ? 1 : 2, d1 = fj == 0 ? 2 : 3;
        constexpr bool two_pairs = fj == 0 || fj == 3;
        const int abase = 4 * l16 + 64 * (kq >> 1) + 2 * (kq & 1);
        const int vb0 = WX_DY + 4 * l16 + 64 * ((kq + d0) >> 1) + 2 * ((kq + d0) & 1);
        const int vb1 = WX_DY + 4 * l16 + 64 * ((kq + d1) >> 1) + 2 * ((kq + d1) & 1);
        // rows first (both columns of a pair at once: v_pk_add_f32), then the columns
        auto colmix = [&](f32x2 v, float sgn) {
            return fj == 0 ? sgn * v.x : fj == 1 ? sgn * v.x + sgn * v.y : fj == 2 ? sgn * v.x - sgn * v.y : -sgn * v.y;
        };
        // operands of k-step j of the tile in `buf` (tile row j>>1, tile columns 4*(j&1) + kq), in two phases each: the raw
        // ds_read_b64 (issued a k-step ahead, in front of MFMAs) and the combination (behind them).  The phases are fenced
        // with sched_barrier: left alone, the scheduler sinks every read next to its use to save registers, and all four
        // waves of a SIMD then sit in the same s_waitcnt while the matrix pipe idles.
        constexpr int NRA = (need_r0 && need_r1) ? 2 : 1;
        constexpr int NRB = two_pairs ? 4 : 2;
        auto load_a = [&](const float* buf, int j, f32x2 (&ra)[5][NRA]) {
            const int tr = j >> 1, jc = j & 1;
#pragma unroll
            for (int mt = 0; mt < 5; ++mt) {
                if (SINDDM_WW_ABL & 2) continue;
                const float* qd = buf + abase + (mt * 4 + 2 * tr) * 256 + 128 * jc;
                if constexpr (NRA == 2) { ra[mt][0] = ww_ld2(qd); ra[mt][1] = ww_ld2(qd + 256); }
                else ra[mt][0] = ww_ld2(qd + (need_r0 ? 0 : 256));
            }
        };
        auto comb_a = [&](int j, const f32x2 (&ra)[5][NRA], float (&a)[5]) {
#pragma unroll
            for (int mt = 0; mt < 5; ++mt) {
                if (SINDDM_WW_ABL & 2) { a[mt] = (float)(j + mt); continue; }
                if constexpr (NRA == 2) a[mt] = colmix(fi == 1 ? ra[mt][0] + ra[mt][1] : ra[mt][0] - ra[mt][1], 1.f);
                else a[mt] = colmix(ra[mt][0], need_r0 ? 1.f : -1.f);
            }
        };
        auto load_b = [&](const float* buf, int j, f32x2 (&rb)[NNT][NRB]) {
            const int tr = j >> 1, jc = j & 1;
#pragma unroll
            for (int nt = 0; nt < NNT; ++nt) {
                if (SINDDM_WW_ABL & 2) continue;
                const int o0 = nt * 2304 + 64 * (6 * (2 * tr + pa0) + 2 * jc);
                const int o1 = nt * 2304 + 64 * (6 * (2 * tr + pa1) + 2 * jc);
                rb[nt][0] = ww_ld2(buf + vb0 + o0);
                rb[nt][1] = ww_ld2(buf + vb0 + o1);
                if constexpr (two_pairs) {
                    rb[nt][2] = ww_ld2(buf + vb1 + o0);
                    rb[nt][3] = ww_ld2(buf + vb1 + o1);
                }
            }
        };
        auto comb_b = [&](int j, const f32x2 (&rb)[NNT][NRB], float (&bv)[NNT]) {
#pragma unroll
            for (int nt = 0; nt < NNT; ++nt) {
                if (SINDDM_WW_ABL & 2) { bv[nt] = (float)(j - nt); continue; }
                const f32x2 u = sa0 * rb[nt][0] + sa1 * rb[nt][1];
                if constexpr (two_pairs) {
                    const f32x2 v = sa0 * rb[nt][2] + sa1 * rb[nt][3];
                    bv[nt] = fj == 0 ? sb0 * u.y + sb1 * v.y : sb0 * u.x + sb1 * v.x;
                } else {
                    bv[nt] = sb0 * u.x + sb1 * u.y;
                }
            }
        };
        auto mfma_nt = [&](int nt, const float (&a)[5], const float (&bv)[NNT]) {
#pragma unroll
            for (int mt = 0; mt < 5; ++mt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], bv[nt], acc[mt][nt], 0, 0, 0);
        };
        float a[2][5], bv[2][NNT];
        f32x2 ra[5][NRA], rb[NNT][NRB];
        dma_barrier();                // tile t_begin landed
        load_a(b0, 0, ra); load_b(b0, 0, rb);
        comb_a(0, ra, a[0]); comb_b(0, rb, bv[0]);
        for (int tile = t_begin; tile < t_end; ++tile) {
            dma_barrier();            // tile + 1 landed and is visible; every wave is done with the reads of tile - 1
            const TileAddr ta = next_tile(tile + 2 < t_end && !(SINDDM_WW_ABL & 1));
#pragma unroll
            for (int j = 0; j < 4; ++j) {              // k-step: tile columns 4j .. 4j+3 (lane group kq)
                const float* nbuf = j < 3 ? b0 : b1;   // next k-step's tile (behind the last tile: values nobody uses)
                const int nj = (j + 1) & 3;
                if (j < 3 && !(SINDDM_WW_ABL & 4)) issue(ta, b2, j);
                load_a(nbuf, nj, ra);
                if (NNT == 1) load_b(nbuf, nj, rb);
                if constexpr (XI == 5) {               // frequency (1,1): dM = sum of the 2x2 block = the bias gradient
#pragma unroll
                    for (int mt = 0; mt < 5; ++mt) bsum[mt] += a[j & 1][mt];
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma_nt(0, a[j & 1], bv[j & 1]);
                __builtin_amdgcn_sched_barrier(0);
                comb_a(nj, ra, a[(j + 1) & 1]);
                if (NNT > 1) {
                    load_b(nbuf, nj, rb);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int nt = 1; nt < NNT; ++nt) mfma_nt(nt, a[j & 1], bv[j & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                comb_b(nj, rb, bv[(j + 1) & 1]);
            }
            float* t = b0; b0 = b1; b1 = b2; b2 = t;
        }
    };
    auto run_nnt = [&](auto xi_c) {
        if (nnt == 3) run_tiles(xi_c, std::integral_constant<int, 3>{});
        else if (nnt == 2) run_tiles(xi_c, std::integral_constant<int, 2>{});
        else run_tiles(xi_c, std::integral_constant<int, 1>{});
    };
    switch (xi) {
#define WW_CASE(n) case n: run_nnt(std::integral_constant<int, n>{}); break;
        WW_CASE(0) WW_CASE(1) WW_CASE(2) WW_CASE(3) WW_CASE(4) WW_CASE(5) WW_CASE(6) WW_CASE(7)
        WW_CASE(8) WW_CASE(9) WW_CASE(10) WW_CASE(11) WW_CASE(12) WW_CASE(13) WW_CASE(14) default: run_nnt(std::integral_constant<int, 15>{});
#undef WW_CASE
    }
    ww_epilogue(p, smem, acc, bsum, dobias, xi, lane, cb, ci0);
}

}  // namespace sinddm
