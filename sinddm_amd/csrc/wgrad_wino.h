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
constexpr int WW_MAXWG = 512, WW_MAXSLAB = 64;
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
    WwMap map;
};

// returns the number of workgroups, or 0 when the table cannot describe the launch (caller reports BADSHAPE)
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
        __syncthreads();              // tile t_begin landed (vmcnt(0) is part of the barrier)
        build(b0, 0, a[0], bv[0]);
        for (int tile = t_begin; tile < t_end; ++tile) {
            __syncthreads();          // tile + 1 landed and is visible; every wave is done with the reads of tile - 1
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

    // ---- epilogue: dg = G^T dU G per (co, ci), one 16-channel M tile at a time through LDS ----
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

}  // namespace sinddm
