// 3x3 weight gradient in the Winograd F(2x2,3x3) domain on the gfx950 BINARY16 matrix pipe, fp32-equivalent: the scheme of
// conv_wh.h applied to wgrad_wino_wide_kernel (wgrad_wino.h has the algebra, the slabs, the launch table, the LDS image of a
// tile, its three-stage LDS-DMA ring and the epilogue -- all reused).  Both operands of the 16 frequency GEMMs
//     dU_xi(co, ci) = sum over 2x2 tiles of  dM_xi(co, tile) * V_xi(ci, tile),      dM = A dY A^T,  V = B^T d B
// are activations, so both are transformed (+-1 sums: exact in fp32 up to the usual rounding), scaled by an exact power of
// two and split into two binary16 pieces (hi = rn16(a), lo = rn16(a - hi): 22 significand bits) IN the kernel, and every
// product runs as all four terms on v_mfma_f32_16x16x32_f16 with fp32 accumulation.
//
// One LDS tile (4 x 16 pixels = 2 x 8 2x2-tiles) is ONE k-step: K = 32 = 16 tiles x {hi, lo}.  Lane (l16, kg) owns tile row
// kg & 1 and tile columns 4 (kg >> 1) .. + 3 of channel l16: it reads the 16-byte groups of its rows (ds_read_b128, the
// slot layout of the wide kernel is conflict-free for them too), combines, scales, splits, and holds
//     B = [V_hi(4 tiles) | V_lo(4 tiles)],   A1 = [dM_lo | dM_lo],   A2 = [dM_hi | dM_hi]
// -- the K order is free as long as A and B agree, so no lane computes what another lane also computes.  Two MFMAs per
// (m-tile, n-tile): lo terms first.  30 MFMAs of 16 cycles per tile and wave against 60 of 32 cycles in the fp32 kernel.
// (In registers: B = [V_hi | V_lo] is kept per n-tile, A1 = [dM_lo | dM_lo], A2 = [dM_hi | dM_hi] are made per m-tile.)
//
// Scales: ONE pair per launch (max over the batch of the per-sample running maxima the producers of dY and of the input
// maintain): a weight gradient is a sum over the batch anyway, and a common scale lets all samples share the accumulators.
// Scaled max in [2^12, 2^13): |dM|, |V| <= 4 max < 65 504.
// Rounding bias of the binary16 MFMA (-0.008 ulp per instruction toward -infinity, profiles/NOTES_r05.md section 6): the
// pixel splits of a slab run with alternating signs of dM (the sign rides on the scale; the epilogue's factor undoes it), so
// the bias cancels between the splits that are summed into one gradient.
#pragma once
#include "wgrad_wino.h"

namespace sinddm {

using wgh8 = __attribute__((ext_vector_type(8))) _Float16;
using wgh4 = __attribute__((ext_vector_type(4))) _Float16;

// compile-time timing ablations (-DWGH_ABL=bits; results are WRONG, never ship):
//   1 no DMA traffic (empty descriptors)   2 no LDS operand reads   4 no DMA instructions   8 no MFMAs   16 no split (one conversion per value)
#ifndef WGH_ABL
#define WGH_ABL 0
#endif
constexpr int WGH_TARGET_EXP = 12;

__device__ __forceinline__ int wgh_shift_for(float m) {
    const unsigned bits = __float_as_uint(m) & 0x7fffffffu;
    const int e = (int)(bits >> 23);
    if (e == 0 || e == 255) return 0;
    int s = WGH_TARGET_EXP - (e - 127);
    return s > 100 ? 100 : (s < -100 ? -100 : s);
}
__device__ __forceinline__ float wgh_pow2(int s) { return __uint_as_float((unsigned)(s + 127) << 23); }

// (volatile, as ww_ld2: no compiler-inserted vmcnt wait in front of reads of an LDS-DMA destination; the tile barrier publishes)
__device__ __forceinline__ f32x4 wgh_ld4(const float* q) {
    typedef const volatile __attribute__((address_space(3))) f32x4* lds_v4;
    if (WGH_ABL & 2) return f32x4{1.f, 2.f, 3.f, 4.f};
    return *(lds_v4)q;
}
__device__ __forceinline__ void wgh_split(f32x4 v, float sc, wgh4& hi, wgh4& lo) {
    const f32x4 s = v * sc;
    hi = __builtin_convertvector(s, wgh4);
    if (WGH_ABL & 16) { lo = hi; return; }
    // remainder as one fused multiply-add per value with the binary16 piece as an operand (v_fma_mix_f32): v * sc is exact
    // (power of two), so fma(v, sc, -hi) = s - hi exactly
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = __builtin_fmaf(v[i], sc, -(float)hi[i]);
    lo = __builtin_convertvector(r, wgh4);
}

__global__ __launch_bounds__(WW_THREADS) void wgrad_wh_kernel(WwArgs p) {
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

    // ---- operand scales of the launch ----
    float ma = 0.f, mv = 0.f;
    for (int b = lane; b < p.B; b += 64) {
        ma = fmaxf(ma, p.amax_d[(size_t)b * AMAX_STRIDE]);
        mv = fmaxf(mv, p.amax_i[(size_t)b * AMAX_STRIDE]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ma = fmaxf(ma, __shfl_xor(ma, o));
        mv = fmaxf(mv, __shfl_xor(mv, o));
    }
    const int sa = wgh_shift_for(ma), sv = wgh_shift_for(mv);
    const float sgn = (s & 1) ? -1.0f : 1.0f;
    const float sca = sgn * wgh_pow2(sa), scv = wgh_pow2(sv);
    const float out_a = sgn * wgh_pow2(-sa), out_v = wgh_pow2(-sv);

    // ---- the (at most) three DMA instructions of this wave (wgrad_wino_wide_kernel's image of a tile) ----
    constexpr unsigned OOB = 0x40000000u;
    unsigned dloc[3];
    int dyx[3];
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
        const int idx = min(xi + 16 * sl, WX_NDMA - 1);
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
    int nb = t_begin / tpi;
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
        } else {
            ta.rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dout), 0, 0, 0x00020000);
            ta.ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, 0, 0x00020000);
        }
        return ta;
    };
    // (Measured alternative: the tile through registers -- plain 16-byte buffer loads, ds_write_b128 at the end of the tile,
    // two stages -- instead of LDS-DMA: 39.2 against 38.4 ms per training step; the kernel is bound by its VALU work and
    // the socket's power limit, not by the 47 DMA instructions per tile.)
    auto issue = [&](const TileAddr& ta, float* buf, int sl) {
        const int idx = min(xi + 16 * sl, WX_NDMA - 1);
        const int gy = ta.y0 - 1 + (dyx[sl] & 0xff), gx = ta.x0 - 4 + (dyx[sl] >> 8);
        const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const unsigned voff = (dloc[sl] + (unsigned)(ta.y0 * W + ta.x0) * 4u) | (ok ? 0u : OOB);
        const __amdgpu_buffer_rsrc_t rs = idx < 20 ? ta.rd : ta.ri;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(buf + idx * 256), 16, (int)voff, 0, 0, 0);
    };

    f32x4 acc[5][3];
#pragma unroll
    for (int mt = 0; mt < 5; ++mt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const bool dobias = p.gb != nullptr && cib == 0 && xi == 5;

    float* b0 = smem;                 // three-stage ring: tile T is read while T + 1 has landed and T + 2 is in flight
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
        // dM = A dY A^T on the 2x2 block: rows  fi 0: +r0   1: r0 + r1   2: r0 - r1   3: -r1;  columns (x, y) alike with fj
        constexpr bool need_r0 = fi != 3, need_r1 = fi != 0;
        // V = B^T d B: rows  0: +d0 -d2   1: +d1 +d2   2: -d1 +d2   3: +d1 -d3   (same for the columns)
        constexpr int pa0 = fi == 0 ? 0 : 1, pa1 = fi == 3 ? 3 : 2;
        constexpr float sa0 = fi == 2 ? -1.f : 1.f, sa1 = (fi == 0 || fi == 3) ? -1.f : 1.f;
        constexpr int pb0 = fj == 0 ? 0 : 1, pb1 = fj == 3 ? 3 : 2;
        constexpr float sb0 = fj == 2 ? -1.f : 1.f, sb1 = (fj == 0 || fj == 3) ? -1.f : 1.f;
        const int tr = kq & 1, half = kq >> 1;
        // dY: block (m-tile, row) of 256 floats, slot 16 q + plane: this lane's tiles = column groups q = 2 half, 2 half + 1
        const int abase = (2 * tr) * 256 + 128 * half + 4 * l16;
        // input: group e = 6 hr + k (hr = halo row, k = column group from x0 - 4), slot 16 e + plane; this lane's window =
        // groups k = 2 half .. 2 half + 3 (columns 8 half .. 8 half + 15 from x0 - 4); tile i, patch column pb -> window
        // index 2 i + 3 + pb
        const int vbase = WX_DY + 128 * half + 4 * l16;
        const int vrow0 = 384 * (2 * tr + pa0), vrow1 = 384 * (2 * tr + pa1);
        auto colmix = [&](float x, float y) { return fj == 0 ? x : fj == 1 ? x + y : fj == 2 ? x - y : -y; };
        for (int tile = t_begin; tile < t_end; ++tile) {
            // This tile's three DMA instructions of this wave have landed (vector memory returns in order: at most the three of
            // tile + 1 are still in flight -- they keep two tile times to land); every wave is done with the reads of tile - 1.
            asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const TileAddr ta = next_tile(tile + 2 < t_end && !(WGH_ABL & 1));
            if (!(WGH_ABL & 4)) {
#pragma unroll
                for (int sl = 0; sl < 3; ++sl) issue(ta, b2, sl);
            }
            const float* buf = b0;
            // ---- B fragments of the tile: V of 16 input channels per n-tile, this lane's 4 tiles ----
            wgh8 bv[NNT];             // [V_hi (4 tiles) | V_lo (4 tiles)]
#pragma unroll
            for (int nt = 0; nt < NNT; ++nt) {
                const float* qv = buf + vbase + nt * 2304;
                float w[16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bool need = (g == 1 || g == 2) || (g == 0 && fj == 0) || (g == 3 && fj == 3);
                    if (!need) continue;
                    const f32x4 x0 = wgh_ld4(qv + 64 * g + vrow0), x1 = wgh_ld4(qv + 64 * g + vrow1);
                    const f32x4 u = sa0 * x0 + sa1 * x1;
                    w[4 * g + 0] = u.x; w[4 * g + 1] = u.y; w[4 * g + 2] = u.z; w[4 * g + 3] = u.w;
                }
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = sb0 * w[2 * i + 3 + pb0] + sb1 * w[2 * i + 3 + pb1];
                wgh4 hi, lo;
                wgh_split(v, scv, hi, lo);
                bv[nt] = __builtin_shufflevector(hi, lo, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            // ---- per m-tile: dM of 16 output channels, split, 2 MFMAs per n-tile ----
#pragma unroll
            for (int mt = 0; mt < 5; ++mt) {
                const float* qd = buf + abase + mt * 1024;
                f32x4 t0, t1;
                if constexpr (need_r0 && need_r1) {
                    const f32x4 r00 = wgh_ld4(qd), r01 = wgh_ld4(qd + 64), r10 = wgh_ld4(qd + 256), r11 = wgh_ld4(qd + 320);
                    t0 = fi == 1 ? r00 + r10 : r00 - r10;
                    t1 = fi == 1 ? r01 + r11 : r01 - r11;
                } else if constexpr (need_r0) {
                    t0 = wgh_ld4(qd); t1 = wgh_ld4(qd + 64);
                } else {
                    t0 = -wgh_ld4(qd + 256); t1 = -wgh_ld4(qd + 320);
                }
                const f32x4 dm{colmix(t0.x, t0.y), colmix(t0.z, t0.w), colmix(t1.x, t1.y), colmix(t1.z, t1.w)};
                if constexpr (XI == 5) bsum[mt] += (dm.x + dm.y) + (dm.z + dm.w);      // frequency (1,1): the bias gradient
                wgh4 hi, lo;
                wgh_split(dm, sca, hi, lo);
                const wgh8 ah = __builtin_shufflevector(hi, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                const wgh8 al = __builtin_shufflevector(lo, lo, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int nt = 0; nt < NNT; ++nt) {
                    if (WGH_ABL & 8) { acc[mt][nt].x += (float)ah[0] + (float)al[5] + (float)bv[nt][1]; continue; }
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bv[nt], acc[mt][nt], 0, 0, 0);     // dM_lo x (V_hi + V_lo)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bv[nt], acc[mt][nt], 0, 0, 0);     // dM_hi x (V_hi + V_lo)
                }
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
#define WGH_CASE(n) case n: run_nnt(std::integral_constant<int, n>{}); break;
        WGH_CASE(0) WGH_CASE(1) WGH_CASE(2) WGH_CASE(3) WGH_CASE(4) WGH_CASE(5) WGH_CASE(6) WGH_CASE(7)
        WGH_CASE(8) WGH_CASE(9) WGH_CASE(10) WGH_CASE(11) WGH_CASE(12) WGH_CASE(13) WGH_CASE(14) default: run_nnt(std::integral_constant<int, 15>{});
#undef WGH_CASE
    }
    // remove the scales (and the split's sign); the bias sum was taken from the unscaled dM
#pragma unroll
    for (int mt = 0; mt < 5; ++mt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) acc[mt][nt] = (acc[mt][nt] * out_a) * out_v;
    ww_epilogue(p, smem, acc, bsum, dobias, xi, lane, cb, ci0);
}

}  // namespace sinddm
