// Winograd F(2x4, 3x3) 3x3 convolution on the gfx950 fp32 matrix cores -- fifth generation: weights shared by two
// n-tiles (as conv_wino4.h) AND two waves per SIMD (as conv_wino3.h).
//
// What conv_wino4.h (one wave per SIMD, 240 accumulator registers) taught (tools/ubench/mfma_fillers.hip, tools/w4_seg.py):
//   * a wave's OWN VALU instructions never overlap its fp32 MFMAs: behind an MFMA a VALU waits ~8 cycles for the matrix
//     pipe and then issues at 4 cycles per instruction with the next MFMA held back meanwhile; a DIFFERENT wave's VALU on
//     the same SIMD costs the MFMA stream nothing.  LDS / vector-memory / scalar instructions are free in both cases.
//   * so the one-wave kernel pays for every VALU twice: the input-transform bursts take 6-7 % of its main loop, and its
//     output-transform epilogue -- ~1 200 VALU per item at 4-8 cycles each plus barrier and LDS round trips, nothing
//     else running on the CU -- is 11-26 % of an item.
// This kernel keeps the weight sharing (every A fragment of the packed image is loaded ONCE per CU and k-step and feeds
// two MFMAs) but splits a frequency row's 60 accumulator tiles over TWO waves of the same SIMD:
//   * one persistent 8-wave workgroup per CU (512 threads, 2 waves per SIMD, 256 registers each).  Wave (i, MH): vertical
//     frequency row i, half MH of the 80 output channels x 6 horizontal frequencies of BOTH n-tiles of an 8x32-pixel
//     item: half 0 = m-tiles 0, 2 and frequencies 0..2 of m-tile 4, half 1 = m-tiles 1, 3 and frequencies 3..5 of
//     m-tile 4 -- 15 (m-tile, frequency) pairs x 2 n-tiles = 30 accumulator tiles = 120 AGPRs (a0..a119 by number),
//     30 MFMAs per k-step and wave.  While one wave runs its transform burst, waits at the chunk barrier or is in its
//     epilogue VALU, the other one's MFMAs own the matrix pipe.
//   * the packed F(2x4) weight image orders a (row, k-step)'s 30 fragments by (half, pair) -- 16 slots per half, slot 15
//     padding (w3_pos_e in conv_wino3.h) -- so a wave's fragments are four consecutive 16-byte groups: 4 KB per wave and
//     k-step, 8 KB per SIMD: the same L2 traffic per MFMA as conv_wino4 (128 B), half of conv_wino3.
//   * both waves of a pair need all six B operands of both n-tiles: each reads the raw patches and runs the packed
//     18-instruction transform itself (the duplicate runs beside the partner's MFMAs, i.e. for free).
//   * epilogue: three passes of 32 channels through the 64 KB exchange area as before; every wave transforms its own
//     tiles (half as many as in conv_wino4), the two partial column transforms of m-tile 4 are summed by the reader;
//     512 reader threads (two channels each per pass).  All eight waves are in it at the same time, so VALU issues from
//     two waves per SIMD (2 cycles per instruction instead of 4) and one wave's LDS / memory waits hide the other's.
// Everything else is conv_wino4.h: asm MFMAs on numbered AGPRs, ring of two k-steps of A registers, conflict-free raw
// tile (row stride 41 / plane stride 411), XOR buffer swap, packed transform burst, pass-ahead epilogue operands with
// the channel in the scalar offset, 16-byte stores + wait state in one asm.  The two halves are two instantiations of
// the kernel body (template MH) behind a wave-uniform branch at the top: (m-tile, frequency) of an accumulator is a
// compile-time fact in both.
#pragma once
#include "conv_wino4.h"

namespace sinddm {

#ifndef W5_ABL          // timing ablations as W4_ABL (results WRONG): 1 no staging, 2 weights once, 4 no LDS reads, 8 no epilogue, 16 no transform
#define W5_ABL 0
#endif
#ifndef W5_XF_SLOT0      // slot of a k-step behind whose MFMA the transform burst sits, per half: the two waves of a SIMD run
#define W5_XF_SLOT0 24   // in lock step (they alternate on the matrix pipe), and bursts at the same slot would leave the pipe
#define W5_XF_SLOT1 29   // idle -- five slots apart one wave's burst lies beside the other's MFMAs
#endif
#ifdef W5_TIMING
__device__ unsigned long long g_w5_seg[8 * 256 * 8 * 32];
#define W5_SEG(slot) do { if (seg) g_w5_seg[((p.mtp * 256 + blockIdx.x) * 8 + wv) * 32 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W5_SEG(slot) do {} while (0)
#endif
constexpr int W5_NE = 15;                      // (m-tile, frequency) pairs per wave
// Accumulator tile t of a wave lives in AGPRs a[8 + 4t : 11 + 4t], t = 0..29 (a8 .. a127), addressed by number from inline
// asm (conv_wino4.h).  At two waves per SIMD the compiler budgets 128 VGPRs + 128 AGPRs for itself and, when it runs out of
// VGPRs, parks values in the LOWEST AGPRs (ascending allocation order), blind to the accumulators.  So the kernel has to
// fit 128 VGPRs almost without that (one operand set in the epilogue, no operand prefetch inside the main loop), a0..a7 are
// left to the compiler, and tests/test_build_isa.py checks on its output that it never touches an AGPR above a7.
constexpr int W5_ACC0 = 8;
template <int T>
__device__ __forceinline__ void w5_mfma(float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "n"(W5_ACC0 + 4 * T), "n"(W5_ACC0 + 4 * T + 3));
}
constexpr int W5_KS_BYTES = W3_KS_BYTES;       // image of one (row, k-step): 2 halves x 4 groups x 1 KB
constexpr int W5_HALF_BYTES = W5_KS_BYTES / 2;

template <int ACT, int EDGE, int MH>
__device__ __forceinline__ void conv_wino5_body(const ConvArgs& p, int items_per_xcd, int wg_per_xcd, float* smem) {
    constexpr int MT = W3_MT;
    float* sX = smem + 2 * W4_BUF;
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    using f32x3 = __attribute__((ext_vector_type(3))) float;
    using u32x3 = __attribute__((ext_vector_type(3))) unsigned;
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

    const int xcd = blockIdx.x & 7;
    const int ls = blockIdx.x >> 3;
    const int tpi = p.tilesX * p.tilesY;
    auto decode = [&](int k, Wino4Item& it) -> bool {
        const int li = ls + k * wg_per_xcd;
        if (li >= items_per_xcd) return false;
        const int tl = li / p.coblks;
        const int tile = xcd * p.tiles_per_xcd + tl;
        if (tile >= p.ntiles) return false;
        it.cb = li - tl * p.coblks;
        it.b = tile / tpi;
        const int trm = tile - it.b * tpi;
        const int ty = trm / p.tilesX;
        it.y0 = ty * W4_TH;
        it.x0 = (trm - ty * p.tilesX) * W4_TW;
        return true;
    };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave 0..7 = (half MH, row i)
    const int wi = wv & 3;
    const int l16 = lane & 15, kq = lane >> 4;
    const int H = p.H, W = p.W;
    const int HW = H * W;

    // vertical B^T rows (F(2,3)):  0: d0 - d2   1: d1 + d2   2: d1 - d2 (U_2j stored negated)   3: d1 - d3
    const int pa0 = wi == 0 ? 0 : 1, pa1 = wi == 3 ? 3 : 2;
    const float sgn = wi == 1 ? 1.f : -1.f;
    const int tr_ = l16 >> 3, tc_ = l16 & 7;
    typedef __attribute__((address_space(3))) float lds_f;
    const unsigned lds0 = (unsigned)(size_t)(lds_f*)smem;
    unsigned rd[2][2];                              // LDS byte addresses [row a / row b][n-tile] in the buffer being read
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        rd[0][h] = lds0 + 4u * (kq * W4_PS + (4 * h + 2 * tr_ + pa0) * W4_RS + 4 * tc_ + 3);
        rd[1][h] = lds0 + 4u * (kq * W4_PS + (4 * h + 2 * tr_ + pa1) * W4_RS + 4 * tc_ + 3);
    }
    auto lds_ld = [](unsigned addr, int foff) __attribute__((always_inline)) { return ((const lds_f*)addr)[foff]; };
    const int nch = p.nch3;

    // ---- raw tile staging: wave wv brings channels 2 wv, 2 wv + 1 of a chunk; 100 sixteen-byte groups per plane = two loads ----
    constexpr unsigned OOB = 0x40000000u;
    auto stage_li = [&](int s) { return s == 0 ? lane : 64 + (lane < 36 ? lane : lane - 36); };
    unsigned goff[2], goffn[2];                     // global byte offsets of the lane's two groups: this item / the next one
    int swo[2];
    int nv = 6, nvn = 6;                            // EDGE builds: patch column c is inside the image iff c < nv
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int li = stage_li(s);
        const int row = li / 10, grp = li - row * 10;
        swo[s] = row * W4_RS + grp * 4;
    }
    auto make_goff = [&](const Wino4Item& it) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int li = stage_li(s);
            const int row = li / 10, grp = li - row * 10;
            const int gy = it.y0 + row - 1, gx = it.x0 - 4 + 4 * grp;
            const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
            goffn[s] = ok ? (unsigned)(gy * W + gx) * 4u : OOB;
        }
        nvn = W - (it.x0 + 4 * tc_ - 1);
    };
    // The four 16-byte groups a wave stages per chunk are requested a WHOLE chunk (~7 700 cycles) before they are written
    // to LDS: each register set is reloaded for chunk c + 2 right behind its LDS write of chunk c + 1.  (One k-step of
    // distance, as in conv_wino4.h, is less than an HBM round trip under load: the wave stalled at the LDS write and,
    // vector memory returning in order, so did every weight refill queued behind; measured 5 points of the main loop.)
    f32x4 stg[4];
    const unsigned HW4 = (unsigned)HW * 4u;
    auto plane_ptr = [&](int ib) { return p.in + ((size_t)ib * p.Cin + wv * 2) * HW; };
    __amdgpu_buffer_rsrc_t rs_st;
    unsigned gsel[2];                               // goff or goffn: offsets of the item the chunk being requested belongs to
    auto stage_load = [&](int n) __attribute__((always_inline)) {                  // n = 2 g + s: group s of channel 2 wv + g
        if (W5_ABL & 1) return;
        stg[n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_st, (int)gsel[n & 1], (n >> 1) * (int)HW4, 0));
    };
    unsigned swb[2];                                // byte address of the lane's group in plane 2 wv of the buffer being WRITTEN
#pragma unroll
    for (int s = 0; s < 2; ++s) swb[s] = lds0 + 4u * (W4_BUF + wv * 2 * W4_PS + swo[s]);
    auto stage_store = [&](auto NE) __attribute__((always_inline)) {                // one write per MFMA slot: NE = 4 n + e
        constexpr int n = decltype(NE)::value >> 2, e = decltype(NE)::value & 3;
        if (W5_ABL & 1) return;
        constexpr int off = (n >> 1) * W4_PS * 4;
        const unsigned addr = swb[n & 1];
        const float val = stg[n][e];
        asm volatile("ds_write_b32 %0, %1 offset:%c2" ::"v"(addr), "v"(val), "n"(off + 4 * e) : "memory");
    };

    // ---- weights: this wave's 15 fragments of a (row, k-step) = groups 0..2 (16 bytes per lane) + group 3 (12 bytes) ----
    const __amdgpu_buffer_rsrc_t rsw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w3), 0, 0x7FFFFFF0, 0x00020000);
    const int wlane = lane * 16;
    auto wbase = [&](int cb) -> int { return cb * nch * W3_CH_BYTES + wi * (4 * W5_KS_BYTES) + MH * W5_HALF_BYTES; };
    f32x4 aq[2][3];                                 // ring of two k-steps
    f32x3 aq3[2];                                   // (group 3 as 12 bytes: no dead fourth register for the allocator to reuse under a pending load)
    const __amdgpu_buffer_rsrc_t rs_none = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.zero), 0, 0u, 0x00020000);
    auto load_a = [&](const __amdgpu_buffer_rsrc_t& rw, int ring, int g, int soff) __attribute__((always_inline)) {
        if (g < 3)
            aq[ring][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, wlane + g * 1024, soff, 0));
        else
            aq3[ring] = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(rw, wlane + g * 1024, soff, 0));
    };
    auto chunk_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // ---- input transform of both n-tiles of a k-step: one burst of 18 packed VALU (conv_wino4.h) ----
    const f32x2 sgn2{sgn, sgn};
    auto xf_burst = [&](f32x2 (&ra)[6], f32x2 (&rb)[6], f32x2 (&v)[W3_NF], int nvalid) __attribute__((always_inline)) {
        asm volatile("" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]));
        asm volatile("" : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(rb[4]), "+v"(rb[5]));
        if (W5_ABL & 16) {
#pragma unroll
            for (int c = 0; c < 6; ++c) v[c] = ra[c];
        } else {
            f32x2 r[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                r[c] = sgn2 * rb[c] + ra[c];
                if (EDGE && c >= 2) {
                    const bool ok = c < nvalid;
                    r[c].x = ok ? r[c].x : 0.f;
                    r[c].y = ok ? r[c].y : 0.f;
                }
            }
            // F(4,3) B^T:  [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
            const f32x2 s24 = r[4] - 4.f * r[2], s13 = r[3] - 4.f * r[1];
            const f32x2 u24 = r[4] - r[2], d31 = r[3] - r[1];
            v[0] = 4.f * r[0] + (r[4] - 5.f * r[2]);
            v[1] = s24 + s13;
            v[2] = s24 - s13;
            v[3] = u24 + 2.f * d31;
            v[4] = u24 - 2.f * d31;
            v[5] = 4.f * r[1] + (r[5] - 5.f * r[3]);
        }
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]));
    };

    Wino4Item it;
    int l = 0;
    if (!decode(l, it)) return;
    asm volatile("" ::: "a8", "a127");
    w4_static_for<2 * W5_NE * 4>([&](auto R) __attribute__((always_inline)) { w4_acc_zero<W5_ACC0 + decltype(R)::value>(); });
    make_goff(it);
    nv = nvn;
    goff[0] = goffn[0];
    goff[1] = goffn[1];
    gsel[0] = goff[0];
    gsel[1] = goff[1];
    int wb_it = wbase(it.cb);
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
        for (int g = 0; g < 4; ++g) load_a(rsw, r2, g, wb_it + r2 * W5_KS_BYTES);
    // first item: chunk 0 straight into LDS, chunk 1 into the staging registers
    rs_st = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(plane_ptr(it.b)), 0, 2 * (int)HW4, 0x00020000);
#pragma unroll
    for (int n = 0; n < 4; ++n) stage_load(n);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        float* d = smem + (wv * 2 + (n >> 1)) * W4_PS + swo[n & 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = stg[n][e];
    }
    rs_st = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(plane_ptr(it.b) + (size_t)16 * HW), 0, 2 * (int)HW4, 0x00020000);
#pragma unroll
    for (int n = 0; n < 4; ++n) stage_load(n);
    __syncthreads();
    int wcur = wb_it;
    f32x2 v[2][W3_NF];                              // [k-step parity][frequency] (n-tile 0, n-tile 1)
    f32x2 raw[2][6];                                // [row a / row b][column]      (n-tile 0, n-tile 1)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        raw[0][c] = f32x2{lds_ld(rd[0][0], c), lds_ld(rd[0][1], c)};
        raw[1][c] = f32x2{lds_ld(rd[1][0], c), lds_ld(rd[1][1], c)};
    }
    xf_burst(raw[0], raw[1], v[0], nv);

    // ---- epilogue reader role: thread = 2x4 tile (n-tile, tile row, tile column) x channels cg + 16 k of a pass ----
    const int tile = tid & 31;
    const int hr = tile >> 4, trr = (tile >> 3) & 1, tcr = tile & 7;
    const int cg = tid >> 5;                        // 0..15
    constexpr int NP = EDGE == 0 ? 1 : (EDGE == 1 ? 2 : 4);
    auto ep_geo = [&](const Wino4Item& g, unsigned (&vo)[2][NP]) __attribute__((always_inline)) {
        const int y = g.y0 + 4 * hr + 2 * trr, x = g.x0 + 4 * tcr;
        const unsigned base = ((unsigned)cg * (unsigned)HW + (unsigned)(y * W + x)) * 4u;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) {
                const int px = pc * (4 / NP);
                const bool ok = (y + pp < H) & (x + px < W);
                vo[pp][pc] = ok ? base + (unsigned)(pp * W + px) * 4u : OOB;
            }
    };
    unsigned vo[2][NP];
    ep_geo(it, vo);
    auto ep_load = [&](const __amdgpu_buffer_rsrc_t& r, int pp, int soff) __attribute__((always_inline)) -> f32x4 {
        if constexpr (EDGE == 0) {
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)vo[pp][0], soff, 0));
        } else if constexpr (EDGE == 1) {
            const u32x2 a0 = __builtin_amdgcn_raw_buffer_load_b64(r, (int)vo[pp][0], soff, 0);
            const u32x2 a1 = __builtin_amdgcn_raw_buffer_load_b64(r, (int)vo[pp][1], soff, 0);
            return __builtin_bit_cast(f32x4, u32x4{a0[0], a0[1], a1[0], a1[1]});
        } else {
            f32x4 t;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                t[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)vo[pp][e], soff, 0));
            return t;
        }
    };
    auto ep_store = [&](const __amdgpu_buffer_rsrc_t& r, int pp, int soff, f32x4 vv) __attribute__((always_inline)) {
        const u32x4 u = __builtin_bit_cast(u32x4, vv);
        if constexpr (EDGE == 0) {
            // (store + one wait state as ONE asm: see conv_wino4.h -- the 16-byte store-data hazard with a scalar offset)
            const unsigned voff = vo[pp][0];
            asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(u), "v"(voff), "s"(r), "s"(soff) : "memory");
        } else if constexpr (EDGE == 1) {
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{u[0], u[1]}, r, (int)vo[pp][0], soff, 0);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{u[2], u[3]}, r, (int)vo[pp][1], soff, 0);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) __builtin_amdgcn_raw_buffer_store_b32(u[e], r, (int)vo[pp][e], soff, 0);
        }
    };
    f32x4 opv[2][2];                                // [k][output row]: operands of the pass in flight
    float bsv[2];

    for (;;) {
#ifdef W5_TIMING
        const bool seg = l == 3 && blockIdx.x < 256;
#endif
        W5_SEG(0);
        Wino4Item nx;
        l += 1;
        const bool have_next = decode(l, nx);
        if (!have_next) nx = it;
        const int wb_nx = wbase(nx.cb);
        const float* base_nx = plane_ptr(nx.b);
        const unsigned plane_b = HW4;
        const unsigned samp_b = (unsigned)p.Cout * plane_b;
        const size_t samp_o = (size_t)it.b * p.Cout * HW;
        auto rsrc_of = [&](const float* base) {
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base ? base + samp_o : p.zero), 0,
                                                     base ? samp_b : 0u, 0x00020000);
        };
        const __amdgpu_buffer_rsrc_t rs_out = rsrc_of(p.out);
        const __amdgpu_buffer_rsrc_t rs_op = rsrc_of(ACT == 2 ? p.aux : p.resid);      // the one per-pixel operand
        const __amdgpu_buffer_rsrc_t rs_pre = rsrc_of(ACT == 1 ? p.out_pre : nullptr);
        const __amdgpu_buffer_rsrc_t rs_bias = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.bias ? p.bias : p.zero), 0, p.bias ? (unsigned)(p.coblks * MT * 16) * 4u : 0u, 0x00020000);
        const int cb_ch = it.cb * (MT * 16);
        // pass P covers channels 32 P .. 32 P + 31 of the block (pass 2: 16 channels); a thread handles cg + 16 k
        auto ep_nk = [](int pass) { return pass < 2 ? 2 : 1; };
        auto ep_soff = [&](int pass, int k) -> int { return (cb_ch + pass * 32 + 16 * k) * (int)plane_b; };
        auto ep_fetch = [&](int pass) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (k >= ep_nk(pass)) continue;
                if (ACT != 2)
                    bsv[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        rs_bias, cg * 4, (cb_ch + pass * 32 + 16 * k) * 4, 0));
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) opv[k][pp] = ep_load(rs_op, pp, ep_soff(pass, k));
            }
        };

        for (int c = 0; c < nch; ++c) {
            const bool last = c + 1 == nch;
            if (c + 2 == nch) make_goff(nx);                 // (the first chunk whose requests belong to the next item)
            // weights of k-step + 2 (the ring): k-steps 2, 3 of this chunk, then 0, 1 of the next chunk / item
            const int wnext = last ? wb_nx : wcur + W3_CH_BYTES;
            const int w_pre[4] = {wcur + 2 * W5_KS_BYTES, wcur + 3 * W5_KS_BYTES, wnext, wnext + W5_KS_BYTES};
            // raw tile requested in this chunk: chunk c + 2 of the item, or chunk c + 2 - nch of the next one
            const bool own = c + 2 < nch;
            const int rch = own ? c + 2 : c + 2 - nch;
            const float* rbase = (own ? plane_ptr(it.b) : base_nx) + (size_t)rch * 16 * HW;
            const bool live = (own || have_next) && rch * 16 + wv * 2 < p.Cin;
            rs_st = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rbase), 0, live && !(W5_ABL & 32) ? 2 * (int)HW4 : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsw_tail = last ? rs_none : rsw;
            gsel[0] = own ? goff[0] : goffn[0];
            gsel[1] = own ? goff[1] : goffn[1];
            w4_static_for<4>([&](auto KS) __attribute__((always_inline)) {
                constexpr int ks = decltype(KS)::value;
                if constexpr (ks == 3) {
                    __builtin_amdgcn_sched_barrier(0);
                    chunk_barrier();
#pragma unroll
                    for (int i = 0; i < 4; ++i) rd[i >> 1][i & 1] ^= 4u * W4_BUF;
                    swb[0] ^= 4u * W4_BUF;
                    swb[1] ^= 4u * W4_BUF;
                    asm volatile("" : "+v"(rd[0][0]), "+v"(rd[0][1]), "+v"(rd[1][0]), "+v"(rd[1][1]), "+v"(swb[0]), "+v"(swb[1]));
                }
                constexpr int rd_off = ks < 3 ? (ks + 1) * 4 * W4_PS : 0;
                const int mk = (ks == 3 && last) ? nvn : nv;
                // 30 slots: one MFMA + the non-VALU fillers dealt behind it
                w4_static_for<2 * W5_NE>([&](auto S) __attribute__((always_inline)) {
                    constexpr int s = decltype(S)::value;
                    constexpr int el = s >> 1, h = s & 1;
                    constexpr int j = el < 12 ? el % 6 : 3 * MH + el - 12;
                    if constexpr ((el >> 2) < 3) w5_mfma<s>(aq[ks & 1][el >> 2][el & 3], v[ks & 1][j][h]);
                    else w5_mfma<s>(aq3[ks & 1][el & 3], v[ks & 1][j][h]);
                    // raw-patch reads of the next k-step: slots 0..23
                    if constexpr (s < 24 && !(W5_ABL & 4)) {
                        constexpr int hh = s & 1, m = s >> 1, cc = m >> 1, wh = m & 1;
                        raw[wh][cc][hh] = lds_ld(rd[wh][hh], rd_off + cc);
                    }
                    if constexpr (s == (MH ? W5_XF_SLOT1 : W5_XF_SLOT0)) xf_burst(raw[0], raw[1], v[(ks + 1) & 1], mk);
                    // raw-tile staging: the groups of chunk c + 1 go to LDS, their registers are reloaded for chunk c + 2
                    if constexpr (ks == 1 && s >= 12 && s < 20) stage_store(std::integral_constant<int, s - 12>{});
                    if constexpr (ks == 1 && s >= 20 && s < 22) stage_load(s - 20);
                    if constexpr (ks == 2 && s >= 12 && s < 20) stage_store(std::integral_constant<int, s - 12 + 8>{});
                    if constexpr (ks == 2 && s >= 20 && s < 22) stage_load(s - 20 + 2);
                    // weight refills: group g is free behind slot 8 g + 7 (the last one behind slot 29)
                    // (the last chunk's k-steps 2, 3 would request the NEXT item's first two k-steps: 30 registers alive through
                    // the whole epilogue, more than the 128-VGPR budget leaves -- they go through an empty descriptor there
                    // and the real requests are issued inside the epilogue's last pass)
                    if constexpr (!(W5_ABL & 2) && ((s & 7) == 7 || s == 29)) load_a(ks < 2 || (W5_ABL & 8) ? rsw : rsw_tail, ks & 1, s >> 3, w_pre[ks]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            wcur = wnext;
        }

        W5_SEG(1);
        // ---- output transform + epilogue: column half (6 -> 4) in registers, row half through LDS, 32 channels per pass ----
        if (!(W5_ABL & 8)) {
        asm volatile("s_nop 15\n\ts_nop 15");
        w4_static_for<3>([&](auto PASS) __attribute__((always_inline)) {
            constexpr int pass = decltype(PASS)::value;
            // residual / bias / GELU' operands of this pass: requested now, used behind the column transform, the exchange
            // barrier and the LDS reads (one register set: two sets in flight cost the 16 registers that made the
            // compiler park values in AGPRs)
            ep_fetch(pass);
            // this wave's tile of the pass: a whole m-tile (passes 0, 1) or its three frequencies of m-tile 4 (pass 2)
            w4_static_for<8>([&](auto I) __attribute__((always_inline)) {
                constexpr int h = decltype(I)::value >> 2, r = decltype(I)::value & 3;
                f32x4 t;
                if constexpr (pass < 2) {
                    // (M A4)[i][q]:  A4^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
                    constexpr int R0 = W5_ACC0 + ((pass * 6) * 2 + h) * 4 + r;          // register of (pair pass*6 + j, n-tile h, row r) = R0 + 8 j
                    const float m_0 = w4_acc_read<R0>(), m_1 = w4_acc_read<R0 + 8>(), m_2 = w4_acc_read<R0 + 16>();
                    const float m_3 = w4_acc_read<R0 + 24>(), m_4 = w4_acc_read<R0 + 32>(), m_5 = w4_acc_read<R0 + 40>();
                    w4_acc_zero<R0>(); w4_acc_zero<R0 + 8>(); w4_acc_zero<R0 + 16>();
                    w4_acc_zero<R0 + 24>(); w4_acc_zero<R0 + 32>(); w4_acc_zero<R0 + 40>();
                    const float s12 = m_1 + m_2, d12 = m_1 - m_2, s34 = m_3 + m_4, d34 = m_3 - m_4;
                    t = f32x4{m_0 + s12 + s34, fmaf(2.f, d34, d12), fmaf(4.f, s34, s12), fmaf(8.f, d34, d12) + m_5};
                } else {
                    constexpr int R0 = W5_ACC0 + (12 * 2 + h) * 4 + r;
                    const float a_ = w4_acc_read<R0>(), b_ = w4_acc_read<R0 + 8>(), c_ = w4_acc_read<R0 + 16>();
                    w4_acc_zero<R0>(); w4_acc_zero<R0 + 8>(); w4_acc_zero<R0 + 16>();
                    if constexpr (MH == 0) {          // frequencies 0, 1, 2
                        const float s12 = b_ + c_, d12 = b_ - c_;
                        t = f32x4{a_ + s12, d12, s12, d12};
                    } else {                          // frequencies 3, 4, 5
                        const float s34 = a_ + b_, d34 = a_ - b_;
                        t = f32x4{s34, 2.f * d34, 4.f * s34, fmaf(8.f, d34, c_)};
                    }
                }
                *reinterpret_cast<f32x4*>(sX + ((wi * 32 + MH * 16 + kq * 4 + r) * 32 + h * 16 + l16) * 4) = t;
            });
            W5_SEG(2 + 6 * pass);
            if constexpr (pass == 2) {             // weights of the next item's k-steps 0 and 1 (see the main loop)
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                    for (int g = 0; g < 4; ++g) load_a(rsw, r2, g, wb_nx + r2 * W5_KS_BYTES);
            }
            lds_barrier();
            W5_SEG(3 + 6 * pass);
            f32x4 yv[2][2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (k >= ep_nk(pass)) continue;
                const int cl = cg + 16 * k;
                f32x4 t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    t[i] = *reinterpret_cast<const f32x4*>(sX + ((i * 32 + cl) * 32 + tile) * 4);
                    if (pass == 2) t[i] += *reinterpret_cast<const f32x4*>(sX + ((i * 32 + 16 + cl) * 32 + tile) * 4);
                }
                yv[k][0] = t[0] + t[1] + t[2];                     // Y[pp] = sum_i A2^T[pp][i] t[i]
                yv[k][1] = t[1] - t[2] - t[3];
            }
            lds_barrier();
            W5_SEG(4 + 6 * pass);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (k >= ep_nk(pass)) continue;
                const int so = ep_soff(pass, k);
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    f32x4 w_ = yv[k][pp];
                    if (ACT != 2) w_ += bsv[k];
                    if (ACT == 1) {
                        ep_store(rs_pre, pp, so, w_);              // pre-activation (the training forward saves it)
#pragma unroll
                        for (int e = 0; e < 4; ++e) w_[e] = gelu_erf(w_[e]);
                    }
                    if (ACT == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) w_[e] *= gelu_erf_grad(opv[k][pp][e]);
                    } else {
                        w_ += opv[k][pp];
                    }
                    ep_store(rs_out, pp, so, w_);
                }
            }
        });
        ep_geo(nx, vo);                            // reader geometry of the next item
        W5_SEG(20);
        }
        if (!have_next) break;
        it = nx;
        goff[0] = goffn[0];
        goff[1] = goffn[1];
        wb_it = wb_nx;
        nv = nvn;
    }
}

template <int ACT, int EDGE>
__global__ __launch_bounds__(512, 2) void conv_wino5_kernel(ConvArgs p, int items_per_xcd, int wg_per_xcd) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (threadIdx.x >> 8) conv_wino5_body<ACT, EDGE, 1>(p, items_per_xcd, wg_per_xcd, smem);
    else conv_wino5_body<ACT, EDGE, 0>(p, items_per_xcd, wg_per_xcd, smem);
}

#ifndef SINDDM_V5_MIN_ITEMS_PER_CU
#define SINDDM_V5_MIN_ITEMS_PER_CU 4
#endif

inline bool conv_wino5_applies(int B, int H, int W, int coblks) {
    return (long long)B * ((W + W4_TW - 1) / W4_TW) * ((H + W4_TH - 1) / W4_TH) * coblks >=
           (long long)SINDDM_V5_MIN_ITEMS_PER_CU * wino2_cu_count();
}

inline int conv_wino5_launch(const ConvArgs& a_in, hipStream_t st) {
    ConvArgs a = a_in;
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
    a.tilesX = (a.W + W4_TW - 1) / W4_TW;
    a.tilesY = (a.H + W4_TH - 1) / W4_TH;
    a.ntiles = a.B * a.tilesX * a.tilesY;
    a.tiles_per_xcd = (a.ntiles + 7) / 8;
    a.mtp = W3_MT;
#ifdef W5_TIMING
    static int w5_launch_no = 0;
    a.mtp = w5_launch_no++ % 8;
#endif
    const int ipx = a.tiles_per_xcd * a.coblks;
    int wpx = wino2_cu_count() / 8;              // one 8-wave workgroup per CU
    if (wpx < 1) wpx = 1;
    if (wpx > ipx) wpx = ipx;
    const unsigned grid = (unsigned)(wpx * 8);
    constexpr size_t lds = W4_LDS_FLOATS * sizeof(float);
#define W5_GO(ACT, EDGE)                                                                                                \
    do {                                                                                                                \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino5_kernel<ACT, EDGE>),                         \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                \
        hipLaunchKernelGGL((conv_wino5_kernel<ACT, EDGE>), dim3(grid), dim3(512), lds, st, a, ipx, wpx);                \
    } while (0)
    const int edge = a.W % 4 == 0 ? 0 : (a.W % 2 == 0 ? 1 : 2);
    switch ((a.act & 0xff) * 3 + edge) {
        case 0: W5_GO(0, 0); break;
        case 1: W5_GO(0, 1); break;
        case 2: W5_GO(0, 2); break;
        case 3: W5_GO(1, 0); break;
        case 4: W5_GO(1, 1); break;
        case 5: W5_GO(1, 2); break;
        case 6: W5_GO(2, 0); break;
        case 7: W5_GO(2, 1); break;
        default: W5_GO(2, 2);
    }
#undef W5_GO
    if (rec) {
        (void)hipEventRecord(prof.ev[2 * prof.used + 1], st);
        const double fl = 2.0 * a.B * a.H * a.W * (double)a.Cout * 9.0 * a.Cin;   // algorithmic (direct-conv) FLOPs
        prof.note(1, fl, fl * (24.0 / 72.0));                                      // F(2x4): 24 multiplies per 8 outputs
    }
    SINDDM_LAUNCH_CHECK();
    return 0;
}

}  // namespace sinddm
