// Winograd F(2x4, 3x3) 3x3 convolution on the gfx950 fp32 matrix cores -- third generation (forward and data gradient
// of the big launches).
//
// The vertical direction keeps F(2,3) (2 output rows from 4 patch rows), the horizontal one uses F(4,3) (4 output
// columns from 6 patch columns): 4 x 6 = 24 frequencies per 2x4 output tile = 3 multiplies per output instead of the
// 4 of F(2x2,3x3) -- 25 % fewer MFMAs for the same pixels (tests/experiments/wino_f43_numerics.py: the extra fp32
// rounding of the F(4,3) transforms stays far inside the parity budget).
//
// Same skeleton as conv_wino2.h (two persistent 4-wave workgroups per CU, wave i = vertical frequency row i, raw halo
// tile 16 channels x 6 x 40 through registers into LDS, one LDS-only barrier per 16-channel chunk, burst schedule,
// incremental scalar addressing), with these differences:
//   * a work item is still a 4x32-pixel tile x 80 output channels, but the tile is ONE n-tile of sixteen 2x4 output
//     tiles (2 tile rows x 8 tile columns); wave i owns the SIX frequencies (i, 0..5): 5 x 6 accumulator tiles = 120
//     registers, 30 MFMAs per k-step (conv_wino2: 160 registers, 40 MFMAs);
//   * per k-step a lane reads 2 x 6 floats of raw patch (two rows of its channel), combines the rows (F(2,3), shared by
//     the six frequencies) and applies the 6-point F(4,3) input transform: 6 + 20 VALU for six B operands;
//   * weights: U = G2 g G4^T in the register image [co-blk][chunk][i][k-step][q = 0..7][lane][4]: the 30 A operands of
//     a (wave, k-step) are eight 16-byte loads (the last one half empty); group q is refilled right after the MFMAs of
//     (m-tile, frequency) pairs 4q .. 4q+3;
//   * output transform: the column half (A4^T, 6 -> 4 values) in registers, the row half (A2^T over the four waves)
//     through LDS, 32 channels per pass; a reader thread owns one 2x4 tile of two channels: 16-byte stores.
// Restrictions (the caller falls back to conv_wino2 otherwise): C_out % 80 == 0, C_in % 16 == 0,
// launches with at least one item per workgroup slot.  Same epilogue contract as conv_wino2.h.
#pragma once
#include "conv_wino2.h"

namespace sinddm {

constexpr int W3_MT = 5;
constexpr int W3_NF = 6;                       // horizontal frequencies
constexpr int W3_Q = 8;                        // 16-byte A groups per (wave, k-step): ceil(5 * 6 / 4)
constexpr int W3_KS_BYTES = W3_Q * 1024;       // weights of one (i, k-step)
constexpr int W3_CH_BYTES = 4 * 4 * W3_KS_BYTES;   // ... of one 16-channel chunk (4 waves x 4 k-steps)
// Order of a (row i, k-step)'s 30 A fragments in the packed image (round 3): 32 four-byte slots per lane = two halves of
// 16; half MH holds m-tiles MH and 2 + MH (six frequencies each) and frequencies 3 MH .. 3 MH + 2 of m-tile 4, its slot 15
// is padding (the order a two-waves-per-SIMD kernel of round 3 needed -- a half per wave; that kernel is gone, the image
// stayed).  conv_wino3 / conv_wino4 walk the slots in this order.  Returns e = m-tile * 6 + frequency, or
// -1 for padding.
constexpr int w3_pos_e(int pos) {
    const int mh = pos >> 4, el = pos & 15;
    if (el == 15) return -1;
    const int blk = el / 6, jj = el - blk * 6;
    return blk < 2 ? (2 * blk + mh) * 6 + jj : 4 * 6 + 3 * mh + jj;
}
// floats of the packed F(2x4) image of one conv
inline long long wino3_packed_floats(int coblks, int nch) { return (long long)coblks * nch * (W3_CH_BYTES / 4); }

// EDGE: 0 when W % 4 == 0; 1 when W % 4 == 2 (patch columns past the row end are masked, 8-byte edge stores); 2 when W
// is odd (4-byte edge stores)
template <int ACT, int EDGE>
__global__ __launch_bounds__(W2_THREADS, 2) void conv_wino3_kernel(ConvArgs p, int items_per_xcd, int wg_per_xcd) {
    constexpr int MT = W3_MT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sX = smem + 2 * W2_BUF;

    const int xcd = blockIdx.x & 7;
    const int ls = blockIdx.x >> 3;
    const int tpi = p.tilesX * p.tilesY;
    const bool by_tile = p.tiles_per_xcd >= wg_per_xcd;
    auto decode = [&](int k, Wino2Item& it) -> bool {          // k-th work item of this workgroup (as conv_wino2)
        int tl, cb;
        if (by_tile) {
            tl = ls + (k / p.coblks) * wg_per_xcd;
            cb = k % p.coblks;
        } else {
            const int li = ls + k * wg_per_xcd;
            if (li >= items_per_xcd) return false;
            tl = li / p.coblks;
            cb = li % p.coblks;
        }
        if (tl >= p.tiles_per_xcd) return false;
        const int tile = xcd * p.tiles_per_xcd + tl;
        if (tile >= p.ntiles) return false;
        it.cb = cb;
        it.b = tile / tpi;
        const int trm = tile - it.b * tpi;
        const int ty = trm / p.tilesX;
        it.y0 = ty * W2_TH;
        it.x0 = (trm - ty * p.tilesX) * W2_TW;
        return true;
    };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wi = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = vertical frequency row i
    const int l16 = lane & 15, kq = lane >> 4;
    const int H = p.H, W = p.W;
    const int HW = H * W;

    // vertical B^T rows (F(2,3)):  0: d0 - d2   1: d1 + d2   2: d1 - d2 (U_2j stored negated)   3: d1 - d3
    const int pa0 = wi == 0 ? 0 : 1, pa1 = wi == 3 ? 3 : 2;
    const float sgn = wi == 1 ? 1.f : -1.f;
    // lane -> 2x4 output tile (tile row tr, tile column tc) of the 4x32 item, channel kq of the k-step:
    // patch rows = halo rows 2 tr + (0..3), patch columns = image x0 + 4 tc - 1 .. + 4 = halo columns 4 tc + 3 .. + 8
    const int tr_ = l16 >> 3, tc_ = l16 & 7;
    const int oa = kq * W2_PS + (2 * tr_ + pa0) * W2_RS + 4 * tc_ + 3;
    const int ob = kq * W2_PS + (2 * tr_ + pa1) * W2_RS + 4 * tc_ + 3;

    const int nch = p.nch3;

    // ---- raw tile staging (as conv_wino2) ----
    constexpr unsigned OOB = 0x40000000u;
    unsigned goff;
    bool cm[6], cmn[6];                             // patch column c of this lane's tile lies inside the image (item / next item)
    auto make_goff = [&](const Wino2Item& it) {
        const int row = lane / W2_GRP, grp = lane - row * W2_GRP;
        const int gy = it.y0 + row - 1, gx = it.x0 - 4 + 4 * grp;
        const bool ok = lane < W2_HR * W2_GRP && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goff = ok ? (unsigned)(gy * W + gx) * 4u : OOB;
        // (a 16-byte group that starts inside the image may run past its right edge into the next row when W % 4 != 0)
#pragma unroll
        for (int c = 0; c < 6; ++c) cmn[c] = it.x0 + 4 * tc_ - 1 + c < W;
    };
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto issue_dma = [&](int ib, int c, float* buf) {            // prologue only: chunk c of sample ib by LDS-DMA
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int kc = wi * 4 + g;
            const int ch = c * 16 + kc;
            const bool live = ch < p.Cin;
            const float* sbase = p.in + ((size_t)ib * p.Cin + (live ? ch : 0)) * HW;
            const __amdgpu_buffer_rsrc_t rs =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sbase), 0, live ? HW * 4 : 0, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(buf + kc * W2_PS), 16, (int)goff, 0, 0, 0);
        }
    };
    f32x4 stg[2];
    const unsigned HW4 = (unsigned)HW * 4u;
    auto plane_ptr = [&](int ib) { return p.in + ((size_t)ib * p.Cin + wi * 4) * HW; };
    __amdgpu_buffer_rsrc_t rs_st;
    auto stage_load = [&](int slot, int g) {
        stg[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_st, (int)goff, g * (int)HW4, 0));
    };
    auto stage_store = [&](int slot, float* buf, int g) {
        *reinterpret_cast<f32x4*>(buf + (wi * 4 + g) * W2_PS + lane * 4) = stg[slot];
    };

    // ---- weights ----
    const __amdgpu_buffer_rsrc_t rsw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w3), 0, 0x7FFFFFF0, 0x00020000);
    const int wlane = lane * 16;
    auto wbase = [&](int cb) -> int { return cb * nch * W3_CH_BYTES + wi * (4 * W3_KS_BYTES); };
    f32x4 aq[W3_Q];
    auto chunk_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // B operands of one k-step: six horizontal frequencies of this wave's row
    auto read_raw = [&](const float* base, float (&ra)[6], float (&rb)[6]) {
        const float* qa = base + oa;
        const float* qb = base + ob;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            ra[c] = qa[c];
            rb[c] = qb[c];
        }
    };
    auto transform = [&](const float (&ra)[6], const float (&rb)[6], float (&v)[W3_NF], const bool (&mk)[6]) {
        float r[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            r[c] = fmaf(sgn, rb[c], ra[c]);
            if (EDGE) r[c] = mk[c] ? r[c] : 0.f;
        }
        // F(4,3) B^T:  [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
        const float s24 = r[4] - 4.f * r[2], s13 = r[3] - 4.f * r[1];          // r4 - 4 r2,  r3 - 4 r1
        const float u24 = r[4] - r[2], u13 = 2.f * (r[3] - r[1]);              // r4 - r2,  2 (r3 - r1)
        v[0] = fmaf(4.f, r[0], fmaf(-5.f, r[2], r[4]));
        v[1] = s24 + s13;
        v[2] = s24 - s13;
        v[3] = u24 + u13;
        v[4] = u24 - u13;
        v[5] = fmaf(4.f, r[1], fmaf(-5.f, r[3], r[5]));
    };

    Wino2Item it;
    int l = 0;
    if (!decode(l, it)) return;
    make_goff(it);
#pragma unroll
    for (int c = 0; c < 6; ++c) cm[c] = cmn[c];
    issue_dma(it.b, 0, smem);
    int wb_it = wbase(it.cb);
#pragma unroll
    for (int q = 0; q < W3_Q; ++q)
        aq[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wlane + q * 1024, wb_it, 0));
    dma_barrier();
    int wcur = wb_it;
    const float* sstage = plane_ptr(it.b) + (size_t)16 * HW;
    int nb = 0;
    float v[2][W3_NF];
    {
        float ra[6], rb[6];
        read_raw(smem, ra, rb);
        transform(ra, rb, v[0], cm);
    }

    for (;;) {
        Wino2Item nx;
        l += 1;
        const bool have_next = decode(l, nx);
        if (!have_next) nx = it;
        const int wb_nx = wbase(nx.cb);
        const float* base_nx = plane_ptr(nx.b);
        f32x4 acc[MT][W3_NF];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < W3_NF; ++j) acc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int c = 0; c < nch; ++c) {
            const float* cur = smem + nb * W2_BUF;
            float* nxt = smem + (nb ^ 1) * W2_BUF;
            const bool last = __builtin_amdgcn_readfirstlane(c + 1 == nch) != 0;
            if (last) make_goff(nx);
            const int dch = last ? 0 : c + 1;
            const bool dval = !last || have_next;
            const int w_k[4] = {wcur + W3_KS_BYTES, wcur + 2 * W3_KS_BYTES, wcur + 3 * W3_KS_BYTES,
                                last ? wb_nx : wcur + W3_CH_BYTES};
            if (last) sstage = base_nx;
            const bool live = dval && dch * 16 + wi * 4 < p.Cin;
            rs_st = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sstage), 0, live ? 4 * (int)HW4 : 0, 0x00020000);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks == 3) {
                    __builtin_amdgcn_sched_barrier(0);
                    chunk_barrier();
                }
                const float* rsrc_ = ks < 3 ? cur + (ks + 1) * 4 * W2_PS : nxt;
                float ra[6], rb[6];
                // head of the k-step: everything that is not an MFMA
                if (ks == 1) { stage_store(0, nxt, 0); stage_store(1, nxt, 1); }
                if (ks == 2) { stage_store(0, nxt, 2); stage_store(1, nxt, 3); }
                read_raw(rsrc_, ra, rb);
                if (ks == 0) { stage_load(0, 0); stage_load(1, 1); }
                if (ks == 1) { stage_load(0, 2); stage_load(1, 3); }
                // (the operands built in k-step 3 of the last chunk belong to the next item's tile)
                bool mk[6];
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) mk[cc] = (ks == 3 && last) ? cmn[cc] : cm[cc];
                transform(ra, rb, v[(ks + 1) & 1], mk);
                __builtin_amdgcn_sched_barrier(0);
                // burst: 30 MFMAs in the order of the packed image (w3_pos_e), the A group of four slots refilled right
                // behind its last MFMA
#pragma unroll
                for (int pos = 0; pos < 32; ++pos) {
                    const int e = w3_pos_e(pos);
                    if (e >= 0) {
                        const int mt = e / W3_NF, j = e - mt * W3_NF;
                        acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[pos >> 2][pos & 3], v[ks & 1][j], acc[mt][j], 0, 0, 0);
                    }
                    if ((pos & 3) == 3) {
                        aq[pos >> 2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wlane + (pos >> 2) * 1024, w_k[ks], 0));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            nb ^= 1;
            wcur = w_k[3];
            sstage += (size_t)16 * HW;
        }

        // ---- output transform + epilogue: column half (6 -> 4) in registers, row half through LDS, 32 channels per pass ----
        const int tile = tid & 15;                     // reader role: 2x4 tile (tile row, tile column) ...
        const int trr = tile >> 3, tcr = tile & 7;
        const int cg = tid >> 4;                       // ... and channels cg + 16 k of a pass
        const int y = it.y0 + 2 * trr, x = it.x0 + 4 * tcr;
        using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
        const unsigned plane_b = (unsigned)HW * 4u;
        const unsigned samp_b = (unsigned)p.Cout * plane_b;
        const size_t samp_o = (size_t)it.b * p.Cout * HW;
        auto rsrc_of = [&](const float* base) {
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base ? base + samp_o : p.zero), 0,
                                                     base ? samp_b : 0u, 0x00020000);
        };
        const __amdgpu_buffer_rsrc_t rs_out = rsrc_of(p.out), rs_res = rsrc_of(p.resid);
        const __amdgpu_buffer_rsrc_t rs_aux = rsrc_of(ACT == 2 ? p.aux : nullptr);
        const __amdgpu_buffer_rsrc_t rs_pre = rsrc_of(ACT == 1 ? p.out_pre : nullptr);
        const __amdgpu_buffer_rsrc_t rs_bias = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.bias ? p.bias : p.zero), 0, p.bias ? (unsigned)(p.coblks * MT * 16) * 4u : 0u, 0x00020000);
        const bool pix_ok = (y < H) & (x < W);
        const bool okpp[2] = {pix_ok, bool(pix_ok & (y + 1 < H))};
        const int cl_lim = p.Cout - it.cb * (MT * 16) - cg;
        const unsigned pix_o = ((unsigned)(it.cb * (MT * 16) + cg) * (unsigned)HW + (unsigned)(y * W + x)) * 4u;
        // padded rows (ConvArgs::Wt): the quad's columns beyond the true width are written as zeros
        const bool padded = p.Wt > 0 && p.Wt < W;
        const int nvq = p.Wt - x;
        const f32x4 pm{nvq > 0 ? 1.f : 0.f, nvq > 1 ? 1.f : 0.f, nvq > 2 ? 1.f : 0.f, nvq > 3 ? 1.f : 0.f};
        auto in_block = [](int m0, int k) { return m0 * 16 + 16 * k + 16 <= MT * 16; };
        auto off_of = [&](int m0, int k, int pp) -> unsigned {
            const bool ok = okpp[pp] & (m0 * 16 + 16 * k < cl_lim);
            unsigned o = ok ? pix_o + (unsigned)(m0 * 16 + 16 * k) * plane_b + (unsigned)(pp * W) * 4u : OOB;
            asm volatile("" : "+v"(o));
            return o;
        };
        // 4 pixels per access; at the right image edge (EDGE builds) in 8- or 4-byte pieces with per-piece validity
        using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
        auto piece_off = [&](unsigned o, int px) -> unsigned {          // offset of pixel px of the quad, or OOB
            unsigned q = (o != OOB && x + px < W) ? o + 4u * px : OOB;
            asm volatile("" : "+v"(q));
            return q;
        };
        auto ld4 = [&](const __amdgpu_buffer_rsrc_t& r, unsigned o) -> f32x4 {
            if constexpr (EDGE == 0) {
                return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)o, 0, 0));
            } else if constexpr (EDGE == 1) {
                const u32x2 a0 = __builtin_amdgcn_raw_buffer_load_b64(r, (int)piece_off(o, 0), 0, 0);
                const u32x2 a1 = __builtin_amdgcn_raw_buffer_load_b64(r, (int)piece_off(o, 2), 0, 0);
                return __builtin_bit_cast(f32x4, u32x4{a0[0], a0[1], a1[0], a1[1]});
            } else {
                f32x4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)piece_off(o, e), 0, 0));
                return t;
            }
        };
        auto st4 = [&](const __amdgpu_buffer_rsrc_t& r, unsigned o, f32x4 vv) {
            const u32x4 u = __builtin_bit_cast(u32x4, vv);
            if constexpr (EDGE == 0) {
                __builtin_amdgcn_raw_buffer_store_b128(u, r, (int)o, 0, 0);
            } else if constexpr (EDGE == 1) {
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{u[0], u[1]}, r, (int)piece_off(o, 0), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{u[2], u[3]}, r, (int)piece_off(o, 2), 0, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) __builtin_amdgcn_raw_buffer_store_b32(u[e], r, (int)piece_off(o, e), 0, 0);
            }
        };
        f32x4 rs_v[2][2], ax_v[2][2];
        float bs_v[2];
        auto prefetch = [&](int m0) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!in_block(m0, k)) continue;
                bs_v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rs_bias, (it.cb * (MT * 16) + m0 * 16 + cg + 16 * k) * 4, 0, 0));
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const unsigned o = off_of(m0, k, pp);
                    rs_v[k][pp] = ld4(rs_res, o);
                    if (ACT == 2) ax_v[k][pp] = ld4(rs_aux, o);
                }
            }
        };
#pragma unroll
        for (int m0 = 0; m0 < MT; m0 += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (m0 + h < MT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // (M A4)[i][q]:  A4^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
                        const float m_0 = acc[m0 + h][0][r], m_1 = acc[m0 + h][1][r], m_2 = acc[m0 + h][2][r];
                        const float m_3 = acc[m0 + h][3][r], m_4 = acc[m0 + h][4][r], m_5 = acc[m0 + h][5][r];
                        const float s12 = m_1 + m_2, d12 = m_1 - m_2, s34 = m_3 + m_4, d34 = m_3 - m_4;
                        f32x4 t{m_0 + s12 + s34, fmaf(2.f, d34, d12), fmaf(4.f, s34, s12), fmaf(8.f, d34, d12) + m_5};
                        *reinterpret_cast<f32x4*>(sX + ((wi * 32 + h * 16 + kq * 4 + r) * 16 + l16) * 4) = t;
                    }
                }
            }
            if (m0 == 0) prefetch(0);
            lds_barrier();
            f32x4 yv[2][2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!in_block(m0, k)) continue;
                const int cl = cg + 16 * k;                        // channel of the pass (0..31)
                f32x4 t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) t[i] = *reinterpret_cast<const f32x4*>(sX + ((i * 32 + cl) * 16 + tile) * 4);
                yv[k][0] = t[0] + t[1] + t[2];                     // Y[pp] = sum_i A2^T[pp][i] t[i]
                yv[k][1] = t[1] - t[2] - t[3];
            }
            lds_barrier();
            f32x4 val[2][2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!in_block(m0, k)) continue;
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    f32x4 w_ = yv[k][pp] + bs_v[k];
                    if (ACT == 1) {
                        yv[k][pp] = w_;                            // pre-activation (the training forward saves it)
                        w_ = gelu_erf4(w_);
                    } else if (ACT == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) w_[e] *= gelu_erf_grad(ax_v[k][pp][e]);
                    }
                    val[k][pp] = w_ + rs_v[k][pp];
                }
            }
            if (m0 + 2 < MT) prefetch(m0 + 2);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!in_block(m0, k)) continue;
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const unsigned o = off_of(m0, k, pp);
                    if (ACT == 1) st4(rs_pre, o, yv[k][pp]);       // (empty descriptor when out_pre is null: dropped)
                    st4(rs_out, o, padded ? val[k][pp] * pm : val[k][pp]);
                }
            }
        }
        if (!have_next) break;
        it = nx;                                   // goff already describes nx
        wb_it = wb_nx;
#pragma unroll
        for (int c = 0; c < 6; ++c) cm[c] = cmn[c];
    }
}

inline int conv_wino3_launch(const ConvArgs& a_in, hipStream_t st) {
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
    a.tilesX = (a.W + W2_TW - 1) / W2_TW;
    a.tilesY = (a.H + W2_TH - 1) / W2_TH;
    a.ntiles = a.B * a.tilesX * a.tilesY;
    a.tiles_per_xcd = (a.ntiles + 7) / 8;
    a.mtp = W3_MT;
    const int ipx = a.tiles_per_xcd * a.coblks;
    int wpx = wino2_cu_count() / 8 * 2;
    if (wpx < 1) wpx = 1;
    if (wpx > ipx) wpx = ipx;
    const unsigned grid = (unsigned)(wpx * 8);
    constexpr size_t lds = W2_LDS_FLOATS * sizeof(float);
#define W3_GO(ACT, EDGE) hipLaunchKernelGGL((conv_wino3_kernel<ACT, EDGE>), dim3(grid), dim3(W2_THREADS), lds, st, a, ipx, wpx)
    const int edge = a.W % 4 == 0 ? 0 : (a.W % 2 == 0 ? 1 : 2);
    switch ((a.act & 0xff) * 3 + edge) {
        case 0: W3_GO(0, 0); break;
        case 1: W3_GO(0, 1); break;
        case 2: W3_GO(0, 2); break;
        case 3: W3_GO(1, 0); break;
        case 4: W3_GO(1, 1); break;
        case 5: W3_GO(1, 2); break;
        case 6: W3_GO(2, 0); break;
        case 7: W3_GO(2, 1); break;
        default: W3_GO(2, 2);
    }
#undef W3_GO
    if (rec) {
        (void)hipEventRecord(prof.ev[2 * prof.used + 1], st);
        const double fl = 2.0 * a.B * a.H * (a.Wt > 0 ? a.Wt : a.W) * (double)a.Cout * 9.0 * a.Cin;   // algorithmic (direct-conv) FLOPs
        prof.note(1, fl, fl * (24.0 / 72.0), 3);                                      // F(2x4): 24 multiplies per 8 outputs
    }
    SINDDM_LAUNCH_CHECK();
    return 0;
}

}  // namespace sinddm
