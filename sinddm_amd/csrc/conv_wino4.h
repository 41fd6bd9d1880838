// Winograd F(2x4, 3x3) 3x3 convolution on the gfx950 fp32 matrix cores -- fourth generation: ONE wave per SIMD with
// the whole 512-entry register file (forward and data gradient of the big launches).
//
// Why another kernel (round-2 verdict on conv_wino3.h): with one n-tile per wave every MFMA consumed a fresh A
// register from L2 (256 B per 2048 FLOP, 11.8 TB/s of L2 traffic per launch), the raw-patch reads were 4-way bank
// conflicted, the two 80-channel blocks of a tile and the 6x40 halo of a 4x32 tile re-read the input 3.2x, and the
// chip clocked down to 2.05 GHz under that traffic.  The two-workgroups-per-CU skeleton cannot share weights: sharing
// needs both n-tiles in the same K phase, i.e. in the same wave.  So:
//   * one persistent 4-wave workgroup per CU, __launch_bounds__(256, 1): a wave owns its SIMD and 512 registers.  Wave i
//     = vertical frequency row i (as before) but of TWO n-tiles: a work item is an 8x32-pixel tile (n-tile h = rows
//     4h .. 4h+3 = 2 x 8 output tiles of 2x4) x 80 output channels; 2 x 5 x 6 accumulator tiles = 240 registers (the
//     compiler keeps them in the AGPR half), 60 MFMAs per k-step, every A fragment feeds two MFMAs: 128 B of L2
//     traffic per MFMA, half of conv_wino3.  Same packed weight image as conv_wino3 ([co-blk][chunk][i][k-step][q][lane][4]).
//   * with nobody else on the SIMD there is no partner to hide a k-step head, so there is no head: the k-step is
//     software pipelined INSIDE the wave.  A 16x16x4 fp32 MFMA occupies the matrix pipe for 32 cycles and issues in 4;
//     the raw-patch reads, the 18 VALU of each n-tile's input transform, the weight refills and the raw-tile staging
//     of the NEXT k-step / chunk are dealt one or two per MFMA into the 60 gaps (pinned with sched_barrier: <= 3
//     fillers per gap, the gap hides 5).
//   * A registers are a ring of TWO k-steps (64 registers): a refill is issued right behind the last MFMA that reads
//     the group it overwrites and is needed 120 MFMAs (~3 800 cycles) later -- an L2 round trip under load and even an
//     HBM-latency raw-tile load in front of it (vector memory returns in order) fit inside that.
//   * raw halo tile 16 channels x 10 rows x 40 columns, row stride 41 / plane stride 411 floats: the twelve 4-byte
//     patch reads of a (lane, n-tile, k-step) are bank-conflict free (2 tr RS = 2 and kq PS = 3 (mod 4) spread the four
//     (channel, tile-row) combinations of a 32-lane group over the four residues the 4 tc column term leaves free);
//     the halo of an 8x32 tile is 1.56x its pixels (conv_wino3: 1.875x).
//   * the work-item list of an XCD hands the two 80-channel blocks of a tile to NEIGHBOURING workgroups at the same
//     time: the second reader of a raw tile finds it in the XCD's L2.
//   * epilogue as conv_wino3 (column half of the output transform in registers, row half over the four waves through
//     LDS, 32 channels per pass), now for 32 tiles per pass.  It is exposed (no second workgroup runs meanwhile); the
//     next item's first raw chunk, weights and B operands are already in flight / in registers when it starts.
// Restrictions (the caller falls back to conv_wino3 / conv_wino2 otherwise): C_out % 80 == 0, C_in % 16 == 0, at least
// SINDDM_V4_MIN_ITEMS_PER_CU work items per CU.  Same ConvArgs / epilogue contract as conv_wino2.h.
#pragma once
#include <utility>
#include "conv_wino3.h"

namespace sinddm {

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}).  The k-step body
// is far beyond the size at which `#pragma unroll` still unrolls fully (pragma-unroll-threshold), and a loop that stays
// rolled turns the accumulator array into scratch memory.
template <class F, int... I>
__device__ __forceinline__ void w4_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void w4_static_for(F&& f) {
    w4_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Compile-time timing ablations (-DW4_ABL=bits; results are WRONG, never ship):
//   1 no raw-tile staging   2 weights loaded once   4 no raw-patch LDS reads   8 no epilogue   16 no input transform
//   128 raw-tile requests of a chunk contiguous (tile-major what-if)   256 epilogue without global loads / stores (2048: without the loads, 4096: without the stores)   512 ... without its LDS writes   1024 ... without its LDS reads
#ifndef W4_ABL
#define W4_ABL 0
#endif
#ifdef W4_TIMING
// s_memtime stamps of every workgroup's item 3 (debug builds only; tools/w4_seg.py): [launch % 8][workgroup][wave][32]
__device__ unsigned long long g_w4_seg[8 * 256 * 4 * 32];
#define W4_SEG(slot) do { if (seg) g_w4_seg[((p.mtp * 256 + blockIdx.x) * 4 + wi) * 32 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W4_SEG(slot) do {} while (0)
#endif
#ifdef W4_KSTAMP
// s_memtime stamps inside the k-steps of chunk 2 of every workgroup's item 3 (debug builds; tools/w4_kstamp.py):
// [launch % 8][workgroup][wave][k-step 4][stamp 16]
__device__ unsigned long long g_w4_ks[8 * 256 * 4 * 64];
#define W4_KS(ks, i) do { if (kst) g_w4_ks[((p.mtp * 256 + blockIdx.x) * 4 + wi) * 64 + (ks) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W4_KS(ks, i) do {} while (0)
#endif
#ifndef W4_STG_SLOT
#define W4_STG_SLOT 24         // first of the four slots of k-steps 0 / 1 in which the raw-tile groups of the next chunk are requested
#endif
#ifndef W4_PF_KS
#define W4_PF_KS 3            // k-step / slot of a chunk in which the epilogue's pass-0 operands are requested (slot -1: at the
#define W4_PF_SLOT 58         // top of the epilogue instead).  Late in the LAST chunk: what is queued behind these loads
#endif                        // (vector memory returns in order) is only needed after the epilogue
constexpr int W4_PF_AHEAD = 1;  // the epilogue's per-pass operands are requested one pass ahead (two: no gain, profiles/r04_w4_stagger_prefetch_ab.txt)
#ifndef W4_XF_SLOT
#define W4_XF_SLOT 40          // the slot of a k-step behind whose MFMA the input-transform burst of the next k-step sits
#endif
// The 240 accumulator registers are the AGPRs a0 .. a239, addressed by NUMBER from inline asm: tile T = (h * 5 + mt) * 6 + j
// lives in a[4T : 4T+3].  (Left to the register allocator, a third of the MFMAs came out with dst != src and every
// chunk iteration paid 300-500 v_accvgpr copies to undo the permutation at the loop back edge.)  The compiler never
// allocates an AGPR in this kernel -- it sees no MFMA and spills nothing -- and learns the count from the clobber
// list of w4_acc_declare(); `tests/test_build_isa.py` checks that every AGPR access in the ISA is one of these asms.
// ZC: the accumulator input is the constant 0 -- the first k-step of a work item starts its sums that way, so the 240
// registers are never zeroed (240 v_accvgpr_write per wave and item in round 3)
template <int T, bool ZC>
__device__ __forceinline__ void w4_mfma(float a, float b) {
    if constexpr (ZC)
        asm volatile("v_mfma_f32_16x16x4_f32 a[%c2:%c3], %0, %1, 0" ::"v"(a), "v"(b), "n"(4 * T), "n"(4 * T + 3));
    else
        asm volatile("v_mfma_f32_16x16x4_f32 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "n"(4 * T), "n"(4 * T + 3));
}
template <int R>
__device__ __forceinline__ float w4_acc_read() {
    float r;
    asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(r) : "n"(R));
    return r;
}
__device__ __forceinline__ void w4_acc_declare() { asm volatile("" ::: "a0", "a239"); }

// Position pins.  The pre-RA list scheduler orders pure nodes (VALU, MFMA) by register pressure, not by source position;
// an empty volatile asm that "modifies" a value is ordered with every other side-effecting node (loads, barriers, the
// sched_barrier fences), so an operation whose inputs are pinned before it and whose result is pinned behind it stays
// in the MFMA gap it was dealt to.  (Only the "v" form: behind an asm that touches an AGPR the hazard recogniser puts an
// s_nop in front of every following VALU / MFMA.)
#define W4_PINV(x) asm volatile("" : "+v"(x))
constexpr int W4_TH = 8, W4_TW = 32;          // pixel tile of a work item
constexpr int W4_HR = W4_TH + 2;              // halo rows
constexpr int W4_RS = 41;                     // LDS row stride (40 columns: image x0-4 .. x0+35)
constexpr int W4_PS = 411;                    // LDS plane stride (10 x 41 = 410)
constexpr int W4_BUF = 8192;                  // floats per raw-tile buffer: 16 planes (6 576) padded to 32 KB -- a power of two, so that
                                              // the two buffers swap by XOR of every LDS address register with 0x8000
static_assert(16 * W4_PS <= W4_BUF, "raw-tile buffer");
constexpr int W4_XB = 4 * 4 * 32 * 16;        // one exchange buffer: an m-tile's column-transformed values, [i][column][32 tiles][16 channels]
constexpr int W4_XCH = 2 * W4_XB;             // two of them: m-tile p + 1 is written while p is read
constexpr int W4_LDS_FLOATS = 2 * W4_BUF + W4_XCH;   // 64 + 64 KB
static_assert(W4_LDS_FLOATS * 4 <= 160 * 1024, "LDS of a gfx950 CU");
static_assert((2 * W4_BUF) % 4 == 0, "exchange area must stay 16-byte aligned");

struct Wino4Item {
    int b, y0, x0, cb;
};

template <int ACT, int EDGE>
__global__ __launch_bounds__(256, 1) void conv_wino4_kernel(ConvArgs p, int items_per_xcd, int wg_per_xcd) {
    constexpr int MT = W3_MT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sX = smem + 2 * W4_BUF;

    // persistent workgroups: XCD `xcd` owns a contiguous range of tiles; its (tile, block) items go round-robin over
    // its workgroups, so the blocks of one tile run at the same time on neighbouring workgroups
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
    const int wi = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = vertical frequency row i
    const int l16 = lane & 15, kq = lane >> 4;
    const int H = p.H, W = p.W;
    const int HW = H * W;

    // vertical B^T rows (F(2,3)):  0: d0 - d2   1: d1 + d2   2: d1 - d2 (U_2j stored negated)   3: d1 - d3
    const int pa0 = wi == 0 ? 0 : 1, pa1 = wi == 3 ? 3 : 2;
    const float sgn = wi == 1 ? 1.f : -1.f;
    // lane -> 2x4 output tile (tile row tr, tile column tc) of an n-tile, channel kq of the k-step; n-tile h covers the
    // item's rows 4h .. 4h+3: patch rows = halo rows 4h + 2 tr + (0..3), patch columns = halo columns 4 tc + 3 .. + 8
    const int tr_ = l16 >> 3, tc_ = l16 & 7;
    // LDS BYTE addresses of the two patch rows ([row a / row b][n-tile]) in the buffer being read; every access of the
    // main loop is such a register + an immediate, and `^= 0x8000` at the chunk barrier moves them to the other buffer
    typedef __attribute__((address_space(3))) float lds_f;
    const unsigned lds0 = (unsigned)(size_t)(lds_f*)smem;
    unsigned rd[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        rd[0][h] = lds0 + 4u * (kq * W4_PS + (4 * h + 2 * tr_ + pa0) * W4_RS + 4 * tc_ + 3);
        rd[1][h] = lds0 + 4u * (kq * W4_PS + (4 * h + 2 * tr_ + pa1) * W4_RS + 4 * tc_ + 3);
    }
    auto lds_ld = [](unsigned addr, int foff) __attribute__((always_inline)) { return ((const lds_f*)addr)[foff]; };
    const int nch = p.nch3;

    // ---- raw tile staging: 100 sixteen-byte groups per channel plane = two loads per channel ----
    constexpr unsigned OOB = 0x40000000u;
    // lane -> (halo row, 16-byte group) of its two requests per plane.  First request: rows 0..7 x groups 0..7 -- with the
    // odd row stride a half wave's four ds_write_b32 (rows r..r+3 x 8 groups) hit 32 distinct banks; second request: groups
    // 8, 9 of all ten rows (lanes 0..19) and groups 0..7 of rows 8, 9 (lanes 20..35); lanes 36..63 repeat lanes 0..27 --
    // same data to the same address -- so that neither the load nor the LDS write needs an exec mask.  (Row-major over
    // all ten groups, the first form, was 2- to 3-way conflicted: 0.17 of the kernel's LDS cycles.)
    auto stage_li = [&](int s) {
        if (s == 0) return (lane >> 3) * 10 + (lane & 7);
        const int m = lane < 36 ? lane : lane - 36;
        return m < 20 ? (m >> 1) * 10 + 8 + (m & 1) : (8 + ((m - 20) >> 3)) * 10 + ((m - 20) & 7);
    };
    unsigned goff[2];
    int swo[2];                                     // LDS float offset of this lane's group inside a plane
    // EDGE builds: patch column c of this lane's tiles lies inside the image iff c < nv (item) / nvn (next item).  (A count
    // in a VGPR, compared inside the transform burst: twelve per-lane bools in SGPR pairs ran the kernel out of SGPRs.)
    int nv = 6, nvn = 6;
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
            goff[s] = ok ? (unsigned)(gy * W + gx) * 4u : OOB;
        }
        // (a 16-byte group that starts inside the image may run past its right edge into the next row when W % 4 != 0)
        nvn = W - (it.x0 + 4 * tc_ - 1);
    };
    // the eight raw-tile requests of a chunk go out W4_STG_GAP MFMA slots apart, each is written to LDS W4_STG_DIST slots
    // later (two batches of four back to back: +2-3 % per launch at C3 -- bursts of HBM-latency loads hold up the weight
    // refills queued behind them, vector memory returns in order)
    constexpr int W4_STG_GAP = 15, W4_STG_DIST = 60;
    f32x4 stg[8];                                   // the chunk's eight 16-byte groups in flight
    const unsigned HW4 = (unsigned)HW * 4u;
    auto plane_ptr = [&](int ib) { return p.in + ((size_t)ib * p.Cin + wi * 4) * HW; };
    __amdgpu_buffer_rsrc_t rs_st;
    int lin_soff = 0;
    auto tile_lin = [&](const Wino4Item& g) { return (g.y0 / W4_TH) * p.tilesX + g.x0 / W4_TW; };
    auto stage_load = [&](int n) __attribute__((always_inline)) {                  // n = 2 g + s: group s of channel wi*4 + g
        if (W4_ABL & 1) return;
        // (128: the chunk's eight requests of a wave read ONE contiguous 8 KB of the image -- what a tile-major activation
        // layout would make of the ten 160-byte row segments x 4 planes; same bytes, same sharing between the two blocks)
        if (W4_ABL & 128) stg[n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_st, lane * 16 + n * 1024, lin_soff, 0));
        else
        stg[n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_st, (W4_ABL & 64) ? lane * 16 : (int)goff[n & 1], (n >> 1) * (int)HW4, 0));
    };
    // LDS writes of a staged group: four ds_write_b32 at (register + IMMEDIATE) -- left to the compiler they became
    // ds_write2_b32 pairs whose 8-bit offsets need a v_add_u32 per pair, i.e. lone VALU in the MFMA stream.  (The asm
    // hides four LDS operations from the compiler's lgkmcnt bookkeeping, which only makes its waits for earlier reads
    // stricter; the chunk barrier waits for lgkmcnt(0) itself.)
    // (swb: byte address of the lane's group in plane wi*4 of the buffer being WRITTEN -- the other one)
    unsigned swb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) swb[s] = lds0 + 4u * (W4_BUF + wi * 4 * W4_PS + swo[s]);
    // (ONE write per MFMA slot -- NE = 4 n + e: tools/ubench/mfma_fillers.hip: two LDS writes in a gap are free, four cost 16 cycles)
    auto stage_store = [&](auto NE) __attribute__((always_inline)) {
        constexpr int n = decltype(NE)::value >> 2, e = decltype(NE)::value & 3;
        if (W4_ABL & 1) return;
        constexpr int off = (n >> 1) * W4_PS * 4;
        static_assert(off + 12 < 65536, "ds_write_b32 immediate offset");
        const unsigned addr = swb[n & 1];                   // (locals: asm operands do not capture in a generic lambda)
        const float val = stg[n][e];
        asm volatile("ds_write_b32 %0, %1 offset:%c2" ::"v"(addr), "v"(val), "n"(off + 4 * e) : "memory");
    };

    // ---- weights ----
    const __amdgpu_buffer_rsrc_t rsw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w3), 0, 0x7FFFFFF0, 0x00020000);
    const int wlane = lane * 16;
    auto wbase = [&](int cb) -> int { return cb * nch * W3_CH_BYTES + wi * (4 * W3_KS_BYTES); };
    f32x4 aq[2][W3_Q];                              // ring of two k-steps (groups 0-2, 4-6: four fragments)
    // (groups 3 and 7 of the packed order -- w3_pos_e -- hold three fragments + a padding slot: loaded as 12 bytes into
    // their own rings; as the idle quarter of a 16-byte load the dead register was reused at once and the waitcnt pass
    // answered the pending load into it with a vmcnt(0) drain)
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    using f32x3 = __attribute__((ext_vector_type(3))) float;
    f32x3 aq3[2][2];
    auto load_a = [&](int ring, int q, int soff) __attribute__((always_inline)) {
        if ((q & 3) != 3)
            aq[ring][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wlane + q * 1024, soff, 0));
        else
            aq3[ring][q >> 2] = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(rsw, wlane + q * 1024, soff, 0));
    };
    auto chunk_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // ---- input transform of BOTH n-tiles of a k-step as one burst of packed fp32 VALU ----
    // tools/ubench/mfma_fillers.hip: a VALU instruction behind an fp32 MFMA of the SAME wave waits ~8 cycles for the
    // matrix pipe and then issues at 4 cycles per instruction with the next MFMA held back meanwhile (a wave's own VALU
    // and MFMAs never overlap; only another wave's do) -- one VALU per gap costs 12.4 cycles, 36 in one burst 5.8 each,
    // 18 v_pk_*_f32 in one burst 7 each.  So the two n-tiles are transformed together, (n-tile 0, n-tile 1) in the halves
    // of a register pair: 18 packed instructions per k-step, in ONE block between two MFMAs.
    const f32x2 sgn2{sgn, sgn};
    f32x2 n1{-1.f, -1.f};                           // opaque to the optimiser (see the epilogue)
    asm volatile("" : "+v"(n1));
    const f32x4 n4{n1.x, n1.y, n1.x, n1.y};
    auto xf_burst = [&](f32x2 (&ra)[6], f32x2 (&rb)[6], f32x2 (&v)[W3_NF], int nvalid) __attribute__((always_inline)) {
        // (pins: everything below stays behind this point -- the loads have landed -- and in front of the next MFMA)
        asm volatile("" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]));
        asm volatile("" : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(rb[4]), "+v"(rb[5]));
        if (W4_ABL & 16) {
#pragma unroll
            for (int c = 0; c < 6; ++c) v[c] = ra[c];
        } else {
            f32x2 r[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                r[c] = sgn2 * rb[c] + ra[c];
                // (columns 0 and 1 of a tile that has any pixel inside the image are inside it)
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
    w4_acc_declare();
    make_goff(it);
    nv = nvn;
    int wb_it = wbase(it.cb);
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
        for (int q = 0; q < W3_Q; ++q)
            load_a(r2, q, wb_it + r2 * W3_KS_BYTES);
    // first chunk of the first item
    auto stage_store0 = [&](int n) {
        float* d = smem + (wi * 4 + (n >> 1)) * W4_PS + swo[n & 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = stg[n][e];
    };
    rs_st = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(plane_ptr(it.b)), 0, 4 * (int)HW4, 0x00020000);
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
        for (int n = 0; n < 4; ++n) stage_load(4 * hb + n);
#pragma unroll
        for (int n = 0; n < 4; ++n) stage_store0(4 * hb + n);
    }
    __syncthreads();
    int wcur = wb_it;
    const float* sstage = plane_ptr(it.b) + (size_t)16 * HW;
    f32x2 v[2][W3_NF];                              // [k-step parity][frequency] (n-tile 0, n-tile 1)
    f32x2 raw[2][6];                                // [row a / row b][column]      (n-tile 0, n-tile 1)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        raw[0][c] = f32x2{lds_ld(rd[0][0], c), lds_ld(rd[0][1], c)};
        raw[1][c] = f32x2{lds_ld(rd[1][0], c), lds_ld(rd[1][1], c)};
    }
    xf_burst(raw[0], raw[1], v[0], nv);

    // ---- epilogue reader role: thread = 2x4 tile (n-tile, tile row, tile column) x TWO consecutive channels (half `hf` of
    // quad kqr of the pass's m-tile).  An accumulator tile's four registers are four consecutive channels, so the
    // exchange area is channel-minor: 16-byte pieces on the way in (writer lane (l16, kq): 1 KB contiguous per wave and
    // instruction), 8-byte pieces on the way out (reader lanes (hf, kqr, tile): 512 B contiguous) -- conflict-free both ways
    const int hf = tid & 1, kqr = (tid >> 1) & 3;
    const int tile = tid >> 3;
    const int hr = tile >> 4, trr = (tile >> 3) & 1, tcr = tile & 7;
    const int cg = kqr * 4 + hf * 2;              // first of the thread's two channels inside the m-tile
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
    // outputs / epilogue operands of tensors far beyond the 256 MB of last-level cache are written / read once: non-temporal
    // hint (C3, 8.6 GB per tensor: -0.6 % per launch, -1.4 % per step; C2, 0.47 GB: +2 % -- profiles/r04_w4_nontemporal_ab.txt)
    const bool nt = __builtin_amdgcn_readfirstlane((size_t)p.B * p.Cout * HW > ((size_t)1 << 28)) != 0;
    // a 4-pixel output quad goes out in NP pieces: one 16-byte access, or at images with W % 4 != 0 two 8-byte / four
    // 4-byte ones with per-piece validity.  vo[pp][piece] = byte offset of (channel cg, row y + pp, pixel x + piece
    // start) inside the sample, or OOB (hardware drop / zero fill); the channel of a pass rides in the SCALAR offset.
    constexpr int EE = EDGE;
    constexpr int NP = EE == 0 ? 1 : (EE == 1 ? 2 : 4);
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
    unsigned vo[2][NP], vo_nx[2][NP];
    ep_geo(it, vo);
    // padded rows (ConvArgs::Wt): the quad's columns beyond the true width are written as zeros
    const bool padded = __builtin_amdgcn_readfirstlane(p.Wt > 0 && p.Wt < W) != 0;
    auto pad_mask = [&](const Wino4Item& g) __attribute__((always_inline)) -> f32x4 {
        const int nvq = p.Wt - (g.x0 + 4 * tcr);
        return f32x4{nvq > 0 ? 1.f : 0.f, nvq > 1 ? 1.f : 0.f, nvq > 2 ? 1.f : 0.f, nvq > 3 ? 1.f : 0.f};
    };
    f32x4 pm = {1.f, 1.f, 1.f, 1.f};
    if (padded) pm = pad_mask(it);
    auto ep_load = [&](const __amdgpu_buffer_rsrc_t& r, int pp, int soff) __attribute__((always_inline)) -> f32x4 {
        if constexpr (EE == 0) {
            if (nt) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)vo[pp][0], soff, 2));
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)vo[pp][0], soff, 0));
        } else if constexpr (EE == 1) {
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
        if (W4_ABL & (256 | 4096)) { asm volatile("" ::"v"(vv)); return; }
        const u32x4 u = __builtin_bit_cast(u32x4, vv);
        if constexpr (EE == 0) {
            // (store + one wait state as ONE asm: a 16-byte buffer store reads its data registers after it has issued,
            // the compiler's hazard recogniser assumes that cannot bite when the scalar offset is a register and lets
            // the next channel's v_pk_add overwrite them in the following cycle -- on gfx950 it does bite)
            const unsigned voff = vo[pp][0];
            if (nt) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 1" ::"v"(u), "v"(voff), "s"(r), "s"(soff) : "memory");
            else asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(u), "v"(voff), "s"(r), "s"(soff) : "memory");
        } else if constexpr (EE == 1) {
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{u[0], u[1]}, r, (int)vo[pp][0], soff, 0);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{u[2], u[3]}, r, (int)vo[pp][1], soff, 0);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) __builtin_amdgcn_raw_buffer_store_b32(u[e], r, (int)vo[pp][e], soff, 0);
        }
    };
    // two register sets of pass operands: a pass computes from one while the next pass's arrive in the other
    constexpr int NS = W4_PF_AHEAD + 1;
    f32x4 opv[NS][2][2];                            // [set][channel][output row]
    f32x2 bsv[NS];                                  // [set] bias of the two channels
    // the lane's slot in its wave's window of an exchange buffer (writer side) and the thread's (reader side)
    float* xwrite = sX + (wi * 4 * 32 + l16) * 16 + kq * 4;
    const float* xread = sX + tile * 16 + kqr * 4 + hf * 2;

    for (;;) {
#ifdef W4_TIMING
        const bool seg = l == 3 && blockIdx.x < 256;
#endif
        W4_SEG(0);
        Wino4Item nx;
        l += 1;
        const bool have_next = decode(l, nx);
        if (!have_next) nx = it;
        const int wb_nx = wbase(nx.cb);
        const float* base_nx = plane_ptr(nx.b);
        // epilogue descriptors of THIS item (scalar work, free beside the MFMA stream): its pass-0 operands are
        // requested inside the last chunk of the main loop
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
        auto ep_soff = [&](int m0, int r) -> int { return (cb_ch + m0 * 16 + r) * (int)plane_b; };
        // operands of one pass (= m-tile m0) into register set SET: the bias of the thread's two channels and, per channel
        // and output row, the residual (ACT 0 / 1) or the pre-activation whose GELU' multiplies the result (ACT 2; the
        // data-gradient convs carry no residual)
        auto ep_fetch = [&](auto SET, int m0, const __amdgpu_buffer_rsrc_t& rop, const __amdgpu_buffer_rsrc_t& rbias)
                            __attribute__((always_inline)) {
            constexpr int set = decltype(SET)::value;
            if (W4_ABL & (256 | 2048)) return;
            if (ACT != 2)
                bsv[set] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rbias, cg * 4, (cb_ch + m0 * 16) * 4, 0));
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) opv[set][r][pp] = ep_load(rop, pp, ep_soff(m0, r));
        };
        auto chunk = [&](int c, auto ZC) __attribute__((always_inline)) {
            constexpr bool zc = decltype(ZC)::value;     // the item's first chunk: its k-step 0 starts the sums (C = 0)
#ifdef W4_KSTAMP
            const bool kst = l == 4 && c == 2 && blockIdx.x < 256;
#endif
            const bool last = __builtin_amdgcn_readfirstlane(c + 1 == nch) != 0;
            if (last) make_goff(nx);
            const int dch = last ? 0 : c + 1;
            const bool dval = !last || have_next;
            // weights of k-step + 2 (the ring): k-steps 2, 3 of this chunk, then 0, 1 of the next chunk / item
            const int wnext = last ? wb_nx : wcur + W3_CH_BYTES;
            const int w_pre[4] = {wcur + 2 * W3_KS_BYTES, wcur + 3 * W3_KS_BYTES, wnext, wnext + W3_KS_BYTES};
            if (last) sstage = base_nx;
            const bool live = dval && dch * 16 + wi * 4 < p.Cin;
            rs_st = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sstage), 0, live && !(W4_ABL & 32) ? 4 * (int)HW4 : 0, 0x00020000);
            if (W4_ABL & 128) {
                rs_st = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(last ? base_nx : plane_ptr(it.b)), 0, live ? (1 << 26) : 0, 0x00020000);
                lin_soff = ((last ? tile_lin(nx) : tile_lin(it)) * nch + dch) * 8192;
            }
            w4_static_for<4>([&](auto KS) __attribute__((always_inline)) {
                constexpr int ks = decltype(KS)::value;
                if constexpr (ks == 3) {
                    // every wave has written its planes of the next chunk: swap the buffers (six VALU in one block,
                    // where the wave waits anyway)
                    __builtin_amdgcn_sched_barrier(0);
                    chunk_barrier();
#pragma unroll
                    for (int i = 0; i < 4; ++i) rd[i >> 1][i & 1] ^= 4u * W4_BUF;
                    swb[0] ^= 4u * W4_BUF;
                    swb[1] ^= 4u * W4_BUF;
                    asm volatile("" : "+v"(rd[0][0]), "+v"(rd[0][1]), "+v"(rd[1][0]), "+v"(rd[1][1]), "+v"(swb[0]), "+v"(swb[1]));
                }
                constexpr int rd_off = ks < 3 ? (ks + 1) * 4 * W4_PS : 0;
                // (the operands built in k-step 3 of the last chunk belong to the next item's tile)
                const int mk = (ks == 3 && last) ? nvn : nv;
                // 60 slots: one MFMA + the non-VALU fillers dealt to its gap
                w4_static_for<2 * MT * W3_NF>([&](auto S) __attribute__((always_inline)) {
                    constexpr int s = decltype(S)::value;
                    if constexpr (s % 8 == 0) W4_KS(ks, s / 8);                   // stamps 0..7: slots 0, 8, .. 56
                    if constexpr (s == W4_XF_SLOT + 1) W4_KS(ks, 9);             // behind the transform burst
                    if constexpr (s == 59) W4_KS(ks, 10);
                    if constexpr (s >= 42 && s <= 46) W4_KS(ks, 11 + s - 42);       // slots 42..46 one by one
                    constexpr int idx = s >> 1, h = s & 1;
                    constexpr int pos = idx + (idx >= 15 ? 1 : 0);          // slot of the packed order (15 and 31 are padding)
                    constexpr int e = w3_pos_e(pos);
                    constexpr int mt = e / W3_NF, j = e - mt * W3_NF;
                    if constexpr ((pos >> 2 & 3) != 3) w4_mfma<(h * MT + mt) * W3_NF + j, zc && ks == 0>(aq[ks & 1][pos >> 2][pos & 3], v[ks & 1][j][h]);
                    else w4_mfma<(h * MT + mt) * W3_NF + j, zc && ks == 0>(aq3[ks & 1][pos >> 4][pos & 3], v[ks & 1][j][h]);
                    // raw-patch reads of the next k-step: slots 0..23
                    if constexpr (s < 24 && !(W4_ABL & 4)) {
                        constexpr int hh = s & 1, m = s >> 1, cc = m >> 1, wh = m & 1;
                        raw[wh][cc][hh] = lds_ld(rd[wh][hh], rd_off + cc);
                    }
                    // input transform of the next k-step: one packed burst
                    if constexpr (s == W4_XF_SLOT) W4_KS(ks, 8);
                    if constexpr (s == W4_XF_SLOT) xf_burst(raw[0], raw[1], v[(ks + 1) & 1], mk);
                    // raw-tile staging of the next chunk, four 16-byte groups per batch: loaded in k-step 0 / 1, written
                    // to LDS a k-step later
                    {   // request n at chunk slot 2 + GAP n, its four LDS writes from chunk slot 2 + DIST + GAP n on
                        constexpr int G = ks * 60 + s;
                        constexpr int G0 = 2, G1 = 2 + W4_STG_DIST;
                        static_assert(G1 + 7 * W4_STG_GAP + 3 < 180, "the last LDS write must precede the chunk barrier");
                        if constexpr (G >= G0 && G < G0 + W4_STG_GAP * 8 && (G - G0) % W4_STG_GAP == 0) stage_load((G - G0) / W4_STG_GAP);
                        if constexpr (G >= G1 && G < G1 + W4_STG_GAP * 8 && (G - G1) % W4_STG_GAP < 4)
                            stage_store(std::integral_constant<int, (G - G1) / W4_STG_GAP * 4 + (G - G1) % W4_STG_GAP>{});
                    }
                    // pass-0 operands of the epilogue, requested late in the item's LAST chunk (W4_PF_KS, W4_PF_SLOT: ahead of
                    // their use) behind a uniform branch (issued in every chunk through an empty descriptor, twelve such
                    // loads cost ~600 cycles per chunk)
                    if constexpr (ks == W4_PF_KS && s == W4_PF_SLOT && !(W4_ABL & 8)) {
                        if (last) ep_fetch(std::integral_constant<int, 0>{}, 0, rs_op, rs_bias);
                    }
                    // weight refills: a group is free behind the MFMAs of its last slot
                    if constexpr (!(W4_ABL & 2) && h == 1 && ((pos & 3) == 3 || pos == 14 || pos == 30)) load_a(ks & 1, pos >> 2, w_pre[ks]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            wcur = wnext;
            sstage += (size_t)16 * HW;
        };
        chunk(0, std::true_type{});
        for (int c = 1; c < nch; ++c) chunk(c, std::false_type{});

        W4_SEG(1);
        // ---- output transform + epilogue, one m-tile (16 channels) per pass.  Writer half (every wave, its frequency row
        // i): the m-tile's 48 accumulators are read FOUR CHANNELS AT A TIME (the registers of an accumulator tile), the
        // column transform A4 (6 -> 4) runs as packed fp32 on those channel vectors and each of the four results goes to LDS
        // as one 16-byte piece.  Reader half: a thread owns a 2x4 tile of two channels, adds the four waves' values (A2),
        // and finishes (bias, residual / GELU / GELU', stores).  Two exchange buffers: m-tile p + 1 is transformed and
        // written between the issue of p's LDS reads and their use -- one barrier per pass.  The accumulators are not
        // zeroed: the next item's first k-step runs with C = 0. ----
        if (!(W4_ABL & 8)) {
        // (the accumulators of m-tiles 0 / 1 were last written 36+ MFMAs ago; the nops cover the tail of the matrix pipe)
        asm volatile("s_nop 15\n\ts_nop 15");
        if constexpr (W4_PF_SLOT < 0) ep_fetch(std::integral_constant<int, 0>{}, 0, rs_op, rs_bias);
        auto colxf = [&](auto M0) __attribute__((always_inline)) {
            constexpr int m0 = decltype(M0)::value;
            w4_static_for<2>([&](auto H) __attribute__((always_inline)) {
                constexpr int h = decltype(H)::value;
                constexpr int R0 = ((h * MT + m0) * W3_NF) * 4;
                f32x4 m_[6];
                w4_static_for<6>([&](auto J) __attribute__((always_inline)) {
                    constexpr int j = decltype(J)::value;
                    m_[j] = f32x4{w4_acc_read<R0 + 4 * j>(), w4_acc_read<R0 + 4 * j + 1>(), w4_acc_read<R0 + 4 * j + 2>(),
                                  w4_acc_read<R0 + 4 * j + 3>()};
                });
                // (M A4)[i][q]:  A4^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
                // (differences as a + n1 b with an opaque -1: a plain fsub of a vector is not selected as v_pk_add_f32)
                const f32x4 s12 = m_[1] + m_[2], d12 = n4 * m_[2] + m_[1], s34 = m_[3] + m_[4], d34 = n4 * m_[4] + m_[3];
                float* d = xwrite + (m0 & 1) * W4_XB + h * 16 * 16;
                const f32x4 t0 = m_[0] + s12 + s34, t1 = 2.f * d34 + d12, t2 = 4.f * s34 + s12, t3 = 8.f * d34 + d12 + m_[5];
                if (W4_ABL & 512) { asm volatile("" ::"v"(t0), "v"(t1), "v"(t2), "v"(t3)); return; }
                *reinterpret_cast<f32x4*>(d) = t0;
                *reinterpret_cast<f32x4*>(d + 512) = t1;
                *reinterpret_cast<f32x4*>(d + 1024) = t2;
                *reinterpret_cast<f32x4*>(d + 1536) = t3;
            });
        };
        colxf(std::integral_constant<int, 0>{});
        w4_static_for<MT>([&](auto M0) __attribute__((always_inline)) {
            constexpr int m0 = decltype(M0)::value;
            constexpr int set = m0 % NS;
            W4_SEG(2 + 3 * m0);
            // operands of a LATER pass, W4_PF_AHEAD passes ahead of their use (into the set the previous pass has just used)
            if constexpr (m0 + W4_PF_AHEAD < MT)
                ep_fetch(std::integral_constant<int, (m0 + W4_PF_AHEAD) % NS>{}, m0 + W4_PF_AHEAD, rs_op, rs_bias);
            lds_barrier();                          // m-tile m0 is in its buffer; the other buffer has been read
            W4_SEG(3 + 3 * m0);
            const float* xr = xread + (m0 & 1) * W4_XB;
            f32x2 tt[4][4];                         // [i][column] x channel pair
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    tt[i][q] = (W4_ABL & 1024) ? f32x2{n1.x * (i + q), n1.y} : *reinterpret_cast<const f32x2*>(xr + (i * 4 + q) * 512);
            // (behind the reads in the LDS queue, and beside their latency: the next m-tile's writer half)
            if constexpr (m0 + 1 < MT) colxf(std::integral_constant<int, m0 + 1>{});
            W4_SEG(4 + 3 * m0);
            f32x2 y0[4], y1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // Y[pp] = sum_i A2^T[pp][i] t_i:  row 0 = t_0 + t_1 + t_2, row 1 = t_1 - t_2 - t_3
                y0[q] = tt[0][q] + tt[1][q] + tt[2][q];
                y1[q] = n1 * tt[3][q] + (n1 * tt[2][q] + tt[1][q]);
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int so = ep_soff(m0, r);
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    f32x4 w_ = pp == 0 ? f32x4{y0[0][r], y0[1][r], y0[2][r], y0[3][r]}
                                       : f32x4{y1[0][r], y1[1][r], y1[2][r], y1[3][r]};
                    if (ACT != 2) w_ += bsv[set][r];
                    if (ACT == 1) {
                        ep_store(rs_pre, pp, so, w_);          // pre-activation (the training forward saves it;
                                                               // empty descriptor when out_pre is null: dropped)
                        w_ = gelu_erf4(w_);
                    }
                    if (ACT == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) w_[e] *= gelu_erf_grad(opv[set][r][pp][e]);
                    } else {
                        w_ += opv[set][r][pp];
                    }
                    if (padded) w_ *= pm;
                    ep_store(rs_out, pp, so, w_);
                }
            }
        });
        // reader geometry of the next item (VALU here, where nothing is hidden anyway, instead of in its main loop)
        ep_geo(nx, vo_nx);
        if (padded) pm = pad_mask(nx);
        W4_SEG(20);
        }
        if (!have_next) break;
        it = nx;                                   // goff already describes nx
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) vo[pp][pc] = vo_nx[pp][pc];
        wb_it = wb_nx;
        nv = nvn;
    }
}

#ifndef SINDDM_V4_MIN_ITEMS_PER_CU   // launches with at least this many (8x32 tile, 80-channel block) items per CU take conv_wino4.h
#define SINDDM_V4_MIN_ITEMS_PER_CU 2
#endif

inline bool conv_wino4_applies(int B, int H, int W, int coblks) {
    return (long long)B * ((W + W4_TW - 1) / W4_TW) * ((H + W4_TH - 1) / W4_TH) * coblks >=
           (long long)SINDDM_V4_MIN_ITEMS_PER_CU * wino2_cu_count();
}

inline int conv_wino4_launch(const ConvArgs& a_in, hipStream_t st) {
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
#if defined(W4_TIMING) || defined(W4_KSTAMP)
    static int w4_launch_no = 0;
    a.mtp = w4_launch_no++ % 8;                  // (the kernel does not read mtp: stamp row of this launch)
#endif
    const int ipx = a.tiles_per_xcd * a.coblks;
    int wpx = wino2_cu_count() / 8;              // one workgroup per CU
    if (wpx < 1) wpx = 1;
    if (wpx > ipx) wpx = ipx;
    const unsigned grid = (unsigned)(wpx * 8);
    constexpr size_t lds = W4_LDS_FLOATS * sizeof(float);
    // (more than 64 KB of dynamic LDS needs the per-function opt-in: once per kernel and process, its result checked)
#define W4_GO(ACT, EDGE)                                                                                                \
    do {                                                                                                                \
        static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino4_kernel<ACT, EDGE>), \
                                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
        if (attr_rc != hipSuccess) return (int)attr_rc;                                                                 \
        hipLaunchKernelGGL((conv_wino4_kernel<ACT, EDGE>), dim3(grid), dim3(256), lds, st, a, ipx, wpx);                \
    } while (0)
    const int edge = a.W % 4 == 0 ? 0 : (a.W % 2 == 0 ? 1 : 2);
    switch ((a.act & 0xff) * 3 + edge) {
        case 0: W4_GO(0, 0); break;
        case 1: W4_GO(0, 1); break;
        case 2: W4_GO(0, 2); break;
        case 3: W4_GO(1, 0); break;
        case 4: W4_GO(1, 1); break;
        case 5: W4_GO(1, 2); break;
        case 6: W4_GO(2, 0); break;
        case 7: W4_GO(2, 1); break;
        default: W4_GO(2, 2);
    }
#undef W4_GO
    if (rec) {
        (void)hipEventRecord(prof.ev[2 * prof.used + 1], st);
        const double fl = 2.0 * a.B * a.H * (a.Wt > 0 ? a.Wt : a.W) * (double)a.Cout * 9.0 * a.Cin;   // algorithmic (direct-conv) FLOPs
        prof.note(1, fl, fl * (24.0 / 72.0), 4);                                      // F(2x4): 24 multiplies per 8 outputs
    }
    SINDDM_LAUNCH_CHECK();
    return 0;
}

}  // namespace sinddm
