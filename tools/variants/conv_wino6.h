// Winograd F(4x4, 3x3) 3x3 convolution on the gfx950 fp32 matrix cores: 36 multiplies per 16 outputs (2.25 per output)
// instead of the 3.0 of F(2x4) -- a quarter of the MFMAs of conv_wino4.h gone.
//
// Why it was built: the dominant kernel runs at the socket's power limit (profiles/NOTES_r04.md section 6: 1.34 kW of 1.4 kW
// at 2.31 of 2.4 GHz); schedules that need fewer cycles at the same energy per item buy a lower clock, so the hypothesis
// was that only less matrix work buys time.
//
// STATUS (round 4): parity-green (tests/test_gpu_wino6.py, SQ_INSTS_MFMA = 0.75 x conv_wino4's) and NO FASTER than
// conv_wino4 (C3 step 104.8 against 103.7 ms, profiles/r04e_f44_*.txt): the matrix pipe is not what bounds that kernel
// (profiles/NOTES_r04.md section 9).  Experimental: compiled only with -DSINDDM_WINO_F44_BUILD=1, dispatched only behind
// sinddm_debug_set_f44().  Its loads are asm the compiler does not track ("=v" outputs, AGPR destinations, LDS-DMA) with
// the kernel's own vmcnt bookkeeping: correct in this build, but nothing stops a compiler from copying such a value
// before the kernel's wait (the -DW4_TIMING / -DW4_KSTAMP builds of this file fault) -- pin the registers before shipping.
//
// Skeleton of conv_wino4.h (one persistent 4-wave workgroup per CU, a wave owns its SIMD and 512 registers, an item is
// an 8x32-pixel tile x 80 output channels, raw halo tile 16 channels x 10 rows x 40 columns double-buffered in LDS,
// accumulators in numbered AGPRs, the item's first k-step with C = 0), with the work split the 6 x 6 frequency grid
// forces: 6 does not divide over 4 waves by rows, so WAVE (a, b), a, b in {0, 1}, owns the 3 x 3 BLOCK
// i in {3a .. 3a+2}, j in {3b .. 3b+2}: nine frequency GEMMs x five m-tiles x ONE n-tile of sixteen 4x4 output tiles
// (2 x 8 tiles = the item's 8 x 32 pixels): 45 MFMAs per k-step (conv_wino4: 60), 180 accumulator registers, every A
// fragment feeds one MFMA (256 B of L2 weight traffic per MFMA, 1.5x conv_wino4's bytes per k-step).
//   * input transform per wave: rows 3a..3a+2 of B^T need patch rows a..a+4, columns 3b..3b+2 patch columns b..b+4 --
//     25 of the 36 patch values; 6 operations per column for the row triple (packed over column pairs), 6 per row for the
//     column triple: 36 VALU per k-step behind a wave-uniform branch on a / b.
//   * A ring of FOUR k-steps (stage = k-step of the chunk; two stages in the 74 AGPRs the accumulators leave free).
//   * output transform: writer half  T_ab[p][jj] = sum_{i in block} A^T[p][i] M[i][3b+jj]  (4 x 3 values per tile and
//     channel) goes to LDS channel-minor ([wave][p][jj][tile][16 channels], 16-byte pieces, 48 KB per m-tile, two
//     buffers); reader half: a thread owns rows 2 p' .. 2 p'+1 of a 4x4 tile for two channels, adds the two a-halves and
//     applies A^T along the columns -- from there on the epilogue IS conv_wino4's (same thread -> pixel map: a thread
//     finishes a 2x4 pixel block of two channels).
//   * LDS: 64 KB raw tiles + 96 KB exchange = all 160 KB.  Raw tiles arrive by 16-byte LDS-DMA (rows of eleven 16-byte
//     groups, plane stride 448 floats): the 4-byte patch reads are 4-way bank-conflicted, which the LDS pipe absorbs.
// Restrictions (the caller keeps conv_wino4 otherwise): W % 4 == 0 (EDGE 0 only), C_out % 80 == 0, C_in % 16 == 0,
// C_in >= 32, the same items-per-CU rule.  Same ConvArgs / epilogue contract.  Numerics: tools/f44_model.py (the device's
// operation order in numpy): 1.4e-6 rel-L2 per convolution at C_in = 160 (F(2x4): 6e-7).
#pragma once
#include "conv_wino4.h"

namespace sinddm {

constexpr int W6_MT = 5;
constexpr int W6_NF = 9;                        // frequencies of a wave: the 3 x 3 block, f = ii * 3 + jj
constexpr int W6_NPOS = W6_MT * W6_NF;          // 45 A fragments per (wave, k-step): pos = mt * 9 + f
constexpr int W6_Q = 12;                        // 16-byte groups per (wave, k-step): 48 slots, the last three padding
constexpr int W6_KS_BYTES = W6_Q * 1024;
constexpr int W6_CH_BYTES = 4 * 4 * W6_KS_BYTES;      // one 16-channel chunk: 4 waves x 4 k-steps = 192 KB
inline long long wino6_packed_floats(int coblks, int nch) { return (long long)coblks * nch * (W6_CH_BYTES / 4); }

constexpr int W6_RS = 44, W6_PS = 448;          // raw-tile row / plane stride (floats): rows are eleven 16-byte groups (ten + a gap),
                                                // a wave's four planes 448 groups = seven 16-byte DMA instructions.  Everything a
                                                // multiple of 4 floats: the 4-byte patch reads of a 32-lane group fall on 8 of the
                                                // 32 banks (4-way, ~200 LDS cycles per wave and k-step -- the LDS pipe has them;
                                                // the alternative, 28 dword DMA instructions per wave and chunk, cost 14 % of the step)
constexpr int W6_BUF = 8192;                    // floats per raw-tile buffer (16 planes of 448, padded to 32 KB: XOR swap)
static_assert(16 * W6_PS <= W6_BUF && 10 * W6_RS <= W6_PS && W6_PS % 4 == 0 && W6_RS % 4 == 0, "raw-tile buffer");
constexpr int W6_XB = 4 * 12 * 256;             // one exchange buffer: [wave][p][jj][16 tiles][16 channels] = 48 KB
constexpr int W6_LDS_FLOATS = 2 * W6_BUF + 2 * W6_XB;
static_assert(W6_LDS_FLOATS * 4 <= 160 * 1024, "LDS of a gfx950 CU");

template <int T, bool ZC>
__device__ __forceinline__ void w6_mfma(float a, float b) {
    static_assert(T < 45, "accumulator tile");
    if constexpr (ZC)
        asm volatile("v_mfma_f32_16x16x4_f32 a[%c2:%c3], %0, %1, 0" ::"v"(a), "v"(b), "n"(4 * T), "n"(4 * T + 3));
    else
        asm volatile("v_mfma_f32_16x16x4_f32 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "n"(4 * T), "n"(4 * T + 3));
}
// ... the same with the A fragment in AGPR a<AR> (weight-ring stages 2 and 3, see the kernel)
template <int T, bool ZC, int AR>
__device__ __forceinline__ void w6_mfma_a(float b) {
    static_assert(T < 45 && AR >= 180 && AR < 254, "register map");
    if constexpr (ZC)
        asm volatile("v_mfma_f32_16x16x4_f32 a[%c1:%c2], a%c3, %0, 0" ::"v"(b), "n"(4 * T), "n"(4 * T + 3), "n"(AR));
    else
        asm volatile("v_mfma_f32_16x16x4_f32 a[%c1:%c2], a%c3, %0, a[%c1:%c2]" ::"v"(b), "n"(4 * T), "n"(4 * T + 3), "n"(AR));
}
// 16 / 4 bytes per lane straight into AGPRs a<AR> .. (the compiler neither sees the destination nor counts the load: the
// kernel waits for these itself, see w6_wait)
template <int AR>
__device__ __forceinline__ void w6_load_a4(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    static_assert(AR % 2 == 0, "AGPR tuples start at an even register");
    asm volatile("buffer_load_dwordx4 a[%c3:%c4], %0, %1, %2 offen" ::"v"(voff), "s"(rs), "s"(soff), "n"(AR), "n"(AR + 3) : "memory");
}
template <int AR>
__device__ __forceinline__ void w6_load_a1(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    asm volatile("buffer_load_dword a%c3, %0, %1, %2 offen" ::"v"(voff), "s"(rs), "s"(soff), "n"(AR) : "memory");
}
// 64 dwords global -> LDS (lane l: global byte offset voff + soff of the descriptor -> LDS byte address ldsaddr + 4 l; an
// out-of-range lane writes zero).  asm, not the builtin: the compiler fences every later LDS read it cannot prove disjoint
// with a vmcnt wait for the DMA; this kernel waits once, in front of the chunk barrier (w6_wait).
template <int LOFF>
__device__ __forceinline__ void w6_dma(unsigned ldsbase, __amdgpu_buffer_rsrc_t rs, unsigned voff) {
    // (one wait state between the SALU write of m0 and the DMA that reads it; the scalar offset is the literal 0: the form
    // with an `s_add_i32 m0, base, literal` and the offset in a register assembled, and faulted)
    const unsigned la = ldsbase + LOFF;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" ::"v"(voff), "s"(la), "s"(rs) : "memory");
}
// ... and into VGPRs the compiler picks ("=v") but does not TRACK: with some loads counted by the compiler and some not, its
// vmcnt(N) in front of every first use of a refilled group was N too small by the hidden requests since -- it waited for
// raw-tile requests issued two k-steps AFTER the group it needed, and the four-stage ring bought nothing.  Every vector-memory
// request of the main loop is asm now and the kernel keeps the count itself (w6_wait).
__device__ __forceinline__ f32x4 w6_load_v4(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    f32x4 r;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r) : "v"(voff), "s"(rs), "s"(soff) : "memory");
    return r;
}
__device__ __forceinline__ float w6_load_v1(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    float r;
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(r) : "v"(voff), "s"(rs), "s"(soff) : "memory");
    return r;
}
template <int N>
__device__ __forceinline__ void w6_wait() {
    static_assert(N >= 0 && N <= 63, "vmcnt");
    asm volatile("s_waitcnt vmcnt(%c0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void w6_acc_declare() { asm volatile("" ::: "a0", "a253"); }
constexpr int W6_A2 = 180;                      // ring stage 2: fragments 0..44 in a180 .. a224
constexpr int W6_A3 = 226;                      // ring stage 3: fragments 0..27 (groups 0..6) in a226 .. a253 (tuples start even), the rest in VGPRs
constexpr int W6_G3 = 7;                        // first group of stage 3 that lives in VGPRs
constexpr int W6_G1 = 6;                        // first group of stage 1 whose refill an item's last chunk leaves to the end of the epilogue

// timing ablations (-DW6_ABL=bits; results are WRONG, never ship): 1 no raw-tile staging  2 no weight refills  4 no raw-patch reads  16 no input transform  64 raw-tile requests to one L2-resident kilobyte
#ifndef W6_ABL
#define W6_ABL 0
#endif
#ifndef W4_SEG_ITEM
#define W4_SEG_ITEM 3             // (-DW4_TIMING builds: the item of every workgroup whose segments are stamped)
#endif
#ifndef W6_XF_SLOT
#define W6_XF_SLOT 30          // the slot behind whose MFMA the input-transform burst of the next k-step sits
#endif
#ifndef W6_STG_GAP
#define W6_STG_GAP 10
#endif
#ifndef W6_STG_DIST
#define W6_STG_DIST 48
#endif   // raw-tile requests of the next chunk: one every GAP slots, written DIST slots later

template <int ACT>
__global__ __launch_bounds__(256, 1) void conv_wino6_kernel(ConvArgs p, int items_per_xcd, int wg_per_xcd) {
    constexpr int MT = W6_MT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sX = smem + 2 * W6_BUF;

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
    const int wi = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = frequency block (a, b)
    const bool wa = (wi >> 1) != 0, wb = (wi & 1) != 0;
    const int l16 = lane & 15, kq = lane >> 4;
    const int H = p.H, W = p.W;
    const int HW = H * W;
    // lane -> 4x4 output tile (tile row tr, tile column tc), channel kq of the k-step: patch rows = halo rows 4 tr + (0..5),
    // patch columns = halo columns 4 tc + 3 .. 4 tc + 8; this wave reads rows a .. a+4, columns b .. b+4
    const int tr_ = l16 >> 3, tc_ = l16 & 7;
    typedef __attribute__((address_space(3))) float lds_f;
    const unsigned lds0 = (unsigned)(size_t)(lds_f*)smem;
    unsigned rdr[5];                                // LDS byte address of (patch row r, first column) in plane kq of k-step 0
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const int R = 4 * tr_ + (wa ? 1 : 0) + r;
        rdr[r] = lds0 + 4u * (kq * W6_PS + R * W6_RS + 4 * tc_ + 3 + (wb ? 1 : 0));
    }
    auto lds_ld = [](unsigned addr, int foff) __attribute__((always_inline)) { return ((const lds_f*)addr)[foff]; };
    const int nch = p.nch3;

    // ---- raw tile staging by 16-byte LDS-DMA: a wave's four channel planes (10 halo rows x 11 groups of 4 columns, the 11th a
    // gap) are a LINEAR run of 448 sixteen-byte groups = seven instructions; lane l of instruction i owns group 64 i + l =
    // (plane, row, column group) or a gap (out of range: zeros).  No staging registers, no LDS write instructions, requests a
    // whole chunk ahead (4 behind the barrier that frees the buffer, 3 in the next k-step 0).  (conv_wino4 stages through
    // registers: with the request -> LDS write distance the chunk leaves, every chunk of THIS kernel waited for HBM; dword
    // DMA -- which could keep an odd, conflict-free row stride -- is bound by its instruction count, 28 per wave and chunk.) ----
    constexpr unsigned OOB = 0x40000000u;
    unsigned goffd[7];                              // global byte offset (plane included) of this lane's group, per DMA instruction
    const unsigned HW4g = (unsigned)HW * 4u;
    auto make_goffd = [&](const Wino4Item& g) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int P = 64 * i + lane;            // group index inside the wave's 448
            const int pl = P / 112, q = P - pl * 112;
            const int row = q / 11, grp = q - row * 11;
            const int gy = g.y0 + row - 1, gx = g.x0 - 4 + 4 * grp;
            const bool ok = grp < 10 && row < 10 && gy >= 0 && gy < H && gx >= 0 && gx < W;
            goffd[i] = ok ? (unsigned)pl * HW4g + (unsigned)(gy * W + gx) * 4u : OOB;
        }
    };
    const unsigned HW4 = HW4g;
    auto plane_ptr = [&](int ib) { return p.in + ((size_t)ib * p.Cin + wi * 4) * HW; };
    unsigned rbuf = 0;                              // byte offset of the raw-tile buffer being READ (flips at the chunk barrier)
    const unsigned dbase = lds0 + 4u * W6_BUF + 4u * (wi * 4 * W6_PS);   // this wave's planes in the buffer NOT being read: dbase ^ rbuf
    // DMA instruction n = 0..6 of a chunk: groups 64 n .. 64 n + 63 of this wave's region
    auto dma_n = [&](auto N, const __amdgpu_buffer_rsrc_t& rs) __attribute__((always_inline)) {
        constexpr int n = decltype(N)::value;
        if (W6_ABL & 1) return;
        w6_dma<1024 * n>(dbase ^ rbuf, rs, goffd[n]);
    };
    // descriptor of chunk t of image b (empty beyond the last channel / when there is no such item)
    // (never an EMPTY descriptor: an LDS-DMA through num_records = 0 is not range-checked at all -- its out-of-range lanes,
    // 1 GB past the base, fault (measured, profiles/NOTES_r04.md).  A workgroup's last item requests its own tile again.)
    auto chunk_rsrc = [&](const float* base, int t, bool) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + (size_t)t * 16 * HW), 0, 4 * (int)HW4, 0x00020000);
    };

    // ---- weights: [co-blk][chunk][wave][k-step][group 0..11][lane][4], slot 4 q + s = pos = mt * 9 + f ----
    const __amdgpu_buffer_rsrc_t rsw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w3), 0, p.coblks * p.nch3 * W6_CH_BYTES, 0x00020000);
    const int wlane = lane * 16;
    auto wbase = [&](int cb) -> int { return cb * nch * W6_CH_BYTES + wi * (4 * W6_KS_BYTES); };
    // ---- A ring of FOUR k-steps: stage = k-step of the chunk, refilled behind its last use with the same k-step of the NEXT
    // chunk -- 180 MFMAs (~5 800 cycles) ahead.  Vector memory returns in order: a refill queued behind a raw-tile request
    // that goes to HBM cannot land before it, and with a ring of two k-steps (2 900 cycles) that cost 10 % of the step
    // (profiles/r04e_w6_ablations.txt).  Stages 0, 1: VGPRs; stage 2 and groups 0..6 of stage 3: the 74 AGPRs the
    // accumulators leave free, loaded by asm; the rest of stage 3: VGPRs. ----
    f32x4 aq[2][11];                                // stages 0, 1: groups 0..10 (44 fragments)
    float aq1[2];                                   // ... and fragment 44 (group 11 holds one fragment + padding)
    f32x4 aq3v[4];                                  // stage 3: groups 7..10
    float aq3s;                                     // stage 3: fragment 44
    auto load_a = [&](auto ST, auto Q, int soff) __attribute__((always_inline)) {
        constexpr int st = decltype(ST)::value, q = decltype(Q)::value;
        if constexpr (st < 2) {
            if constexpr (q < 11)
                aq[st][q] = w6_load_v4(rsw, wlane + q * 1024, soff);
            else
                aq1[st] = w6_load_v1(rsw, wlane + 11 * 1024, soff);
        } else if constexpr (st == 2) {
            if constexpr (q < 11) w6_load_a4<W6_A2 + 4 * q>(rsw, wlane + q * 1024, soff);
            else w6_load_a1<W6_A2 + 44>(rsw, wlane + 11 * 1024, soff);
        } else {
            if constexpr (q < W6_G3) w6_load_a4<W6_A3 + 4 * q>(rsw, wlane + q * 1024, soff);
            else if constexpr (q < 11)
                aq3v[q - W6_G3] = w6_load_v4(rsw, wlane + q * 1024, soff);
            else
                aq3s = w6_load_v1(rsw, wlane + 11 * 1024, soff);
        }
    };
    // ---- the kernel's own vmcnt bookkeeping (vector memory retires in order; n = requests per k-step: 12 refills, + 3 DMA in
    // k-step 0 (slots 0, 10, 20), + 4 DMA in k-step 3; an item's LAST chunk leaves 6 refills of k-step 1 and 5 of k-step 3 to
    // the end of the epilogue, in this order, and adds the epilogue's pass-0 operands).  Stage ks was refilled during k-step
    // ks of the chunk before; everything issued in the three k-steps since is younger:
    //     k-step 0: n1 + n2 + n3 = 40     k-step 1: n2 + n3 + n0 = 43     k-step 2: n3 + n0 + n1 = 43 (LAST: 37)
    //     k-step 3: n0 + n1 + n2 = 39 (LAST: 33)            [behind a LAST chunk + epilogue: more everywhere]
    // first chunk of an item: the late refills of stage 1 (groups 6..11, first used at slot 24 of k-step 1) are followed by 5
    // late refills of stage 3, the 15 requests of k-step 0 and the 6 refills of slots 3..23: 26; the late refills of stage 3
    // by the 39 of k-steps 0..2.  DMA -> chunk barrier: the last request (slot 20 of k-step 0) is followed by 7 + 12 + 12
    // refills (LAST: 7 + 6 + 12).
    auto chunk_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    static_assert(!(W6_ABL & 2), "the vmcnt bookkeeping counts the weight refills");
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // ---- input transform: 25 patch values -> the nine B operands of the block ----
    // three rows of B^T (F(4,3): [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]) on
    // the five values d0..d4 the triple needs (the upper triple counts from patch index 1): six operations
    auto bt3 = [](bool hi, auto d0, auto d1, auto d2, auto d3, auto d4, auto& o0, auto& o1, auto& o2) __attribute__((always_inline)) {
        if (!hi) {
            const auto s = d4 - 4.f * d2, t = d3 - 4.f * d1;
            o0 = 4.f * d0 + (d4 - 5.f * d2);
            o1 = s + t;
            o2 = s - t;
        } else {
            const auto u = d3 - d1, w = d2 - d0;
            o0 = u + 2.f * w;
            o1 = u - 2.f * w;
            o2 = 4.f * d0 + (d4 - 5.f * d2);
        }
    };
    float raw[5][5];
    float v[2][W6_NF];                              // [k-step parity][f = ii * 3 + jj]
    auto xf_burst = [&](float (&vo_)[W6_NF]) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 5; ++r) asm volatile("" : "+v"(raw[r][0]), "+v"(raw[r][1]), "+v"(raw[r][2]), "+v"(raw[r][3]), "+v"(raw[r][4]));
        if (W6_ABL & 16) {
#pragma unroll
            for (int f = 0; f < W6_NF; ++f) vo_[f] = raw[f / 3][f % 3];
            return;
        }
        // rows: packed over column pairs (0,1), (2,3), (4,4)
        f32x2 P[5][3], O[3][3];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            P[r][0] = f32x2{raw[r][0], raw[r][1]};
            P[r][1] = f32x2{raw[r][2], raw[r][3]};
            P[r][2] = f32x2{raw[r][4], raw[r][4]};
        }
        if (!wa) {
#pragma unroll
            for (int k = 0; k < 3; ++k) bt3(false, P[0][k], P[1][k], P[2][k], P[3][k], P[4][k], O[0][k], O[1][k], O[2][k]);
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) bt3(true, P[0][k], P[1][k], P[2][k], P[3][k], P[4][k], O[0][k], O[1][k], O[2][k]);
        }
        if (!wb) {
#pragma unroll
            for (int ii = 0; ii < 3; ++ii)
                bt3(false, O[ii][0].x, O[ii][0].y, O[ii][1].x, O[ii][1].y, O[ii][2].x, vo_[ii * 3 + 0], vo_[ii * 3 + 1], vo_[ii * 3 + 2]);
        } else {
#pragma unroll
            for (int ii = 0; ii < 3; ++ii)
                bt3(true, O[ii][0].x, O[ii][0].y, O[ii][1].x, O[ii][1].y, O[ii][2].x, vo_[ii * 3 + 0], vo_[ii * 3 + 1], vo_[ii * 3 + 2]);
        }
#pragma unroll
        for (int f = 0; f < W6_NF; ++f) asm volatile("" : "+v"(vo_[f]));
    };

    Wino4Item it;
    int l = 0;
    if (!decode(l, it)) return;
    w6_acc_declare();
    make_goffd(it);
    int wb_it = wbase(it.cb);
    w4_static_for<4>([&](auto ST) __attribute__((always_inline)) {
        w4_static_for<W6_Q>([&](auto Q) __attribute__((always_inline)) { load_a(ST, Q, wb_it + decltype(ST)::value * W6_KS_BYTES); });
    });
    // first chunk of the first item into buffer 0 (rbuf = the other one meanwhile), the first four requests of its chunk 1
    {
        rbuf = 4u * W6_BUF;
        const __amdgpu_buffer_rsrc_t rs0 = chunk_rsrc(plane_ptr(it.b), 0, true);
        w4_static_for<7>([&](auto N) __attribute__((always_inline)) { dma_n(N, rs0); });
        rbuf = 0;
        const __amdgpu_buffer_rsrc_t rs1 = chunk_rsrc(plane_ptr(it.b), 1, true);
        w4_static_for<4>([&](auto N) __attribute__((always_inline)) { dma_n(N, rs1); });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // (once per workgroup: the asm loads among them)
    __syncthreads();
    int wcur = wb_it;
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int c = 0; c < 5; ++c) raw[r][c] = lds_ld(rdr[r], c);
    xf_burst(v[0]);

    // ---- epilogue reader role: thread = rows 2 p' .. 2 p' + 1 of a 4x4 tile (tile row trr, tile column tcr) x TWO consecutive
    // channels; in pixels: the 2x4 block at (y0 + 4 trr + 2 p', x0 + 4 tcr) -- conv_wino4's reader with the roles of its
    // (n-tile, tile row) pair exchanged ----
    const int hf = tid & 1, kqr = (tid >> 1) & 3;
    const int tile16 = (tid >> 3) & 15;
    const int phalf = tid >> 7;                     // wave-uniform
    const int trr = tile16 >> 3, tcr = tile16 & 7;
    const int cg = kqr * 4 + hf * 2;
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    const bool nt = __builtin_amdgcn_readfirstlane((size_t)p.B * p.Cout * HW > ((size_t)1 << 28)) != 0;
    auto ep_geo = [&](const Wino4Item& g, unsigned (&vo)[2]) __attribute__((always_inline)) {
        const int y = g.y0 + 4 * trr + 2 * phalf, x = g.x0 + 4 * tcr;
        const unsigned base = ((unsigned)cg * (unsigned)HW + (unsigned)(y * W + x)) * 4u;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const bool ok = (y + pp < H) & (x < W);
            vo[pp] = ok ? base + (unsigned)(pp * W) * 4u : OOB;
        }
    };
    unsigned vo[2];                                 // (rebuilt per item where it is first needed: nothing of the epilogue is carried
                                                    // through the main loop, whose register budget the four-stage ring has used up)
    const bool padded = __builtin_amdgcn_readfirstlane(p.Wt > 0 && p.Wt < W) != 0;
    auto pad_mask = [&](const Wino4Item& g) __attribute__((always_inline)) -> f32x4 {
        const int nvq = p.Wt - (g.x0 + 4 * tcr);
        return f32x4{nvq > 0 ? 1.f : 0.f, nvq > 1 ? 1.f : 0.f, nvq > 2 ? 1.f : 0.f, nvq > 3 ? 1.f : 0.f};
    };
    auto ep_load = [&](const __amdgpu_buffer_rsrc_t& r, int pp, int soff) __attribute__((always_inline)) -> f32x4 {
        if (nt) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)vo[pp], soff, 2));
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)vo[pp], soff, 0));
    };
    auto ep_store = [&](const __amdgpu_buffer_rsrc_t& r, int pp, int soff, f32x4 vv) __attribute__((always_inline)) {
        const u32x4 u = __builtin_bit_cast(u32x4, vv);
        const unsigned voff = vo[pp];
        // (store + one wait state as ONE asm: see conv_wino4.h)
        if (nt) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 1" ::"v"(u), "v"(voff), "s"(r), "s"(soff) : "memory");
        else asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(u), "v"(voff), "s"(r), "s"(soff) : "memory");
    };
    f32x4 opv[2][2][2];                             // [set][channel][output row]
    f32x2 bsv[2];
    f32x2 n1{-1.f, -1.f};                           // opaque to the optimiser: differences as a + n1 b stay packed
    asm volatile("" : "+v"(n1));
    const f32x4 n4{n1.x, n1.y, n1.x, n1.y};
    float* xwrite = sX + (wi * 12 * 16 + l16) * 16 + kq * 4;
    const float* xread = sX + phalf * (2 * 3 * 256) + tile16 * 16 + kqr * 4 + hf * 2;

    for (;;) {
#ifdef W4_TIMING
        const bool seg = l == W4_SEG_ITEM && blockIdx.x < 256;
#endif
        W4_SEG(0);
        Wino4Item nx;
        l += 1;
        const bool have_next = decode(l, nx);
        if (!have_next) nx = it;
        const int wb_nx = wbase(nx.cb);
        const float* base_nx = plane_ptr(nx.b);
        const float* base_it = plane_ptr(it.b);
        const unsigned plane_b = HW4;
        const unsigned samp_b = (unsigned)p.Cout * plane_b;
        const size_t samp_o = (size_t)it.b * p.Cout * HW;
        auto rsrc_of = [&](const float* base) {
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base ? base + samp_o : p.zero), 0,
                                                     base ? samp_b : 0u, 0x00020000);
        };
        auto mk_op = [&]() { return rsrc_of(ACT == 2 ? p.aux : p.resid); };
        auto mk_bias = [&]() {
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias ? p.bias : p.zero), 0,
                                                     p.bias ? (unsigned)(p.coblks * MT * 16) * 4u : 0u, 0x00020000);
        };
        const int cb_ch = it.cb * (MT * 16);
        auto ep_soff = [&](int m0, int r) -> int { return (cb_ch + m0 * 16 + r) * (int)plane_b; };
        auto ep_fetch = [&](auto SET, int m0, const __amdgpu_buffer_rsrc_t& rs_op, const __amdgpu_buffer_rsrc_t& rs_bias) __attribute__((always_inline)) {
            constexpr int set = decltype(SET)::value;
            if (ACT != 2)
                bsv[set] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_bias, cg * 4, (cb_ch + m0 * 16) * 4, 0));
            if constexpr (ACT != 1) {               // (the GELU launches -- conv1 of a block -- carry no residual: launch check)
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) opv[set][r][pp] = ep_load(rs_op, pp, ep_soff(m0, r));
            }
        };
        // LAST: the item's last chunk is its own copy of the body (compile-time: what it prefetches for the epilogue would
        // otherwise be carried around the chunk loop in registers the loop does not have)
        auto chunk = [&](int c, auto ZC, auto LAST) __attribute__((always_inline)) {
            constexpr bool zc = decltype(ZC)::value;
            constexpr bool last = decltype(LAST)::value;
#ifdef W4_KSTAMP
            const bool kst = l == 4 && c == 2 && blockIdx.x < 256;
#endif
            const int wnext = last ? wb_nx : wcur + W6_CH_BYTES;
            // (stage ks is refilled with k-step ks of the next chunk / item)
            // raw tiles: k-step 0 issues the last 13 requests of chunk c + 1 (behind the last chunk: the next item's chunk 0),
            // k-step 3 -- behind the barrier that frees the buffer -- the first 15 of chunk c + 2
            __amdgpu_buffer_rsrc_t rsd;              // descriptor of the chunk whose requests the current k-step issues
            {
                int cc = c;
                asm volatile("" : "+s"(cc));        // (built HERE, not hoisted to the top of the item and spilled)
                rsd = last ? chunk_rsrc(base_nx, 0, have_next) : chunk_rsrc(base_it, cc + 1, true);
            }
            w4_static_for<4>([&](auto KS) __attribute__((always_inline)) {
                constexpr int ks = decltype(KS)::value;
                if constexpr (ks == 3) {
                    __builtin_amdgcn_sched_barrier(0);
                    w6_wait<last ? 25 : 31>();
                    chunk_barrier();
#pragma unroll
                    for (int r = 0; r < 5; ++r) rdr[r] ^= 4u * W6_BUF;
                    rbuf ^= 4u * W6_BUF;
                    asm volatile("" : "+v"(rdr[0]), "+v"(rdr[1]), "+v"(rdr[2]), "+v"(rdr[3]), "+v"(rdr[4]));
                    // (from the second-to-last chunk on the requests are the next item's)
                    if (c + 2 == nch) make_goffd(nx);
                    {
                        int cc = c;
                        asm volatile("" : "+s"(cc));
                        const bool a_nx = cc + 2 >= nch;
                        rsd = a_nx ? chunk_rsrc(base_nx, cc + 2 - nch, have_next) : chunk_rsrc(base_it, cc + 2, true);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                constexpr int rd_off = ks < 3 ? (ks + 1) * 4 * W6_PS : 0;
                w4_static_for<W6_NPOS>([&](auto S) __attribute__((always_inline)) {
                    constexpr int pos = decltype(S)::value;
                    constexpr int f = pos % W6_NF;
                    if constexpr (pos % 5 == 0) W4_KS(ks, pos / 5);                // stamps 0..8: slots 0, 5, .. 40
                    if constexpr (pos == 44) W4_KS(ks, 9);
                    if constexpr (pos == W6_XF_SLOT + 1) W4_KS(ks, 11);             // behind the transform burst
                    if constexpr (pos == 0) w6_wait<ks == 0 ? 40 : ks == 1 ? 43 : ks == 2 ? (last ? 37 : 43) : (last ? 33 : 39)>();
                    if constexpr (zc && ks == 1 && pos == 4 * W6_G1) w6_wait<26>();
                    if constexpr (ks < 2) {
                        if constexpr (pos < 44) w6_mfma<pos, zc && ks == 0>(aq[ks][pos >> 2][pos & 3], v[ks & 1][f]);
                        else w6_mfma<pos, zc && ks == 0>(aq1[ks], v[ks & 1][f]);
                    } else if constexpr (ks == 2) {
                        w6_mfma_a<pos, false, W6_A2 + pos>(v[ks & 1][f]);
                    } else {
                        if constexpr (pos < 4 * W6_G3) w6_mfma_a<pos, false, W6_A3 + pos>(v[ks & 1][f]);
                        else if constexpr (pos < 44) w6_mfma<pos, false>(aq3v[(pos >> 2) - W6_G3][pos & 3], v[ks & 1][f]);
                        else w6_mfma<pos, false>(aq3s, v[ks & 1][f]);
                    }
                    // raw-patch reads of the next k-step: slots 0..24
                    if constexpr (pos < 25 && !(W6_ABL & 4)) raw[pos / 5][pos % 5] = lds_ld(rdr[pos / 5], rd_off + pos % 5);
                    if constexpr (pos == W6_XF_SLOT) W4_KS(ks, 10);
                    if constexpr (pos == W6_XF_SLOT) xf_burst(v[(ks + 1) & 1]);
                    // raw-tile requests: one every ten slots
                    if constexpr (ks == 0 && pos % 10 == 0 && pos <= 20) dma_n(std::integral_constant<int, 4 + pos / 10>{}, rsd);
                    if constexpr (ks == 3 && pos % 10 == 0 && pos <= 30) dma_n(std::integral_constant<int, pos / 10>{}, rsd);
                    if constexpr (ks == 3 && pos == 43) {
                        if constexpr (last) {
                            ep_geo(it, vo);
                            ep_fetch(std::integral_constant<int, 0>{}, 0, mk_op(), mk_bias());
                        }
                    }
                    // weight refills: a group is free behind the MFMA of its last slot
                    if constexpr (!(W6_ABL & 2) && ((pos & 3) == 3 || pos == 44)) {
                        // (the VGPR part of stage 3 is not needed before the next item's k-step 3, the second half of stage 1 not
                        // before the middle of its k-step 1: in an item's last chunk their refills wait until the epilogue -- which
                        // is where the register pressure peaks -- is over)
                        if constexpr ((ks == 3 && (pos >> 2) >= W6_G3) || (ks == 1 && (pos >> 2) >= W6_G1)) {
                            if constexpr (!last) load_a(KS, std::integral_constant<int, (pos >> 2)>{}, wnext + ks * W6_KS_BYTES);
                        } else {
                            load_a(KS, std::integral_constant<int, (pos >> 2)>{}, wnext + ks * W6_KS_BYTES);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            wcur = wnext;
        };
        chunk(0, std::true_type{}, std::false_type{});
        for (int c = 1; c + 1 < nch; ++c) chunk(c, std::false_type{}, std::false_type{});
        chunk(nch - 1, std::false_type{}, std::true_type{});

        W4_SEG(1);
        // ---- output transform + epilogue, one m-tile (16 channels) per pass ----
        asm volatile("s_nop 15\n\ts_nop 15");
        // (the epilogue's descriptors are built here: carried through the main loop they were scalar spills)
        const __amdgpu_buffer_rsrc_t rs_out = rsrc_of(p.out);
        const __amdgpu_buffer_rsrc_t rs_op = mk_op();
        const __amdgpu_buffer_rsrc_t rs_pre = rsrc_of(ACT == 1 ? p.out_pre : nullptr);
        const __amdgpu_buffer_rsrc_t rs_bias = mk_bias();
        auto rowxf = [&](auto M0) __attribute__((always_inline)) {
            constexpr int m0 = decltype(M0)::value;
            float* d = xwrite + (m0 & 1) * W6_XB;
            w4_static_for<3>([&](auto JJ) __attribute__((always_inline)) {
                constexpr int jj = decltype(JJ)::value;
                f32x4 m_[3];
                w4_static_for<3>([&](auto II) __attribute__((always_inline)) {
                    constexpr int ii = decltype(II)::value;
                    constexpr int R0 = (m0 * W6_NF + ii * 3 + jj) * 4;
                    m_[ii] = f32x4{w4_acc_read<R0>(), w4_acc_read<R0 + 1>(), w4_acc_read<R0 + 2>(), w4_acc_read<R0 + 3>()};
                });
                // T[p] = sum_i A^T[p][i] M[i]:  A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
                f32x4 t0, t1, t2, t3;
                if (!wa) {
                    const f32x4 s = m_[1] + m_[2], dd = n4 * m_[2] + m_[1];
                    t0 = m_[0] + s; t1 = dd; t2 = s; t3 = dd;
                } else {
                    const f32x4 s = m_[0] + m_[1], dd = n4 * m_[1] + m_[0];
                    t0 = s; t1 = 2.f * dd; t2 = 4.f * s; t3 = 8.f * dd + m_[2];
                }
                *reinterpret_cast<f32x4*>(d + (0 * 3 + jj) * 256) = t0;
                *reinterpret_cast<f32x4*>(d + (1 * 3 + jj) * 256) = t1;
                *reinterpret_cast<f32x4*>(d + (2 * 3 + jj) * 256) = t2;
                *reinterpret_cast<f32x4*>(d + (3 * 3 + jj) * 256) = t3;
            });
        };
        rowxf(std::integral_constant<int, 0>{});
        f32x4 pm = {1.f, 1.f, 1.f, 1.f};
        if (padded) pm = pad_mask(it);
        w4_static_for<MT>([&](auto M0) __attribute__((always_inline)) {
            constexpr int m0 = decltype(M0)::value;
            constexpr int set = m0 & 1;
            W4_SEG(2 + 3 * m0);
            if constexpr (m0 + 1 < MT) ep_fetch(std::integral_constant<int, (m0 + 1) & 1>{}, m0 + 1, rs_op, rs_bias);
            lds_barrier();                          // m-tile m0 is in its buffer; the other buffer has been read
            W4_SEG(3 + 3 * m0);
            const float* xr = xread + (m0 & 1) * W6_XB;
            // (one output row at a time: twelve 8-byte reads -- the two a-halves of T[p][0..5] -- are live at once, not 24)
            f32x2 y[2][4];                          // [output row][column] x channel pair
            f32x2 xa0[2][6];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    xa0[a][j] = *reinterpret_cast<const f32x2*>(xr + ((a * 2 + j / 3) * 12 + 0 * 3 + j % 3) * 256);
            if constexpr (m0 + 1 < MT) rowxf(std::integral_constant<int, m0 + 1>{});
            auto cols = [&](const f32x2 (&x)[2][6], f32x2 (&yo)[4]) __attribute__((always_inline)) {
                f32x2 t[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) t[j] = x[0][j] + x[1][j];
                const f32x2 s12 = t[1] + t[2], d12 = n1 * t[2] + t[1], s34 = t[3] + t[4], d34 = n1 * t[4] + t[3];
                yo[0] = t[0] + s12 + s34;
                yo[1] = 2.f * d34 + d12;
                yo[2] = 4.f * s34 + s12;
                yo[3] = 8.f * d34 + d12 + t[5];
            };
            W4_SEG(4 + 3 * m0);
            cols(xa0, y[0]);
            {
                f32x2 xa1[2][6];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int j = 0; j < 6; ++j)
                        xa1[a][j] = *reinterpret_cast<const f32x2*>(xr + ((a * 2 + j / 3) * 12 + 1 * 3 + j % 3) * 256);
                cols(xa1, y[1]);
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int so = ep_soff(m0, r);
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    f32x4 w_ = f32x4{y[pp][0][r], y[pp][1][r], y[pp][2][r], y[pp][3][r]};
                    if (ACT != 2) w_ += bsv[set][r];
                    if (ACT == 1) {
                        ep_store(rs_pre, pp, so, w_);
                        w_ = gelu_erf4(w_);
                    }
                    if (ACT == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) w_[e] *= gelu_erf_grad(opv[set][r][pp][e]);
                    } else if (ACT == 0) {
                        w_ += opv[set][r][pp];
                    }
                    if (padded) w_ *= pm;
                    ep_store(rs_out, pp, so, w_);
                }
            }
        });
        w4_static_for<W6_Q - W6_G1>([&](auto Q) __attribute__((always_inline)) {
            load_a(std::integral_constant<int, 1>{}, std::integral_constant<int, W6_G1 + decltype(Q)::value>{}, wb_nx + 1 * W6_KS_BYTES);
        });
        w4_static_for<W6_Q - W6_G3>([&](auto Q) __attribute__((always_inline)) {
            load_a(std::integral_constant<int, 3>{}, std::integral_constant<int, W6_G3 + decltype(Q)::value>{}, wb_nx + 3 * W6_KS_BYTES);
        });
        W4_SEG(20);
        if (!have_next) break;
        it = nx;
        wb_it = wb_nx;
    }
}

// Run-time switch (process-global, sinddm_debug_set_f44): 1 = launches that qualify take this kernel
#ifndef SINDDM_WINO_F44_DEFAULT
#define SINDDM_WINO_F44_DEFAULT 0
#endif
inline int& conv_wino6_flag() {
    static int on = SINDDM_WINO_F44_DEFAULT != 0;
    return on;
}
inline bool conv_wino6_enabled() { return conv_wino6_flag() != 0; }

// (callers: a launch with act == 1 AND a residual operand is not this kernel's -- none exists in the network)
inline bool conv_wino6_applies(int B, int H, int W, int coblks, int Cin) {
    return conv_wino6_enabled() && W % 4 == 0 && Cin >= 32 && Cin % 16 == 0 && conv_wino4_applies(B, H, W, coblks);
}

inline int conv_wino6_launch(const ConvArgs& a_in, hipStream_t st) {
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
    a.mtp = W6_MT;
#if defined(W4_TIMING) || defined(W4_KSTAMP)
    static int w6_launch_no = 0;
    a.mtp = w6_launch_no++ % 8;                  // (the kernel does not read mtp: stamp row of this launch)
#endif
    const int ipx = a.tiles_per_xcd * a.coblks;
    int wpx = wino2_cu_count() / 8;
    if (wpx < 1) wpx = 1;
    if (wpx > ipx) wpx = ipx;
    const unsigned grid = (unsigned)(wpx * 8);
    constexpr size_t lds = W6_LDS_FLOATS * sizeof(float);
#define W6_GO(ACT)                                                                                                     \
    do {                                                                                                               \
        static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino6_kernel<ACT>), \
                                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
        if (attr_rc != hipSuccess) return (int)attr_rc;                                                                \
        hipLaunchKernelGGL((conv_wino6_kernel<ACT>), dim3(grid), dim3(256), lds, st, a, ipx, wpx);                     \
    } while (0)
    switch (a.act & 0xff) {
        case 0: W6_GO(0); break;
        case 1: W6_GO(1); break;
        default: W6_GO(2);
    }
#undef W6_GO
    if (rec) {
        (void)hipEventRecord(prof.ev[2 * prof.used + 1], st);
        const double fl = 2.0 * a.B * a.H * (a.Wt > 0 ? a.Wt : a.W) * (double)a.Cout * 9.0 * a.Cin;   // algorithmic (direct-conv) FLOPs
        prof.note(1, fl, fl * (36.0 / 144.0), 6);                                     // F(4x4): 36 multiplies per 16 outputs
    }
    SINDDM_LAUNCH_CHECK();
    return 0;
}

}  // namespace sinddm
