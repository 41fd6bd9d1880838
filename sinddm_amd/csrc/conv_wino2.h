// Winograd F(2x2, 3x3) 3x3 convolution on the gfx950 fp32 matrix cores -- second generation ("W4x2").
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 2x2 output tile, d = its 4x4 input patch
//
// 16 "frequency" GEMMs  M_ij[co][tile] = sum_ci U_ij[co][ci] * V_ij[ci][tile]  replace the 9-tap implicit GEMM
// (16 multiplies per 4 outputs instead of 36), all in fp32 on v_mfma_f32_16x16x4_f32.
//
// What changed against conv_wino.h (one 16-wave workgroup per CU, wave = one frequency), and why:
//   * TWO independent 4-wave workgroups per CU (256 registers per wave).  The first-generation kernel kept all 16
//     waves of a CU in the same phase, so the matrix pipe idled through every output-transform epilogue (15 % of the
//     launch), every chunk-barrier ramp (7 %) and every item prologue.  Two workgroups drift against each other:
//     while one is in its epilogue / at a barrier / waiting for its first operands, the other one's wave on the same
//     SIMD owns the matrix pipe.
//   * wave i owns the FOUR frequencies (i, 0..3) of a 4x32-pixel tile (2x16 output 2x2-tiles) for MT*16 output
//     channels: 4 x MT x 2 accumulator tiles (160 registers at MT = 5).  The row half of the input transform
//     (B^T d) is shared by its four frequencies: per k-step and tile-row TWO 16-byte LDS reads + 8 VALU build four B
//     operands (the old kernel: 16 four-byte reads + 12 FMAs), and the column half of the OUTPUT transform
//     (M A) happens in registers, so only 8 instead of 16 values per (channel, tile) cross waves through LDS.
//   * weights (U, pre-transformed by the pack kernel) are laid out [co-block][16-ch chunk][i][k-step][mt][lane][j]:
//     ONE 16-byte buffer load per (k-step, mt) brings a lane its four A operands; every load is issued right after
//     the last MFMA that reads the registers it overwrites, i.e. two k-steps (80 MFMAs) ahead of its use.
//   * raw input halo tile (16 channels x 6 x 40, plane padded to 256 floats), double buffered, one 4-wave barrier per
//     16-channel chunk.  Big launches (MT >= 3) bring it through registers: one 16-byte buffer load per channel with
//     per-lane global offsets (hardware bounds check zero-fills halo / missing channels) and a ds_write_b128 a k-step
//     later; the small-launch kernels (MT < 3) use LDS-DMA (no registers to spare for latency there).
//   * the instruction mix is the design constraint (tools/ubench/valu_next_to_mfma.hip): while the other wave of the
//     SIMD streams MFMAs, a wave issues VALU every 2-3 cycles but SALU / LDS / VMEM / s_waitcnt only once per 16
//     cycles.  Address arithmetic of both streams is therefore incremental (one descriptor per chunk, one scalar
//     offset per k-step: 17 SALU per 160 MFMAs), validity tests are VALU selects on offsets (out-of-range offset =
//     hardware drop / zero fill) instead of exec-mask branches.
//   * burst schedule: all non-MFMA work of a k-step (stage stores, raw-patch reads, stage loads, input transform) sits
//     at its head, then the 8 * MT MFMAs run back to back with only the A refills between the m-tiles (pinned with
//     sched_barrier): the two waves of a SIMD alternate bursts and heads.  Anything else inside a burst costs 1-6 %.
// Same ConvArgs / epilogue contract as conv_mfma.h (bias, GELU / GELU'(aux) *, identity residual, pre-activation
// save); 1x1 residual projections are not fused (the caller passes their result as `resid`).
#pragma once
#include "conv_mfma.h"
#include "conv1x1.h"

namespace sinddm {

// Compile-time timing ablations (-DW2_ABL=bits; results are WRONG, never ship):
//   1 no raw-tile DMA   2 weights loaded once   4 no LDS reads   8 no epilogue   16 no input-transform VALU
//   32 raw patches read from the exchange area (LDS the DMA never writes)   64 no chunk barrier   128 chunk barrier
//   without its waitcnt
#ifndef W2_ABL
#define W2_ABL 0
#endif
#ifndef W2_THEAD
#define W2_THEAD 0           // (experiment) transform at the head of the k-step that uses it (no LDS wait in a head): +-0
#endif
#ifndef W2_ADB
#define W2_ADB 0             // (experiment) double-buffered A registers: all weight loads at the head of the k-step
#endif
#ifndef W2_WMERGE
#define W2_WMERGE 0          // (experiment) 2 + 1 merged waits per k-step instead of 5 + 3: measured -1.4 % (waits come earlier)
#endif
#ifndef W2_STAGE
#define W2_STAGE 1           // raw tiles of the big launches through registers (0: LDS-DMA, as the small launches do)
#endif
#ifdef W2_TIMING
// s_memtime stamps of workgroups 8 and 9 (debug builds only; tools/w2_timing.py): [wg][item][slot][wave]
__device__ unsigned long long g_w2_dbg[2 * 4 * 40 * 4];
#endif
#ifdef W2_PHASE
// per workgroup HW_ID | XCC_ID << 32, then (epilogue start, end) of its first 9 items (tools/w2_phase.py)
__device__ unsigned long long g_w2_phase[1024 * 20];
// per workgroup and wave: 16 stamps inside the epilogue of item 3 (tools/w2_phase.py seg)
__device__ unsigned long long g_w2_seg[1024 * 4 * 16];
#define W2_SEG(slot) do { if (seg) g_w2_seg[(blockIdx.x * 4 + wi) * 16 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W2_SEG(slot) do {} while (0)
#endif
constexpr int W2_THREADS = 256;
constexpr int W2_TW = 32, W2_TH = 4;           // pixel tile
// LDS image of a channel's halo tile: 6 rows (y0-1 .. y0+4) of 40 floats = image columns x0-4 .. x0+35, i.e. ten
// 16-byte groups per row that are ALIGNED to the tile (x0 is a multiple of 32): a group is either entirely left of
// the image (x0 = 0: hardware zero fill through the per-lane out-of-range offset) or starts inside it.  60 groups =
// ONE 16-bytes-per-lane LDS-DMA instruction per channel.
constexpr int W2_RS = 40, W2_HR = W2_TH + 2;
constexpr int W2_GRP = W2_RS / 4;              // 16-byte groups per row
constexpr int W2_PLANE = W2_HR * W2_RS;        // 240 floats
constexpr int W2_PS = 256;                     // plane stride (lanes 60..63 of the DMA write zeros into the pad)
constexpr int W2_BUF = 16 * W2_PS;             // floats per raw-tile buffer (16-channel chunk)
constexpr int W2_XCH = 4 * 32 * 32 * 2;        // exchange area: [i][32 channels][32 tiles][q]
constexpr int W2_LDS_FLOATS = 2 * W2_BUF + W2_XCH;

struct Wino2Item {          // one unit of work: a 4x32 pixel tile x one block of MT*16 output channels
    int b, y0, x0, cb;
};

// EDGE >= 1 when W % 4 != 0 (2 when W is odd: the last column is then stored with 4-byte stores): a 16-byte group of the halo tile can then straddle the right image edge, and the columns
// past it (the next row's first pixels) are zeroed when the patch is transformed -- 8 v_cndmask per k-step that images
// with W % 4 == 0 (every group is entirely inside or entirely outside a row: hardware zero fill) do not pay.
template <int MT, int ACT, int MTP, int EDGE>
__global__ __launch_bounds__(W2_THREADS, 2) void conv_wino2_kernel(ConvArgs p, int items_per_xcd, int wg_per_xcd) {
    constexpr int NT = 2;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sX = smem + 2 * W2_BUF;

    // persistent workgroups: XCD `xcd` owns a contiguous range of tiles (its L2 serves their halos); workgroup `ls` of
    // that XCD takes tiles ls, ls + wg_per_xcd, ... and runs ALL output-channel blocks of a tile back to back: the raw
    // tile of the second block comes out of L2 (a few hundred cycles instead of an HBM round trip).  That matters
    // more than it looks: memory operations complete in order, so every weight load a wave issues after a raw-tile
    // DMA is held up until that DMA has landed.
    const int xcd = blockIdx.x & 7;
    const int ls = blockIdx.x >> 3;
    const int tpi = p.tilesX * p.tilesY;
    // (launches with fewer tiles than workgroups -- the coarse pyramid scales -- spread the channel blocks over
    // workgroups instead: items ls, ls + wg_per_xcd, ... of the (tile, block) list)
    const bool by_tile = p.tiles_per_xcd >= wg_per_xcd;
    auto decode = [&](int k, Wino2Item& it) -> bool {          // k-th work item of this workgroup
        int tl, cb;
        if (by_tile) {
            tl = ls + (k / p.coblks) * wg_per_xcd;              // tile inside the XCD's range
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
    const int wi = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = Winograd frequency row i
    const int l16 = lane & 15, kq = lane >> 4;
    const int H = p.H, W = p.W;
    const int HW = H * W;

    // B^T rows: frequency row i combines patch rows  0: d0 - d2   1: d1 + d2   2: d1 - d2 (*)   3: d1 - d3
    // (*) B^T row 2 is -d1 + d2; the sign lives in the packed weights (U_2j is stored negated), so every wave
    //     evaluates  ra + sgn * rb  with sgn = +1 for i = 1 and -1 otherwise.
    const int pa0 = wi == 0 ? 0 : 1, pa1 = wi == 3 ? 3 : 2;
    const float sgn = wi == 1 ? 1.f : -1.f;
    // patch columns of tile column tc: image x0+2tc-1 .. x0+2tc+2 = row positions 2tc+3 .. 2tc+6
    const int oa = kq * W2_PS + pa0 * W2_RS + 2 * l16 + 3;        // + nt * 2 * RS + ks * 4 * PS
    const int ob = kq * W2_PS + pa1 * W2_RS + 2 * l16 + 3;

    const int nch = p.nch3;                                        // 16-channel chunks of the reduction
    const int nks_total = nch * 4;

    // ---- raw-tile DMA: wave w stages channels 4w..4w+3 of a chunk, one 16-bytes-per-lane instruction per channel ----
    constexpr unsigned OOB = 0x40000000u;
    unsigned goff;                                  // this lane's 16-byte group of the halo tile (per item)
    bool cm[4];                                     // patch column c of this lane's tile column lies inside the image
    bool cmn[4];                                    // ... for the item whose raw tile is being staged (the next one)
    auto make_goff = [&](const Wino2Item& it) {
        const int row = lane / W2_GRP, grp = lane - row * W2_GRP;
        const int gy = it.y0 + row - 1, gx = it.x0 - 4 + 4 * grp;
        const bool ok = lane < W2_HR * W2_GRP && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goff = ok ? (unsigned)(gy * W + gx) * 4u : OOB;
        // a group that starts inside the image may run past its right edge (into the next row): those columns are
        // replaced by the zero padding when the patch is read
#pragma unroll
        for (int c = 0; c < 4; ++c) cmn[c] = it.x0 + 2 * l16 - 1 + c < W;
    };
    // `valid` = false issues the same instructions with empty descriptors (zero fill): the instruction stream of a
    // chunk is the same for every chunk, so the compiler's vmcnt bookkeeping is exact (no control-flow merge)
    auto issue1 = [&](int ib, int c, bool valid, float* buf, int g) {     // channel g (0..3) of this wave; ib = sample
        const int kc = wi * 4 + g;
        const int ch = c * 16 + kc;
        const bool live = valid && ch < p.Cin;
        const float* sbase = p.in + ((size_t)ib * p.Cin + (live ? ch : 0)) * HW;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sbase), 0, live ? HW * 4 : 0, 0x00020000);
        float* pl = buf + kc * W2_PS;
        if (W2_ABL & 1) return;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)pl, 16, (int)goff, 0, 0, 0);
    };
    auto issue = [&](int ib, int c, bool valid, float* buf) {
#pragma unroll
        for (int g = 0; g < 4; ++g) issue1(ib, c, valid, buf, g);
    };
    // The same transfer through registers (big launches): the CU's LDS-DMA path moves ~10 B/clk, so the 16 KB raw
    // tile of a chunk keeps the vector-memory pipe of the CU busy for ~60 % of a chunk when both workgroups stream --
    // every other memory instruction (weights, epilogue loads and stores) queues behind it.  An ordinary 16-byte load
    // + ds_write_b128 a k-step later moves the same bytes at the pipe's full rate; two staging slots (8 VGPRs).
    f32x4 stg[2];
    // Scalar instructions are the expensive ones here: next to a wave that streams MFMAs, a SALU / LDS / VMEM / s_waitcnt
    // instruction issues once per 16 cycles (VALU: every 2-3; tools/ubench/valu_next_to_mfma.hip), so the address
    // arithmetic of the streams is incremental -- one descriptor per chunk (the wave's 4 channel planes of the chunk,
    // the plane picked by the instruction's scalar offset), one scalar byte offset per k-step for the weights.
    const unsigned HW4 = (unsigned)HW * 4u;
    auto plane_ptr = [&](int ib) { return p.in + ((size_t)ib * p.Cin + wi * 4) * HW; };   // (sample, channel 4 wi)
    __amdgpu_buffer_rsrc_t rs_st;                   // planes of the chunk being staged (Cin % 4 == 0: all four or none)
    auto stage_load = [&](int slot, int g) {
        if (W2_ABL & 1) return;
        stg[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_st, (int)goff, g * (int)HW4, 0));
    };
    auto stage_store = [&](int slot, float* buf, int g) {
        if (W2_ABL & 1) return;
        *reinterpret_cast<f32x4*>(buf + (wi * 4 + g) * W2_PS + lane * 4) = stg[slot];
    };

    // ---- weights: register image [coblk][chunk][i][ks][mt][lane][j], 16-byte buffer loads ----
    const __amdgpu_buffer_rsrc_t rsw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w3), 0, 0x7FFFFFF0, 0x00020000);
    const int wlane = lane * 16;
    // byte offset of (item block cb, chunk 0, this wave's frequency row, k-step 0, m-tile 0) in the packed image -- the
    // one integer division of the weight addressing, once per item instead of once per load
    auto wbase = [&](int cb) -> int {
        const int mg = cb * MT;                                    // first global m-tile of the item (MT divides MTP)
        const int pcb = mg / MTP, pmt = mg - pcb * MTP;
        return (pcb * nch * 4 + wi) * (4096 * MTP) + pmt * 1024;
    };
    f32x4 a[2][MT];                                                // A operands of the current k-step ([1]: W2_ADB double buffer)
    // g = k-step index inside the item; g == nks_total means "k-step 0 of the next item" (output-channel block ncb):
    // the weight stream, like the raw-tile stream, runs across items without a branch inside a k-step -- a chunk stays
    // one basic block, which the register allocator needs to accumulate in place
    auto load_w = [&](int mt, int cb, int ncb, int g) {
        if ((W2_ABL & 2) && g > 0) return;
        const bool wrap = g >= nks_total;
        const int gg = wrap ? 0 : g;
        // the packed image is blocked by MTP m-tiles per output-channel block; an item covers MT of them (MT == MTP
        // for the big launches, MT = 1 when a launch has too few tiles to fill the chip with whole blocks)
        // (`cb` / `ncb` arrive already split into packed block and first packed m-tile: wbase())
        const int wb = wrap ? ncb : cb;
        const int so = wb + (gg >> 2) * (4 * 4096 * MTP) + (gg & 3) * (1024 * MTP) + mt * 1024;
        a[0][mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wlane, so, 0));
    };
    // Barriers are written in assembly: hipcc puts a full `s_waitcnt vmcnt(0)` in front of every s_barrier on gfx9,
    // which would drain the weight loads issued a moment ago (main loop) and expose the completion latency of every
    // global store (epilogue).
    //   chunk barrier: what it needs is (a) this wave's raw-tile DMA of the next chunk landed -- it is OLDER than the MT
    //   weight loads of the k-step before the barrier and memory operations complete in order, so vmcnt(MT) covers it;
    //   (b) this wave's LDS reads of the current chunk done: lgkmcnt(0).
    auto chunk_barrier = [&]() {
        if (W2_ABL & 64) return;
        if (W2_ABL & 128) { asm volatile("s_barrier" ::: "memory"); return; }
        if (W2_STAGE && MT >= 3) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); return; }
        if (W2_ADB && MT >= 3) { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(3 * MT) : "memory"); return; }
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(MT) : "memory");
    };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // B operands of one k-step: V[nt][j] from the two patch rows of this wave's frequency row
    auto read_raw = [&](const float* base, f32x4 (&ra)[NT], f32x4 (&rb)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (W2_ABL & 4) { ra[nt] = f32x4{1.f, 2.f, 3.f, 4.f}; rb[nt] = f32x4{(float)lane, 1.f, 0.f, 2.f}; continue; }
            const float* qa = base + oa + nt * 2 * W2_RS;
            const float* qb = base + ob + nt * 2 * W2_RS;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                ra[nt][c] = qa[c];
                rb[nt][c] = qb[c];
            }
        }
    };
    auto transform = [&](const f32x4 (&ra)[NT], const f32x4 (&rb)[NT], float (&v)[NT][4], const bool (&cm)[4]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (W2_ABL & 4) { v[nt][0] = ra[nt][0]; v[nt][1] = rb[nt][0]; v[nt][2] = ra[nt][2]; v[nt][3] = rb[nt][1]; continue; }
            // columns right of the image edge hold the next row's pixels (the 16-byte groups run past it): zero padding
            float r0 = fmaf(sgn, rb[nt][0], ra[nt][0]), r1 = fmaf(sgn, rb[nt][1], ra[nt][1]);
            float r2 = fmaf(sgn, rb[nt][2], ra[nt][2]), r3 = fmaf(sgn, rb[nt][3], ra[nt][3]);
            if (EDGE) { r0 = cm[0] ? r0 : 0.f; r1 = cm[1] ? r1 : 0.f; r2 = cm[2] ? r2 : 0.f; r3 = cm[3] ? r3 : 0.f; }
            v[nt][0] = r0 - r2;      // B columns: 0: c0 - c2   1: c1 + c2   2: c2 - c1   3: c1 - c3
            v[nt][1] = r1 + r2;
            v[nt][2] = r2 - r1;
            v[nt][3] = r1 - r3;
        }
    };

    Wino2Item it;
    int l = 0;
    if (!decode(l, it)) return;
    make_goff(it);
#pragma unroll
    for (int c = 0; c < 4; ++c) cm[c] = cmn[c];
    issue(it.b, 0, true, smem);
    int wb_it = wbase(it.cb);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) load_w(mt, wb_it, wb_it, 0);
    dma_barrier();
    int wcur = wb_it;                              // weights of (item, chunk, this wave's frequency row, k-step 0, m-tile 0)
    const float* sstage = plane_ptr(it.b) + (size_t)16 * HW;   // planes of the chunk to stage next (chunk 1 of the item)
    int nb = 0;                                    // raw buffer holding the current chunk (runs across items)
    float v[2][NT][4];                             // B operands, double buffered by k-step parity (runs across items)
    // W2_THEAD (big launches): the raw patches of a k-step are read at the head of the PREVIOUS k-step and transformed at
    // the head of their own -- no LDS round trip in a head, one B-operand set; they stay in registers across the epilogue
    f32x4 pra[NT], prb[NT];
    if (W2_THEAD && MT >= 3) {
        read_raw(smem, pra, prb);
    } else {
        f32x4 ra[NT], rb[NT];
        read_raw(smem, ra, rb);
        transform(ra, rb, v[0], cm);
    }
#ifdef W2_TIMING
    int dbg_item = 0;
    const int dbg_wg = blockIdx.x == 8 ? 0 : 1;
#endif

    for (;;) {
        Wino2Item nx;
        l += 1;
        const bool have_next = decode(l, nx);
        if (!have_next) nx = it;
        const int wb_nx = wbase(nx.cb);
        const float* base_nx = plane_ptr(nx.b);
#ifdef W2_TIMING
        const bool dbg = (blockIdx.x == 8 || blockIdx.x == 8 + 8 * 32) && dbg_item < 4 && lane == 0;
        if (dbg) g_w2_dbg[((dbg_wg * 4 + dbg_item) * 40 + 39) * 4 + wi] = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
#endif
        f32x4 acc[MT][NT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[mt][nt][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // Schedule of a 16-channel chunk (4 k-steps of 8*MT MFMAs):
        //   k-step 0   : LDS-DMA of the NEXT chunk of this workgroup's stream (chunk c+1, or chunk 0 of the next item)
        //                into the other raw buffer
        //   k-step ks  : per m-tile 8 MFMAs, then the refill of that m-tile's A registers with k-step ks+1; the raw
        //                patches of k-step ks+1 are read from LDS beside the MFMAs of the first m-tile and transformed
        //                into B operands beside those of the third
        //   barrier    : BEFORE the last k-step, not after it: every read of the current raw buffer has been issued by
        //                then (k-step 3's operands are in registers), the next chunk's DMA landed long ago, and each wave
        //                leaves the barrier with 8*MT MFMAs ready whose shadow hides the first LDS round trip of the next
        //                chunk.  (A barrier at the end of the chunk exposes that round trip: 10 % of a chunk.)
        for (int c = 0; c < nch; ++c) {
            const float* cur = smem + nb * W2_BUF;
            float* nxt = smem + (nb ^ 1) * W2_BUF;
            // (opaque: the optimizer must not peel or unswitch the chunk loop on it -- two copies of the body do not fit
            // the register file)
            const bool last = __builtin_amdgcn_readfirstlane(c + 1 == nch) != 0;
#ifdef W2_TIMING
            if (dbg && c < 12) g_w2_dbg[((dbg_wg * 4 + dbg_item) * 40 + 3 * c + 0) * 4 + wi] = __builtin_amdgcn_s_memtime();
#endif
            if (last) make_goff(nx);
            const int dsrc = last ? nx.b : it.b;
            const int dch = last ? 0 : c + 1;
            const bool dval = !last || have_next;
            if (MT < 3) issue(dsrc, dch, dval, nxt);
            // byte offsets of the weights of k-steps 1..3 of this chunk and of k-step 0 of the next one (next chunk, or
            // the next item's first)
            constexpr int W_KS = 1024 * MTP, W_CH = 4 * 4096 * MTP;
            const int w_k[4] = {wcur + W_KS, wcur + 2 * W_KS, wcur + 3 * W_KS, last ? wb_nx : wcur + W_CH};
            if constexpr (MT >= 3 && W2_STAGE) {
                if (last) sstage = base_nx;
                const bool live = dval && dch * 16 + wi * 4 < p.Cin;
                rs_st = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sstage), 0, live ? 4 * (int)HW4 : 0, 0x00020000);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int g = c * 4 + ks;
                if (ks == 3) {
                    __builtin_amdgcn_sched_barrier(0);
#ifdef W2_TIMING
                    if (dbg && c < 12) g_w2_dbg[((dbg_wg * 4 + dbg_item) * 40 + 3 * c + 1) * 4 + wi] = __builtin_amdgcn_s_memtime();
#endif
                    chunk_barrier();
#ifdef W2_TIMING
                    if (dbg && c < 12) g_w2_dbg[((dbg_wg * 4 + dbg_item) * 40 + 3 * c + 2) * 4 + wi] = __builtin_amdgcn_s_memtime();
#endif
                }
                const float* rsrc_ = ks < 3 ? cur + (ks + 1) * 4 * W2_PS : nxt;
                f32x4 ra[NT], rb[NT];
                // the B operands built in k-step 3 belong to the NEXT chunk: in the last chunk of an item that is the next
                // item's tile, whose right-edge column masks differ
                bool mk[4];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) mk[cc] = (ks == 3 && last) ? cmn[cc] : cm[cc];
                if constexpr (MT >= 3) {
                    // Everything that is not an MFMA sits at the head of the k-step, then 8 * MT MFMAs run back to back (only
                    // the A refills between the m-tiles): the partner wave's burst covers this wave's head.  (A fine-grained
                    // version -- one LDS read / three VALU pinned behind every MFMA of m-tiles 0 and 2 -- measured 1 % slower
                    // at two waves per SIMD and no faster for a lone wave.)
#if W2_STAGE
                    if (ks == 1) { stage_store(0, nxt, 0); stage_store(1, nxt, 1); }
                    if (ks == 2) { stage_store(0, nxt, 2); stage_store(1, nxt, 3); }
#else
                    if (ks == 0) issue(dsrc, dch, dval, nxt);
#endif
#if W2_THEAD
                    transform(pra, prb, v[0], cm);                   // this k-step's operands (raw data read a k-step ago)
                    __builtin_amdgcn_sched_barrier(0);
                    read_raw(rsrc_, pra, prb);                       // the next k-step's raw patches
#else
                    read_raw(rsrc_, ra, rb);
#endif
#if W2_STAGE
                    if (ks == 0) { stage_load(0, 0); stage_load(1, 1); }
                    if (ks == 1) { stage_load(0, 2); stage_load(1, 3); }
#endif
#if W2_ADB
                    // all five A refills of the NEXT k-step here, into the other register set: the burst below is pure MFMA
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        a[(ks + 1) & 1][mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wlane + mt * 1024, w_k[ks], 0));
#endif
#if !W2_THEAD
                    transform(ra, rb, v[(ks + 1) & 1], mk);
#endif
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            acc[mt][q >> 2][q & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[W2_ADB ? (ks & 1) : 0][mt][q & 3], v[W2_THEAD ? 0 : (ks & 1)][q >> 2][q & 3],
                                                                                          acc[mt][q >> 2][q & 3], 0, 0, 0);
#if !W2_ADB
                        if (!((W2_ABL & 2) && g + 1 > 0))
                            a[0][mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wlane + mt * 1024, w_k[ks], 0));
                        __builtin_amdgcn_sched_barrier(0);
#endif
                    }
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    read_raw(rsrc_, ra, rb);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        if (mt == MT - 1) transform(ra, rb, v[(ks + 1) & 1], mk);
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            acc[mt][q >> 2][q & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][mt][q & 3], v[ks & 1][q >> 2][q & 3],
                                                                                          acc[mt][q >> 2][q & 3], 0, 0, 0);
                        load_w(mt, wb_it, wb_nx, g + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            nb ^= 1;
            wcur = w_k[3];
            sstage += (size_t)16 * HW;
        }

        // ---- output transform + epilogue: column half in registers, row half through LDS, 32 channels per pass ----
#ifdef W2_TIMING
        if (dbg) g_w2_dbg[((dbg_wg * 4 + dbg_item) * 40 + 36) * 4 + wi] = __builtin_amdgcn_s_memtime();
#endif
#ifdef W2_PHASE
        const bool ph = wi == 0 && lane == 0 && p.Cin == 160 && p.Cout == 160 && (l - 1) < 9 && blockIdx.x < 1024;
        if (ph) {
            if (l == 1) g_w2_phase[blockIdx.x * 20] = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4) |
                                                        ((unsigned long long)__builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 20) << 32);
            g_w2_phase[blockIdx.x * 20 + 2 * (l - 1) + 1] = __builtin_amdgcn_s_memtime();
        }
#endif
        if (W2_ABL & 8) {
            float sink = 0.f;                                   // keep every accumulator chain alive
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) sink += acc[mt][nt][j][0] + acc[mt][nt][j][1] + acc[mt][nt][j][2] + acc[mt][nt][j][3];
            if (sink == 123.456f) p.out[tid] = sink;
            if (!have_next) break;
            it = nx;
            wb_it = wb_nx;
#pragma unroll
            for (int c = 0; c < 4; ++c) cm[c] = cmn[c];
            continue;
        }
#ifdef W2_PHASE
        const bool seg = lane == 0 && p.Cin == 160 && p.Cout == 160 && l == 4 && blockIdx.x < 1024;
#endif
        W2_SEG(0);
        const int tile = tid & 31;                     // reader role: 2x2 tile (tile-row, tile-col) ...
        const int tr = tile >> 4, tc = tile & 15;
        const int cg = tid >> 5;                       // ... and channels cg + 8k of a pass
        const int y = it.y0 + 2 * tr, x = it.x0 + 2 * tc;
        const bool x1ok = x + 1 < W;
        // Every global access of the epilogue is a BUFFER access on a per-sample descriptor (one sample's Cout planes:
        // < 4 GB even at 411x512): lanes without a pixel / channel carry an out-of-range offset, which the hardware
        // drops (stores) or zero-fills (loads) -- no exec-masked branches, so all loads of a pass are in flight together.
        using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
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
        // padded rows (ConvArgs::Wt): the pair's columns beyond the true width are written as zeros
        const bool padded = p.Wt > 0 && p.Wt < W;
        const f32x2 pm{x < p.Wt ? 1.f : 0.f, x + 1 < p.Wt ? 1.f : 0.f};
        const int cl_lim = p.Cout - it.cb * (MT * 16) - cg;        // channel (m0, k) of this lane exists iff 16 m0 + 8 k < cl_lim
        const unsigned pix_o = ((unsigned)(it.cb * (MT * 16) + cg) * (unsigned)HW + (unsigned)(y * W + x)) * 4u;
        // channel k of pass m0 exists in the item's block for every lane, or for none (cg < 8): known at compile time
        auto in_block = [](int m0, int k) { return m0 * 16 + 8 * k + 8 <= MT * 16; };
        // byte offset of (channel k of pass m0, row pp) or OOB
        auto off_of = [&](int m0, int k, int pp) -> unsigned {
            const bool ok = okpp[pp] & (m0 * 16 + 8 * k < cl_lim);
            unsigned o = ok ? pix_o + (unsigned)(m0 * 16 + 8 * k) * plane_b + (unsigned)(pp * W) * 4u : OOB;
            // opaque to the optimizer: it would otherwise turn "access at (ok ? o : OOB)" back into an exec-masked branch
            // around the access, and the control-flow merges bring full vmcnt(0) drains (= waits for every store)
            asm volatile("" : "+v"(o));
            return o;
        };
        auto ld2 = [&](const __amdgpu_buffer_rsrc_t& r, unsigned o) -> f32x2 {
            return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)o, 0, 0));
        };
        // an 8-byte store covers pixels (x, x+1); in the last odd column only pixel x exists: 4-byte store instead
        // (only when W is odd -- EDGE = 2 builds; issued unconditionally there, with an out-of-range offset where unused)
        auto st2 = [&](const __amdgpu_buffer_rsrc_t& r, unsigned o, f32x2 vv) {
            if (W2_ABL & 512) { if (vv[0] == 123.4f) p.out[tid] = vv[1]; return; }
            if constexpr (EDGE == 2) {
                unsigned o2 = x1ok ? o : OOB, o1 = x1ok ? OOB : o;
                asm volatile("" : "+v"(o2), "+v"(o1));
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, vv), r, (int)o2, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vv[0]), r, (int)o1, 0, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, vv), r, (int)o, 0, 0);
            }
        };
        // residual / pre-activation / bias operands of a pass: requested one pass AHEAD, before the stores of the pass in
        // between -- memory operations complete in order, so a wait for them never waits for a store
        f32x2 rs_v[4][2], ax_v[4][2];
        float bs_v[4];
        auto prefetch = [&](int m0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!in_block(m0, k)) continue;
                if (W2_ABL & 256) { bs_v[k] = 0.f; rs_v[k][0] = rs_v[k][1] = f32x2{0.f, 0.f}; ax_v[k][0] = ax_v[k][1] = f32x2{1.f, 1.f}; continue; }
                bs_v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rs_bias, (it.cb * (MT * 16) + m0 * 16 + cg + 8 * k) * 4, 0, 0));
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const unsigned o = off_of(m0, k, pp);
                    rs_v[k][pp] = ld2(rs_res, o);
                    if (ACT == 2) ax_v[k][pp] = ld2(rs_aux, o);
                }
            }
        };
#pragma unroll
        for (int m0 = 0; m0 < MT; m0 += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (m0 + h < MT) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            // (M A)[i][q]:  q = 0: m0 + m1 + m2,  q = 1: m1 - m2 - m3
                            const float m_0 = acc[m0 + h][nt][0][r], m_1 = acc[m0 + h][nt][1][r];
                            const float m_2 = acc[m0 + h][nt][2][r], m_3 = acc[m0 + h][nt][3][r];
                            f32x2 t{m_0 + m_1 + m_2, m_1 - m_2 - m_3};
                            *reinterpret_cast<f32x2*>(sX + (((wi * 32 + h * 16 + kq * 4 + r) * 32) + nt * 16 + l16) * 2) = t;
                        }
                }
            }
            if (m0 == 0) prefetch(0);                              // (its accumulators are dead: registers are free)
            W2_SEG(1 + (m0 / 2) * 5);
            lds_barrier();
            W2_SEG(2 + (m0 / 2) * 5);
            f32x2 yv[4][2];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!in_block(m0, k)) continue;
                const int cl = cg + 8 * k;                         // channel of the pass (0..31)
                f32x2 t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) t[i] = *reinterpret_cast<const f32x2*>(sX + ((i * 32 + cl) * 32 + tile) * 2);
                yv[k][0] = t[0] + t[1] + t[2];                     // Y[pp][q] = sum_i A^T[pp][i] t[i][q]
                yv[k][1] = t[1] - t[2] - t[3];
            }
            W2_SEG(3 + (m0 / 2) * 5);
            lds_barrier();                                         // the exchange area is free for the next pass
            W2_SEG(4 + (m0 / 2) * 5);
            f32x2 val[4][2];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!in_block(m0, k)) continue;
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    f32x2 w_ = yv[k][pp] + bs_v[k];
                    if (ACT == 1) {
                        yv[k][pp] = w_;                            // pre-activation (training forward saves it)
                        w_ = gelu_erf2(w_);
                    } else if (ACT == 2) {
                        w_[0] *= gelu_erf_grad(ax_v[k][pp][0]);
                        w_[1] *= gelu_erf_grad(ax_v[k][pp][1]);
                    }
                    val[k][pp] = w_ + rs_v[k][pp];
                }
            }
            if (m0 + 2 < MT) prefetch(m0 + 2);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!in_block(m0, k)) continue;
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const unsigned o = off_of(m0, k, pp);
                    if (ACT == 1) st2(rs_pre, o, yv[k][pp]);       // (empty descriptor when out_pre is null: dropped)
                    st2(rs_out, o, padded ? val[k][pp] * pm : val[k][pp]);
                }
            }
            W2_SEG(5 + (m0 / 2) * 5);
        }
#ifdef W2_TIMING
        if (dbg) g_w2_dbg[((dbg_wg * 4 + dbg_item) * 40 + 37) * 4 + wi] = __builtin_amdgcn_s_memtime();
        ++dbg_item;
#endif
#ifdef W2_PHASE
        if (ph) g_w2_phase[blockIdx.x * 20 + 2 * (l - 1) + 2] = __builtin_amdgcn_s_memtime();
#endif
        if (!have_next) break;
        it = nx;                                   // goff already describes nx
        wb_it = wb_nx;
#pragma unroll
        for (int c = 0; c < 4; ++c) cm[c] = cmn[c];
    }
}

template <int MT, int MTP, int EDGE>
inline void conv_wino2_launch_e(const ConvArgs& a, unsigned grid, int ipx, int wpx, hipStream_t st) {
#ifdef W2_ONE_WG
    constexpr size_t lds = 100 * 1024;                       // (experiment) only one workgroup fits a CU
#else
    constexpr size_t lds = W2_LDS_FLOATS * sizeof(float);
#endif
    switch (a.act & 0xff) {
        case 1: hipLaunchKernelGGL((conv_wino2_kernel<MT, 1, MTP, EDGE>), dim3(grid), dim3(W2_THREADS), lds, st, a, ipx, wpx); break;
        case 2: hipLaunchKernelGGL((conv_wino2_kernel<MT, 2, MTP, EDGE>), dim3(grid), dim3(W2_THREADS), lds, st, a, ipx, wpx); break;
        default: hipLaunchKernelGGL((conv_wino2_kernel<MT, 0, MTP, EDGE>), dim3(grid), dim3(W2_THREADS), lds, st, a, ipx, wpx);
    }
}

template <int MT, int MTP>
inline void conv_wino2_launch_t(const ConvArgs& a, unsigned grid, int ipx, int wpx, hipStream_t st) {
    if (a.W % 4 == 0) conv_wino2_launch_e<MT, MTP, 0>(a, grid, ipx, wpx, st);
    else if (a.W % 2 == 0) conv_wino2_launch_e<MT, MTP, 1>(a, grid, ipx, wpx, st);     // column masks only
    else conv_wino2_launch_e<MT, MTP, 2>(a, grid, ipx, wpx, st);                         // + 4-byte stores of the last odd column
}

// CU count of the CURRENT device (kernel selection thresholds and persistent grid sizes).  Cached per device id: the
// only mutable state of the library besides the debug profiler -- idempotent, and a race on it writes the same value.
inline int wino2_cu_count() {
    static int cache[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int v = cache[dev];
    if (v <= 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
        cache[dev] = v;
    }
    return v;
}

inline int conv_wino2_launch(const ConvArgs& a_in, int mt, hipStream_t st) {
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
    // Small launches (coarse pyramid scales at small batch): with whole 80-channel blocks there are fewer work items
    // than CUs and every item walks the full reduction alone -- one m-tile per item gives 5x the items at a fifth of
    // the latency each (the packed weight image is addressed by global m-tile, so it serves both).
    a.mtp = mt;
    if (mt > 1 && a.ntiles * a.coblks < wino2_cu_count()) {
        a.coblks *= mt;
        mt = 1;
    }
    // persistent launch: two 4-wave workgroups per CU, each walking its share of the XCD's work items
    const int ipx = a.tiles_per_xcd * a.coblks;                  // work items per XCD
#ifdef W2_ONE_WG
    int wpx = wino2_cu_count() / 8;
#else
    int wpx = wino2_cu_count() / 8 * 2;                          // workgroups per XCD
#endif
    if (wpx < 1) wpx = 1;
    if (wpx > ipx) wpx = ipx;
    const unsigned grid = (unsigned)(wpx * 8);
    switch (mt * 8 + a.mtp) {
        case 5 * 8 + 5: conv_wino2_launch_t<5, 5>(a, grid, ipx, wpx, st); break;
        case 2 * 8 + 2: conv_wino2_launch_t<2, 2>(a, grid, ipx, wpx, st); break;
        case 1 * 8 + 1: conv_wino2_launch_t<1, 1>(a, grid, ipx, wpx, st); break;
        case 1 * 8 + 2: conv_wino2_launch_t<1, 2>(a, grid, ipx, wpx, st); break;
        case 1 * 8 + 5: conv_wino2_launch_t<1, 5>(a, grid, ipx, wpx, st); break;
        default: return SINDDM_E_BADSHAPE;
    }
    if (rec) {
        (void)hipEventRecord(prof.ev[2 * prof.used + 1], st);
        const double fl = 2.0 * a.B * a.H * (a.Wt > 0 ? a.Wt : a.W) * (double)a.Cout * 9.0 * a.Cin;   // algorithmic (direct-conv) FLOPs
        prof.note(1, fl, fl * (16.0 / 36.0), 2);
    }
    SINDDM_LAUNCH_CHECK();
    return 0;
}

}  // namespace sinddm
