// Micro-test: does a STRUCTURED buffer descriptor (stride = row pitch, index = row, offset = column * 4) clip a 16-byte
// access at the END OF THE ROW per dword?  If it does, 16-byte loads / stores can serve images whose width is not a
// multiple of 4 without per-piece masks.  Rows of 13 floats; lane l reads row l / 4,
// columns 4 (l % 4) .. + 3: the group at column 12 has one float inside the row and three behind its end.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

__global__ void probe(const float* src, float* dst_loaded, float* store_target, int rows, int W) {
    const int lane = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), (short)(W * 4), rows, 0x00020000);
    i32x2 va = {lane >> 2, (lane & 3) * 16};          // {index = row, byte offset inside the row}
    f32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 idxen offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(va), "s"(rs) : "memory");
    for (int e = 0; e < 4; ++e) dst_loaded[lane * 4 + e] = v[e];
    // store side: write 100 + lane into the same groups of a second (pre-zeroed) image
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(store_target, (short)(W * 4), rows, 0x00020000);
    f32x4 w = {100.f + lane, 200.f + lane, 300.f + lane, 400.f + lane};
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 idxen offen\n\ts_waitcnt vmcnt(0)" ::"v"(w), "v"(va), "s"(rd) : "memory");
}

int main() {
    const int rows = 8, W = 13, n = rows * W;
    std::vector<float> h(n + 16);
    for (int i = 0; i < n + 16; ++i) h[i] = (float)i;
    float *src, *ld, *st;
    hipMalloc(&src, (n + 16) * 4); hipMalloc(&ld, 64 * 4 * 4); hipMalloc(&st, (n + 16) * 4);
    hipMemcpy(src, h.data(), (n + 16) * 4, hipMemcpyHostToDevice);
    hipMemset(st, 0, (n + 16) * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, ld, st, rows, W);
    std::vector<float> out(256), sto(n + 16);
    hipMemcpy(out.data(), ld, 256 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(sto.data(), st, (n + 16) * 4, hipMemcpyDeviceToHost);
    int clipped = 0, spilled = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int row = lane >> 2, c0 = (lane & 3) * 4;
        for (int e = 0; e < 4; ++e) {
            const float got = out[lane * 4 + e];
            const bool inside = row < rows && c0 + e < W;
            const float want_in = (float)(row * W + c0 + e);
            if (inside && got != want_in) { printf("LOAD MISMATCH lane %d e %d got %g want %g\n", lane, e, got, want_in); }
            if (!inside) { if (got == 0.f) ++clipped; else { ++spilled; if (spilled < 6) printf("load behind the row end: lane %d (row %d col %d) returned %g\n", lane, row, c0 + e, got); } }
        }
    }
    printf("loads : dwords behind the row end / behind the last row: %d returned 0 (clipped), %d returned memory\n", clipped, spilled);
    int wrong = 0;
    for (int i = 0; i < n; ++i) {
        const int row = i / W, col = i % W, lane = row * 4 + col / 4, e = col % 4;
        const float want = (e + 1) * 100.f + lane;
        if (sto[i] != want) { if (++wrong < 6) printf("store: element (row %d col %d) = %g, want %g\n", row, col, sto[i], want); }
    }
    int tail = 0;
    for (int i = n; i < n + 16; ++i) tail += sto[i] != 0.f;
    printf("stores: %d of %d in-row elements wrong (a store behind a row end that was NOT clipped overwrites the next row), %d writes behind the image\n", wrong, n, tail);
    return 0;
}
