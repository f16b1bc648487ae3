// l2_bw_probe.hip -- aggregate load bandwidth into the CUs by footprint (standalone; hipcc --offload-arch=gfx950 -O3).
// Every workgroup streams `bytes_per_wg` with 16-byte requests, 8 in flight per lane, cyclically over a footprint of F bytes
// that all workgroups share (staggered starts): F = 2 MB stays in every XCD's 4 MB L2, 64 MB in the 256 MB Infinity Cache,
// 2 GB comes from HBM.  Prints TB/s and B/clk/CU (2.4 GHz, 256 CUs) for 1, 2, 4 and 8 resident waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void probe(const uint4 *__restrict__ buf, size_t f_vec, size_t per_wg_vec, unsigned *out) {
    const size_t lane = threadIdx.x;
    size_t pos = ((size_t)blockIdx.x * 9973u * 256u) % f_vec;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = 0; i < per_wg_vec; i += 256 * 8) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            size_t p = pos + lane + 256 * j;
            if (p >= f_vec) p -= f_vec;
            v[j] = buf[p];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w; }
        pos += 256 * 8;
        if (pos >= f_vec) pos -= f_vec;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}

int main() {
    const size_t cap = (size_t)2 << 30;
    uint4 *buf; unsigned *out;
    hipMalloc(&buf, cap + 4096); hipMalloc(&out, 4);
    hipMemset(buf, 1, cap);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t fps[] = {(size_t)512 << 10, (size_t)2 << 20, (size_t)16 << 20, (size_t)64 << 20, (size_t)192 << 20, cap};
    for (size_t F : fps)
        for (int wgs_per_cu : {1, 2, 4, 8}) {
            const int grid = 256 * wgs_per_cu;
            const size_t per_wg = (size_t)8 << 20;      // 8 MB per workgroup
            const size_t f_vec = F / 16, per_vec = per_wg / 16;
            probe<<<grid, 256>>>(buf, f_vec, per_vec, out);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            probe<<<grid, 256>>>(buf, f_vec, per_vec, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)grid * per_wg;
            printf("footprint %8.1f MB  wgs/CU %d (waves/SIMD %d): %7.2f TB/s  %6.1f B/clk/CU\n", F / 1048576.0, wgs_per_cu, wgs_per_cu,
                   bytes / ms / 1e9, bytes / (ms * 1e-3) / 2.4e9 / 256);
        }
    return 0;
}
