// Probe of ds_read_b64_tr_b16 (gfx950): prints which LDS element every (lane, j) receives.
// build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC tr_b16_probe.hip -o tr_b16_probe.so
#include <hip/hip_runtime.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(short *out, int mode) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int elem;                                  // element offset (shorts) of this lane's 8-byte chunk
    if (mode == 0) elem = l * 4;               // contiguous chunks
    else elem = (l >> 4) * 256 + ((l & 15) >> 2) * 64 + (l & 3) * 4;   // rows 64 elements apart inside a group
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lds + elem));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
extern "C" int run_probe(short *out, int mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, mode);
    return (int)hipDeviceSynchronize();
}
