// Single-instantiation build of the tile GEMM kernels for ISA inspection (build container only):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S scripts/isa/gemm_one.hip -o /tmp/isa/one.s
//   python scripts/isa/sched.py /tmp/isa/one.s 'gemm_bf16c_k64_kernel<128, 128' --loop
#include "../../detr-tensorflow_amd/csrc/gemm_kernels.h"
namespace detr {
template __global__ void gemm_bf16c_k64_kernel<128, 128, 2, 2, false, false>(GemmArgs);   // split-K weight gradients (1 workgroup / CU)
template __global__ void gemm_bf16c_k64_kernel<64, 64, 2, 2, true, true>(GemmArgs);
template __global__ void gemm_bf16c_kernel<128, 128, 2, 2, true, true, true, true>(GemmArgs);
template __global__ void gemm_bf16c_kernel<64, 64, 2, 2, true, true, false, true>(GemmArgs);      // transformer linears (fp32 activations x bf16 weights)   // big 1x1-conv forward / dgrad
}
