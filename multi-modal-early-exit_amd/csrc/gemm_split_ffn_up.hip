// FFN-up instantiation of the split GEMM kernel (256 x 256 tile, GELU epilogue, split output, three MFMA terms) in a translation unit of
// its own, so that the Makefile can give it its own instruction-scheduling strategy (gemm_split_kernel.h).
#include "gemm_split_kernel.h"

namespace mmee {

void launch_split_ffn_up(const GemmArgs& a, int max_m, int num_cus, hipStream_t s) {
    launch_split_one<CfgC, EPI_GELU, true>(a, max_m, num_cus, s);
}

}  // namespace mmee
