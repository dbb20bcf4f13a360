// Dense-block instantiations of the bf16 forward kernel (layered path: one layer, a block of its output tiles).
// Split-bf16: up to 8 tiles per workgroup.  Plain bf16 (one plane): a third of the registers and LDS per tile, so up to
// 16 tiles -- a 512-wide layer in two launches instead of four.
#include "fwd_bf16_kernel.hpp"

#define NTCASE(N_, P_) case N_: return launch_pipe_nt<N_, P_, true>(a, W16, st);

int isk_launch_mlp_fwd_bf16_dense(const MlpFwdArgs& a, const void* W16, int planes, hipStream_t st) {
    if (planes == 1) {
        switch (a.nt1) {
            NTCASE(1, 1) NTCASE(2, 1) NTCASE(3, 1) NTCASE(4, 1) NTCASE(5, 1) NTCASE(6, 1) NTCASE(7, 1) NTCASE(8, 1)
            NTCASE(9, 1) NTCASE(10, 1) NTCASE(11, 1) NTCASE(12, 1) NTCASE(13, 1) NTCASE(14, 1) NTCASE(15, 1) NTCASE(16, 1)
            default: return ISK_EUNSUPPORTED;
        }
    }
    switch (a.nt1) {
        NTCASE(1, 3) NTCASE(2, 3) NTCASE(3, 3) NTCASE(4, 3) NTCASE(5, 3) NTCASE(6, 3) NTCASE(7, 3) NTCASE(8, 3)
        default: return ISK_EUNSUPPORTED;
    }
}
