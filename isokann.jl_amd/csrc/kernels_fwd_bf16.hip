// Split-bf16 / plain-bf16 forward, fused networks (layer 1 + tail layers in one launch): instantiations and dispatch.
// The kernel itself is in fwd_bf16_kernel.hpp; the dense-block instantiations of the layered path are compiled in
// kernels_fwd_bf16_dense.hip (a translation unit of their own: the templates take minutes to compile).
#include "fwd_bf16_kernel.hpp"

// fp32 fragment-linear layer-1 weights [chunk16][nt][64][4]  ->  bf16 planes [chunk32][3][nt][64][8]
// lane = (kq << 4) | o_local holds W[o][k = 32*c + 8*kq + j], j = 0..7: the B operand of v_mfma_f32_16x16x32_bf16
__global__ __launch_bounds__(256) void fold_bf16_kernel(const float* __restrict__ Wt, __bf16* __restrict__ out, int nchunk32,
                                                        int nt, int nchunk16, int planes) {
    const long long total = (long long)nchunk32 * nt * 64 * 8;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int j = e & 7;
        const int lane = (e >> 3) & 63;
        const long long ct = e >> 9;
        const int t = (int)(ct % nt);
        const int c32 = (int)(ct / nt);
        const int li = lane & 15, kq = lane >> 4;
        const int c16 = 2 * c32 + (kq >> 1);
        const int k16 = 8 * (kq & 1) + j;
        float w = 0.f;
        if (c16 < nchunk16) w = Wt[(((long long)c16 * nt + t) * 64 + (((k16 >> 2) << 4) | li)) * 4 + (k16 & 3)];
        const __bf16 b0 = (__bf16)w;
        const float r1 = w - (float)b0;
        const __bf16 b1 = (__bf16)r1;
        const __bf16 b2 = (__bf16)(r1 - (float)b1);
        const long long base = (long long)c32 * planes * nt;
        out[((base + 0 * nt + t) * 64 + lane) * 8 + j] = b0;
        if (planes == 3) {
            out[((base + 1 * nt + t) * 64 + lane) * 8 + j] = b1;
            out[((base + 2 * nt + t) * 64 + lane) * 8 + j] = b2;
        }
    }
}

// chunks in the plane image: padded with zero planes (the pipelined kernel runs an even number of stages)
int isk_bf16_chunks(int Kp) { return ((Kp + 31) / 32 + 5) / 6 * 6; }

void isk_launch_fold_bf16(const float* Wt, void* out, int Kp, int nt, int planes, hipStream_t st) {
    const int nchunk32 = isk_bf16_chunks(Kp), nchunk16 = Kp / 16;
    long long total = (long long)nchunk32 * nt * 64 * 8;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fold_bf16_kernel, dim3(blocks), dim3(256), 0, st, Wt, (__bf16*)out, nchunk32, nt, nchunk16, planes);
}


int isk_launch_mlp_fwd_bf16_dense(const MlpFwdArgs& a, const void* W16, int planes, hipStream_t st);   // kernels_fwd_bf16_dense.hip

#ifdef ISK_DEV_NT9        // developer builds (asm inspection, diagnostic libraries): the C3 shape only
#define NTCASE(N_, P_) case N_: if (N_ == 9) return launch_pipe_nt<9, P_, false>(a, W16, st); return ISK_EUNSUPPORTED;
#else
#define NTCASE(N_, P_) case N_: return launch_pipe_nt<N_, P_, false>(a, W16, st);
#endif

int isk_launch_mlp_fwd_bf16x3(const MlpFwdArgs& a, const void* W16, int planes, hipStream_t st) {
    if (a.S <= 0) return 0;
    if (a.Fp % 32 != 0) return ISK_EUNSUPPORTED;
    if (a.dense_mode) return isk_launch_mlp_fwd_bf16_dense(a, W16, planes, st);
    // up to ISK_MAX_NT output tiles per workgroup (first hidden width <= 160)
    if (planes == 1) {
        switch (a.nt1) {
            NTCASE(1, 1) NTCASE(2, 1) NTCASE(3, 1) NTCASE(4, 1) NTCASE(5, 1) NTCASE(6, 1) NTCASE(7, 1) NTCASE(8, 1) NTCASE(9, 1) NTCASE(10, 1)
            default: return ISK_EUNSUPPORTED;
        }
    }
    switch (a.nt1) {
        NTCASE(1, 3) NTCASE(2, 3) NTCASE(3, 3) NTCASE(4, 3) NTCASE(5, 3) NTCASE(6, 3) NTCASE(7, 3) NTCASE(8, 3) NTCASE(9, 3) NTCASE(10, 3)
        default: return ISK_EUNSUPPORTED;
    }
}
