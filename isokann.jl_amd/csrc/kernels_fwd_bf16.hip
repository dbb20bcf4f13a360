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

// split-f16 x2 weight planes (forward_mode 3): fp32 fragment-linear [chunk16][nt][64][4] -> f16 [chunk32][2][nt][64][8],
// every weight scaled by the power of two that brings the tensor maximum (from fold_kernel) into [2^14, 2^15) (isk_f16_scale):
//   hi = f16(w 2^s), lo = f16(w 2^s - hi);   wscale_inv[0] = 2^-s for the forward epilogue
__global__ __launch_bounds__(256) void fold_f16_kernel(const float* __restrict__ Wt, _Float16* __restrict__ out, int nchunk32, int nt,
                                                       int nchunk16, const unsigned* __restrict__ wmax, float* __restrict__ wscale_inv) {
    float sc, inv;
    isk_f16_scale(__builtin_bit_cast(float, wmax[0]), sc, inv);
    if (blockIdx.x == 0 && threadIdx.x == 0) wscale_inv[0] = inv;
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
        if (c16 < nchunk16) w = Wt[(((long long)c16 * nt + t) * 64 + (((k16 >> 2) << 4) | li)) * 4 + (k16 & 3)] * sc;
        const _Float16 hi = (_Float16)w;
        const _Float16 lo = (_Float16)(w - (float)hi);
        const long long base = (long long)c32 * 2 * nt;
        out[((base + t) * 64 + lane) * 8 + j] = hi;
        out[((base + nt + t) * 64 + lane) * 8 + j] = lo;
    }
}

void isk_launch_fold_f16(const float* Wt, void* out, int Kp, int nt, const unsigned* wmax, float* wscale_inv, hipStream_t st) {
    const int nchunk32 = isk_bf16_chunks(Kp), nchunk16 = Kp / 16;
    const long long total = (long long)nchunk32 * nt * 64 * 8;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fold_f16_kernel, dim3(blocks), dim3(256), 0, st, Wt, (_Float16*)out, nchunk32, nt, nchunk16, wmax, wscale_inv);
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

// fits(): dry run of the dispatch (LDS footprint of the shape)
static int f16x2_dispatch(const MlpFwdArgs& a, const void* W16, hipStream_t st, bool dry) {
    if (a.Fp % 32 != 0 || a.dense_mode || !a.xscale || !a.wscale) return ISK_EUNSUPPORTED;
#ifdef ISK_DEV_NT9
#define F16CASE(N_) case N_: if (N_ == 9) return launch_pipe_f16<9>(a, W16, st, dry); return ISK_EUNSUPPORTED;
#else
#define F16CASE(N_) case N_: return launch_pipe_f16<N_>(a, W16, st, dry);
#endif
    switch (a.nt1) {
        F16CASE(1) F16CASE(2) F16CASE(3) F16CASE(4) F16CASE(5) F16CASE(6) F16CASE(7) F16CASE(8) F16CASE(9) F16CASE(10)
        default: return ISK_EUNSUPPORTED;
    }
#undef F16CASE
}
int isk_launch_mlp_fwd_f16x2(const MlpFwdArgs& a, const void* W16, hipStream_t st) {
    if (a.S <= 0) return 0;
    return f16x2_dispatch(a, W16, st, false);
}
int isk_mlp_fwd_f16x2_fits(const MlpFwdArgs& a) { return f16x2_dispatch(a, nullptr, nullptr, true) == 0; }

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
