// the optimizer rule of every kernel that updates parameters (opt_step_kernel, train_grad1n carrying the step, the step-boundary
// kernel of the wide-network path)
#pragma once
#include "isk_internal.h"

// WeightDecay + Adam | Nesterov on one parameter (Optimisers.jl 0.4.7).  ONE definition for opt_step_kernel and for the kernels
// that carry the step themselves, with every product / sum pinned (explicit fmaf, __fmul_rn): left to the compiler, which product
// of `gs * gscale + reg * th` is contracted into the fma depends on the surrounding code, and the fused and unfused steps then
// differ in the last bit of a third of the parameters.
__device__ __forceinline__ void opt_rule(const OptArgs& a, float bc1, float bc2, float gs, float& th, float& mi, float& vi) {
    const float g = fmaf(a.reg, th, __fmul_rn(gs, a.gscale));
    float d;
    if (a.kind == ISK_OPT_ADAM) {
        const float m = fmaf(a.beta1, mi, __fmul_rn(1.f - a.beta1, g));
        const float v = fmaf(a.beta2, vi, __fmul_rn(__fmul_rn(1.f - a.beta2, g), g));
        mi = m; vi = v;
        d = __fmul_rn(__fdiv_rn(__fdiv_rn(m, bc1), __fadd_rn(sqrtf(__fdiv_rn(v, bc2)), a.eps)), a.lr);
    } else {
        d = fmaf(-__fmul_rn(a.rho, a.rho), mi, __fmul_rn(__fmul_rn(1.f + a.rho, a.lr), g));
        mi = fmaf(a.rho, mi, -__fmul_rn(a.lr, g));
    }
    th = __fsub_rn(th, d);
}

