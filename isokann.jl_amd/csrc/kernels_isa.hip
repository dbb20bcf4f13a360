// Device-side inner simplex algorithm (TransformISA, src/isotarget.jl:74-127 of the reference) as a
// short chain of row-parallel kernels, Float64 throughout like the reference's host solve:
//   X = Float64(ks')                          N x d
//   ind = PCCAPlus.indexmap(X or X*C^-1/2)     d steps: max-norm row, translate / normalise + project out
//   A = inv(X[ind, :]);  T = A' * ks
//   fixperm: argmin over the d! row permutations (lexicographic, first minimiser) of sum |T[p] - chi|
//   loss weights 1 ./ std(target, dims=2) (corrected) from the Float32-rounded target
// Every kernel is row-parallel over G = ceil(N / 1024) workgroups and leaves per-workgroup partials
// (argmax candidate incl. its row, cost matrix, moments); the NEXT kernel's workgroups each finalize
// those G partials redundantly, so there is no grid barrier, no atomics and no host round trip, and the
// cost does not grow with N per workgroup (data-parallel runs feed the all-gathered N_global rows).
// Launches: 1 (+1 whitening) + (d-1) + 1 + 1 + 1.  status bit 1 (value 2) = singular / non-finite simplex.
#include "isk_internal.h"

#define WAVE 64
#define ISA_T 256
#define ISA_R 4                 // rows per thread: loads of a thread's rows are issued together
#define ISA_ROWS (ISA_T * ISA_R)
#define ISA_MAXD 8

// K chi as the simplex solve sees it: the unrounded double K-mean where the engine has it (IsaArgs.ks64), else the widened Float32
__device__ __forceinline__ double isa_ks(const IsaArgs& a, long long i) { return a.ks64 ? a.ks64[i] : (double)a.ks[i]; }

// partial record of one workgroup: [0] max norm, [1] row index (as double), [2..2+D) that row of X
template <int D> struct IsaLayout { static constexpr int PREC = 2 + D; };

template <int M>
__device__ __forceinline__ void block_sum_multi(double (&v)[M], double* sh /* [M][ISA_T/64] */) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[m] += __shfl_xor(v[m], o, WAVE);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int m = 0; m < M; ++m) sh[m * (ISA_T / 64) + (threadIdx.x >> 6)] = v[m];
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < M; ++m) {
        double t = 0;
#pragma unroll
        for (int i = 0; i < ISA_T / 64; ++i) t += sh[m * (ISA_T / 64) + i];
        v[m] = t;
    }
}

// sum of G partial vectors (length M, at P[g * M + i]) in workgroup order g = 0 .. G-1, result in out[i] (LDS), i < M.  The partials
// are requested by ALL threads at once and staged in LDS: written as `for g: s += P[g * M + tid]` the loop was G dependent memory
// round trips (each load waited for before the next was issued: ~1 us apiece beside the featurizer), now it is one.
#define ISA_STAGE 2048             /* doubles of LDS staging (16 KB): G * M beyond that falls back to the plain loop */
__device__ __forceinline__ void sum_partials(const double* __restrict__ P, int G, int M, double* stage, double* out) {
    const int tot = G * M;
    if (tot <= ISA_STAGE) {
        for (int i = threadIdx.x; i < tot; i += blockDim.x) stage[i] = P[i];
        __syncthreads();
        if ((int)threadIdx.x < M) {
            double s = 0;
            for (int g = 0; g < G; ++g) s += stage[g * M + threadIdx.x];
            out[threadIdx.x] = s;
        }
    } else if ((int)threadIdx.x < M) {
        double s = 0;
        for (int g = 0; g < G; ++g) s += P[(long long)g * M + threadIdx.x];
        out[threadIdx.x] = s;
    }
    __syncthreads();
}

// workgroup argmax of (value, first index); winner broadcast through shd[0] / shi[0]
__device__ __forceinline__ void block_argmax(double bv, long long bi, double* shd, long long* shi) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(bv, o, WAVE);
        const long long oi = __shfl_xor(bi, o, WAVE);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { shd[threadIdx.x >> 6] = bv; shi[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < ISA_T / 64; ++i)
            if (shd[i] > bv || (shd[i] == bv && shi[i] < bi)) { bv = shd[i]; bi = shi[i]; }
        shd[0] = bv;
        shi[0] = bi;
    }
    __syncthreads();
}

// finalize the G partial argmax records of the previous kernel: every workgroup does this redundantly.
// On return shd[0] = max norm, shi[0] = row index, vrow_sh[k] = that row of X (old values).
// When NO row is a candidate (every norm is NaN: the model has diverged) the records carry index -1: the selection
// then falls back to row 0 with a zero row, the status word gets bit 2 (the reference throws DomainError at
// src/isotarget.jl:94-97) and every index the later kernels derive from shi[0] stays inside the arrays.
template <int D>
__device__ __forceinline__ void finalize_argmax(const double* __restrict__ P, int G, double* shd, long long* shi,
                                                double* vrow_sh, int* __restrict__ status, long long N) {
    constexpr int PREC = IsaLayout<D>::PREC;
    double bv = -1.0;
    long long bi = 0x7fffffffffffffffLL;
    double brow[D];                                             // the row of this thread's best record: it travels with the record, so
#pragma unroll                                                  // the winner's row needs no second (dependent) round trip
    for (int k = 0; k < D; ++k) brow[k] = 0.0;
    for (int g = threadIdx.x; g < G; g += ISA_T) {
        const double v = P[(long long)g * PREC];
        const long long i = (long long)P[(long long)g * PREC + 1];
        double row[D];
#pragma unroll
        for (int k = 0; k < D; ++k) row[k] = P[(long long)g * PREC + 2 + k];
        if (i >= 0 && (v > bv || (v == bv && i < bi))) {
            bv = v; bi = i;
#pragma unroll
            for (int k = 0; k < D; ++k) brow[k] = row[k];
        }
    }
    const long long mine = bi;
    block_argmax(bv, bi, shd, shi);
    const long long sel = shi[0];
    const bool none = !(sel >= 0 && sel < N);                  // workgroup-uniform
    __syncthreads();                                            // everyone has read the winner before it is replaced
    if (none) {
        if (threadIdx.x == 0) {
            shi[0] = 0;
            shd[0] = 0.0;
            if (blockIdx.x == 0) atomicOr(status, 2);
        }
        if (threadIdx.x < D) vrow_sh[threadIdx.x] = 0.0;
    } else if (mine == sel) {                                   // exactly one thread: a row index names one record
#pragma unroll
        for (int k = 0; k < D; ++k) vrow_sh[k] = brow[k];
    }
    __syncthreads();
}

// write this workgroup's partial argmax record (value, index, row values).  brow: the values of this thread's own best row (the
// winner writes its row from registers: re-reading it from X behind the barrier was one more dependent round trip per kernel)
template <int D>
__device__ __forceinline__ void write_partial_argmax(double bv, long long bi, const double (&brow)[D], double* __restrict__ P,
                                                     double* shd, long long* shi) {
    constexpr int PREC = IsaLayout<D>::PREC;
    const long long mine = bi;
    block_argmax(bv, bi, shd, shi);
    const long long sel = shi[0];
    const bool any = sel != 0x7fffffffffffffffLL;              // false: no row of this slab was a candidate (all NaN, or an empty slab)
    if (threadIdx.x == 0) {
        P[(long long)blockIdx.x * PREC] = shd[0];
        P[(long long)blockIdx.x * PREC + 1] = any ? (double)sel : -1.0;
        if (!any) {
#pragma unroll
            for (int k = 0; k < D; ++k) P[(long long)blockIdx.x * PREC + 2 + k] = 0.0;
        }
    }
    if (any && mine == sel) {                                   // exactly one thread owns the winning row
#pragma unroll
        for (int k = 0; k < D; ++k) P[(long long)blockIdx.x * PREC + 2 + k] = brow[k];
    }
}

// ---- k0: X = Float64(ks'), finiteness; norms -> partial argmax, or whitening moments -------------------
template <int D>
__global__ __launch_bounds__(ISA_T) void isa_init_kernel(IsaArgs a, double* __restrict__ P, double* __restrict__ Pm) {
    ISK_LATENCY_CHAIN_PRIO();
    __shared__ double shd[ISA_MAXD * ISA_MAXD * (ISA_T / 64)];
    __shared__ long long shi[ISA_T / 64];
    const long long N = a.N;
    double* X = a.work;
    double bv = -1.0;
    long long bi = 0x7fffffffffffffffLL;
    int nf = 0;
    double brow[D];
#pragma unroll
    for (int k = 0; k < D; ++k) brow[k] = 0.0;
    double cm[D * D];
#pragma unroll
    for (int i = 0; i < D * D; ++i) cm[i] = 0;
    const long long base = (long long)blockIdx.x * ISA_ROWS + threadIdx.x;
    double kv[ISA_R][D];
#pragma unroll
    for (int r = 0; r < ISA_R; ++r) {
        const long long n = base + (long long)r * ISA_T;
#pragma unroll
        for (int k = 0; k < D; ++k) kv[r][k] = (n < N) ? isa_ks(a, n * D + k) : 0.0;
    }
#pragma unroll
    for (int r = 0; r < ISA_R; ++r) {
        const long long n = base + (long long)r * ISA_T;
        if (n < N) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                nf |= !isfinite(kv[r][k]);
                const double x = (double)kv[r][k];
                X[n * D + k] = x;
                s += x * x;
            }
            if (a.whitening) {
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) cm[i * D + j] += (double)kv[r][i] * (double)kv[r][j];
            }
            s = sqrt(s);
            if (s > bv) {
                bv = s; bi = n;
#pragma unroll
                for (int k = 0; k < D; ++k) brow[k] = (double)kv[r][k];
            }
        }
    }
    if (nf) atomicOr(a.status, 2);
    if (a.whitening) {
        block_sum_multi<D * D>(cm, shd);
        if (threadIdx.x < D * D) {
            double v = 0;
#pragma unroll
            for (int i = 0; i < D * D; ++i) v = (threadIdx.x == i) ? cm[i] : v;
            Pm[(long long)blockIdx.x * D * D + threadIdx.x] = v;
        }
    } else {
        write_partial_argmax<D>(bv, bi, brow, P, shd, shi);
    }
}

// ---- whitening: C = sum of partial moments / N, W = C^(-1/2) (Jacobi), X <- X*W, norms -------------------
template <int D>
__global__ __launch_bounds__(ISA_T) void isa_whiten_kernel(IsaArgs a, const double* __restrict__ Pm, int G,
                                                           double* __restrict__ P) {
    ISK_LATENCY_CHAIN_PRIO();
    __shared__ double shd[ISA_MAXD * ISA_MAXD * (ISA_T / 64)];
    __shared__ long long shi[ISA_T / 64];
    __shared__ double Wm[D * D];
    __shared__ double stage[ISA_STAGE];
    __shared__ int badw;
    const long long N = a.N;
    double* X = a.work;
    if (threadIdx.x == 0) badw = 0;
    sum_partials(Pm, G, D * D, stage, Wm);
    if (threadIdx.x < D * D) Wm[threadIdx.x] = Wm[threadIdx.x] / (double)N;
    __syncthreads();
    if (threadIdx.x == 0) {
        double V[D * D], Aj[D * D];
        for (int i = 0; i < D * D; ++i) { Aj[i] = Wm[i]; V[i] = 0; }
        for (int i = 0; i < D; ++i) V[i * D + i] = 1;
        for (int sweep = 0; sweep < 100; ++sweep) {
            double off = 0;
            for (int i = 0; i < D; ++i)
                for (int j = i + 1; j < D; ++j) off += Aj[i * D + j] * Aj[i * D + j];
            if (off < 1e-300) break;
            for (int p = 0; p < D; ++p)
                for (int q = p + 1; q < D; ++q) {
                    if (fabs(Aj[p * D + q]) < 1e-300) continue;
                    const double th = (Aj[q * D + q] - Aj[p * D + p]) / (2 * Aj[p * D + q]);
                    const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1));
                    const double c = 1 / sqrt(t * t + 1), sn = t * c;
                    for (int k = 0; k < D; ++k) {
                        const double akp = Aj[k * D + p], akq = Aj[k * D + q];
                        Aj[k * D + p] = c * akp - sn * akq; Aj[k * D + q] = sn * akp + c * akq;
                    }
                    for (int k = 0; k < D; ++k) {
                        const double apk = Aj[p * D + k], aqk = Aj[q * D + k];
                        Aj[p * D + k] = c * apk - sn * aqk; Aj[q * D + k] = sn * apk + c * aqk;
                    }
                    for (int k = 0; k < D; ++k) {
                        const double vkp = V[k * D + p], vkq = V[k * D + q];
                        V[k * D + p] = c * vkp - sn * vkq; V[k * D + q] = sn * vkp + c * vkq;
                    }
                }
        }
        for (int i = 0; i < D; ++i)
            if (!(Aj[i * D + i] > 0)) badw = 1;
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) {
                double sacc = 0;
                for (int k = 0; k < D; ++k) sacc += V[i * D + k] * V[j * D + k] / sqrt(Aj[k * D + k]);
                Wm[i * D + j] = sacc;
            }
        if (badw && blockIdx.x == 0) atomicOr(a.status, 2);
    }
    __syncthreads();
    double Wl[D * D];
#pragma unroll
    for (int i = 0; i < D * D; ++i) Wl[i] = Wm[i];
    double bv = -1.0;
    long long bi = 0x7fffffffffffffffLL;
    double brow[D];
#pragma unroll
    for (int k = 0; k < D; ++k) brow[k] = 0.0;
    const long long base = (long long)blockIdx.x * ISA_ROWS + threadIdx.x;
    double xv[ISA_R][D];
#pragma unroll
    for (int r = 0; r < ISA_R; ++r) {
        const long long n = base + (long long)r * ISA_T;
#pragma unroll
        for (int k = 0; k < D; ++k) xv[r][k] = (n < N) ? X[n * D + k] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < ISA_R; ++r) {
        const long long n = base + (long long)r * ISA_T;
        if (n < N) {
            double s = 0;
            double yv[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double y = 0;
#pragma unroll
                for (int k = 0; k < D; ++k) y += xv[r][k] * Wl[k * D + j];
                X[n * D + j] = y;
                yv[j] = y;
                s += y * y;
            }
            s = sqrt(s);
            if (s > bv) {
                bv = s; bi = n;
#pragma unroll
                for (int k = 0; k < D; ++k) brow[k] = yv[k];
            }
        }
    }
    write_partial_argmax<D>(bv, bi, brow, P, shd, shi);
}

// ---- indexmap step j (j = 0 .. d-2): finalize the argmax of step j, update the rows, partial argmax of j+1 ----
template <int D>
__global__ __launch_bounds__(ISA_T) void isa_step_kernel(IsaArgs a, int j, const double* __restrict__ Pin, int G,
                                                         double* __restrict__ Pout, int* __restrict__ ind) {
    ISK_LATENCY_CHAIN_PRIO();
    __shared__ double shd[ISA_MAXD * ISA_MAXD * (ISA_T / 64)];
    __shared__ long long shi[ISA_T / 64];
    __shared__ double vrow[D];
    const long long N = a.N;
    double* X = a.work;
    // this thread's rows do not depend on the argmax: requested before it is finalized (two dependent round trips and
    // three barriers), not behind it
    const long long base = (long long)blockIdx.x * ISA_ROWS + threadIdx.x;
    double xv[ISA_R][D];
#pragma unroll
    for (int r = 0; r < ISA_R; ++r) {
        const long long n = base + (long long)r * ISA_T;
#pragma unroll
        for (int k = 0; k < D; ++k) xv[r][k] = (n < N) ? X[n * D + k] : 0.0;
    }
    finalize_argmax<D>(Pin, G, shd, shi, vrow, a.status, N);
    const double rn = shd[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) ind[j] = (int)shi[0];
    const double inv = (j == 0) ? 1.0 : 1.0 / rn;
    double vr[D];
#pragma unroll
    for (int k = 0; k < D; ++k) vr[k] = vrow[k] * inv;
    double bv = -1.0;
    long long bi = 0x7fffffffffffffffLL;
    double brow[D];
#pragma unroll
    for (int k = 0; k < D; ++k) brow[k] = 0.0;
    __syncthreads();          // every thread holds its (old) rows before any row of this slab is overwritten
#pragma unroll
    for (int r = 0; r < ISA_R; ++r) {
        const long long n = base + (long long)r * ISA_T;
        if (n < N) {
            double s = 0;
            if (j == 0) {
#pragma unroll
                for (int k = 0; k < D; ++k) xv[r][k] -= vr[k];
            } else {
                double dot = 0;
#pragma unroll
                for (int k = 0; k < D; ++k) { xv[r][k] *= inv; dot += xv[r][k] * vr[k]; }
#pragma unroll
                for (int k = 0; k < D; ++k) xv[r][k] -= dot * vr[k];
            }
#pragma unroll
            for (int k = 0; k < D; ++k) { X[n * D + k] = xv[r][k]; s += xv[r][k] * xv[r][k]; }
            s = sqrt(s);
            if (s > bv) {
                bv = s; bi = n;
#pragma unroll
                for (int k = 0; k < D; ++k) brow[k] = xv[r][k];
            }
        }
    }
    write_partial_argmax<D>(bv, bi, brow, Pout, shd, shi);
}

// ---- finalize the last index, A = inv(ks[ind, :]), T = A' ks, partial fixperm cost matrices ------------------
template <int D>
__global__ __launch_bounds__(ISA_T) void isa_target_kernel(IsaArgs a, const double* __restrict__ Pin, int G,
                                                           int* __restrict__ ind, double* __restrict__ Pc) {
    ISK_LATENCY_CHAIN_PRIO();
    __shared__ double shd[ISA_MAXD * ISA_MAXD * (ISA_T / 64)];
    __shared__ long long shi[ISA_T / 64];
    __shared__ double vrow[D];
    __shared__ double Am[D * D];
    __shared__ double Msh[D * 2 * D];
    __shared__ int bad;
    const long long N = a.N;
    double* T = a.work + N * D;
    // this thread's ks / chi rows: requested before the argmax is finalized and thread 0 inverts the simplex
    const long long base = (long long)blockIdx.x * ISA_ROWS + threadIdx.x;
    double kv[ISA_R][D];
    float cv[ISA_R][D];
#pragma unroll
    for (int r = 0; r < ISA_R; ++r) {
        const long long n = base + (long long)r * ISA_T;
#pragma unroll
        for (int k = 0; k < D; ++k) { kv[r][k] = (n < N) ? isa_ks(a, n * D + k) : 0.0; cv[r][k] = (n < N) ? a.chi[n * D + k] : 0.f; }
    }
    // thread 0: the simplex rows of the earlier steps (indices written by the kernels before this one) are requested before this
    // kernel's own argmax is finalized -- two dependent round trips that used to stand behind it
    int idx[D];
    double kvs[D * D];
    int bad0 = 0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < D - 1; ++i) idx[i] = ind[i];
#pragma unroll
        for (int i = 0; i < D - 1; ++i)
            if (idx[i] < 0 || idx[i] >= N) { idx[i] = 0; bad0 = 1; }      // belt and braces: finalize_argmax keeps them in range
#pragma unroll
        for (int i = 0; i < D - 1; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) kvs[i * D + j] = isa_ks(a, (long long)idx[i] * D + j);
    }
    finalize_argmax<D>(Pin, G, shd, shi, vrow, a.status, N);
    if (threadIdx.x == 0) {
        bad = bad0;
        idx[D - 1] = (int)shi[0];
        if (idx[D - 1] < 0 || idx[D - 1] >= N) { idx[D - 1] = 0; bad = 1; }
        if (blockIdx.x == 0) ind[D - 1] = idx[D - 1];
        // the last row in ONE more round trip (registers first): filled element by element, the pivoting matrix lives
        // in scratch memory and every load waited for the one before it -- d*d dependent round trips on one thread
#pragma unroll
        for (int j = 0; j < D; ++j) kvs[(D - 1) * D + j] = isa_ks(a, (long long)idx[D - 1] * D + j);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (D <= 4) {
            // d <= 4 (the usual ISOKANN dimensions): the augmented matrix in REGISTERS -- every loop unrolled, the pivot row picked and
            // swapped with selects, so no element is indexed at run time.  In LDS (below) the elimination is ~100 dependent LDS
            // accesses of one thread, ~5 us that every workgroup spends in front of its row work
            double R[D][2 * D];
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int jj = 0; jj < D; ++jj) { R[i][jj] = (double)kvs[i * D + jj]; R[i][D + jj] = (i == jj) ? 1.0 : 0.0; }
            int badl = bad;
#pragma unroll
            for (int c = 0; c < D; ++c) {
                if (!badl) {
                    int pr = c;
                    double pa = fabs(R[c][c]), pv = R[c][c];
#pragma unroll
                    for (int r = c + 1; r < D; ++r) {
                        const double av = fabs(R[r][c]);
                        if (av > pa) { pa = av; pr = r; pv = R[r][c]; }
                    }
                    if (!(fabs(pv) > 0.0) || !isfinite(pv)) badl = 1;
                    else {
#pragma unroll
                        for (int r = c + 1; r < D; ++r)
                            if (pr == r) {
#pragma unroll
                                for (int jj = 0; jj < 2 * D; ++jj) { const double t = R[r][jj]; R[r][jj] = R[c][jj]; R[c][jj] = t; }
                            }
#pragma unroll
                        for (int r = 0; r < D; ++r) {
                            if (r == c) continue;
                            const double f = R[r][c] / pv;
                            if (f != 0.0) {
#pragma unroll
                                for (int jj = 0; jj < 2 * D; ++jj) R[r][jj] -= f * R[c][jj];
                                R[r][c] = 0.0;
                            }
                        }
                    }
                }
            }
            if (!badl) {
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    const double inv = 1.0 / R[c][c];
#pragma unroll
                    for (int jj = 0; jj < 2 * D; ++jj) R[c][jj] *= inv;
                }
            }
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int jj = 0; jj < D; ++jj) {
                    const double v = badl ? 0.0 : R[i][D + jj];
                    Am[i * D + jj] = v;
                    if (!isfinite(v)) badl = 1;
                }
            bad = badl;
        } else {
            // the pivoting matrix is indexed at run time: in LDS (one thread uses it), not in a private array -- that one
            // lives in scratch memory and every access of the elimination is a dependent memory round trip
            double* M = Msh;
    #pragma unroll
            for (int i = 0; i < D; ++i)
    #pragma unroll
                for (int j = 0; j < D; ++j) { M[i * 2 * D + j] = (double)kvs[i * D + j]; M[i * 2 * D + D + j] = (i == j) ? 1.0 : 0.0; }
            // elimination with the multiplier taken against the UNSCALED pivot row (as LU does), so exactly
            // dependent rows cancel to exact zeros and are reported singular like Julia's inv()
            for (int c = 0; c < D && !bad; ++c) {
                int p = c;
                for (int r = c + 1; r < D; ++r)
                    if (fabs(M[r * 2 * D + c]) > fabs(M[p * 2 * D + c])) p = r;
                const double pv = M[p * 2 * D + c];
                if (!(fabs(pv) > 0.0) || !isfinite(pv)) { bad = 1; break; }
                if (p != c)
                    for (int j = 0; j < 2 * D; ++j) { const double t = M[p * 2 * D + j]; M[p * 2 * D + j] = M[c * 2 * D + j]; M[c * 2 * D + j] = t; }
                for (int r = 0; r < D; ++r) {
                    if (r == c) continue;
                    const double f = M[r * 2 * D + c] / pv;
                    if (f == 0.0) continue;
                    for (int j = 0; j < 2 * D; ++j) M[r * 2 * D + j] -= f * M[c * 2 * D + j];
                    M[r * 2 * D + c] = 0.0;
                }
            }
            for (int c = 0; c < D && !bad; ++c) {
                const double inv = 1.0 / M[c * 2 * D + c];
                for (int j = 0; j < 2 * D; ++j) M[c * 2 * D + j] *= inv;
            }
            for (int i = 0; i < D; ++i)
                for (int j = 0; j < D; ++j) {
                    Am[i * D + j] = bad ? 0.0 : M[i * 2 * D + D + j];
                    if (!isfinite(Am[i * D + j])) bad = 1;
                }
        }
        if (bad && blockIdx.x == 0) atomicOr(a.status, 2);
    }
    __syncthreads();
    double Al[D * D];
#pragma unroll
    for (int i = 0; i < D * D; ++i) Al[i] = Am[i];
    double cst[D * D];
#pragma unroll
    for (int i = 0; i < D * D; ++i) cst[i] = 0;
#pragma unroll
    for (int r = 0; r < ISA_R; ++r) {
        const long long n = base + (long long)r * ISA_T;
        if (n < N) {
            double tq[D];
#pragma unroll
            for (int q = 0; q < D; ++q) {
                double sacc = 0;
#pragma unroll
                for (int b = 0; b < D; ++b) sacc += Al[b * D + q] * (double)kv[r][b];
                tq[q] = sacc;
                T[n * D + q] = sacc;
            }
            if (a.permute) {
#pragma unroll
                for (int q = 0; q < D; ++q)
#pragma unroll
                    for (int i = 0; i < D; ++i) cst[q * D + i] += fabs(tq[q] - (double)cv[r][i]);
            }
        }
    }
    if (a.permute) {
        block_sum_multi<D * D>(cst, shd);
        if (threadIdx.x < D * D) {
            double v = 0;
#pragma unroll
            for (int i = 0; i < D * D; ++i) v = (threadIdx.x == i) ? cst[i] : v;
            Pc[(long long)blockIdx.x * D * D + threadIdx.x] = v;
        }
    }
}

// ---- fixperm from the summed cost matrix, Float32 target (local window), partial loss-weight moments ------
template <int D>
__global__ __launch_bounds__(ISA_T) void isa_perm_kernel(IsaArgs a, const double* __restrict__ Pc, int G,
                                                         double* __restrict__ Pmom) {
    ISK_LATENCY_CHAIN_PRIO();
    __shared__ double shd[ISA_MAXD * ISA_MAXD * (ISA_T / 64)];
    __shared__ double Cst[D * D];
    __shared__ double stage[ISA_STAGE];
    __shared__ int rperm[D];
    __shared__ int psh[2 * D];
    const long long N = a.N;
    const double* T = a.work + N * D;
    // this thread's target rows: requested before the cost sums and the permutation search
    const long long base = (long long)blockIdx.x * ISA_ROWS + threadIdx.x;
    double tv[ISA_R][D];
#pragma unroll
    for (int r = 0; r < ISA_R; ++r) {
        const long long n = base + (long long)r * ISA_T;
#pragma unroll
        for (int k = 0; k < D; ++k) tv[r][k] = (n < N) ? T[n * D + k] : 0.0;
    }
    if (threadIdx.x < D) rperm[threadIdx.x] = threadIdx.x;
    if (a.permute) {
        sum_partials(Pc, G, D * D, stage, Cst);
        if (threadIdx.x == 0) {
            int* p = psh;                           // run-time indexed: LDS, not scratch (see isa_target_kernel)
            int* best = psh + D;
            for (int i = 0; i < D; ++i) p[i] = i;
            double bc = 0;
            bool first = true;
            while (true) {
                double c = 0;
                for (int i = 0; i < D; ++i) c += Cst[p[i] * D + i];
                if (first || c < bc) { bc = c; first = false; for (int i = 0; i < D; ++i) best[i] = p[i]; }
                int i = D - 2;                      // next lexicographic permutation
                while (i >= 0 && p[i] >= p[i + 1]) --i;
                if (i < 0) break;
                int j = D - 1;
                while (p[j] <= p[i]) --j;
                { const int t = p[i]; p[i] = p[j]; p[j] = t; }
                for (int l = i + 1, r = D - 1; l < r; ++l, --r) { const int t = p[l]; p[l] = p[r]; p[r] = t; }
            }
            for (int i = 0; i < D; ++i) rperm[i] = best[i];
        }
    }
    __syncthreads();
    int rp[D];
#pragma unroll
    for (int i = 0; i < D; ++i) rp[i] = rperm[i];
    // shift by row 0's value: keeps sum(t^2) - sum(t)^2/N well conditioned in Float64
    double sh0[D];
#pragma unroll
    for (int i = 0; i < D; ++i) sh0[i] = (double)(float)T[rp[i]];
    double mom[2 * D];
#pragma unroll
    for (int i = 0; i < 2 * D; ++i) mom[i] = 0;
#pragma unroll
    for (int r = 0; r < ISA_R; ++r) {
        const long long n = base + (long long)r * ISA_T;
        if (n < N) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double v = 0;
#pragma unroll
                for (int k = 0; k < D; ++k) v = (rp[i] == k) ? tv[r][k] : v;
                const float vf = (float)v;                 // T(target): back to Float32
                if (n >= a.out_first && n < a.out_first + a.out_count) a.target[(n - a.out_first) * D + i] = vf;
                const double c = (double)vf - sh0[i];
                mom[i] += c;
                mom[D + i] += c * c;
            }
        }
    }
    block_sum_multi<2 * D>(mom, shd);
    if (threadIdx.x < 2 * D) {
        double v = 0;
#pragma unroll
        for (int i = 0; i < 2 * D; ++i) v = (threadIdx.x == i) ? mom[i] : v;
        Pmom[(long long)blockIdx.x * 2 * D + threadIdx.x] = v;
    }
}

template <int D>
__global__ __launch_bounds__(64) void isa_weights_kernel(IsaArgs a, const double* __restrict__ Pmom, int G) {
    ISK_LATENCY_CHAIN_PRIO();
    __shared__ double stage[ISA_STAGE];
    __shared__ double mom[2 * D];
    sum_partials(Pmom, G, 2 * D, stage, mom);
    const int i = threadIdx.x;
    if (i >= D) return;
    const double s1 = mom[i], s2 = mom[D + i];
    const double N = (double)a.N;
    a.lossw[i] = (float)(1.0 / sqrt((s2 - s1 * s1 / N) / (N - 1.0)));
}

long long isk_isa_work_doubles(long long N, int d) {
    const long long G = (N + ISA_ROWS - 1) / ISA_ROWS;
    // X, T, two ping-pong argmax partial sets, cost / moment partials, index list
    return 2 * N * d + 2 * G * (2 + d) + G * d * d + G * 2 * d + 16;
}

template <int D>
static void launch_isa_d(const IsaArgs& a, hipStream_t st) {
    const long long N = a.N;
    const int G = (int)((N + ISA_ROWS - 1) / ISA_ROWS);
    double* p = a.work + 2 * N * D;
    double* P0 = p; p += (long long)G * (2 + D);
    double* P1 = p; p += (long long)G * (2 + D);
    double* Pc = p; p += (long long)G * D * D;
    double* Pm = p; p += (long long)G * 2 * D;
    int* ind = (int*)p;
    hipLaunchKernelGGL(isa_init_kernel<D>, dim3(G), dim3(ISA_T), 0, st, a, P0, Pc);
    if (a.whitening) hipLaunchKernelGGL(isa_whiten_kernel<D>, dim3(G), dim3(ISA_T), 0, st, a, Pc, G, P0);
    double* pin = P0;
    double* pout = P1;
    for (int j = 0; j < D - 1; ++j) {
        hipLaunchKernelGGL(isa_step_kernel<D>, dim3(G), dim3(ISA_T), 0, st, a, j, pin, G, pout, ind);
        double* t = pin; pin = pout; pout = t;
    }
    hipLaunchKernelGGL(isa_target_kernel<D>, dim3(G), dim3(ISA_T), 0, st, a, pin, G, ind, Pc);
    hipLaunchKernelGGL(isa_perm_kernel<D>, dim3(G), dim3(ISA_T), 0, st, a, Pc, G, Pm);
    hipLaunchKernelGGL(isa_weights_kernel<D>, dim3(1), dim3(64), 0, st, a, Pm, G);
}

void isk_launch_isa(const IsaArgs& a, hipStream_t st) {
    switch (a.d) {
        case 2: launch_isa_d<2>(a, st); break;
        case 3: launch_isa_d<3>(a, st); break;
        case 4: launch_isa_d<4>(a, st); break;
        case 5: launch_isa_d<5>(a, st); break;
        case 6: launch_isa_d<6>(a, st); break;
        case 7: launch_isa_d<7>(a, st); break;
        default: launch_isa_d<8>(a, st); break;
    }
}
