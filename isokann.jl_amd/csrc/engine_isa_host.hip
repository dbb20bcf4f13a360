// Host twin of the device ISA chain (kernels_isa.hip): Float64 inner simplex, fixperm and loss weights on the CPU, for
// the host-driven data-parallel hooks (ragged shards, gloo groups) and isk_isa_solve.  src/isotarget.jl:81-127 of the
// reference; PCCAPlus.indexmap restated from the published algorithm.
#include "engine.hpp"
#include <limits>

// ---- host Float64 ISA (src/isotarget.jl:81-127; PCCAPlus.indexmap restated) -----------
static std::vector<int> indexmap_host(std::vector<double> X /* N x d row-major */, long long N, int d) {
    std::vector<int> ind(d);
    std::vector<double> rn(N);
    for (int j = 0; j < d; ++j) {
        long long best = 0;
        double bv = -1.0;
        for (long long n = 0; n < N; ++n) {
            double s = 0;
            for (int k = 0; k < d; ++k) s += X[n * d + k] * X[n * d + k];
            rn[n] = std::sqrt(s);
            if (rn[n] > bv) { bv = rn[n]; best = n; }      // first maximum, like Julia's argmax
        }
        ind[j] = (int)best;
        if (j == 0) {
            std::vector<double> x0(X.begin() + best * d, X.begin() + best * d + d);
            for (long long n = 0; n < N; ++n)
                for (int k = 0; k < d; ++k) X[n * d + k] -= x0[k];
        } else {
            const double inv = 1.0 / rn[best];
            for (auto& v : X) v *= inv;
            std::vector<double> vt(X.begin() + best * d, X.begin() + best * d + d);
            for (long long n = 0; n < N; ++n) {
                double dot = 0;
                for (int k = 0; k < d; ++k) dot += X[n * d + k] * vt[k];
                for (int k = 0; k < d; ++k) X[n * d + k] -= dot * vt[k];
            }
        }
    }
    return ind;
}

// Gauss-Jordan inverse with partial pivoting; false if singular / non-finite
static bool invert_host(std::vector<double>& A, int d) {
    std::vector<double> M(d * 2 * d, 0.0);
    for (int i = 0; i < d; ++i) {
        for (int j = 0; j < d; ++j) M[i * 2 * d + j] = A[i * d + j];
        M[i * 2 * d + d + i] = 1.0;
    }
    // multiplier against the unscaled pivot row (as LU does): exactly dependent rows cancel to exact
    // zeros and are reported singular, like Julia's inv()
    for (int c = 0; c < d; ++c) {
        int p = c;
        for (int r = c + 1; r < d; ++r)
            if (std::fabs(M[r * 2 * d + c]) > std::fabs(M[p * 2 * d + c])) p = r;
        const double pv = M[p * 2 * d + c];
        if (!(std::fabs(pv) > 0.0) || !std::isfinite(pv)) return false;
        if (p != c)
            for (int j = 0; j < 2 * d; ++j) std::swap(M[p * 2 * d + j], M[c * 2 * d + j]);
        for (int r = 0; r < d; ++r) {
            if (r == c) continue;
            const double f = M[r * 2 * d + c] / pv;
            if (f == 0.0) continue;
            for (int j = 0; j < 2 * d; ++j) M[r * 2 * d + j] -= f * M[c * 2 * d + j];
            M[r * 2 * d + c] = 0.0;
        }
    }
    for (int c = 0; c < d; ++c) {
        const double inv = 1.0 / M[c * 2 * d + c];
        for (int j = 0; j < 2 * d; ++j) M[c * 2 * d + j] *= inv;
    }
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            A[i * d + j] = M[i * 2 * d + d + j];
            if (!std::isfinite(A[i * d + j])) return false;
        }
    return true;
}

// symmetric C -> C^(-1/2) by cyclic Jacobi (whitening option, src/isotarget.jl:85-89)
static bool inv_sqrt_sym(std::vector<double>& C, int d) {
    std::vector<double> V(d * d, 0.0), A = C;
    for (int i = 0; i < d; ++i) V[i * d + i] = 1.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0;
        for (int i = 0; i < d; ++i)
            for (int j = i + 1; j < d; ++j) off += A[i * d + j] * A[i * d + j];
        if (off < 1e-300) break;
        for (int p = 0; p < d; ++p)
            for (int q = p + 1; q < d; ++q) {
                if (std::fabs(A[p * d + q]) < 1e-300) continue;
                const double theta = (A[q * d + q] - A[p * d + p]) / (2 * A[p * d + q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
                const double c = 1 / std::sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < d; ++k) {
                    const double akp = A[k * d + p], akq = A[k * d + q];
                    A[k * d + p] = c * akp - s * akq; A[k * d + q] = s * akp + c * akq;
                }
                for (int k = 0; k < d; ++k) {
                    const double apk = A[p * d + k], aqk = A[q * d + k];
                    A[p * d + k] = c * apk - s * aqk; A[q * d + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < d; ++k) {
                    const double vkp = V[k * d + p], vkq = V[k * d + q];
                    V[k * d + p] = c * vkp - s * vkq; V[k * d + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < d; ++i)
        if (!(A[i * d + i] > 0)) return false;
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            double s = 0;
            for (int k = 0; k < d; ++k) s += V[i * d + k] * V[j * d + k] / std::sqrt(A[k * d + k]);
            C[i * d + j] = s;
        }
    return true;
}

// fixperm for larger systems: argmin over row permutations p of sum_i Cst[p_i * d + i] as a linear assignment problem -- what the
// reference's own TODO names ("use the hungarian algorithm for larger systems", src/isotarget.jl:121); its brute force is d! steps.
// Shortest augmenting paths with row / column potentials (Kuhn-Munkres in its O(d^3) form), Float64.  Where the minimiser is
// unique -- always, up to exact ties of Float64 sums -- it is the brute force's.
std::vector<int> assign_min_cost(const std::vector<double>& Cst, int d) {
    const double INF = std::numeric_limits<double>::infinity();
    std::vector<double> u(d + 1, 0.0), v(d + 1, 0.0);
    std::vector<int> match(d + 1, 0), way(d + 1, 0);          // match[a'] = chi row assigned to target row a' (1-based, 0 = none)
    for (int i = 1; i <= d; ++i) {                           // rows: chi row i - 1; columns: target row a - 1; cost = Cst[(a-1) d + (i-1)]
        match[0] = i;
        int j0 = 0;
        std::vector<double> minv(d + 1, INF);
        std::vector<char> used(d + 1, 0);
        do {
            used[j0] = 1;
            const int i0 = match[j0];
            double delta = INF;
            int j1 = 0;
            for (int j = 1; j <= d; ++j) {
                if (used[j]) continue;
                const double cur = Cst[(size_t)(j - 1) * d + (i0 - 1)] - u[i0] - v[j];
                if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
                if (minv[j] < delta) { delta = minv[j]; j1 = j; }
            }
            if (!(delta < INF)) { j1 = 0; for (int j = 1; j <= d; ++j) if (!used[j]) { j1 = j; break; } delta = 0; }   // non-finite costs: any completion
            for (int j = 0; j <= d; ++j) {
                if (used[j]) { u[match[j]] += delta; v[j] -= delta; }
                else minv[j] -= delta;
            }
            j0 = j1;
        } while (match[j0] != 0);
        do { const int j1 = way[j0]; match[j0] = match[j1]; j0 = j1; } while (j0);
    }
    std::vector<int> p(d);
    for (int a = 1; a <= d; ++a) p[match[a] - 1] = a - 1;    // p[i] = target row that lands on chi row i
    return p;
}

// ks, chi: [N][d] (fp32, as downloaded).  Produces At (A' row-major: target = At * ks[n]) and the row permutation.
int isa_solve_host(const isk_engine* e, const std::vector<float>& ks, const std::vector<float>& chi, long long N, int d,
                          std::vector<double>& At, std::vector<int>& rowperm, std::vector<double>* target_out) {
    for (float v : ks)
        if (!std::isfinite(v)) return fail(ISK_ESINGULAR, isk_status_string(ISK_ESINGULAR));
    std::vector<double> X(ks.begin(), ks.end());           // X = Float64.(ks')  (N x d)
    std::vector<int> ind;
    if (e->cfg.isa_whitening) {
        std::vector<double> C(d * d, 0.0);
        for (long long n = 0; n < N; ++n)
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j) C[i * d + j] += X[n * d + i] * X[n * d + j];
        for (auto& v : C) v /= (double)N;
        if (!inv_sqrt_sym(C, d)) return fail(ISK_ESINGULAR, isk_status_string(ISK_ESINGULAR));
        std::vector<double> XW((size_t)N * d);
        for (long long n = 0; n < N; ++n)
            for (int j = 0; j < d; ++j) {
                double s = 0;
                for (int k = 0; k < d; ++k) s += X[n * d + k] * C[k * d + j];
                XW[n * d + j] = s;
            }
        ind = indexmap_host(XW, N, d);
    } else {
        ind = indexmap_host(X, N, d);
    }
    std::vector<double> A(d * d);
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) A[i * d + j] = X[(long long)ind[i] * d + j];
    if (!invert_host(A, d)) return fail(ISK_ESINGULAR, isk_status_string(ISK_ESINGULAR));
    // target = A' * ks  ->  target[n][a] = sum_b A[b][a] ks[n][b]
    At.assign(d * d, 0.0);
    for (int a = 0; a < d; ++a)
        for (int b = 0; b < d; ++b) At[a * d + b] = A[b * d + a];
    rowperm.resize(d);
    std::iota(rowperm.begin(), rowperm.end(), 0);
    std::vector<double> tgt((size_t)N * d);
    for (long long n = 0; n < N; ++n)
        for (int a = 0; a < d; ++a) {
            double s = 0;
            for (int b = 0; b < d; ++b) s += At[a * d + b] * X[n * d + b];
            tgt[n * d + a] = s;
        }
    if (e->cfg.isa_permute) {
        // fixperm (src/isotarget.jl:120-127): cost(p) = sum_i sum_n |new[p_i, n] - old[i, n]|
        std::vector<double> Cst(d * d, 0.0);
        for (long long n = 0; n < N; ++n)
            for (int a = 0; a < d; ++a)
                for (int i = 0; i < d; ++i) Cst[a * d + i] += std::fabs(tgt[n * d + a] - (double)chi[n * d + i]);
        if (d <= ISK_ISA_DEV_MAXD) {
            // the reference's search itself (and the device chain's): every permutation in lexicographic order, first minimiser
            std::vector<int> p(d), best(d);
            std::iota(p.begin(), p.end(), 0);
            double bc = INFINITY;
            bool first = true;
            do {
                double c = 0;
                for (int i = 0; i < d; ++i) c += Cst[p[i] * d + i];
                if (first || c < bc) { bc = c; best = p; first = false; }
            } while (std::next_permutation(p.begin(), p.end()));
            rowperm = best;
        } else {
            rowperm = assign_min_cost(Cst, d);
        }
    }
    if (target_out) {
        target_out->resize((size_t)N * d);
        for (long long n = 0; n < N; ++n)
            for (int i = 0; i < d; ++i) (*target_out)[n * d + i] = tgt[n * d + rowperm[i]];
    }
    return ISK_OK;
}

// loss weights 1 ./ std(target, dims=2) (corrected), src/iso.jl:183
void loss_weights_host(const std::vector<float>& tgt, long long N, int d, std::vector<float>& w) {
    w.assign(d, 1.0f);
    if (d <= 1) return;
    for (int i = 0; i < d; ++i) {
        double mean = 0;
        for (long long n = 0; n < N; ++n) mean += tgt[n * d + i];
        mean /= (double)N;
        double ss = 0;
        for (long long n = 0; n < N; ++n) { const double t = tgt[n * d + i] - mean; ss += t * t; }
        w[i] = (float)(1.0 / std::sqrt(ss / (double)(N - 1)));
    }
}

extern "C" int isk_isa_solve(const float* ks, const float* chi, int64_t N, int32_t d, int32_t permute, int32_t whitening,
                             double* At, int32_t* rowperm) {
    if (!ks || !chi || !At || !rowperm || N < 1 || d < 2 || d > ISK_ISA_HOST_MAXD) return fail(ISK_EINVAL, "bad argument");
    isk_engine tmp;
    tmp.cfg.isa_permute = permute; tmp.cfg.isa_whitening = whitening;
    std::vector<float> k(ks, ks + (size_t)N * d), c(chi, chi + (size_t)N * d);
    std::vector<double> A;
    std::vector<int> rp;
    CHK(isa_solve_host(&tmp, k, c, N, d, A, rp, nullptr));
    for (int i = 0; i < d * d; ++i) At[i] = A[i];
    for (int i = 0; i < d; ++i) rowperm[i] = rp[i];
    return ISK_OK;
}
