/* jtk_eigen.h -- symmetric eigendecomposition shared by the product (jtk_amd/csrc/correction.cpp) and the CPU oracle
 * (oracle/correction.c), the way jtk_math.h shares exp / log: the reference takes it from nalgebra
 * (`DMatrix::symmetric_eigen`, haplotyper/src/phmm_likelihood_correction.rs:418), which is not under /root/reference.
 * OWN SPECIFICATION: cyclic Jacobi, rows/columns swept in index order, rotations as in Golub & Van Loan 8.4; stops when
 * no off-diagonal entry exceeds 1e-14 x the largest diagonal magnitude (or after 64 sweeps).  Eigenvalue i is a[i][i],
 * eigenvector i is COLUMN i of v; the order is whatever the sweeps leave (the caller sorts).  Eigenvectors are defined
 * up to sign (and up to rotation inside a degenerate eigenspace); what the caller does with them -- a distance-based
 * k-means after per-column normalisation -- does not depend on the sign. */
#ifndef JTK_EIGEN_H
#define JTK_EIGEN_H
#include <stddef.h>

/* a: n x n row-major symmetric, destroyed (its diagonal holds the eigenvalues on return); v: n x n row-major output */
static inline void jtk_symmetric_eigen(double *a, size_t n, double *v) {
    for (size_t i = 0; i < n; i++)
        for (size_t j = 0; j < n; j++) v[i * n + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = 0.0, diag = 0.0;
        for (size_t i = 0; i < n; i++) {
            const double d = a[i * n + i] < 0 ? -a[i * n + i] : a[i * n + i];
            if (d > diag) diag = d;
            for (size_t j = i + 1; j < n; j++) {
                const double o = a[i * n + j] < 0 ? -a[i * n + j] : a[i * n + j];
                if (o > off) off = o;
            }
        }
        if (!(off > 1e-14 * diag)) break;
        for (size_t p = 0; p + 1 < n; p++)
            for (size_t q = p + 1; q < n; q++) {
                const double apq = a[p * n + q];
                if (apq == 0.0) continue;
                const double tau = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
                const double root = __builtin_sqrt(1.0 + tau * tau);
                const double t = tau >= 0.0 ? 1.0 / (tau + root) : -1.0 / (-tau + root);
                const double c = 1.0 / __builtin_sqrt(1.0 + t * t), s = t * c;
                for (size_t k = 0; k < n; k++) { /* A <- A J (columns p, q) */
                    const double akp = a[k * n + p], akq = a[k * n + q];
                    a[k * n + p] = c * akp - s * akq;
                    a[k * n + q] = s * akp + c * akq;
                }
                for (size_t k = 0; k < n; k++) { /* A <- J^T A (rows p, q) */
                    const double apk = a[p * n + k], aqk = a[q * n + k];
                    a[p * n + k] = c * apk - s * aqk;
                    a[q * n + k] = s * apk + c * aqk;
                }
                for (size_t k = 0; k < n; k++) { /* V <- V J */
                    const double vkp = v[k * n + p], vkq = v[k * n + q];
                    v[k * n + p] = c * vkp - s * vkq;
                    v[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
}
#endif
