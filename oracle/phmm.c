/* phmm.c -- CPU ORACLE (test infrastructure).  OWN SPECIFICATION of the pair-HMM routines the reference
 * takes from the third-party crate kiley 0.3.0 @34ebbda0cb358335e22e20b054d357ea34d8326d
 * (Cargo.lock:452-454), whose source is NOT under /root/reference and for which the reference holds no
 * test or golden vector: **parity with kiley is unpinned**.  Call sites this stands in for:
 *   modification_table_antidiagonal      pseudo_mcmc.rs:62-63
 *   polish_until_converge_antidiagonal   local_clustering/mod.rs:105-106,154-156
 *   likelihood_antidiagonal_bootstrap    likelihood_gains.rs:27-28,282-283,301-302
 *   Generate::gen, gen_seq::generate_seq likelihood_gains.rs:17,26,270-271,281,300
 * The HIP kernels (jtk_amd/csrc/phmm_kernels.hip) implement exactly this specification; DESIGN.md
 * "Pair-HMM specification" is the prose version.
 *
 * Model.  States Match/Ins/Del.  Ins consumes a read base, Del consumes a template base (as in
 * definitions/src/lib.rs:816-823).  F_S(i,j): probability of read[0..j) and a path that has consumed i
 * template bases, j read bases and is in state S.  Start: F_M(0,0)=1.  End: lk = F_M+F_I+F_D at (L,n).
 *   F_M(i,j) = eM[x[i-1]][y[j-1]] * toM(i-1,j-1)      toM = F_M*a_MM + F_I*a_IM + F_D*a_DM
 *   F_I(i,j) = eI[ctx(j)][y[j-1]] * toI(i,j-1)        toI = F_M*a_MI + F_I*a_II + F_D*a_DI
 *   F_D(i,j) =                       toD(i-1,j)        toD = F_M*a_MD + F_I*a_ID + F_D*a_DD
 * with ctx(j) = y[j-2] (previous READ base) or 4 when j == 1, and every `to*` evaluated as
 * fma(F_D, a_D*, fma(F_I, a_I*, F_M * a_M*)).
 * Band.  Anti-diagonal t = i+j.  The ops path gives a centre c[t] (the i-coordinate of the path on
 * anti-diagonal t; a Match step visits c = i+1 on both of its two anti-diagonals); cells with
 * |i - c[t]| <= radius are filled, everything else is 0.
 * Scaling.  Values are kept as x * 2^-E with one integer exponent E per 64-anti-diagonal block
 * (re-normalised by the exact power of two that brings the block's first diagonal's maximum into [1,2)),
 * so scaling never changes a mantissa.
 * Modification table.  With b_S(i,j) the backward quantity (probability of read[j..) given state S at
 * (i,j)), hatM = eM*b_M, every edited-template likelihood is one "row crossing"
 *   V(i1,i2,base) = sum_j toM(i1,j)*eM[base][y[j]]*b_M(i2,j+1) + toD(i1,j)*b_D(i2,j)
 * (sub b@p: (p,p+1,b); ins b@p: (p,p,b); copy c@p: (p+c,p+1,x[p]); del d@p: (p,p+d+1,x[p+d])), summed over
 * j in DESCENDING order (the order a backward sweep meets them), M-term before D-term.
 */
#include <stdlib.h>
#include <string.h>

#include "jtk_math.h"
#include "jtk_oracle.h"

#define NUM_ROW JTK_NUM_ROW
#define LN2 0.6931471805599453094
#define MIN_GAIN 0.1
#define MAX_POLISH_ROUNDS 20

static inline int base_code(uint8_t c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 0;
    }
}

static inline double pow2i(int e) { return jtk_scalbn(1.0, e); }

int jo_band_centers(const uint8_t *ops, size_t n_ops, size_t tl, size_t rl, uint32_t *c) {
    size_t i = 0, j = 0, t = 0;
    c[0] = 0;
    for (size_t k = 0; k < n_ops; k++) {
        switch (ops[k]) {
            case JTK_OP_MATCH:
            case JTK_OP_MISMATCH:
                if (t + 2 > tl + rl) return -1;
                c[t + 1] = (uint32_t)(i + 1);
                c[t + 2] = (uint32_t)(i + 1);
                i++;
                j++;
                t += 2;
                break;
            case JTK_OP_DEL:
                if (t + 1 > tl + rl) return -1;
                c[t + 1] = (uint32_t)(i + 1);
                i++;
                t++;
                break;
            case JTK_OP_INS:
                if (t + 1 > tl + rl) return -1;
                c[t + 1] = (uint32_t)i;
                j++;
                t++;
                break;
            default:
                return -1;
        }
    }
    return (i == tl && j == rl) ? 0 : -1;
}

typedef struct fwd_tab {
    size_t L, n, T, W;
    int r;
    uint32_t *c;   /* T+1 centres */
    double *toM;   /* (T+1) x W, scaled */
    double *toD;
    int *E;        /* T+1 cumulative exponents: true = scaled * 2^E[t] */
    uint8_t *x, *y; /* base codes */
    double lk;
} fwd_tab_t;

static inline double tab_get(const double *a, const fwd_tab_t *f, int64_t t, int64_t i) {
    if (t < 0 || t > (int64_t)f->T) return 0.0;
    int64_t w = i - ((int64_t)f->c[t] - f->r);
    if (w < 0 || w >= (int64_t)f->W) return 0.0;
    return a[(size_t)t * f->W + (size_t)w];
}

static void fwd_free(fwd_tab_t *f) {
    free(f->c);
    free(f->toM);
    free(f->toD);
    free(f->E);
    free(f->x);
    free(f->y);
}

/* forward sweep; keep_tables == 0 keeps only what the likelihood needs (still allocates full tables:
 * this is the checker, not the product) */
static int forward(const jtk_hmm_t *h, const uint8_t *tmpl, size_t L, const uint8_t *read, size_t n,
                   const uint8_t *ops, size_t n_ops, size_t radius, fwd_tab_t *f) {
    memset(f, 0, sizeof *f);
    f->L = L;
    f->n = n;
    f->T = L + n;
    f->r = (int)radius;
    f->W = 2 * radius + 1;
    f->c = (uint32_t *)malloc((f->T + 1) * sizeof(uint32_t));
    f->x = (uint8_t *)malloc(L + 1);
    f->y = (uint8_t *)malloc(n + 1);
    for (size_t i = 0; i < L; i++) f->x[i] = (uint8_t)base_code(tmpl[i]);
    for (size_t j = 0; j < n; j++) f->y[j] = (uint8_t)base_code(read[j]);
    if (jo_band_centers(ops, n_ops, L, n, f->c) != 0) {
        f->lk = JO_LOG_ZERO;
        return -1;
    }
    size_t W = f->W, T = f->T;
    f->toM = (double *)calloc((T + 1) * W, sizeof(double));
    f->toD = (double *)calloc((T + 1) * W, sizeof(double));
    double *toI = (double *)calloc((T + 1) * W, sizeof(double));
    f->E = (int *)calloc(T + 1, sizeof(int));
    double *FM = (double *)malloc(W * sizeof(double)), *FI = (double *)malloc(W * sizeof(double)),
           *FD = (double *)malloc(W * sizeof(double));
    double endM = 0, endI = 0, endD = 0;
    for (size_t t = 0; t <= T; t++) {
        int Eprev1 = t >= 1 ? f->E[t - 1] : 0, Eprev2 = t >= 2 ? f->E[t - 2] : 0;
        /* raw values of diagonal t are produced in the scale of diagonal t-1 */
        int Ecur = Eprev1;
        double s2 = Eprev2 == Ecur ? 1.0 : pow2i(Eprev2 - Ecur); /* re-express diagonal t-2 (previous block at most) */
        double m = 0.0;
        for (size_t w = 0; w < W; w++) {
            int64_t i = (int64_t)f->c[t] - f->r + (int64_t)w, j = (int64_t)t - i;
            double fm = 0, fi = 0, fd = 0;
            if (i >= 0 && i <= (int64_t)L && j >= 0 && j <= (int64_t)n) {
                if (t == 0) {
                    fm = 1.0;
                } else {
                    if (i >= 1 && j >= 1)
                        fm = h->mat_emit[4 * f->x[i - 1] + f->y[j - 1]] *
                             (tab_get(f->toM, f, (int64_t)t - 2, i - 1) * s2);
                    if (j >= 1) {
                        int ctx = j >= 2 ? f->y[j - 2] : 4;
                        fi = h->ins_emit[4 * ctx + f->y[j - 1]] * tab_get(toI, f, (int64_t)t - 1, i);
                    }
                    if (i >= 1) fd = tab_get(f->toD, f, (int64_t)t - 1, i - 1);
                }
            }
            FM[w] = fm;
            FI[w] = fi;
            FD[w] = fd;
            if (fm > m) m = fm;
            if (fi > m) m = fi;
            if (fd > m) m = fd;
        }
        if (t > 0 && (t & (JO_SCALE_BLOCK - 1)) == 0 && m > 0.0) {
            int e = jtk_ilogb_pos(m);
            double s = pow2i(-e);
            for (size_t w = 0; w < W; w++) {
                FM[w] *= s;
                FI[w] *= s;
                FD[w] *= s;
            }
            Ecur += e;
        }
        f->E[t] = Ecur;
        for (size_t w = 0; w < W; w++) {
            double fm = FM[w], fi = FI[w], fd = FD[w];
            f->toM[t * W + w] = __builtin_fma(fd, h->del_mat, __builtin_fma(fi, h->ins_mat, fm * h->mat_mat));
            toI[t * W + w] = __builtin_fma(fd, h->del_ins, __builtin_fma(fi, h->ins_ins, fm * h->mat_ins));
            f->toD[t * W + w] = __builtin_fma(fd, h->del_del, __builtin_fma(fi, h->ins_del, fm * h->mat_del));
        }
        if (t == T) {
            endM = FM[radius];
            endI = FI[radius];
            endD = FD[radius]; /* c[T] == L, so w == radius is cell (L,n) */
        }
    }
    double tot = (endM + endI) + endD;
    f->lk = tot > 0.0 ? jtk_log(tot) + (double)f->E[T] * LN2 : JO_LOG_ZERO;
    free(toI);
    free(FM);
    free(FI);
    free(FD);
    return 0;
}

double jo_phmm_likelihood(const jtk_hmm_t *hmm, const uint8_t *tmpl, size_t tl, const uint8_t *read,
                          size_t rl, const uint8_t *ops, size_t n_ops, size_t radius) {
    fwd_tab_t f;
    forward(hmm, tmpl, tl, read, rl, ops, n_ops, radius, &f);
    double lk = f.lk;
    fwd_free(&f);
    return lk;
}

/* accumulator slots per template row */
enum { A_SUB0 = 0, A_SUBD = 4, A_INS0 = 5, A_INSD = 9, A_COPY = 10, A_DEL = 13, A_N = 16 };

static void finalize_row(const jtk_hmm_t *h, int64_t iota, size_t L, const double *acc, int G,
                         double *table) {
    /* row iota owns: sub[iota-1], ins[iota], copy_c[iota-1], del_d[iota-d-1] */
    const double gl = (double)G * LN2;
    const double *a = acc + (size_t)iota * A_N;
    if (iota >= 1) {
        size_t p = (size_t)iota - 1;
        for (int b = 0; b < 4; b++) {
            double v = h->mat_emit[4 * b + 0] * a[A_SUB0 + 0];
            v = __builtin_fma(h->mat_emit[4 * b + 1], a[A_SUB0 + 1], v);
            v = __builtin_fma(h->mat_emit[4 * b + 2], a[A_SUB0 + 2], v);
            v = __builtin_fma(h->mat_emit[4 * b + 3], a[A_SUB0 + 3], v);
            v = v + a[A_SUBD];
            table[p * NUM_ROW + b] = v > 0.0 ? jtk_log(v) + gl : JO_LOG_ZERO;
        }
        for (int c = 1; c <= 3; c++) {
            double v = a[A_COPY + c - 1];
            table[p * NUM_ROW + 8 + (c - 1)] = v > 0.0 ? jtk_log(v) + gl : JO_LOG_ZERO;
        }
        for (int d = 1; d <= 3; d++) {
            if (iota - d - 1 < 0) continue;
            size_t q = (size_t)(iota - d - 1);
            double v = a[A_DEL + d - 1];
            table[q * NUM_ROW + 11 + (d - 1)] = v > 0.0 ? jtk_log(v) + gl : JO_LOG_ZERO;
        }
    }
    {
        size_t p = (size_t)iota;
        for (int b = 0; b < 4; b++) {
            double v = h->mat_emit[4 * b + 0] * a[A_INS0 + 0];
            v = __builtin_fma(h->mat_emit[4 * b + 1], a[A_INS0 + 1], v);
            v = __builtin_fma(h->mat_emit[4 * b + 2], a[A_INS0 + 2], v);
            v = __builtin_fma(h->mat_emit[4 * b + 3], a[A_INS0 + 3], v);
            v = v + a[A_INSD];
            table[p * NUM_ROW + 4 + b] = v > 0.0 ? jtk_log(v) + gl : JO_LOG_ZERO;
        }
    }
    (void)L;
}

double jo_phmm_modification_table(const jtk_hmm_t *h, const uint8_t *tmpl, size_t L, const uint8_t *read,
                                  size_t n, const uint8_t *ops, size_t n_ops, size_t radius,
                                  double *table) {
    size_t cols = NUM_ROW * (L + 1);
    for (size_t p = 0; p < cols; p++) table[p] = JO_LOG_ZERO;
    fwd_tab_t f;
    if (forward(h, tmpl, L, read, n, ops, n_ops, radius, &f) != 0 || !(f.lk > JO_LOG_ZERO)) {
        double lk = f.lk;
        fwd_free(&f);
        return lk;
    }
    const size_t W = f.W, T = f.T;
    const int r = f.r;
    /* backward diagonals t+1, t+2 kept (scaled with EB of their own diagonal) */
    double *bM = (double *)calloc(3 * W, sizeof(double)), *bD = (double *)calloc(3 * W, sizeof(double)),
           *hM = (double *)calloc(3 * W, sizeof(double)), *hI = (double *)calloc(3 * W, sizeof(double));
    int *EB = (int *)calloc(T + 3, sizeof(int));
    double *acc = (double *)calloc((L + 2) * A_N, sizeof(double));
    double *cbI_buf = (double *)malloc(W * sizeof(double));
    /* backward band accessor for diagonal tt stored in slot tt % 3 */
#define BGET(arr, tt, ii)                                                                        \
    (((tt) > (int64_t)T || (ii) < (int64_t)f.c[tt] - r || (ii) > (int64_t)f.c[tt] + r)            \
         ? 0.0                                                                                   \
         : (arr)[((size_t)(tt) % 3) * W + (size_t)((ii) - ((int64_t)f.c[tt] - r))])
    int Gprev = 0;
    int64_t live_hi = (int64_t)L; /* rows > live_hi are finalized */
    for (int64_t t = (int64_t)T; t >= 0; t--) {
        int64_t lo_i = (int64_t)f.c[t] - r, hi_i = (int64_t)f.c[t] + r;
        /* (0) rows that left the band are final, in the exponent of the previous step */
        while (live_hi > hi_i) {
            finalize_row(h, live_hi, L, acc, Gprev, table);
            live_hi--;
        }
        /* (1) backward values of diagonal t, produced in the scale of diagonal t+1 */
        int Ecur = t < (int64_t)T ? EB[t + 1] : 0;
        int E1 = t + 1 <= (int64_t)T ? EB[t + 1] : 0, E2 = t + 2 <= (int64_t)T ? EB[t + 2] : 0;
        double s2 = E2 == Ecur ? 1.0 : pow2i(E2 - Ecur);
        (void)E1;
        double *cbM = bM + ((size_t)t % 3) * W, *cbD = bD + ((size_t)t % 3) * W,
               *chM = hM + ((size_t)t % 3) * W, *chI = hI + ((size_t)t % 3) * W;
        double *cbI = cbI_buf;
        double m = 0.0;
        for (size_t w = 0; w < W; w++) {
            int64_t i = lo_i + (int64_t)w, j = t - i;
            double vm = 0, vi = 0, vd = 0;
            if (i >= 0 && i <= (int64_t)L && j >= 0 && j <= (int64_t)n) {
                if (t == (int64_t)T) {
                    vm = vi = vd = 1.0;
                } else {
                    double xm = BGET(hM, t + 2, i + 1) * s2; /* hatM(i+1,j+1) */
                    double xi = BGET(hI, t + 1, i);          /* hatI(i,j+1)   */
                    double xd = BGET(bD, t + 1, i + 1);      /* b_D(i+1,j)    */
                    vm = __builtin_fma(h->mat_del, xd, __builtin_fma(h->mat_ins, xi, h->mat_mat * xm));
                    vi = __builtin_fma(h->ins_del, xd, __builtin_fma(h->ins_ins, xi, h->ins_mat * xm));
                    vd = __builtin_fma(h->del_del, xd, __builtin_fma(h->del_ins, xi, h->del_mat * xm));
                }
            }
            cbM[w] = vm;
            cbI[w] = vi;
            cbD[w] = vd;
            if (vm > m) m = vm;
            if (vi > m) m = vi;
            if (vd > m) m = vd;
        }
        if (t < (int64_t)T && (t & (JO_SCALE_BLOCK - 1)) == JO_SCALE_BLOCK - 1 && m > 0.0) {
            int e = jtk_ilogb_pos(m);
            double s = pow2i(-e);
            for (size_t w = 0; w < W; w++) {
                cbM[w] *= s;
                cbI[w] *= s;
                cbD[w] *= s;
            }
            Ecur += e;
        }
        EB[t] = Ecur;
        for (size_t w = 0; w < W; w++) {
            int64_t i = lo_i + (int64_t)w, j = t - i;
            double a = 0, b = 0;
            if (i >= 1 && i <= (int64_t)L && j >= 1 && j <= (int64_t)n)
                a = h->mat_emit[4 * f.x[i - 1] + f.y[j - 1]] * cbM[w];
            if (i >= 0 && i <= (int64_t)L && j >= 1 && j <= (int64_t)n) {
                int ctx = j >= 2 ? f.y[j - 2] : 4;
                b = h->ins_emit[4 * ctx + f.y[j - 1]] * cbI[w];
            }
            chM[w] = a;
            chI[w] = b;
        }
        /* (2) common exponent of this step; rescale the live accumulators when it changes */
        int G = f.E[t] + EB[t];
        if (t < (int64_t)T && G != Gprev) {
            double s = pow2i(Gprev - G);
            int64_t from = lo_i < 0 ? 0 : lo_i;
            for (int64_t ii = from; ii <= live_hi; ii++)
                for (int k = 0; k < A_N; k++) acc[(size_t)ii * A_N + k] *= s;
        }
        Gprev = G;
        /* (3) accumulate the terms of every band cell (row iota = i, column j2 = j) */
        double fsc[8]; /* 2^(E_F[tt]-E_F[t]) for tt = t-5 .. t+2: exact re-expression of a source diagonal */
        for (int q = 0; q < 8; q++) {
            const int64_t tt = t - 5 + q;
            const int de = (tt >= 0 && tt <= (int64_t)T) ? f.E[tt] - f.E[t] : 0;
            fsc[q] = de == 0 ? 1.0 : pow2i(de);
        }
        for (size_t w = 0; w < W; w++) {
            int64_t i = lo_i + (int64_t)w, j2 = t - i;
            if (i < 0 || i > (int64_t)L || j2 < 0 || j2 > (int64_t)n) continue;
            double *a = acc + (size_t)i * A_N;
            double vM = cbM[w], vD = cbD[w], vH = chM[w];
#define FSC(tt) fsc[(tt) - t + 5]
            /* sub (entry i-1): M toM(i-1,j2-1) diag t-2 ; D toD(i-1,j2) diag t-1 */
            if (i >= 1) {
                if (j2 >= 1) {
                    int cy = f.y[j2 - 1];
                    double fm = tab_get(f.toM, &f, t - 2, i - 1) * FSC(t - 2);
                    a[A_SUB0 + cy] = __builtin_fma(fm, vM, a[A_SUB0 + cy]);
                }
                double fd = tab_get(f.toD, &f, t - 1, i - 1) * FSC(t - 1);
                a[A_SUBD] = __builtin_fma(fd, vD, a[A_SUBD]);
            }
            /* ins (entry i): M toM(i,j2-1) diag t-1 ; D toD(i,j2) diag t */
            if (j2 >= 1) {
                int cy = f.y[j2 - 1];
                double fm = tab_get(f.toM, &f, t - 1, i) * FSC(t - 1);
                a[A_INS0 + cy] = __builtin_fma(fm, vM, a[A_INS0 + cy]);
            }
            {
                double fd = tab_get(f.toD, &f, t, i);
                a[A_INSD] = __builtin_fma(fd, vD, a[A_INSD]);
            }
            if (i >= 1) {
                /* copy c (entry i-1): M toM(i-1+c,j2-1) diag t+c-2 ; D toD(i-1+c,j2) diag t+c-1 */
                for (int c = 1; c <= 3; c++) {
                    double v = a[A_COPY + c - 1];
                    if (j2 >= 1) {
                        double fm = tab_get(f.toM, &f, t + c - 2, i - 1 + c) * FSC(t + c - 2);
                        v = __builtin_fma(fm, vH, v);
                    }
                    double fd = tab_get(f.toD, &f, t + c - 1, i - 1 + c) * FSC(t + c - 1);
                    v = __builtin_fma(fd, vD, v);
                    a[A_COPY + c - 1] = v;
                }
                /* del d (entry i-d-1): M toM(i-d-1,j2-1) diag t-d-2 ; D toD(i-d-1,j2) diag t-d-1 */
                for (int d = 1; d <= 3; d++) {
                    if (i - d - 1 < 0) continue;
                    double v = a[A_DEL + d - 1];
                    if (j2 >= 1) {
                        double fm = tab_get(f.toM, &f, t - d - 2, i - d - 1) * FSC(t - d - 2);
                        v = __builtin_fma(fm, vH, v);
                    }
                    double fd = tab_get(f.toD, &f, t - d - 1, i - d - 1) * FSC(t - d - 1);
                    v = __builtin_fma(fd, vD, v);
                    a[A_DEL + d - 1] = v;
                }
            }
        }
    }
    while (live_hi >= 0) {
        finalize_row(h, live_hi, L, acc, Gprev, table);
        live_hi--;
    }
#undef BGET
#undef FSC
    double lk = f.lk;
    free(bM);
    free(bD);
    free(hM);
    free(hI);
    free(EB);
    free(acc);
    free(cbI_buf);
    fwd_free(&f);
    return lk;
}

/* ------------------------------------------------------------------------------------------------
 * Polishing (own spec of kiley polish_until_converge_antidiagonal).
 * Round t: total[p][row] = sum over the first take_num reads, in read order, of (table_r - lk_r).
 * Scan p = ignore_edge .. L-ignore_edge-1 left to right; at p take the FIRST best row; if its total
 * exceeds MIN_GAIN apply it and skip the bases it touches plus inactive(t) = 5 + (5t mod 21) further
 * positions.  Stop when a round applies nothing or after MAX_POLISH_ROUNDS rounds.
 * Ops are re-threaded locally: a deleted template base turns its Match/Mismatch column into Ins and drops
 * its Del column; inserted template bases become Del columns placed right after the column that consumed
 * the preceding template base; Match/Mismatch tags are then recomputed from the bases.
 */
typedef struct edit {
    size_t pos;
    int row;
} edit_t;

static size_t select_edits(const double *total, size_t L, size_t ignore_edge, size_t inactive,
                           edit_t *edits) {
    size_t ne = 0;
    size_t pos = ignore_edge;
    while (pos + ignore_edge < L) {
        int best = 0;
        double g = total[pos * NUM_ROW];
        for (int row = 1; row < NUM_ROW; row++)
            if (total[pos * NUM_ROW + row] > g) {
                g = total[pos * NUM_ROW + row];
                best = row;
            }
        if (g > MIN_GAIN) {
            edits[ne].pos = pos;
            edits[ne].row = best;
            ne++;
            size_t span = best >= 11 ? (size_t)(best - 10) : 1;
            pos += span + inactive;
        } else {
            pos++;
        }
    }
    return ne;
}

static size_t apply_edits_template(const uint8_t *tmpl, size_t L, const edit_t *edits, size_t ne,
                                   uint8_t *out) {
    size_t w = 0, e = 0, p = 0;
    while (p < L) {
        if (e < ne && edits[e].pos == p) {
            int row = edits[e].row;
            e++;
            if (row < 4) {
                out[w++] = (uint8_t)"ACGT"[row];
                p++;
            } else if (row < 8) {
                out[w++] = (uint8_t)"ACGT"[row - 4];
                out[w++] = tmpl[p++];
            } else if (row < 11) {
                size_t c = (size_t)(row - 7);
                for (size_t q = 0; q < c && p + q < L; q++) out[w++] = tmpl[p + q];
                out[w++] = tmpl[p++];
            } else {
                p += (size_t)(row - 10);
            }
        } else {
            out[w++] = tmpl[p++];
        }
    }
    return w;
}

/* number of template bases an insertion-type edit adds */
static size_t edit_inserted(const edit_t *e, size_t L) {
    if (e->row >= 4 && e->row < 8) return 1;
    if (e->row >= 8 && e->row < 11) {
        size_t c = (size_t)(e->row - 7);
        return e->pos + c <= L ? c : L - e->pos;
    }
    return 0;
}

static size_t rethread_ops(const uint8_t *ops, size_t n_ops, size_t L, const edit_t *edits, size_t ne,
                           uint8_t *out) {
    size_t w = 0, e = 0, ti = 0;
#define PENDING_INSERTS()                                                         \
    while (e < ne && edits[e].pos == ti && edits[e].row >= 4 && edits[e].row < 11) { \
        size_t k = edit_inserted(&edits[e], L);                                    \
        for (size_t q = 0; q < k; q++) out[w++] = JTK_OP_DEL;                     \
        e++;                                                                      \
    }
    PENDING_INSERTS();
    for (size_t k = 0; k < n_ops; k++) {
        uint8_t op = ops[k];
        if (op == JTK_OP_INS) {
            out[w++] = op;
            continue;
        }
        /* consumes template base ti */
        if (e < ne && edits[e].row >= 11 && edits[e].pos <= ti && ti < edits[e].pos + (size_t)(edits[e].row - 10)) {
            if (op != JTK_OP_DEL) out[w++] = JTK_OP_INS;
            ti++;
            if (ti == edits[e].pos + (size_t)(edits[e].row - 10)) e++;
        } else {
            out[w++] = op;
            if (e < ne && edits[e].row < 4 && edits[e].pos == ti) e++;
            ti++;
        }
        PENDING_INSERTS();
    }
#undef PENDING_INSERTS
    return w;
}

static void retag_ops(uint8_t *ops, size_t n_ops, const uint8_t *tmpl, const uint8_t *read) {
    size_t i = 0, j = 0;
    for (size_t k = 0; k < n_ops; k++) {
        if (ops[k] == JTK_OP_INS) {
            j++;
        } else if (ops[k] == JTK_OP_DEL) {
            i++;
        } else {
            ops[k] = tmpl[i] == read[j] ? JTK_OP_MATCH : JTK_OP_MISMATCH;
            i++;
            j++;
        }
    }
}

int64_t jo_phmm_polish(const jtk_hmm_t *fwd, const jtk_hmm_t *rev, const uint8_t *tmpl, size_t tl, size_t n,
                       const uint8_t *const *reads, const size_t *read_len, uint8_t **ops, size_t *ops_len,
                       size_t ops_cap, const uint8_t *strands, size_t radius, size_t take_num,
                       size_t ignore_edge, uint8_t *cons, size_t cons_cap, uint32_t *rounds_out) {
    if (tl > cons_cap) return -1;
    uint8_t *cur = (uint8_t *)malloc(cons_cap + 8), *nxt = (uint8_t *)malloc(cons_cap + 8);
    uint8_t *ops_tmp = (uint8_t *)malloc(ops_cap + 8);
    memcpy(cur, tmpl, tl);
    size_t L = tl;
    if (take_num > n) take_num = n;
    uint32_t round = 0;
    int64_t rc = 0;
    for (; round < MAX_POLISH_ROUNDS; round++) {
        size_t cols = NUM_ROW * (L + 1);
        double *total = (double *)calloc(cols, sizeof(double));
        double *tab = (double *)malloc(cols * sizeof(double));
        for (size_t r = 0; r < take_num; r++) {
            const jtk_hmm_t *h = strands[r] ? fwd : rev;
            double lk = jo_phmm_modification_table(h, cur, L, reads[r], read_len[r], ops[r], ops_len[r],
                                                   radius, tab);
            for (size_t p = 0; p < cols; p++) total[p] += tab[p] - lk;
        }
        edit_t *edits = (edit_t *)malloc((L + 1) * sizeof(edit_t));
        size_t inactive = 5 + (5 * (size_t)round) % 21;
        size_t ne = select_edits(total, L, ignore_edge, inactive, edits);
        free(total);
        free(tab);
        if (ne == 0) {
            free(edits);
            round++;
            break;
        }
        size_t grow = 0;
        for (size_t e = 0; e < ne; e++) grow += edit_inserted(&edits[e], L);
        if (L + grow > cons_cap) {
            free(edits);
            rc = -1;
            break;
        }
        size_t newL = apply_edits_template(cur, L, edits, ne, nxt);
        for (size_t r = 0; r < n && rc == 0; r++) {
            if (ops_len[r] + grow > ops_cap) {
                rc = -1;
                break;
            }
            size_t m = rethread_ops(ops[r], ops_len[r], L, edits, ne, ops_tmp);
            retag_ops(ops_tmp, m, nxt, reads[r]);
            memcpy(ops[r], ops_tmp, m);
            ops_len[r] = m;
        }
        free(edits);
        if (rc) break;
        uint8_t *t = cur;
        cur = nxt;
        nxt = t;
        L = newL;
    }
    if (rc == 0) {
        memcpy(cons, cur, L);
        rc = (int64_t)L;
    }
    if (rounds_out) *rounds_out = round;
    free(cur);
    free(nxt);
    free(ops_tmp);
    return rc;
}

/* Global unit-cost alignment; ties: diagonal, then Del, then Ins (traceback from the end). */
size_t jo_edit_ops(const uint8_t *tmpl, size_t tl, const uint8_t *read, size_t rl, uint8_t *ops) {
    size_t W = rl + 1;
    uint32_t *D = (uint32_t *)malloc((tl + 1) * W * sizeof(uint32_t));
    for (size_t j = 0; j <= rl; j++) D[j] = (uint32_t)j;
    for (size_t i = 1; i <= tl; i++) {
        D[i * W] = (uint32_t)i;
        for (size_t j = 1; j <= rl; j++) {
            uint32_t a = D[(i - 1) * W + j - 1] + (tmpl[i - 1] != read[j - 1]);
            uint32_t b = D[(i - 1) * W + j] + 1, c = D[i * W + j - 1] + 1;
            uint32_t m = a < b ? a : b;
            D[i * W + j] = m < c ? m : c;
        }
    }
    size_t i = tl, j = rl, k = 0;
    while (i > 0 || j > 0) {
        if (i > 0 && j > 0 && D[i * W + j] == D[(i - 1) * W + j - 1] + (tmpl[i - 1] != read[j - 1])) {
            ops[k++] = tmpl[i - 1] == read[j - 1] ? JTK_OP_MATCH : JTK_OP_MISMATCH;
            i--;
            j--;
        } else if (i > 0 && D[i * W + j] == D[(i - 1) * W + j] + 1) {
            ops[k++] = JTK_OP_DEL;
            i--;
        } else {
            ops[k++] = JTK_OP_INS;
            j--;
        }
    }
    for (size_t a = 0; a < k / 2; a++) {
        uint8_t t = ops[a];
        ops[a] = ops[k - 1 - a];
        ops[k - 1 - a] = t;
    }
    free(D);
    return k;
}

double jo_phmm_likelihood_bootstrap(const jtk_hmm_t *hmm, const uint8_t *tmpl, size_t tl,
                                    const uint8_t *read, size_t rl, size_t radius) {
    uint8_t *ops = (uint8_t *)malloc(tl + rl + 1);
    size_t k = jo_edit_ops(tmpl, tl, read, rl, ops);
    double lk = jo_phmm_likelihood(hmm, tmpl, tl, read, rl, ops, k, radius);
    free(ops);
    return lk;
}

void jo_generate_seq(jo_rng_t *rng, size_t len, uint8_t *out) {
    for (size_t i = 0; i < len; i++) out[i] = (uint8_t)"ACGT"[jo_gen_index(rng, 4)];
}

/* Read simulation from the model: start in Match at template position 0; repeatedly draw the next
 * state from the current state's transition row (choose_weighted over [->M, ->I, ->D]); Match emits from
 * mat_emit[x[i]][.] and advances, Ins emits from ins_emit[prev read base or 4][.], Del advances; stop when
 * the template is exhausted. */
size_t jo_phmm_gen(const jtk_hmm_t *h, const uint8_t *tmpl, size_t tl, jo_rng_t *rng, uint8_t *out,
                   size_t cap) {
    size_t i = 0, w = 0;
    int state = 0, prev = 4;
    while (i < tl && w + 1 < cap) {
        double tr[3];
        if (state == 0) {
            tr[0] = h->mat_mat; tr[1] = h->mat_ins; tr[2] = h->mat_del;
        } else if (state == 1) {
            tr[0] = h->ins_mat; tr[1] = h->ins_ins; tr[2] = h->ins_del;
        } else {
            tr[0] = h->del_mat; tr[1] = h->del_ins; tr[2] = h->del_del;
        }
        int64_t ns = jo_choose_weighted(rng, tr, 3);
        if (ns < 0) break;
        state = (int)ns;
        if (state == 0) {
            int64_t b = jo_choose_weighted(rng, h->mat_emit + 4 * base_code(tmpl[i]), 4);
            if (b < 0) break;
            out[w++] = (uint8_t)"ACGT"[b];
            prev = (int)b;
            i++;
        } else if (state == 1) {
            int64_t b = jo_choose_weighted(rng, h->ins_emit + 4 * prev, 4);
            if (b < 0) break;
            out[w++] = (uint8_t)"ACGT"[b];
            prev = (int)b;
        } else {
            i++;
        }
    }
    return w;
}
