/* model_fit.c -- CPU ORACLE (test infrastructure).  The pair-HMM refit of the stage preamble,
 * haplotyper/src/model_tune.rs:96-156 (`estimate_model_parameters_on_both_strands`, called at
 * local_clustering/mod.rs:58): TRAIN_ROUND = 10 rounds of [polish every training pile-up with
 * HMMPolishConfig::new(band / 2, N, 0) (:137-143), then kiley `fit_antidiagonal_par_multiple(&packs, max_band / 2)` (:144-151)].
 * The driver is a restatement of the in-tree Rust.  The fit itself lives in kiley 0.3.0 (not under /root/reference, no
 * test pins it): OWN SPECIFICATION below -- one Baum-Welch step on the banded pair-HMM of phmm.c; **parity with kiley is
 * unpinned**.
 *
 * Expected counts of one read (band, scaling, F / b as in phmm.c; P = likelihood):
 *   S -> M at (i,j):  F_S(i,j) a_SM hatM(i+1,j+1)      S -> I:  F_S(i,j) a_SI hatI(i,j+1)      S -> D:  F_S(i,j) a_SD b_D(i+1,j)
 *   mat_emit[x[i-1]][y[j-1]] += F_M(i,j) b_M(i,j)       ins_emit[ctx(j)][y[j-1]] += F_I(i,j) b_I(i,j)          (all / P)
 * Summation order (the device kernel keeps 64 partial sums, one per lane): a cell of band offset w adds to partial sum
 * w mod 64, diagonals in DESCENDING order and offsets ascending within a diagonal; a cell's weight is the exact power of
 * two 2^(E_F[t] + E_B - E_F[T]) (E_B = exponent of diagonal t+1 for transitions, of diagonal t for emissions) times
 * 1 / tot; the 64 partial sums are added in lane order, reads in read order, pile-ups in order.
 * M-step: every transition row, every mat_emit row and every ins_emit row is divided by its sum (rows whose sum is not
 * positive keep their old values). */
#include <stdlib.h>
#include <string.h>

#include "jtk_math.h"
#include "jtk_oracle.h"

#define NCNT JTK_FIT_COUNTS /* 9 transitions + 16 mat_emit + 20 ins_emit */

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

/* expected counts of one read; counts[NCNT] is overwritten; returns lk (JO_LOG_ZERO: counts are all zero) */
double jo_phmm_counts(const jtk_hmm_t *h, const uint8_t *tmpl, size_t L, const uint8_t *read, size_t n,
                      const uint8_t *ops, size_t n_ops, size_t radius, double *counts) {
    for (int k = 0; k < NCNT; k++) counts[k] = 0.0;
    const size_t T = L + n, W = 2 * radius + 1;
    const int r = (int)radius;
    uint32_t *c = (uint32_t *)malloc((T + 1) * sizeof(uint32_t));
    if (jo_band_centers(ops, n_ops, L, n, c) != 0) {
        free(c);
        return JO_LOG_ZERO;
    }
    uint8_t *x = (uint8_t *)malloc(L + 1), *y = (uint8_t *)malloc(n + 1);
    for (size_t i = 0; i < L; i++) x[i] = (uint8_t)base_code(tmpl[i]);
    for (size_t j = 0; j < n; j++) y[j] = (uint8_t)base_code(read[j]);
    double *FM = (double *)calloc((T + 1) * W, sizeof(double)), *FI = (double *)calloc((T + 1) * W, sizeof(double)),
           *FD = (double *)calloc((T + 1) * W, sizeof(double));
    double *toM = (double *)calloc(3 * W, sizeof(double)), *toI = (double *)calloc(3 * W, sizeof(double)),
           *toD = (double *)calloc(3 * W, sizeof(double));
    int *E = (int *)calloc(T + 1, sizeof(int));
#define RGET(arr, tt, ii)                                                                                 \
    (((tt) < 0 || (tt) > (int64_t)T || (ii) < (int64_t)c[tt] - r || (ii) > (int64_t)c[tt] + r)            \
         ? 0.0                                                                                            \
         : (arr)[((size_t)(tt) % 3) * W + (size_t)((ii) - ((int64_t)c[tt] - r))])
    /* ---- forward, exactly as phmm.c's forward() */
    for (int64_t t = 0; t <= (int64_t)T; t++) {
        int Eprev1 = t >= 1 ? E[t - 1] : 0, Eprev2 = t >= 2 ? E[t - 2] : 0;
        int Ecur = Eprev1;
        double s2 = Eprev2 == Ecur ? 1.0 : pow2i(Eprev2 - Ecur);
        double m = 0.0;
        double *fm = FM + (size_t)t * W, *fi = FI + (size_t)t * W, *fd = FD + (size_t)t * W;
        for (size_t w = 0; w < W; w++) {
            int64_t i = (int64_t)c[t] - r + (int64_t)w, j = t - i;
            double a = 0, b = 0, d = 0;
            if (i >= 0 && i <= (int64_t)L && j >= 0 && j <= (int64_t)n) {
                if (t == 0) {
                    a = 1.0;
                } else {
                    if (i >= 1 && j >= 1) a = h->mat_emit[4 * x[i - 1] + y[j - 1]] * (RGET(toM, t - 2, i - 1) * s2);
                    if (j >= 1) b = h->ins_emit[4 * (j >= 2 ? y[j - 2] : 4) + y[j - 1]] * RGET(toI, t - 1, i);
                    if (i >= 1) d = RGET(toD, t - 1, i - 1);
                }
            }
            fm[w] = a;
            fi[w] = b;
            fd[w] = d;
            if (a > m) m = a;
            if (b > m) m = b;
            if (d > m) m = d;
        }
        if (t > 0 && (t & (JO_SCALE_BLOCK - 1)) == 0 && m > 0.0) {
            int e = jtk_ilogb_pos(m);
            double s = pow2i(-e);
            for (size_t w = 0; w < W; w++) {
                fm[w] *= s;
                fi[w] *= s;
                fd[w] *= s;
            }
            Ecur += e;
        }
        E[t] = Ecur;
        double *oM = toM + ((size_t)t % 3) * W, *oI = toI + ((size_t)t % 3) * W, *oD = toD + ((size_t)t % 3) * W;
        for (size_t w = 0; w < W; w++) {
            oM[w] = __builtin_fma(fd[w], h->del_mat, __builtin_fma(fi[w], h->ins_mat, fm[w] * h->mat_mat));
            oI[w] = __builtin_fma(fd[w], h->del_ins, __builtin_fma(fi[w], h->ins_ins, fm[w] * h->mat_ins));
            oD[w] = __builtin_fma(fd[w], h->del_del, __builtin_fma(fi[w], h->ins_del, fm[w] * h->mat_del));
        }
    }
    const double tot = (FM[T * W + radius] + FI[T * W + radius]) + FD[T * W + radius];
    const double lk = tot > 0.0 ? jtk_log(tot) + (double)E[T] * 0.6931471805599453094 : JO_LOG_ZERO;
    if (!(tot > 0.0)) goto done;
    {
        /* ---- backward + counts */
        double *hM = toM, *hI = toI, *bD = toD; /* rings reused: three diagonals of hatM, hatI, b_D */
        memset(hM, 0, 3 * W * sizeof(double));
        memset(hI, 0, 3 * W * sizeof(double));
        memset(bD, 0, 3 * W * sizeof(double));
        int *EB = (int *)calloc(T + 3, sizeof(int));
        double(*part)[NCNT] = (double(*)[NCNT])calloc(64, sizeof(*part));
        double *vmv = (double *)malloc(W * sizeof(double)), *viv = (double *)malloc(W * sizeof(double)),
               *vdv = (double *)malloc(W * sizeof(double));
        const double inv = 1.0 / tot;
        for (int64_t t = (int64_t)T; t >= 0; t--) {
            int Ecur = t < (int64_t)T ? EB[t + 1] : 0;
            int E2 = t + 2 <= (int64_t)T ? EB[t + 2] : 0;
            double s2 = E2 == Ecur ? 1.0 : pow2i(E2 - Ecur);
            /* transitions leaving diagonal t: neighbours are in the scale of diagonal t+1 */
            const double wt = pow2i(E[t] + Ecur - E[T]) * inv;
            double m = 0.0;
            for (size_t w = 0; w < W; w++) {
                int64_t i = (int64_t)c[t] - r + (int64_t)w, j = t - i;
                double vm = 0, vi = 0, vd = 0;
                if (i >= 0 && i <= (int64_t)L && j >= 0 && j <= (int64_t)n) {
                    if (t == (int64_t)T) {
                        vm = vi = vd = 1.0;
                    } else {
                        double xm = RGET(hM, t + 2, i + 1) * s2, xi = RGET(hI, t + 1, i), xd = RGET(bD, t + 1, i + 1);
                        vm = __builtin_fma(h->mat_del, xd, __builtin_fma(h->mat_ins, xi, h->mat_mat * xm));
                        vi = __builtin_fma(h->ins_del, xd, __builtin_fma(h->ins_ins, xi, h->ins_mat * xm));
                        vd = __builtin_fma(h->del_del, xd, __builtin_fma(h->del_ins, xi, h->del_mat * xm));
                        const double fm = FM[(size_t)t * W + w], fi = FI[(size_t)t * W + w], fd = FD[(size_t)t * W + w];
                        double *p = part[w & 63];
                        p[0] += ((fm * h->mat_mat) * xm) * wt;
                        p[1] += ((fm * h->mat_ins) * xi) * wt;
                        p[2] += ((fm * h->mat_del) * xd) * wt;
                        p[3] += ((fi * h->ins_mat) * xm) * wt;
                        p[4] += ((fi * h->ins_ins) * xi) * wt;
                        p[5] += ((fi * h->ins_del) * xd) * wt;
                        p[6] += ((fd * h->del_mat) * xm) * wt;
                        p[7] += ((fd * h->del_ins) * xi) * wt;
                        p[8] += ((fd * h->del_del) * xd) * wt;
                    }
                }
                vmv[w] = vm;
                viv[w] = vi;
                vdv[w] = vd;
                if (vm > m) m = vm;
                if (vi > m) m = vi;
                if (vd > m) m = vd;
            }
            if (t < (int64_t)T && (t & (JO_SCALE_BLOCK - 1)) == JO_SCALE_BLOCK - 1 && m > 0.0) {
                int e = jtk_ilogb_pos(m);
                double s = pow2i(-e);
                for (size_t w = 0; w < W; w++) {
                    vmv[w] *= s;
                    viv[w] *= s;
                    vdv[w] *= s;
                }
                Ecur += e;
            }
            EB[t] = Ecur;
            const double we = pow2i(E[t] + Ecur - E[T]) * inv; /* emissions at diagonal t: b in the scale of diagonal t */
            double *oM = hM + ((size_t)t % 3) * W, *oI = hI + ((size_t)t % 3) * W, *oD = bD + ((size_t)t % 3) * W;
            for (size_t w = 0; w < W; w++) {
                int64_t i = (int64_t)c[t] - r + (int64_t)w, j = t - i;
                double a = 0, b = 0;
                if (i >= 1 && i <= (int64_t)L && j >= 1 && j <= (int64_t)n) {
                    a = h->mat_emit[4 * x[i - 1] + y[j - 1]] * vmv[w];
                    part[w & 63][9 + 4 * x[i - 1] + y[j - 1]] += (FM[(size_t)t * W + w] * vmv[w]) * we;
                }
                if (i >= 0 && i <= (int64_t)L && j >= 1 && j <= (int64_t)n) {
                    const int ctx = j >= 2 ? y[j - 2] : 4;
                    b = h->ins_emit[4 * ctx + y[j - 1]] * viv[w];
                    part[w & 63][25 + 4 * ctx + y[j - 1]] += (FI[(size_t)t * W + w] * viv[w]) * we;
                }
                oM[w] = a;
                oI[w] = b;
                oD[w] = vdv[w];
            }
        }
        for (int l = 0; l < 64; l++)
            for (int k = 0; k < NCNT; k++) counts[k] += part[l][k];
        free(EB);
        free(part);
        free(vmv);
        free(viv);
        free(vdv);
    }
done:
#undef RGET
    free(c);
    free(x);
    free(y);
    free(FM);
    free(FI);
    free(FD);
    free(toM);
    free(toI);
    free(toD);
    free(E);
    return lk;
}

/* the M-step on summed counts; rows without mass keep the old values */
void jo_fit_mstep(const jtk_hmm_t *old, const double *cnt, jtk_hmm_t *out) {
    *out = *old;
    double *tr[3] = {&out->mat_mat, &out->ins_mat, &out->del_mat};
    for (int s = 0; s < 3; s++) {
        const double sum = (cnt[3 * s] + cnt[3 * s + 1]) + cnt[3 * s + 2];
        if (sum > 0.0)
            for (int q = 0; q < 3; q++) tr[s][q] = cnt[3 * s + q] / sum;
    }
    for (int xr = 0; xr < 4; xr++) {
        const double *e = cnt + 9 + 4 * xr;
        const double sum = ((e[0] + e[1]) + e[2]) + e[3];
        if (sum > 0.0)
            for (int q = 0; q < 4; q++) out->mat_emit[4 * xr + q] = e[q] / sum;
    }
    for (int cx = 0; cx < 5; cx++) {
        const double *e = cnt + 25 + 4 * cx;
        const double sum = ((e[0] + e[1]) + e[2]) + e[3];
        if (sum > 0.0)
            for (int q = 0; q < 4; q++) out->ins_emit[4 * cx + q] = e[q] / sum;
    }
}

/* estimate_model_parameters_on_both_strands (model_tune.rs:119-152) on already selected training pile-ups (the
 * selection -- median coverage +-2, sorted by chunk id, first TRAIN_UNIT_SIZE = 5 -- is host glue, :99-118).
 * band_frac -> band_width(len) = ceil(len * frac) of the UNPOLISHED chunk (:123, fixed over the rounds). */
int jo_fit_model(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_chunk_t *chunks, const uint8_t *tmpl_bases,
                 const uint8_t *read_bases, const uint64_t *read_off, const uint8_t *ops, const uint64_t *ops_off,
                 const uint8_t *strand, uint32_t rounds, jtk_hmm_t *fwd_out, jtk_hmm_t *rev_out) {
    jtk_lc_params_t cur = *params;
    size_t n_reads = 0, tmpl_total = 0, max_bw = 0;
    for (size_t c = 0; c < n_chunks; c++) {
        n_reads += chunks[c].n_reads;
        tmpl_total += (size_t)chunks[c].tmpl_len;
        size_t bw = (size_t)__builtin_ceil((double)chunks[c].tmpl_len * params->band_frac);
        if (bw > max_bw) max_bw = bw;
    }
    if (n_chunks == 0 || n_reads == 0) return JTK_ERR_INVALID_ARG; /* assert!(!polishing_pairs.is_empty()) :135 */
    /* working copies: consensus and ops change from round to round */
    size_t ccap = 2 * tmpl_total + 64 * n_chunks + 64, ocap = 2 * (size_t)ops_off[n_reads] + 64 * n_reads + 64;
    uint8_t *cons = (uint8_t *)malloc(ccap), *cons2 = (uint8_t *)malloc(ccap);
    uint8_t *cops = (uint8_t *)malloc(ocap), *cops2 = (uint8_t *)malloc(ocap);
    uint64_t *coff = (uint64_t *)malloc((n_chunks + 1) * 8), *coff2 = (uint64_t *)malloc((n_chunks + 1) * 8);
    uint64_t *ooff = (uint64_t *)malloc((n_reads + 1) * 8), *ooff2 = (uint64_t *)malloc((n_reads + 1) * 8);
    jtk_lc_chunk_t *ch = (jtk_lc_chunk_t *)malloc(n_chunks * sizeof(*ch));
    jtk_lc_result_t *res = (jtk_lc_result_t *)malloc(n_chunks * sizeof(*res));
    uint32_t *radius = (uint32_t *)malloc(n_chunks * sizeof(uint32_t));
    memcpy(ooff, ops_off, (n_reads + 1) * 8);
    memcpy(cops, ops, (size_t)ops_off[n_reads]);
    uint64_t o = 0;
    for (size_t c = 0; c < n_chunks; c++) {
        ch[c] = chunks[c];
        coff[c] = o;
        memcpy(cons + o, tmpl_bases + chunks[c].tmpl_off, (size_t)chunks[c].tmpl_len);
        o += chunks[c].tmpl_len;
        radius[c] = (uint32_t)((size_t)__builtin_ceil((double)chunks[c].tmpl_len * params->band_frac) / 2);
    }
    coff[n_chunks] = o;
    int rc = 0;
    for (uint32_t round = 0; round < rounds && rc == 0; round++) {
        /* polish every pile-up with HMMPolishConfig::new(bw / 2, seqs.len(), 0)  (:137-143) */
        uint64_t co = 0, oo = 0;
        for (size_t c = 0; c < n_chunks && rc == 0; c++) {
            jtk_lc_chunk_t one = ch[c];
            one.tmpl_off = coff[c];
            one.tmpl_len = coff[c + 1] - coff[c];
            uint64_t co1[2];
            jtk_lc_result_t r1;
            /* one window at a time keeps each pile-up's own radius */
            const uint64_t r0 = one.read_first;
            jtk_lc_chunk_t loc = one;
            loc.read_first = 0;
            rc = jo_polish_chunks(&cur, 1, &loc, cons, read_bases, read_off + r0, cops, ooff + r0, strand + r0, radius[c], 0, 0,
                                  cons2 + co, co1, cops2 + oo, ooff2 + r0, &r1, 1);
            if (rc) break;
            coff2[c] = co;
            for (uint32_t r = 0; r <= one.n_reads; r++) ooff2[r0 + r] += oo;
            co += co1[1];
            oo = ooff2[r0 + one.n_reads];
        }
        if (rc) break;
        coff2[n_chunks] = co;
        uint8_t *t8 = cons; cons = cons2; cons2 = t8;
        t8 = cops; cops = cops2; cops2 = t8;
        uint64_t *t64 = coff; coff = coff2; coff2 = t64;
        t64 = ooff; ooff = ooff2; ooff2 = t64;
        /* fit_antidiagonal_par_multiple(&packs, bw / 2) with bw = the LARGEST band width (:136,:151) */
        double sum[2][NCNT];
        memset(sum, 0, sizeof sum);
        double cnt[NCNT];
        for (size_t c = 0; c < n_chunks; c++)
            for (uint32_t r = 0; r < ch[c].n_reads; r++) {
                const uint64_t g = ch[c].read_first + r;
                const int s = strand[g] ? 0 : 1;
                jo_phmm_counts(s == 0 ? &cur.forward : &cur.reverse, cons + coff[c], (size_t)(coff[c + 1] - coff[c]),
                               read_bases + read_off[g], (size_t)(read_off[g + 1] - read_off[g]), cops + ooff[g],
                               (size_t)(ooff[g + 1] - ooff[g]), max_bw / 2, cnt);
                for (int k = 0; k < NCNT; k++) sum[s][k] += cnt[k];
            }
        jtk_hmm_t nf, nr;
        jo_fit_mstep(&cur.forward, sum[0], &nf);
        jo_fit_mstep(&cur.reverse, sum[1], &nr);
        cur.forward = nf;
        cur.reverse = nr;
    }
    *fwd_out = cur.forward;
    *rev_out = cur.reverse;
    free(cons); free(cons2); free(cops); free(cops2); free(coff); free(coff2); free(ooff); free(ooff2);
    free(ch); free(res); free(radius);
    return rc;
}
