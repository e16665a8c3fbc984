/* jtk_math.h -- deterministic f64 exp/log/scalbn shared by the HIP kernels and the CPU oracle.
 *
 * Why this exists: the reference (Rust) calls the platform libm for f64::exp / f64::ln
 * (haplotyper/src/local_clustering/pseudo_mcmc.rs:637,736; haplotyper/src/misc.rs:89;
 * haplotyper/src/likelihood_gains.rs:116-136).  Integer cluster labels depend on ~10^6 Metropolis
 * accept/reject decisions per chunk, each of which compares a u64 draw with exp(diff)*2^64, so the
 * device and the CPU checker must evaluate exp/log bit-identically.  Neither glibc's nor OCML's
 * implementation is available on the other side, so both sides use this one: the classic fdlibm
 * e_exp.c / e_log.c algorithms (<1 ulp), written with explicit operation order and NO fused
 * multiply-add (translation units that include this header are compiled with -ffp-contract=off).
 * tests/test_oracle_math.py checks it against libm (<= 1 ulp) and test_gpu_math checks device == host
 * bit for bit.
 */
#ifndef JTK_MATH_H
#define JTK_MATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define JTK_HD __host__ __device__ static inline
#else
#define JTK_HD static inline
#endif

JTK_HD uint64_t jtk_f64_bits(double x) {
    uint64_t u;
    memcpy(&u, &x, sizeof u);
    return u;
}
JTK_HD double jtk_bits_f64(uint64_t u) {
    double x;
    memcpy(&x, &u, sizeof x);
    return x;
}

/* x * 2^n, exact unless the result is subnormal/overflows (musl scalbn.c structure). */
JTK_HD double jtk_scalbn(double x, int n) {
    double y = x;
    if (n > 1023) {
        y *= 0x1p1023;
        n -= 1023;
        if (n > 1023) {
            y *= 0x1p1023;
            n -= 1023;
            if (n > 1023) n = 1023;
        }
    } else if (n < -1022) {
        y *= 0x1p-1022 * 0x1p53;
        n += 1022 - 53;
        if (n < -1022) {
            y *= 0x1p-1022 * 0x1p53;
            n += 1022 - 53;
            if (n < -1022) n = -1022;
        }
    }
    return y * jtk_bits_f64((uint64_t)(0x3ff + n) << 52);
}

/* Unbiased binary exponent of a positive normal double (floor(log2 x)). */
JTK_HD int jtk_ilogb_pos(double x) { return (int)((jtk_f64_bits(x) >> 52) & 0x7ff) - 1023; }

JTK_HD double jtk_exp(double x) {
    const double ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                 P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    double hi, lo, c, xx, y;
    int k, sign;
    uint32_t hx = (uint32_t)(jtk_f64_bits(x) >> 32);
    sign = (int)(hx >> 31);
    hx &= 0x7fffffff;
    if (hx >= 0x4086232b) { /* |x| >= 708.39 */
        if (x != x) return x;
        if (x > 709.782712893383973096) return x * 0x1p1023;
        if (x < -745.13321910194110842) return 0.0;
    }
    if (hx > 0x3fd62e42) { /* |x| > 0.5 ln2 */
        if (hx >= 0x3ff0a2b2)
            k = (int)(invln2 * x + (sign ? -0.5 : 0.5));
        else
            k = 1 - sign - sign;
        hi = x - (double)k * ln2hi;
        lo = (double)k * ln2lo;
        x = hi - lo;
    } else if (hx > 0x3e300000) { /* |x| > 2^-28 */
        k = 0;
        hi = x;
        lo = 0.0;
    } else {
        return 1.0 + x;
    }
    xx = x * x;
    c = x - xx * (P1 + xx * (P2 + xx * (P3 + xx * (P4 + xx * P5))));
    y = 1.0 + (x * c / (2.0 - c) - lo + hi);
    if (k == 0) return y;
    return jtk_scalbn(y, k);
}

JTK_HD double jtk_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t ui = jtk_f64_bits(x);
    uint32_t hx = (uint32_t)(ui >> 32);
    int k = 0;
    double hfsq, f, s, z, R, w, t1, t2, dk;
    if (hx < 0x00100000 || (hx >> 31)) {
        if ((ui << 1) == 0) return -1.0 / (x * x); /* log(+-0) = -inf */
        if (hx >> 31) return (x - x) / 0.0;        /* log(-#)  = NaN  */
        k -= 54;
        x *= 0x1p54;
        ui = jtk_f64_bits(x);
        hx = (uint32_t)(ui >> 32);
    } else if (hx >= 0x7ff00000) {
        return x;
    } else if (hx == 0x3ff00000 && (ui << 32) == 0) {
        return 0.0;
    }
    hx += 0x3ff00000 - 0x3fe6a09e;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffff) + 0x3fe6a09e;
    ui = ((uint64_t)hx << 32) | (ui & 0xffffffffu);
    x = jtk_bits_f64(ui);
    f = x - 1.0;
    hfsq = 0.5 * f * f;
    s = f / (2.0 + f);
    z = s * s;
    w = z * z;
    t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    R = t2 + t1;
    dk = (double)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

/* f64::max semantics of Rust (NaN-ignoring), used by pseudo_mcmc.rs:644,678,791. */
JTK_HD double jtk_fmax(double a, double b) {
    if (a != a) return b;
    if (b != b) return a;
    return a < b ? b : a;
}

#endif /* JTK_MATH_H */
