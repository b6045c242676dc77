// decq.h -- exact arithmetic emulation of the reference's two decimal text round trips.
//
// The reference writes every feature with C "%.4g" (fv.cpp:133, via ostream << setprecision(4)) and
// svm-scale re-reads it with sscanf("%lf") (svm-scale.c:270); svm-scale prints the scaled value with
// "%g" (svm-scale.c:350) and svm-predict re-reads it with strtod (svm-predict.c:108).  Each round trip is
//      x  ->  N * 10^q   (N = the P most significant decimal digits of x, round-half-even on the EXACT
//                         binary value, as glibc printf does)  ->  nearest double to N * 10^q (strtod).
// haf_decq(x, P) computes that double without ever forming text:
//   * N comes from an exact product/quotient: x*10^k as an unevaluated sum hi+lo (one multiply + one fma)
//     or x/10^j as quotient + exact fma remainder, so ties and near-ties are decided exactly;
//   * the result N*10^j or N/10^k is ONE IEEE operation on exactly representable operands, hence correctly
//     rounded, which is what strtod returns.
// This is exact whenever |k|, |j| <= 22 (10^22 is the largest exact power of ten in binary64), i.e. for
// 1e-19 <= |x| < 1e26 (P=4) and 1e-17 <= |x| < 1e28 (P=6).  Outside that window (never reached by height
// data in metres) a double-double evaluation is used whose decision can differ from glibc only when x lies
// within ~1e-30 relative of a rounding boundary; tests/test_decq.py exercises both windows against glibc.
//
// The same source is compiled for the device (kernels.hip) and for the host (decq_host.cpp -> unit tests).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HAF_HD __host__ __device__ __forceinline__
#else
#define HAF_HD inline
#endif

namespace hafq {

HAF_HD double pow10_exact(int k)   // 0 <= k <= 22
{
    // switch keeps the constants in the instruction stream on the device (no constant-memory table needed)
    switch (k) {
        case 0: return 1e0;   case 1: return 1e1;   case 2: return 1e2;   case 3: return 1e3;
        case 4: return 1e4;   case 5: return 1e5;   case 6: return 1e6;   case 7: return 1e7;
        case 8: return 1e8;   case 9: return 1e9;   case 10: return 1e10; case 11: return 1e11;
        case 12: return 1e12; case 13: return 1e13; case 14: return 1e14; case 15: return 1e15;
        case 16: return 1e16; case 17: return 1e17; case 18: return 1e18; case 19: return 1e19;
        case 20: return 1e20; case 21: return 1e21; default: return 1e22;
    }
}

// round-half-even of the exact positive value hi+lo (|lo| <= ulp(hi)/2, hi < 2^52)
HAF_HD double rhe(double hi, double lo)
{
    double fl = floor(hi);
    double frac = hi - fl;                 // exact
    if (frac > 0.5) return fl + 1.0;
    if (frac < 0.5) return fl;             // also covers frac == 0 with lo < 0: nearest integer is still fl
    if (lo > 0.0) return fl + 1.0;
    if (lo < 0.0) return fl;
    double half = fl * 0.5;                // exact tie: to even
    return (floor(half) == half) ? fl : fl + 1.0;
}

struct dd { double hi, lo; };

HAF_HD dd dd_mul_d(dd a, double b)        // (a.hi + a.lo) * b, double-double
{
    double p = a.hi * b;
    double e = fma(a.hi, b, -p);
    e = fma(a.lo, b, e);
    double s = p + e;
    dd r; r.hi = s; r.lo = e - (s - p);
    return r;
}

HAF_HD dd dd_div_d(dd a, double b)        // (a.hi + a.lo) / b, double-double
{
    double q1 = a.hi / b;
    double r = fma(-q1, b, a.hi);          // exact remainder of the rounded quotient
    r = r + a.lo;
    double q2 = r / b;
    double s = q1 + q2;
    dd o; o.hi = s; o.lo = q2 - (s - q1);
    return o;
}

// Slow path: |k| > 22.  Scales by 10^22 repeatedly in double-double.
HAF_HD double decq_wide(double a, int P, int e)
{
    const double lo_bound = pow10_exact(P - 1), hi_bound = pow10_exact(P);
    for (int iter = 0; iter < 3; iter++) {
        int k = P - 1 - e;                 // t = a * 10^k
        dd t; t.hi = a; t.lo = 0.0;
        int rem = k;
        while (rem > 22) { t = dd_mul_d(t, 1e22); rem -= 22; }
        while (rem < -22) { t = dd_div_d(t, 1e22); rem += 22; }
        if (rem >= 0) t = dd_mul_d(t, pow10_exact(rem)); else t = dd_div_d(t, pow10_exact(-rem));
        if (t.hi > hi_bound || (t.hi == hi_bound && t.lo >= 0.0)) { e++; continue; }
        if (t.hi < lo_bound || (t.hi == lo_bound && t.lo < 0.0)) { e--; continue; }
        double N = rhe(t.hi, t.lo);
        dd r; r.hi = N; r.lo = 0.0;        // N * 10^-k
        rem = -k;
        while (rem > 22) { r = dd_mul_d(r, 1e22); rem -= 22; }
        while (rem < -22) { r = dd_div_d(r, 1e22); rem += 22; }
        if (rem >= 0) r = dd_mul_d(r, pow10_exact(rem)); else r = dd_div_d(r, pow10_exact(-rem));
        return r.hi + r.lo;
    }
    return a;
}

// |x| -> nearest double to the P-significant-digit decimal nearest to |x| (see file header).
HAF_HD double decq_abs(double a, int P)
{
    // decimal exponent estimate from the binary exponent: floor(log10(a)) is e0 or e0+1
    int b = ilogb(a);
    int e = (b * 1233) >> 12;              // floor(b*log10(2)) for |b| < 1100 (checked in tests)
    const double lo_bound = pow10_exact(P - 1), hi_bound = pow10_exact(P);
    for (int iter = 0; iter < 3; iter++) {
        int k = P - 1 - e;                 // t = a * 10^k in [10^(P-1), 10^P)
        if (k > 22 || k < -22) return decq_wide(a, P, e);
        if (k >= 0) {
            double T = pow10_exact(k);
            double hi = a * T;
            double lo = fma(a, T, -hi);    // exact: a*T = hi + lo
            if (hi > hi_bound || (hi == hi_bound && lo >= 0.0)) { e++; continue; }
            if (hi < lo_bound || (hi == lo_bound && lo < 0.0)) { e--; continue; }
            double N = rhe(hi, lo);
            return N / T;                  // one correctly rounded division of exact operands == strtod
        } else {
            double T = pow10_exact(-k);
            double q = a / T;
            double r = fma(-q, T, a);      // exact: a = q*T + r, sign(r) = sign(a/T - q)
            if (q > hi_bound || (q == hi_bound && r >= 0.0)) { e++; continue; }
            if (q < lo_bound || (q == lo_bound && r < 0.0)) { e--; continue; }
            double N = rhe(q, r);
            return N * T;                  // one correctly rounded product of exact operands == strtod
        }
    }
    return a;
}

// strtod(sprintf("%.{P}g", x)) for any x; zeros, infinities and NaNs pass through like the text forms do.
HAF_HD double decq(double x, int P)
{
    double a = fabs(x);
    if (!(a > 0.0) || !(a < INFINITY)) return x;
    double r = decq_abs(a, P);
    return x < 0.0 ? -r : r;
}

// svm-scale output() (svm-scale.c:333-353) + "%g" round trip.  q4 is the value svm-scale parsed.
// Returns the attribute value svm-predict parses (0.0 when the attribute is omitted from the text).
HAF_HD double scale_q6(double q4, double fmin, double fmax, double lower, double upper)
{
    double value;
    if (q4 == fmin) value = lower;
    else if (q4 == fmax) value = upper;
    else value = lower + (upper - lower) * (q4 - fmin) / (fmax - fmin);   // no contraction: built with -ffp-contract=off
    if (value == 0.0) return 0.0;          // "if(value != 0)" attribute omitted
    return decq(value, 6);
}

}  // namespace hafq
