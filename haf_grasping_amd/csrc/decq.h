// decq.h -- exact arithmetic emulation of the reference's two decimal text round trips.
//
// The reference writes every feature with C "%.4g" (fv.cpp:133, via ostream << setprecision(4)) and
// svm-scale re-reads it with sscanf("%lf") (svm-scale.c:270); svm-scale prints the scaled value with
// "%g" (svm-scale.c:350) and svm-predict re-reads it with strtod (svm-predict.c:108).  Each round trip is
//      x  ->  N * 10^q   (N = the P most significant decimal digits of x, round-half-even on the EXACT
//                         binary value, as glibc printf does)  ->  nearest double to N * 10^q (strtod).
// decq(x, P) computes that double without ever forming text:
//   * N comes from an exact product/quotient: x*10^k as an unevaluated sum hi+lo (one multiply + one fma)
//     or x/10^j as quotient + exact fma remainder, so ties and near-ties are decided exactly;
//   * the result N*10^j or N/10^k is ONE correctly rounded IEEE operation on exactly representable operands,
//     which is what strtod returns.
// This is exact whenever |k|, |j| <= 22 (10^22 is the largest exact power of ten in binary64), i.e. for
// 1e-19 <= |x| < 1e26 (P=4) and 1e-17 <= |x| < 1e28 (P=6).  Outside that window (never reached by height
// data in metres) a double-double evaluation is used whose decision can differ from glibc only when x lies
// within ~1e-30 relative of a rounding boundary; tests/test_host_cpu.py exercises both windows against glibc.
//
// Two code paths, same results: a branch-light FAST path for the common magnitudes (0 <= k <= 22: the decimal exponent
// is decided exactly by ONE comparison against a table of "smallest double >= 10^e", no loop, no per-lane-divergent
// switch; the final division N/10^k is q = N*y, q += fma(-q, T, N)*y with y = RN(10^-k) -- verified EXHAUSTIVELY equal
// to IEEE division for every N in [10^3,10^4] U [10^5,10^6] and k in 0..22, tools/divtest.c), and the general SLOW path
// (loops, true divisions) for everything else.
//
// The same source is compiled for the device (features.hip, recheck.hip, prob.hip) and for the host (engine.cpp -> unit tests).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HAF_HD __host__ __device__ __forceinline__
#define HAF_HD_NOINLINE static __host__ __device__ __attribute__((noinline))
#else
#define HAF_HD inline
#define HAF_HD_NOINLINE inline
#endif

namespace hafq {

#define HAFQ_P10_LIST                                                                                              \
    1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, \
        1e20, 1e21, 1e22
#define HAFQ_P10INV_LIST                                                                                                 \
    1e-0, 1e-1, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6, 1e-7, 1e-8, 1e-9, 1e-10, 1e-11, 1e-12, 1e-13, 1e-14, 1e-15, 1e-16, 1e-17, \
        1e-18, 1e-19, 1e-20, 1e-21, 1e-22
// kBnd[e + 24] = the smallest double >= 10^e, e = -24..24 (generated with exact rational arithmetic): for a double a,
// a >= 10^e  <=>  a >= kBnd[e + 24], exactly, also where 10^e itself is not representable.
#define HAFQ_BND_LIST                                                                                              \
    0x1.357c299a88ea8p-80, 0x1.82db34012b252p-77, 0x1.e392010175ee6p-74, 0x1.2e3b40a0e9b50p-70, \
    0x1.79ca10c924224p-67, 0x1.d83c94fb6d2adp-64, 0x1.2725dd1d243acp-60, 0x1.70ef54646d497p-57, \
    0x1.cd2b297d889bdp-54, 0x1.203af9ee75616p-50, 0x1.6849b86a12b9cp-47, 0x1.c25c268497682p-44, \
    0x1.19799812dea12p-40, 0x1.5fd7fe1796496p-37, 0x1.b7cdfd9d7bdbbp-34, 0x1.12e0be826d695p-30, \
    0x1.5798ee2308c3ap-27, 0x1.ad7f29abcaf49p-24, 0x1.0c6f7a0b5ed8ep-20, 0x1.4f8b588e368f1p-17, \
    0x1.a36e2eb1c432dp-14, 0x1.0624dd2f1a9fcp-10, 0x1.47ae147ae147bp-7, 0x1.999999999999ap-4, \
    0x1.0000000000000p+0, 0x1.4000000000000p+3, 0x1.9000000000000p+6, 0x1.f400000000000p+9, \
    0x1.3880000000000p+13, 0x1.86a0000000000p+16, 0x1.e848000000000p+19, 0x1.312d000000000p+23, \
    0x1.7d78400000000p+26, 0x1.dcd6500000000p+29, 0x1.2a05f20000000p+33, 0x1.74876e8000000p+36, \
    0x1.d1a94a2000000p+39, 0x1.2309ce5400000p+43, 0x1.6bcc41e900000p+46, 0x1.c6bf526340000p+49, \
    0x1.1c37937e08000p+53, 0x1.6345785d8a000p+56, 0x1.bc16d674ec800p+59, 0x1.158e460913d00p+63, \
    0x1.5af1d78b58c40p+66, 0x1.b1ae4d6e2ef50p+69, 0x1.0f0cf064dd592p+73, 0x1.52d02c7e14af7p+76, \
    0x1.a784379d99db5p+79
// 10^k exactly (k <= 22) and RN(10^-k): a decimal literal IS the correctly rounded double of its value
static const double kP10_host[23] = {HAFQ_P10_LIST};
static const double kP10inv_host[23] = {HAFQ_P10INV_LIST};
static const double kBnd_host[49] = {HAFQ_BND_LIST};
#if defined(__HIP__)
__device__ static const double kP10_dev[23] = {HAFQ_P10_LIST};
__device__ static const double kP10inv_dev[23] = {HAFQ_P10INV_LIST};
__device__ static const double kBnd_dev[49] = {HAFQ_BND_LIST};
#endif

HAF_HD double pow10_exact(int k)   // 0 <= k <= 22
{
#if defined(__HIP_DEVICE_COMPILE__)
    return kP10_dev[k];
#else
    return kP10_host[k];
#endif
}
HAF_HD double pow10_inv(int k)     // RN(10^-k), 0 <= k <= 22
{
#if defined(__HIP_DEVICE_COMPILE__)
    return kP10inv_dev[k];
#else
    return kP10inv_host[k];
#endif
}

HAF_HD double pow10_bound(int e)   // smallest double >= 10^e, -24 <= e <= 24
{
#if defined(__HIP_DEVICE_COMPILE__)
    return kBnd_dev[e + 24];
#else
    return kBnd_host[e + 24];
#endif
}

// Table providers: the default reads the constant arrays above (host: static data, device: global memory through L1);
// kernels that evaluate millions of values copy the 95 doubles into LDS once per workgroup and pass a PtrTabs, which
// turns three dependent global loads per round trip into LDS reads.
struct GlobalTabs {
    HAF_HD double p10(int k) const { return pow10_exact(k); }
    HAF_HD double p10inv(int k) const { return pow10_inv(k); }
    HAF_HD double bnd(int e) const { return pow10_bound(e); }
};
constexpr int kTabDoubles = 23 + 23 + 49;
struct PtrTabs {
    const double *t;      // [0,23): 10^k   [23,46): RN(10^-k)   [46,95): smallest double >= 10^(e), e = -24..24
    HAF_HD double p10(int k) const { return t[k]; }
    HAF_HD double p10inv(int k) const { return t[23 + k]; }
    HAF_HD double bnd(int e) const { return t[46 + 24 + e]; }
};
HAF_HD double tab_entry(int i)    // the value PtrTabs expects at index i
{
    return i < 23 ? pow10_exact(i) : (i < 46 ? pow10_inv(i - 23) : pow10_bound(i - 46 - 24));
}

// round-half-even of the exact positive value hi+lo (|lo| <= ulp(hi)/2, hi < 2^52), branch-free
HAF_HD double rhe(double hi, double lo)
{
    double fl = floor(hi);
    double frac = hi - fl;                 // exact
    double half = fl * 0.5;
    bool odd = floor(half) != half;
    // frac < 0.5 (incl. frac == 0 with lo < 0): nearest integer is fl; exact tie only when frac == 0.5 and lo == 0
    bool up = (frac > 0.5) || ((frac == 0.5) && ((lo > 0.0) || ((lo == 0.0) && odd)));
    return up ? fl + 1.0 : fl;
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

// |k| > 22: scales by 10^22 repeatedly in double-double (not proven exact, see header)
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

// General (slow) path: any positive finite a.
HAF_HD_NOINLINE double decq_abs_slow(double a, int P)
{
    int b = ilogb(a);
    int e = (b * 1233) >> 12;              // floor(b*log10(2)) for |b| < 1100; floor(log10(a)) is e or e+1
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

// |x| -> nearest double to the P-significant-digit decimal nearest to |x| (see file header); P is 4 or 6.
template <class Tabs>
HAF_HD double decq_abs(double a, int P, const Tabs &tb)
{
    const int b = ilogb(a);
    const int e0 = (b * 1233) >> 12;       // floor(b*log10(2)); for -80 <= b <= 80 (checked exhaustively with exact
                                           // rationals) the decimal exponent floor(log10(a)) is e0 or e0 + 1
    if (e0 >= -24 && e0 <= 23) {
        const int e = e0 + ((a >= tb.bnd(e0 + 1)) ? 1 : 0);          // exact decision, see kBnd
        const int k = P - 1 - e;                                      // a * 10^k lies in [10^(P-1), 10^P)
        if (k >= 0 && k <= 22) {
            const double T = tb.p10(k), y = tb.p10inv(k);
            const double hi = a * T, lo = fma(a, T, -hi);             // exact: a*T = hi + lo
            const double lo_b = (P == 4) ? 1e3 : 1e5, hi_b = (P == 4) ? 1e4 : 1e6;
            if (hi >= lo_b && hi < hi_b) {                             // always true here; anything else goes the slow way
                const double N = rhe(hi, lo);
                double q = N * y;                                      // N / 10^k, correctly rounded (exhaustively verified):
                q = fma(fma(-q, T, N), y, q);                          //   q + (N - q*T) * RN(1/T)
                return q;
            }
        }
    }
    return decq_abs_slow(a, P);
}

// strtod(sprintf("%.{P}g", x)) for any x (P = 4 or 6); zeros, infinities and NaNs pass through like the text forms do.
template <class Tabs>
HAF_HD double decq(double x, int P, const Tabs &tb)
{
    double a = fabs(x);
    if (!(a > 0.0) || !(a < INFINITY)) return x;
    double r = decq_abs(a, P, tb);
    return x < 0.0 ? -r : r;
}
HAF_HD double decq(double x, int P) { return decq(x, P, GlobalTabs()); }

// "%.4g" round trip of an fp32 value (what fv.cpp:133 prints).  A float has 24 significant bits and 5^12 < 2^28, so for
// k <= 12 the product a * 10^k is EXACT in binary64 and round-half-even of it is a single v_rndne_f64.
template <class Tabs>
HAF_HD double decq4_float(float v, const Tabs &tb)
{
    const double x = (double)v;
    const double a = fabs(x);
    if (!(a > 0.0) || !(a < INFINITY)) return x;
    const int b = ilogb(a);
    const int e0 = (b * 1233) >> 12;
    double r;
    bool done = false;
    if (e0 >= -24 && e0 <= 23) {
        const int e = e0 + ((a >= tb.bnd(e0 + 1)) ? 1 : 0);
        const int k = 3 - e;
        if (k >= 0 && k <= 12) {
            const double T = tb.p10(k), y = tb.p10inv(k);
            const double t = a * T;                                    // exact
            if (t >= 1e3 && t < 1e4) {
                const double N = rint(t);                              // round-half-even, like glibc on the exact value
                double q = N * y;
                q = fma(fma(-q, T, N), y, q);                          // N / 10^k, correctly rounded
                r = q;
                done = true;
            }
        }
    }
    if (!done) r = decq_abs(a, 4, tb);
    return x < 0.0 ? -r : r;
}
HAF_HD double decq4_float(float v) { return decq4_float(v, GlobalTabs()); }

// ---- "%.4g" of an fp32 value for the SCREENING pass (features.hip: screening features) ----------------------------
// Table-driven and branch-free: the fp32 exponent byte E picks {thr, i0} from a 256-entry table, where thr is the smallest
// float >= 10^(e0+1) (e0 = floor((E-127) log10 2); floor(log10 |v|) is e0 or e0+1, decided exactly by |v| >= thr) and i0
// the slot of k = 3 - e0 in a table of {10^k, RN(10^-k)} pairs; the digits are N = rint(|v| 10^k), exact for k <= 12 as in
// decq4_float, and the result N * RN(10^-k) stands in for the correctly rounded quotient (2^-52 relative, carried by the
// screening band).  Supported: v == 0 and 1e-9 <= |v| < 1e4.  Everything else (and Inf/NaN) lands on a NaN pair and
// returns NaN, which poisons the norms of that evaluation, so it is never trusted; fp32 subnormals return 0 (an absolute
// error below 1.2e-38).
#define HAFQ_THR10_LIST                                                                                            \
    0x1.12e0c0p-30f, 0x1.5798f0p-27f, 0x1.ad7f2ap-24f, 0x1.0c6f7cp-20f, 0x1.4f8b5ap-17f, 0x1.a36e30p-14f,          \
        0x1.0624dep-10f, 0x1.47ae16p-7f, 0x1.99999ap-4f, 0x1.0p+0f, 0x1.4p+3f, 0x1.9p+6f, 0x1.f4p+9f, 0x1.388p+13f
static const float kThr10_host[14] = {HAFQ_THR10_LIST};      // smallest float >= 10^e, e = -9..4 (exact rationals)
#if defined(__HIP__)
__device__ static const float kThr10_dev[14] = {HAFQ_THR10_LIST};
#endif
constexpr int kScrExpEntries = 256, kScrPairs = 15;
constexpr int kScrTabWords = kScrExpEntries + 2 * kScrPairs;   // 8-byte words: [0,256) exponent entries, then 15 pairs

HAF_HD unsigned long long scr_tab_word(int i)                  // the 8-byte word the screening tables hold at index i
{
    const unsigned nan32 = 0x7fc00000u;
    if (i >= kScrExpEntries) {                                 // pair slot s <-> k = s - 1; slots 0 and 14 are NaN
        const int s = (i - kScrExpEntries) >> 1, k = s - 1;
        double v = __builtin_nan("");
        if (k >= 0 && k <= 12) v = ((i - kScrExpEntries) & 1) ? pow10_inv(k) : pow10_exact(k);
        return __builtin_bit_cast(unsigned long long, v);
    }
    if (i == 0) return ((unsigned long long)1 << 32) | nan32;  // zero and subnormals: k = 0, never a carry
    const int e0 = ((i - 127) * 1233) >> 12;
    const int i0 = 4 - e0;
    if (i0 < 1) return nan32;                                   // |v| >= 1e4 (and Inf/NaN): slot 0
    if (i0 > 14) return ((unsigned long long)14 << 32) | nan32; // |v| < 1e-10: slot 14
#if defined(__HIP_DEVICE_COMPILE__)
    const float thr = kThr10_dev[e0 + 10];
#else
    const float thr = kThr10_host[e0 + 10];
#endif
    return ((unsigned long long)i0 << 32) | __builtin_bit_cast(unsigned, thr);
}

struct ScrTabs {
    const unsigned long long *w;                               // kScrTabWords words (LDS on the device)
};
HAF_HD double decq4_float_scr(float v, const ScrTabs &st)
{
    const unsigned bits = __builtin_bit_cast(unsigned, v);
    const unsigned long long ent = st.w[(bits >> 23) & 0xffu];
    const float a = __builtin_bit_cast(float, bits & 0x7fffffffu);
    const float thr = __builtin_bit_cast(float, (unsigned)ent);
    const int s = (int)(ent >> 32) - ((a >= thr) ? 1 : 0);
#if defined(HAF_ABL) && HAF_ABL == 5                   // timing experiment: an 8-byte pair entry (two floats) instead of 16 bytes
    const float *pf = reinterpret_cast<const float *>(st.w + kScrExpEntries + s);
    return rint((double)v * (double)pf[0]) * (double)pf[1];
#endif
    const double *pr = reinterpret_cast<const double *>(st.w + kScrExpEntries) + 2 * s;
    return rint((double)v * pr[0]) * pr[1];          // round-half-even is symmetric: the sign (of a zero too) rides along
}

// svm-scale output() (svm-scale.c:333-353) + "%g" round trip.  q4 is the value svm-scale parsed; range = fmax - fmin
// and inv_range = RN(1/range) are per-attribute constants.  Returns the attribute value svm-predict parses
// (0.0 when the attribute is omitted from the text).
template <class Tabs>
HAF_HD double scale_q6(double q4, double fmin, double fmax, double range, double inv_range, double lower, double upper,
                       const Tabs &tb)
{
    double value;
    if (q4 == fmin) value = lower;
    else if (q4 == fmax) value = upper;
    else {
        const double num = (upper - lower) * (q4 - fmin);     // svm-scale.c:344-346, unfused (-ffp-contract=off)
        const double an = fabs(num);
        double q;
        if ((an < 1e290) && ((an > 1e-290) || (num == 0.0))) {
            // num / range, correctly rounded without a hardware division: y = RN(1/range); two Markstein steps
            // (q1 is a faithful quotient, so q2 = RN(q1 + r1*y) is the correctly rounded one)
            double q0 = num * inv_range;
            double q1 = fma(fma(-q0, range, num), inv_range, q0);
            q = fma(fma(-q1, range, num), inv_range, q1);
        } else {
            q = num / range;
        }
        value = lower + q;
    }
    if (value == 0.0) return 0.0;          // "if(value != 0)": attribute omitted
    return decq(value, 6, tb);
}
HAF_HD double scale_q6(double q4, double fmin, double fmax, double range, double inv_range, double lower, double upper)
{
    return scale_q6(q4, fmin, fmax, range, inv_range, lower, upper, GlobalTabs());
}

}  // namespace hafq
