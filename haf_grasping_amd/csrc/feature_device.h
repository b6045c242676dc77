// feature_device.h -- one feature value / one scaled attribute from the 15x15 integral window: the device functions shared by the
// feature kernels (features.hip) and the exact re-evaluation kernels (recheck.hip).
//   CIntImage_to_Featurevec::calc_featurevalue (fv.cpp:141-199), the "%.4g" text round trip (fv.cpp:133 -> svm-scale.c:270),
//   svm-scale restore+output (svm-scale.c:333-353) and the "%g" round trip (svm-scale.c:350 -> svm-predict.c:108)
#pragma once
#include "device_common.h"

namespace haf {

// ---------------------------------------------------------------------------------------------------
// a5/a6: one feature value from the 15x15 integral window (fv.cpp:141-199).  fp32, strict order, unfused.
// ---------------------------------------------------------------------------------------------------
// Integral-image reads go through a buffer descriptor: address = descriptor base + 32-bit VGPR byte offset (the window
// origin of the lane's cell) + SGPR byte offset (the region corner from the wave-uniform feature descriptor), i.e.
// `buffer_load_dword v, v_off, s[rsrc], s_off offen` with NO vector address arithmetic per load.  UNI = false (feature
// index differs per lane: recheck kernels) folds the corner offset into the VGPR instead.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_ii_rsrc(const float *ii, Dims d)
{
    const unsigned bytes = (unsigned)d.B * (unsigned)d.R * (unsigned)((d.H + 1) * (d.W + 1)) * 4u;   // < 2^32, checked in haf_create
    return __builtin_amdgcn_make_buffer_rsrc((void *)ii, 0, (int)bytes, 0x00020000);
}

template <bool UNI>
__device__ __forceinline__ float ii_load(rsrc_t r, unsigned w0b, int off)
{
    if (UNI) return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)w0b, off * 4, 0));
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)(w0b + (unsigned)off * 4u), 0, 0));
}

// Where the corners of a feature's regions come from: the integral image through the buffer descriptor (corner offset in
// an SGPR when the feature is wave-uniform, UNI, else folded into the VGPR), or a copy of the evaluation's 15x15 window in
// LDS (k_features).  Same values, same arithmetic.
template <bool UNI>
struct SrcBuf {
    rsrc_t r;
    unsigned w0b;
    template <class FD>
    __device__ __forceinline__ float corner(const FD &f, int k, int j) const { return ii_load<UNI>(r, w0b, f.off[k][j]); }
};
struct SrcWin {
    const float *win;             // this lane's window, row pitch 15
    template <class FD>
    __device__ __forceinline__ float corner(const FD &f, int k, int j) const { return win[f.offw[k][j]]; }
};

// LDS forms of a descriptor for the workgroups whose lanes take DIFFERENT attributes (k_features_small, k_small_direct: the attribute
// index differs between the quarters of a wave, so the descriptor words are vector loads).  Read from global memory where they are
// needed they were seven dependent round trips per attribute -- corner offsets, weight, the next region's, the scaling constants --
// and those round trips, not the arithmetic, were the kernels' time (round 5: k_small_direct 35 us for 4 072 evaluations).  The
// workgroup copies what its mode reads once; window offsets fit a byte (15 x 15 - 1 = 224).  Same members, same values.
struct FeatDescX {                // exact attributes (attribute_value_rec): 64 bytes
    unsigned char offw[3][4];
    unsigned char active, shaf, skip, pad;
    float w[3];
    float pad1;
    double fmin, fmax, range, inv_range;
};
struct FeatDescS {                // screening attributes (screen_attribute): 48 bytes
    unsigned char offw[3][4];
    unsigned char active, shaf, skip, pad;
    float w[3];
    float scr_extra;
    double scr_mul, scr_add;
};
template <int THREADS>
__device__ __forceinline__ void stage_descriptors(const FeatDesc *__restrict__ fd, int n, FeatDescX *s_fd)
{
    for (int f = threadIdx.x; f < n; f += THREADS) {
        const FeatDesc &F = fd[f];
        FeatDescX L;
#pragma unroll
        for (int k = 0; k < 3; k++) {
#pragma unroll
            for (int j = 0; j < 4; j++) L.offw[k][j] = (unsigned char)F.offw[k][j];
            L.w[k] = F.w[k];
        }
        L.active = (unsigned char)F.active; L.shaf = F.shaf ? 1 : 0; L.skip = F.skip ? 1 : 0; L.pad = 0; L.pad1 = 0.0f;
        L.fmin = F.fmin; L.fmax = F.fmax; L.range = F.range; L.inv_range = F.inv_range;
        s_fd[f] = L;
    }
}
template <int THREADS>
__device__ __forceinline__ void stage_descriptors(const FeatDesc *__restrict__ fd, int n, FeatDescS *s_fd)
{
    for (int f = threadIdx.x; f < n; f += THREADS) {
        const FeatDesc &F = fd[f];
        FeatDescS L;
#pragma unroll
        for (int k = 0; k < 3; k++) {
#pragma unroll
            for (int j = 0; j < 4; j++) L.offw[k][j] = (unsigned char)F.offw[k][j];
            L.w[k] = F.w[k];
        }
        L.active = (unsigned char)F.active; L.shaf = F.shaf ? 1 : 0; L.skip = F.skip ? 1 : 0; L.pad = 0;
        L.scr_extra = F.scr_extra; L.scr_mul = F.scr_mul; L.scr_add = F.scr_add;
        s_fd[f] = L;
    }
}

template <class Src, class FD>
__device__ __forceinline__ float region_sum(const Src &src, const FD &f, int k)
{
    float s = __fsub_rn(src.corner(f, k, 0), src.corner(f, k, 1));
    s = __fsub_rn(s, src.corner(f, k, 2));
    return __fadd_rn(s, src.corner(f, k, 3));                            // fv.cpp:161-162 / 183-184
}

template <class Src, class FD>
__device__ __forceinline__ float feature_value(const Src &src, const FD &f)
{
    if (!f.shaf) {
        float rv = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; k++)
            if (f.active & (1 << k)) rv = __fadd_rn(rv, __fmul_rn(f.w[k], region_sum(src, f, k)));
        return rv;
    }
    float r[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 3; k++)
        if (f.active & (1 << k)) r[k] = __fmul_rn(f.w[k], region_sum(src, f, k));
    if (r[1] > r[0] && r[1] > r[2]) {                                    // fv.cpp:187-191
        float a = __fsub_rn(r[1], r[0]), b = __fsub_rn(r[1], r[2]);
        return (b < a) ? b : a;
    }
    return -1.0f;
}

// fp32 feature -> attribute value svm-predict would parse (both decimal text round trips emulated exactly)
template <class Src, class FD, class Tabs>
__device__ __forceinline__ double attribute_value(const Src &src, const FD &f, double lower, double upper, const Tabs &tb)
{
    float v = feature_value(src, f);
    double q4 = hafq::decq4_float(v, tb);
    return hafq::scale_q6(q4, f.fmin, f.fmax, f.range, f.inv_range, lower, upper, tb);
}

// The same, leaving the three stages of the attribute behind for haf_debug_fetch_attr (HAF_FLAG_KEEP_DEBUG): rec == nullptr in
// every production call.  An attribute svm-scale drops (f.skip) is 0 for the contraction; its feature and "%.4g" value are
// still what fv.cpp writes into the text file, so the record keeps them.
template <class Src, class FD, class Tabs>
__device__ __forceinline__ double attribute_value_rec(const Src &src, const FD &f, double lower, double upper, const Tabs &tb,
                                                      AttrRecord *rec)
{
    if (f.skip && !rec) return 0.0;
    const float v = feature_value(src, f);
    const double q4 = hafq::decq4_float(v, tb);
    const double x = f.skip ? 0.0 : hafq::scale_q6(q4, f.fmin, f.fmax, f.range, f.inv_range, lower, upper, tb);
    if (rec) { rec->feature = v; rec->pad = 0.0f; rec->q4 = q4; rec->scaled = x; }
    return x;
}

// Attribute for the SCREENING pass only, already multiplied by c (kernels.h: ScreenParams): the "%.4g" round trip through the
// table-driven decq4_float_scr, svm-scale's formula in plain fp64 with the constants folded on the host (u' = fma(q4,
// scr_mul, scr_add): one instruction with two scalar operands), and NO "%g" round trip.  With x the value svm-predict would parse and u = c x, the result u' satisfies
// |u' - u| <= 5e-6 |u'| (six significant decimal digits: half a unit of the sixth digit is <= 5e-6 relative) plus, in norm over
// the attributes, ScreenParams::eta_abs (engine.cpp: the fp64 roundings of both evaluations of the formula, the exact-zero
// omission and the min/max shortcuts).  screen_finish() carries that difference through the guard band; evaluations the
// screening pass cannot decide get the exact attributes in the three-pass tier.  An fp32 feature outside the decimal
// path's range comes back NaN and poisons the norms: that evaluation is never trusted.
// (round 5: + 3 u for the fp32 scaling -- fl32(q4), the product's and the sum's rounding inside the fma are one, the constants' own roundings --
// relative to |u'|; the part relative to |scr_add| is in eta_abs)
constexpr double kScreenEtaRel = 5.0e-6 * (1.0 + 1e-6) + 1.8e-7;
template <class Src, class FD>
__device__ __forceinline__ float screen_attribute(const Src &src, const FD &f, const hafq::ScrTabs &st)
{
    const float v = feature_value(src, f);
    return fmaf((float)hafq::decq4_float_scr(v, st), (float)f.scr_mul, (float)f.scr_add);     // (the same floats ScrDesc holds: engine_tables.cpp)
}

// The same with the low-rank form's noise bound (kernels.h: kLrK; features.hip: k_features_serial, LR) for a lane of a wave that is
// not a run of neighbours: nb >= |u' - u_lin| -- the "%.4g" rounding, the products' rounding errors (exact, by fma) and the sums'
// -- and, for a region whose sum is not provably EXACT in the reference's order ((a - b) - c) + d, the worst its three roundings can
// add (region_round_bound).  The exactness test, per region, for a grid without negative heights (monotone integral image, R >= 0):
// a - b by Sterbenz (a <= 2b) or b = 0; then s2 = s1 - c = R - d and s3 = R are multiples of ulp(d) below d when R < d
// (c = d = 0: nothing to round).  A SHAF slot (pad2 = 0) is passed through as it is and needs none.
// what the three roundings of ((a - b) - c) + d can add up to when the region does not pass the exactness test: u (|s1| + |s2| + |R|)
// (each operation is off by at most half an ulp of its own result) -- 0 when it passes.
// The test is LOCAL -- on the four corners and the computed sum alone, no assumption about the rest of the grid: a - b is exact by
// Sterbenz (b/2 <= a <= 2b) or with b = 0; a, b, c >= d >= 0 makes a, b, c and therefore s1 and s2 = s1 - c = R - d multiples of
// ulp(d), and 0 <= R < d keeps |s2| <= d and s3 = R below 2^24 ulp(d): representable.  (Were the arithmetic inexact the computed R
// could not pass 0 <= R < d with a true R outside: s2 = fl(s1 - c) keeps the sign of s1 - c.)
__device__ __forceinline__ float region_round_bound(float a, float b, float c, float d, float s1, float s2, float R)
{
    const bool e1 = (a <= 2.0f * b && b <= 2.0f * a) || b == 0.0f;
    const bool e2 = (c == 0.0f && d == 0.0f) || (d >= 0.0f && fminf(fminf(a, b), c) >= d && R >= 0.0f && R < d);
    return (e1 && e2) ? 0.0f : 6.1e-8f * (fabsf(s1) + fabsf(s2) + fabsf(R));
}
template <class Src>
__device__ __forceinline__ float screen_attribute_lr(const Src &src, const FeatDesc &f, const hafq::ScrTabs &st, float &nb)
{
    if (f.shaf) { nb = 0.0f; return screen_attribute(src, f, st); }
    float rv = 0.0f, ee = 0.0f, esum = 0.0f;
    bool first = true;
#pragma unroll
    for (int k = 0; k < 3; k++)
        if (f.active & (1 << k)) {
            const float a = src.corner(f, k, 0), b = src.corner(f, k, 1), c = src.corner(f, k, 2), d = src.corner(f, k, 3);
            const float s1 = __fsub_rn(a, b), s2 = __fsub_rn(s1, c);
            const float R = __fadd_rn(s2, d);
            esum = fmaf(fabsf(f.w[k]), region_round_bound(a, b, c, d, s1, s2, R), esum);
            const float p = __fmul_rn(f.w[k], R);
            ee += fmaf(f.w[k], R, -p);
            rv = __fadd_rn(rv, p);
            if (!first) esum = fmaf(6.1e-8f, fabsf(rv), esum);           // (0 + p is exact; every later addition rounds its result once)
            first = false;
        }
    const float q4f = (float)hafq::decq4_float_scr(rv, st);
    nb = f.pad2 * (fabsf(q4f - rv) + fabsf(ee) * 1.000001f + esum + 6.1e-8f * fabsf(rv));
    return fmaf(q4f, (float)f.scr_mul, (float)f.scr_add);
}

// ---- the fast form of the screening feature pass ------------------------------------------------------------------
// What bounds the per-lane form (buffer loads at window origin + corner offset) is the vector L1: the texture addresser
// coalesces 16 lanes at a time, 64 consecutive floats at an arbitrary alignment cost ~7.5 tag accesses per load, and with
// ~2400 loads per evaluation the TA is 97 % busy (profiles/README.md).  So a wave whose 64 evaluations are 64 neighbouring
// cells of one row (k_scan's order makes that the rule) first copies the band of the integral image its windows cover --
// 15 rows x 78 columns -- into LDS, and then reads every corner with ds_read_addtid_b32: LDS address = M0 + lane * 4, M0 =
// band + corner offset from the wave-uniform descriptor, so a corner costs two scalar instructions and one conflict-free
// LDS read, no vector address arithmetic, no L1 traffic.
constexpr int kBandRows = 15;
constexpr int kBandFloats4 = kBandRows * kBandPitch;  // per wave

// Four attribute slots of a "fast" group (ScreenParams::fast_groups: plain HAF features of at most two regions): the 32
// corner reads go out back to back before anything waits on them, and nothing branches.  An inactive region has weight 0
// and its corners at the window origin: it adds 0.0f * 0.0f, which leaves the sum of fv.cpp:164 as it is.
// hipcc does not know that the asm reads are asynchronous: the registers are handed on only through the s_waitcnt statement.
// The descriptors are read through the constant address space: the memory clobbers around the band would otherwise make
// hipcc fetch every wave-uniform descriptor word with a vector load.
typedef const ScrDesc __attribute__((address_space(4))) *ScrDescK;
__device__ __forceinline__ ScrDescK constant_ptr(const ScrDesc *p) { return (ScrDescK)(unsigned long long)p; }

// NB (low-rank form, kernels.h: kLrK): nb[q] >= |u' - u_lin| of the slot, u_lin the EXACTLY linear functional of the window behind
// the attribute -- the "%.4g" rounding |q4 - v| (formed in fp32: its cast costs another u |q4|) plus the fp32 roundings of the
// products and of their sum (<= 3.1 u sum|w_k R_k|), valid when the region sums themselves are exact (k_features_serial checks that
// per wave), times |scr_mul| (ScrDesc::pad, rounded up; 0 for a SHAF slot, which is passed through as it is).
// u = 2^-24: 3 u (1 + 1e-3) and 4 u (1 + 1e-3), rounded up (the fp32 roundings of the bound's own three operations included)
constexpr float kNbRound = 1.80e-7f, kNbRound3 = 2.40e-7f;
template <int NB>
__device__ __forceinline__ void screen_quad(unsigned band, ScrDescK sd, const hafq::ScrTabs &st, float *ud, float &nu2, float &rmin)
{
    float c[4][8];
    unsigned adr[4][8];                               // all descriptor words first: a volatile asm pins what follows it
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int j = 0; j < 8; j++) adr[q][j] = band + (unsigned)sd[q].off[j];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const unsigned a = adr[q][j];
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_read_addtid_b32 %0" : "=v"(c[q][j]) : "s"(a) : "m0");
        }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(c[0][0]), "+v"(c[0][1]), "+v"(c[0][2]), "+v"(c[0][3]), "+v"(c[0][4]), "+v"(c[0][5]), "+v"(c[0][6]), "+v"(c[0][7]));
#pragma unroll
    for (int q = 1; q < 4; q++)
        asm volatile("" : "+v"(c[q][0]), "+v"(c[q][1]), "+v"(c[q][2]), "+v"(c[q][3]), "+v"(c[q][4]), "+v"(c[q][5]), "+v"(c[q][6]),
                          "+v"(c[q][7]));
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float s10 = __fsub_rn(c[q][0], c[q][1]), s20 = __fsub_rn(s10, c[q][2]), s11 = __fsub_rn(c[q][4], c[q][5]), s21 = __fsub_rn(s11, c[q][6]);
        const float R0 = __fadd_rn(s20, c[q][3]);
        const float R1 = __fadd_rn(s21, c[q][7]);
        const float r0 = __fmul_rn(sd[q].w[0], R0);
        const float r1 = __fmul_rn(sd[q].w[1], R1);
        const float v = __fadd_rn(r0, r1);      // 0.0f + r0 first (fv.cpp:164) only turns a -0 into +0: same decimal, same u'
        const float q4f = (float)hafq::decq4_float_scr(v, st);
        ud[q] = fmaf(q4f, sd[q].scr_mul, sd[q].scr_add);
        if (NB) {
            // round 5: the fp32 roundings BOUNDED instead of measured -- two products (u |r_k| each), their sum (u |v|) and the cast of q4
            // (u |q4| <= 1.0005 u |v|), |v| <= (1 + u)(|r0| + |r1|): at most kNbRound (|r0| + |r1|), a thousandth of the "%.4g" term beside
            // it (|q4 - v| ~ 1e-4 |v|).  Six vector instructions per slot instead of eleven.
            float ar = 0.0f;                               // NB == 2: a wave that did not pass the exactness test as a whole (features.hip)
            if (NB == 2)
                ar = fabsf(sd[q].w[0]) * region_round_bound(c[q][0], c[q][1], c[q][2], c[q][3], s10, s20, R0) +
                     fabsf(sd[q].w[1]) * region_round_bound(c[q][4], c[q][5], c[q][6], c[q][7], s11, s21, R1);
            float nbq = fmaf(kNbRound, fabsf(r0) + fabsf(r1), fabsf(q4f - v));
            if (NB == 2) nbq += ar;
            // path A (NB == 1, the wave passed the exactness test as a whole): that test takes R >= 0 from the monotone integral image, which
            // holds for the TRUE sums; the fp32-rounded corners can leave a near-empty region at -1..-3 ulp(d), and then (a - b) - c may have
            // rounded.  The smallest computed R of the evaluation rides along (one v_min3 per slot); a negative one voids the
            // evaluation's screening pass (features.hip)
            if (NB == 1) rmin = fminf(fminf(R0, R1), rmin);
            nbq *= sd[q].pad;
            nu2 = fmaf(nbq, nbq, nu2);
            // (the sum is tied to the sequence of the volatile LDS reads: left to float, the temporaries of eight slots stay alive until
            // the sums are finally formed -- 45 registers, two waves of occupancy)
            asm volatile("" : "+v"(nu2));
        }
    }
}
__device__ __forceinline__ void screen_quad(unsigned band, ScrDescK sd, const hafq::ScrTabs &st, float *ud) { float dummy = 0.0f, dmin = 0.0f; screen_quad<0>(band, sd, st, ud, dummy, dmin); }

// Two attribute slots of any other group, from the band: three regions each, the HAF sum or the SHAF rule (feature_value).
// A slot of a dropped or absent attribute has scr_mul = scr_add = 0: its u' is 0 (NaN if its feature value left the decimal
// path's range, which only costs that evaluation the screening pass).
typedef const ScrDesc3 __attribute__((address_space(4))) *ScrDesc3K;
__device__ __forceinline__ ScrDesc3K constant_ptr(const ScrDesc3 *p) { return (ScrDesc3K)(unsigned long long)p; }

// (NQ slots per call: two in the plain form; ONE with the low-rank form's noise bound, whose temporaries would otherwise cost the
// kernel 48 registers -- a wave of occupancy -- for the four groups of 40 that take this path)
template <int NB, int NQ>
__device__ __forceinline__ void screen_pair3(unsigned band, ScrDesc3K sd, const hafq::ScrTabs &st, float *ud, float &nu2, float &rmin)
{
    float c[NQ][12];
    unsigned adr[NQ][12];
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
        for (int j = 0; j < 12; j++) adr[q][j] = band + (unsigned)sd[q].off[j];
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const unsigned a = adr[q][j];
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_read_addtid_b32 %0" : "=v"(c[q][j]) : "s"(a) : "m0");
        }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(c[0][0]), "+v"(c[0][1]), "+v"(c[0][2]), "+v"(c[0][3]), "+v"(c[0][4]), "+v"(c[0][5]), "+v"(c[0][6]), "+v"(c[0][7]),
                   "+v"(c[0][8]), "+v"(c[0][9]), "+v"(c[0][10]), "+v"(c[0][11]));
    if (NQ > 1)
        asm volatile("" : "+v"(c[NQ - 1][0]), "+v"(c[NQ - 1][1]), "+v"(c[NQ - 1][2]), "+v"(c[NQ - 1][3]), "+v"(c[NQ - 1][4]), "+v"(c[NQ - 1][5]), "+v"(c[NQ - 1][6]), "+v"(c[NQ - 1][7]),
                          "+v"(c[NQ - 1][8]), "+v"(c[NQ - 1][9]), "+v"(c[NQ - 1][10]), "+v"(c[NQ - 1][11]));
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        float r[3], Rk[3], ar = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float s1 = __fsub_rn(c[q][4 * k], c[q][4 * k + 1]), s2 = __fsub_rn(s1, c[q][4 * k + 2]);
            Rk[k] = __fadd_rn(s2, c[q][4 * k + 3]);
            r[k] = __fmul_rn(sd[q].w[k], Rk[k]);
            if (NB == 2) ar = fmaf(fabsf(sd[q].w[k]), region_round_bound(c[q][4 * k], c[q][4 * k + 1], c[q][4 * k + 2], c[q][4 * k + 3], s1, s2, Rk[k]), ar);
        }
        float v;
        if (sd[q].shaf) {                                              // wave-uniform
            v = -1.0f;
            if (r[1] > r[0] && r[1] > r[2]) {                          // fv.cpp:187-191
                const float a = __fsub_rn(r[1], r[0]), b = __fsub_rn(r[1], r[2]);
                v = (b < a) ? b : a;
            }
        } else {
            v = __fadd_rn(__fadd_rn(r[0], r[1]), r[2]);
        }
        const float q4f = (float)hafq::decq4_float_scr(v, st);
        ud[q] = fmaf(q4f, sd[q].scr_mul, sd[q].scr_add);
        if (NB) {
            // (HAF slots only -- pad[0] = 0 for SHAF: three products, the partial sum r0 + r1, the total and the cast of q4 rounded once each:
            // at most 4 u (|r0| + |r1| + |r2|) <= kNbRound3 of it)
            const float nbq = sd[q].pad[0] * (fmaf(kNbRound3, (fabsf(r[0]) + fabsf(r[1])) + fabsf(r[2]), fabsf(q4f - v)) + ar);
            nu2 = fmaf(nbq, nbq, nu2);
            if (NB == 1 && !sd[q].shaf) rmin = fminf(fminf(Rk[0], Rk[1]), fminf(Rk[2], rmin));   // (as in screen_quad; a SHAF slot is passed through, not bounded)
            asm volatile("" : "+v"(nu2));
        }
    }
}
__device__ __forceinline__ void screen_pair3(unsigned band, ScrDesc3K sd, const hafq::ScrTabs &st, float *ud) { float dummy = 0.0f, dmin = 0.0f; screen_pair3<0, 2>(band, sd, st, ud, dummy, dmin); }

// the decimal tables (95 doubles) in LDS: call from every thread of the workgroup before any divergent return
__device__ __forceinline__ hafq::PtrTabs load_decimal_tables(double *lds_tab)
{
    if (threadIdx.x < hafq::kTabDoubles) lds_tab[threadIdx.x] = hafq::tab_entry((int)threadIdx.x);
    __syncthreads();
    hafq::PtrTabs tb;
    tb.t = lds_tab;
    return tb;
}

// the screening decimal tables (decq.h: 256 exponent entries + 15 pairs, 2288 bytes) in LDS; workgroups of >= 256 threads
__device__ __forceinline__ hafq::ScrTabs load_screen_tables(unsigned long long *lds_tab)
{
    if (threadIdx.x < hafq::kScrExpEntries) lds_tab[threadIdx.x] = hafq::scr_tab_word((int)threadIdx.x);
    if (threadIdx.x < 2 * hafq::kScrPairs) lds_tab[hafq::kScrExpEntries + threadIdx.x] = hafq::scr_tab_word(hafq::kScrExpEntries + (int)threadIdx.x);
    __syncthreads();
    hafq::ScrTabs st;
    st.w = lds_tab;
    return st;
}

// BYTE offset of the 15x15 window origin II[i-7][j-7] of a cell id (br*H + i)*W + j inside the integral-image buffer
__device__ __forceinline__ unsigned window_origin(int cell, int H, int W)
{
    const int br = cell / (H * W);
    const int rem = cell - br * H * W;
    const int i = rem / W, j = rem - i * W;
    return ((unsigned)br * (unsigned)((H + 1) * (W + 1)) + (unsigned)((i - 7) * (W + 1) + (j - 7))) * 4u;
}

// X image, fp32 form: tiles of 32 evals, k-major inside a tile ([tile][kDP][32] fp32) -- the exact register image of
// the fp32 MFMA A operand, so the contraction kernel fills its A fragments with fully coalesced 256-byte loads.

}  // namespace haf
