// screen.hip -- the screening pass (tier 0) of the RBF decision: k_svm_screen.  Its own translation unit because it is
// built with -fno-slp-vectorize: hipcc's SLP pass packs pairs of the epilogue's fp32 FMAs into v_pk_fma_f32, which
// cannot be placed one by one between the MFMAs (and measured wrong sums on gfx950 when fed straight from v_exp_f32).
#include "kernels.h"
#include "screen_band.h"
#include <algorithm>
#include <cmath>
#include <vector>

namespace haf {

// (The A/B and ablation switches the measurements of DESIGN.md §5 were taken with are not part of this file:
// tools/screen_experiments.patch puts them back, tools/build_variant.sh builds such a variant next to the product library.)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------
// a8, screening pass (tier 0): the same decision function as ONE fp16 MFMA pass.  Operands are pre-scaled by
// c = sqrt(2*gamma*log2 e) and rounded to fp16 (u^ = fp16(c x), w^ = fp16(c s), attributes that are the same function of the
// window sharing one of 320 K slots: kernels.h).  Only u.w goes through the matrix core:
//     exp2 argument of (evaluation e, SV n) = u_e.w_n + t_n - |u_e|^2/2,    t_n = -|v_n|^2/2.
// t_n is the INITIAL VALUE of the accumulator column (the C operand of the first MFMA of each chain, one fp32 per SV from the
// tile image), -|u|^2/2 is common to all SVs of an evaluation and multiplies the two class sums once after the sweep.  So
// the epilogue is one v_exp_f32 and one fma per (evaluation, SV), and K is exactly ten 16x16x32 steps.
// (The coefficient stays a multiplier: as log2|coef_n| inside t_n it would save nothing -- hipcc splits a packed add next to
// MFMAs into two adds anyway -- and a tiny coefficient would put |t_n| ~ 12 into the accumulator, which the worst-case bound
// of the matrix core's fp32 accumulation pays for with a 2x wider band: measured 2.9 % instead of 1.9 % refined.)
// The result is only trusted outside a rigorous per-evaluation band
//     |dec| > min(gA |w|_2, gC S) + (guard_acc0 + gB) S + cm (|dec| + |rho|) + guard_abs,   S = sum|coef|K,  w_n = coef_n K_n,
// {gA, gB, gC, cm} from k_features (screen_finish) -- or, for the CENTRED estimate dec - corr (the first-order error of the part of
// w that does not depend on the evaluation, computed by k_features and subtracted here), outside
//     abs_c + (guard_acc0 + gB) S + cm (|dec - corr| + |corr| + |rho|) + guard_abs, whichever band is narrower (kernels.h),
// which is ~20x wider than the three-pass kernel's, so a few per cent of the evaluations go on to that kernel (in list
// mode) and from there to the fp64 tiers as before: the labels stay those of libsvm, the bulk costs a third.
//   * a wave keeps 64 evals x 320 slots in 160 VGPRs (twice the rows of the three-pass kernel: every B fragment read
//     from LDS feeds 4 MFMAs); workgroup = 4 waves = 256 evals and TWO workgroups share a CU (one wave of each per SIMD):
//     their tile barriers, LDS-DMA bursts and prologues fall at different times, so one's stalls sit beside the other's MFMAs;
//   * SV tiles (32 SVs, 21 KiB) stream through a 3-deep LDS ring by LDS-DMA exactly as in k_svm_rbf_h;
//   * the exp/add epilogue of a 16-SV column block is issued BETWEEN the MFMAs of the next block (two accumulator
//     sets in ping-pong), so it overlaps the matrix pipe inside one wave instead of relying on the partner wave.
// ---------------------------------------------------------------------------------------------------
// One LDS-DMA piece (1 KiB): wave-uniform global base + lane*16, LDS destination M0 + lane*16.  Base and M0 must come out
// of SALU arithmetic (kernel arguments, readfirstlane results computed long before): an SGPR fresh from v_readfirstlane
// needs 5 wait states before a vector-memory instruction reads it, and nothing inside an asm string is padded.
__device__ __forceinline__ void dma_piece(const char *gbase, unsigned lds_dst, unsigned lane16)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(lane16), "s"(gbase) : "memory", "m0");
}

// the five pieces of one SV tile's operand image this wave stages (20 pieces over 4 waves; round 1 spread 22 pieces as 4 x 6 with
// two of them staged twice -- an eighth of the DMA instructions for nothing).  The 21st piece (t_n and the coefficients, 256
// bytes of it used) is staged by wave 0 alone, outside the MFMA stream.
struct TileDma {
    const char *g[kS0WavePieces];
    unsigned l[kS0WavePieces];
};
__device__ __forceinline__ TileDma tile_dma(const char *gtile, unsigned lds_slot, const int (&poff)[kS0WavePieces])
{
    TileDma d;
#pragma unroll
    for (int q = 0; q < kS0WavePieces; q++) { d.g[q] = gtile + poff[q]; d.l[q] = lds_slot + (unsigned)poff[q]; }
    return d;
}
__device__ __forceinline__ void stage_sv_tile_s0(const TileDma &d, unsigned lane16)
{
    asm volatile("s_nop 4");                                         // prologue only: the bases may be fresh from v_readfirstlane
#pragma unroll
    for (int q = 0; q < kS0WavePieces; q++) dma_piece(d.g[q], d.l[q], lane16);
}

// MFMAs of column block n (16 SVs) of the tile at `cur` into acc, with the epilogue of the PREVIOUS block (old) spread over the
// k-steps: behind the four MFMAs of step s come the two v_exp_f32 of element pair s-1 and the two fmas of pair s-2.
// The chains start from t (this lane's column of the block: -|v|^2/2) as the C operand of their first MFMA.
// HAZARD (measured on gfx950, two waves per SIMD): a VALU instruction that reads a v_exp_f32 result within ~4 instructions
// of the v_exp_f32 -- an MFMA in between does not help -- can read the register BEFORE the transcendental unit has written
// it (wrong sums on some waves of some launches; hipcc pads one wait state, which is not enough).  Every exp result
// here is consumed one whole k-step (>= 4 MFMAs, >= 8 instructions) after it was issued; haf_grasping_amd/build.py checks
// that distance in the ISA of every build.
// The wave's LDS-DMA pieces of the tile two ahead are issued behind the first MFMAs of k-step 0 (which carries no
// epilogue work), inside the MFMA stream instead of all waves paying for them together behind the tile barrier.
// No branch may sit inside this stream: with the DMA under a wave-uniform `if`, hipcc's code motion carries the epilogue of
// the block out of the MFMA stream (into the next basic block), and with two template instantiations in the arms of an
// if/else it hoists the epilogue they have in common in front of the branch.  So every wave issues the same DMA in every block:
// its pieces 0..2 of the tile two ahead in column block 0, its pieces 3..4 in column block 1 (FIRST = first piece, COUNT = how many).
// SUMSQ (the variant for ill-conditioned models): the epilogue also accumulates q += (coef K)^2, so that the sqrt(S) form of the
// band can use |w|_2^2 = sum_n (coef_n K_n)^2 itself instead of its bound max|coef| * S: three VALU instructions
// per element behind the exp instead of one.
// B fragments are read one k-step ahead of their MFMAs, ACROSS the boundary between the two column blocks of a tile: block 0
// (n = 0) reads its own first fragment on entry and leaves that of block 1 in b on exit, so that block 1's first MFMA does not
// wait for an LDS round trip.  (Across tiles that is not possible: the next tile is only known to have landed behind the
// barrier.)  Two steps ahead measured the same (round 1) and costs four VGPRs, which now hold a second level of the
// coefficient sum.
// CR_EXP / CR_POLY (round 4, kernels.h: the centred-remainder form): the chains start from 0, the accumulator is z = p^.q^_n and the
// epilogue accumulates b_n psi(z), psi(z) = 2^z - 1 - z ln2 -- by v_exp_f32, a subtraction and an fma (CR_EXP: three VALU
// instructions behind the exp, like SUMSQ), or with no transcendental at all as z^2 (a2 + a3 z + a4 z^2 + a5 z^3), a_k = ln2^k / k!
// (CR_POLY: five VALU instructions per element -- the coefficient is folded into the constants per block --, for models whose z stay small and whose sum|b| is so large that an ulp of 2^z is
// too much: relative accuracy instead of absolute).
constexpr float kLn2f = 0.693147180559945f;
constexpr float kPsiA2 = 0.240226506959101f, kPsiA3 = 0.0555041086648216f, kPsiA4 = 0.00961812910762848f, kPsiA5 = 0.00133335581464284f;
template <int FIRST, int COUNT, int VAR>
__device__ __forceinline__ void screen_block(const char *cur, int n, int lane, const half8 (&a)[kHFull][4], f32x4 (&acc)[4],
                                             const f32x4 (&old)[4], float t, float cf_old, float (&sum)[4][4], float (&sq)[4][4],
                                             const TileDma &dma, unsigned lane16, half8 &b, half8 &b1)
{
    constexpr bool SUMSQ = VAR == SCREEN_SUMSQ, CRE = VAR == SCREEN_CR_EXP, CRP = VAR == SCREEN_CR_POLY, PLAIN = VAR == SCREEN_PLAIN;
    const char *bl = cur + n * 1024 + lane * 16;
    __builtin_amdgcn_sched_barrier(0);                               // DMA issue and address arithmetic stay in front
    const f32x4 t4 = {t, t, t, t};                                   // rows differ, the column (this lane's SV) is the same
    // CR_POLY: b psi(z) = z^2 (b a2 + b a3 z + b a4 z^2 + b a5 z^3) -- the coefficient of the PREVIOUS block's column folded into the
    // four constants once per block (four multiplications per lane) instead of one per element: five VALU instructions per element
    const float ba2 = CRP ? cf_old * kPsiA2 : 0.0f, ba3 = CRP ? cf_old * kPsiA3 : 0.0f, ba4 = CRP ? cf_old * kPsiA4 : 0.0f, ba5 = CRP ? cf_old * kPsiA5 : 0.0f;
    if (n == 0) b = *reinterpret_cast<const half8 *>(bl);            // B[k = 32s + 8(lane>>4) + j][col 16n + (lane&15)]
    float k0 = 0.0f, k1 = 0.0f;                                      // exp2 of the pair issued in the previous k-step
    // The issue order of every k-step is pinned instruction by instruction (a scheduling barrier after each): B read of the
    // next step, then MFMA | exp | MFMA | exp | MFMA | fma | MFMA | fma, where the exps belong to pair s-1 and the fmas to
    // pair s-2.  hipcc's own scheduling (also with sched_group_barrier hints) pulls an fma right behind its exp.
#define HAF_SB() __builtin_amdgcn_sched_barrier(0)
#pragma unroll
    for (int s = 0; s < kHFull; s++) {
        // fragment s + 1 of this block, or -- in the last step of block 0 -- fragment 0 of block 1 (1 KiB further on)
        if (s + 1 < kHFull) b1 = *reinterpret_cast<const half8 *>(bl + (s + 1) * 2048);
        else if (n == 0) b1 = *reinterpret_cast<const half8 *>(bl + 1024);
        HAF_SB();
        const bool ex = s >= 1 && s < 9, fm = s >= 2;
        const int e0 = 2 * (s - 1), e1 = e0 + 1, f0 = 2 * (s - 2), f1 = f0 + 1;
        float q0 = 0.0f, q1 = 0.0f;
        float pt0 = 0.0f, ph0 = 0.0f, pt1 = 0.0f, ph1 = 0.0f;       // CR_POLY: z^2 and the Horner value of this step's pair
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s][0], b, s == 0 ? t4 : acc[0], 0, 0, 0);
        HAF_SB();
        if (COUNT > 0 && s == 0) { dma_piece(dma.g[FIRST], dma.l[FIRST], lane16); HAF_SB(); }
        if (ex && !CRP) { q0 = __builtin_amdgcn_exp2f(old[e0 >> 2][e0 & 3]); HAF_SB(); }
        if (ex && CRP) {                                             // the polynomial form works on this step's own pair: no exp to wait for
            const float z = old[e0 >> 2][e0 & 3];
            pt0 = z * z;
            HAF_SB();
            ph0 = fmaf(z, ba5, ba4);
            HAF_SB();
            ph0 = fmaf(ph0, z, ba3);
            HAF_SB();
        }
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s][1], b, s == 0 ? t4 : acc[1], 0, 0, 0);
        HAF_SB();
        if (COUNT > 1 && s == 0) { dma_piece(dma.g[FIRST + 1], dma.l[FIRST + 1], lane16); HAF_SB(); }
        if (ex && !CRP) { q1 = __builtin_amdgcn_exp2f(old[e1 >> 2][e1 & 3]); HAF_SB(); }
        if (ex && CRP) {
            const float z = old[e1 >> 2][e1 & 3];
            pt1 = z * z;
            HAF_SB();
            ph1 = fmaf(z, ba5, ba4);
            HAF_SB();
            ph1 = fmaf(ph1, z, ba3);
            HAF_SB();
        }
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s][2], b, s == 0 ? t4 : acc[2], 0, 0, 0);
        HAF_SB();
        if (COUNT > 2 && s == 0) { dma_piece(dma.g[FIRST + 2], dma.l[FIRST + 2], lane16); HAF_SB(); }
        if (fm && PLAIN) { sum[f0 >> 2][f0 & 3] = fmaf(cf_old, k0, sum[f0 >> 2][f0 & 3]); HAF_SB(); }
        if (fm && SUMSQ) {
            const float ck = cf_old * k0;
            HAF_SB();
            sum[f0 >> 2][f0 & 3] += ck;
            HAF_SB();
            sq[f0 >> 2][f0 & 3] = fmaf(ck, ck, sq[f0 >> 2][f0 & 3]);
            HAF_SB();
        }
        if (fm && CRE) {                                             // b psi(z) = b ((2^z - 1) - z ln2)
            const float em1 = k0 - 1.0f;
            HAF_SB();
            const float ps = fmaf(old[f0 >> 2][f0 & 3], -kLn2f, em1);
            HAF_SB();
            sum[f0 >> 2][f0 & 3] = fmaf(cf_old, ps, sum[f0 >> 2][f0 & 3]);
            HAF_SB();
        }
        if (ex && CRP) {
            ph0 = fmaf(ph0, old[e0 >> 2][e0 & 3], ba2);
            HAF_SB();
            sum[e0 >> 2][e0 & 3] = fmaf(pt0, ph0, sum[e0 >> 2][e0 & 3]);
            HAF_SB();
        }
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s][3], b, s == 0 ? t4 : acc[3], 0, 0, 0);
        HAF_SB();
        if (fm && PLAIN) { sum[f1 >> 2][f1 & 3] = fmaf(cf_old, k1, sum[f1 >> 2][f1 & 3]); HAF_SB(); }
        if (fm && SUMSQ) {
            const float ck = cf_old * k1;
            HAF_SB();
            sum[f1 >> 2][f1 & 3] += ck;
            HAF_SB();
            sq[f1 >> 2][f1 & 3] = fmaf(ck, ck, sq[f1 >> 2][f1 & 3]);
            HAF_SB();
        }
        if (fm && CRE) {
            const float em1 = k1 - 1.0f;
            HAF_SB();
            const float ps = fmaf(old[f1 >> 2][f1 & 3], -kLn2f, em1);
            HAF_SB();
            sum[f1 >> 2][f1 & 3] = fmaf(cf_old, ps, sum[f1 >> 2][f1 & 3]);
            HAF_SB();
        }
        if (ex && CRP) {
            ph1 = fmaf(ph1, old[e1 >> 2][e1 & 3], ba2);
            HAF_SB();
            sum[e1 >> 2][e1 & 3] = fmaf(pt1, ph1, sum[e1 >> 2][e1 & 3]);
            HAF_SB();
        }
        k0 = q0;
        k1 = q1;
        b = b1;
    }
    // (n == 0: b now holds fragment 0 of block 1)
#undef HAF_SB
    __builtin_amdgcn_sched_barrier(0);                               // nothing crosses from one column block into the next
}

// The decision tail of one evaluation (slot e of the operand images / bands): the two class sums (and the sum of squares) -> decision
// value, band, label; returns "undecided".  COMBINED: the sums are fp64 sums of k_svm_screen<., PART>'s partial sums (k_screen_combine).
template <int VAR, bool COMBINED>
__device__ __forceinline__ bool screen_tail_vals(double Psd, double Nsd, float qs, long e, float4 g, float4 g2, float sc,
                                                 const SvmParams &p, const CrParams &crp, const int *__restrict__ idx_list,
                                                 const int *__restrict__ evalcell, float *__restrict__ dec, int8_t *__restrict__ labels,
                                                 float *__restrict__ margin)
{
    constexpr bool SUMSQ = VAR == SCREEN_SUMSQ, CR = VAR == SCREEN_CR_EXP || VAR == SCREEN_CR_POLY;
    const float Ps = (float)Psd, Ns = (float)Nsd;
    {
        float val, err;
        if (CR) {
            // centred-remainder form: dec^ = A^ (B0 + L + R^) - rho in fp64 (B0 + L cancels against rho), R^ = the two class sums of
            // b psi(z^), S_psi^ = their difference; band {L, c_abs, k_psi, cm} from screen_finish_cr (features.hip)
            const double T = (crp.B0 + (double)g.x) + (Psd + Nsd);
            const double dvd = (double)sc * T - crp.rho;
            val = (float)dvd;
            const float spsi = COMBINED ? (float)(Psd - Nsd) * 1.0000003f : Ps - Ns;   // sum |b| psi, raw (before the common factor)
            // fp32 class sums: two-level, as in the plain variant (guard_acc0); the casts of this tail: 3 u of the terms
            const float raw = g.y + (p.guard_acc0 * 1.04f + g.z) * spsi + 1.8e-7f * (fabsf(g.x) + fabsf(Ps) + fabsf(Ns));
            err = (raw * sc * 1.002f + (g.w + 2.4e-7f) * (fabsf(val) + fabsf((float)crp.rho))) * 1.002f + p.guard_abs;
        } else {
        const float P = Ps * sc, N = Ns * sc;
        const float dv = (P + N) - p.rho;
        const float sabs = P - N;                                   // sum |coef| K
        // {gA, gB, gC, cm} (screen_finish): linear term through the spectral norms (~sqrt(S)) or per SV (~S), whichever is
        // smaller; S-proportional terms; the common factor on (|dec^| + |rho|)
        const float adv = fabsf(dv);
        // |w|_2 of w_n = coef_n K_n: measured (SUMSQ; the common factor enters squared) or bounded by sqrt(max|coef| * S)
        const float w2 = SUMSQ ? sqrtf(qs) * sc : p.sqrt_cmax * sqrtf(sabs);
        const float lin = fminf(g.x * w2, g.z * sabs);
        const float gacc = SUMSQ ? p.guard_acc0_s : p.guard_acc0;   // single- / two-level coefficient sum
        // (COMBINED: the partial class sums were added in fp64 and rounded to fp32 once more: one more unit of S)
        const float sterm = (gacc * 1.04f + g.y + (COMBINED ? 1.2e-7f : 0.0f)) * sabs;
        const float err1 = (lin + sterm + g.w * (adv + fabsf(p.rho))) * 1.002f + p.guard_abs;
        // the centred estimate (kernels.h): the first-order error of the evaluation-independent part of w is subtracted, the rest
        // of the bilinear term is the absolute bound g2.y; the two fp32 operations here go with the common factor's term
        const float corr = g2.x * sc;
        const float dvc = dv - corr;
        const float err2 = (g2.y * sc + sterm + (g.w + 2.4e-7f) * (fabsf(dvc) + fabsf(corr) + fabsf(p.rho))) * 1.002f + p.guard_abs;
        const bool centred = err2 < err1;                           // decide from the estimate with the narrower band
        val = centred ? dvc : dv;
        err = centred ? err2 : err1;
        }
        const long eid = idx_list ? (long)idx_list[e] : e;          // the evaluation this slot holds
        dec[eid] = val;
        labels[evalcell[eid]] = (int8_t)(val > 0.0f ? p.gv0 : p.gv1);
        const bool flagged = !(fabsf(val) > err);                   // also catches NaN
        if (margin) margin[eid] = flagged ? 0.0f : fabsf(val) / err;  // HAF_FLAG_KEEP_DEBUG only: how far outside its band the tier decided
        return flagged;
    }
}
template <int VAR, bool COMBINED>
__device__ __forceinline__ bool screen_tail(double Psd, double Nsd, float qs, long e, const float *__restrict__ gband, const float *__restrict__ nax,
                                            const SvmParams &p, const CrParams &crp, const int *__restrict__ idx_list,
                                            const int *__restrict__ evalcell, float *__restrict__ dec, int8_t *__restrict__ labels,
                                            float *__restrict__ margin)
{
    // the common factor 2^(-|u|^2/2) of every term of both sums (its v_exp_f32 is consumed many instructions later:
    // the LDS reads and their wait sit in between)
    float sc = __builtin_amdgcn_exp2f(nax[e]);
    const float4 g = *reinterpret_cast<const float4 *>(gband + kBandFloats * e);
    const float4 g2 = *reinterpret_cast<const float4 *>(gband + kBandFloats * e + 4);
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(sc));
    return screen_tail_vals<VAR, COMBINED>(Psd, Nsd, qs, e, g, g2, sc, p, crp, idx_list, evalcell, dec, labels, margin);
}

// number of SV ranges a request of n_evals evaluations is split over (PART form of k_svm_screen, below): as many as it takes to put
// ~512 workgroups on the chip, at most kS0MaxParts, and never more than either coefficient group has tiles
constexpr int kS0MaxParts = 16;
constexpr int kS0PartBlocks = 256;                                   // requests of up to this many workgroups take the PART form
constexpr int kS0MinPartTiles = 8;                                   // ... if the model has at least four times this many SV tiles (1024 SVs)
__device__ __forceinline__ int screen_parts(int n_evals, const Dims &d, int forced)
{
    const int blocks = (n_evals + kS0BlockEvals - 1) / kS0BlockEvals;
    int k = forced > 0 ? forced : (blocks > 0 ? 512 / blocks : 1);
    k = min(k, kS0MaxParts);
    if (forced <= 0) k = min(k, d.n_sv_tiles / kS0MinPartTiles);       // a slice worth a workgroup's prologue (A fragments, ring start)
    k = min(k, min(d.sv_tile_neg, d.n_sv_tiles - d.sv_tile_neg));
    return max(k, 1);
}

// PART = true (round 4: requests that do not fill the chip -- up to 65 536 evaluations, e.g. C3 against a model of thousands of SVs):
// the grid is (workgroups, kS0MaxParts); workgroup (x, k) sweeps the k-th slice of the non-negative tiles and the k-th slice of the
// negative ones (K = screen_parts(), from the LIVE evaluation count: no host round trip) and writes its two class sums (and the sum
// of squares) to part_out[k][evaluation]; k_screen_combine adds the K partial sums in fp64 and runs the same decision tail.
template <int VAR, bool PART>
__global__ __launch_bounds__(kS0Waves * 64, 2) void k_svm_screen(const char *__restrict__ X0, const float *__restrict__ gband,
                                                               const float *__restrict__ nax,
                                                               const char *__restrict__ svt0,
                                                               const int *__restrict__ evalcell,
                                                               const int *__restrict__ counters, SvmParams p,
                                                               float *__restrict__ dec, int8_t *__restrict__ labels,
                                                               unsigned long long *__restrict__ flag0_words, Dims d,
                                                               float *__restrict__ margin, CrParams crp,
                                                               const int *__restrict__ idx_list, int count_slot,
                                                               float4 *__restrict__ part_out, int forced_parts, int in_cap)
{
    // LIST mode (idx_list != nullptr; round 4, "tier 0b"): the operand images, bands and common factors are indexed by list slot
    // (the feature kernel's list mode wrote them), slot j holds evaluation idx_list[j], counters[count_slot] says how many
    constexpr bool SUMSQ = VAR == SCREEN_SUMSQ, CRE = VAR == SCREEN_CR_EXP, CRP = VAR == SCREEN_CR_POLY;
    // the ONLY LDS object: 3 SV tile images + per wave one row of positive-group sums and one row of final sums
    __shared__ __attribute__((aligned(16))) char lds[kS0Buffers * kS0SvTileBytes + 3 * kS0Waves * kS0WaveEvals * 4];
    // (list mode: the counter holds how many evaluations the previous pass left undecided, which may EXCEED what its list holds --
    // the host finds out after the request and redoes the decision stage; until then no consumer may walk past the list's end: the
    // entries beyond it are stale flag words and foreign memory, and evaluation ids read from there were written through)
    const int n_evals = min(counters[count_slot], in_cap);
    const long base = (long)blockIdx.x * kS0BlockEvals;
    if (base >= n_evals) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long tile32 = (base >> 5) + 2 * wave;                      // this wave's two 32-eval operand images
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    // the tiles this workgroup sweeps: ring positions 0..nt-1 hold tiles tile_of(0..nt-1), the first np of them non-negative
    int np = d.sv_tile_neg, nt = d.n_sv_tiles, p0 = 0, n0 = d.sv_tile_neg;
    if (PART) {
        const int K = screen_parts(n_evals, d, forced_parts), k = blockIdx.y;
        if (k >= K) return;
        const int nn = d.n_sv_tiles - d.sv_tile_neg;
        p0 = d.sv_tile_neg * k / K;
        np = d.sv_tile_neg * (k + 1) / K - p0;
        n0 = d.sv_tile_neg + nn * k / K;
        nt = np + (d.sv_tile_neg + nn * (k + 1) / K - n0);
    }
    auto tile_of = [&](int i) { return PART ? (i < np ? p0 + i : n0 + (i - np)) : i; };
    float *pos = reinterpret_cast<float *>(lds + kS0Buffers * kS0SvTileBytes) + wave * kS0WaveEvals;
    float *fin = pos + kS0Waves * kS0WaveEvals;
    float *qrow = fin + kS0Waves * kS0WaveEvals;                     // SUMSQ: sum of (coef K)^2 over both sweeps

    const unsigned lane16 = (unsigned)lane * 16u;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int poff[kS0WavePieces];                                         // byte offsets of this wave's five image pieces inside a tile
#pragma unroll
    for (int q = 0; q < kS0WavePieces; q++) poff[q] = (wave_u + kS0Waves * q) * 1024;    // pieces 0..19, each exactly once
    static_assert(kS0Waves * kS0WavePieces * 1024 == kS0MatBytes && kS0MatBytes + 1024 == kS0SvTileBytes, "20 image pieces + 1 tail piece");
    {
        const char *g0 = svt0 + (size_t)tile_of(0) * kS0SvTileBytes;
        stage_sv_tile_s0(tile_dma(g0, lds0, poff), lane16);                                                    // tile 0
        if (wave_u == 0) dma_piece(g0 + kS0MatBytes, lds0 + kS0MatBytes, lane16);
    }
    if (nt > 1) {                                                                                              // tile 1
        const char *g1 = svt0 + (size_t)tile_of(1) * kS0SvTileBytes;
        stage_sv_tile_s0(tile_dma(g1, lds0 + kS0SvTileBytes, poff), lane16);
        if (wave_u == 0) dma_piece(g1 + kS0MatBytes, lds0 + kS0SvTileBytes + kS0MatBytes, lane16);
    }

    // A fragments: row block m = 0..3 (rows 16m..16m+15 of the wave's 64); lane holds A[16m + (lane&15)][32s + 8(lane>>4) + j]
    // read once, with the nt hint: 5 GB of operands stream past the 2.8 MB of SV tiles that every workgroup re-reads from L2
#define SCREEN_A_LOAD(p) __builtin_nontemporal_load(p)
    half8 a[kHFull][4];
    {
        const char *xt = X0 + (size_t)tile32 * kS0MatBytes;
#pragma unroll
        for (int s = 0; s < kHFull; s++)
#pragma unroll
            for (int m = 0; m < 4; m++)
                a[s][m] = SCREEN_A_LOAD(reinterpret_cast<const half8 *>(xt + (m >> 1) * kS0MatBytes + (s * 2 + (m & 1)) * 1024 + lane * 16));
    }
    // pin the compiler-issued loads before any further (asm, uncounted) DMA is queued behind them (see k_svm_rbf)
#pragma unroll
    for (int s = 0; s < kHFull; s++)
#pragma unroll
        for (int m = 0; m < 4; m++) asm volatile("" : "+v"(a[s][m]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // tiles 0 and 1 (this wave's pieces) have landed
    __syncthreads();

    float sum[4][4];                                                 // rows 16m + 4(lane>>4) + r, this lane's columns
    // Plain variant: TWO-LEVEL coefficient sum.  `sum` takes the products of at most eight tiles (16 fmas), then folds into
    // `part`: a term passes through 16 + tiles/8 roundings instead of 2 * tiles, and the worst-case bound of this fp32 sum --
    // guard_acc0, the largest single term of the screening band at 4096 SVs -- shrinks from 268 u to 41 u (refined share at C5:
    // 2.05 % -> see DESIGN.md 5).  The SUMSQ variant has no registers for it (its 16 go to the squares): single level there.
    float part[4][4];
    float sq[4][4];                                                  // SUMSQ only (dead otherwise): runs on across the two sweeps
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int r = 0; r < 4; r++) sq[m][r] = 0.0f;
    f32x4 acc0[4], acc1[4];
    // Two sweeps: the tile images hold the non-negative coefficients first, so the first sweep yields
    // P = sum_{coef>0} coef*K and the second N = sum_{coef<0} coef*K; dec = P + N - rho and the guard scale
    // sum|coef|K = P - N come from the same registers.  The DMA ring runs on across the two sweeps.
    for (int ph = 0; ph < 2; ph++) {
        const int t_end = ph ? nt : np;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            acc1[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int r = 0; r < 4; r++) { sum[m][r] = 0.0f; part[m][r] = 0.0f; }
        }
        int fold = 0;
        float cf_prev = 0.0f;                                        // the first deferred epilogue adds 0 * exp2(0)
        for (int t = ph ? np : 0; t < t_end; t++) {
            const char *cur = lds + (t % kS0Buffers) * kS0SvTileBytes;
            // always the same DMA pieces per wave and tile, so the wait below is one constant per wave: past the last tile the
            // ring slot that nobody reads any more is refilled with tile (t+2) mod nt
            const int tn = tile_of((t + 2) % nt);
            const TileDma dma = tile_dma(svt0 + (size_t)tn * kS0SvTileBytes, lds0 + ((t + 2) % kS0Buffers) * kS0SvTileBytes, poff);
            // wave 0: the tail piece of that tile, here -- in front of the MFMA stream, where a branch does no harm
            if (wave_u == 0) dma_piece(svt0 + (size_t)tn * kS0SvTileBytes + kS0MatBytes,
                                       lds0 + ((t + 2) % kS0Buffers) * kS0SvTileBytes + kS0MatBytes, lane16);
            const float *tt = reinterpret_cast<const float *>(cur + kS0MatBytes);
            const float t0 = tt[lane & 15], t1 = tt[16 + (lane & 15)];   // t_n of this lane's column in either block
            const float cf0 = tt[32 + (lane & 15)], cf1 = tt[48 + (lane & 15)];   // and its coefficient (0 for padding SVs)
            // block 0 | epilogue of the previous tile's block 1, then block 1 | epilogue of block 0
            half8 bf0, bf1;                                          // B fragments in flight, handed from block 0 to block 1
            screen_block<0, 3, VAR>(cur, 0, lane, a, acc0, acc1, t0, cf_prev, sum, sq, dma, lane16, bf0, bf1);
            screen_block<3, 2, VAR>(cur, 1, lane, a, acc1, acc0, t1, cf0, sum, sq, dma, lane16, bf0, bf1);
            cf_prev = cf1;
            if (!SUMSQ && ++fold == 8) {                             // wave-uniform, outside the MFMA stream
                fold = 0;
#pragma unroll
                for (int m = 0; m < 4; m++)
#pragma unroll
                    for (int r = 0; r < 4; r++) { part[m][r] += sum[m][r]; sum[m][r] = 0.0f; }
            }
            // tile t+1 must have landed before anyone reads it.  ALL of this wave's pieces are waited for, those of tile t+2 (issued
            // behind the first MFMAs of this tile's blocks, a whole tile ago) included: a counted wait -- vmcnt(5), "the five just
            // issued may stay in flight" -- assumes that LDS-DMA loads complete in issue order, and the bulk three-pass kernel has
            // shown that they do not ("Counted waits on LDS-DMA do not hold").  Measured free: 13.59 against 13.65 ms.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");                          // no LDS read of the next tile may move above the barrier
        }
        // epilogue of the sweep's last block, then the sum over the 16 column lanes
        float *dst = ph ? fin : pos;
        // (all sixteen v_exp_f32 first, their consumers behind wait states that hang on the data: see the hazard note)
        f32x4 zz[4];                                                 // CR_EXP: the arguments, still needed behind the exps
#pragma unroll
        for (int m = 0; m < 4; m++) zz[m] = acc1[m];
        if (!CRP) {
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc1[m][r] = __builtin_amdgcn_exp2f(acc1[m][r]);
            asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc1[0]), "+v"(acc1[1]), "+v"(acc1[2]), "+v"(acc1[3]));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float ck;
                if (CRE) {
                    ck = cf_prev * fmaf(zz[m][r], -kLn2f, acc1[m][r] - 1.0f);
                } else if (CRP) {
                    const float z = zz[m][r];
                    ck = (cf_prev * (z * z)) * fmaf(fmaf(fmaf(z, kPsiA5, kPsiA4), z, kPsiA3), z, kPsiA2);
                } else {
                    ck = cf_prev * acc1[m][r];
                }
                float v = ck + sum[m][r];
                if (!SUMSQ) v += part[m][r];
                if (SUMSQ) sq[m][r] = fmaf(ck, ck, sq[m][r]);
                v += __shfl_xor(v, 8, 64);
                v += __shfl_xor(v, 4, 64);
                v += __shfl_xor(v, 2, 64);
                v += __shfl_xor(v, 1, 64);
                if ((lane & 15) == 0) dst[16 * m + 4 * (lane >> 4) + r] = v;
            }
    }
    if (SUMSQ) {
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float v = sq[m][r];
                v += __shfl_xor(v, 8, 64);
                v += __shfl_xor(v, 4, 64);
                v += __shfl_xor(v, 2, 64);
                v += __shfl_xor(v, 1, 64);
                if ((lane & 15) == 0) qrow[16 * m + 4 * (lane >> 4) + r] = v;
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the refills past the last tile
    __syncthreads();
    // one evaluation per lane from here: coalesced stores, one flag word per wave
    const long e = base + wave * kS0WaveEvals + lane;
    if (PART) {                                                      // partial sums of this slice; the tail runs in k_screen_combine
        part_out[(size_t)blockIdx.y * ((size_t)gridDim.x * kS0BlockEvals) + e] = float4{pos[lane], fin[lane], SUMSQ ? qrow[lane] : 0.0f, 0.0f};
        return;
    }
    const bool live = e < n_evals;
    bool flagged = false;
    if (live) flagged = screen_tail<VAR, false>((double)pos[lane], (double)fin[lane], SUMSQ ? qrow[lane] : 0.0f, e, gband, nax, p, crp, idx_list,
                                                evalcell, dec, labels, margin);
    // one 64-bit word per wave (64 consecutive evaluations): k_screen_compact turns the words into the ORDERED list of
    // undecided evaluations -- neighbours in the list are neighbours on the grid, so the feature kernel that follows reads
    // overlapping windows, and the list (hence every later tile) is the same from run to run
    const unsigned long long bal = __ballot(flagged);
    if (lane == 0) flag0_words[(base >> 6) + wave] = bal;
}

// The K partial sums of k_svm_screen<VAR, true> -> decision, band, label, flag words: one evaluation per thread, workgroups of 256 like
// the tail of the unsplit kernel (same flag-word layout).  The partial sums are added in fp64, in the order of the slices.
template <int VAR>
__global__ __launch_bounds__(kS0BlockEvals) void k_screen_combine(const float4 *__restrict__ part_out, const float *__restrict__ gband,
                                                                  const float *__restrict__ nax, const int *__restrict__ evalcell,
                                                                  const int *__restrict__ counters, SvmParams p, float *__restrict__ dec,
                                                                  int8_t *__restrict__ labels, unsigned long long *__restrict__ flag0_words,
                                                                  Dims d, float *__restrict__ margin, CrParams crp,
                                                                  const int *__restrict__ idx_list, int count_slot, int forced_parts, int in_cap)
{
    const int n_evals = min(counters[count_slot], in_cap);
    const long base = (long)blockIdx.x * kS0BlockEvals;
    if (base >= n_evals) return;
    const long e = base + threadIdx.x;
    const int K = screen_parts(n_evals, d, forced_parts);
    const size_t stride = (size_t)gridDim.x * kS0BlockEvals;
    bool flagged = false;
    if (e < n_evals) {
        double Ps = 0.0, Ns = 0.0, qs = 0.0;
        for (int k = 0; k < K; k++) {
            const float4 v = part_out[(size_t)k * stride + e];
            Ps += (double)v.x;
            Ns += (double)v.y;
            qs += (double)v.z;
        }
        flagged = screen_tail<VAR, true>(Ps, Ns, (float)qs, e, gband, nax, p, crp, idx_list, evalcell, dec, labels, margin);
    }
    const unsigned long long bal = __ballot(flagged);
    if ((threadIdx.x & 63) == 0) flag0_words[(base >> 6) + (threadIdx.x >> 6)] = bal;
}

// Ordered compaction of the screening pass's flag words in two small launches: k_screen_count sums the popcounts of
// 256 words per workgroup; k_screen_compact gives every word its slot (sum of the preceding workgroups' counts + a
// workgroup scan) and writes the set bits in ascending order.  counters[CNT_FLAGGED0] receives the total, which may exceed
// the list's capacity (the host then falls back to the three-pass kernel for everything).
constexpr int kCompactWords = 256;

__device__ __forceinline__ int screen_words(const int *counters, int count_slot, int in_cap)
{
    return (min(counters[count_slot], in_cap) + kS0BlockEvals - 1) / kS0BlockEvals * (kS0BlockEvals / 64);
}

__global__ __launch_bounds__(kCompactWords) void k_screen_count(const unsigned long long *__restrict__ words,
                                                                int *__restrict__ wgcount, const int *__restrict__ counters, int count_slot, int in_cap)
{
    __shared__ int red[kCompactWords / 64];
    const int n_words = screen_words(counters, count_slot, in_cap);
    const int w = blockIdx.x * kCompactWords + threadIdx.x;
    int c = (w < n_words) ? __popcll(words[w]) : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) wgcount[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(kCompactWords) void k_screen_compact(const unsigned long long *__restrict__ words,
                                                                  const int *__restrict__ wgcount, int n_wg,
                                                                  int *__restrict__ list, int cap, int *__restrict__ counters, int also_counter,
                                                                  const int *__restrict__ idx_list, int count_slot, int out_slot, int in_cap)
{
    __shared__ int part[kCompactWords];
    __shared__ int s_base;
    const int n_words = screen_words(counters, count_slot, in_cap);
    const int t = threadIdx.x;
    // slots taken by the preceding workgroups (the last workgroup also publishes the total)
    int before = 0;
    const int upto = (blockIdx.x == gridDim.x - 1) ? n_wg : blockIdx.x;
    int total = 0;
    for (int j = t; j < upto; j += kCompactWords) {
        const int c = wgcount[j];
        total += c;
        if (j < (int)blockIdx.x) before += c;
    }
    part[t] = before;
    __syncthreads();
    for (int o = kCompactWords / 2; o > 0; o >>= 1) {
        if (t < o) part[t] += part[t + o];
        __syncthreads();
    }
    if (t == 0) s_base = part[0];
    __syncthreads();
    if (blockIdx.x == gridDim.x - 1) {
        part[t] = total;
        __syncthreads();
        for (int o = kCompactWords / 2; o > 0; o >>= 1) {
            if (t < o) part[t] += part[t + o];
            __syncthreads();
        }
        if (t == 0) {
            counters[out_slot] = part[0];
            if (also_counter >= 0) counters[also_counter] = min(part[0], cap);    // small requests: the list IS the exact tier's (engine.cpp)
        }
        __syncthreads();
    }
    const int w = blockIdx.x * kCompactWords + t;
    unsigned long long m = (w < n_words) ? words[w] : 0ull;
    const int cnt = __popcll(m);                                      // (the flag words of a partly filled block hold no bits beyond the live count: screen_tail)
    part[t] = cnt;
    __syncthreads();
    for (int o = 1; o < kCompactWords; o <<= 1) {
        int v = (t >= o) ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int slot = s_base + part[t] - cnt;
    while (m) {
        const int b = __ffsll((long long)m) - 1;
        m &= m - 1;
        if (slot < cap) list[slot] = idx_list ? idx_list[w * 64 + b] : w * 64 + b;   // (list mode: slots back to evaluations, still in ascending order)
        slot++;
    }
}

// Small requests -- at most kListCompactThreads flag words, 65 536 evaluations: the reference's own grids -- count, scan and ordered
// write in ONE workgroup, a word per thread (round 5: the two launches above were 4.5 + 6.7 us and a gap of C3's 0.26 ms).  Same list.
__global__ __launch_bounds__(kListCompactThreads) void k_screen_compact_small(const unsigned long long *__restrict__ words, int *__restrict__ list,
                                                                              int cap, int *__restrict__ counters, int also_counter,
                                                                              const int *__restrict__ idx_list, int count_slot, int out_slot, int in_cap)
{
    __shared__ int s_wave[kListCompactThreads / 64];
    const int n_words = screen_words(counters, count_slot, in_cap);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    unsigned long long m = (t < n_words) ? words[t] : 0ull;
    const int cnt = __popcll(m);
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kListCompactThreads / 64; k++) {
        const int c = s_wave[k];
        if (k < wave) before += c;
        total += c;
    }
    int slot = before + incl - cnt;
    while (m) {
        const int b = __ffsll((long long)m) - 1;
        m &= m - 1;
        if (slot < cap) list[slot] = idx_list ? idx_list[t * 64 + b] : t * 64 + b;
        slot++;
    }
    if (t == 0) {
        counters[out_slot] = total;
        if (also_counter >= 0) counters[also_counter] = min(total, cap);
    }
}

// the flag words of every workgroup that can hold evaluations (the kernels clip to the live ones) -> the ordered list
static void launch_screen_compaction(const unsigned long long *flag0_words, int *wgcount, long blocks, int *flag0_list, int flag0_cap,
                                     int *counters_rw, int also_counter, const int *idx_list, int count_slot, int out_slot, int in_cap,
                                     hipStream_t s)
{
    const long words = blocks * (kS0BlockEvals / 64);
    if (words <= kListCompactThreads) {
        hipLaunchKernelGGL(k_screen_compact_small, dim3(1), dim3(kListCompactThreads), 0, s, flag0_words, flag0_list, flag0_cap, counters_rw,
                           also_counter, idx_list, count_slot, out_slot, in_cap);
        return;
    }
    const int n_wg = (int)((words + kCompactWords - 1) / kCompactWords);
    hipLaunchKernelGGL(k_screen_count, dim3(n_wg), dim3(kCompactWords), 0, s, flag0_words, wgcount, counters_rw, count_slot, in_cap);
    hipLaunchKernelGGL(k_screen_compact, dim3(n_wg), dim3(kCompactWords), 0, s, flag0_words, wgcount, n_wg, flag0_list, flag0_cap,
                       counters_rw, also_counter, idx_list, count_slot, out_slot, in_cap);
}

// Does the matrix core take fp16 subnormal A operands at their value?  The screening feature kernel stores u^ = fp16(u') without
// flushing small magnitudes (two vector instructions per attribute saved) and counts |u^ - u'| from the stored value; that is
// only right if the MFMA multiplies what is stored.  haf_create runs this once per process and refuses a device that flushes
// (build with -DHAF_FLUSH_F16_SUBNORMALS there).  All A elements 2^-20 (subnormal), all B elements 2^4: every output element is
// K * 2^-16, exactly.
__global__ __launch_bounds__(64) void k_probe_f16_subnormal(float *out)
{
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    const _Float16 tiny = (_Float16)9.5367431640625e-07f, big = (_Float16)16.0f;
    h8 a, b;
    h4 at, bt;
    for (int i = 0; i < 8; i++) { a[i] = tiny; b[i] = big; }
    for (int i = 0; i < 4; i++) { at[i] = tiny; bt[i] = big; }
    asm volatile("" : "+v"(a), "+v"(b), "+v"(at), "+v"(bt));          // no constant folding
    f4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    const f4 c32 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, z, 0, 0, 0);
    const f4 c16 = __builtin_amdgcn_mfma_f32_16x16x16f16(at, bt, z, 0, 0, 0);
    bool ok = true;
    for (int r = 0; r < 4; r++) ok = ok && c32[r] == 32.0f * 1.52587890625e-05f && c16[r] == 16.0f * 1.52587890625e-05f;
    const unsigned long long all = __ballot(ok);
    if (threadIdx.x == 0) out[0] = (all == ~0ull) ? 1.0f : 0.0f;
}

int probe_f16_subnormal_mfma(hipStream_t s)
{
    float *d = nullptr, h = -1.0f;
    if (hipMalloc((void **)&d, sizeof(float)) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_probe_f16_subnormal, dim3(1), dim3(64), 0, s, d);
    const bool fine = hipMemcpyAsync(&h, d, sizeof(float), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    (void)hipFree(d);
    if (!fine) return -1;
    return h == 1.0f ? 1 : 0;
}

// How large is the error of ONE v_mfma_f32_16x16x32_f16?  Per output element the instruction forms c + sum_{k<32} a_k b_k; the
// products are exact in fp32, the 33-term sum is not, and the architecture manuals do not say how it is rounded.  Measured
// (tools/mfma_rounding_probe.py, profiles/r03_mfma_rounding.json): a single large product keeps the sum of 31 tiny ones (no
// sequential rounding), results are within 2 ulp whenever nothing cancels, and against scale = |c| + sum|a_k b_k| the error
// reaches 5.3 u (u = 2^-24) when two large products cancel over thirty small ones: the terms are aligned to the largest exponent
// and truncated with a few guard bits.  The guard bands of the fp16 tiers take  |error| <= kappa u scale  per instruction with
// kappa = max(12, 1.5 x the largest ratio this probe sees) -- on the device and in the process the engine is created in, over the
// same adversarial families (seeded, 64 trials each: 147 456 sums, and as many of the 16-wide shape of the three-pass kernel's K
// tail on the first sixteen products of the same data), so a matrix core that rounds worse than the one the
// constants were chosen on widens the bands by itself.  A measured property with a margin, not a theorem: DESIGN.md 2 says so.
// Round 4: families 7 and 8 (one term of order 1 over 31 dense-mantissa products) were added after an adversarial search in the
// tests found 9.0 u where the first seven families reach 5.5 u: the alignment drops about a quarter of a unit per term (two guard
// bits), 31 x 0.25 + the final rounding.  That is also the largest value the two-guard-bit reading allows (33 x 0.25 + 1 ~ 9.3).
__global__ __launch_bounds__(64) void k_probe_mfma_rounding(const _Float16 *__restrict__ a, const _Float16 *__restrict__ b,
                                                            const float *__restrict__ c, float *__restrict__ d, float *__restrict__ d16)
{
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int t = blockIdx.x, l = threadIdx.x, rc = l & 15, kb = l >> 4;
    a += (size_t)t * 512; b += (size_t)t * 512; c += (size_t)t * 256; d += (size_t)t * 256; d16 += (size_t)t * 256;
    h8 fa, fb;
    for (int j = 0; j < 8; j++) {                    // A[row][k] row-major, B[k][col] row-major
        fa[j] = a[rc * 32 + 8 * kb + j];
        fb[j] = b[(8 * kb + j) * 16 + rc];
    }
    f4 acc;
    for (int r = 0; r < 4; r++) acc[r] = c[(4 * kb + r) * 16 + rc];
    const f4 c0 = acc;
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) d[(4 * kb + r) * 16 + rc] = acc[r];
    // the 16-wide shape of the three-pass kernel's K tail on the first sixteen products of the same data: lane holds k = 4 kb + j
    h4 ga, gb;
    for (int j = 0; j < 4; j++) {
        ga[j] = a[rc * 32 + 4 * kb + j];
        gb[j] = b[(4 * kb + j) * 16 + rc];
    }
    const f4 t16 = __builtin_amdgcn_mfma_f32_16x16x16f16(ga, gb, c0, 0, 0, 0);
    for (int r = 0; r < 4; r++) d16[(4 * kb + r) * 16 + rc] = t16[r];
}

// returns the largest |d - exact| / (u (|c| + sum|a_k b_k|)) over the families for the 32-wide shape (the screening kernel's only one)
// and, in *worst16, for the 16-wide shape; a negative number when HIP fails
double probe_mfma_rounding(hipStream_t s, double *worst16)
{
    constexpr int kFam = 9, kTrials = 64, T = kFam * kTrials;
    std::vector<_Float16> A((size_t)T * 512), B((size_t)T * 512);
    std::vector<float> Cm((size_t)T * 256), D((size_t)T * 256), D16((size_t)T * 256);
    unsigned long long st = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (double)(st >> 11) * (1.0 / 9007199254740992.0); };   // [0, 1)
    auto sgn = [&]() { return rnd() < 0.5 ? -1.0 : 1.0; };
    for (int f = 0; f < kFam; f++)
        for (int tr = 0; tr < kTrials; tr++) {
            const int t = f * kTrials + tr;
            _Float16 *a = A.data() + (size_t)t * 512, *b = B.data() + (size_t)t * 512;
            float *c = Cm.data() + (size_t)t * 256;
            for (int i = 0; i < 256; i++) c[i] = 0.0f;
            if (f == 0) {                            // random signs and magnitudes over 2^-3 .. 2^3, c comparable
                for (int i = 0; i < 512; i++) { a[i] = (_Float16)(sgn() * (0.25 + rnd()) * std::ldexp(1.0, (int)(rnd() * 7) - 3)); b[i] = (_Float16)(sgn() * (0.25 + rnd()) * std::ldexp(1.0, (int)(rnd() * 7) - 3)); }
                for (int i = 0; i < 256; i++) c[i] = (float)(sgn() * rnd() * 16.0);
            } else if (f <= 3) {                     // one product in [1, 2), 31 of one sign around 2^-22 / 2^-20 / 2^-24: nothing on a grid
                const int lo = f == 1 ? -22 : f == 2 ? -20 : -24, big = (5 * tr) % 32;
                for (int i = 0; i < 512; i++) a[i] = (_Float16)1.0f;
                for (int k = 0; k < 32; k++)
                    for (int j = 0; j < 16; j++) b[k * 16 + j] = (_Float16)((1.0 + rnd()) * (k == big ? 1.0 : std::ldexp(1.0, lo)));
            } else if (f == 4) {                     // the accumulator dominates, 32 small products of one sign
                const double sg = (tr & 1) ? -1.0 : 1.0;
                for (int i = 0; i < 512; i++) { a[i] = (_Float16)1.0f; b[i] = (_Float16)(sg * (1.0 + rnd()) * std::ldexp(1.0, -22)); }
                for (int i = 0; i < 256; i++) c[i] = (float)(1.0 + rnd());
            } else if (f == 5) {                     // two large products that cancel, thirty small ones
                const int p0 = (3 * tr) % 32, p1 = (p0 + 1 + tr % 7) % 32;
                for (int i = 0; i < 512; i++) a[i] = (_Float16)1.0f;
                for (int k = 0; k < 32; k++)
                    for (int j = 0; j < 16; j++) b[k * 16 + j] = (_Float16)(k == p0 ? 1024.0 : k == p1 ? -1024.0 : (0.5 + 0.5 * rnd()) * std::ldexp(1.0, -13));
            } else if (f == 6) {                     // |c| = 2^12 against products of order 1
                for (int i = 0; i < 512; i++) { a[i] = (_Float16)(sgn() * (0.5 + rnd())); b[i] = (_Float16)(sgn() * (0.5 + rnd())); }
                for (int i = 0; i < 256; i++) c[i] = (float)(sgn() * 4096.0 * (1.0 + rnd()));
            } else {
                // Round 4 (found by the adversarial search of tests/: 9.0 where the seven families above reach 5.5): ONE term of order 1
                // -- a product (f == 7) or the accumulator (f == 8) -- and 31 / 32 products of one sign, 2^-14 .. 2^-29 of it, whose
                // factors have DENSE mantissas in [1.5, 2): every product carries 22 significant bits, and an adder that aligns the
                // terms to the largest exponent and drops what lies below a couple of guard bits loses almost a full unit per term
                const int e = -14 - (tr % 16);
                const int ea = e / 2, eb = e - ea;
                const double sg = (tr & 16) ? -1.0 : 1.0;
                for (int i = 0; i < 512; i++) {
                    a[i] = (_Float16)((1.5 + 0.5 * rnd()) * std::ldexp(1.0, ea));
                    b[i] = (_Float16)(sg * (1.5 + 0.5 * rnd()) * std::ldexp(1.0, eb));
                }
                if (f == 7) {
                    for (int i = 0; i < 16; i++) a[i * 32] = (_Float16)1.0f;                   // A[row][k = 0]
                    for (int j = 0; j < 16; j++) b[j] = (_Float16)(1.0 + std::floor(rnd() * 1024.0) / 1024.0);   // B[k = 0][col]
                } else {
                    for (int i = 0; i < 256; i++) c[i] = (float)(1.0 + rnd());
                }
            }
        }
    _Float16 *da = nullptr, *db = nullptr;
    float *dc = nullptr, *dd = nullptr, *dd16 = nullptr;
    bool ok = hipMalloc((void **)&da, A.size() * 2) == hipSuccess && hipMalloc((void **)&db, B.size() * 2) == hipSuccess &&
              hipMalloc((void **)&dc, Cm.size() * 4) == hipSuccess && hipMalloc((void **)&dd, D.size() * 4) == hipSuccess &&
              hipMalloc((void **)&dd16, D16.size() * 4) == hipSuccess;
    ok = ok && hipMemcpyAsync(da, A.data(), A.size() * 2, hipMemcpyHostToDevice, s) == hipSuccess &&
         hipMemcpyAsync(db, B.data(), B.size() * 2, hipMemcpyHostToDevice, s) == hipSuccess &&
         hipMemcpyAsync(dc, Cm.data(), Cm.size() * 4, hipMemcpyHostToDevice, s) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_probe_mfma_rounding, dim3(T), dim3(64), 0, s, da, db, dc, dd, dd16);
        ok = hipMemcpyAsync(D.data(), dd, D.size() * 4, hipMemcpyDeviceToHost, s) == hipSuccess &&
             hipMemcpyAsync(D16.data(), dd16, D16.size() * 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    }
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dc); (void)hipFree(dd); (void)hipFree(dd16);
    if (!ok) return -1.0;
    // the exact sums: products of two fp16 numbers are exact in fp64; 33 of them in long double (64 significant bits) are exact to
    // 2^-58 of the scale, far below the 2^-24 being measured
    double worst = 0.0, w16 = 0.0;
    for (int t = 0; t < T; t++)
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 16; j++) {
                long double sum = (long double)Cm[(size_t)t * 256 + i * 16 + j];
                double scale = std::fabs((double)Cm[(size_t)t * 256 + i * 16 + j]);
                for (int k = 0; k < 32; k++) {
                    const double pr = (double)(float)A[(size_t)t * 512 + i * 32 + k] * (double)(float)B[(size_t)t * 512 + k * 16 + j];
                    sum += (long double)pr;
                    scale += std::fabs(pr);
                    if (k == 15) {                                                    // the 16-wide shape: the first sixteen products
                        const double e16 = std::fabs((double)((long double)D16[(size_t)t * 256 + i * 16 + j] - sum));
                        if (!(e16 == e16)) return 1e30;
                        if (scale > 0.0) w16 = std::max(w16, e16 / (5.9604644775390625e-08 * scale));
                    }
                }
                const double err = std::fabs((double)((long double)D[(size_t)t * 256 + i * 16 + j] - sum));
                if (!(err == err)) return 1e30;                                       // NaN: never trusted
                if (scale > 0.0) worst = std::max(worst, err / (5.9604644775390625e-08 * scale));
            }
    *worst16 = w16;
    return worst;
}

// ---------------------------------------------------------------------------------------------------
// Low-rank form (kernels.h: kLrK; DESIGN.md 2, "projected operand").  k_project: Y = fp16(B^' P^) on the matrix core -- per 32
// output slots one "projection tile" (the rows of B^' as a 10-step fp16 image, exactly the layout of an SV tile of the 10-step
// form) through a double-buffered LDS slot; the tile's fragments are the A operand (rows = output slots), the evaluations' 10-step
// images the B operand (columns = evaluations), so D[output slot][evaluation] comes out with FOUR CONSECUTIVE output slots per lane
// for evaluation lane & 15: the host permutes the rows of B^' such that MFMA row 4g + r of block 2s' (2s' + 1) is output slot
// 32 s' + 8 g + r (+ 4), and the two blocks of a pair are, converted to fp16, exactly the lane's 8 halves of the 6-step A-operand
// image the sweep reads (h_image_offset): no transpose.  |y^ - y32|^2 per evaluation (exact differences, fp32 sum) goes to raw[5].
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kS0Waves * 64, 2) void k_project(const char *__restrict__ X0, const char *__restrict__ btiles, char *__restrict__ Y,
                                                            float *__restrict__ raw, const int *__restrict__ counters)
{
    __shared__ __attribute__((aligned(16))) char lds[2 * kLrProjTileBytes];
    const int n_evals = counters[CNT_EVALS];
    const long base = (long)blockIdx.x * kS0BlockEvals;
    if (base >= n_evals) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long tile32 = (base >> 5) + 2 * wave;
    half8 a[kHFull][4];
    {
        const char *xt = X0 + (size_t)tile32 * kS0MatBytes;
#pragma unroll
        for (int s = 0; s < kHFull; s++)
#pragma unroll
            for (int m = 0; m < 4; m++)
                a[s][m] = __builtin_nontemporal_load(reinterpret_cast<const half8 *>(xt + (m >> 1) * kS0MatBytes + (s * 2 + (m & 1)) * 1024 + lane * 16));
    }
    // tile 0 into slot 0 (plain loads + stores: 20 KiB per workgroup and tile, 5 x 16 bytes per thread)
    auto stage = [&](int t, int slot) {
        const char *g = btiles + (size_t)t * kLrProjTileBytes;
        char *l = lds + slot * kLrProjTileBytes;
#pragma unroll
        for (int q = 0; q < kLrProjTileBytes / (kS0Waves * 64 * 16); q++) {
            const int o = (q * kS0Waves * 64 + tid) * 16;
            *reinterpret_cast<float4 *>(l + o) = *reinterpret_cast<const float4 *>(g + o);
        }
    };
    stage(0, 0);
    __syncthreads();
    float sdy[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    char *yt = Y + (size_t)tile32 * kLrMatBytes;
    for (int sp = 0; sp < kLrSteps; sp++) {
        if (sp + 1 < kLrSteps) stage(sp + 1, (sp + 1) & 1);
        const char *cur = lds + (sp & 1) * kLrProjTileBytes;
        f32x4 acc[2][4];
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int m = 0; m < 4; m++) acc[n][m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s = 0; s < kHFull; s++) {
#pragma unroll
            for (int n = 0; n < 2; n++) {
                const half8 bf = *reinterpret_cast<const half8 *>(cur + (s * 2 + n) * 1024 + lane * 16);
#pragma unroll
                for (int m = 0; m < 4; m++) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf, a[s][m], acc[n][m], 0, 0, 0);
            }
        }
        // lane (g = lane >> 4, e = lane & 15): output slots 32 sp + 8 g + 0..3 (block 0) and + 4..7 (block 1) of evaluation 16 m + e
#pragma unroll
        for (int m = 0; m < 4; m++) {
            half8 h;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float y0 = acc[0][m][r], y1 = acc[1][m][r];
                const _Float16 h0 = (_Float16)y0, h1 = (_Float16)y1;     // RN; subnormals stay (probe_f16_subnormal_mfma)
                h[r] = h0; h[4 + r] = h1;
                const float d0 = (float)h0 - y0, d1 = (float)h1 - y1;    // exact
                sdy[m] = fmaf(d0, d0, sdy[m]);
                sdy[m] = fmaf(d1, d1, sdy[m]);
            }
            __builtin_nontemporal_store(h, reinterpret_cast<half8 *>(yt + (m >> 1) * kLrMatBytes + (sp * 2 + (m & 1)) * 1024 + lane * 16));
        }
        __syncthreads();                                                 // the next tile has been stored; this one may be overwritten
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
        float v = sdy[m];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        const long e = base + wave * kS0WaveEvals + 16 * m + (lane & 15);
        if (lane < 16) raw[kBandFloats * e + 5] = v;
    }
}

void launch_project(const void *X0, const void *btiles, void *Y, float *raw, const int *counters, long max_evals, hipStream_t s)
{
    const long blocks = (max_evals + kS0BlockEvals - 1) / kS0BlockEvals;
    if (blocks <= 0) return;
    hipLaunchKernelGGL(k_project, dim3((unsigned)blocks), dim3(kS0Waves * 64), 0, s, (const char *)X0, (const char *)btiles, (char *)Y, raw, counters);
}

#ifndef HAF_LR_WGS
#define HAF_LR_WGS 2
#endif
#ifndef HAF_LR_EXPLAG
#define HAF_LR_EXPLAG 4
#endif
// HAF_LR_ABL (experiment builds only, haf_grasping_amd/build.py: build_variant; results are garbage, only the kernel's time means anything):
// bit 0 no in-loop LDS-DMA, 1 no epilogue VALU, 2 no B-fragment reads, 3 no tile barrier / wait, 4 no projection MFMAs
#ifndef HAF_LR_ABL
#define HAF_LR_ABL 0
#endif
constexpr int kLrExpLag = HAF_LR_EXPLAG;        // MFMAs between an element's v_exp_f32 and its first consumer (>= 2: ten instructions)
constexpr bool kLrTwoLevel = HAF_LR_WGS < 3;   // three workgroups per CU need the sixteen registers of the second summation level (its band term is relative to S_psi: small either way)
// One column block (16 SVs) of the 6-step sweep: 24 MFMAs; the epilogue of the PREVIOUS block's 16 elements rides between them --
// element j's v_exp_f32 behind MFMA j (j < 16), its three VALU instructions behind MFMA j + 4 (a whole k-step later: the hazard note
// above); the polynomial form has no exp and puts element j's five instructions behind MFMA j + 2.
template <int FIRST, int COUNT, int VAR>
__device__ __forceinline__ void screen_block_lr(const char *cur, int n, int lane, const half8 (&a)[kLrSteps][4], f32x4 (&acc)[4],
                                                const f32x4 (&old)[4], float cf_old, float (&sum)[4][4],
                                                const TileDma &dma, unsigned lane16, half8 &b, half8 &b1)
{
    constexpr bool CRP = VAR == SCREEN_CR_POLY, PLN = VAR == SCREEN_PLAIN;
    static_assert(VAR == SCREEN_CR_EXP || VAR == SCREEN_CR_POLY || VAR == SCREEN_PLAIN, "the low-rank form serves the plain and the centred-remainder variants");
    const char *bl = cur + n * 1024 + lane * 16;
    __builtin_amdgcn_sched_barrier(0);
    const f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
    const float ba2 = CRP ? cf_old * kPsiA2 : 0.0f, ba3 = CRP ? cf_old * kPsiA3 : 0.0f, ba4 = CRP ? cf_old * kPsiA4 : 0.0f;
    if (n == 0) b = *reinterpret_cast<const half8 *>(bl);
    if (HAF_LR_ABL & 2) asm volatile("" :: "v"(old[0]), "v"(old[1]), "v"(old[2]), "v"(old[3]));
    float kq[16];                                                    // exp2 results in flight (CR_EXP): at most five live at a time
#define HAF_SB() __builtin_amdgcn_sched_barrier(0)
#pragma unroll
    for (int s = 0; s < kLrSteps; s++) {
        if (HAF_LR_ABL & 4) b1 = b;
        else if (s + 1 < kLrSteps) b1 = *reinterpret_cast<const half8 *>(bl + (s + 1) * 2048);
        else if (n == 0) b1 = *reinterpret_cast<const half8 *>(bl + 1024);
        HAF_SB();
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int j = 4 * s + i;
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s][i], b, s == 0 ? z4 : acc[i], 0, 0, 0);
            HAF_SB();
            if (s == 0 && i < COUNT && !(HAF_LR_ABL & 1)) { dma_piece(dma.g[FIRST + i], dma.l[FIRST + i], lane16); HAF_SB(); }
            if (HAF_LR_ABL & 2) continue;
            if (!CRP) {
                if (j < 16) { kq[j] = __builtin_amdgcn_exp2f(old[j >> 2][j & 3]); HAF_SB(); }
                if (PLN && j >= 16) {
                    // plain epilogue: b 2^z, one fma per element -- TWO of them behind each of the last eight MFMAs, the sixteen exps behind the
                    // first sixteen: every MFMA then shares the issue port with 8 cycles of vector work, exactly half its own 16 (with the fma
                    // four MFMAs behind its exp the middle twelve carried 12 cycles and the last four none)
                    const int e0 = 2 * (j - 16), e1 = e0 + 1;
                    sum[e0 >> 2][e0 & 3] = fmaf(cf_old, kq[e0], sum[e0 >> 2][e0 & 3]);
                    HAF_SB();
                    sum[e1 >> 2][e1 & 3] = fmaf(cf_old, kq[e1], sum[e1 >> 2][e1 & 3]);
                    HAF_SB();
                }
                if (!PLN && j >= kLrExpLag && j < 16 + kLrExpLag) {
                    const int e = j - kLrExpLag;
                    const float em1 = kq[e] - 1.0f;
                    HAF_SB();
                    const float ps = fmaf(old[e >> 2][e & 3], -kLn2f, em1);
                    HAF_SB();
                    sum[e >> 2][e & 3] = fmaf(cf_old, ps, sum[e >> 2][e & 3]);
                    HAF_SB();
                }
            } else if (j >= 2 && j < 18) {
                // DEGREE 4 here (the ten-step kernel's polynomial has degree 5): psi(t) = t^2/2 + t^3/6 + t^4/24, four instructions per element;
                // what it drops is at most 0.03 |t|^3 of psi (screen_band.h), 1.5e-5 for a model whose z stay within 0.1 -- the trained one
                const int e = j - 2;
                const float z = old[e >> 2][e & 3];
                const float pt = z * z;
                HAF_SB();
                float ph = fmaf(z, ba4, ba3);
                HAF_SB();
                ph = fmaf(ph, z, ba2);
                HAF_SB();
                sum[e >> 2][e & 3] = fmaf(pt, ph, sum[e >> 2][e & 3]);
                HAF_SB();
            }
        }
        b = b1;
    }
#undef HAF_SB
    __builtin_amdgcn_sched_barrier(0);
}

// The sweep of the low-rank form: k_svm_screen's structure (two 4-wave workgroups per CU, 3-deep LDS-DMA ring, epilogue of a column
// block between the MFMAs of the next, two sign-grouped sweeps, two-level coefficient sum) on 6-step images; the tail finishes the band
// from the raw sums (lr_finish_band) before the usual decision tail.  Whole requests only (no list mode, no SV-range split).
// FUSED: the projection is the sweep's prologue -- the ten input k-steps of B^' ("projection tiles", 12 KiB each: the twelve 16-row blocks
// of output slots for input slots 32 s .. 32 s + 31, in the SV tiles' piece layout) stream through the SAME LDS-DMA ring in front of the
// SV tiles, the wave's 64 evaluations x 192 outputs accumulate in 192 registers while the 10-step operand image passes by one k-step at
// a time (16 registers, the next step's loads issued behind the first MFMAs of the current one), and the converted fp16 fragments never
// leave the registers: no 6-step image in HBM (3 GB written and read per C5 step), no second launch.  Y = the 10-step images then.
// GATHER (round 5, "tier 0b" behind a low-rank first pass with the plain epilogue): the kernel serves a LIST -- slot j is evaluation
// idx_list[j], counters[count_slot] of them, never more than in_cap -- straight from what the first pass left in memory: the wave's 64
// operand images are gathered from the 10-step images by evaluation id (a lane's 16 bytes of a k-step lie where the feature kernel put
// them: screen.hip header), the raw sums and the common factor are read by evaluation id in the tail (the linear term L of the
// centred-remainder form rides in raw[6] since round 5: features.hip), flag words and the compaction behind them go by slot.  No
// second feature kernel, no second set of operand images (0.30 of tier 0b's 0.68 ms at C5).
template <int VAR, bool FUSED, bool GATHER>
__global__ __launch_bounds__(kS0Waves * 64, HAF_LR_WGS) void k_svm_screen_lr(const char *__restrict__ Y, const float *__restrict__ raw,
                                                                  const float *__restrict__ nax, const char *__restrict__ svt,
                                                                  const int *__restrict__ evalcell, const int *__restrict__ counters,
                                                                  SvmParams p, float *__restrict__ dec, int8_t *__restrict__ labels,
                                                                  unsigned long long *__restrict__ flag0_words, Dims d,
                                                                  float *__restrict__ margin, CrParams crp, LrBand lb,
                                                                  const char *__restrict__ ptiles,
                                                                  const int *__restrict__ idx_list, int count_slot, int in_cap)
{
    static_assert(!GATHER || FUSED, "the gather form reads the 10-step images");
    constexpr bool CRP = VAR == SCREEN_CR_POLY;
    constexpr int kV0 = FUSED ? kHFull : 0;                          // virtual tiles in front of the SV tiles: ring slot of SV tile t = (t + kV0) % 3
    __shared__ __attribute__((aligned(16))) char lds[kS0Buffers * kLrSvTileBytes + 3 * kS0Waves * kS0WaveEvals * 4];
    const int n_evals = GATHER ? min(counters[count_slot], in_cap) : counters[CNT_EVALS];
    const long base = (long)blockIdx.x * kS0BlockEvals;
    if (base >= n_evals) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long tile32 = (base >> 5) + 2 * wave;
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    const int np = d.sv_tile_neg, nt = d.n_sv_tiles;
    float *pos = reinterpret_cast<float *>(lds + kS0Buffers * kLrSvTileBytes) + wave * kS0WaveEvals;
    float *fin = pos + kS0Waves * kS0WaveEvals;
    float *sdyrow = fin + kS0Waves * kS0WaveEvals;                   // FUSED: |y^ - y32|^2 per evaluation (this wave's row)
    const unsigned lane16 = (unsigned)lane * 16u;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int poff[kS0WavePieces];                                         // (TileDma is sized for the 10-step form: three pieces used)
#pragma unroll
    for (int q = 0; q < kS0WavePieces; q++) poff[q] = (q < kLrWavePieces) ? (wave_u + kS0Waves * q) * 1024 : 0;
    static_assert(kS0Waves * kLrWavePieces * 1024 == kLrMatBytes && kLrMatBytes + 1024 == kLrSvTileBytes, "12 image pieces + 1 tail piece");
    auto stage3 = [&](const char *g, unsigned l) {
        asm volatile("s_nop 4");
#pragma unroll
        for (int q = 0; q < kLrWavePieces; q++) dma_piece(g + poff[q], l + (unsigned)poff[q], lane16);
    };
    // virtual tile v: the projection tiles (FUSED) in front of the SV tiles; its global image and whether it has a tail piece
    auto vtile = [&](int v) -> const char * { return (FUSED && v < kV0) ? ptiles + (size_t)v * kLrMatBytes : svt + (size_t)(v - kV0) * kLrSvTileBytes; };
    {
        stage3(vtile(0), lds0);
        if (!FUSED && wave_u == 0) dma_piece(svt + kLrMatBytes, lds0 + kLrMatBytes, lane16);
    }
    if (FUSED || nt > 1) {
        stage3(vtile(1), lds0 + kLrSvTileBytes);
        if (!FUSED && wave_u == 0) dma_piece(vtile(1) + kLrMatBytes, lds0 + kLrSvTileBytes + kLrMatBytes, lane16);
    }
    half8 a[kLrSteps][4];
    if (!FUSED) {
        const char *xt = Y + (size_t)tile32 * kLrMatBytes;
#pragma unroll
        for (int s = 0; s < kLrSteps; s++)
#pragma unroll
            for (int m = 0; m < 4; m++)
                a[s][m] = __builtin_nontemporal_load(reinterpret_cast<const half8 *>(xt + (m >> 1) * kLrMatBytes + (s * 2 + (m & 1)) * 1024 + lane * 16));
#pragma unroll
        for (int s = 0; s < kLrSteps; s++)
#pragma unroll
            for (int m = 0; m < 4; m++) asm volatile("" : "+v"(a[s][m]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else {
        // ---- the projection: D[output slot][evaluation] += B^'[output slot][input slot] X[input slot][evaluation], one input k-step per ring tile ----
        const char *xt = Y + (size_t)tile32 * kS0MatBytes;           // this wave's two 10-step images
        // GATHER: row block m of this wave holds the evaluations of list slots base + 64 wave + 16 m + (lane & 15); a lane's fragment of
        // k-step sx of evaluation e lies at image e / 32, piece (sx, (e / 16) % 2), lane 16 (lane / 16) + e % 16 (slots beyond the list: evaluation 0)
        const char *xg[4];
        if (GATHER) {
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const long sl = base + wave * kS0WaveEvals + 16 * m + (lane & 15);
                const int eg = sl < n_evals ? idx_list[sl] : 0;
                xg[m] = Y + (size_t)(eg >> 5) * kS0MatBytes + ((eg >> 4) & 1) * 1024 + ((lane >> 4) * 16 + (eg & 15)) * 16;
            }
        }
        auto xload = [&](int sx, half8 (&x)[4]) {
#pragma unroll
            for (int m = 0; m < 4; m++)
                x[m] = GATHER ? *reinterpret_cast<const half8 *>(xg[m] + sx * 2048)
                              : __builtin_nontemporal_load(reinterpret_cast<const half8 *>(xt + (m >> 1) * kS0MatBytes + (sx * 2 + (m & 1)) * 1024 + lane * 16));
        };
        f32x4 accp[2 * kLrSteps][4];                                 // twelve 16-row blocks of outputs x four 16-evaluation blocks
#pragma unroll
        for (int rb = 0; rb < 2 * kLrSteps; rb++)
#pragma unroll
            for (int m = 0; m < 4; m++) accp[rb][m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        half8 xc[4], xn[4];
        xload(0, xc);
#pragma unroll
        for (int m = 0; m < 4; m++) xn[m] = xc[m];
#pragma unroll
        for (int m = 0; m < 4; m++) asm volatile("" : "+v"(xc[m]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // projection tiles 0 and 1, step 0 of the operand images
        __syncthreads();
#pragma unroll 1
        for (int v = 0; v < kV0; v++) {
            const char *cur = lds + (v % kS0Buffers) * kLrSvTileBytes;
#pragma unroll
            for (int rb = 0; rb < 2 * kLrSteps; rb++) {
                const half8 bf = *reinterpret_cast<const half8 *>(cur + rb * 1024 + lane * 16);
                if (!(HAF_LR_ABL & 16)) {
#pragma unroll
                for (int m = 0; m < 4; m++) accp[rb][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf, xc[m], accp[rb][m], 0, 0, 0);
                } else if (rb < 4) {
                    accp[rb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf, xc[rb], accp[rb][0], 0, 0, 0);
                }
                if (rb == 0) {
                    // behind the first MFMAs (which have waited for nothing: everything issued so far has landed): the next step's operand
                    // fragments and the ring tile two ahead -- a projection tile or one of the first two SV tiles (with its tail piece)
                    __builtin_amdgcn_sched_barrier(0);
                    if (v + 1 < kV0) xload(v + 1, xn);
                    const int vn = v + 2;
                    if (vn < kV0 + nt) {
                        const unsigned ls = lds0 + (vn % kS0Buffers) * kLrSvTileBytes;
                        stage3(vtile(vn), ls);
                        if (vn >= kV0 && wave_u == 0) dma_piece(vtile(vn) + kLrMatBytes, ls + kLrMatBytes, lane16);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int m = 0; m < 4; m++) { asm volatile("" : "+v"(xn[m])); xc[m] = xn[m]; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        // fp16 fragments of the sweep (screen.hip: k_project for the row order) and |y^ - y32|^2
        float sdy[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int sp = 0; sp < kLrSteps; sp++)
#pragma unroll
            for (int m = 0; m < 4; m++) {
                half8 h;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float y0 = accp[2 * sp][m][r], y1 = accp[2 * sp + 1][m][r];
                    const _Float16 h0 = (_Float16)y0, h1 = (_Float16)y1;
                    h[r] = h0; h[4 + r] = h1;
                    const float d0 = (float)h0 - y0, d1 = (float)h1 - y1;
                    sdy[m] = fmaf(d0, d0, sdy[m]);
                    sdy[m] = fmaf(d1, d1, sdy[m]);
                }
                a[sp][m] = h;
            }
#pragma unroll
        for (int m = 0; m < 4; m++) {
            float sv = sdy[m];
            sv += __shfl_xor(sv, 16, 64);
            sv += __shfl_xor(sv, 32, 64);
            if (lane < 16) sdyrow[16 * m + lane] = sv;           // (sdyrow is this wave's own row, like pos and fin)
        }
    }

    float sum[4][4], part[4][4];
    f32x4 acc0[4], acc1[4];
    for (int ph = 0; ph < 2; ph++) {
        const int t_end = ph ? nt : np;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            acc1[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int r = 0; r < 4; r++) { sum[m][r] = 0.0f; part[m][r] = 0.0f; }
        }
        int fold = 0;
        float cf_prev = 0.0f;
        for (int t = ph ? np : 0; t < t_end; t++) {
            const char *cur = lds + ((t + kV0) % kS0Buffers) * kLrSvTileBytes;
            const int tn = (t + 2) % nt;
            const TileDma dma = tile_dma(svt + (size_t)tn * kLrSvTileBytes, lds0 + ((t + 2 + kV0) % kS0Buffers) * kLrSvTileBytes, poff);
            if (wave_u == 0 && !(HAF_LR_ABL & 1)) dma_piece(svt + (size_t)tn * kLrSvTileBytes + kLrMatBytes,
                                       lds0 + ((t + 2 + kV0) % kS0Buffers) * kLrSvTileBytes + kLrMatBytes, lane16);
            const float *tt = reinterpret_cast<const float *>(cur + kLrMatBytes);
            const float cf0 = tt[32 + (lane & 15)], cf1 = tt[48 + (lane & 15)];
            half8 bf0, bf1;
            screen_block_lr<0, 2, VAR>(cur, 0, lane, a, acc0, acc1, cf_prev, sum, dma, lane16, bf0, bf1);
            screen_block_lr<2, 1, VAR>(cur, 1, lane, a, acc1, acc0, cf0, sum, dma, lane16, bf0, bf1);
            cf_prev = cf1;
            if (kLrTwoLevel && ++fold == 8) {
                fold = 0;
#pragma unroll
                for (int m = 0; m < 4; m++)
#pragma unroll
                    for (int r = 0; r < 4; r++) { part[m][r] += sum[m][r]; sum[m][r] = 0.0f; }
            }
            if (!(HAF_LR_ABL & 8)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            }
            asm volatile("" ::: "memory");
        }
        float *dst = ph ? fin : pos;
        f32x4 zz[4];
#pragma unroll
        for (int m = 0; m < 4; m++) zz[m] = acc1[m];
        if (!CRP) {
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc1[m][r] = __builtin_amdgcn_exp2f(acc1[m][r]);
            asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc1[0]), "+v"(acc1[1]), "+v"(acc1[2]), "+v"(acc1[3]));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float ck;
                if (VAR == SCREEN_PLAIN) {
                    ck = cf_prev * acc1[m][r];
                } else if (!CRP) {
                    ck = cf_prev * fmaf(zz[m][r], -kLn2f, acc1[m][r] - 1.0f);
                } else {
                    const float z = zz[m][r];
                    ck = (cf_prev * (z * z)) * fmaf(fmaf(z, kPsiA4, kPsiA3), z, kPsiA2);
                }
                float v = ck + sum[m][r];
                if (kLrTwoLevel) v += part[m][r];
                v += __shfl_xor(v, 8, 64);
                v += __shfl_xor(v, 4, 64);
                v += __shfl_xor(v, 2, 64);
                v += __shfl_xor(v, 1, 64);
                if ((lane & 15) == 0) dst[16 * m + 4 * (lane >> 4) + r] = v;
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const long e = base + wave * kS0WaveEvals + lane;
    const bool live = e < n_evals;
    bool flagged = false;
    if (!kLrTwoLevel) p.guard_acc0 = p.guard_acc0_s;                  // single-level coefficient sum: (2 tiles + 14) u instead of (34 + tiles / 8) u
    if (live) {
        const long es = GATHER ? (long)idx_list[e] : e;                  // the evaluation this slot holds
        float sc = __builtin_amdgcn_exp2f(nax[es]);
        float4 g, g2 = float4{0.0f, 0.0f, 0.0f, 0.0f};
        float rw[6];
#pragma unroll
        for (int k = 0; k < 6; k++) rw[k] = raw[kBandFloats * es + k];
        if (GATHER) rw[3] = raw[kBandFloats * es + 6];                   // L = ln2 p.g of the centred-remainder form (the first pass's raw[3] is ITS correction sum)
        if (FUSED) rw[5] = sdyrow[lane];
        if (VAR == SCREEN_PLAIN) lr_finish_band_plain(rw, lb, g, g2);
        else lr_finish_band(rw, lb, g.x, g.y, g.z, g.w);
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(sc));
        flagged = screen_tail_vals<VAR, false>((double)pos[lane], (double)fin[lane], 0.0f, e, g, g2, sc, p, crp, GATHER ? idx_list : nullptr, evalcell, dec, labels, margin);
    }
    const unsigned long long bal = __ballot(flagged);
    if (lane == 0) flag0_words[(base >> 6) + wave] = bal;
}

void launch_svm_screen(const void *X0, const float *gband, const float *nax, const void *svt0, const int *evalcell, const int *counters,
                       SvmParams p, float *dec, int8_t *labels, unsigned long long *flag0_words, int *wgcount, int *flag0_list,
                       int flag0_cap, int *counters_rw, Dims d, long max_evals, float *margin, int variant, CrParams cr, hipStream_t s,
                       int also_counter, const int *idx_list, int count_slot, int out_slot, void *part_buf, int parts)
{
    long blocks = (max_evals + kS0BlockEvals - 1) / kS0BlockEvals;
    if (blocks <= 0) return;
    // list mode: what the INPUT list holds (it is the same size as the output list; whole requests: everything the launch covers)
    const int in_cap = idx_list ? (int)std::min<long>(flag0_cap, max_evals) : 0x7fffffff;
    // parts: 0 = the engine's rule (requests of up to kS0PartBlocks workgroups are split over SV ranges, as many as the live count asks
    // for), 1 = never, > 1 = that many (tests); the buffer holds kS0MaxParts x kS0PartBlocks x 256 partial sums (screen_part_bytes())
    const bool split = part_buf && parts != 1 && blocks <= kS0PartBlocks && (parts > 1 || d.n_sv_tiles >= 4 * kS0MinPartTiles);
#define HAF_SCREEN_LAUNCH(V)                                                                                                      \
    if (split) {                                                                                                                  \
        hipLaunchKernelGGL((k_svm_screen<V, true>), dim3((unsigned)blocks, kS0MaxParts), dim3(kS0Waves * 64), 0, s, (const char *)X0, gband, nax, \
                           (const char *)svt0, evalcell, counters, p, dec, labels, flag0_words, d, margin, cr, idx_list, count_slot,      \
                           (float4 *)part_buf, parts, in_cap);                                                                    \
        hipLaunchKernelGGL(k_screen_combine<V>, dim3((unsigned)blocks), dim3(kS0BlockEvals), 0, s, (const float4 *)part_buf, gband, nax,   \
                           evalcell, counters, p, dec, labels, flag0_words, d, margin, cr, idx_list, count_slot, parts, in_cap);   \
    } else {                                                                                                                      \
        hipLaunchKernelGGL((k_svm_screen<V, false>), dim3((unsigned)blocks), dim3(kS0Waves * 64), 0, s, (const char *)X0, gband, nax,     \
                           (const char *)svt0, evalcell, counters, p, dec, labels, flag0_words, d, margin, cr, idx_list, count_slot,      \
                           (float4 *)nullptr, 0, in_cap);                                                                         \
    }
    switch (variant) {
        case SCREEN_SUMSQ: HAF_SCREEN_LAUNCH(SCREEN_SUMSQ); break;
        case SCREEN_CR_EXP: HAF_SCREEN_LAUNCH(SCREEN_CR_EXP); break;
        case SCREEN_CR_POLY: HAF_SCREEN_LAUNCH(SCREEN_CR_POLY); break;
        default: HAF_SCREEN_LAUNCH(SCREEN_PLAIN); break;
    }
#undef HAF_SCREEN_LAUNCH
    launch_screen_compaction(flag0_words, wgcount, blocks, flag0_list, flag0_cap, counters_rw, also_counter, idx_list, count_slot, out_slot, in_cap, s);
}

void launch_svm_screen_lr(const void *Y, float *raw, const float *nax, const void *svt_lr, const int *evalcell, const int *counters,
                          SvmParams p, float *dec, int8_t *labels, unsigned long long *flag0_words, int *wgcount, int *flag0_list,
                          int flag0_cap, int *counters_rw, Dims d, long max_evals, float *margin, int variant, CrParams cr, LrBand lb,
                          hipStream_t s, int also_counter, const void *ptiles, const int *idx_list, int count_slot, int out_slot)
{
    const long blocks = (max_evals + kS0BlockEvals - 1) / kS0BlockEvals;
    if (blocks <= 0) return;
    lb.poly = variant == SCREEN_CR_POLY;
    // gather form (idx_list != nullptr; CR_EXP on the fused kernel only): what the INPUT list holds, as in launch_svm_screen
    const int in_cap = idx_list ? (int)std::min<long>(flag0_cap, max_evals) : 0x7fffffff;
#define HAF_LR_LAUNCH(V, F, G)                                                                                                        \
    hipLaunchKernelGGL((k_svm_screen_lr<V, F, G>), dim3((unsigned)blocks), dim3(kS0Waves * 64), 0, s, (const char *)Y, raw, nax,        \
                       (const char *)svt_lr, evalcell, counters, p, dec, labels, flag0_words, d, margin, cr, lb, (const char *)ptiles,  \
                       idx_list, count_slot, in_cap)
    if (idx_list) { HAF_LR_LAUNCH(SCREEN_CR_EXP, true, true); }
    else if (variant == SCREEN_CR_POLY) { if (ptiles) HAF_LR_LAUNCH(SCREEN_CR_POLY, true, false); else HAF_LR_LAUNCH(SCREEN_CR_POLY, false, false); }
    else if (variant == SCREEN_PLAIN) { if (ptiles) HAF_LR_LAUNCH(SCREEN_PLAIN, true, false); else HAF_LR_LAUNCH(SCREEN_PLAIN, false, false); }
    else { if (ptiles) HAF_LR_LAUNCH(SCREEN_CR_EXP, true, false); else HAF_LR_LAUNCH(SCREEN_CR_EXP, false, false); }
#undef HAF_LR_LAUNCH
    const int cs = idx_list ? count_slot : CNT_EVALS;
    launch_screen_compaction(flag0_words, wgcount, blocks, flag0_list, flag0_cap, counters_rw, also_counter, idx_list, cs,
                             idx_list ? out_slot : CNT_FLAGGED0, in_cap, s);
}

size_t screen_part_bytes() { return (size_t)kS0MaxParts * kS0PartBlocks * kS0BlockEvals * sizeof(float4); }

}  // namespace haf
