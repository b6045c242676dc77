#!/usr/bin/env python3
"""Timing-only ablations of k_svm_screen as a bit mask (results are WRONG; every evaluation is reported as decided so that the engine
does not re-run anything):  python tools/abl_build.py MASK  ->  haf_grasping_amd/libhafgrasp_vablMASK.so
  1 no epilogue VALU (v_exp_f32, fma)   2 no LDS-DMA inside the loop   4 no tile wait/barrier   8 no B-fragment LDS reads in the loop
  16 no sched_barrier pinning (hipcc orders the stream)
Time with  HAF_LIB=haf_grasping_amd/libhafgrasp_vablMASK.so python tools/time_svm_stage.py  on ONE GPU box."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "haf_grasping_amd", "csrc")
mask = int(sys.argv[1])
s = open(os.path.join(CS, "screen.hip")).read()
def rep(a, b, count=1):
    global s
    assert s.count(a) >= 1, a
    s = s.replace(a, b)
rep("        flagged = !(adv > err);                                     // also catches NaN", "        flagged = false;")
if mask & 1:
    rep("        if (ex) { q0 = __builtin_amdgcn_exp2f(old[e0 >> 2][e0 & 3]); HAF_SB(); }", '        if (ex) { asm volatile("" ::"v"(old[e0 >> 2][e0 & 3])); HAF_SB(); }')
    rep("        if (ex) { q1 = __builtin_amdgcn_exp2f(old[e1 >> 2][e1 & 3]); HAF_SB(); }", '        if (ex) { asm volatile("" ::"v"(old[e1 >> 2][e1 & 3])); HAF_SB(); }')
    rep("        if (fm && !SUMSQ) { sum[f0 >> 2][f0 & 3] = fmaf(cf_old, k0, sum[f0 >> 2][f0 & 3]); HAF_SB(); }", "")
    rep("        if (fm && !SUMSQ) { sum[f1 >> 2][f1 & 3] = fmaf(cf_old, k1, sum[f1 >> 2][f1 & 3]); HAF_SB(); }", "")
if mask & 2:
    for q in ("FIRST", "FIRST + 1", "FIRST + 2"):
        rep("dma_piece(dma.g[%s], dma.l[%s], lane16); HAF_SB();" % (q, q), "")
    rep("            if (wave_u == 0) dma_piece(svt0 + (size_t)tn * kS0SvTileBytes + kS0MatBytes,\n                                       lds0 + ((t + 2) % kS0Buffers) * kS0SvTileBytes + kS0MatBytes, lane16);", "")
    rep('            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");', '            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");')
if mask & 4:
    rep('            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");', "") if not (mask & 2) else rep('            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n            __builtin_amdgcn_s_barrier();', "            __builtin_amdgcn_s_barrier();")
    rep("            __builtin_amdgcn_s_barrier();", "")
if mask & 8:
    rep("        if (s + 1 < kHFull) b1 = *reinterpret_cast<const half8 *>(bl + (s + 1) * 2048);\n        else if (n == 0) b1 = *reinterpret_cast<const half8 *>(bl + 1024);", "")
if mask & 16:
    rep("#define HAF_SB() __builtin_amdgcn_sched_barrier(0)", "#define HAF_SB() do {} while (0)")
tmp = tempfile.mkdtemp()
open(os.path.join(tmp, "screen.hip"), "w").write(s)
hipcc = "/opt/rocm/bin/hipcc"
subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-inline-asm",
                       "-fno-slp-vectorize", "-I" + CS, "-c", os.path.join(tmp, "screen.hip"), "-o", os.path.join(tmp, "screen.o")])
out = os.path.join(ROOT, "haf_grasping_amd", "libhafgrasp_vabl%d.so" % mask)
subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(CS, o) for o in ("kernels.o", "prob.o", "engine.o", "parsers.o", "multi.o")] +
                      [os.path.join(tmp, "screen.o"), "-L/opt/rocm/lib", "-lrccl", "-lpthread", "-Wl,-rpath,/opt/rocm/lib", "-o", out])
print("built", out)
