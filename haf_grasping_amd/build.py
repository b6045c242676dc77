"""Builds the gfx950 libraries with hipcc, in-tree, so the .so files travel to the GPU box.

  libhafgrasp.so           the product: C-ABI of include/hafgrasp.h, nothing else exported for tests or experiments
  libhafgrasp_testing.so   the same kernels + the engine*.cpp units compiled with -DHAF_TESTING (+ engine_testing.cpp): haf_test_* hooks and the
                           environment switches that scale the guard bands (tests/ only)
  haf_grasp_cli            ROS-free command line front end (C++ over the C-ABI)

After linking (under temporary names), check_exp_hazard() disassembles the contraction kernels and fails the build when a
v_exp_f32 result is read too soon (a measured gfx950 hazard that a compiler update could silently re-open); only a
build that passes gets the library names up_to_date() looks for.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhafgrasp.so")
LIB_TESTING = os.path.join(HERE, "libhafgrasp_testing.so")
# the engine's host side: every one of these is compiled twice, without and with -DHAF_TESTING (csrc/engine_state.h: test_env)
ENGINE_SOURCES = ["engine.cpp", "engine_tables.cpp", "engine_request.cpp", "engine_geometry.cpp", "engine_debug.cpp"]
TESTING_ONLY = ["engine_testing.cpp", "testkernels.hip"]         # libhafgrasp_testing.so only
SOURCES = ["prestages.hip", "features.hip", "contraction.hip", "screen.hip", "recheck.hip", "exact8.hip", "vote.hip", "prob.hip"] + \
          ENGINE_SOURCES + ["parsers.cpp", "multi.cpp"]
# per-file extra flags (screen.hip: see its header)
EXTRA = {"screen.hip": ["-fno-slp-vectorize"]}
HEADERS = ["kernels.h", "device_common.h", "feature_device.h", "screen_band.h", "parsers.h", "decq.h", "engine_internal.h", "engine_state.h"] + TESTING_ONLY + [ os.path.join("..", "..", "include", "hafgrasp.h"),
           os.path.join("..", "cli", "haf_grasp_cli.cpp"), os.path.join("..", "..", "ros_shim", "shim_core.h")]
# -ffp-contract=off: the bit-exact stages spell out every rounding; nothing may be fused behind their back
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-result", "-Wno-inline-asm"]
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
OBJDUMP = os.path.join(ROCM, "lib", "llvm", "bin", "llvm-objdump")
# kernels whose epilogue consumes v_exp_f32 results next to MFMAs, and the distance (instructions between the exp and the
# first reader of its destination register) the build insists on.  Measured: 3 fails on hardware, >= 8 never did.
EXP_HAZARD_KERNELS = ["k_svm_screen", "k_svm_rbf_h", "k_svm_rbf"]
EXP_MIN_DISTANCE = 8


def up_to_date():
    if not (os.path.exists(LIB) and os.path.exists(LIB_TESTING)):
        return False
    t = min(os.path.getmtime(LIB), os.path.getmtime(LIB_TESTING))
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return all(os.path.getmtime(d) <= t for d in deps if os.path.exists(d))


def _regs(tok):
    """Registers named by one operand token: v12 -> {12}; v[12:15] -> {12..15}; anything else -> {}."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def exp_hazard_report(lib=None, kernels=None):
    """Per kernel: the smallest number of instructions between a v_exp_f32 and the first later instruction that reads
    its destination VGPR (straight-line distance inside the kernel's disassembly; a write to the register ends the
    search; s_nop N counts as N + 1 instructions because it is that many wait states).  {kernel: (min_distance, n_exps)}"""
    lib = lib or LIB
    kernels = kernels or EXP_HAZARD_KERNELS
    # the gfx950 code objects are bundled inside the host .so: llvm-objdump --offloading writes them out, one per
    # translation unit, into the working directory
    import glob
    import shutil
    import tempfile
    tmp = tempfile.mkdtemp(prefix="haf_isa_")
    try:
        shutil.copy(lib, os.path.join(tmp, "lib.so"))          # the bundles are written next to the input file
        subprocess.check_call([OBJDUMP, "--offloading", "lib.so"], cwd=tmp, stdout=subprocess.DEVNULL)
        text = ""
        for co in sorted(glob.glob(os.path.join(tmp, "*gfx950*"))):
            text += subprocess.check_output([OBJDUMP, "-d", co]).decode(errors="replace")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    report = {}
    cur, body = None, {}
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            body[cur] = []
            continue
        if cur is None:
            continue
        ins = line.split("//")[0].strip()
        if ins and not ins.endswith(":"):
            body[cur].append(ins)
    for want in kernels:
        names = [k for k in body if want in k and not k.endswith(".kd")]
        worst, count = None, 0
        for name in names:
            code = body[name]
            for i, ins in enumerate(code):
                if not ins.startswith("v_exp_f32"):
                    continue
                ops = [t.strip() for t in ins.split(None, 1)[1].split(",")]
                dst = _regs(ops[0])
                if not dst:
                    continue
                count += 1
                dist = 0
                for later in code[i + 1:]:
                    parts = later.split(None, 1)
                    op = parts[0]
                    toks = [t.strip() for t in parts[1].split(",")] if len(parts) > 1 else []
                    if op.startswith("s_branch") or op.startswith("s_cbranch") or op.startswith("s_endpgm") or op.startswith("s_setpc"):
                        break                      # leaves the straight line: the loop back edge is >= a whole tile away
                    srcs = set()
                    for t in toks[1:]:
                        srcs |= _regs(t)
                    # stores and MFMA C operands read their "first" operand too
                    reads_first = op.startswith(("global_store", "buffer_store", "ds_write", "ds_store", "flat_store", "scratch_store"))
                    if reads_first and toks:
                        srcs |= _regs(toks[0])
                    if srcs & dst:
                        worst = dist if worst is None else min(worst, dist)
                        break
                    if toks and (_regs(toks[0]) & dst) and not reads_first:
                        break                      # overwritten before anybody read it
                    m2 = re.fullmatch(r"s_nop\s+(\d+)", later)
                    dist += (int(m2.group(1)) + 1) if m2 else 1
        report[want] = (worst, count)
    return report


def check_exp_hazard(lib=None, verbose=False):
    rep = exp_hazard_report(lib)
    bad = []
    for k, (dist, n) in rep.items():
        if verbose:
            print("  v_exp_f32 hazard check: %-14s %3d exps, nearest reader %s instructions behind" % (k, n, dist))
        if n == 0:
            bad.append("%s: no v_exp_f32 found in the disassembly (kernel renamed? check EXP_HAZARD_KERNELS)" % k)
        elif dist is not None and dist < EXP_MIN_DISTANCE:
            bad.append("%s: a v_exp_f32 result is read %d instructions after the exp (< %d): the gfx950 transcendental "
                       "hazard of DESIGN.md §2 is open again" % (k, dist, EXP_MIN_DISTANCE))
    if bad:
        raise RuntimeError("build check failed:\n  " + "\n  ".join(bad))
    return rep


# Round 5: the group-parallel feature kernels (k_features<MODE, WAVES>) read the descriptor words of a wave's attribute group by SCALAR
# loads -- the wave index is held in an SGPR (readfirstlane).  As a VGPR expression the compiler fetched them with ~17 vector loads and
# as many s_waitcnt vmcnt per slot (138 global_load in the slot loop of k_features<2, 8>, 40 us of a 49 us launch at C3).  The build
# refuses an instance with more vector loads than the staging, list and operand traffic account for.
DESCRIPTOR_LOAD_KERNELS = "k_featuresILi"
MAX_VECTOR_LOADS = 80


def descriptor_load_report(lib=None):
    """{kernel symbol: vector load instructions (global_load / buffer_load)} of every k_features<., .> instance"""
    import glob
    import shutil
    import tempfile
    lib = lib or LIB
    tmp = tempfile.mkdtemp(prefix="haf_isa_")
    try:
        shutil.copy(lib, os.path.join(tmp, "lib.so"))
        subprocess.check_call([OBJDUMP, "--offloading", "lib.so"], cwd=tmp, stdout=subprocess.DEVNULL)
        text = ""
        for co in sorted(glob.glob(os.path.join(tmp, "*gfx950*"))):
            text += subprocess.check_output([OBJDUMP, "-d", co]).decode(errors="replace")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    rep, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1) if (DESCRIPTOR_LOAD_KERNELS in m.group(1) and not m.group(1).endswith(".kd")) else None
            if cur:
                rep[cur] = 0
            continue
        if cur and line.split("//")[0].strip().startswith(("global_load", "buffer_load")):
            rep[cur] += 1
    return rep


def check_descriptor_loads(lib=None, verbose=False):
    rep = descriptor_load_report(lib)
    if verbose:
        for k, n in sorted(rep.items()):
            print("  descriptor-load check: %-60s %3d vector loads" % (k[:60], n))
    if not rep:
        raise RuntimeError("check_descriptor_loads: no %s instance found in %s" % (DESCRIPTOR_LOAD_KERNELS, lib or LIB))
    bad = {k: n for k, n in rep.items() if n > MAX_VECTOR_LOADS}
    if bad:
        raise RuntimeError("build check failed: the group-parallel feature kernels fetch descriptor words by vector loads again "
                           "(> %d global/buffer loads): %r" % (MAX_VECTOR_LOADS, bad))
    return rep


READELF = os.path.join(ROCM, "lib", "llvm", "bin", "llvm-readelf")
# kernels that live at the edge of the register file (two waves of ~250 VGPRs per SIMD): a scratch spill in their inner loops
# would be a silent 2x -- the build refuses it (round 4: a packed-fp32 form of the polynomial epilogue compiled to 54 spills)
NO_SPILL_KERNELS = ["k_svm_screen", "k_svm_rbf_h", "k_svm_rbf", "k_recheck_i8"]


def check_no_spills(lib=None, verbose=False):
    """{kernel symbol: (vgprs, spills)} of the kernels in NO_SPILL_KERNELS from the code objects' metadata; raises on any spill."""
    import glob
    import shutil
    import tempfile
    lib = lib or LIB
    tmp = tempfile.mkdtemp(prefix="haf_meta_")
    report = {}
    try:
        shutil.copy(lib, os.path.join(tmp, "lib.so"))
        subprocess.check_call([OBJDUMP, "--offloading", "lib.so"], cwd=tmp, stdout=subprocess.DEVNULL)
        for co in sorted(glob.glob(os.path.join(tmp, "*gfx950*"))):
            text = subprocess.check_output([READELF, "--notes", co]).decode(errors="replace")
            name = vg = None
            for line in text.splitlines():
                line = line.strip()
                if line.startswith(".name:"):
                    name = line.split(":", 1)[1].strip()
                elif line.startswith(".vgpr_count:"):
                    vg = int(line.split(":", 1)[1])
                elif line.startswith(".vgpr_spill_count:") and name:
                    if any(k in name for k in NO_SPILL_KERNELS):
                        report[name] = (vg, int(line.split(":", 1)[1]))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    bad = {k: v for k, v in report.items() if v[1] != 0}
    if verbose:
        for k, v in sorted(report.items()):
            print("  %-60s %3d VGPRs, %d spilled" % (k[:60], v[0] or -1, v[1]))
    if bad:
        raise RuntimeError("register spills in a contraction kernel: %r" % bad)
    if not report:
        raise RuntimeError("check_no_spills: no contraction kernel found in %s" % lib)
    return report


def _compile_and_link(out_lib, out_testing, obj_dir, extra_flags=(), verbose=False, checks=True):
    """Compiles every translation unit into obj_dir and links the two libraries under the given names (checked for the v_exp_f32
    hazard and for spills before they get those names)."""
    hipcc = os.environ.get("HIPCC", os.path.join(ROCM, "bin", "hipcc"))
    os.makedirs(obj_dir, exist_ok=True)

    def compile_one(src, suffix="", defs=()):
        obj = os.path.join(obj_dir, os.path.splitext(src)[0] + suffix + ".o")
        # HAF_EXPERIMENT_FLAGS: extra compiler flags for ablation builds (tools/ablate_h.sh); never set for a build that is kept
        cmd = [hipcc] + FLAGS + EXTRA.get(src, []) + os.environ.get("HAF_EXPERIMENT_FLAGS", "").split() + list(extra_flags) + list(defs) + \
              ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    # the translation units are independent: compile them side by side (HAF_BUILD_JOBS, default 4 -- a device compile holds ~1 GiB)
    from concurrent.futures import ThreadPoolExecutor
    jobs = [(src, "", ()) for src in SOURCES] + [(src, "_testing", ("-DHAF_TESTING",)) for src in ENGINE_SOURCES + TESTING_ONLY]
    with ThreadPoolExecutor(max_workers=max(1, int(os.environ.get("HAF_BUILD_JOBS", "4")))) as pool:
        done = list(pool.map(lambda j: compile_one(*j), jobs))
    objs = dict(zip(SOURCES, done[:len(SOURCES)]))
    testing_objs = dict(zip(ENGINE_SOURCES + TESTING_ONLY, done[len(SOURCES):]))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"]
    libs = ["-L" + os.path.join(ROCM, "lib"), "-lrccl", "-lpthread", "-Wl,-rpath," + os.path.join(ROCM, "lib")]
    # The libraries are linked under temporary names and get their real ones only after the v_exp_f32 check has passed: a build
    # that fails the check (or cannot run it: llvm-objdump missing) leaves NO libhafgrasp.so behind, so the next build() cannot
    # mistake it for an up-to-date one.
    staged = []
    try:
        for out in (out_lib, out_testing):
            tmp = out + ".unchecked"
            if out == out_lib:
                members = [objs[s] for s in SOURCES]
            else:
                members = [testing_objs.get(s, objs[s]) for s in SOURCES] + [testing_objs[s] for s in TESTING_ONLY]
            cmd = link + members + libs + ["-o", tmp]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            staged.append((tmp, out))
        for out in (out_lib, out_testing):
            if os.path.exists(out):
                os.remove(out)                  # whatever happens below, a stale library must not survive a failed build
        if checks:
            check_exp_hazard(staged[0][0], verbose=verbose)
            check_no_spills(staged[0][0], verbose=verbose)
            check_descriptor_loads(staged[0][0], verbose=verbose)
        for tmp, out in staged:
            os.replace(tmp, out)
    finally:
        for tmp, _ in staged:
            if os.path.exists(tmp):
                os.remove(tmp)


def build_variant(name, flags, verbose=False, checks=True):
    """An experiment build NEXT TO the product: haf_grasping_amd/variants/libhafgrasp_<name>.so and libhafgrasp_testing_<name>.so from
    the same sources with extra compiler flags (e.g. -DHAF_LR_WGS=3), objects under csrc/_variants/<name>/.  The product's names
    are never touched; load a variant with HAF_LIB / HAF_TESTLIB (capi.py).  Same ISA checks as the product build unless checks=False
    (--no-checks: TIMING-ONLY ablation variants whose results are garbage by construction; never for a build whose results are used)."""
    vdir = os.path.join(HERE, "variants")
    os.makedirs(vdir, exist_ok=True)
    out = os.path.join(vdir, "libhafgrasp_%s.so" % name)
    out_t = os.path.join(vdir, "libhafgrasp_testing_%s.so" % name)
    _compile_and_link(out, out_t, os.path.join(CSRC, "_variants", name), extra_flags=list(flags), verbose=verbose, checks=checks)
    return out, out_t


def build(force=False, verbose=False):
    if not force and up_to_date():
        return LIB
    hipcc = os.environ.get("HIPCC", os.path.join(ROCM, "bin", "hipcc"))
    _compile_and_link(LIB, LIB_TESTING, CSRC, verbose=verbose)
    # ROS-free command line front end (C++ host code over the C-ABI)
    cli = os.path.join(HERE, "haf_grasp_cli")
    cmd = [hipcc, "-O2", "-std=c++17", "-I" + os.path.join(HERE, "..", "ros_shim"), "-I" + os.path.join(HERE, "..", "include"),
           os.path.join(HERE, "cli", "haf_grasp_cli.cpp"),
           "-o", cli, "-L" + HERE, "-lhafgrasp", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":          # python -m haf_grasping_amd.build --variant NAME -DFLAG ...
        rest = [a for a in sys.argv[3:] if a != "--no-checks"]
        print(build_variant(sys.argv[2], rest, verbose=True, checks="--no-checks" not in sys.argv))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
