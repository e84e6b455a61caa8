"""Build librecguru_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

Every source is compiled with -save-temps so that its device ISA stays beside the object (recguru_amd/build/isa/<name>.s), and
every build ends with the ISA screen of recguru_amd/isa_screen.py: hipcc (ROCm 7.2) can place register spills in front of the
exec restore of a join block -- silent wrong values in the lanes that skipped the branch -- and a kernel that
shows the pattern FAILS the build (RG_BUILD_NO_SCREEN=1 to look at such a build anyway).  build/BUILD_INFO.json records the
compiler version, the flags and the hash of the library the screen passed on."""
import glob
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librecguru_hip.so")
LIB_DET = os.path.join(HERE, "librecguru_hip_det.so")
# -Wno-pass-failed: the L > 224 attention instantiations hold > 80 KB of LDS per workgroup, so their launch-bounds
# occupancy hint (2 waves per SIMD) cannot be met -- expected, not worth a warning per instantiation
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wno-unused-value", "-Wno-pass-failed"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _compile(srcs, hdrs, objdir, extra, force, verbose, jobs):
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs) or not os.path.exists(_isa_of(o)):
            cmd = ["hipcc"] + FLAGS + extra + ["-save-temps=obj", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            if len(procs) >= jobs:
                _drain(procs)
    _drain(procs)
    for o in objs:                              # keep the device ISA, drop the other intermediates of -save-temps
        stem = o[:-2]
        tmp = stem + "-hip-amdgcn-amd-amdhsa-gfx950.s"
        if os.path.exists(tmp):
            os.replace(tmp, _isa_of(o))
        for junk in glob.glob(stem + "-hip-amdgcn-*") + glob.glob(stem + "-host-*") + glob.glob(stem + ".hip-hip-*"):
            os.remove(junk)
    return objs


def _link_and_screen(target, objs, screened_objs, info_name, force, verbose):
    if force or _stale(target, objs):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", target] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    try:
        _screen_and_record([_isa_of(o) for o in screened_objs], verbose, target, info_name)
    except RuntimeError:
        # a library whose ISA the screen refuses must not stay loadable (hip.lib() loads whatever is on disk -- ADVICE r4), and the
        # record of an EARLIER library of that name must not stay behind to vouch for nothing
        for stale in (target, os.path.join(HERE, "build", info_name)):
            if os.path.exists(stale):
                os.remove(stale)
        raise
    return target


def _sources():
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.hip.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return srcs, hdrs


def build(force=False, verbose=False, jobs=4, det=True):
    """librecguru_hip.so, and (det) librecguru_hip_det.so beside it."""
    srcs, hdrs = _sources()
    objs = _compile(srcs, hdrs, os.path.join(HERE, "build"), [], force, verbose, jobs)
    _link_and_screen(LIB, objs, objs, "BUILD_INFO.json", force, verbose)
    if det:
        # the deterministic library is test infrastructure (tests/test_det_gpu.py, RG_DETERMINISTIC=1): a refusal of ITS build -- an ISA-screen
        # finding that exists only under -DRG_DETERMINISTIC -- must not fail the build of the production library, which is complete and
        # screened at this point.  RG_BUILD_DET_STRICT=1 (the CPU test of the det ABI sets it) turns the warning back into the error.
        try:
            build_det(force, verbose, jobs)
        except RuntimeError as e:
            if os.environ.get("RG_BUILD_DET_STRICT"):
                raise
            sys.stderr.write("recguru_amd.build: librecguru_hip_det.so NOT built (%s); librecguru_hip.so is built and screened\n" % str(e).splitlines()[0])
    return LIB


def build_det(force=False, verbose=False, jobs=4):
    """The deterministic-reduction library (csrc/rg_det.hip.h): the translation units that accumulate are compiled a second time
    with -DRG_DETERMINISTIC into build/det/, the others' objects are shared with librecguru_hip.so.  Same ISA screen."""
    srcs, hdrs = _sources()
    acc = [s for s in srcs if '#include "rg_det.hip.h"' in open(s).read()]
    dobjs = _compile(acc, hdrs, os.path.join(HERE, "build", "det"), ["-DRG_DETERMINISTIC=1"], force, verbose, jobs)
    names = set(os.path.basename(o) for o in dobjs)
    shared = [os.path.join(HERE, "build", os.path.basename(s)[:-4] + ".o") for s in srcs if os.path.basename(s)[:-4] + ".o" not in names]
    missing = [o for o in shared if not os.path.exists(o)]
    if missing:
        raise RuntimeError("build_det: build() first (%s missing)" % ", ".join(missing))
    return _link_and_screen(LIB_DET, dobjs + shared, dobjs, "BUILD_INFO_det.json", force, verbose)


def build_variant(name, extra, only=None, force=False, verbose=False, jobs=4):
    """An A/B library: every source (or those named in `only`, the rest shared with the shipped build) compiled with the `extra`
    flags into build/variants/<name>/, linked as tools/variants/v_<name>.so (travels to the GPU box; load it with RG_HIP_LIB).
    Same ISA screen as the shipped library."""
    srcs, hdrs = _sources()
    mine = [s for s in srcs if only is None or os.path.basename(s)[:-4] in only]
    vobjs = _compile(mine, hdrs, os.path.join(HERE, "build", "variants", name), list(extra), force, verbose, jobs)
    names = set(os.path.basename(o) for o in vobjs)
    shared = [os.path.join(HERE, "build", os.path.basename(s)[:-4] + ".o") for s in srcs if os.path.basename(s)[:-4] + ".o" not in names]
    out = os.path.join(HERE, "..", "tools", "variants")
    os.makedirs(out, exist_ok=True)
    return _link_and_screen(os.path.join(out, "v_%s.so" % name), vobjs + shared, vobjs, os.path.join("variants", "BUILD_INFO_%s.json" % name), force, verbose)


def _isa_of(obj):
    d = os.path.join(os.path.dirname(obj), "isa")          # (a directory of its own: .gpurunignore keeps the ~30 MB of text off the GPU box)
    os.makedirs(d, exist_ok=True)
    return os.path.join(d, os.path.basename(obj)[:-2] + ".s")


def _screen_and_record(isa_files, verbose, lib_path=None, info_name="BUILD_INFO.json"):
    from . import isa_screen
    lib_path = lib_path or LIB
    info_path = os.path.join(HERE, "build", info_name)
    sha = hashlib.sha256(open(lib_path, "rb").read()).hexdigest()
    try:
        with open(info_path) as f:
            if json.load(f).get("library_sha256") == sha:
                return                           # this very library has been screened
    except (OSError, ValueError):
        pass
    flagged, warnings, opsel = [], 0, []
    for fn in isa_files:
        bad, warn = isa_screen.screen(fn)
        warnings += len(warn)
        flagged += [(os.path.basename(fn), kernel, no, block, len(ins)) for kernel, no, block, ins in bad]
        opsel += [(os.path.basename(fn), kernel, no, text) for kernel, no, text in isa_screen.packed_opsel(fn)]
    ver = subprocess.run(["hipcc", "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode().strip().splitlines()
    bypass = bool(os.environ.get("RG_BUILD_NO_SCREEN"))
    info = {"hipcc": ver[:3], "flags": FLAGS, "library_sha256": sha, "isa_files": len(isa_files), "screen": "recguru_amd/isa_screen.py",
            "flagged_join_blocks": [list(x) for x in flagged], "spill_in_front_of_exec_restore_warnings": warnings,
            "packed_f32_high_half_selects": [list(x) for x in opsel]}
    if bypass and (flagged or opsel or warnings):
        # a flagged library kept on purpose (RG_BUILD_NO_SCREEN=1): never record its hash as "screened", so that the next
        # ordinary build screens -- and refuses -- it again instead of returning early on a matching hash
        info["screen_bypassed"] = True
        info["library_sha256"] = None
        info["bypassed_library_sha256"] = sha
    if opsel and not os.environ.get("RG_BUILD_NO_SCREEN"):
        msg = "\n".join("  %s: %s: line %d: %s" % x for x in opsel[:12])
        raise RuntimeError("hipcc emitted a packed-f32 operation that feeds a LOW result from the HIGH half of a source pair (%d site(s); "
                           "recguru_amd/isa_screen.py, DESIGN.md 2a finding 1): on gfx950 that form returns wrong values in lanes 48-63 "
                           "when a second wave shares the SIMD.  Keep per-row statistics out of one register pair (scalar temporaries, "
                           "-fno-slp-vectorize on the function) until the pattern is gone (RG_BUILD_NO_SCREEN=1 builds anyway):\n%s"
                           % (len(opsel), msg))
    if warnings and not os.environ.get("RG_BUILD_NO_SCREEN"):
        # (round 5) the copies that are a conditional body's own masked writes -- source defined inside the block -- are recognised and
        # dropped by the screen's dataflow triage; what is left is a VGPR <-> AGPR / scratch copy of a value that was live BEFORE the
        # narrowed region, executed under the narrowed exec: the spill defect without a join label
        raise RuntimeError("hipcc placed %d spill-like copy run(s) of live-in registers in front of an exec restore (recguru_amd/isa_screen.py, "
                           "DESIGN.md 2a; `python tools/isa_exec_screen.py recguru_amd/build/isa/*.s` lists them; RG_BUILD_NO_SCREEN=1 builds anyway)"
                           % warnings)
    if flagged and not os.environ.get("RG_BUILD_NO_SCREEN"):
        msg = "\n".join("  %s: %s: %d vector instruction(s) in front of the exec restore at line %d (join block %s)" % (f, k, n, no, b)
                        for f, k, no, b, n in flagged)
        raise RuntimeError("hipcc placed spill code under a narrowed exec mask (recguru_amd/isa_screen.py, DESIGN.md 2a) -- these kernels "
                           "return wrong values in the lanes that skipped the branch; change the source until the pattern is gone "
                           "(RG_BUILD_NO_SCREEN=1 builds anyway):\n" + msg)
    with open(info_path, "w") as f:
        json.dump(info, f, indent=1)
    if verbose:
        print("ISA screen: %d files, %d flagged join blocks, %d packed-f32 high-half selects, %d warnings" % (len(isa_files), len(flagged), len(opsel), warnings))


def _drain(procs):
    while procs:
        s, p = procs.pop(0)
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on %s" % s)
        txt = out.decode()
        if "warning" in txt:
            sys.stderr.write(txt)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
