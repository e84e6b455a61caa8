"""Build-time screen for a hipcc (ROCm 7.2 LLVM AMDGPU) miscompile: a register spill placed in FRONT of the exec restore of a
join block.

What goes wrong.  `if (lane-dependent condition) { ... }` becomes

        s_and_saveexec_b64 s[A:B], <cond>      ; exec narrowed to the lanes that take the branch
        s_cbranch_execz .Ljoin
        ...body...
    .Ljoin:
        s_or_b64 exec, exec, s[A:B]             ; exec restored -- must be the FIRST thing the join block does

The register allocator runs after this lowering.  When it decides to park live VGPRs in AGPRs (or scratch) at the top of
.Ljoin and emits the copies IN FRONT of the s_or_b64, they execute under the NARROWED exec: only the lanes that took the
branch are saved, the other lanes of the spill slot keep what an earlier use of the slot left there, and the reload -- much
later, under full exec -- hands those lanes stale values.  Nothing in the source is wrong and any edit that changes the register
allocation moves or removes it.  Found in round 4 in post_attn_fwd_kernel<x3, 2, true, false> (fused.hip): LayerNorm's
`if (lg == 0)` exchange block, eleven v_accvgpr_write_b32 in front of the exec restore, lanes 16-63 of the output-copy addresses
stale: three rows of four of every tile wrong.

The screen: in every kernel, between a label and the first `s_or_b64 exec, exec, ...` of that block there may be no vector
instruction that writes a register file other lanes depend on later -- v_accvgpr_write / v_accvgpr_read / scratch_* /
buffer_* spill traffic -- (a plain VALU instruction there would be just as wrong, so every v_* / ds_* / global_* / scratch_* /
buffer_* instruction is flagged).

  python tools/isa_exec_screen.py file.s [file.s ...]      exit code 1 if any kernel is flagged
  python tools/isa_exec_screen.py --build                   compiles every recguru_amd/csrc/*.hip to ISA first (a few minutes)
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VEC = re.compile(r"^\s*(v_|ds_|global_|scratch_|buffer_|flat_)")
LABEL = re.compile(r"^([.\w$]+):")
RESTORE = re.compile(r"^\s*s_or_b64\s+exec,\s*exec,")
# instructions that may sit between a label and the exec restore: scalar ALU / moves / waits, comments, directives
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wno-unused-value", "-Wno-pass-failed"]


SAVEEXEC = re.compile(r"^\s*s_and_saveexec_b64\s+(s\[\d+:\d+\])")
EXECZ = re.compile(r"^\s*s_cbranch_execz\s+([.\w$]+)")
RESTORE_OF = re.compile(r"^\s*s_or_b64\s+exec,\s*exec,\s*(s\[\d+:\d+\])")
SPILL = re.compile(r"^\s*(v_accvgpr_write_b32\s+a\d+,\s*v\d+|v_accvgpr_read_b32|scratch_store|scratch_load|buffer_store_dword.*offen|buffer_load_dword.*offen)")


def kernels(path):
    """{kernel: [(line number, text)]} -- instruction and label lines of every function in the file."""
    out, cur = {}, None
    with open(path, errors="replace") as f:
        for no, line in enumerate(f, 1):
            m = LABEL.match(line)
            if m and not m.group(1).startswith("."):
                cur = out.setdefault(m.group(1), [])
            s = line.strip()
            if cur is None or not s or s.startswith(";"):
                continue
            if s.startswith(".") and not LABEL.match(line):
                continue                          # directive
            cur.append((no, s))
    return out


def screen(path):
    """Two findings per kernel:
    join  -- `s_and_saveexec_b64 sX, c ; s_cbranch_execz .L` whose join block .L runs vector instructions BEFORE its
             `s_or_b64 exec, exec, sX` (they execute under the narrowed -- or empty -- exec): the miscompile;
    tail  -- spill-like copies (VGPR <-> AGPR, scratch) immediately in front of ANY exec restore (branch-free `if` bodies have no
             join label; a spill at their end is the same defect): reported as a warning, it can also be the body's own code."""
    bad, warn = [], []
    for kernel, ins in kernels(path).items():
        labels = {s[:-1] if s.endswith(":") else s.split(":")[0]: i for i, (no, s) in enumerate(ins) if LABEL.match(s)}
        for i, (no, s) in enumerate(ins):
            m = SAVEEXEC.match(s)
            if m and i + 1 < len(ins):
                saved = m.group(1)
                mb = EXECZ.match(ins[i + 1][1])
                if mb and mb.group(1) in labels:
                    j = labels[mb.group(1)] + 1
                    pending = []
                    while j < len(ins):
                        t = ins[j][1]
                        mr = RESTORE_OF.match(t)
                        if mr:
                            if mr.group(1) == saved and pending:
                                bad.append((kernel, ins[j][0], mb.group(1), pending))
                            break
                        if LABEL.match(t) or t.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_barrier", "s_and_saveexec", "s_setpc", "s_swappc")):
                            break
                        if VEC.match(t):
                            pending.append("%d: %s" % ins[j])
                        j += 1
            if RESTORE.match(s):
                j, run = i - 1, []
                while j >= 0 and (SPILL.match(ins[j][1]) or (ins[j][1].startswith("s_") and not ins[j][1].startswith(("s_cbranch", "s_branch", "s_barrier", "s_or_b64 exec", "s_and_saveexec")))):
                    if SPILL.match(ins[j][1]):
                        run.append("%d: %s" % ins[j])
                    j -= 1
                if run:
                    warn.append((kernel, no, "-", run[::-1]))
    return bad, warn


def build_isa(outdir):
    srcs = sorted(glob.glob(os.path.join(ROOT, "recguru_amd", "csrc", "*.hip")))
    procs = []
    for s in srcs:
        o = os.path.join(outdir, os.path.basename(s)[:-4] + ".s")
        procs.append((o, subprocess.Popen(["hipcc"] + FLAGS + ["-S", "--cuda-device-only", s, "-o", o], stdout=subprocess.DEVNULL,
                                          stderr=subprocess.DEVNULL)))
        if len(procs) >= 4:
            procs.pop(0)[1].wait()
    for _, p in procs:
        p.wait()
    return sorted(glob.glob(os.path.join(outdir, "*.s")))


def main():
    args = sys.argv[1:]
    tmp = None
    if args and args[0] == "--build":
        tmp = tempfile.mkdtemp(prefix="rg_isa_")
        files = build_isa(tmp)
    else:
        files = args
    if not files:
        sys.exit(__doc__)
    total = 0
    for fn in files:
        bad, warn = screen(fn)
        total += len(bad)
        for tag, lst in (("MISCOMPILED JOIN", bad), ("spill-like copies in front of an exec restore (check)", warn)):
            for kernel, no, block, ins in lst:
                print("%s: %s: %s, exec restore at line %d%s: %d instruction(s)" % (os.path.basename(fn), kernel, tag, no,
                                                                                    (" (join block %s)" % block) if block != "-" else "", len(ins)))
                for i in ins[:4]:
                    print("      " + i)
                if len(ins) > 4:
                    print("      ... %d more" % (len(ins) - 4))
        print("%-24s %s" % (os.path.basename(fn), ("FLAGGED: %d join block(s)" % len(bad)) if bad else ("clean (%d warnings)" % len(warn))))
    print("join blocks that run vector instructions under the narrowed exec: %d" % total)
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
