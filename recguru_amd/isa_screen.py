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
stale: three rows of four of every tile wrong, and different from launch to launch.

The screen: for every `s_and_saveexec_b64 sX ; s_cbranch_execz .L` the join block .L may run no vector instruction (v_* / ds_* /
global_* / scratch_* / buffer_*) in front of its `s_or_b64 exec, exec, sX` -- whatever stands there executes under the narrowed
(or empty) exec.  Spill-like copies directly in front of any other exec restore are listed as warnings (branch-free `if` bodies have
no join label; there the copies can also be the body's own code).

Second screen (round 4, DESIGN.md 2a finding 1): `v_pk_add_f32 vD, vA, vB op_sel:[0,1]` -- a packed-f32 operation whose LOW result
takes the HIGH half of its second source pair.  On gfx950 the low result of that form intermittently comes out as if the selected
source were 0, in lanes 48-63 only, whenever ANOTHER wave on the SIMD is issuing MFMAs (two workgroups per CU, or another process
on the GPU; stand-alone reproducer: tools/hazard/opsel_repro.hip):
the fused block's LayerNorm subtracted a mean of 0 from one element of a row in 4 % of the rows of every bench-shape launch for
three rounds.  Established on the ISA of the faulty kernel (tools/hazard/isa_variants.py, 720 instances): dropping the op_sel,
replacing the instruction by two v_sub_f32, or moving the selection to src0 of a v_pk_fma_f32 (op_sel:[1,0,0]) removes the defect;
wait states, sleeps, waits on LDS and a copied source do not.  hipcc forms the instruction by itself when two per-row statistics end
up in one register pair (SLP vectorisation).  The stand-alone probe maps the hazard to the SECOND source operand: v_pk_add_f32 and
v_pk_mul_f32 with op_sel:[0,1] and v_pk_fma_f32 with op_sel:[0,1,0] fail (src1 read as 0); op_sel:[1,0] (src0) of v_pk_add_f32 /
v_pk_mov_b32, op_sel:[1,0,0] and op_sel:[0,0,1] (src0, src2) of v_pk_fma_f32 and every op_sel_hi form do not.
packed_opsel() flags every packed-f32 operation with the op_sel bit of its second source set: the build fails on it.

recguru_amd/build.py keeps the device ISA of every source (-save-temps) and calls screen() and packed_opsel() on it: a flagged
kernel fails the build.  tools/isa_exec_screen.py is the command-line front.
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))        # repo root (this file: recguru_amd/isa_screen.py)
# (v_readlane / v_writelane / v_readfirstlane address lanes explicitly and ignore exec: SGPR spills to VGPR lanes in front of an exec
# restore are harmless)
VEC = re.compile(r"^\s*(?!v_writelane_b32|v_readlane_b32|v_readfirstlane_b32)(v_|ds_|global_|scratch_|buffer_|flat_)")
LABEL = re.compile(r"^([.\w$]+):")
RESTORE = re.compile(r"^\s*s_or_b64\s+exec,\s*exec,")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wno-unused-value", "-Wno-pass-failed"]
SAVEEXEC = re.compile(r"^\s*s_(?:and|andn2)_saveexec_b64\s+(s\[\d+:\d+\])")     # the exec-NARROWING saveexec forms
EXECZ = re.compile(r"^\s*s_cbranch_execz\s+([.\w$]+)")
RESTORE_OF = re.compile(r"^\s*s_or_b64\s+exec,\s*exec,\s*(s\[\d+:\d+\])")
SPILL = re.compile(r"^\s*(v_accvgpr_write_b32\s+a\d+,\s*v\d+|v_accvgpr_read_b32|scratch_store|scratch_load|buffer_store_dword.*offen|buffer_load_dword.*offen)")
STOP = ("s_cbranch", "s_branch", "s_endpgm", "s_barrier", "s_and_saveexec", "s_setpc", "s_swappc")


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
    """Two findings per kernel, each a list of (kernel, line of the exec restore, join label or "-", [instructions]):
    join  -- `s_and_saveexec_b64 sX, c ; s_cbranch_execz .L` whose join block .L runs vector instructions BEFORE its
             `s_or_b64 exec, exec, sX` (they execute under the narrowed -- or empty -- exec): the miscompile;
    tail  -- spill-like copies (VGPR <-> AGPR, scratch) immediately in front of ANY exec restore (branch-free `if` bodies have no
             join label; a spill at their end is the same defect): a warning, it can also be the body's own code."""
    bad, warn = [], []
    for kernel, ins in kernels(path).items():
        labels = {s.split(":")[0]: i for i, (no, s) in enumerate(ins) if LABEL.match(s)}
        for i, (no, s) in enumerate(ins):
            m = SAVEEXEC.match(s)
            if m and i + 1 < len(ins):
                saved = m.group(1)
                nx = i + 1                      # scalar instructions may be scheduled between the saveexec and its branch
                while nx + 1 < len(ins) and ins[nx][1].startswith("s_") and not ins[nx][1].startswith(STOP + ("s_or_b64 exec",)) \
                        and saved not in ins[nx][1].split(None, 1)[-1].split(",")[0]:
                    nx += 1
                mb = EXECZ.match(ins[nx][1])
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
                        if LABEL.match(t) or t.startswith(STOP):
                            break
                        if VEC.match(t):
                            pending.append("%d: %s" % ins[j])
                        j += 1
            if RESTORE.match(s):
                j, run = i - 1, []
                while j >= 0 and (SPILL.match(ins[j][1]) or (ins[j][1].startswith("s_") and not ins[j][1].startswith(STOP + ("s_or_b64 exec",)))):
                    if SPILL.match(ins[j][1]):
                        run.append(j)
                    j -= 1
                # Triage (round 5): a VGPR -> AGPR copy under a narrowed exec is the conditional body's OWN masked write when the
                # source VGPR was DEFINED inside the same straight-line block (after the block's label / the exec-narrowing
                # instruction): the lanes outside exec keep the variable's previous value in the AGPR, as the source says.  It is
                # a SPILL (the defect: lanes outside exec keep whatever the slot held) when the source was live before the block.
                # Copies of the first kind are dropped; the rest stay warnings -- and `unresolved` marks them for the build.
                left = []
                for jj in run:
                    if _under_restored_mask(ins, jj, i):
                        continue
                    mw = ACC_WRITE.match(ins[jj][1])
                    if mw and _defined_in_block(ins, jj, int(mw.group(1))):
                        continue
                    mr = ACC_READ.match(ins[jj][1])              # AGPR -> VGPR of a value this very block computed (an MFMA result)
                    if mr and _defined_in_block(ins, jj, int(mr.group(1)), "a"):
                        continue
                    left.append("%d: %s" % ins[jj])
                if left:
                    warn.append((kernel, no, "-", left[::-1]))
    return bad, warn


OR_SAVEEXEC = re.compile(r"^\s*s_or_saveexec_b64\s+(s\[\d+:\d+\]),\s*(s\[\d+:\d+\])\s*$")
XOR_EXEC = re.compile(r"^\s*s_xor_b64\s+exec,\s*exec,\s*(s\[\d+:\d+\])\s*$")


def _under_restored_mask(ins, at, restore):
    """(round 6) True when the copy ins[at] runs under EXACTLY the mask the restore ins[restore] re-establishes -- the structurizer's
    else-flow block with an empty else body:

            s_or_saveexec_b64 sX, sX          ; sX' = exec (E0), exec = S | E0
            <copy>                            ; under S | E0
            s_xor_b64 exec, exec, sX          ; exec = (S | E0) ^ E0 = S & ~E0      (the else lanes)
            s_or_b64 exec, exec, sX           ; exec = (S & ~E0) | E0 = S | E0      -- the mask the copy ran under

    Nothing but scalar instructions that leave exec and sX alone may stand between the four.  (Met in round 6 in the f32-tier
    instantiations of post_attn_fwd_kernel once the kernel was compiled without packed-f32 instructions: a v_accvgpr_read_b32 that the
    scheduler moved up from behind the restore.)"""
    mr = RESTORE_OF.match(ins[restore][1])
    if not mr:
        return False
    reg = mr.group(1)
    j, seen_xor = restore - 1, False
    while j > at:                                   # between the copy and the restore: one xor of exec with sX, other copies, plain scalars
        t = ins[j][1]
        mx = XOR_EXEC.match(t)
        if mx:
            if seen_xor or mx.group(1) != reg:
                return False
            seen_xor = True
        elif t.startswith(BLOCK_START) or t.startswith(STOP) or LABEL.match(t) or (t.startswith("s_") and reg in t):
            return False
        elif not (t.startswith("s_") or SPILL.match(t)):
            return False
        j -= 1
    if not seen_xor:
        return False
    j = at - 1
    while j >= 0:                                   # in front of the copy: other copies / plain scalars, then the s_or_saveexec sX, sX
        t = ins[j][1]
        mo = OR_SAVEEXEC.match(t)
        if mo:
            return mo.group(1) == reg and mo.group(2) == reg
        if LABEL.match(t) or t.startswith(BLOCK_START) or t.startswith(STOP) or (t.startswith("s_") and reg in t):
            return False
        if not (t.startswith("s_") or SPILL.match(t)):
            return False
        j -= 1
    return False


ACC_READ = re.compile(r"^\s*v_accvgpr_read_b32\s+v\d+,\s*a(\d+)\s*$")
ADEST = re.compile(r"^\s*(?:v_mfma|v_smfmac|v_accvgpr_write)\S*\s+(a\d+|a\[\d+:\d+\])")
ACC_WRITE = re.compile(r"^\s*v_accvgpr_write_b32\s+a\d+,\s*v(\d+)\s*$")
DEST = re.compile(r"^\s*(?:v_|ds_read|ds_load|global_load|buffer_load|scratch_load|flat_load)\S*\s+(v\d+|v\[\d+:\d+\])")
BLOCK_START = ("s_and_saveexec", "s_or_saveexec", "s_andn2_saveexec", "s_xor_saveexec", "s_or_b64 exec", "s_and_b64 exec", "s_andn2_b64 exec",
               "s_xor_b64 exec", "s_mov_b64 exec")


def _defined_in_block(ins, at, vreg, bank="v"):
    """True when register `vreg` of the VGPR ("v") or AGPR ("a") bank is written by an instruction between the start of the
    straight-line block that holds ins[at] (the nearest label or exec-changing instruction in front of it) and ins[at]."""
    j = at - 1
    while j >= 0:
        t = ins[j][1]
        if LABEL.match(t) or t.startswith(BLOCK_START) or t.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
            return False
        m = (ADEST if bank == "a" else DEST).match(t)
        if m and (bank == "a" or not t.startswith(("v_cmp", "v_cmpx", "v_accvgpr_write"))):
            d = m.group(1)
            if d[1] == "[":
                lo, hi = (int(x) for x in d[2:-1].split(":"))
            else:
                lo = hi = int(d[1:])
            if lo <= vreg <= hi:
                return True
        j -= 1
    return False


PK_F32 = re.compile(r"^\s*(v_pk_(?:add|mul|fma)_f32)\s+.*?\bop_sel:\[([01,]+)\]")


def packed_opsel(path):
    """[(kernel, line, text)]: packed-f32 operations that feed a LOW result from the HIGH half of their SECOND source (module
    docstring)."""
    out = []
    for kernel, ins in kernels(path).items():
        for no, s in ins:
            m = PK_F32.match(s)
            if not m:
                continue
            bits = [int(b) for b in m.group(2).split(",")]
            if len(bits) > 1 and bits[1]:      # src1 (a bit on src0 measured clean for add, fma and v_pk_mov_b32; on src2 for fma)
                out.append((kernel, no, s))
    return out


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
    if args and args[0] == "--build":
        files = build_isa(tempfile.mkdtemp(prefix="rg_isa_"))
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
        pk = packed_opsel(fn)
        total += len(pk)
        for kernel, no, text in pk[:6]:
            print("%s: %s: HIGH-HALF SELECT ON A PACKED-F32 SOURCE, line %d: %s" % (os.path.basename(fn), kernel, no, text))
        print("%-24s %s" % (os.path.basename(fn), ("FLAGGED: %d join block(s), %d packed-f32 op_sel form(s)" % (len(bad), len(pk))) if bad or pk
                            else ("clean (%d warnings)" % len(warn))))
    print("flagged (join blocks under a narrowed exec + packed-f32 high-half selects): %d" % total)
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
