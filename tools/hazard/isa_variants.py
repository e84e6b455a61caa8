"""ISA-level experiments on the fused block that carried DESIGN 2a's finding 1 (wrong LayerNorm-1 elements in lanes 48-63 at two
workgroups per CU).  Checks out that kernel file as it was before the fix (commit 2779adb^: rsqrtf() in ln_regs), compiles its device
side to ISA with today's hipcc, applies one regex edit per named variant, rebuilds every variant from the edited ISA
(tools/hazard/roundtrip.sh: no second trip through the register allocator) and links it with the current objects of the other kernel
files into tools/hazard/v_<name>.so.  On the GPU box:  bash tools/hazard/run_variants.sh   (tools/race_post_attn.py at M = 65536 on
each library: a launch that differs from the first one is the defect).

    python tools/hazard/isa_variants.py <variant> [<variant> ...]        then   gpurun -- 'bash tools/hazard/run_variants.sh 6'

Variants and what round 4 measured with them (differing launches of 36; inference / training / decoder forms, dropout 0 and 0.5):
    roundtrip        unchanged ISA through the round trip                                             36   (the defect survives it)
    sleep_after_rsq  s_sleep 8 behind every v_rsq_f32                                                 36
    sleep_between    s_sleep 8 between a v_rsq_f32 and a ds_read_b128 that overwrites its source      36
    copy_src         the v_rsq_f32 source copied to a fresh register first                            36
    dsw_wait         s_waitcnt lgkmcnt(0) behind every LDS store whose data registers are rewritten   36
    sub_nop          s_nop 3 in front of every v_pk_add_f32 ... op_sel:[0,1]                          36
    pk_noneg         ... the same instructions without their neg modifiers (op_sel kept)              36
    pk_noopsel       ... without the op_sel (arithmetic changes, determinism is what is tested)         0
    pk_lowbcast      ... with op_sel_hi:[1,0] instead (the LOW half to both results)                    0
    sub_scalar       ... replaced by two v_sub_f32 (same arithmetic)                                    0
    as_fma           ... as v_pk_fma_f32 d, b, (-1,-1), a op_sel:[1,0,0] where registers are free      0 in the rewritten kernels
The trigger is the op_sel:[0,1] of v_pk_add_f32 (LOW result from the HIGH half of src1); recguru_amd/isa_screen.py fails the build on it.
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OLD = "2779adb^"
RSQ = re.compile(r"^\s*v_rsq_f32_e32\s+(v\d+),\s*(v\d+)\s*$")
DSR = re.compile(r"^\s*ds_read_b128\s+v\[(\d+):(\d+)\]")
PKA = re.compile(r"^\s*v_pk_add_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel:\[0,1\] neg_lo:\[0,1\] neg_hi:\[0,1\]\s*$")


def is_ins(l):
    s = l.strip()
    return bool(s) and not s.startswith((";", ".")) and not s.endswith(":")


def vregs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def old_tree(tmp):
    """The kernel file and the headers of commit OLD under tmp/, device ISA in tmp/fused_dev.s."""
    os.makedirs(os.path.join(tmp, "recguru_amd", "csrc"))
    os.makedirs(os.path.join(tmp, "include"))
    for rel in ("recguru_amd/csrc/fused.hip", "recguru_amd/csrc/rg_common.hip.h", "include/recguru_hip.h"):
        with open(os.path.join(tmp, rel), "wb") as f:
            f.write(subprocess.check_output(["git", "show", "%s:%s" % (OLD, rel)], cwd=ROOT))
    src = os.path.join(tmp, "recguru_amd", "csrc", "fused.hip")
    dev = os.path.join(tmp, "fused_dev.s")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wno-unused-value", "-Wno-pass-failed",
                           "-S", "--cuda-device-only", os.path.basename(src), "-o", dev], cwd=os.path.dirname(src), stderr=subprocess.DEVNULL)
    return src, dev


def variant(name, lines):
    txt = "\n".join(lines)
    nfree = {m.group(1): int(re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", m.group(2)).group(1))
             for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S)}
    out, n, cur = [], 0, None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1)
        roomy = cur is not None and nfree.get(cur, 999) <= 240          # v250 / v251 are free there
        r, pk = RSQ.match(l), PKA.match(l)
        nxt = ""
        if r:
            j = i + 1
            while j < len(lines) and not is_ins(lines[j]):
                j += 1
            nxt = lines[j] if j < len(lines) else ""
        if name == "sleep_after_rsq" and r:
            out += [l, "\ts_sleep 8"]; n += 1; continue
        if name == "sleep_between" and r and DSR.match(nxt):
            out += [l, "\ts_sleep 8"]; n += 1; continue
        if name == "copy_src" and r and roomy and DSR.match(nxt):
            d = DSR.match(nxt)
            if int(d.group(1)) <= int(r.group(2)[1:]) <= int(d.group(2)):
                out += ["\tv_mov_b32_e32 v250, %s" % r.group(2), "\ts_nop 1", "\tv_rsq_f32_e32 %s, v250" % r.group(1)]; n += 1; continue
        if name == "dsw_wait" and l.strip().startswith("ds_write"):
            ops = [t.strip() for t in l.strip().split(None, 1)[1].split(" offset")[0].split(",")]
            data = set().union(*[vregs(o) for o in ops[1:]])
            j, k, hit = i + 1, 0, False
            while j < len(lines) and k < 3:
                if is_ins(lines[j]):
                    k += 1
                    t = lines[j].strip()
                    if t.startswith(("v_", "ds_read", "global_load")) and not t.startswith("v_cmp") and len(t.split(None, 1)) > 1:
                        hit = hit or bool(vregs(t.split(None, 1)[1].split(",")[0].strip()) & data)
                j += 1
            out.append(l)
            if hit:
                out.append("\ts_waitcnt lgkmcnt(0)"); n += 1
            continue
        if pk:
            d, a, b = int(pk.group(1)), int(pk.group(3)), int(pk.group(5))
            base = l.split(" op_sel:")[0]
            if name == "sub_nop":
                out += ["\ts_nop 3", l]; n += 1; continue
            if name == "pk_noneg":
                out.append(base + " op_sel:[0,1]"); n += 1; continue
            if name == "pk_noopsel":
                out.append(base + " neg_lo:[0,1] neg_hi:[0,1]"); n += 1; continue
            if name == "pk_lowbcast":
                out.append(base + " op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]"); n += 1; continue
            if name == "sub_scalar":
                out += ["\tv_sub_f32_e32 v%d, v%d, v%d" % (d, a, b + 1), "\tv_sub_f32_e32 v%d, v%d, v%d" % (d + 1, a + 1, b + 1)]; n += 1; continue
            if name == "as_fma" and roomy:
                out.append("\tv_pk_fma_f32 v[%d:%d], v[%d:%d], v[250:251], v[%d:%d] op_sel:[1,0,0]" % (d, d + 1, b, b + 1, a, a + 1)); n += 1; continue
        if name == "as_fma" and m and roomy:
            out += [l, "\tv_mov_b32_e32 v250, -1.0", "\tv_mov_b32_e32 v251, -1.0"]; continue
        out.append(l)
    t = "\n".join(out)
    if name in ("copy_src", "as_fma"):       # the kernels that use v250 / v251 declare them
        def bump(mm):
            k, b = mm.group(1), mm.group(2)
            if nfree.get(k, 999) <= 240:
                b = re.sub(r"(\.amdhsa_next_free_vgpr\s+)\d+", r"\g<1>256", b)
                b = re.sub(r"(\.amdhsa_accum_offset\s+)\d+", r"\g<1>256", b)
            return ".amdhsa_kernel " + k + b + ".end_amdhsa_kernel"
        t = re.sub(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", bump, t, flags=re.S)
    return t, n


def main():
    names = sys.argv[1:]
    if not names:
        sys.exit(__doc__)
    subprocess.check_call([sys.executable, "-m", "recguru_amd.build"], cwd=ROOT, stdout=subprocess.DEVNULL)
    others = [os.path.join(ROOT, "recguru_amd", "build", f) for f in os.listdir(os.path.join(ROOT, "recguru_amd", "build"))
              if f.endswith(".o") and f != "fused.o"]
    with tempfile.TemporaryDirectory(prefix="rg_hazard_") as tmp:
        src, dev = old_tree(tmp)
        lines = open(dev).read().split("\n")
        for name in names:
            t, n = (("\n".join(lines), 0) if name == "roundtrip" else variant(name, lines))
            s_path, o_path = os.path.join(tmp, "var_%s.s" % name), os.path.join(tmp, "var_%s.o" % name)
            open(s_path, "w").write(t)
            subprocess.check_call([os.path.join(ROOT, "tools", "hazard", "roundtrip.sh"), src, s_path, o_path], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            so = os.path.join(ROOT, "tools", "hazard", "v_%s.so" % name)
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, o_path] + others)
            print("%-16s %4d edits -> %s" % (name, n, os.path.relpath(so, ROOT)))


if __name__ == "__main__":
    main()
