#!/usr/bin/env python
"""Fold the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) into
profiles/pmc_traffic.json: per kernel, HBM-side bytes per launch.

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dirF> -- python3 bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d <dirW> -- python3 bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline
  python tools/pmc_traffic.py <dirF> <dirW> [out.json]

Units and gfx950 corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB
(bytes = value * 1024); FETCH_SIZE tallies the L2's 128-byte fabric read requests at 64 B, so read bytes =
2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for 16-byte-per-lane stores and float atomics.  Infinity-Cache hits are
counted (these are L2 <-> fabric bytes, an upper bound on DRAM traffic).
"""
import csv
import glob
import json
import os
import re
import sys


def _targs_mangled(t):
    """Template arguments of an Itanium-mangled kernel name (only the forms our kernels use)."""
    out, i = [], 0
    while i < len(t):
        if t.startswith("DF16b", i):
            out.append("bf16"); i += 5
        elif t[i] == "f":
            out.append("f32"); i += 1
        elif t[i] == "L":                     # Li14E / Lb0E
            j = t.index("E", i)
            out.append(t[i + 2:j]); i = j + 1
        else:
            break
    return out


def short(name):
    """rocprofv3 kernel name (mangled or demangled) -> the label bench.py uses."""
    m = re.match(r"_Z(\d+)", name)
    if m:
        n = int(m.group(1))
        base = name[m.end():m.end() + n]
        rest = name[m.end() + n:]
        parts = _targs_mangled(rest[1:]) if rest.startswith("I") else []
    else:
        m = re.match(r"(?:void )?(\w+)(?:<(.*)>)?\(", name)
        base, targs = (m.group(1), m.group(2) or "") if m else (name.split("(")[0], "")
        # rocprofv3's demangler renders the bf16 template argument (DF16b) as "bool _Accum"
        targs = targs.replace("bool _Accum", "bf16").replace("__hip_bfloat16", "bf16").replace("__bf16", "bf16").replace("float", "f32")
        parts = [p.strip() for p in targs.split(",")] if targs else []
    if base in ("post_attn_fwd_kernel", "attn_fwd_kernel", "gemm_tn_kernel"):
        return "%s<%s>" % (base, parts[0])
    if base == "attn_bwd_bf16_kernel":
        return "attn_bwd_kernel<bf16>"
    if base == "attn_bwd_kernel":
        return "attn_bwd_kernel<f32>"
    if base in ("gemm_nt_kernel", "gemm_ws_kernel", "gemm_tn_big_kernel", "gemm_tn_dma_kernel"):
        return "%s<%s,%s>" % (base, parts[0], parts[1])
    return base


def demangle(n):
    return n


def collect(d, counter):
    out, cache = {}, {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            raw = row["Kernel_Name"]
            if raw not in cache:
                cache[raw] = short(demangle(raw))
            e = out.setdefault(cache[raw], [0, 0.0])
            e[0] += 1
            e[1] += float(row["Counter_Value"])
    return out


def main():
    dF, dW = sys.argv[1], sys.argv[2]
    outp = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    F, W = collect(dF, "FETCH_SIZE"), collect(dW, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(F) | set(W)):
        nf, vf = F.get(k, [0, 0.0])
        nw, vw = W.get(k, [0, 0.0])
        rd = 2.0 * vf * 1024 / nf if nf else None          # gfx950: FETCH_SIZE reads half
        wr = vw * 1024 / nw if nw else None
        kernels[k] = {"launches_fetch_pass": nf, "launches_write_pass": nw,
                      "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                      "hbm_bytes_per_launch": (rd or 0.0) + (wr or 0.0)}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of "
                         "`python3 bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline`",
               "corrections": "bytes = KiB * 1024; read bytes = 2 * FETCH_SIZE (gfx950, MI355X_MICROARCH.md HBM section); "
                              "L2<->fabric bytes, Infinity-Cache hits included",
               "kernels": kernels}, open(outp, "w"), indent=1, sort_keys=True)
    for k, v in sorted(kernels.items(), key=lambda kv: -(kv[1]["hbm_bytes_per_launch"] * max(kv[1]["launches_fetch_pass"], 1)))[:25]:
        print("%-34s n=%5d  read %9.1f MB  write %9.1f MB" % (k, v["launches_fetch_pass"], (v["read_bytes_per_launch"] or 0) / 1e6,
                                                             (v["write_bytes_per_launch"] or 0) / 1e6))


if __name__ == "__main__":
    main()
