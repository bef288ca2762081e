#!/usr/bin/env python3
"""tools/isa_report.py -- per-kernel resources and instruction counts of the BUILT gfx950 code objects (no GPU, no recompile).

For every kernel in dasp_amd/csrc/build/{kernels,mgstep,devpack,mgx}.o: VGPRs, SGPRs, scratch bytes, spills, static LDS, kernarg bytes (the
code object's metadata note) and the number of s_load / s_buffer_load, global_load / global_store, scratch, ds, v_mfma and
buffer-wide instructions in its disassembly.  The table is what DESIGN.md's performance cliffs are stated in (scalar loads of the row tables,
the 64-register cap of the windowed kernels, no scratch in the plain kernels); tests/test_isa_guard.py asserts them on every build.

    python tools/isa_report.py            # markdown table on stdout
    python tools/isa_report.py --json     # the same as one JSON object
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
OBJS = ("kernels", "mgstep", "mgx")      # (devpack.o: the device packers + hipCUB's kernels, off the hot path -- pass --all to include it)


def demangle(names):
    try:
        # (binutils' c++filt does not know DF16_ = _Float16: demangle it as Dh = half)
        out = subprocess.run(["c++filt"], input="\n".join(n.replace("DF16_", "Dh") for n in names) + "\n", capture_output=True, text=True, check=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def short_name(d):
    """dasp::dasp_spmv_kernel<double, true, true, false, false>(dasp::DevArgs) -> dasp_spmv_kernel<double,1,1,0,0>"""
    d = re.sub(r"^void ", "", d)
    d = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", d)
    d = d.replace("dasp::", "").replace("(anonymous namespace)::", "")
    d = d.replace("true", "1").replace("false", "0").replace(" ", "")
    return d


def code_object(obj_path, work):
    """the gfx950 code object embedded in a host object (llvm-objdump --offloading writes it next to its input)"""
    local = os.path.join(work, os.path.basename(obj_path))
    shutil.copy(obj_path, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], capture_output=True, text=True, check=True)
    cands = [f for f in os.listdir(work) if f.startswith(os.path.basename(obj_path) + ".") and "amdgcn" in f]
    if not cands:
        raise RuntimeError("no gfx950 bundle inside " + obj_path)
    return os.path.join(work, cands[0])


def metadata(co):
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    kern, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(\S+)", line)
        if line.startswith("  - .") or line.startswith("  - "):
            if re.match(r"  - \.", line):          # a new kernel entry (two-space indent)
                cur = {}
                kern[id(cur)] = cur
        if m and cur is not None:
            k, v = m.group(1), m.group(2)
            if k in ("vgpr_count", "sgpr_count", "agpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "kernarg_segment_size",
                     "vgpr_spill_count", "sgpr_spill_count", "max_flat_workgroup_size"):
                cur[k] = int(v)
            elif k == "name":
                cur["name"] = v
    return {d["name"]: d for d in kern.values() if "name" in d}


COUNTS = (("s_load", r"\ts_load_"), ("s_buffer_load", r"\ts_buffer_load_"), ("global_load", r"\tglobal_load_"), ("global_store", r"\tglobal_store_"),
          ("flat", r"\tflat_(load|store)_"), ("scratch", r"\tscratch_(load|store)_"), ("ds", r"\tds_"), ("mfma", r"\tv_mfma_"), ("dpp", r"row_(shl|shr|ror)|dpp"),
          ("s_barrier", r"\ts_barrier"), ("waitcnt", r"\ts_waitcnt"))


def disassembly(co):
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
    out, cur = {}, None
    pats = [(k, re.compile(p)) for k, p in COUNTS]
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = out.setdefault(m.group(1), {k: 0 for k, _ in COUNTS})
            cur["insts"] = 0
            continue
        if cur is None or not line.startswith("\t"):
            continue
        cur["insts"] += 1
        for k, p in pats:
            if p.search(line):
                cur[k] += 1
    return out


def report(build_dir=None, objs=OBJS):
    build_dir = build_dir or os.path.join(ROOT, "dasp_amd", "csrc", "build")
    rows = {}
    with tempfile.TemporaryDirectory() as work:
        for o in objs:
            path = os.path.join(build_dir, o + ".o")
            if not os.path.exists(path):
                continue
            co = code_object(path, work)
            md, dis = metadata(co), disassembly(co)
            names = demangle(list(md))
            for n, d in md.items():
                r = dict(d)
                r.pop("name")
                r.update(dis.get(n, {}))
                r["object"] = o
                rows[short_name(names[n])] = r
    return rows


COLS = ("object", "vgpr_count", "sgpr_count", "private_segment_fixed_size", "vgpr_spill_count", "group_segment_fixed_size", "kernarg_segment_size",
        "insts", "s_load", "global_load", "global_store", "scratch", "ds", "mfma", "s_barrier")
HEAD = ("object", "VGPR", "SGPR", "scratch B", "VGPR spills", "LDS B", "kernarg B", "insts", "s_load", "global_load", "global_store", "scratch ops", "ds ops",
        "v_mfma", "s_barrier")


def main():
    rows = report(objs=OBJS + ("devpack",) if "--all" in sys.argv else OBJS)
    if "--json" in sys.argv:
        print(json.dumps(rows, indent=1, sort_keys=True))
        return
    print("| kernel | " + " | ".join(HEAD) + " |")
    print("|---|" + "---|" * len(HEAD))
    for k in sorted(rows):
        print("| `%s` | " % k + " | ".join(str(rows[k].get(c, "")) for c in COLS) + " |")


if __name__ == "__main__":
    main()
