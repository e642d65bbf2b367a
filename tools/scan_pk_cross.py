#!/usr/bin/env python3
"""Static guard behind DESIGN.md 16.1: on gfx950 a packed fp32 operation (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) that takes a
LOW-half operand from the HIGH dword of a register pair (op_sel:[..1..]) intermittently lost its low-half result for lanes 16-31 /
48-63 in isg_gatv2_tile_conv (tools/flake/: 48-417 wrong launches of 1600 in every variant that has such an operation, 0 of 1600 in
the variants without -- same loop, same LDS reads, one v_mov_b32 more).  The library must not contain one.

  python3 tools/scan_pk_cross.py                 every code object of the BUILT libraries: libisg_hip.so, libisg_hip_strict.so and
                                                 whatever tools/build_variant.py left in tools/_build/ (tests/test_host_cpu.py runs
                                                 the first two; a variant library is scanned by build_variant.py when it is made)
  python3 tools/scan_pk_cross.py lib.so ...      the given libraries
  python3 tools/scan_pk_cross.py file.hip ...    compile the given sources to assembly and scan that (while editing a kernel)
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "intrinsic-subgraph-generation-for-vqa_amd", "csrc")
LIB = os.path.join(CSRC, "libisg_hip.so")
STRICT_LIB = os.path.join(CSRC, "libisg_hip_strict.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
# every packed fp32 ARITHMETIC operation of gfx950 (v_pk_mov_b32 moves bits, it computes nothing: its op_sel is how the compiler
# swaps halves and is not the form that failed)
PK = re.compile(r"\b(v_pk_(?:fma|mul|add|max|min|maximum3|minimum3)_f32)\b(.*)")
CROSS = re.compile(r"op_sel:\[[01,]*1")          # some source's LOW half comes from a high dword


def scan_text(lines, symbol_re):
    """-> (packed fp32 operations seen, [(kernel, instruction)] of the cross-selecting ones)"""
    kern, total, hits = None, 0, []
    for line in lines:
        m = symbol_re.match(line)
        if m:
            kern = m.group(1)
            continue
        m = PK.search(line)
        if m:
            total += 1
            text = (m.group(1) + m.group(2)).split("//")[0].split(";")[0].strip()
            if CROSS.search(text):
                hits.append((kern, text))
    return total, hits


def scan_library(lib=LIB):
    tmp = tempfile.mkdtemp(prefix="isg_scan_")
    try:
        copy = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, copy)               # llvm-objdump --offloading extracts beside its input
        subprocess.run([OBJDUMP, "--offloading", copy], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        objs = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)
        if not objs:
            raise RuntimeError(f"no gfx950 code object found in {lib}")
        total, hits = 0, []
        for f in objs:
            dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            t, h = scan_text(dis.splitlines(), re.compile(r"^[0-9a-f]+ <(\S+)>:"))
            total += t
            hits += h
        return len(objs), total, hits
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def scan_source(src):
    with tempfile.TemporaryDirectory(prefix="isg_scan_") as tmp:
        asm = os.path.join(tmp, "k.s")
        sys.path.insert(0, ROOT)
        from __graft_entry__ import flags_for               # the build's own flags for this source
        subprocess.run(["/opt/rocm/bin/hipcc", *flags_for(src), "--cuda-device-only", "-S", os.path.abspath(src), "-o", asm],
                       check=True, stderr=subprocess.DEVNULL, cwd=CSRC)
        return scan_text(open(asm), re.compile(r"^(_Z\w+):"))


def demangle(name):
    try:
        return subprocess.run(["c++filt", name or "?"], capture_output=True, text=True).stdout.strip()[:100]
    except OSError:
        return name


def built_libraries():
    """The shipped pair and any variant library tools/build_variant.py left behind."""
    var = os.path.join(ROOT, "tools", "_build")
    extra = sorted(os.path.join(var, f) for f in os.listdir(var) if f.endswith(".so")) if os.path.isdir(var) else []
    return [p for p in (LIB, STRICT_LIB) if os.path.exists(p)] + extra


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and all(a.endswith(".hip") for a in args):
        total, hits = 0, []
        for src in args:
            t, h = scan_source(src)
            total += t
            hits += h
        where = ", ".join(os.path.basename(s) for s in args)
    else:
        n, total, hits = 0, 0, []
        libs = args or built_libraries()
        for lib in libs:
            o, t, h = scan_library(lib)
            n, total, hits = n + o, total + t, hits + [(f"{os.path.basename(lib)}: {k}", x) for k, x in h]
        where = f"{n} code objects of {', '.join(os.path.relpath(l, ROOT) for l in libs)}"
    for kern, text in hits:
        print(f"{demangle(kern)}\n    {text}")
    print(f"{len(hits)} cross-selecting packed fp32 operation(s) among {total} in {where}")
    sys.exit(1 if hits else 0)
