#!/usr/bin/env python3
"""Variant builds of isg_gatv2_tile_conv's aggregation loop (C2) for the ONE experiment that separates the hypotheses about its
round-4 intermittent wrong sums.  Each variant is the shipped source with the in-edge loop replaced, compiled
to its own object and linked against the shipped objects -> tools/_build/libisg_agg_<name>.so (git-ignored, travels to the GPU box).
  python3 tools/flake/make_variants.py            (CPU; ~40 s per variant, run in parallel)"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "intrinsic-subgraph-generation-for-vqa_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_build")
SRC = open(os.path.join(CSRC, "isg_layer_tile.hip")).read()

BEGIN = "      // One in-edge per trip,"
END = "      if (a.bias) { o.x += b4.x;"
i0, i1 = SRC.index(BEGIN), SRC.index(END)

LOAD = """const float4 {u} = *reinterpret_cast<const float4 *>(&sXl[s_tab[{s}].x][fr * 4]);"""
FMA = "{o}.x = fmaf({u}.x, {w}, {o}.x); {o}.y = fmaf({u}.y, {w}, {o}.y); {o}.z = fmaf({u}.z, {w}, {o}.z); {o}.w = fmaf({u}.w, {w}, {o}.w);"
ASM_FMA = ('asm volatile("v_fma_f32 %0, %4, %8, %0\\n\\tv_fma_f32 %1, %5, %8, %1\\n\\tv_fma_f32 %2, %6, %8, %2\\n\\tv_fma_f32 %3, %7, %8, %3" '
           ': "+v"({o}.x), "+v"({o}.y), "+v"({o}.z), "+v"({o}.w) : "v"({u}.x), "v"({u}.y), "v"({u}.z), "v"({u}.w), "v"({w}));')


def rolled(unroll):
    return f"""#pragma unroll {unroll}
      for (int s = rb; s < re; ++s) {{
        const float wm = s_w[s];
        {LOAD.format(u='u4', s='s')}
        {FMA.format(o='o', u='u4', w='wm')}
      }}
"""


def pairs(between="", after_loads="", fma0=None, fma1=None, acc1="o", tail=""):
    fma0 = fma0 or FMA.format(o="o", u="u0", w="w0")
    fma1 = fma1 or FMA.format(o=acc1, u="u1", w="w1")
    return f"""      int s = rb;
      float4 o2 = make_float4(0.f, 0.f, 0.f, 0.f);
      (void)o2;
      if ((re - rb) & 1) {{
        const float w0 = s_w[s];
        {LOAD.format(u='u0', s='s')}
        {FMA.format(o='o', u='u0', w='w0')}
        ++s;
      }}
#pragma unroll 1
      for (; s < re; s += 2) {{
        float w0 = s_w[s], w1 = s_w[s + 1];
        {LOAD.format(u='u0', s='s')}
        {LOAD.format(u='u1', s='s + 1')}
        {after_loads}
        {fma0}
        {between}
        {fma1}
      }}
      {tail}
"""


VARIANTS = {
    "rolled": rolled(1),                                   # the shipped form (control: never differed)
    "unroll2": rolled(2),                                  # round 4's failing form (control: must fail for the rest to mean anything)
    "pairs": pairs(),                                      # the same pairs written by hand: does the pattern alone fail?
    "two_acc": pairs(acc1="o2", tail="o.x += o2.x; o.y += o2.y; o.z += o2.z; o.w += o2.w;"),   # no two writes to one accumulator inside a pair
    "nops": pairs(after_loads='asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_nop 7\\n\\ts_nop 7" ::: "memory");'),   # LDS data long landed
    "opaque_w": pairs(after_loads='asm volatile("" : "+v"(w0)); asm volatile("" : "+v"(w1));'),   # no weight through a high dword
    "scalar": pairs(fma0=ASM_FMA.format(o="o", u="u0", w="w0"), fma1=ASM_FMA.format(o="o", u="u1", w="w1")),   # no packed fp32
    "between_nop": pairs(between='asm volatile("s_nop 7" : "+v"(o.x), "+v"(o.y), "+v"(o.z), "+v"(o.w));'),   # issue distance between the pair's two writes
    # round 6 ("the operand form" vs "a timing window the form happens to open"): `opaque_w` differs from the
    # failing `pairs` by ONE v_mov_b32 in front of the first fma -- and by that instruction's issue slot.  These keep the slot and
    # the cross-dword select both: the same v_mov_b32 at the same place (one, and two), its result unused by the fmas.  If they fail
    # like `pairs`, the extra instruction's timing is not what cures `opaque_w`; the operand form is.
    "cross_plus_mov": pairs(after_loads='{ float dm; asm volatile("v_mov_b32 %0, %1" : "=v"(dm) : "v"(w1)); asm volatile("" :: "v"(dm)); }'),
    "cross_plus_2mov": pairs(after_loads='{ float dm, dn; asm volatile("v_mov_b32 %0, %1\\n\\tv_mov_b32 %2, %3" : "=v"(dm), "=v"(dn) : "v"(w1), "v"(w0)); '
                                         'asm volatile("" :: "v"(dm), "v"(dn)); }'),
}
# the next tile's requests issued AFTER the aggregation instead of under it (no VMEM in flight across C2), around the failing form
NO_INFLIGHT = "unroll2_no_inflight"


def source(name):
    if name == NO_INFLIGHT:
        s = SRC[:i0] + rolled(2) + SRC[i1:]
        req = "    TC_REQUEST_TILE(desc_n)     //"
        j0 = s.index(req)
        j1 = s.index("\n", j0) + 1
        line = s[j0:j1]
        s = s[:j0] + s[j1:]
        k = s.index("    TC_STAMP(7)")
        return s[:k] + line + s[k:]
    return SRC[:i0] + VARIANTS[name] + SRC[i1:]


def build(name):
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(CSRC, f"_agg_{name}.hip")             # beside the headers it includes
    obj = os.path.join(OUT, f"agg_{name}.o")
    open(src, "w").write(source(name))
    try:
        flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC"]
        subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-c", src, "-o", obj])
        subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "--cuda-device-only", "-S", src, "-o", os.path.join(OUT, f"agg_{name}.s")],
                              stderr=subprocess.DEVNULL)
    finally:
        os.remove(src)
    objs = [os.path.join(CSRC, "_obj", f) for f in sorted(os.listdir(os.path.join(CSRC, "_obj")))
            if f.endswith(".o") and not f.endswith(".strict.o") and f != "isg_layer_tile.o"]
    lib = os.path.join(OUT, f"libisg_agg_{name}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", obj, *objs, "-o", lib])
    os.remove(obj)
    return lib


if __name__ == "__main__":
    names = sys.argv[1:] or list(VARIANTS) + [NO_INFLIGHT]
    with ThreadPoolExecutor(max_workers=6) as pool:
        for lib in pool.map(build, names):
            print(lib)
