"""Register / scratch / LDS use of every kernel in libdpr.so, read from the code-object metadata
(llvm-readelf --notes on the gfx950 bundles of the shared library).  Runs without a GPU.

    python tools/kernel_resources.py            # kernels with spills or scratch
    python tools/kernel_resources.py --all      # every kernel

`tests/test_abi.py::test_no_kernel_spills_or_uses_scratch` uses `kernel_resources()` as the
zero-scratch gate of the shipped library."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "diffpointrasterisation.jl_amd", "libdpr.so")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return out.splitlines()
    except Exception:
        return list(names)


def kernel_resources(lib=LIB):
    """[{name, vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds}] for every kernel of `lib`."""
    tmp = tempfile.mkdtemp(prefix="dpr_co_")
    try:
        copy = os.path.join(tmp, "lib.so")
        shutil.copy(lib, copy)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", copy], check=True,
                       capture_output=True, cwd=tmp)
        kernels = []
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)],
                                   check=True, capture_output=True, text=True).stdout
            # one YAML map per kernel under amdhsa.kernels; fields are flat "  .key: value" lines
            for block in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
                block = ".agpr_count:" + block
                get = lambda k, b=block: re.search(rf"\.{k}:\s*(\S+)", b)
                name = get("name")
                if not name:
                    continue
                num = lambda k: int(get(k).group(1)) if get(k) else 0
                kernels.append(dict(name=name.group(1), vgpr=num("vgpr_count"), agpr=num("agpr_count"),
                                    sgpr=num("sgpr_count"), vgpr_spill=num("vgpr_spill_count"),
                                    sgpr_spill=num("sgpr_spill_count"),
                                    scratch=num("private_segment_fixed_size"),
                                    lds=num("group_segment_fixed_size")))
        for k, d in zip(kernels, demangle([k["name"] for k in kernels])):
            k["demangled"] = d
        return kernels
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def short(d):
    """`dpr::k_name<template args>` without the parameter list"""
    m = re.match(r"(?:void )?(dpr::\w+(?:<.*?>)?)\(", d)
    return m.group(1) if m else d


if __name__ == "__main__":
    ks = kernel_resources()
    show_all = "--all" in sys.argv
    n = 0
    for k in sorted(ks, key=lambda k: (-k["scratch"], k["demangled"])):
        if show_all or k["vgpr_spill"] or k["scratch"] or k["sgpr_spill"]:
            n += 1
            print(f"vgpr {k['vgpr']:3d} agpr {k['agpr']:3d} spill v{k['vgpr_spill']:3d} s{k['sgpr_spill']:3d} "
                  f"scratch {k['scratch']:4d} B  lds {k['lds']:6d}  {short(k['demangled'])}")
    print(f"{len(ks)} kernels, {n} listed")
