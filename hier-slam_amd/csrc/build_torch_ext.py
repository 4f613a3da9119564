#!/usr/bin/env python3
"""Builds diff_gaussian_rasterization/_hsr_torch.so (hsr_torch_ext.cpp: the torch glue above the C ABI) in-tree with one g++
command — host C++ only, no device code, no ninja, no setup.py.  Needs ../libhsr_rast.so (make -C . first)."""
import os
import subprocess
import sys
import sysconfig

import torch
from torch.utils import cpp_extension

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(HERE), "diff_gaussian_rasterization")
OUT = os.path.join(PKG, "_hsr_torch.so")
SRC = os.path.join(HERE, "hsr_torch_ext.cpp")


def main(force=False):
    deps = [SRC, os.path.join(os.path.dirname(HERE), "..", "include", "hsr_rasterizer.h"), os.path.abspath(__file__)]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    inc = [p for p in cpp_extension.include_paths() if os.path.isdir(p)] + [sysconfig.get_paths()["include"]]
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", SRC, "-o", OUT,
           "-DTORCH_EXTENSION_NAME=_hsr_torch", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    cmd += ["-I" + p for p in inc]
    cmd += ["-L" + libdir, "-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10",
            "-L" + os.path.dirname(HERE), "-lhsr_rast",
            "-Wl,-rpath," + libdir, "-Wl,-rpath,$ORIGIN/.."]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(main(force="--force" in sys.argv))
