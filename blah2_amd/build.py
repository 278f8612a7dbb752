"""Builds the HIP library (and the host-side C++ classes) in-tree for gfx950.

    python -m blah2_amd.build            # build everything that is stale
    python -m blah2_amd.build --force

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting ``blah2_amd/libblah2hip.so`` travels to the GPU box with the tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
LIB = os.path.join(PKG, "libblah2hip.so")
HOSTLIB = os.path.join(PKG, "libblah2host.so")
ARCH = "gfx950"


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _sources(d, exts):
    out = []
    for base, _, files in os.walk(d):
        for f in files:
            if f.endswith(exts):
                out.append(os.path.join(base, f))
    return sorted(out)


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return exe


def build_hip(force=False, verbose=True):
    srcs = _sources(CSRC, (".hip",))
    deps = srcs + _sources(CSRC, (".hpp",)) + [os.path.join(ROOT, "include", "blah2hip.h")]
    if not force and not _newer(LIB, deps):
        return LIB
    # one hipcc per translation unit, side by side (capi.hip alone is 40 s of the build), then one link
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-fno-slp-vectorize",
             "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [hipcc(), *flags, "-c", src, "-o", obj]
        if verbose:
            print("[blah2_amd.build]", " ".join(cmd), flush=True)
        procs.append((cmd, obj, subprocess.Popen(cmd)))
    objs = []
    for cmd, obj, p_ in procs:
        if p_.wait() != 0:
            raise subprocess.CalledProcessError(p_.returncode, cmd)
        objs.append(obj)
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-fPIC", "-shared", *objs, "-o", LIB]
    if verbose:
        print("[blah2_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


def build_host(force=False, verbose=True):
    """C++ classes with the reference's own surface (Ambiguity, Map, IqData, ...)."""
    srcs = _sources(HOST, (".cpp",))
    srcs = [s for s in srcs if not os.path.basename(s).startswith("test_")]
    if not srcs:
        return None
    deps = srcs + _sources(HOST, (".h",)) + [os.path.join(ROOT, "include", "blah2hip.h"), LIB]
    if not force and not _newer(HOSTLIB, deps):
        return HOSTLIB
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-pthread", "-I", os.path.join(ROOT, "include"),
           "-I", HOST, *srcs, "-o", HOSTLIB, "-L", PKG, "-lblah2hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print("[blah2_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return HOSTLIB


HOSTTEST = os.path.join(HOST, "test", "test_ambiguity")


def build_host_test(force=False, verbose=True):
    """C++ test programs for the drop-in classes: test_ambiguity (mirrors the reference's TestAmbiguity.cpp) and
    test_golden (the classes against the compiled reference's values, tests/test_host_cpp_gpu.py)."""
    if not os.path.exists(HOSTLIB):
        return None
    last = None
    for name in ("test_ambiguity", "test_golden"):
        src = os.path.join(HOST, "test", name + ".cpp")
        exe = os.path.join(HOST, "test", name)
        if not os.path.exists(src):
            continue
        last = exe
        if not force and not _newer(exe, [src, HOSTLIB, LIB]):
            continue
        cmd = ["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", HOST, src, "-o", exe,
               "-L", PKG, "-lblah2host", "-lblah2hip", "-Wl,-rpath,$ORIGIN/../.."]
        if verbose:
            print("[blah2_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return last


def build_all(force=False, verbose=True):
    out = [build_hip(force, verbose)]
    h = build_host(force, verbose)
    if h:
        out.append(h)
        t = build_host_test(force, verbose)
        if t:
            out.append(t)
    return out


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
