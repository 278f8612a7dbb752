"""Builds the HIP library (and the host-side C++ classes) in-tree for gfx950.

    python -m blah2_amd.build            # build everything that is stale
    python -m blah2_amd.build --force

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting ``blah2_amd/libblah2hip.so`` travels to the GPU box with the tree.
"""
from __future__ import annotations

import contextlib
import fcntl
import os
import shlex
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


@contextlib.contextmanager
def _build_lock():
    """One builder at a time per tree: ranks of a job and pytest workers that import against a stale library would otherwise
    compile into the same object files and link each other's half-written ones."""
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    with open(os.path.join(objdir, ".lock"), "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


def extra_flags():
    """Extra hipcc flags for the product build from BLAH2HIP_EXTRA_HIPCC_FLAGS (shell-quoted).  They are part of the
    staleness check (a library built with other flags is rebuilt).  Experiment builds that must not replace the product
    library go through tools/build_variant.sh instead."""
    return shlex.split(os.environ.get("BLAH2HIP_EXTRA_HIPCC_FLAGS", ""))


def build_hip(force=False, verbose=True):
    with _build_lock():
        return _build_hip_locked(force, verbose)


def _build_hip_locked(force, verbose):
    srcs = _sources(CSRC, (".hip",))
    deps = srcs + _sources(CSRC, (".hpp",)) + [os.path.join(ROOT, "include", "blah2hip.h")]
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-fno-slp-vectorize",
             *extra_flags(), "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    objdir = os.path.join(PKG, "build")
    stamp = os.path.join(objdir, "flags.txt")
    flag_text = " ".join(flags[:-4])
    same_flags = os.path.exists(stamp) and open(stamp).read() == flag_text
    # (a library that came with the tree and has no stamp beside it was built by this script with the default flags)
    if not force and not _newer(LIB, deps) and (same_flags or (not os.path.exists(stamp) and not extra_flags())):
        return LIB
    # one hipcc per translation unit, side by side (capi.hip alone is 40 s of the build), then one link
    procs = []
    try:
        for src in srcs:
            obj = os.path.join(objdir, os.path.relpath(src, CSRC).replace(os.sep, "__") + ".o")  # same-named files in subdirectories do not collide
            cmd = [hipcc(), *flags, "-c", src, "-o", obj]
            if verbose:
                print("[blah2_amd.build]", " ".join(cmd), flush=True)
            procs.append((cmd, obj, subprocess.Popen(cmd)))
        objs = []
        for cmd, obj, p_ in procs:
            if p_.wait() != 0:
                raise subprocess.CalledProcessError(p_.returncode, cmd)
            objs.append(obj)
    except BaseException:
        for _, _, p_ in procs:  # a failed unit (or an interrupt): the siblings do not keep compiling behind our back
            if p_.poll() is None:
                p_.kill()
            p_.wait()
        raise
    tmp = LIB + f".{os.getpid()}.tmp"
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-fPIC", "-shared", *objs, "-o", tmp]
    if verbose:
        print("[blah2_amd.build]", " ".join(cmd), flush=True)
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)  # a process that has the old library mapped keeps it; nobody sees a half-written file
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    with open(stamp, "w") as f:
        f.write(flag_text)
    return LIB


def build_host(force=False, verbose=True):
    """C++ classes with the reference's own surface (Ambiguity, Map, IqData, ...)."""
    with _build_lock():
        return _build_host_locked(force, verbose)


def _build_host_locked(force, verbose):
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
