#!/usr/bin/env python3
"""Can the replay upload straight out of the page cache?  mmap a capture-sized file of /dev/shm, hipHostRegister windows of
it (the three ways a read-only capture can be mapped), and time registration, the H2D copy from the registered window,
and unregistration against pread into a pinned buffer (GPU box).

    python tools/gpu_hostreg.py [MiB per window] [windows]
"""
import ctypes as C
import mmap
import os
import sys
import time

import numpy as np
import torch

hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipGetErrorString.restype = C.c_char_p
H2D = 1


def main():
    if "--numa" in sys.argv:
        sys.argv.remove("--numa")
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from blah2_amd import replay as R
        print("pinned to the GPU's NUMA node:", R.pin_to_device_node(torch, 0), "cpus", len(os.sched_getaffinity(0)), flush=True)
    win = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
    nwin = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    path = "/dev/shm/blah2_hostreg.bin"
    rng = np.random.default_rng(1)
    with open(path, "wb") as f:
        for _ in range(nwin):
            f.write(rng.integers(-2000, 2000, win // 2, dtype=np.int16).tobytes())
    dev = torch.empty(win, dtype=torch.uint8, device="cuda")
    pinned = torch.empty(win, dtype=torch.uint8).pin_memory()
    st = torch.cuda.current_stream().cuda_stream
    ref = np.fromfile(path, dtype=np.uint8)

    # baseline: pread into a pinned buffer (1 thread), then H2D
    fd = os.open(path, os.O_RDONLY)
    for k in range(nwin):
        t0 = time.perf_counter()
        os.preadv(fd, [memoryview(pinned.numpy())], k * win)
        t1 = time.perf_counter()
        dev.copy_(pinned, non_blocking=True)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"pread -> pinned window {k}: {win / (t1 - t0) / 1e9:6.1f} GB/s, H2D {win / (t2 - t1) / 1e9:6.1f} GB/s", flush=True)
    os.close(fd)

    variants = [
        ("O_RDWR   MAP_SHARED  rw, flags 0", os.O_RDWR, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE, 0),
        ("O_RDONLY MAP_SHARED  ro, ReadOnly", os.O_RDONLY, mmap.MAP_SHARED, mmap.PROT_READ, 0x08),
        ("O_RDONLY MAP_SHARED  ro, flags 0", os.O_RDONLY, mmap.MAP_SHARED, mmap.PROT_READ, 0),
        ("O_RDONLY MAP_PRIVATE rw, flags 0", os.O_RDONLY, mmap.MAP_PRIVATE, mmap.PROT_READ | mmap.PROT_WRITE, 0),
    ]
    for name, oflag, mflag, prot, rflag in variants:
        fd = os.open(path, oflag)
        mm = mmap.mmap(fd, nwin * win, flags=mflag, prot=prot)
        arr = np.frombuffer(mm, dtype=np.uint8)
        base = arr.ctypes.data
        print(name, flush=True)
        for k in range(nwin):
            p = base + k * win
            t0 = time.perf_counter()
            rc = hip.hipHostRegister(p, win, rflag)
            t1 = time.perf_counter()
            if rc:
                print(f"  hipHostRegister failed: {rc} {hip.hipGetErrorString(rc).decode()}", flush=True)
                break
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            rc = hip.hipMemcpyAsync(dev.data_ptr(), p, win, H2D, st)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            ok = rc == 0 and bool((dev.cpu().numpy() == ref[k * win:(k + 1) * win]).all())
            t4 = time.perf_counter()
            rc2 = hip.hipHostUnregister(p)
            t5 = time.perf_counter()
            print(f"  window {k}: register {win / (t1 - t0) / 1e9:6.1f} GB/s ({(t1 - t0) * 1e3:.1f} ms), H2D {win / (t3 - t2) / 1e9:6.1f} GB/s rc {rc} "
                  f"same bytes {ok}, unregister {(t5 - t4) * 1e3:.1f} ms rc {rc2}", flush=True)
        del arr
        try:
            mm.close()
        except BufferError:
            pass
        os.close(fd)
    # page-table population ahead of the registration: madvise(MADV_POPULATE_READ) (Linux 5.14+) from 1 / 2 / 4 threads on a
    # fresh mapping, then the registration and copy of the populated window
    from concurrent.futures import ThreadPoolExecutor
    MADV_POPULATE_READ = 22
    libc = C.CDLL(None, use_errno=True)
    libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    for threads in (1, 2, 4):
        fd = os.open(path, os.O_RDONLY)
        mm = mmap.mmap(fd, nwin * win, flags=mmap.MAP_SHARED, prot=mmap.PROT_READ)
        arr = np.frombuffer(mm, dtype=np.uint8)
        base = arr.ctypes.data
        pool = ThreadPoolExecutor(threads)
        print(f"madvise(MADV_POPULATE_READ) with {threads} thread(s), then register", flush=True)
        for k in range(nwin):
            p = base + k * win
            part = win // threads
            t0 = time.perf_counter()
            rcs = list(pool.map(lambda i: libc.madvise(p + i * part, part, MADV_POPULATE_READ), range(threads)))
            t1 = time.perf_counter()
            rcr = list(pool.map(lambda i: hip.hipHostRegister(p + i * part, part, 0), range(threads)))
            t2 = time.perf_counter()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            for i in range(threads):
                hip.hipMemcpyAsync(dev.data_ptr() + i * part, p + i * part, part, H2D, st)
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            ok = bool((dev.cpu().numpy() == ref[k * win:(k + 1) * win]).all())
            for i in range(threads):
                hip.hipHostUnregister(p + i * part)
            print(f"  window {k}: populate {win / (t1 - t0) / 1e9:6.1f} GB/s rc {max(rcs)} errno {C.get_errno()}, register {win / (t2 - t1) / 1e9:6.1f} GB/s rc {max(rcr)}, "
                  f"H2D {win / (t4 - t3) / 1e9:6.1f} GB/s same bytes {ok}", flush=True)
        pool.shutdown()
        del arr
        try:
            mm.close()
        except BufferError:
            pass
        os.close(fd)
    # user-space copy out of the (populated) mapping into a pinned buffer: memmove (glibc: non-temporal stores for large sizes)
    # against pread (the kernel's copy_to_user), 1 / 2 / 4 / 8 threads
    for threads in (1, 2, 4, 8):
        fd = os.open(path, os.O_RDONLY)
        mm = mmap.mmap(fd, nwin * win, flags=mmap.MAP_SHARED, prot=mmap.PROT_READ)
        arr = np.frombuffer(mm, dtype=np.uint8)
        base = arr.ctypes.data
        pool = ThreadPoolExecutor(threads)
        dstp = pinned.data_ptr()
        part = win // threads

        def cp(i, p):
            libc.madvise(p + i * part, part, MADV_POPULATE_READ)
            C.memmove(dstp + i * part, p + i * part, part)

        def rd(i, k):
            os.preadv(fd, [memoryview(pinned.numpy())[i * part:(i + 1) * part]], k * win + i * part)

        rates = []
        for k in range(nwin):
            p = base + k * win
            t0 = time.perf_counter()
            list(pool.map(lambda i: cp(i, p), range(threads)))
            t1 = time.perf_counter()
            ok = bool((pinned.numpy() == ref[k * win:(k + 1) * win]).all())
            t2 = time.perf_counter()
            list(pool.map(lambda i: rd(i, k), range(threads)))
            t3 = time.perf_counter()
            rates.append((win / (t1 - t0) / 1e9, win / (t3 - t2) / 1e9, ok))
        print(f"{threads} thread(s): populate + memmove " + ", ".join(f"{r[0]:.1f}" for r in rates) + " GB/s (same bytes " +
              str(all(r[2] for r in rates)) + "); pread " + ", ".join(f"{r[1]:.1f}" for r in rates) + " GB/s", flush=True)
        pool.shutdown()
        del arr
        try:
            mm.close()
        except BufferError:
            pass
        os.close(fd)
    os.unlink(path)


if __name__ == "__main__":
    main()
