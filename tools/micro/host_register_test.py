#!/usr/bin/env python3
"""How fast is hipHostRegister on memory-mapped (page-cached) .beta files, and how fast does the DMA run from registered pages,
against the library's staged upload?  (Decides whether the upload path should register the caller's rows instead of copying them
through page-locked staging pieces.)
    python tools/micro/host_register_test.py [--files 32] [--mb 56]
"""
import argparse
import ctypes as C
import mmap
import os
import tempfile
import time

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--files', type=int, default=32)
    ap.add_argument('--mb', type=int, default=56)
    a = ap.parse_args()
    hip = C.CDLL('libamdhip64.so')
    hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
    hip.hipHostUnregister.argtypes = [C.c_void_p]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    n = a.mb << 20
    d = tempfile.mkdtemp(dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
    rng = np.random.default_rng(0)
    paths = []
    for i in range(a.files):
        p = os.path.join(d, 'f%d.beta' % i)
        rng.integers(0, 255, n, dtype=np.uint8).tofile(p)
        paths.append(p)
    dev = torch.empty((a.files, n), dtype=torch.uint8, device='cuda:0')
    torch.cuda.synchronize()
    for flags, name in ((0, 'default'), (2, 'hipHostRegisterMapped'), (8, 'hipHostRegisterReadOnly?')):
        maps, addrs = [], []
        fds = [os.open(p, os.O_RDONLY) for p in paths]
        for fd in fds:
            m = mmap.mmap(fd, n, prot=mmap.PROT_READ)
            maps.append(m)
            addrs.append(np.frombuffer(m, dtype=np.uint8).ctypes.data)
        t0 = time.perf_counter()
        rcs = [hip.hipHostRegister(C.c_void_p(ad), n, flags) for ad in addrs]
        t1 = time.perf_counter()
        if any(rcs):
            print('%s: hipHostRegister failed: %s' % (name, rcs[:4]))
        else:
            for i, ad in enumerate(addrs):
                hip.hipMemcpyAsync(C.c_void_p(dev[i].data_ptr()), C.c_void_p(ad), n, 1, None)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for ad in addrs:
                hip.hipHostUnregister(C.c_void_p(ad))
            t3 = time.perf_counter()
            print('%s: register %d x %d MB: %.1f ms; DMA from the registered pages: %.1f ms = %.1f GB/s; unregister %.1f ms; total %.1f ms' % (
                name, a.files, a.mb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, a.files * n / (t2 - t1) / 1e9, (t3 - t2) * 1e3, (t3 - t0) * 1e3))
        del addrs
        for m in maps:
            try:
                m.close()
            except BufferError:
                pass
        for fd in fds:
            os.close(fd)
    # plain pageable copy for comparison
    fds = [os.open(p, os.O_RDONLY) for p in paths]
    maps = [mmap.mmap(fd, n, prot=mmap.PROT_READ) for fd in fds]
    arrs = [np.frombuffer(m, dtype=np.uint8) for m in maps]
    t0 = time.perf_counter()
    for i, ar in enumerate(arrs):
        hip.hipMemcpy(C.c_void_p(dev[i].data_ptr()), C.c_void_p(ar.ctypes.data), n, 1)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print('pageable hipMemcpy, one thread: %.1f ms = %.1f GB/s' % ((t1 - t0) * 1e3, a.files * n / (t1 - t0) / 1e9))
    for p in paths:
        os.unlink(p)
    os.rmdir(d)


if __name__ == '__main__':
    main()
