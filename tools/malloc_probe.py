#!/usr/bin/env python3
"""hipMalloc cost by allocation size in a fresh process (the presence matrix, the result arrays and the arena of the
per-sample lists are such allocations), and the first-touch cost (hipMemset) right after.
usage: tools/malloc_probe.py"""
import ctypes
import time

hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipSetDevice(0)
p0 = ctypes.c_void_p()
hip.hipMalloc(ctypes.byref(p0), 256)
hip.hipDeviceSynchronize()


def one(size):
    p = ctypes.c_void_p()
    t = time.perf_counter()
    rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(size))
    t1 = time.perf_counter()
    hip.hipMemset(p, 0, ctypes.c_size_t(size))
    hip.hipDeviceSynchronize()
    t2 = time.perf_counter()
    hip.hipMemset(p, 0, ctypes.c_size_t(size))
    hip.hipDeviceSynchronize()
    t3 = time.perf_counter()
    return rc, p, (t1 - t) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3


for rep in range(2):
    held = []
    for mib in (64, 256, 1024, 1025, 1536, 2048, 4096, 8192, 16384):
        rc, p, a, m1, m2 = one(mib << 20)
        print("pass %d  %6d MiB: hipMalloc %8.2f ms  first memset %8.2f ms  second memset %8.2f ms  rc=%d"
              % (rep, mib, a, m1, m2, rc), flush=True)
        held.append(p)
    for p in held:
        t = time.perf_counter()
        hip.hipFree(p)
    print("freed")

# 1 GiB chunks until 240 GiB are held: where does the cost per chunk change?
held, ts = [], []
for i in range(240):
    p = ctypes.c_void_p()
    t = time.perf_counter()
    rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(1 << 30))
    ts.append((time.perf_counter() - t) * 1e3)
    if rc != 0:
        print("hipMalloc failed at chunk", i, rc)
        break
    held.append(p)
print("ms per 1 GiB chunk while holding 0..%d GiB:" % len(held), " ".join("%.0f" % t for t in ts))
