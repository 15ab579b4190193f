#!/usr/bin/env python3
"""What giving device memory back costs on this box, and which ways around it exist (psk_build_presence spent
0.45 s of a 2.3-s run in hipFree of the inflate's ~65 GB of buffers).  Measurement only: HIP through ctypes, no libpsk.

  python tools/free_probe.py            the table (JSON on stdout)
  python tools/free_probe.py child GB   (internal) allocates + touches GB gigabytes, prints a time stamp, exits WITHOUT hipFree

Questions: (1) hipFree of touched memory by size; (2) the stream-ordered allocator (hipMallocAsync / hipFreeAsync with a release
threshold that keeps freed blocks in the pool): free, and the next allocation out of the pool; (3) hipFree on a helper thread while
the calling thread keeps a stream busy -- does the stream stall?; (4) a process that exits without freeing: how long after its last
line does wait() return?"""
import ctypes
import json
import subprocess
import sys
import threading
import time

hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
GB = 1 << 30


def chk(rc, what):
    if rc != 0:
        raise RuntimeError("%s: hip error %d" % (what, rc))


def malloc(nbytes):
    p = ctypes.c_void_p()
    chk(hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes)), "hipMalloc")
    return p


def touch(p, nbytes, stream=None):
    chk(hip.hipMemsetAsync(p, 1, ctypes.c_size_t(nbytes), stream), "hipMemsetAsync")


def sync():
    chk(hip.hipDeviceSynchronize(), "hipDeviceSynchronize")


def child(gb):
    chk(hip.hipSetDevice(0), "hipSetDevice")
    bufs = [malloc(GB) for _ in range(int(gb))]
    for p in bufs:
        touch(p, GB)
    sync()
    sys.stdout.write("%.6f\n" % time.time())
    sys.stdout.flush()
    # no hipFree: the driver takes the memory back when the process is gone


def alloc_child(sizes_gb):
    """A fresh process: hipMalloc of each size in turn (kept), each touched; prints the milliseconds of every hipMalloc."""
    chk(hip.hipSetDevice(0), "hipSetDevice")
    hip.hipFree(None)
    ms = []
    for gb in sizes_gb:
        t0 = time.time()
        p = malloc(int(gb * GB))
        ms.append(round((time.time() - t0) * 1e3, 2))
        touch(p, int(gb * GB))
        sync()
    print(json.dumps(ms))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "allocs":
        # r06: is it the SIZE of one hipMalloc that is slow (24 GB: 750 ms, <= 12 GB: 0.2 ms in the first table), or the total?
        res = {}
        for name, sizes in (("1x24", [24]), ("2x12", [12, 12]), ("3x8", [8, 8, 8]), ("6x4", [4] * 6), ("1x16", [16]), ("1x13", [13]), ("1x20", [20]),
                            ("12 then 24", [12, 24]), ("8x12", [12] * 8), ("2x48", [48, 48])):
            p = subprocess.run([sys.executable, __file__, "alloc_child"] + [str(x) for x in sizes], capture_output=True, text=True)
            res[name] = json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 else "failed: " + p.stderr[-200:]
            time.sleep(3)     # (the driver clears what the child held)
        print(json.dumps({"hipMalloc_ms_fresh_process": res}, indent=1))
        return
    chk(hip.hipSetDevice(0), "hipSetDevice")
    hip.hipFree(None)
    out = {"hipFree_ms_by_GB": {}, "hipMalloc_ms_by_GB": {}}
    for gb in (1, 4, 12, 24, 48):
        t0 = time.time()
        p = malloc(gb * GB)
        t1 = time.time()
        touch(p, gb * GB)
        sync()
        t2 = time.time()
        chk(hip.hipFree(p), "hipFree")
        t3 = time.time()
        out["hipMalloc_ms_by_GB"][gb] = round((t1 - t0) * 1e3, 2)
        out["hipFree_ms_by_GB"][gb] = round((t3 - t2) * 1e3, 2)
    # many buffers freed one after the other (the inflate holds seven)
    bufs = [malloc(8 * GB) for _ in range(6)]
    for p in bufs:
        touch(p, 8 * GB)
    sync()
    t0 = time.time()
    for p in bufs:
        chk(hip.hipFree(p), "hipFree")
    out["hipFree_ms_6x8GB"] = round((time.time() - t0) * 1e3, 2)

    # (2) the stream-ordered allocator
    try:
        stream = ctypes.c_void_p()
        chk(hip.hipStreamCreate(ctypes.byref(stream)), "hipStreamCreate")
        pool = ctypes.c_void_p()
        chk(hip.hipDeviceGetDefaultMemPool(ctypes.byref(pool), 0), "hipDeviceGetDefaultMemPool")
        thr = ctypes.c_uint64(0xFFFFFFFFFFFFFFFF)
        chk(hip.hipMemPoolSetAttribute(pool, 4, ctypes.byref(thr)), "hipMemPoolSetAttribute(release threshold)")   # hipMemPoolAttrReleaseThreshold = 4
        res = {}
        for gb in (12, 24):
            p = ctypes.c_void_p()
            t0 = time.time()
            chk(hip.hipMallocAsync(ctypes.byref(p), ctypes.c_size_t(gb * GB), stream), "hipMallocAsync")
            chk(hip.hipStreamSynchronize(stream), "sync")
            t1 = time.time()
            touch(p, gb * GB, stream)
            chk(hip.hipStreamSynchronize(stream), "sync")
            t2 = time.time()
            chk(hip.hipFreeAsync(p, stream), "hipFreeAsync")
            chk(hip.hipStreamSynchronize(stream), "sync")
            t3 = time.time()
            q = ctypes.c_void_p()
            chk(hip.hipMallocAsync(ctypes.byref(q), ctypes.c_size_t(gb * GB // 2), stream), "hipMallocAsync")
            chk(hip.hipStreamSynchronize(stream), "sync")
            t4 = time.time()
            chk(hip.hipFreeAsync(q, stream), "hipFreeAsync")
            chk(hip.hipStreamSynchronize(stream), "sync")
            res[gb] = {"mallocAsync_ms": round((t1 - t0) * 1e3, 2), "freeAsync_ms": round((t3 - t2) * 1e3, 2),
                       "mallocAsync_half_from_pool_ms": round((t4 - t3) * 1e3, 2)}
        t0 = time.time()
        chk(hip.hipMemPoolTrimTo(pool, ctypes.c_size_t(0)), "hipMemPoolTrimTo")
        res["trim_pool_ms"] = round((time.time() - t0) * 1e3, 2)
        out["stream_ordered"] = res
    except Exception as e:   # noqa: BLE001
        out["stream_ordered"] = "failed: %s" % e

    # (3) hipFree on a helper thread while this thread keeps a stream busy with 64-MB memsets
    try:
        big = malloc(32 * GB)
        touch(big, 32 * GB)
        small = malloc(GB)
        sync()
        lat = []
        t_free = [0.0, 0.0]

        def helper():
            t_free[0] = time.time()
            hip.hipFree(big)
            t_free[1] = time.time()
        th = threading.Thread(target=helper)
        t_begin = time.time()
        th.start()
        while th.is_alive() or len(lat) < 20:
            t0 = time.time()
            touch(small, 64 << 20, stream)
            chk(hip.hipStreamSynchronize(stream), "sync")
            lat.append((time.time() - t0) * 1e3)
            if time.time() - t_begin > 20:
                break
        th.join()
        quiet = []
        for _ in range(50):
            t0 = time.time()
            touch(small, 64 << 20, stream)
            chk(hip.hipStreamSynchronize(stream), "sync")
            quiet.append((time.time() - t0) * 1e3)
        out["free_on_helper_thread"] = {"hipFree_32GB_ms": round((t_free[1] - t_free[0]) * 1e3, 2), "launches_meanwhile": len(lat),
                                        "memset64MB_ms_max_meanwhile": round(max(lat), 3), "memset64MB_ms_median_meanwhile": round(sorted(lat)[len(lat) // 2], 3),
                                        "memset64MB_ms_median_quiet": round(sorted(quiet)[len(quiet) // 2], 3)}
        hip.hipFree(small)
    except Exception as e:   # noqa: BLE001
        out["free_on_helper_thread"] = "failed: %s" % e

    # (4) exit without freeing
    res = {}
    for gb in (1, 48):
        p = subprocess.Popen([sys.executable, __file__, "child", str(gb)], stdout=subprocess.PIPE, text=True)
        stamp = float(p.stdout.readline())
        p.wait()
        res[gb] = round((time.time() - stamp) * 1e3, 1)
    out["exit_without_free_ms_after_last_line_by_GB"] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(float(sys.argv[2]))
    elif len(sys.argv) > 2 and sys.argv[1] == "alloc_child":
        alloc_child([float(x) for x in sys.argv[2:]])
    else:
        main()
