"""What does giving 7.2 MB back to the kernel cost a process — before and after it has opened the GPU?
(bench.py frees every step's cloud inside the timed loop; this separates the operating system's share.)"""
import mmap
import sys
import time

import numpy as np


def once(nbytes):
    m = mmap.mmap(-1, nbytes, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    a = np.frombuffer(m, dtype=np.uint8)
    t0 = time.perf_counter()
    a[:] = 1
    t1 = time.perf_counter()
    del a
    t2 = time.perf_counter()
    m.close()
    return (t1 - t0) * 1e3, (time.perf_counter() - t2) * 1e3


def report(tag):
    for nbytes in (2_400_000, 7_200_000):
        r = np.array([once(nbytes) for _ in range(50)])
        print(f"{tag}: {nbytes / 1e6:.1f} MB  first touch {np.median(r[:, 0]):.3f} ms (max {r[:, 0].max():.3f}), "
              f"munmap {np.median(r[:, 1]):.3f} ms (max {r[:, 1].max():.3f})", flush=True)
    # malloc / free of the same size (numpy): what glibc does with it
    t = []
    for _ in range(50):
        a = np.empty(7_200_000, dtype=np.uint8)
        a[:] = 1
        t0 = time.perf_counter()
        del a
        t.append((time.perf_counter() - t0) * 1e3)
    print(f"{tag}: free() of a 7.2 MB numpy array: median {np.median(t):.3f} ms, max {max(t):.3f}", flush=True)


report("before the GPU is opened")
import torch  # noqa: E402
torch.zeros(1, device="cuda")
torch.cuda.synchronize()
report("after torch opened the GPU")
sys.path.insert(0, ".")
from eskf_lio_amd import capi  # noqa: E402
with capi.Context(0) as ctx:
    ctx.map_reset(0.3, 1000)
    report("with a vgicp context alive")

    # the bench's situation: 500 clouds (2.4 + 7.2 MB each) allocated up front, written once, unmapped one by one later
    def many(aligned):
        two_mb = 2 << 20
        maps = []
        for _ in range(500):
            pair = []
            for nbytes in (2_400_000, 7_200_000):
                size = ((nbytes + two_mb - 1) // two_mb * two_mb + two_mb) if aligned else nbytes
                m = mmap.mmap(-1, size, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
                a = np.frombuffer(m, dtype=np.uint8)
                if aligned:
                    off = (-a.ctypes.data) % two_mb
                    a = a[off:off + nbytes]
                a[:] = 1
                del a
                pair.append(m)
            maps.append(pair)
        t = []
        for pair in maps:
            t0 = time.perf_counter()
            for m in pair:
                m.close()
            t.append((time.perf_counter() - t0) * 1e3)
        t = np.array(t)
        print(f"500 clouds written once, then unmapped one by one ({'2 MB-aligned' if aligned else 'as mmap places them'}): "
              f"median {np.median(t):.3f} ms per cloud, p99 {np.percentile(t, 99):.3f}, max {t.max():.3f}", flush=True)
    many(False)
    many(True)
