"""Developer probe: the in-process multi-device context (vgicp_create_multi) with N sub-contexts on device 0.
Prints parity against a single-device context and timings.  usage: python tools/probe_multi.py [N ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eskf_lio_amd import capi, synth  # noqa: E402


def main():
    worlds = [int(a) for a in sys.argv[1:]] or [2, 4]
    big = os.environ.get("PROBE_C2") == "1"
    vmap = synth.make_map(1_000_000 if big else 50_000)
    pts, covs = synth.make_uniform_scan(100_000 if big else 5_000, vmap)
    g = synth.default_guess()
    with capi.Context(0) as one:
        one.map_reset(vmap.voxel_size, vmap.keys.shape[0])
        one.map_upsert(vmap.keys, vmap.means, vmap.covs)
        ref = one.align(pts, covs, g, 20, 1e-6, 2.0)
        t0 = time.perf_counter()
        for _ in range(20):
            one.align(pts, covs, g, 20, 1e-6, 2.0)
        t_one = (time.perf_counter() - t0) / 20
        one.scan_upload(pts, covs)
        t0 = time.perf_counter()
        for _ in range(20):
            rr = one.align_resident(g, 20, 1e-6, 2.0)
        t_one_res = (time.perf_counter() - t0) / 20
        print(f"single: align {t_one*1e3:.3f} ms, resident {t_one_res*1e3:.3f} ms, device {rr.device_seconds*1e6:.1f} us")
    for n in worlds:
        t0 = time.perf_counter()
        with capi.Context([0] * n) as ctx:
            print(f"N={n}: created in {time.perf_counter()-t0:.2f} s; device_info {ctx.device_info()}")
            ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0])
            ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
            print("  map", ctx.map_size())
            got = ctx.align(pts, covs, g, 20, 1e-6, 2.0)
            print(f"  align: iterations {got.iterations} world {got.world_size} launches {got.launches} counts equal "
                  f"{np.array_equal(got.corr_count, ref.corr_count)} pose diff {np.abs(got.pose-ref.pose).max():.2e} "
                  f"fallbacks {ctx.counter(1)} launches {ctx.counter(0)}")
            t0 = time.perf_counter()
            for _ in range(20):
                got = ctx.align(pts, covs, g, 20, 1e-6, 2.0)
            t_n = (time.perf_counter() - t0) / 20
            ctx.scan_upload(pts, covs)
            t0 = time.perf_counter()
            for _ in range(20):
                rr = ctx.align_resident(g, 20, 1e-6, 2.0)
            t_res = (time.perf_counter() - t0) / 20
            print(f"  timing: align {t_n*1e3:.3f} ms, resident {t_res*1e3:.3f} ms, device {rr.device_seconds*1e6:.1f} us, fallbacks {ctx.counter(1)}")
            h = ctx.align(pts, covs, g, 20, 1e-6, 2.0, flags=capi.FLAG_NO_PERSISTENT)
            print(f"  host-summed loop: launches {h.launches} counts equal {np.array_equal(h.corr_count, ref.corr_count)} "
                  f"pose diff {np.abs(h.pose-ref.pose).max():.2e} seconds {h.seconds*1e3:.2f} ms")
            for m in (0, 1, 3, 449):
                a = ctx.align(pts[:m], covs[:m], g, 5, 1e-6, 2.0, allow_degenerate=True)
                with capi.Context(0) as one:
                    one.map_reset(vmap.voxel_size, vmap.keys.shape[0])
                    one.map_upsert(vmap.keys, vmap.means, vmap.covs)
                    b = one.align(pts[:m], covs[:m], g, 5, 1e-6, 2.0, allow_degenerate=True)
                print(f"  n={m}: status {a.status}/{b.status} counts {a.corr_count.tolist()} vs {b.corr_count.tolist()} "
                      f"pose diff {np.nanmax(np.abs(a.pose-b.pose)):.2e}")
            # resident insertion from shards
            ctx.scan_upload(pts, covs)
            fresh = ctx.map_insert_resident(np.eye(4), 5)
            with capi.Context(0) as one:
                one.map_reset(vmap.voxel_size, vmap.keys.shape[0])
                one.map_upsert(vmap.keys, vmap.means, vmap.covs)
                one.scan_upload(pts, covs)
                fresh1 = one.map_insert_resident(np.eye(4), 5)
                print(f"  insert_resident: new voxels {fresh} vs {fresh1}; map {ctx.map_size()} vs {one.map_size()}")


if __name__ == "__main__":
    main()
