"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol the header
declares, and refuses to compute without a gfx950 device (no fallback). No GPU needed."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAS_GPU = torch.cuda.device_count() > 0  # device_count() does not initialise the GPU


def header_symbols():
    text = open(os.path.join(ROOT, "include", "vgicp_hip.h")).read()
    return sorted(set(re.findall(r"\b(vgicp_[a-z_0-9]+)\s*\(", text)))


def test_build_entry_point_compiles_everything():
    import __graft_entry__ as g
    g.build()
    for rel in ("eskf_lio_amd/lib/libvgicp_hip.so", "eskf_lio_amd/lib/libvgicp_host.so",
                "oracle/libvgicp_oracle.so"):
        assert os.path.exists(os.path.join(ROOT, rel)), rel


def test_library_exports_every_declared_symbol():
    from eskf_lio_amd import capi
    lib = capi.load_library()
    declared = header_symbols()
    assert len(declared) == 47 and set(declared) == set(capi.EXPORTS)
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True,
                         check=True).stdout
    exported = set(re.findall(r" T (vgicp_[a-z_0-9]+)", out))
    assert set(declared) <= exported
    assert lib.vgicp_abi_version() == 6


def test_library_is_a_gfx950_code_object_without_torch_or_oracle():
    from eskf_lio_amd import capi
    needed = subprocess.run(["readelf", "-d", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    libs = re.findall(r"NEEDED.*\[(.*)\]", needed)
    assert any("amdhip64" in n for n in libs)
    assert not any("torch" in n or "c10" in n or "oracle" in n or "rccl" in n for n in libs)  # RCCL is dlopen'ed
    raw = open(capi.LIB_PATH, "rb").read()
    # every device code object in the fat binary targets gfx950 (hipCUB's host-side arch-name table
    # mentions other gfx names as plain strings; code objects are what counts)
    targets = set(re.findall(rb"hipv4-amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", raw))
    assert targets == {b"gfx950"}
    assert b"nvptx" not in raw and b"sm_90" not in raw and b"sm_80" not in raw


def test_struct_layouts_match_the_header():
    from eskf_lio_amd import capi
    assert C.sizeof(capi.Params) == 32
    assert capi.Params.translation_sq_threshold.offset == 8 and capi.Params.flags.offset == 24
    assert C.sizeof(capi.Stats) == 56 and capi.Stats.corr_count.offset == 32


@pytest.mark.skipif(HAS_GPU, reason="checks the no-device behaviour")
def test_no_device_means_loud_failure_not_fallback():
    from eskf_lio_amd import capi
    with pytest.raises(capi.VgicpError) as e:
        capi.Context(0)
    assert e.value.code == capi.ERR_NO_DEVICE
    with pytest.raises(capi.VgicpError) as e:                       # the multi-device context: the same, no host-side stand-in
        capi.Context([0, 0])
    assert e.value.code == capi.ERR_NO_DEVICE
    with pytest.raises(capi.VgicpError) as e:
        capi.Context([0] * 17)                                      # at most 16 devices
    assert e.value.code == capi.ERR_BAD_ARGUMENT
    lib = capi.load_library()
    # NULL context: every entry point rejects it instead of crashing
    assert lib.vgicp_map_reset(None, 0.3, 0) == capi.ERR_BAD_ARGUMENT
    assert lib.vgicp_destroy(None) == capi.OK
    from eskf_lio_amd import host
    with pytest.raises(RuntimeError):
        host.LocalMap(0.3, 1)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "eskf_lio_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "vgicp_oracle" not in text, f
    for base, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            assert "oracle" not in open(os.path.join(base, f)).read().lower(), f


def test_shard_bounds_tile_the_scan():
    from eskf_lio_amd.distributed import shard_bounds
    for n in (0, 1, 7, 8, 9, 100_000, 1_000_003):
        for w in (1, 2, 3, 4, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def test_pose_abi_roundtrip():
    from eskf_lio_amd import capi, synth
    T = synth.se3_to_SE3([1, 2, 3, 0.1, 0.2, 0.3])
    v = capi.pose_to_abi(T)
    assert v[12] == T[0, 3] and v[1] == T[1, 0] and v[15] == 1.0      # column-major, as Eigen
    assert np.array_equal(capi.pose_from_abi(v), T)
    rows = np.arange(27.0)[None]
    JTJ, JTr = capi.expand_normal_eq(rows)
    assert JTJ[0, 3, 1] == 7 and JTJ[0, 1, 3] == 7 and JTJ[0, 5, 5] == 20 and JTr[0, 0] == 21


def test_bench_refuses_to_run_without_a_gpu_and_prices_bytes_as_designed():
    """bench.py has no CPU fallback; its byte model is SURVEY.md 8(d)'s 112 N + 96 M per round, and its
    traffic figure comes from the PMC summary taken at the same scan size."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"],
                          capture_output=True, text=True, cwd=root)
    assert proc.returncode != 0 and "no HIP device" in (proc.stderr + proc.stdout)
    assert proc.stdout.strip() == ""                                 # no JSON line without a measurement
    sys.path.insert(0, root)
    import bench
    assert bench.algorithmic_bytes(100_000, 49_912.8) == 112 * 100_000 + 96 * 49_912.8
    # the committed summaries: quoted only with their provenance, or not at all with the reason
    quoted = {}
    for n in (100_000, 1_000_000):
        t, src = bench.measured_traffic(n, 1, "persistent_kernel")
        if t is None:
            assert "reason" in src
        else:
            assert src["file"].startswith("profiles/") and src["tag"] and "match" in src
            quoted[n] = t
    if len(quoted) == 2:
        assert quoted[1_000_000] > 50 * quoted[100_000]              # per launch of 20 rounds
    t8, src8 = bench.measured_traffic(100_000, 8, "persistent_kernel")
    assert t8 is None and "reason" in src8


def test_traffic_is_quoted_only_for_the_build_it_was_measured_on(tmp_path, monkeypatch):
    """roofline.traffic comes from profiles/, not from the run: a summary taken on these kernel sources (or this very
    library) is quoted with file / tag / hashes, one taken on another build is refused with the reason."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from eskf_lio_amd import provenance
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    body = {"traffic": {"persistent_kernel_total_calibrated": 123.0}, "bench": {"config": {"points": 100_000}}}
    (prof / "a_old_summary.json").write_text(json.dumps(dict(body, tag="a_old")))                     # no hashes at all
    t, src = bench.measured_traffic(100_000, 1, "persistent_kernel")
    assert t is None and "another build" in src["reason"] and src["kernel_source_sha256"] == provenance.kernel_source_sha256()
    (prof / "b_src_summary.json").write_text(json.dumps(dict(body, tag="b_src", library_sha256="0" * 64,
                                                             kernel_source_sha256=provenance.kernel_source_sha256())))
    t, src = bench.measured_traffic(100_000, 1, "persistent_kernel")
    assert t == 123.0 and src["tag"] == "b_src" and "kernel sources" in src["match"] and src["file"] == "profiles/b_src_summary.json"
    (prof / "c_lib_summary.json").write_text(json.dumps(dict(body, tag="c_lib", library_sha256=provenance.library_sha256(),
                                                             kernel_source_sha256="1" * 64)))
    t, src = bench.measured_traffic(100_000, 1, "persistent_kernel")
    assert t == 123.0 and src["tag"] == "c_lib" and "byte for byte" in src["match"]
    (prof / "d_new_summary.json").write_text(json.dumps(dict(body, tag="d_new", library_sha256="2" * 64,
                                                             kernel_source_sha256="3" * 64)))
    t, src = bench.measured_traffic(100_000, 1, "persistent_kernel")
    assert t == 123.0 and src["tag"] == "c_lib"                      # the newest one that matches, never a stale one
    assert bench.measured_traffic(5_000, 1, "persistent_kernel")[0] is None


def test_module_fallback_solve_is_the_oracles_ldlt_bit_for_bit(oracle):
    """csrc/vgicp_math.h's ldlt6_solve — what the kernels run when the normal equations are not safely
    positive definite — compiled for the host, against the oracle's restatement of Eigen's pivoted LDLT
    (reference call site src/Registration.cpp:78) on well-conditioned, indefinite and rank-deficient
    systems with noise pivots: same operation order, no contraction, hence the same bits."""
    from eskf_lio_amd import host
    rng = np.random.default_rng(5)
    for trial in range(600):
        kind = trial % 4
        if kind == 0:
            Q, _ = np.linalg.qr(rng.normal(size=(6, 6)))
            A = (Q * np.geomspace(1.0, 10.0 ** rng.uniform(0, 10), 6)) @ Q.T
        elif kind == 1:
            Q, _ = np.linalg.qr(rng.normal(size=(6, 6)))
            A = (Q * rng.normal(size=6)) @ Q.T
        elif kind == 2:
            B = rng.normal(size=(6, 3))
            A = B @ B.T
        else:
            B = rng.integers(-3, 4, size=(6, int(rng.integers(1, 6)))).astype(np.float64)
            A = B @ B.T
        A = 0.5 * (A + A.T)
        b = rng.normal(size=6)
        ours = host.math_ldlt6_solve(A, -b)
        theirs, _ = oracle.solve_step(A, b)
        assert np.array_equal(ours, theirs, equal_nan=True), trial
    assert np.array_equal(host.math_ldlt6_solve(np.zeros((6, 6)), np.zeros(6)), np.zeros(6))   # K3
    for xi in ([0.1, -0.2, 0.3, 0.0, 0.0, 0.0], [0.1, -0.2, 0.3, 6e-8, 0.0, 8e-8], [0, 0, 0, 0, 0, np.pi / 2],
               [1.0, 2.0, 3.0, 0.3, -0.4, 1.2]):
        assert np.allclose(host.math_se3_exp(xi), oracle.se3_to_SE3(np.array(xi, dtype=np.float64)), rtol=0, atol=1e-15)


@pytest.mark.skipif(not os.path.isdir("/root/reference/include/ESKF_LIO"),
                    reason="build-container check: needs the reference's own Types.hpp on the include path")
def test_shim_native_types_branch_meets_a_compiler(tmp_path):
    """include/eskf_lio_shim/*.hpp have two branches: dependency-free stand-in types (compiled and tested everywhere
    in this repository) and the reference's REAL types — Eigen, Open3D, yaml-cpp — which this image does not have.
    tests/compile_native/ gives that second branch compiler contact: minimal stand-in headers that declare only the
    members the branch and the reference's call sites use (src/ErrorStateKF.cpp:126-130, src/Odometry.cpp:11-16,61,74,
    86, reproduced in call_sites.cpp), plus the reference's own include/ESKF_LIO/Types.hpp read where it lies.
    It pins NOTHING numerically (the stand-ins compute nothing); it retires "never compiled"."""
    out = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-c", "-o", str(tmp_path / "call_sites.o"),
                          "-I" + os.path.join(ROOT, "tests", "compile_native", "stubs"), "-I" + os.path.join(ROOT, "include"),
                          "-I/root/reference/include", os.path.join(ROOT, "tests", "compile_native", "call_sites.cpp")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-4000:]
    syms = subprocess.run(["nm", "-C", str(tmp_path / "call_sites.o")], capture_output=True, text=True, check=True).stdout
    assert "ESKF_LIO::ICP::align" in syms and "vgicp_align" in syms and "vgicp_map_upsert" in syms


def test_copy_crew_survives_helpers_that_come_late(tmp_path):
    """The threads that copy a scan into page-locked memory take units of a job from one atomic word (job number and
    next unit).  A helper that was woken for a job which the caller has meanwhile finished alone finds the NEXT job's
    word there: it must leave without taking anything (compare-and-swap on the job number) — a blind increment took a
    unit away from everybody and the upload never finished (round 5: a soak that hung once per ~200 000 uploads;
    this stress reproduced it within a few thousand jobs).  tests/native/crew_stress.cpp: 150 000 tiny jobs whose SHAPE
    changes from job to job (round 5's advisor: a late helper checked its old ticket against the next job's larger unit
    count — jobs are closed by finish() now), every third one not announced to the helpers; every unit published once,
    every byte copied, no job stuck.  Then a helper that never returns from its copy: finish() reports it within its
    deadline (the module returns VGICP_ERR_TIMEOUT), the crew goes on alone, the late completion is dropped."""
    exe = tmp_path / "crew_stress"
    out = subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                          "-I" + os.path.join(ROOT, "eskf_lio_amd", "csrc"), "-I" + os.path.join(ROOT, "include"), "-o", str(exe),
                          os.path.join(ROOT, "tests", "native", "crew_stress.cpp")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    for helpers in ("1", "3"):
        run = subprocess.run([str(exe), "150000", helpers, "1"], capture_output=True, text=True, timeout=300)
        assert run.returncode == 0 and "ok 150000 jobs" in run.stdout, run.stdout[-500:] + run.stderr[-500:]
        assert "stuck helper: reported, crew went on alone, late completion dropped" in run.stdout, run.stdout[-500:]
    # the same under ThreadSanitizer (the sanitizers run on the CPU build only): no data race between the caller's set-up
    # of the next job and a helper still looking at the last one
    tsan = tmp_path / "crew_stress_tsan"
    out = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", "-D__HIP_PLATFORM_AMD__",
                          "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "eskf_lio_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
                          "-o", str(tsan), os.path.join(ROOT, "tests", "native", "crew_stress.cpp")], capture_output=True, text=True)
    if out.returncode != 0:
        pytest.skip("no ThreadSanitizer runtime for this compiler: " + out.stderr[-200:])
    run = subprocess.run([str(tsan), "20000", "2", "1"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "ok 20000 jobs" in run.stdout and "ThreadSanitizer" not in run.stderr, run.stderr[-3000:]


def test_dropin_resident_check_hash_on_the_cpu(tmp_path):
    """The drop-in classes read the DEVICE copy of a cloud only while a hash over every byte of the host cloud still
    matches (shim::ResidentCheck::FullHash, the default since round 6; reference: src/Registration.cpp:11 and
    src/LocalMap.cpp:45-58 always read the host cloud).  tests/native/shim_hash.cpp: the AVX2 lanes equal the plain
    ones, every single-bit edit / in-lane swap / rotation of a buffer changes the value, FullHash sees an edit of an
    element the 64-sample check (ResidentCheck::Sampled, the opt-in) misses."""
    exe = tmp_path / "shim_hash"
    out = subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-Wall", "-I" + os.path.join(ROOT, "include"), "-o", str(exe),
                          os.path.join(ROOT, "tests", "native", "shim_hash.cpp")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and run.stdout.strip() == "ok", run.stdout[-500:] + run.stderr[-500:]
    # the helper threads of the check (shim::HashCrew) under ThreadSanitizer: a job's set-up against helpers still
    # looking at the last one, chunks handed out once, sums published before finish() returns
    tsan = tmp_path / "shim_hash_tsan"
    out = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", "-I" + os.path.join(ROOT, "include"),
                          "-o", str(tsan), os.path.join(ROOT, "tests", "native", "shim_hash.cpp")], capture_output=True, text=True)
    if out.returncode != 0:
        pytest.skip("no ThreadSanitizer runtime for this compiler: " + out.stderr[-200:])
    run = subprocess.run([str(tsan), "400"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and run.stdout.strip() == "ok" and "ThreadSanitizer" not in run.stderr, run.stderr[-3000:]


def test_sort_plan_of_the_preparation_on_the_cpu(tmp_path):
    """Host logic of the hand-written sort (eskf_lio_amd/csrc/vgicp_sort.h: which groups of runs the merge levels take,
    how many launches that is, how much room the splitters need) for sizes up to 20 M pairs: tests/native/sort_plan.hip,
    compiled by hipcc (it cross-compiles without a GPU) and run on the CPU — no device call.  The sort itself is checked on
    the GPU against std::stable_sort (tests/native/sort_check.hip, test_gpu_parity.py)."""
    exe = tmp_path / "sort_plan"
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "--offload-arch=gfx950", "-o", str(exe),
                          os.path.join(ROOT, "tests", "native", "sort_plan.hip")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and run.stdout.startswith("ok"), run.stdout[-500:] + run.stderr[-500:]


def test_shadow_grid_of_the_dropin_map_on_the_cpu(tmp_path):
    """The drop-in LocalMap's defaults keep the grid on the device and a host-side SHADOW of it (a worker thread) for
    save() (include/eskf_lio_shim/LocalMap.hpp; reference src/LocalMap.cpp:10-76,156-167).  tests/native/shadow_stress.cpp
    stubs the C ABI (every call succeeds, nothing happens), so what runs is the class's host side alone: 120 frames with
    insertion and eviction through the worker thread against a host-authoritative map on the caller's thread — save()
    writes the same points; and the same under ThreadSanitizer."""
    src = os.path.join(ROOT, "tests", "native", "shadow_stress.cpp")
    exe = tmp_path / "shadow_stress"
    out = subprocess.run(["g++", "-O1", "-std=c++17", "-pthread", "-Wall", "-I" + os.path.join(ROOT, "include"), "-o", str(exe), src],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    run = subprocess.run([str(exe), "120", str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "ok 120 frames" in run.stdout, run.stdout[-800:] + run.stderr[-800:]
    tsan = tmp_path / "shadow_stress_tsan"
    out = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", "-I" + os.path.join(ROOT, "include"),
                          "-o", str(tsan), src], capture_output=True, text=True)
    if out.returncode != 0:
        pytest.skip("no ThreadSanitizer runtime for this compiler: " + out.stderr[-200:])
    run = subprocess.run([str(tsan), "40", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "ok 40 frames" in run.stdout and "ThreadSanitizer" not in run.stderr, run.stderr[-3000:]


def test_same_voxel_shortcut_implies_an_unchanged_key():
    """The persistent launch asks "is the point still inside last round's voxel?" before it makes a key
    (`same_voxel_coord`, eskf_lio_amd/csrc/vgicp_device_fn.h): r = fma(-k, h, x); yes iff 0 <= r and h - r > 2^-20 h.
    Restated here with exact rational arithmetic for the FMA (one rounding, like the instruction): on coordinates
    within a few ulps of cell faces, for several voxel sizes and keys up to 2^29, a yes must IMPLY
    floor(fl(x / h)) == k (reference: LocalMap::getVoxelIndex, src/LocalMap.cpp:114-118); a no is always allowed.
    The device function itself is pinned by the hook behind `vgicp_voxel_index` (-m gpu)."""
    from fractions import Fraction

    def same(x, k, h):
        r = float(Fraction(x) - Fraction(int(k)) * Fraction(h))  # fma(-k, h, x): exact, then ONE rounding
        return r >= 0.0 and (h - r) > 2.0 ** -20 * h

    rng = np.random.default_rng(11)
    yes = total = 0
    for h in (0.3, 0.1, 0.7, 1.0, 0.05, 2.5, 1.0 / 3.0):
        ks = np.concatenate([np.arange(-40, 40), rng.integers(-2**29, 2**29, size=150),
                             rng.integers(-70_000, 70_000, size=150)])
        for k in ks:
            base = float(k) * h
            xs = [base, base + 0.5 * h, base + h * (1 - 2.0 ** -19), base + h * (1 - 2.0 ** -22)]
            for ulps in (1, 2, 5, 17):
                up, dn = base, base
                for _ in range(ulps):
                    up, dn = np.nextafter(up, np.inf), np.nextafter(dn, -np.inf)
                xs += [float(up), float(dn)]
            for x in xs:
                true_key = int(np.floor(np.float64(x) / np.float64(h)))
                for cand in (true_key - 1, true_key, true_key + 1, int(k)):
                    total += 1
                    if same(x, cand, h):
                        yes += 1
                        assert cand == true_key, (x, h, cand, true_key)
    assert yes > total // 10  # the shortcut does say yes for ordinary points


def test_small_angle_threshold_of_the_series_exponential_is_the_references_branch():
    """se3_exp_device tests n2 < kSmallAngle2 where the reference tests sqrt(n2) < 1e-6 (src/Utils.cpp:47): the constant
    must be the smallest double whose (correctly rounded) square root reaches 1e-6."""
    import math
    import re
    import struct
    src = open(os.path.join(ROOT, "eskf_lio_amd", "csrc", "vgicp_kernels.hip")).read()
    m = re.search(r"constexpr double kSmallAngle2 = (0x[0-9a-fA-F.]+p[-+]?\d+);", src)
    assert m, "kSmallAngle2 not found"
    t = float.fromhex(m.group(1))
    below = struct.unpack("<d", struct.pack("<q", struct.unpack("<q", struct.pack("<d", t))[0] - 1))[0]
    assert math.sqrt(t) >= 1e-6 and math.sqrt(below) < 1e-6
