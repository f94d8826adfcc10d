"""CPU: the two forms of the product's mpr_step (mujoco_amd/csrc/physics_collide.h — default: one branch per portal
phase, -DMPR_BRANCHY; the default: one instruction stream with phase-selected operands) are the SAME function.  Both are
__host__ __device__; tests/host_mpr/mpr_host.hip wraps them, compiled for the host, around an exhaustive support
search, and this test runs thousands of posed hull pairs through both and compares every call's return code, packed
state word and next direction (a trace hash), the support-call and iteration counts, and depth / normal / position, bit for
bit.  (Host arithmetic is not the device's — no v_rsq_f64 — so this pins the LOGIC of the unified form; the device-vs-oracle
contact tests pin the arithmetic of whichever form the library is built with.)"""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def forms(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    d = tmp_path_factory.mktemp("mpr")
    libs = {}
    for name, flag in (("branchy", ["-DMPR_BRANCHY"]), ("unified", [])):
        out = str(d / f"libmpr_{name}.so")
        subprocess.run([HIPCC, "-O2", "-std=c++17", "--cuda-host-only", "-fPIC", "-shared", "-Wno-unused-function", *flag,
                        "-o", out, os.path.join(ROOT, "tests", "host_mpr", "mpr_host.hip")], check=True, capture_output=True)
        L = C.CDLL(out)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.mpr_host_pair.argtypes = [dp, C.c_int, dp, C.c_int, dp, C.c_double, C.c_double, C.c_int, dp, ip, ip, C.POINTER(C.c_uint64)]
        L.mpr_host_pair.restype = C.c_int
        libs[name] = L
    yield libs
    shutil.rmtree(d, ignore_errors=True)


def _run(L, va, vb, pose, margin, tol=1e-6, maxit=50):
    out = np.zeros(7)
    calls, iters, trace = C.c_int(), C.c_int(), C.c_uint64()
    dp = C.POINTER(C.c_double)
    r = L.mpr_host_pair(va.ctypes.data_as(dp), len(va), vb.ctypes.data_as(dp), len(vb), pose.ctypes.data_as(dp), margin, tol, maxit,
                        out.ctypes.data_as(dp), C.byref(calls), C.byref(iters), C.byref(trace))
    return r, out, calls.value, iters.value, trace.value


def _rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _polytope(rng, n):
    p = rng.normal(size=(n, 3))
    p /= np.linalg.norm(p, axis=1)[:, None]
    return np.ascontiguousarray(p * rng.uniform(0.01, 0.04, size=3))


def _hulls_of_the_hand():
    from mujoco_amd import Model
    m = Model.load(os.path.join(ROOT, "tests", "golden", "dexterous_model.npz"))
    hv = np.asarray(m.hull_vert, dtype=np.float64).reshape(-1, 3)
    return [np.ascontiguousarray(hv[a:a + n]) for a, n in zip(m.mesh_vertadr, m.mesh_vertnum) if n >= 4]


def test_unified_mpr_step_is_the_branchy_one_bit_for_bit(forms):
    rng = np.random.default_rng(20261112)
    hulls = _hulls_of_the_hand() + [_polytope(rng, n) for n in (4, 6, 8, 12, 20, 40, 80)]
    cube = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64) * 0.02
    hulls.append(np.ascontiguousarray(cube))  # flat faces and parallel edges: the degenerate tests of the portal
    n_pairs, hits, misses, special, max_calls = 0, 0, 0, 0, 0
    for trial in range(6000):
        va, vb = hulls[rng.integers(len(hulls))], hulls[rng.integers(len(hulls))]
        ra, rb = np.abs(va).max(), np.abs(vb).max()
        Ra, Rb = _rot(rng), _rot(rng)
        if trial % 40 == 0:
            Ra = Rb = np.eye(3)                      # axis-aligned: exact ties in the sign tests
        u = rng.normal(size=3)
        u /= np.linalg.norm(u)
        gap = rng.choice([0.0, 0.3, 0.7, 1.0, 1.3, 2.5]) * (ra + rb)   # from concentric to clearly apart
        xa = rng.normal(size=3) * 0.05
        xb = xa + u * gap
        if trial % 97 == 0:
            xb = xa.copy()                            # coincident centres: mpr_begin's epsilon direction
        pose = np.ascontiguousarray(np.concatenate([xa, Ra.ravel(), xb, Rb.ravel()]))
        margin = float(rng.choice([0.0, 0.001, 0.004]))
        a = _run(forms["branchy"], va, vb, pose, margin)
        b = _run(forms["unified"], va, vb, pose, margin)
        assert a[0] == b[0] and a[2] == b[2] and a[3] == b[3], (trial, a, b)
        assert a[4] == b[4], (trial, "the two forms took different paths", a, b)
        assert a[1].tobytes() == b[1].tobytes(), (trial, a[1], b[1])
        n_pairs += 1
        hits += a[0] == 1
        misses += a[0] == 2
        special += a[0] == 1 and a[3] == 0
        max_calls = max(max_calls, a[2])
    # the sample exercised every exit: portal hits with refinement, early and late misses
    assert hits > 1000 and misses > 1000 and max_calls >= 8, (hits, misses, special, max_calls)
