"""The C-ABI library loads, exports every symbol include/bayesfilt.h declares, and the product
fails loudly (never falls back to a CPU path) when no MI355X is present."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "bayesfilt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bf_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    names = _declared_functions()
    for required in ("bf_version", "bf_last_error", "bf_device_count", "bf_kalman_filter_f32", "bf_bytes_per_step"):
        assert required in names


def test_library_exports_every_declared_symbol():
    from bayesianfiltering_amd import _lib
    lib = _lib.load()
    for name in _declared_functions():
        assert hasattr(lib, name), f"{name} declared in include/bayesfilt.h but not exported"
        assert name in _lib.SYMBOLS, f"{name} has no ctypes prototype in _lib.SYMBOLS"
    assert lib.bf_version() >= 100


def test_bytes_per_step_formula():
    from bayesianfiltering_amd import _lib
    lib = _lib.load()
    assert lib.bf_bytes_per_step(4, 2, 1, None) == 172          # 4m + 4K(1 + 2n + 2n^2)
    assert lib.bf_bytes_per_step(8, 4, 32, None) == 18576
    assert lib.bf_bytes_per_step(64, 32, 1, None) == 33412


def test_struct_layouts_match_header():
    import ctypes as C
    from bayesianfiltering_amd import _lib
    assert C.sizeof(_lib.bf_stream) == 40 and C.sizeof(_lib.bf_cstream) == 40
    assert C.sizeof(_lib.bf_out_desc) == 320 and C.sizeof(_lib.bf_carry) == 48
    assert C.sizeof(_lib.bf_lgssm) == 16 + 8 * 8 + 8


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_gpu_means_loud_failure_not_fallback():
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import _lib
    from tests import common as cm
    a = cm.cv_model_arrays()
    with pytest.raises(_lib.BayesFiltError) as e:
        bfa.kalman_filter(cm.product_params(a), np.zeros((2, 8, 2), np.float32))
    assert e.value.code == _lib.BF_ENOGPU


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "bayesianfiltering_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "oracle/" not in src or f.endswith(".md"), f


def test_registry_functions_match_oracle_models():
    """Host-callable registry functions (bayesianfiltering_amd.nonlinearities) agree with the oracle's models."""
    from bayesianfiltering_amd import nonlinearities as nl
    from oracle import models as om
    rng = np.random.default_rng(0)
    pairs = [(nl.lorenz96(8), om.Lorenz96(8), 8, 8, 0.0), (nl.lorenz96(8, mode="as_written"), om.Lorenz96(8, mode="as_written"), 8, 8, 0.0),
             (nl.lorenz63(), om.Lorenz63(), 3, 3, 0.0), (nl.maneuver_bot(), om.ManeuverBOT(), 4, 2, 1.0),
             (nl.maneuver_bot(), om.ManeuverBOT(), 4, 2, 2.0), (nl.bearing_range(), om.BearingRange(), 4, 2, 0.0), (nl.bearing(), om.Bearing(), 4, 1, 0.0),
             (nl.sine(3), om.Sine(3), 3, 3, 0.0), (nl.quadratic(3, 0.5), om.Quadratic(3, 0.5), 3, 1, 0.0),
             (nl.growth(), om.Growth(), 1, 1, 0.3), (nl.stoch_vol(3), om.StochVol(3), 3, 3, 1.0),
             (nl.pick_even(8), om.PickEven(8), 8, 4, 0.0)]
    for f, g, n, nw, u in pairs:
        x = rng.normal(size=n).astype(np.float32); w = (0.1 * rng.normal(size=nw)).astype(np.float32)
        assert np.allclose(f(x, w, np.float32([u])), g.value(x, w, np.float32([u])), rtol=1e-5, atol=1e-6), f.name


def test_struct_sizes_agree_with_a_c_compiler(tmp_path):
    """Every struct of include/bayesfilt.h, as laid out by gcc, against its ctypes mirror in _lib.py."""
    import ctypes as C
    import subprocess
    from bayesianfiltering_amd import _lib
    names = ["bf_stream", "bf_cstream", "bf_out_desc", "bf_carry", "bf_lgssm", "bf_model", "bf_ukf_params", "bf_bpf_model",
             "bf_bpf_carry", "bf_bpf_out"]
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "bayesfilt.h"\nint main(void) {\n'
                   + "".join(f'  printf("{n} %zu\\n", sizeof({n}));\n' for n in names) + "  return 0;\n}\n")
    exe = tmp_path / "sizes"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    sizes = dict(line.split() for line in out.strip().splitlines())
    for n in names:
        assert int(sizes[n]) == C.sizeof(getattr(_lib, n)), (n, sizes[n], C.sizeof(getattr(_lib, n)))


def test_user_model_source_compiles_without_a_gpu():
    """hiprtc needs no device to COMPILE: a well-formed source gets as far as loading the code object (BF_EHIP without a
    GPU, BF_OK with one), a malformed one is refused with the compiler's message."""
    import ctypes as C
    from bayesianfiltering_amd import _lib
    lib = _lib.load()
    good = b"template <class T> __device__ void dynamics(const T* x, const T* q, T u, const float* th, T* out) { out[0] = sin(x[0]) * th[0] + q[0]; }"
    h = C.c_void_p()
    rc = lib.bf_user_model_create(good, None, 1, 1, 1, 1, C.byref(h))
    assert rc in (_lib.BF_OK, _lib.BF_EHIP, _lib.BF_ENOGPU), lib.bf_last_error()
    if rc != _lib.BF_OK:
        assert b"loading the compiled model" in lib.bf_last_error()
    bad = b"template <class T> __device__ void dynamics(const T* x, const T* q, T u, const float* th, T* out) { out[0] = undefined_fn(x[0]); }"
    assert lib.bf_user_model_create(bad, None, 1, 1, 1, 1, C.byref(h)) == _lib.BF_EINVAL
    assert b"undefined_fn" in lib.bf_last_error()
    assert lib.bf_user_model_create(None, None, 1, 1, 1, 1, C.byref(h)) == _lib.BF_EINVAL


def test_missing_rccl_or_hiprtc_is_reported_not_fatal(tmp_path):
    """A loader candidate that cannot be opened must end in the documented BF_EUNSUPPORTED ('... is not available') with
    dlerror()'s text, never in a crash (dlerror() clears its message: calling it twice handed NULL to std::string).  Runs
    in a child process: the loaders cache their result per process, and $BAYESFILT_*_LIB must be set before first use."""
    import subprocess
    import sys
    code = f"""
import ctypes as C, sys
sys.path.insert(0, {ROOT!r})
from bayesianfiltering_amd import _lib
lib = _lib.load()
buf = (C.c_char * 64)()
rc = lib.bf_allgather_summaries(C.cast(buf, C.c_void_p), C.cast(buf, C.c_void_p), 16, C.c_void_p(1), None)
assert rc == _lib.BF_EUNSUPPORTED, (rc, lib.bf_last_error())
assert b"RCCL is not available" in lib.bf_last_error() and b"no_such_rccl" in lib.bf_last_error(), lib.bf_last_error()
src = b"template <class T> __device__ void dynamics(const T* x, const T* q, T u, const float* th, T* out) {{ out[0] = x[0] * 0.25f + q[0]; }}"
h = C.c_void_p()
rc = lib.bf_user_model_create(src, None, 1, 1, 1, 1, C.byref(h))
assert rc == _lib.BF_EUNSUPPORTED, (rc, lib.bf_last_error())
assert b"hiprtc is not available" in lib.bf_last_error() and b"no_such_hiprtc" in lib.bf_last_error(), lib.bf_last_error()
print("ok")
"""
    env = dict(os.environ, BAYESFILT_RCCL_LIB=str(tmp_path / "no_such_rccl.so"), BAYESFILT_HIPRTC_LIB=str(tmp_path / "no_such_hiprtc.so"),
               BAYESFILT_CACHE_DIR=str(tmp_path / "cache"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_jit_cache_survives_a_corrupt_file_and_concurrent_writers(tmp_path):
    """The disk cache of compiled user models: several processes that miss together each write their own temporary and
    rename it (no truncated file is ever visible under the final name), and a cache file that does not load is deleted and
    rebuilt instead of failing for good.  Without a GPU the load step reports BF_ENOGPU; the cache logic is the same."""
    import subprocess
    import sys
    cache = tmp_path / "cache"
    code = f"""
import ctypes as C, sys, os
sys.path.insert(0, {ROOT!r})
from bayesianfiltering_amd import _lib
lib = _lib.load()
src = b"template <class T> __device__ void dynamics(const T* x, const T* q, T u, const float* th, T* out) {{ out[0] = x[0] * 0.5f + q[0]; }}"
h = C.c_void_p()
rc = lib.bf_user_model_create(src, None, 1, 1, 1, 1, C.byref(h))
assert rc in (_lib.BF_OK, _lib.BF_ENOGPU), (rc, lib.bf_last_error())
print("rc", rc)
"""
    env = dict(os.environ, BAYESFILT_CACHE_DIR=str(cache))
    procs = [subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(3)]
    for pr in procs:
        out, err = pr.communicate(timeout=600)
        assert pr.returncode == 0, (out, err)
    files = sorted(os.listdir(cache))
    assert len(files) == 1 and files[0].endswith(".co") and not any(f.endswith(".tmp") for f in files), files
    good = (cache / files[0]).read_bytes()
    assert good[:4] == b"\x7fELF" or good[:8] == b"__CLANG_"                  # a code object / offload bundle, whole
    if os.path.exists("/dev/kfd"):                                               # with a GPU: a corrupt file is replaced
        (cache / files[0]).write_bytes(good[: len(good) // 3])
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "rc 0" in r.stdout, (r.stdout, r.stderr)
        assert (cache / files[0]).read_bytes() == good
