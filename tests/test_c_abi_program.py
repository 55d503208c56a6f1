"""include/bayesfilt.h used from C: tests/c/kalman_from_c.c (plain C99, hipMalloc'd buffers, no Python in the loop)
is compiled with gcc against the header and the built library, run as its own process, and its printed posterior is
compared with the oracle.  Also: the ctypes stub of INTEGRATION.md is executed as written and must agree with the
library's struct layouts (bf_abi_check)."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32 = np.float32


def _build(tmp_path):
    exe = tmp_path / "kalman_from_c"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                    "-I", "/opt/rocm/include", os.path.join(ROOT, "tests", "c", "kalman_from_c.c"),
                    "-L", os.path.join(ROOT, "bayesianfiltering_amd"), "-lbayesfilt_hip", "-L", "/opt/rocm/lib", "-lamdhip64",
                    "-lm", "-o", str(exe)], check=True)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "bayesianfiltering_amd") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    return exe, env


def test_header_compiles_and_links_from_c(tmp_path):
    """gcc -std=c99 -Wall -Werror on a program that includes only bayesfilt.h + the HIP C API; without a GPU it stops
    after the ABI guard (which must accept the header's own sizes and refuse a six-stream bf_out_desc)."""
    exe, env = _build(tmp_path)
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present: the gpu-marked test runs the program")
    out = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "version 210" in out.stdout and "no-gpu" in out.stdout


def test_abi_guard_names_the_mismatching_struct():
    import ctypes as C
    from bayesianfiltering_amd import _lib
    lib = _lib.load()
    sizes = [C.sizeof(_lib.bf_out_desc), C.sizeof(_lib.bf_lgssm), C.sizeof(_lib.bf_model), C.sizeof(_lib.bf_bpf_model),
             C.sizeof(_lib.bf_bpf_out)]
    assert lib.bf_abi_check(_lib.HEADER_VERSION, *sizes) == _lib.BF_OK
    six_streams = [6 * C.sizeof(_lib.bf_stream)] + sizes[1:]            # round 1's stale documentation stub
    assert lib.bf_abi_check(_lib.HEADER_VERSION, *six_streams) == _lib.BF_EINVAL
    assert "bf_out_desc" in lib.bf_last_error().decode()
    assert lib.bf_abi_check(100, *sizes) == _lib.BF_EINVAL              # a binding written against the 0.1 header
    assert "version" in lib.bf_last_error().decode()


def _integration_stub_namespace():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n# gaussfiltax/_hip.py.*?\n(.*?)```", text, flags=re.S).group(1)
    block = block.replace('C.CDLL("libbayesfilt_hip.so")', 'C.CDLL(%r)' % os.path.join(ROOT, "bayesianfiltering_amd", "libbayesfilt_hip.so"))
    ns = {}
    exec(compile(block, "INTEGRATION.md", "exec"), ns)
    return ns


def test_integration_md_stub_matches_the_library():
    """The stub a maintainer would copy out of INTEGRATION.md: executes, passes the library's ABI guard, and its
    structs have the sizes of _lib.py's (eight streams in bf_out_desc)."""
    import ctypes as C
    from bayesianfiltering_amd import _lib
    ns = _integration_stub_namespace()
    for name in ("bf_stream", "bf_out_desc", "bf_carry", "bf_lgssm"):
        assert C.sizeof(ns[name]) == C.sizeof(getattr(_lib, name)), name
    assert [f[0] for f in ns["bf_out_desc"]._fields_] == [f[0] for f in _lib.bf_out_desc._fields_]


@pytest.mark.gpu
def test_c_program_matches_oracle(tmp_path):
    from tests import common as cm
    exe, env = _build(tmp_path)
    out = subprocess.run([str(exe), "3", "24"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    vals = {}
    for line in out.stdout.splitlines():
        parts = line.split()
        if parts and parts[0] in ("emissions", "weights", "loglik", "means", "predicted_means", "covariances", "predicted_covariances"):
            vals[parts[0]] = np.array([float(v) for v in parts[1:]], F32)
    B, T, n, m = 3, 24, 4, 2
    a = cm.cv_model_arrays()
    ys = vals["emissions"].reshape(B, T, m)
    ref = cm.oracle_kalman_batch(a, ys, np.zeros((B, n), F32))
    shapes = {"weights": (B, 1, T), "loglik": (B, 1, T), "means": (B, 1, T, n), "predicted_means": (B, 1, T, n),
              "covariances": (B, 1, T, n, n), "predicted_covariances": (B, 1, T, n, n)}
    for k, shp in shapes.items():
        assert cm.rel_err(vals[k].reshape(shp), ref[k]) < 1e-5, k


@pytest.mark.gpu
def test_integration_md_stub_runs_the_filter():
    """kalman_scan() of the INTEGRATION.md stub, as written, against the oracle."""
    import torch
    from tests import common as cm
    ns = _integration_stub_namespace()
    a = cm.cv_model_arrays()
    B, T = 5, 40
    ys = cm.simulate_batch(a, B, T, seed=3)
    y = torch.as_tensor(ys, device="cuda")
    m0 = torch.zeros((B, 1, 4), device="cuda")
    P0 = torch.eye(4, device="cuda").reshape(1, 1, 4, 4).expand(B, 1, 4, 4).contiguous()
    outs = ns["kalman_scan"](a["A"], a["G"], a["H"], a["D"], a["q0"], a["r0"], a["Q"], a["R"], y, m0, P0)
    ref = cm.oracle_kalman_batch(a, ys, np.zeros((B, 4), F32))
    for k, rk in (("means", "means"), ("covs", "covariances"), ("pred_means", "predicted_means"), ("pred_covs", "predicted_covariances")):
        assert cm.rel_err(outs[k].cpu().numpy(), ref[rk]) < 1e-5, k
