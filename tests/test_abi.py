"""CPU: the C-ABI library loads, exports every symbol include/ishara_hip.h declares, and its
parameter layout equals the oracle's inventory (no compute: there is no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from ishara_amd import _lib, make_config
from ishara_amd.model import Model, _keras_init
from oracle import ishara_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "ishara_hip.h")).read()
    declared = set(re.findall(r"\b(ishara_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"ishara_model", "ishara_config", "ishara_stream"}
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(lib, name), f"libishara_hip.so does not export {name}"
    assert declared == set(_lib.SIGNATURES), "ctypes SIGNATURES out of sync with the header"


def test_config_struct_matches_header():
    hdr = open(os.path.join(ROOT, "include", "ishara_hip.h")).read()
    body = re.search(r"typedef struct ishara_config \{(.*?)\} ishara_config;", hdr, re.S).group(1)
    fields = re.findall(r"(?:int32_t|float)\s+([a-z_]+)(?:\[\d+\])?;", body)
    assert fields == [f[0] for f in _lib.Config._fields_]


@pytest.mark.parametrize("kw", [
    dict(dim=64, num_conv_squeeze_blocks=1, num_conv_conform_blocks=1),
    dict(dim=256, input_shape=(384, 224)),
    dict(dim=256, num_conv_squeeze_blocks=4, num_conv_conform_blocks=4, num_conv_per_block=0, squeeze_expansion=4, conformer_expansion=2, top_dim=256),
    dict(dim=512, num_conv_squeeze_blocks=6, num_conv_conform_blocks=6, input_shape=(512, 224)),
])
def test_param_layout_equals_oracle_inventory(kw):
    m = Model(make_config(**kw, max_batch=2), device=None)
    ok = {k: v for k, v in kw.items()}
    specs = O.param_specs(O.Config(**ok))
    assert [(n, tuple(s), t) for n, s, _, t in specs] == [(n, s, t) for n, s, _, t in m.entries]
    tot, tr, nt = O.count_params(O.Config(**ok))
    assert (m.n_total, m.n_train) == (tot, tr)
    # trainable entries occupy [0, n_train) contiguously (one all-reduce bucket), BN moving stats after
    offs = sorted((o, int(np.prod(s)), t) for _, s, o, t in m.entries)
    pos = 0
    for o, n, t in offs:
        assert o == pos and (t == (o < m.n_train))
        pos += n
    assert pos == m.n_total
    assert int(m._lib.ishara_workspace_bytes(m._h)) > 0


def test_keras_initialisers_match_oracle():
    cfg = O.Config(dim=64, num_conv_squeeze_blocks=1, num_conv_conform_blocks=1)
    ref = O.init_params(cfg, 5)
    g = np.random.default_rng(5)
    for n, s, _, _ in O.param_specs(cfg):
        assert np.array_equal(_keras_init(n, s, g), ref[n]), n


def test_create_rejects_bad_configs(lib):
    h = C.c_void_p()
    for bad in [dict(dim=60), dict(dim=256, num_heads=5), dict(input_shape=(100, 276)), dict(kernel_sizes=[33]),
                dict(transformer_kernel_size=14), dict(num_classes=100), dict(dim=768, num_heads=8)]:
        rc = lib.ishara_create(C.byref(make_config(**bad)), C.byref(h))
        assert rc != 0 and lib.ishara_last_error(), bad


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.IsharaError):
        Model(make_config(dim=64, num_conv_squeeze_blocks=1, num_conv_conform_blocks=1), device="cuda:0")


def test_product_path_does_not_import_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "ishara_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


@pytest.mark.parametrize("guard", ["0", "1"])
def test_workspace_plan_has_no_overlap(guard, monkeypatch):
    """Host-side audit of the bump-allocated workspace: aligned, in bounds, disjoint — with and without guard zones."""
    monkeypatch.setenv("ISHARA_WS_GUARD", guard)
    for kw in (dict(dim=64, num_conv_squeeze_blocks=1, num_conv_conform_blocks=1, max_batch=1, dtype="f32"),
               dict(dim=256, input_shape=(384, 224), max_batch=256),
               dict(dim=512, num_conv_squeeze_blocks=6, num_conv_conform_blocks=6, input_shape=(512, 224), max_batch=8)):
        m = Model(make_config(**kw), device=None)
        n = m._lib.ishara_workspace_plan_check(m._h)
        assert n > 100, _lib.load().ishara_last_error()


def test_c_caller_walks_the_abi(tmp_path):
    """include/ishara_hip.h compiles as C and a plain C program (gcc + dlopen, no torch, no C++) creates the three families, reads the
    parameter layout and audits the workspace plan through the exported symbols."""
    import subprocess
    exe = str(tmp_path / "abi_walk")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_walk.c"), "-ldl", "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, _lib.LIB_PATH], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 3 and "7577784 parameters" in lines[0] and lines[2].startswith("family 2")
