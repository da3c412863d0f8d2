"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/cae_hip.h declares,
and its geometry-only entry points agree with the reference-generated fixtures.  No compute calls."""
import os
import re

import numpy as np
import pytest

from helpers import MODEL_CASES, GoldenCase
from cae_tools_amd import _lib
from cae_tools_amd.engine import EnginePlan

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = set()
    for (header, prefix) in (("cae_hip.h", "cae_"), ("cae_unet.h", "unet_"), ("cae_vae.h", "vae_"), ("cae_linear.h", "lin_")):
        with open(os.path.join(ROOT, "include", header)) as f:
            text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)     # prose in comments mentions call-like names
        found = set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text))
        assert len(found) >= 16, header
        declared |= found
    declared -= {"cae_engine", "cae_status", "unet_engine", "vae_engine", "lin_engine"}
    for name in sorted(declared):
        assert hasattr(lib, name), f"libcae_hip.so does not export {name}"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.cae_abi_version() == 1


@pytest.mark.parametrize("name", MODEL_CASES)
def test_tensor_table_matches_reference_state_dict(name):
    case = GoldenCase(name)
    plan = EnginePlan(case.spec, case.meta["fc"], case.meta["latent"], max_batch=8)
    ref = {}
    for side in ("enc", "dec"):
        for k, v in case.group(f"init/{side}/").items():
            if not k.endswith("num_batches_tracked"):
                ref[f"{side}/{k}"] = tuple(v.shape)
    got = {n: shape for n, (arena, off, numel, shape) in plan.tensors.items()}
    assert got == ref
    # reference state_dict order == arena order
    assert list(got) == list(ref)
    nparam = sum(int(np.prod(s)) for n, s in ref.items() if "running_" not in n)
    assert nparam <= plan.n_param <= nparam + 4 * len(ref)
    for n, (arena, off, numel, shape) in plan.tensors.items():
        assert off % 4 == 0 and arena == (1 if "running_" in n else 0)
    assert plan.workspace_bytes > 0
    plan.close()


def test_bad_geometry_is_rejected():
    case = GoldenCase("cfg1_b3")
    spec = {"input_layers": [dict(l) for l in case.spec["input_layers"]],
            "output_layers": [dict(l) for l in case.spec["output_layers"]]}
    spec["output_layers"][2]["output_dimensions"] = [8, 30, 31]
    with pytest.raises(_lib.CaeError, match="decoder output size"):
        EnginePlan(spec, 16, 4, 8)
    with pytest.raises(_lib.CaeError):
        EnginePlan(case.spec, 0, 4, 8)


def test_trace_ranges_are_callable_without_a_profiler():
    """cae_trace_range_push / pop (roctx, bound at run time): balanced calls, 1 when a roctx library was found, 0 otherwise -
    and never an error: tracing must not be able to break a run"""
    from cae_tools_amd import _lib
    lib = _lib.load()
    pushed = lib.cae_trace_range_push(b"cae_tools_amd.test")
    assert pushed in (0, 1)
    assert lib.cae_trace_range_pop() == pushed
    assert lib.cae_trace_range_push(None) == 0
