"""The host side of libcae_hip under AddressSanitizer + LeakSanitizer (SURVEY.md §5 "sanitizers"): tools/asan_host_check.sh builds
every .hip source with -fsanitize=address (device code compiled as usual, never launched: GPU ASAN is not available on this
pool) and runs tests/asan/plan_check.cpp on the CPU - engine plans, tensor tables, error paths and destruction of the ConvAE
engine, the var engine (trunk mode) and the UNET engine.  The build takes ~4 minutes, so the test runs only when asked for:
CAE_ASAN=1 python -m pytest tests/test_asan_host_cpu.py  (last run: clean)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(os.environ.get("CAE_ASAN") != "1", reason="set CAE_ASAN=1 (a ~4 minute sanitizer build)")
def test_host_side_is_clean_under_address_sanitizer():
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan_host_check.sh")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         timeout=1800)
    text = out.stdout.decode(errors="replace")
    assert out.returncode == 0 and "host-side plan checks clean" in text and "ERROR: AddressSanitizer" not in text, text[-3000:]
