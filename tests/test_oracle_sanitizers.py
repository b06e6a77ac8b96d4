"""CPU: the oracle's C code under AddressSanitizer + UndefinedBehaviorSanitizer.

The oracle is the parity anchor on the GPU box; a checker that writes out of bounds can corrupt the
thing it checks (round 4: mask mode with ``H == 0, exclude_last`` wrote a row into a zero-byte buffer).
This test builds ``oracle/*.c`` with ``-fsanitize=address,undefined`` and replays a randomised sweep
of every entry point -- degenerate shapes included -- through the Python wrappers in a child process
started under ``LD_PRELOAD=libasan`` (``tests/fuzz/oracle_sweep.py``).  A report aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    path = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


@pytest.mark.parametrize("seed", [501, 502])
def test_oracle_sweep_is_clean_under_asan_and_ubsan(seed):
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("gcc's libasan is not installed")
    sys.path.insert(0, ROOT)
    import oracle

    oracle.build(sanitize=True)
    env = dict(os.environ)
    env.update(
        LD_PRELOAD=asan,
        ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
        UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
        PDT_ORACLE_SANITIZE="1",
    )
    r = subprocess.run(
        [sys.executable, os.path.join(ROOT, "tests", "fuzz", "oracle_sweep.py"), str(seed), "6000"],
        env=env, capture_output=True, text=True, timeout=600,
    )  # fmt: skip
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "no sanitizer report" in r.stdout
