"""AddressSanitizer + UndefinedBehaviorSanitizer run of the CPU oracle (sanitizers on the CPU build only —
GPU ASan is not available on this pool)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_oracle_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_san")
    cmd = ["gcc", "-O1", "-g", "-std=gnu11", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", "-o", exe, os.path.join(ROOT, "tests", "sanitize_oracle_main.c"),
           os.path.join(ROOT, "oracle", "adn_oracle.c"), "-lm"]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("libasan/libubsan not installed")
    assert build.returncode == 0, build.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "sanitized oracle run ok" in run.stdout
