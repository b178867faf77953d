"""CPU: the container parsers (convert-utility/container.cpp: WAV / CAF sniffing, the CAF packet table, MP4 / M4A boxes, the
sample description, the cookie wrappers) take untrusted files — 3 000 mutated, truncated and spliced inputs through an
AddressSanitizer + UBSan build, in a subprocess (the sanitizer runtime has to be preloaded into the interpreter)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parsers_survive_mutated_files_under_asan():
    cu = os.path.join(ROOT, "convert-utility")
    subprocess.check_call(["make", "-C", cu, "libcontainer_asan.so"], stdout=subprocess.DEVNULL)
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "container_fuzz_worker.py"), "3000"], capture_output=True,
                       text=True, timeout=600, env=env)
    assert p.returncode == 0 and p.stdout.strip().startswith("OK"), (p.stdout[-800:], p.stderr[-3000:])
